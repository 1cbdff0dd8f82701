"""``VGGPerceptualLoss`` on MI355X (mirror of reference ``src/lib/loss.py:17-58``).

Same call contract -- ``VGGPerceptualLoss(resize=True)(input, target) -> 0-d tensor`` on NCHW images
in [0, 1] -- and the same arithmetic: channel repeat for non-RGB, ImageNet normalisation, optional
bilinear 224x224 (align_corners=False), the four VGG16 slices ``features[0:4], [4:9], [9:16],
[16:23]`` (3x3 pad-1 convs WITH bias + ReLU, 2x2 max-pool), loss = sum over slices of mean |x - y|.

MI355X-first: input and target run as ONE batch of 2B through the implicit-GEMM conv kernel
(bias + ReLU fused in its epilogue), conv1_1's 3 input channels are fed as 3x3 patches (normalise
fused into the patch kernel) so that it is a 1x1 conv with K = 32, and each slice's L1 is a
two-level fp64 reduction.  Forward only, like the reference (its VGG is frozen and never
back-propagated through in-tree).

The reference takes the weights from ``torchvision.models.vgg16(pretrained=True)`` (a download);
here they come from a state_dict with the reference module's own key names
(``blocks.<slice>.<features index>.{weight,bias}``) or torchvision's (``features.<index>.*``).
Parity: fixture G10 (tests/golden/g10_vgg.npz) is the reference's own ``VGGPerceptualLoss`` run on a
torchvision-free VGG16-D layer list with synthetic weights -- its slicing, channel repeat, normalisation,
resize and L1 lines are pinned; torchvision's layer list itself (third party, absent) stays restated.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import capi

# (slice, features index, cin, cout); 'p' marks a 2x2 max-pool in front of the conv
VGG16_LAYOUT = [
    (0, 0, 3, 64, False), (0, 2, 64, 64, False),
    (1, 5, 64, 128, True), (1, 7, 128, 128, False),
    (2, 10, 128, 256, True), (2, 12, 256, 256, False), (2, 14, 256, 256, False),
    (3, 17, 256, 512, True), (3, 19, 512, 512, False), (3, 21, 512, 512, False),
]
SLICE_END = {1: 0, 3: 1, 6: 2, 9: 3}  # conv position (0-based) -> slice that ends after it


def _dtype_code(name: str) -> int:
    return capi.BF16 if name.lower() in ("bf16", "bfloat16") else capi.F32


class _Plan:
    """Static launch list for one (2B, H, W) shape."""

    def __init__(self, mod: "VGGPerceptualLoss", B2: int, H: int, W: int, dev):
        self.ops = []
        self.keep = []
        dt, esz = mod.dtype, (2 if mod.dtype == capi.BF16 else 4)
        tdt = torch.bfloat16 if dt == capi.BF16 else torch.float32
        self.img = torch.zeros(B2, 3, H, W, device=dev)
        # weights in kernel layout + table
        nconv = len(VGG16_LAYOUT)
        tab = (capi.WPrep * nconv)()
        src = fwd = blk = 0
        offs = []
        for i, (_, _, ci, co, _) in enumerate(VGG16_LAYOUT):
            patch = i == 0
            cip, kk = (32, 1) if patch else (ci, 9)
            e = tab[i]
            e.src_off, e.fwd_off, e.bwd_off = src, fwd, -1
            e.Co, e.Ci, e.ks, e.Cip, e.patch, e.blk0 = co, ci, 3, cip, int(patch), blk
            offs.append(fwd)
            src += co * ci * 9
            fwd += co * kk * cip
            blk += math.ceil(co * ci * 9 / 1024)
        self.wk = torch.zeros(fwd, dtype=tdt, device=dev)
        self.wtab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).clone().to(dev)
        self.wblocks, self.nconv = blk, nconv
        self.partial = torch.zeros(4, 1024, dtype=torch.float64, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)

        def act(b, h, w, c):
            t = torch.empty(b * h * w * c * esz, dtype=torch.uint8, device=dev)
            self.keep.append(t)
            return t

        x = act(B2, H, W, 32)
        self.ops.append(("stl_patch3x3", (dt, self.img.data_ptr(), x.data_ptr(), B2, H, W, 1, mod.mean.data_ptr(), mod.std.data_ptr())))
        h, w, c = H, W, 32
        for i, (_, _, ci, co, pool) in enumerate(VGG16_LAYOUT):
            if pool:
                y = act(B2, h // 2, w // 2, c)
                self.ops.append(("stl_maxpool2x2", (dt, x.data_ptr(), y.data_ptr(), B2, h, w, c)))
                x, h, w = y, h // 2, w // 2
            p = capi.Conv()
            p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co = dt, B2, h, w, c, h, w, co
            p.ks, p.stride, p.shape = (1 if i == 0 else 3), 1, -1
            p.src.x, p.src.mode = x.data_ptr(), capi.SRC_PLAIN
            y = act(B2, h, w, co)
            p.w = self.wk.data_ptr() + offs[i] * esz
            p.out, p.bias, p.out_relu = y.data_ptr(), mod.bias_flat.data_ptr() + 4 * mod.bias_off[i], 1
            capi.call("stl_conv_plan", C.byref(p))
            self.keep.append(p)
            self.ops.append(("stl_conv_forward", (C.byref(p),)))
            x, c = y, co
            if i in SLICE_END:
                s = SLICE_END[i]
                half = (B2 // 2) * h * w * c
                self.ops.append(("stl_l1_partial", (dt, x.data_ptr(), x.data_ptr() + half * esz, half, self.partial[s].data_ptr(), 1024)))
                self.ops.append(("stl_sum_partials", (self.partial[s].data_ptr(), 1024, 1.0 / half, self.loss.data_ptr(), int(s > 0))))


class VGGPerceptualLoss(nn.Module):
    def __init__(self, resize: bool = True, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 compute_dtype: str = "fp32"):
        super().__init__()
        self.resize = resize
        self.dtype = _dtype_code(compute_dtype)
        self.blocks = nn.ModuleList([nn.Module() for _ in range(4)])  # reference: self.blocks[s][features idx]
        for s, idx, ci, co, _ in VGG16_LAYOUT:
            leaf = nn.Module()
            leaf.register_parameter("weight", nn.Parameter(torch.zeros(co, ci, 3, 3), requires_grad=False))
            leaf.register_parameter("bias", nn.Parameter(torch.zeros(co), requires_grad=False))
            self.blocks[s].add_module(str(idx), leaf)
        self.mean = nn.Parameter(torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))   # loss.py:37
        self.std = nn.Parameter(torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))    # loss.py:38
        self._plans: Dict = {}
        self._flat_dev = None
        if state_dict is not None:
            self.load_vgg_weights(state_dict)

    def load_vgg_weights(self, sd: Dict[str, torch.Tensor]):
        """Accepts torchvision's ``features.<idx>.*`` keys or the reference module's ``blocks.<s>.<idx>.*``."""
        with torch.no_grad():
            for s, idx, _, _, _ in VGG16_LAYOUT:
                leaf = getattr(self.blocks[s], str(idx))
                for name in ("weight", "bias"):
                    t = sd.get(f"features.{idx}.{name}", sd.get(f"blocks.{s}.{idx}.{name}"))
                    if t is None:
                        raise KeyError(f"VGG16 weights: missing features.{idx}.{name}")
                    getattr(leaf, name).copy_(t)
        self._flat_dev = None

    def _pack(self, dev):
        ws, bs, self.bias_off = [], [], []
        off = 0
        for s, idx, _, co, _ in VGG16_LAYOUT:
            leaf = getattr(self.blocks[s], str(idx))
            ws.append(leaf.weight.detach().reshape(-1).float())
            bs.append(leaf.bias.detach().float())
            self.bias_off.append(off)
            off += co
        self.w_flat = torch.cat(ws).to(dev).contiguous()
        self.bias_flat = torch.cat(bs).to(dev).contiguous()
        self._flat_dev = dev
        self._plans.clear()

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if not input.is_cuda:
            raise RuntimeError("stlpose_amd.VGGPerceptualLoss runs only on an MI355X (cuda/HIP device); there is no CPU path")
        if input.shape[1] != 3:  # loss.py:43-45
            input, target = input.repeat(1, 3, 1, 1), target.repeat(1, 3, 1, 1)
        dev = input.device
        if self.mean.device != dev:
            self.to(dev)
        if self._flat_dev != dev:
            self._pack(dev)
        st = torch.cuda.current_stream().cuda_stream
        x = torch.cat([input, target.to(dev)], 0).contiguous().float()
        B2, _, H, W = x.shape
        if self.resize:  # bilinear commutes with the per-channel affine normalisation (weights sum to 1)
            r = torch.empty(B2, 3, 224, 224, device=dev)
            capi.call("stl_bilinear_nchw", x.data_ptr(), r.data_ptr(), B2, 3, H, W, 224, 224, st)
            x, H, W = r, 224, 224
        if H < 8 or W < 8:   # three 2x2 max-pools (floor, like nn.MaxPool2d) must leave at least one pixel
            raise RuntimeError(f"VGGPerceptualLoss needs H, W >= 8, got {H}x{W}")
        key = (B2, H, W)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = _Plan(self, B2, H, W, dev)
        plan.img.copy_(x)
        capi.call("stl_weight_prep", self.dtype, self.w_flat.data_ptr(), plan.wk.data_ptr(), plan.wtab.data_ptr(),
                  plan.nconv, plan.wblocks, st)
        lib = capi.lib()
        for name, args in plan.ops:
            rc = getattr(lib, name)(*args, st)
            if rc != 0:
                raise RuntimeError(f"{name}: {lib.stl_last_error().decode()}")
        return plan.loss.clone()
