"""VGG19 perceptual style-transfer loss on MI355X: content (relu4_2 MSE) + Gram-matrix style loss
(relu1_1 .. relu5_1) -- the forward pass BASELINE.json configs[3] names ("VGG19 perceptual style-transfer forward
(content + Gram style loss) 512x512 bs=16 fp32").

**No reference counterpart** (SURVEY.md 8a, row V2): /root/reference contains only the VGG16 + L1
``VGGPerceptualLoss`` (``lib/loss.py:17-58`` -> ``stlpose_amd/vgg.py``); this module follows the published method
(Gatys et al. 2016, Johnson et al. 2016) and is checked against ``oracle.vgg_ref.vgg19_style_content_loss`` only:
PARITY UNPINNED.

MI355X-first, on the same kernels as the V1 path: the stylised / content / style images run as ONE batch of 3B
through the implicit-GEMM conv kernel (bias + ReLU in its epilogue; conv1_1 as 27 -> 32-wide patches with the
ImageNet normalisation fused into the patch kernel), 2x2 max-pools, and every Gram matrix G = F F^T / (C H W) is
the MFMA weight-gradient kernel run as a 1x1 "convolution" of the NHWC feature map with itself
(dw[c1][c2] = sum_pixels F[p][c1] F[p][c2]: K = pixels, transposed LDS reads, split-K slabs), one launch per image
and tap.  The content term is a two-level fp64 reduction of squared differences (``stl_l2_partial``).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import capi

# (features index, cin, cout, 2x2 max-pool in front): torchvision VGG19 "E" up to conv5_1
VGG19_LAYOUT = [(0, 3, 64, False), (2, 64, 64, False), (5, 64, 128, True), (7, 128, 128, False), (10, 128, 256, True),
                (12, 256, 256, False), (14, 256, 256, False), (16, 256, 256, False), (19, 256, 512, True), (21, 512, 512, False),
                (23, 512, 512, False), (25, 512, 512, False), (28, 512, 512, True)]
STYLE_TAPS = (0, 2, 4, 8, 12)   # relu1_1, relu2_1, relu3_1, relu4_1, relu5_1
CONTENT_TAP = 9                 # relu4_2


def vgg19_flops_per_image(H: int, W: int) -> float:
    """2 * MACs of the 13 convolutions up to conv5_1."""
    fl, h, w = 0.0, H, W
    for _, ci, co, pool in VGG19_LAYOUT:
        if pool:
            h, w = h // 2, w // 2
        fl += 2.0 * h * w * co * ci * 9
    return fl


class _Plan:
    def __init__(self, mod: "VGG19StyleLoss", B3: int, H: int, W: int, dev):
        self.ops, self.keep = [], []
        dt, esz = mod.dtype, (2 if mod.dtype == capi.BF16 else 4)
        tdt = torch.bfloat16 if dt == capi.BF16 else torch.float32
        B = B3 // 3
        self.img = torch.zeros(B3, 3, H, W, device=dev)
        nconv = len(VGG19_LAYOUT)
        tab = (capi.WPrep * nconv)()
        src = fwd = blk = 0
        offs = []
        for i, (_, ci, co, _) in enumerate(VGG19_LAYOUT):
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
        self.cpartial = torch.zeros(1024, dtype=torch.float64, device=dev)
        self.content = torch.zeros((), dtype=torch.float32, device=dev)
        self.grams = []   # per style tap: (slabs [2B, nsplit, C, C] fp32, 1 / (C H W))

        def act(b, h, w, c):
            t = torch.empty(b * h * w * c * esz, dtype=torch.uint8, device=dev)
            self.keep.append(t)
            return t

        x = act(B3, H, W, 32)
        self.ops.append(("stl_patch3x3", (dt, self.img.data_ptr(), x.data_ptr(), B3, H, W, 1, mod.mean.data_ptr(), mod.std.data_ptr())))
        h, w, c = H, W, 32
        for i, (_, ci, co, pool) in enumerate(VGG19_LAYOUT):
            if pool:
                y = act(B3, h // 2, w // 2, c)
                self.ops.append(("stl_maxpool2x2", (dt, x.data_ptr(), y.data_ptr(), B3, h, w, c)))
                x, h, w = y, h // 2, w // 2
            p = capi.Conv()
            p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co = dt, B3, h, w, c, h, w, co
            p.ks, p.stride, p.shape = (1 if i == 0 else 3), 1, -1
            p.src.x, p.src.mode = x.data_ptr(), capi.SRC_PLAIN
            y = act(B3, h, w, co)
            p.w = self.wk.data_ptr() + offs[i] * esz
            p.out, p.bias, p.out_relu = y.data_ptr(), mod.bias_flat.data_ptr() + 4 * mod.bias_off[i], 1
            capi.call("stl_conv_plan", C.byref(p))
            self.keep.append(p)
            self.ops.append(("stl_conv_forward", (C.byref(p),)))
            x, c = y, co
            img_elems = h * w * c
            if i == CONTENT_TAP:   # images [0, B) = stylised, [B, 2B) = content
                n = B * img_elems
                self.ops.append(("stl_l2_partial", (dt, x.data_ptr(), x.data_ptr() + n * esz, n, self.cpartial.data_ptr(), 1024)))
                self.ops.append(("stl_sum_partials", (self.cpartial.data_ptr(), 1024, 1.0 / n, self.content.data_ptr(), 0)))
            if i in STYLE_TAPS:    # Gram of the stylised images [0, B) and the style images [2B, 3B)
                from .engine import choose_tile
                th, tw = choose_tile(1, h, w, 1, 1, esz, bn_cols=32, maxhalo=576)
                npt = math.ceil((h + 1) / th) * math.ceil(w / tw)
                wg0 = capi.Wgrad()
                wg0.dtype, wg0.B, wg0.Hi, wg0.Wi, wg0.Ci, wg0.Ho, wg0.Wo, wg0.Co, wg0.ks, wg0.stride = dt, 1, h, w, c, h, w, c, 1, 1
                ctile = capi.lib().stl_wgrad_chunk(C.byref(wg0))
                chunks = math.ceil(c / ctile) ** 2
                nsplit = max(1, min(npt, max(1, 256 // chunks)))
                slabs = torch.zeros(2 * B, nsplit, c, c, dtype=torch.float32, device=dev)
                for j, b in enumerate(list(range(B)) + list(range(2 * B, 3 * B))):
                    wg = capi.Wgrad()
                    wg.dtype, wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co, wg.ks, wg.stride = dt, 1, h, w, c, h, w, c, 1, 1
                    wg.TH, wg.TW, wg.nsplit = th, tw, nsplit
                    ptr = x.data_ptr() + b * img_elems * esz
                    wg.h.x, wg.h.mode, wg.g.x, wg.g.mode = ptr, capi.SRC_PLAIN, ptr, capi.SRC_PLAIN
                    wg.partial = slabs[j].data_ptr()
                    self.keep.append(wg)
                    self.ops.append(("stl_conv_wgrad", (C.byref(wg),)))
                self.grams.append((slabs, 1.0 / (c * h * w)))


class VGG19StyleLoss(nn.Module):
    """``VGG19StyleLoss()(stylised, content, style) -> (total, content_loss, style_loss)`` on NCHW images in [0, 1]."""

    def __init__(self, content_weight: float = 1.0, style_weight: float = 1e5, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 compute_dtype: str = "fp32"):
        super().__init__()
        self.content_weight, self.style_weight = float(content_weight), float(style_weight)
        self.dtype = capi.BF16 if compute_dtype.lower() in ("bf16", "bfloat16") else capi.F32
        self.features = nn.Module()
        for idx, ci, co, _ in VGG19_LAYOUT:
            leaf = nn.Module()
            leaf.register_parameter("weight", nn.Parameter(torch.zeros(co, ci, 3, 3), requires_grad=False))
            leaf.register_parameter("bias", nn.Parameter(torch.zeros(co), requires_grad=False))
            self.features.add_module(str(idx), leaf)
        self.mean = nn.Parameter(torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), requires_grad=False)
        self.std = nn.Parameter(torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1), requires_grad=False)
        self._plans: Dict = {}
        self._flat_dev = None
        if state_dict is not None:
            self.load_state_dict({k: v for k, v in state_dict.items() if k.startswith("features.")}, strict=False)

    def _pack(self, dev):
        ws, bs, self.bias_off = [], [], []
        off = 0
        for idx, _, co, _ in VGG19_LAYOUT:
            leaf = getattr(self.features, str(idx))
            ws.append(leaf.weight.detach().reshape(-1).float())
            bs.append(leaf.bias.detach().float())
            self.bias_off.append(off)
            off += co
        self.w_flat = torch.cat(ws).to(dev).contiguous()
        self.bias_flat = torch.cat(bs).to(dev).contiguous()
        self._flat_dev = dev
        self._plans.clear()

    def forward(self, x: torch.Tensor, content: torch.Tensor, style: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("stlpose_amd.VGG19StyleLoss runs only on an MI355X (cuda/HIP device); there is no CPU path")
        dev = x.device
        if self.mean.device != dev:
            self.to(dev)
        if self._flat_dev != dev:
            self._pack(dev)
        st = torch.cuda.current_stream().cuda_stream
        xin = torch.cat([x, content.to(dev), style.to(dev)], 0).contiguous().float()
        B3, ch, H, W = xin.shape
        if ch != 3 or B3 % 3 or H < 16 or W < 16:
            raise RuntimeError(f"VGG19StyleLoss needs three equal batches of (B, 3, H >= 16, W >= 16) images, got {tuple(xin.shape)}")
        plan = self._plans.get((B3, H, W))
        if plan is None:
            plan = self._plans[(B3, H, W)] = _Plan(self, B3, H, W, dev)
        plan.img.copy_(xin)
        capi.call("stl_weight_prep", self.dtype, self.w_flat.data_ptr(), plan.wk.data_ptr(), plan.wtab.data_ptr(), plan.nconv, plan.wblocks, st)
        lib = capi.lib()
        for name, args in plan.ops:
            rc = getattr(lib, name)(*args, st)
            if rc != 0:
                raise RuntimeError(f"{name}: {lib.stl_last_error().decode()}")
        # C x C Gram matrices: split-K slabs -> sum, scale, squared distance (a few hundred KB of bookkeeping)
        B = B3 // 3
        s_loss = torch.zeros((), dtype=torch.float64, device=dev)
        for slabs, scale in plan.grams:
            g = slabs.double().sum(1) * scale
            s_loss = s_loss + ((g[:B] - g[B:]) ** 2).mean()
        s_loss = s_loss.float()
        c_loss = plan.content.clone()
        return self.content_weight * c_loss + self.style_weight * s_loss, c_loss, s_loss
