"""Drop-in ``PoseHighResolutionNet`` for MI355X.

Host-side mirror of the reference module (``src/models/HRnet.py:275-468``): same constructor
contract (``PoseHighResolutionNet(is_train=False)``, kwargs ignored -- ``lib/model_setup.py:38``),
same ``state_dict`` names/shapes (1754 entries for W32; loads the reference's
``pose_hrnet_w32_256x192.pth`` strictly), ``.train()/.eval()/.to()/.parameters()`` and
``model(img) -> (B, 17, H/4, W/4)`` fp32 NCHW, differentiable through ``loss.backward()``.

Everything between input and output runs in the hand-written HIP kernels of
``libstlpose_hip.so`` through the static plans of ``engine.py``; there is no torch.nn compute and
no CPU fallback (a CPU tensor raises).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import capi, ops
from .arch import ARCHS, Arch, registry
from .engine import Engine, ParamStore


class _Node(nn.Module):
    """Name-only container so that parameters get the reference's dotted state_dict keys."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("container module; call the top-level network")

    def __getitem__(self, i):  # index like the reference's nn.Sequential / nn.ModuleList
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)


def _dtype_code(name: str) -> int:
    name = name.lower()
    if name in ("bf16", "bfloat16"):
        return capi.BF16
    if name in ("fp32", "f32", "float32"):
        return capi.F32
    if name in ("mixed", "f16bf16", "fp16bf16"):   # forward tensors f16, gradients bf16 (capi.MIXED)
        return capi.MIXED
    raise ValueError(f"compute_dtype must be 'mixed', 'bf16' or 'fp32', got {name!r}")


def norm_device(device) -> torch.device:
    """``torch.device('cuda') != torch.device('cuda:0')``: always carry the index, so that a store
    created for 'cuda' is recognised when an input tensor reports 'cuda:0'."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class _Fn(torch.autograd.Function):
    """Autograd bridge: the whole network is one node; its backward replays the HIP backward plan
    and deposits parameter gradients straight into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, img, anchor, net, eng):
        ctx.net, ctx.eng = net, eng
        out = torch.ops.stlpose.hrnet_forward(img, ops.register_engine(eng))   # custom op: the planned forward program
        # The activations backward needs live in the engine's planned buffers, not in ctx: a later
        # forward through the same plan overwrites them.  Remember which forward this node belongs to
        # (Engine.forward counts EVERY pass through the plan, also no-grad ones and TrainStep's).
        ctx.generation = eng.generation
        return out

    @staticmethod
    def backward(ctx, gout):
        net, eng = ctx.net, ctx.eng
        if ctx.generation != eng.generation:
            raise RuntimeError(
                "stlpose_amd.PoseHighResolutionNet: backward through a stale forward -- the plan for this "
                f"(batch, resolution, mode) ran forward #{eng.generation} after the forward (#{ctx.generation}) being "
                "differentiated, and its activations were overwritten (the reference's autograd keeps one set "
                "per call; this engine keeps one per plan).  Call backward before the next forward of the same "
                "shape, or concatenate the inputs into one batch.")
        flat = torch.ops.stlpose.hrnet_backward(gout.contiguous(), ops.register_engine(eng))   # the flat parameter gradient
        net._publish_grads(flat)
        return None, None, None, None


class PoseHighResolutionNet(nn.Module):
    def __init__(self, arch: str = "w32", compute_dtype: Optional[str] = None, **kwargs):
        super().__init__()
        self.arch: Arch = ARCHS[arch] if isinstance(arch, str) else arch
        self.compute_dtype = _dtype_code(compute_dtype or os.environ.get("STLPOSE_DTYPE", "mixed"))
        self._reg = registry(self.arch)
        self._store: Optional[ParamStore] = None
        self._engines: Dict[Tuple, Engine] = {}
        self._anchor: Optional[torch.Tensor] = None
        self._leaf: Dict[str, Tuple[nn.Module, str]] = {}
        gen = torch.Generator().manual_seed(torch.initial_seed() & 0x7FFFFFFF)
        bn_w = {k[: -len("running_mean")] + "weight" for k, _ in self._reg.buffers if k.endswith("running_mean")}
        for key, shape, kind in self._reg.state:
            mod, leaf = self._node_for(key)
            if kind == "param":
                t = torch.empty(shape)
                if len(shape) == 4:      # nn.Conv2d default: kaiming_uniform_(a=sqrt(5)) -> U(-1/sqrt(fan_in), ..)
                    bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
                    t.uniform_(-bound, bound, generator=gen)
                elif key in bn_w:
                    t.fill_(1.0)
                elif key.endswith("final_layer.bias"):
                    bound = 1.0 / math.sqrt(self.arch.widths[0])
                    t.uniform_(-bound, bound, generator=gen)
                else:
                    t.zero_()
                mod.register_parameter(leaf, nn.Parameter(t))
            elif kind == "nbt":
                mod.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
            else:
                mod.register_buffer(leaf, torch.ones(shape) if key.endswith("running_var") else torch.zeros(shape))
            self._leaf[key] = (mod, leaf)

    # ------------------------------------------------------------------ module tree / flat storage
    def _node_for(self, key: str):
        parts = key.split(".")
        mod: nn.Module = self
        for p in parts[:-1]:
            nxt = mod._modules.get(p)
            if nxt is None:
                nxt = _Node()
                mod.add_module(p, nxt)
            mod = nxt
        return mod, parts[-1]

    def _tensor(self, key: str) -> torch.Tensor:
        mod, leaf = self._leaf[key]
        return getattr(mod, leaf)

    def _packed(self, device) -> bool:
        st = self._store
        if st is None or st.device != norm_device(device):
            return False
        first, last = self._reg.params[0][0], self._reg.params[-1][0]
        return (self._tensor(first).data_ptr() == st.master.data_ptr()
                and self._tensor(last).data_ptr() == st.master.data_ptr() + 4 * st.param_off[last])

    def _pack(self, device):
        """(Re)point every parameter/buffer at its slice of the flat fp32 storage on `device`."""
        if self._holder() is not None:
            raise RuntimeError(
                "stlpose_amd.PoseHighResolutionNet: the flat parameter storage is held by a TrainStep / Trainer "
                f"on {self._store.device}; re-packing it for {device} would silently detach the optimiser from the "
                "module's parameters.  Keep the model on the TrainStep's device (or build a new TrainStep).")
        device = norm_device(device)
        st = ParamStore(self._reg, device)
        with torch.no_grad():
            for key, shape in self._reg.params:
                off = st.param_off[key]
                n = int(math.prod(shape))
                view = st.master[off:off + n].view(shape)
                p = self._tensor(key)
                view.copy_(p.data)
                p.data = view
                p.grad = None
            for key, shape in self._reg.buffers:
                b = self._tensor(key)
                if key.endswith("num_batches_tracked"):
                    view = st.nbt[st.nbt_idx[key]]
                else:
                    off = st.buf_off[key]
                    view = st.bufs[off:off + int(math.prod(shape))].view(shape)
                view.copy_(b)
                mod, leaf = self._leaf[key]
                mod._buffers[leaf] = view
        self._store = st
        self._engines.clear()
        self._anchor = torch.zeros(1, device=device, requires_grad=True)

    def _holder(self):
        """The live TrainStep / Trainer that owns the flat storage, or None (the reference is weak: a dropped
        TrainStep releases the model)."""
        ref = self.__dict__.get("_pinned_by")
        return ref() if ref is not None else None

    def __getstate__(self):   # torch.save(model) / pickle: the weak reference and the device plans stay behind
        d = dict(self.__dict__)
        for k in ("_pinned_by", "_engines", "_store", "_anchor", "_grad_pub"):
            d.pop(k, None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._store, self._engines, self._anchor = None, {}, None

    def _publish_grads(self, flat: torch.Tensor):
        """Expose the gradients as per-parameter ``.grad`` views of one flat tensor (`flat`: the copy of the engine's
        gradient buffer the backward op returned; the engine's own buffer is overwritten by the next backward);
        accumulates when the caller has not zeroed its gradients, like autograd would."""
        st = self._store
        pub = getattr(self, "_grad_pub", None)
        first = self._tensor(self._reg.params[0][0]).grad
        if pub is not None and first is not None and first.data_ptr() == pub.data_ptr():
            pub.add_(flat)       # gradient accumulation across backward calls
            return
        pub = self._grad_pub = flat
        for key, shape in self._reg.params:
            off = st.param_off[key]
            self._tensor(key).grad = pub[off:off + int(math.prod(shape))].view(shape)

    def engine(self, B: int, H: int, W: int, training: bool) -> Engine:
        key = (B, H, W, training, self.compute_dtype)
        e = self._engines.get(key)
        if e is None:
            e = self._engines[key] = Engine(self.arch, self._store, B, H, W, self.compute_dtype, training)
        return e

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("stlpose_amd.PoseHighResolutionNet runs only on an MI355X (cuda/HIP device); "
                               "there is no CPU path")
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"expected input (B, 3, H, W), got {tuple(x.shape)}")
        if not self._packed(x.device):
            self._pack(x.device)
        B, _, H, W = x.shape
        eng = self.engine(B, H, W, self.training)
        if self.training and torch.is_grad_enabled():
            return _Fn.apply(x, self._anchor, self, eng)
        with torch.no_grad():
            return torch.ops.stlpose.hrnet_forward(x, ops.register_engine(eng))

    def load_pretrained(self, pretrained: str = ""):
        """reference HRnet.py:470-499: conv ~ N(0, 0.001), BN gamma 1 / beta 0, then an optional
        non-strict load of a checkpoint file."""
        with torch.no_grad():
            bn_w = {k[: -len("running_mean")] + "weight" for k, _ in self._reg.buffers if k.endswith("running_mean")}
            for key, shape in self._reg.params:
                t = self._tensor(key)
                if len(shape) == 4:
                    t.normal_(0.0, 0.001)
                elif key in bn_w:
                    t.fill_(1.0)
                else:
                    t.zero_()
        if pretrained and os.path.isfile(pretrained):
            self.load_state_dict(torch.load(pretrained, map_location="cpu"), strict=False)
        elif pretrained:
            raise ValueError(f"{pretrained} is not exist!")
        return self
