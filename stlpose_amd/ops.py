"""``torch.library`` registration of the hot-path kernels (namespace ``stlpose``).

SURVEY.md 8(b) / BASELINE north_star name PyTorch-ROCm custom ops as the mechanism through which the Python
host reaches the HIP kernels.  Every op below is a thin dispatcher-visible wrapper around ONE entry point of the
C ABI (``include/stlpose_hip.h`` -> ``libstlpose_hip.so``): tensors are allocated by torch, the kernel is
enqueued on the current HIP stream, nothing is computed by ATen.  There is no CPU implementation: on a CPU
tensor the dispatcher raises (no kernel registered for that backend).  Fake (meta) implementations make the ops
traceable.  The stateful whole-network ops (``stlpose::hrnet_forward`` / ``hrnet_backward``) take the handle of
a planned engine (``engine.Engine``), because their plans own the activation buffers.
"""
from __future__ import annotations

import weakref
from typing import Tuple

import torch

from . import capi

_LIB = torch.library.Library("stlpose", "DEF")
_ENGINES = weakref.WeakValueDictionary()   # handle -> Engine (whole-network ops); the model owns its engines


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _define(schema: str, fn, fake=None):
    _LIB.define(schema)
    name = schema.split("(")[0]
    _LIB.impl(name, fn, "CUDA")
    if fake is not None:
        torch.library.register_fake(f"stlpose::{name}", fake, lib=_LIB)


# ------------------------------------------------------------------ loss (reference lib/loss.py:71-94)
def _mse(output: torch.Tensor, target: torch.Tensor, weight: torch.Tensor, scale: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    o, t = output.contiguous().float(), target.contiguous().float()
    b, j = o.shape[:2]
    w = weight.float().reshape(b, j).contiguous()
    dout = torch.empty_like(o)
    partial = torch.empty(256, dtype=torch.float64, device=o.device)
    loss = torch.empty((), dtype=torch.float32, device=o.device)
    capi.call("stl_mse_loss", o.data_ptr(), t.data_ptr(), w.data_ptr(), dout.data_ptr(), partial.data_ptr(), 256, loss.data_ptr(),
              b, j, o[0, 0].numel(), float(scale), _st())
    return loss, dout


_define("person_mse(Tensor output, Tensor target, Tensor weight, float scale=1.0) -> (Tensor, Tensor)", _mse,
        lambda o, t, w, scale=1.0: (o.new_empty(()), torch.empty_like(o)))


# ------------------------------------------------------------------ decode (lib/pose_parsing.py:16-92, lib/transforms.py:147-164)
def _argmax(hm: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    hm = hm.contiguous().float()
    b, j, h, w = hm.shape
    idx = torch.empty(b, j, dtype=torch.int32, device=hm.device)
    mx = torch.empty(b, j, 1, dtype=torch.float32, device=hm.device)
    preds = torch.empty(b, j, 2, dtype=torch.float32, device=hm.device)
    capi.call("stl_heatmap_argmax", hm.data_ptr(), idx.data_ptr(), mx.data_ptr(), preds.data_ptr(), b * j, h, w, _st())
    return idx, mx, preds


_define("heatmap_argmax(Tensor heatmaps) -> (Tensor, Tensor, Tensor)", _argmax,
        lambda hm: (hm.new_empty(hm.shape[:2], dtype=torch.int32), hm.new_empty(*hm.shape[:2], 1), hm.new_empty(*hm.shape[:2], 2)))


def _final_preds(hm: torch.Tensor, center: torch.Tensor, scale: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    hm = hm.contiguous().float()
    b, j, h, w = hm.shape
    preds = torch.empty(b, j, 2, dtype=torch.float32, device=hm.device)
    mx = torch.empty(b, j, 1, dtype=torch.float32, device=hm.device)
    capi.call("stl_final_preds", hm.data_ptr(), center.contiguous().float().data_ptr(), scale.contiguous().float().data_ptr(),
              preds.data_ptr(), mx.data_ptr(), b, j, h, w, _st())
    return preds, mx


_define("final_preds(Tensor heatmaps, Tensor center, Tensor scale) -> (Tensor, Tensor)", _final_preds,
        lambda hm, c, s: (hm.new_empty(*hm.shape[:2], 2), hm.new_empty(*hm.shape[:2], 1)))


def _flip_merge(out: torch.Tensor, out_flipped: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    a, f = out.contiguous().float(), out_flipped.contiguous().float()
    r = torch.empty_like(a)
    b, j, h, w = a.shape
    capi.call("stl_flip_merge", a.data_ptr(), f.data_ptr(), r.data_ptr(), perm.to(torch.int32).contiguous().data_ptr(), b, j, h, w, _st())
    return r


_define("flip_merge(Tensor out, Tensor out_flipped, Tensor perm) -> Tensor", _flip_merge, lambda a, f, p: torch.empty_like(a))


# ------------------------------------------------------------------ data pipeline (data/JointsDataset.py:189-286)
def _targets(joints: torch.Tensor, vis: torch.Tensor, hh: int, wh: int, sx: float, sy: float, sigma: float) -> Tuple[torch.Tensor, torch.Tensor]:
    j, v = joints.contiguous().float(), vis.contiguous().float()
    b, nj = j.shape[:2]
    target = torch.empty(b, nj, hh, wh, dtype=torch.float32, device=j.device)
    tw = torch.empty(b, nj, dtype=torch.float32, device=j.device)
    capi.call("stl_gaussian_targets", j.data_ptr(), v.data_ptr(), target.data_ptr(), tw.data_ptr(), b, nj, hh, wh, float(sx), float(sy),
              float(sigma), _st())
    return target, tw


_define("gaussian_targets(Tensor joints, Tensor vis, int hh, int wh, float sx, float sy, float sigma) -> (Tensor, Tensor)", _targets,
        lambda j, v, hh, wh, sx, sy, sigma: (j.new_empty(j.shape[0], j.shape[1], hh, wh), j.new_empty(j.shape[0], j.shape[1])))


def _crop(src: torch.Tensor, src_off: torch.Tensor, src_hw: torch.Tensor, minv: torch.Tensor, flip: torch.Tensor, ho: int, wo: int,
          mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    b = src_hw.shape[0]
    out = torch.empty(b, 3, ho, wo, dtype=torch.float32, device=src.device)
    capi.call("stl_affine_crop", src.data_ptr(), src_off.data_ptr(), src_hw.data_ptr(), minv.data_ptr(), flip.data_ptr(), out.data_ptr(),
              b, ho, wo, mean.data_ptr(), std.data_ptr(), _st())
    return out


_define("affine_crop(Tensor src, Tensor src_off, Tensor src_hw, Tensor minv, Tensor flip, int ho, int wo, Tensor mean, Tensor std) -> Tensor",
        _crop, lambda s, o, hw, m, f, ho, wo, mean, std: s.new_empty(hw.shape[0], 3, ho, wo, dtype=torch.float32))


# ------------------------------------------------------------------ whole network (models/HRnet.py:433-468 + its autograd backward)
def register_engine(eng) -> int:
    h = id(eng)
    _ENGINES[h] = eng
    return h


def _hrnet_fwd(img: torch.Tensor, handle: int) -> torch.Tensor:
    eng = _ENGINES[handle]
    eng.img.copy_(img)
    eng.forward(_st())
    return eng.out.clone()


def _hrnet_bwd(grad_out: torch.Tensor, handle: int) -> torch.Tensor:
    """Backward of the planned network for the LAST forward of that plan; returns the flat parameter gradient."""
    eng = _ENGINES[handle]
    eng.dout.copy_(grad_out)
    eng.backward(_st())
    return eng.store.grads.clone()   # a fresh tensor: an op must not hand out an alias of the plan's own buffer


def _hrnet_fwd_fake(img, engine):
    eng = _ENGINES[engine]   # joints and output stride come from the planned engine, not from constants
    return img.new_empty(tuple(eng.out.shape), dtype=torch.float32)


def _hrnet_bwd_fake(grad_out, engine):
    return grad_out.new_empty(_ENGINES[engine].store.nparam, dtype=torch.float32)


_define("hrnet_forward(Tensor img, int engine) -> Tensor", _hrnet_fwd, _hrnet_fwd_fake)
_define("hrnet_backward(Tensor grad_out, int engine) -> Tensor", _hrnet_bwd, _hrnet_bwd_fake)

OPS = ["person_mse", "heatmap_argmax", "final_preds", "flip_merge", "gaussian_targets", "affine_crop", "hrnet_forward", "hrnet_backward"]
