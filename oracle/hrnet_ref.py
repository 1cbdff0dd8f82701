"""ORACLE (test infrastructure, never imported by the product path).

Plain ``torch.nn`` fp32 restatement of the reference pose network, written as a
table-driven builder.  It reproduces

* the ``state_dict`` key names and shapes of ``PoseHighResolutionNet``
  (reference ``src/models/HRnet.py:275-339``) so that the same weight tensors
  load into the reference, this oracle and the HIP product module, and
* its arithmetic: stem (``HRnet.py:290-297,434-440``), residual units
  (``:32-61`` two-3x3 unit, ``:64-102`` 1x1/3x3/1x1 unit), transitions
  (``:341-380,443-463``), multi-resolution modules (``:105-266``; branches
  ``:140-186``, cross-resolution sum ``:188-243,255-264``) and the 1x1 head
  (``:331-337,466``).

Pinned against the reference by ``tests/golden/make_golden.py`` (see
``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# Architecture tables.  W32/W48 values = the upstream cfg_hrnet_w{32,48}_*.yaml that the
# reference reads from outside its tree (HRnet.py:280-283); "tiny" is a reduced-width net
# used for fast CPU tests and goldens.
ARCHS: Dict[str, dict] = {
    "w32": dict(widths=[32, 64, 128, 256], modules=[1, 4, 3], blocks=4, joints=17, stem=64),
    "w48": dict(widths=[48, 96, 192, 384], modules=[1, 4, 3], blocks=4, joints=17, stem=64),
    "tiny": dict(widths=[16, 32, 48, 64], modules=[1, 2, 2], blocks=2, joints=17, stem=64),
}

_MOM = 0.1  # HRnet.py:23


def _bn(c: int) -> nn.BatchNorm2d:
    return nn.BatchNorm2d(c, momentum=_MOM)


def _c3(cin: int, cout: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, 3, stride, 1, bias=False)


def _c1(cin: int, cout: int) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, 1, 1, 0, bias=False)


class TwoConvUnit(nn.Module):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + x)  -- HRnet.py:32-61 (expansion 1)."""

    def __init__(self, c: int):
        super().__init__()
        self.conv1, self.bn1 = _c3(c, c), _bn(c)
        self.conv2, self.bn2 = _c3(c, c), _bn(c)

    def forward(self, x):
        h = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(h)) + x)


class ThreeConvUnit(nn.Module):
    """1x1 -> 3x3 -> 1x1(x4) residual unit -- HRnet.py:64-102; projection shortcut :384-391."""

    def __init__(self, cin: int, mid: int, project: bool):
        super().__init__()
        self.conv1, self.bn1 = _c1(cin, mid), _bn(mid)
        self.conv2, self.bn2 = _c3(mid, mid), _bn(mid)
        self.conv3, self.bn3 = _c1(mid, 4 * mid), _bn(4 * mid)
        if project:
            self.downsample = nn.Sequential(_c1(cin, 4 * mid), _bn(4 * mid))
        else:
            self.downsample = None

    def forward(self, x):
        h = F.relu(self.bn1(self.conv1(x)))
        h = F.relu(self.bn2(self.conv2(h)))
        h = self.bn3(self.conv3(h))
        r = x if self.downsample is None else self.downsample(x)
        return F.relu(h + r)


def _down_chain(cin: int, cout: int, hops: int) -> nn.Sequential:
    """`hops` stride-2 3x3 convs; width stays `cin` until the last hop; ReLU after all but
    the last (HRnet.py:213-240)."""
    seq: List[nn.Module] = []
    for k in range(hops):
        last = k == hops - 1
        mods = [_c3(cin, cout if last else cin, 2), _bn(cout if last else cin)]
        if not last:
            mods.append(nn.ReLU(True))
        seq.append(nn.Sequential(*mods))
    return nn.Sequential(*seq)


class ExchangeModule(nn.Module):
    """One multi-resolution module: per-branch residual units then the cross-resolution sum
    (HRnet.py:105-266).  `full_out=False` builds only output 0 (:195,:413-416)."""

    def __init__(self, widths: Sequence[int], nblocks: int, full_out: bool = True):
        super().__init__()
        n = len(widths)
        self.branches = nn.ModuleList(
            nn.Sequential(*[TwoConvUnit(c) for _ in range(nblocks)]) for c in widths)
        rows = []
        for i in range(n if full_out else 1):
            row: List[nn.Module | None] = []
            for j in range(n):
                if j > i:  # coarser -> finer: 1x1, BN, nearest x2^(j-i)  (HRnet.py:198-209)
                    row.append(nn.Sequential(
                        _c1(widths[j], widths[i]), nn.BatchNorm2d(widths[i]),
                        nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(None)
                else:
                    row.append(_down_chain(widths[j], widths[i], i - j))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows)

    def forward(self, xs):
        xs = [b(x) for b, x in zip(self.branches, xs)]
        outs = []
        for i, row in enumerate(self.fuse_layers):
            acc = None
            for j, x in enumerate(xs):  # summation order j = 0,1,2,... (HRnet.py:258-263)
                t = x if row[j] is None else row[j](x)
                acc = t if acc is None else acc + t
            outs.append(F.relu(acc))
        return outs


class RefPoseNet(nn.Module):
    """Oracle counterpart of PoseHighResolutionNet (HRnet.py:275-468)."""

    def __init__(self, arch: str | dict = "w32"):
        super().__init__()
        a = ARCHS[arch] if isinstance(arch, str) else arch
        self.arch = a
        w, stem = a["widths"], a["stem"]
        self.conv1, self.bn1 = _c3(3, stem, 2), _bn(stem)
        self.conv2, self.bn2 = _c3(stem, stem, 2), _bn(stem)
        self.layer1 = nn.Sequential(
            ThreeConvUnit(stem, 64, True), *[ThreeConvUnit(256, 64, False) for _ in range(3)])
        # registration order follows HRnet.py:305-329 so that state_dict key ORDER matches too
        m, nb = a["modules"], a["blocks"]

        def _t(cin, cout):  # one stride-2 transition hop (HRnet.py:364-378)
            return nn.Sequential(nn.Sequential(_c3(cin, cout, 2), nn.BatchNorm2d(cout), nn.ReLU(True)))
        self.transition1 = nn.ModuleList([
            nn.Sequential(_c3(256, w[0]), nn.BatchNorm2d(w[0]), nn.ReLU(True)), _t(256, w[1])])
        self.stage2 = nn.Sequential(*[ExchangeModule(w[:2], nb) for _ in range(m[0])])
        self.transition2 = nn.ModuleList([None, None, _t(w[1], w[2])])
        self.stage3 = nn.Sequential(*[ExchangeModule(w[:3], nb) for _ in range(m[1])])
        self.transition3 = nn.ModuleList([None, None, None, _t(w[2], w[3])])
        self.stage4 = nn.Sequential(*[
            ExchangeModule(w[:4], nb, full_out=(k != m[2] - 1)) for k in range(m[2])])
        self.final_layer = nn.Conv2d(w[0], a["joints"], 1, 1, 0)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        ys = [self.transition1[0](x), self.transition1[1](x)]
        ys = self.stage2(ys)
        ys = self.stage3([ys[0], ys[1], self.transition2[2](ys[-1])])
        ys = self.stage4([ys[0], ys[1], ys[2], self.transition3[3](ys[-1])])
        return self.final_layer(ys[0])


# ----------------------------------------------------------------------------------------
# Framework-independent synthetic weights: every state_dict entry is drawn
# from numpy PCG64 seeded by crc32(key) so that reference, oracle and HIP module can all be
# filled identically without shipping a checkpoint.
# ----------------------------------------------------------------------------------------
def synth_tensor(key: str, shape: Sequence[int], seed: int = 0) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) + 7919 * seed) & 0xFFFFFFFF))
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if key.endswith("running_mean"):
        return rng.normal(0.0, 0.1, shape).astype(np.float32)
    if key.endswith("running_var"):
        return rng.uniform(0.8, 1.6, shape).astype(np.float32)
    if len(shape) == 4:  # conv weight: He-normal on fan_in
        fan_in = shape[1] * shape[2] * shape[3]
        return rng.normal(0.0, np.sqrt(2.0 / fan_in), shape).astype(np.float32)
    if key.endswith("final_layer.bias"):
        return rng.normal(0.0, 0.05, shape).astype(np.float32)
    if key.endswith("weight"):  # BN gamma
        # last BN of a residual unit gets a small gain (cf. zero-init-residual) so that the
        # eval-mode net, whose running stats are synthetic too, stays O(1) through 100+ units
        if key.endswith(".bn3.weight") or (key.endswith(".bn2.weight") and "branches" in key):
            return rng.uniform(0.1, 0.3, shape).astype(np.float32)
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if key.endswith("bias"):  # BN beta
        return rng.uniform(-0.2, 0.2, shape).astype(np.float32)
    raise KeyError(key)


def synth_state_dict(model: nn.Module, seed: int = 0) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in model.state_dict().items():
        out[k] = torch.from_numpy(synth_tensor(k, v.shape, seed))
    return out


def load_synth(model: nn.Module, seed: int = 0) -> nn.Module:
    model.load_state_dict(synth_state_dict(model, seed), strict=True)
    return model


# ----------------------------------------------------------------------------------------
# bf16-STORAGE emulation (calibration of the bf16 path's parity bar, tests only).
# The HIP bf16 path keeps activations and kernel-layout weights in bf16 and accumulates in
# fp32.  The same rounding points are imposed on this fp32 oracle with hooks: every conv
# (the head excepted: fp32 weights, fp32 output) sees its input and its weight rounded to
# bf16 and has its raw output rounded to bf16 before the BatchNorm reads it.  The distance of
# that run from the plain fp32 oracle is what bf16 STORAGE costs on a given batch -- the HIP
# path is held to a small multiple of it, not to a free constant.  The materialised sums
# (residual block ends, exchange sums, transition outputs) are rounded too: they are stored in
# bf16 by the HIP path and travel on along the skip connections, i.e. the residual stream
# itself carries 2^-9 of relative rounding per block.
# ----------------------------------------------------------------------------------------
class bf16_storage:
    def __init__(self, model: nn.Module):
        self.model, self.handles, self.saved = model, [], {}

    @staticmethod
    def _r(t: torch.Tensor) -> torch.Tensor:
        return t.to(torch.bfloat16).to(t.dtype)

    def __enter__(self):
        for name, m in self.model.named_modules():
            if not isinstance(m, nn.Conv2d):
                continue
            head = name == "final_layer"
            self.handles.append(m.register_forward_pre_hook(lambda mod, args: (type(self)._r(args[0]),)))
            if not head:
                self.saved[name] = m.weight.data
                m.weight.data = type(self)._r(m.weight.data)
                self.handles.append(m.register_forward_hook(lambda mod, args, out: type(self)._r(out)))
        import re

        def round_out(mod, args, out):
            return [type(self)._r(t) for t in out] if isinstance(out, (list, tuple)) else type(self)._r(out)
        for name, m in self.model.named_modules():
            if isinstance(m, (TwoConvUnit, ThreeConvUnit, ExchangeModule)) or re.fullmatch(r"transition\d\.\d", name):
                self.handles.append(m.register_forward_hook(round_out))
        return self.model

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
        mods = dict(self.model.named_modules())
        for name, w in self.saved.items():
            mods[name].weight.data = w
        self.handles, self.saved = [], {}
        return False


class f16_storage(bf16_storage):
    """The same rounding points with IEEE half (10 mantissa bits): the FORWARD side of the HIP mixed 16-bit mode (forward
    tensors and forward kernel-layout weights f16).  The mixed-mode parity bars are tied to this run: it must
    pass them itself (tests/test_parity_r3_gpu.py)."""

    @staticmethod
    def _r(t: torch.Tensor) -> torch.Tensor:
        return t.to(torch.float16).to(t.dtype)
