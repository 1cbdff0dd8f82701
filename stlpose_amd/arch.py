"""Architecture tables and the network topology of the pose network, written once as a walk
over an abstract builder.  The same walk yields (a) the ordered parameter/buffer registry whose
names and shapes are the reference's ``state_dict`` ABI (``src/models/HRnet.py:275-339``) and
(b) the HIP execution plan (``engine.py``).

Values of W32 / W48 are the upstream ``cfg_hrnet_w{32,48}`` YAMLs that the reference reads from
outside its tree (``HRnet.py:280-283``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple


@dataclass(frozen=True)
class Arch:
    name: str
    widths: Tuple[int, ...]
    modules: Tuple[int, int, int]  # NUM_MODULES of stage 2/3/4
    blocks: int = 4                # NUM_BLOCKS per branch
    joints: int = 17
    stem: int = 64


ARCHS = {
    "w32": Arch("w32", (32, 64, 128, 256), (1, 4, 3)),
    "w48": Arch("w48", (48, 96, 192, 384), (1, 4, 3)),
    "tiny": Arch("tiny", (16, 32, 48, 64), (1, 2, 2), blocks=2),
}


def walk(g, a: Arch):
    """Drive builder `g` through the network.  Builder protocol:
      g.stem_input()                                  -> act  (3x3/s2 patches of the image)
      g.conv_bn(conv_key, bn_key, x, cout, ks, stride, relu, patch=False) -> act (BN on load)
      g.fuse(terms=[(act, log2_upsample)], relu)      -> act  (materialised)
      g.head(key, x, joints)                          -> output
      g.set_stream(i)                                 scheduling hint: following ops go to HIP stream i
    Call order == reference registration order, so the registry comes out in state_dict order.
    """
    w = a.widths
    # stem (HRnet.py:290-296, 434-439); conv1 runs as a 1x1 conv over 3x3/s2 patches
    x = g.stem_input()
    x = g.conv_bn("conv1", "bn1", x, a.stem, 3, 2, True, patch=True)
    x = g.conv_bn("conv2", "bn2", x, a.stem, 3, 2, True)
    x = g.fuse([(x, 0)], relu=False)  # two consumers below -> materialise once
    # layer1: four 1x1/3x3/1x1 residual units, 64 -> 256 (HRnet.py:64-102, 297, 382-399)
    for i in range(4):
        p = f"layer1.{i}"
        h = g.conv_bn(f"{p}.conv1", f"{p}.bn1", x, 64, 1, 1, True)
        h = g.conv_bn(f"{p}.conv2", f"{p}.bn2", h, 64, 3, 1, True)
        h = g.conv_bn(f"{p}.conv3", f"{p}.bn3", h, 256, 1, 1, False)
        r = g.conv_bn(f"{p}.downsample.0", f"{p}.downsample.1", x, 256, 1, 1, False) if i == 0 else x
        x = g.fuse([(h, 0), (r, 0)], relu=True)
    # transition1 (HRnet.py:341-380, 443-447)
    # (the two transition convolutions read the same 113 MB tensor back to back; putting the second on branch 1's stream
    # measured 14.55-14.61 vs 14.54-14.56 ms per step in round 3 -- both are bandwidth-bound -- and was dropped)
    t0 = g.conv_bn("transition1.0.0", "transition1.0.1", x, w[0], 3, 1, True)
    y0 = g.fuse([(t0, 0)], relu=False)
    t1 = g.conv_bn("transition1.1.0.0", "transition1.1.0.1", x, w[1], 3, 2, True)
    y1 = g.fuse([(t1, 0)], relu=False)
    ys = [y0, y1]
    for stage, nbr in ((2, 2), (3, 3), (4, 4)):
        nmod = a.modules[stage - 2]
        for m in range(nmod):
            full = not (stage == 4 and m == nmod - 1)  # last module emits branch 0 only (:413-416)
            ys = _exchange_module(g, f"stage{stage}.{m}", ys, w[:nbr], a.blocks, full)
        if stage < 4:  # transition to the next stage: new branch from the LAST branch (:451-463)
            key = f"transition{stage}.{nbr}.0"
            g.set_stream(nbr)
            t = g.conv_bn(f"{key}.0", f"{key}.1", ys[-1], w[nbr], 3, 2, True)
            ys = ys + [g.fuse([(t, 0)], relu=False)]
            g.set_stream(0)
    return g.head("final_layer", ys[0], a.joints)


def _exchange_module(g, p: str, xs: List, widths: Sequence[int], nblocks: int, full: bool):
    """HRnet.py:105-266: per-branch two-conv residual units, then out_i = ReLU(sum_j f_ij(x_j))."""
    n = len(widths)
    xs = list(xs)
    for b in range(n):
        g.set_stream(b)  # branches are independent until the exchange -> one HIP stream each
        x = xs[b]
        for k in range(nblocks):
            q = f"{p}.branches.{b}.{k}"
            h = g.conv_bn(f"{q}.conv1", f"{q}.bn1", x, widths[b], 3, 1, True)
            h = g.conv_bn(f"{q}.conv2", f"{q}.bn2", h, widths[b], 3, 1, False)
            x = g.fuse([(h, 0), (x, 0)], relu=True)
        xs[b] = x
    outs = []
    # Exchange: out_i = ReLU(sum_j f_ij(x_j)): everything that feeds out_i runs on stream i.  (Spreading the f_ij chains over
    # the streams by estimated cost measured 18.4 -> 18.8 ms per step in round 2: every moved chain pays a cross-stream event
    # wait at both ends, more than the serialisation it removes.)
    for i in range(n if full else 1):
        terms = []
        g.set_stream(i)
        for j in range(n):
            q = f"{p}.fuse_layers.{i}.{j}"
            if j == i:
                terms.append((xs[j], 0))
            elif j > i:  # 1x1 conv + BN, nearest-upsampled 2^(j-i) inside the sum kernel
                terms.append((g.conv_bn(f"{q}.0", f"{q}.1", xs[j], widths[i], 1, 1, False), j - i))
            else:        # (i-j) stride-2 3x3 hops, ReLU after all but the last
                t = xs[j]
                for k in range(i - j):
                    last = k == i - j - 1
                    t = g.conv_bn(f"{q}.{k}.0", f"{q}.{k}.1", t, widths[i] if last else widths[j], 3, 2, not last)
                terms.append((t, 0))
        outs.append(g.fuse(terms, relu=True))
    g.set_stream(0)
    return outs


# ------------------------------------------------------------------------------------------
@dataclass
class Registry:
    """Ordered state_dict layout: params (fp32 master) and buffers."""
    params: List[Tuple[str, Tuple[int, ...]]] = field(default_factory=list)
    buffers: List[Tuple[str, Tuple[int, ...]]] = field(default_factory=list)
    state: List[Tuple[str, Tuple[int, ...], str]] = field(default_factory=list)  # (key, shape, kind)


class _RegistryBuilder:
    """Dry builder: records parameter creation order only."""

    def __init__(self):
        self.reg = Registry()

    def _p(self, key, shape):
        self.reg.params.append((key, tuple(shape)))
        self.reg.state.append((key, tuple(shape), "param"))

    def _b(self, key, shape, kind="buffer"):
        self.reg.buffers.append((key, tuple(shape)))
        self.reg.state.append((key, tuple(shape), kind))

    def stem_input(self):
        return 3

    def set_stream(self, s):
        pass

    def conv_bn(self, ck, bk, x, cout, ks, stride, relu, patch=False):
        cin = x
        self._p(ck + ".weight", (cout, cin, ks, ks))
        self._p(bk + ".weight", (cout,))
        self._p(bk + ".bias", (cout,))
        self._b(bk + ".running_mean", (cout,))
        self._b(bk + ".running_var", (cout,))
        self._b(bk + ".num_batches_tracked", (), "nbt")
        return cout

    def fuse(self, terms, relu):
        return terms[0][0]

    def head(self, key, x, joints):
        self._p(key + ".weight", (joints, x, 1, 1))
        self._p(key + ".bias", (joints,))
        return joints


def registry(a: Arch) -> Registry:
    b = _RegistryBuilder()
    walk(b, a)
    return b.reg
