"""Static execution planner for the HRNet hot path on MI355X.

The network topology (``arch.walk``) is turned ONCE per (batch, resolution, dtype, mode) into two
flat lists of C-ABI kernel invocations with every device pointer resolved -- a forward program and
its hand-derived backward program -- which are then replayed eagerly or captured in a HIP graph.
No tracing compiler and no autograd tape at run time: the backward program is built here by a
reverse walk over the plan (``_build_backward``).

Data flow choices:
  * activations NHWC, dtype bf16 or fp32, fp32 accumulation, fp64 BatchNorm statistics;
  * a conv writes its RAW output plus per-channel sums; the BatchNorm (+ReLU) is applied by the
    consumer while it stages its input ("normalise on load"), so BN costs no HBM pass of its own;
  * residual adds / exchange sums are the only materialised activations (``fuse``);
  * backward mirrors it: the data-gradient conv masks with the ReLU and reduces the BatchNorm
    backward sums in its epilogue, and dgrad/wgrad apply the BatchNorm backward on load.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import capi
from .arch import Arch, Registry, registry, walk

EPS = 1e-5       # nn.BatchNorm2d default
MOMENTUM = 0.1   # reference HRnet.py:23 (fuse/transition BNs use the default, also 0.1)


def _esz(dtype: int) -> int:
    return 2 if dtype == capi.BF16 else 4


def choose_tile(B: int, Ho: int, Wo: int, stride: int, ks: int, esz: int, bn_cols: int = 64,
                maxpx: int = 128, maxhalo: int = 576) -> Tuple[int, int]:
    """Pick the output tile (TH virtual rows x TW columns, TH*TW <= 128) that wastes the fewest
    MFMA rows / halo loads while fitting LDS."""
    vrows = B * (Ho + 1)
    best, best_tile = -1.0, (1, min(Wo, maxpx))
    for tw in range(min(Wo, 4), min(Wo, maxpx) + 1):
        th = max(1, min(maxpx // tw, vrows))
        hr, hc = (th - 1) * stride + ks, (tw - 1) * stride + ks
        lds = hr * hc * 80 + bn_cols * (ks * ks * 64 + 16) + 8192
        if lds > 150 * 1024 or hr * hc > maxhalo:   # 576 halo pixels = 9 staging vectors per thread
            continue
        cols = math.ceil(Wo / tw) * tw
        rows = math.ceil(vrows / th) * th
        eff = (Wo / cols) * (B * Ho / rows) * (th * tw / float(maxpx))
        eff *= ((th * tw) / float(hr * hc) * stride * stride) ** 0.25  # mild halo penalty
        if eff > best:
            best, best_tile = eff, (th, tw)
    return best_tile


@dataclass
class BNInfo:
    idx: int
    C: int
    param_off: int   # gamma offset in master (beta at +C)
    buf_off: int     # running_mean offset in float buffers (running_var at +C)
    stats_off: int   # offset (doubles) in the stats / rstats arenas
    inv_count: float = 0.0


@dataclass
class ConvInfo:
    key: str
    Co: int
    Ci: int          # real input channels
    ks: int
    stride: int
    patch: bool
    master_off: int
    fwd_off: int = -1
    bwd_off: int = -1
    Cik: int = 0     # input channels as the kernel sees them (32 for the patch conv)


@dataclass
class Act:
    kind: str            # 'plain' | 'bn'
    t: torch.Tensor      # storage (flat uint8)
    B: int
    H: int
    W: int
    C: int
    bn: Optional[BNInfo] = None
    relu: bool = False
    needs_grad: bool = True
    grads: List[torch.Tensor] = field(default_factory=list)  # plain: gradient contributions
    dt: Optional[torch.Tensor] = None                        # bn: grad wrt BN output (masked)
    consumers: int = 0
    bwd_seen: int = 0                                        # consumers already handled by the backward builder
    fused_du: Optional[torch.Tensor] = None                  # plain: masked gradient produced by a fused dgrad
    pending: Optional[Tuple] = None                          # plain block-end sum not consumed yet: (forward op, BN term, skip term)

    @property
    def ptr(self) -> int:
        return self.t.data_ptr()


class ParamStore:
    """Flat fp32 master / grad / buffer storage laid out in state_dict order."""

    def __init__(self, reg: Registry, device):
        self.reg = reg
        self.param_off: Dict[str, int] = {}
        off = 0
        for k, s in reg.params:
            self.param_off[k] = off
            off += int(math.prod(s)) if s else 1
        self.nparam = off
        self.buf_off: Dict[str, int] = {}
        self.nbt_idx: Dict[str, int] = {}
        off, n = 0, 0
        for k, s in reg.buffers:
            if k.endswith("num_batches_tracked"):
                self.nbt_idx[k] = n
                n += 1
            else:
                self.buf_off[k] = off
                off += int(math.prod(s))
        self.nbuf, self.nnbt = off, n
        self.device = device
        self.master = torch.zeros(self.nparam, dtype=torch.float32, device=device)
        self.grads = torch.zeros(self.nparam, dtype=torch.float32, device=device)
        self.bufs = torch.zeros(self.nbuf, dtype=torch.float32, device=device)
        self.nbt = torch.zeros(self.nnbt, dtype=torch.int64, device=device)


class Engine:
    """One plan: fixed batch/resolution/dtype/mode."""

    def __init__(self, arch: Arch, store: ParamStore, B: int, H: int, W: int, dtype: int, training: bool):
        assert H % 32 == 0 and W % 32 == 0, "input H, W must be multiples of 32 (four stride-2 stages + 8x upsample)"
        self.arch, self.store, self.B, self.H, self.W = arch, store, B, H, W
        self.dtype, self.training = dtype, training
        self.dev = store.device
        self.esz = _esz(dtype)
        self.tdtype = torch.bfloat16 if dtype == capi.BF16 else torch.float32
        self.lib = capi.lib()
        self.fwd_ops: List[Tuple] = []
        self.bwd_ops: List[Tuple] = []
        self.tape: List[Tuple] = []
        self.convs: List[ConvInfo] = []
        self.bns: List[BNInfo] = []
        self._keep: List = []  # keep ctypes structs / tensors alive
        self._progs: Dict[int, Tuple] = {}
        self.act_bytes = 0
        self.generation = 0   # forward passes through this plan (hrnet._Fn stale-backward check)
        # static I/O
        self.img = torch.zeros(B, 3, H, W, dtype=torch.float32, device=self.dev)
        self.out: Optional[torch.Tensor] = None
        self.dout: Optional[torch.Tensor] = None
        # pass 1: count BN channels to size the statistics arenas
        nstat = sum(int(math.prod(s)) for k, s in store.reg.params if _is_bn_weight(k, store.reg)) * 2 * capi.NSHARD
        self.stats = torch.zeros(max(nstat, 1), dtype=torch.float64, device=self.dev)
        self.rstats = torch.zeros(max(nstat, 1), dtype=torch.float64, device=self.dev) if training else None
        self.nstreams = int(os.environ.get("STLPOSE_STREAMS", "4"))
        # weight-gradient streams: "0" = none (same stream as the branch), "1" = one per branch stream,
        # "n<k>" = k shared streams (branch s -> weight-gradient stream s % k)
        # "auto": no extra streams -- every off-chain launch (weight gradients, slab reductions) is list-scheduled
        # onto the branch stream that is free first (_balance_streams)
        wgs = os.environ.get("STLPOSE_WGRAD_STREAMS", "auto")
        self.wgrad_auto = wgs == "auto"
        self.wgrad_streams = wgs not in ("0", "auto")
        self.nwstreams = 0 if not self.wgrad_streams else (int(wgs[1:]) if wgs.startswith("n") else self.nstreams)
        self._stream = 0
        self._fwd_streams = max(1, min(self.nstreams, int(os.environ.get("STLPOSE_FWD_STREAMS", str(self.nstreams)))))   # streams of the FORWARD program only
        self._side = None
        self._chain_mask = bool(os.environ.get("STLPOSE_CUMASK_CHAIN", ""))
        self._stats_used = 0
        self._wk_elems = 0
        self._wk_fix: List[Tuple] = []
        walk(self, arch)
        self._finalize_weights()
        self._build_tables()
        if training:
            self._build_backward()
        if os.environ.get("STLPOSE_ISSUE_ORDER", "0") != "0":
            self.fwd_ops = self._issue_order(self.fwd_ops)

    # ------------------------------------------------------------------ allocation helpers
    def _alloc(self, nbytes: int) -> torch.Tensor:
        """Every planned buffer lives as long as the engine: kernels hold raw pointers, so a tensor
        that merely lost its last Python reference must never go back to the caching allocator."""
        self.act_bytes += nbytes
        t = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        self._keep.append(t)
        return t

    def _act_tensor(self, B, H, W, C) -> torch.Tensor:
        return self._alloc(B * H * W * C * self.esz)

    def _src(self, a: Act, relu: Optional[bool] = None) -> capi.Src:
        a.pending = None
        s = capi.Src()
        s.x = a.ptr
        if a.kind == "plain":
            s.mode = capi.SRC_PLAIN
            return s
        bn = a.bn
        s.mode = capi.SRC_BN
        s.relu = int(a.relu if relu is None else relu)
        st = self.store
        s.gamma = st.master.data_ptr() + 4 * bn.param_off
        s.beta = st.master.data_ptr() + 4 * (bn.param_off + bn.C)
        if self.training:
            s.stats = self.stats.data_ptr() + 8 * bn.stats_off
        else:
            s.rmean = st.bufs.data_ptr() + 4 * bn.buf_off
            s.rvar = st.bufs.data_ptr() + 4 * (bn.buf_off + bn.C)
        s.inv_count = bn.inv_count
        s.eps = EPS
        return s

    def _gsrc(self, y: Act) -> capi.Src:
        """Gradient of a conv's raw output, BatchNorm backward applied on load."""
        s = self._src(y)
        s.mode = capi.SRC_BNBWD
        s.x = y.dt.data_ptr()
        s.y = y.ptr
        s.rstats = self.rstats.data_ptr() + 8 * y.bn.stats_off
        return s

    # ------------------------------------------------------------------ builder protocol (arch.walk)
    def set_stream(self, s: int):
        self._stream = s % self.nstreams

    # opt-in: measured 18.4 -> 18.8 ms/step -- every chain moved to another stream pays a cross-stream event wait
    # (~6 us) at both ends, more than the serialisation it removes
    balance_exchange = os.environ.get("STLPOSE_BALANCE_EXCHANGE", "0") != "0"
    # block-end sums formed by the consuming conv1 (STL_SRC_BNADD).  Bit-identical to the two-launch form and time-neutral on
    # MI355X (all 69 eligible sums merged: 15.50 vs 15.33-15.43 ms per step; C <= 32 / C <= 64 / C >= 64 / C >= 128 only:
    # 15.53 / 15.51 / 15.45 / 15.36): the conv re-forms the sum for every halo pixel and every output-channel block, which
    # costs what the saved launch and tensor pass bought.  Default: the C >= 128 layers (STLPOSE_MERGE_MINC / _MAXC).
    merge_block_end = os.environ.get("STLPOSE_MERGE_BLOCK_END", "1") != "0"

    def est_cost(self, x: "Act", cout: int, ks: int, stride: int, hop: int = 0) -> float:
        """Estimated duration (us) of one conv of an exchange chain (arch._exchange_module): launch + latency chain
        plus input and output bytes at ~2 TB/s; hop k of a stride-2 chain sees a map 4^k times smaller."""
        h, w = x.H >> hop, x.W >> hop
        ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
        return 12.0 + (x.B * h * w * x.C + x.B * ho * wo * cout) * self.esz / 2.0e6

    def stem_input(self) -> Act:
        B, H, W = self.B, self.H, self.W
        Ho, Wo = H // 2, W // 2
        t = self._act_tensor(B, Ho, Wo, 32)
        pd = capi.Patch()
        pd.dtype, pd.B, pd.H, pd.W, pd.stride = self.dtype, B, H, W, 2
        pd.img, pd.out = self.img.data_ptr(), t.data_ptr()
        self.fwd_ops.append(("stl_patch3x3", pd, 0, [], [t.data_ptr()]))
        return Act("plain", t, B, Ho, Wo, 32, needs_grad=False)

    def conv_bn(self, ck, bk, x: Act, cout, ks, stride, relu, patch=False) -> Act:
        st = self.store
        real_ci = 3 if patch else x.C
        ci = ConvInfo(ck, cout, real_ci, ks, stride, patch, st.param_off[ck + ".weight"], Cik=x.C)
        kks, kstride = (1, 1) if patch else (ks, stride)
        pad = 1 if kks == 3 else 0
        Ho, Wo = (x.H + 2 * pad - kks) // kstride + 1, (x.W + 2 * pad - kks) // kstride + 1
        # kernel-layout weights: forward [Co][taps][Cik]; data-gradient [Cik][taps][Co]
        ci.fwd_off = self._wk_elems
        self._wk_elems += cout * kks * kks * x.C
        if x.needs_grad and self.training:
            ci.bwd_off = self._wk_elems
            self._wk_elems += cout * kks * kks * x.C
        self.convs.append(ci)
        bn = BNInfo(len(self.bns), cout, st.param_off[bk + ".weight"], st.buf_off[bk + ".running_mean"],
                    self._stats_used, 1.0 / float(x.B * Ho * Wo))
        self._stats_used += capi.NSHARD * 2 * cout
        self.bns.append(bn)
        y = Act("bn", self._act_tensor(x.B, Ho, Wo, cout), x.B, Ho, Wo, cout, bn=bn, relu=relu)
        p = capi.Conv()
        p.dtype = self.dtype
        p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co = x.B, x.H, x.W, x.C, Ho, Wo, cout
        p.ks, p.stride, p.stuff = kks, kstride, 0
        p.TH, p.TW, p.shape = 0, 0, -1
        capi.call("stl_conv_plan", C.byref(p))  # block shape + pixel tile, searched once
        p.grid_pct = self._grid_pct(ck)
        reads, writes = [x.ptr], [y.ptr]
        pend = x.pending
        if (pend is not None and self.merge_block_end and kks == 3 and kstride == 1 and pend[0][2] == self._stream % self._fwd_streams
                and int(os.environ.get("STLPOSE_MERGE_MINC", "128")) <= x.C <= int(os.environ.get("STLPOSE_MERGE_MAXC", "4096"))
                and capi.lib().stl_conv_bnadd_ok(C.byref(p)) == 1):
            # Residual block end z = ReLU(BN(y2) + skip) whose FIRST consumer is this 3x3 convolution (the next unit's
            # conv1): the sum is formed while the conv stages its tiles and written out once from the tile interiors
            # (STL_SRC_BNADD + src_out) -- the stand-alone sum launch and one pass over the tensor go (HRnet.py:58-59).
            fop, ybn, skip = pend
            self.fwd_ops.remove(fop)
            p.src = self._src(ybn, relu=True)
            p.src.mode = capi.SRC_BNADD
            p.src.y = skip.ptr
            p.src_out = x.ptr
            reads, writes = [ybn.ptr, skip.ptr], [y.ptr, x.ptr]
            x.pending = None
        else:
            p.src = self._src(x)
        p.out = y.ptr
        if self.training:
            p.out_stats = self.stats.data_ptr() + 8 * bn.stats_off
        self._wk_fix.append((p, "w", ci.fwd_off))
        self.fwd_ops.append(("stl_conv_forward", p, self._stream % self._fwd_streams, reads, writes))
        x.consumers += 1
        self.tape.append(("conv", x, y, ci, (kks, kstride), self._stream))
        return y

    def fuse(self, terms, relu) -> Act:
        terms = [(a, s, a.relu if a.kind == "bn" else False) for a, s in terms]
        if len(terms) == 1 and terms[0][2] and not relu:
            relu, terms = True, [(terms[0][0], terms[0][1], False)]  # relu(bn(y)) == fuse-level ReLU
        assert not any(tr for _, _, tr in terms), "ReLU inside a multi-term sum is not part of this network"
        base = [a for a, s, _ in terms if s == 0][0]
        B, H, W, Cc = base.B, base.H, base.W, base.C
        z = Act("plain", self._act_tensor(B, H, W, Cc), B, H, W, Cc)
        p = capi.Fuse()
        p.dtype, p.B, p.H, p.W, p.C, p.nterms, p.relu = self.dtype, B, H, W, Cc, len(terms), int(relu)
        for i, (a, s, tr) in enumerate(terms):
            assert a.C == Cc and a.H << s == H and a.W << s == W, "fuse: term shape mismatch"
            p.t[i].src = self._src(a, relu=tr)
            p.t[i].shift = s
            a.consumers += 1
        p.out = z.ptr
        op = ("stl_fuse_forward", p, self._stream % self._fwd_streams, [a.ptr for a, _, _ in terms], [z.ptr])
        self.fwd_ops.append(op)
        self.tape.append(("fuse", terms, z, relu, self._stream))
        if relu and len(terms) == 2 and all(s == 0 for _, s, _ in terms):
            bns = [a for a, _, tr in terms if a.kind == "bn" and not tr]
            pls = [a for a, _, _ in terms if a.kind == "plain"]
            if len(bns) == 1 and len(pls) == 1 and terms[0][0] is bns[0]:   # BN term first: same summation order as the sum kernel
                z.pending = (op, bns[0], pls[0])
        return z

    def head(self, key, x: Act, joints) -> torch.Tensor:
        st = self.store
        self.out = torch.zeros(x.B, joints, x.H, x.W, dtype=torch.float32, device=self.dev)
        self.head_w = st.master.data_ptr() + 4 * st.param_off[key + ".weight"]
        self.head_b = st.master.data_ptr() + 4 * st.param_off[key + ".bias"]
        hd = capi.Head()
        hd.dtype, hd.B, hd.H, hd.W, hd.Ci, hd.J = self.dtype, x.B, x.H, x.W, x.C, joints
        hd.x, hd.w, hd.bias, hd.out = x.ptr, self.head_w, self.head_b, self.out.data_ptr()
        self.fwd_ops.append(("stl_head_forward", hd, 0, [x.ptr], [self.out.data_ptr()]))
        x.consumers += 1
        self.tape.append(("head", x, key, joints))
        return self.out

    # ------------------------------------------------------------------ weights in kernel layout
    def _finalize_weights(self):
        self.wk = torch.zeros(max(self._wk_elems, 1), dtype=self.tdtype, device=self.dev)
        base = self.wk.data_ptr()
        for p, attr, off in self._wk_fix:
            setattr(p, attr, base + off * self.esz)
        tab = (capi.WPrep * len(self.convs))()
        blk = 0
        for i, c in enumerate(self.convs):
            e = tab[i]
            e.src_off, e.fwd_off, e.bwd_off = c.master_off, c.fwd_off, c.bwd_off
            e.Co, e.Ci, e.ks, e.Cip, e.patch, e.blk0 = c.Co, c.Ci, c.ks, c.Cik, int(c.patch), blk
            blk += math.ceil(c.Co * c.Ci * c.ks * c.ks / 1024)
        self._wprep_blocks = blk
        self._wprep_blk0 = [tab[i].blk0 for i in range(len(self.convs))] + [blk]
        self._wprep_tab = _to_device(tab, self.dev)
        self._wprep_n = len(self.convs)
        self.weights_ready = False   # set by a caller that has already run prep_weights_range for every bucket

    def _active_of(self, key: str) -> int:
        """Branch chains in flight around the layer `key` (its stage's branch count)."""
        if key.startswith("stage"):
            return int(key[5])
        if key.startswith("transition") and key[10] in "23":
            return int(key[10])
        return 1 if not key.startswith("final") else self.nstreams

    def _grid_pct(self, key: str) -> int:
        """STLPOSE_CAP_SCALE='p1,p2,p3,p4': persistent-grid size (per cent of the kernel's default) of a launch that
        runs beside 0 / 1 / 2 / 3 other branch chains."""
        spec = os.environ.get("STLPOSE_CAP_SCALE", "")
        if not spec:
            return 0
        pcts = [int(v) for v in spec.split(",")]
        return pcts[min(self._active_of(key), self.nstreams, len(pcts)) - 1]

    # ------------------------------------------------------------------ backward program
    def _new_grad(self, a: Act) -> torch.Tensor:
        return self._act_tensor(a.B, a.H, a.W, a.C)

    def _build_backward(self):
        st = self.store
        ops = self.bwd_ops
        self.slabs: List[Tuple] = []  # (struct_or_None, nelem, entry dict)
        self._slab_elems = 0
        J = self.out.shape[1]
        self.dout = torch.zeros_like(self.out)
        producer = {id(n[2]): n for n in self.tape if n[0] == "fuse"}
        fuse_block_end = os.environ.get("STLPOSE_FUSE_BLOCK_END", "1") != "0"
        # opt-in (STLPOSE_FUSED_BWD=1): measured on MI355X the fused launch is 36-38 us against 22 + 25 us for
        # the two stand-alone launches and moves fewer bytes, but it puts the weight-gradient work on the
        # data-gradient chain (the critical path) at two blocks per CU: 20.4 vs 19.7 ms per step
        fused_bwd = os.environ.get("STLPOSE_FUSED_BWD", "0") != "0"
        fused_c = os.environ.get("STLPOSE_FUSED_C", "32,64").split(",")
        # ---- gradient buckets: contiguous suffixes of the flat gradient buffer, closed as soon as every
        # parameter in them has its slabs / BatchNorm reductions complete (backward finishes the last
        # layers first).  Each bucket gets one ranged slab reduction + BN-gradient launch inside the
        # program, so the step has no serial tail, and an event a data-parallel all-reduce can wait on.
        self.buckets: List[dict] = []
        bucket_min = int(float(os.environ.get("STLPOSE_BUCKET_MB", "32")) * (1 << 20) / 4)   # 16 -> 32 MB: 16.79 -> 16.66 ms/step (fewer, larger reductions and collectives)
        bk = dict(done=0, lo=st.nparam, hi=st.nparam, slab0=0, reads=[], strm=0, wuse=[])
        # The serial tail of backward (layer1 + stem: one branch, 113 MB tensors) finishes last.  Close a bucket
        # where it begins, whatever its size, so that the final slab reduction (the only work left after the last
        # weight gradient, in front of the optimiser) covers just the stem / layer1 slabs instead of every layer
        # since the last 16 MB boundary.
        tail_keys = [k for k in os.environ.get("STLPOSE_BUCKET_TAIL", "transition1.0.0.weight,layer1.1.conv1.weight").split(",") if k]
        force_at = {st.param_off[k] for k in tail_keys if k in st.param_off}

        def bucket_add(off: int, size: int):
            bk["done"] += size
            bk["lo"] = min(bk["lo"], off)

        self._wg_pending: Dict[Tuple, List] = {}   # grouped weight gradients waiting for their group to fill
        # STLPOSE_WGRAD_DEFER=1: a full group is not issued where it fills (in the middle of its branch's data-gradient chain,
        # on that chain's own stream) but when the reverse walk leaves the branch: the chain reaches the exchange earlier and the
        # weight gradients run beside the exchange's small launches
        self._wg_ready: List[Tuple] = []
        self._wg_defer = os.environ.get("STLPOSE_WGRAD_DEFER", "0") != "0"
        # how many layers share each weight-gradient shape.  STLPOSE_WGRAD_COUNT=1 sizes a shape's groups by it, so that a shape
        # which occurs once (transition convs, the stem) gets the whole block budget instead of a quarter -- measured 0.04-0.06 ms
        # per step SLOWER (15.17 / 15.13 / 15.12 vs 15.11 / 15.09 / 15.07): these launches run beside the data-gradient chain of the
        # bandwidth-bound tail, and fewer blocks disturb it less.  Off by default.
        self._wg_count: Dict[Tuple, int] = {}
        for node in self.tape:
            if node[0] == "conv":
                _, x_, y_, _ci, (kk_, ss_), _s = node
                kkey = (x_.C, y_.C, kk_, ss_, x_.H, x_.W)
                self._wg_count[kkey] = self._wg_count.get(kkey, 0) + 1

        def bucket_close(force: bool = False):
            complete = bk["done"] == bk["hi"] - bk["lo"]          # suffix [lo, hi) fully covered
            force = force or (complete and bk["lo"] in force_at)
            if not complete or bk["done"] == 0 or (bk["done"] < bucket_min and not force):
                return
            assert complete
            self._flush_ready_groups(ops)
            for key in list(self._wg_pending):                     # the bucket's slab reduction reads every member's slabs
                self._flush_wgrad_group(ops, key, now=True)
            rr, br = capi.ReduceRange(), capi.BNRange()
            b = dict(lo=bk["lo"], hi=bk["hi"], slab0=bk["slab0"], slab1=len(self.slabs), rr=rr, br=br, wuse=list(bk["wuse"]))
            wstrm = bk["strm"]
            ops.append(("stl_reduce_slabs_range", rr, wstrm, list(bk["reads"]), [("bucket", len(self.buckets))]))
            ops.append(("stl_bn_grads_range", br, wstrm, [("bucket", len(self.buckets))], [("bucketbn", len(self.buckets))]))
            b["op"] = len(ops) - 1
            self.buckets.append(b)
            bk.update(done=0, hi=bk["lo"], slab0=len(self.slabs), reads=[], wuse=[])
        self._nactive: Dict[int, int] = {}   # id(desc) -> branch streams busy with the data-gradient chain around that op
        cur_active = self.nstreams

        active_of = self._active_of
        n_before = 0
        for node in reversed(self.tape):
            kind = node[0]
            for o in ops[n_before:]:
                self._nactive.setdefault(id(o[1]), cur_active)
            n_before = len(ops)
            if kind == "conv":
                cur_active = min(self.nstreams, active_of(node[3].key))
            bucket_close()   # after the previous node's ops: closes a bucket when a complete suffix is large enough
            if kind == "head":
                _, x, key, joints = node
                x.bwd_seen += 1
                nblk = max(1, min(int(os.environ.get("STLPOSE_HEAD_BLOCKS", "256")), math.ceil(x.B * x.H * x.W / 256)))   # <= one 256-pixel chunk per block
                dx = self._new_grad(x)
                nel = joints * x.C + joints
                part_off = self._slab_elems
                self._slab_elems += (nblk * nel + 3) // 4 * 4   # keep every entry 16-byte aligned
                hb = capi.HeadBwd()
                hb.dtype, hb.B, hb.H, hb.W, hb.Ci, hb.J, hb.nblk = self.dtype, x.B, x.H, x.W, x.C, joints, nblk
                hb.x, hb.w, hb.dout, hb.dx = x.ptr, self.head_w, self.dout.data_ptr(), dx.data_ptr()
                self._head_bwd_args = (hb, part_off)
                ops.append(("stl_head_backward", hb, 0, [self.dout.data_ptr(), x.ptr], [dx.data_ptr(), id(hb)]))
                bk["reads"].append(id(hb))
                bucket_add(st.param_off[key + ".weight"], joints * x.C)
                bucket_add(st.param_off[key + ".bias"], joints)
                x.grads.append(dx)
                self.slabs.append(dict(part_off=part_off, grad_off=st.param_off[key + ".weight"], nsplit=nblk,
                                       Co=joints, Ci=x.C, ks=1, Cip=x.C, patch=0, stride=nel))
                self.slabs.append(dict(part_off=part_off + joints * x.C, grad_off=st.param_off[key + ".bias"],
                                       nsplit=nblk, Co=joints, Ci=1, ks=1, Cip=1, patch=0, stride=nel))
            elif kind == "fuse":
                _, terms, z, relu, strm = node
                for a, _s, _ in terms:
                    a.bwd_seen += 1
                if z.fused_du is not None:
                    # the ReLU mask, the BatchNorm reductions and the sum of contributions were done in the
                    # epilogue of the data gradient that produced the last contribution (mask_z)
                    assert not z.grads
                    z.grads.append(z.fused_du)
                assert 1 <= len(z.grads) <= 4, f"fuse output has {len(z.grads)} gradient contributions"
                p = capi.FuseBwd()
                p.dtype, p.B, p.H, p.W, p.C = self.dtype, z.B, z.H, z.W, z.C
                p.ngrads, p.relu = len(z.grads), int(relu)
                for i, gt in enumerate(z.grads):
                    p.dz[i] = gt.data_ptr()
                p.z = z.ptr
                same_bn = [a for a, s, _ in terms if a.kind == "bn" and s == 0]
                p.nbn = len(same_bn)
                for i, a in enumerate(same_bn):
                    p.bn[i] = self._src(a)
                    p.rstats[i] = self.rstats.data_ptr() + 8 * a.bn.stats_off
                trivial = (len(z.grads) == 1 and not relu and not same_bn) or z.fused_du is not None
                du = z.grads[0] if trivial else self._new_grad(z)
                p.du = du.data_ptr()
                if not trivial:
                    ops.append(("stl_fuse_backward", p, strm, [gt.data_ptr() for gt in z.grads], [du.data_ptr()]))
                for a, s, _ in terms:
                    if a.kind == "plain":
                        assert s == 0, "upsampled plain terms do not occur in this network"
                        if a.needs_grad:
                            a.grads.append(du)
                    elif s == 0:
                        a.dt = du
                    else:
                        u = capi.UpBwd()
                        u.dtype, u.B, u.H, u.W, u.C, u.shift = self.dtype, a.B, a.H, a.W, a.C, s
                        u.du = du.data_ptr()
                        a.dt = self._new_grad(a)
                        u.dt = a.dt.data_ptr()
                        u.bn = self._src(a)
                        u.rstats = self.rstats.data_ptr() + 8 * a.bn.stats_off
                        ops.append(("stl_upsample_backward", u, strm, [du.data_ptr()], [a.dt.data_ptr()]))
            else:  # conv
                _, x, y, ci, (kks, kstride), strm = node
                x.bwd_seen += 1
                assert y.consumers == 1 and y.dt is not None, f"{ci.key}: BN activation must have exactly one consumer"
                if self._wg_ready and self._wg_ready[0][0][:6] != (x.C, y.C, kks, kstride, x.H, x.W):
                    self._flush_ready_groups(ops)      # the walk has left the branch whose groups are waiting
                g = self._gsrc(y)
                wstrm = (self.nstreams + strm % self.nwstreams) if self.wgrad_streams else strm
                # Fused backward (conv_core.hip, NCO > 0): the two-conv units' 3x3 stride-1 C -> C convolutions with
                # C = 32 / 64 -- the bandwidth-bound half of the network -- compute the weight gradient inside the
                # data-gradient launch: dt and y are fetched once for both, one launch instead of two.
                fuse_wg = (fused_bwd and kks == 3 and kstride == 1 and x.needs_grad and x.C == y.C and x.C % 32 == 0
                           and x.C <= 64 and x.H == y.H and x.W == y.W and str(x.C) in fused_c)
                if not fuse_wg:
                    self._emit_wgrad(ops, bk, x, y, ci, g, kks, kstride, strm, wstrm)
                bucket_add(ci.master_off, ci.Co * ci.Ci * ci.ks * ci.ks)
                bucket_add(y.bn.param_off, 2 * y.bn.C)   # gamma, beta of the BatchNorm behind this conv
                # ---- data gradient
                if not x.needs_grad:
                    continue
                d = capi.Conv()
                d.dtype = self.dtype
                d.B, d.Hi, d.Wi, d.Ci = y.B, y.H, y.W, y.C
                d.Ho, d.Wo, d.Co = x.H, x.W, x.C
                d.ks, d.stride, d.stuff = kks, 1, int(kstride == 2)
                d.TH, d.TW, d.shape = 0, 0, -1
                if fuse_wg:
                    d.partial = 1   # plan for the fused block shape; the slab pointer is patched in below
                capi.call("stl_conv_plan", C.byref(d))
                d.grid_pct = self._grid_pct(ci.key)
                d.src = g
                d.w = self.wk.data_ptr() + ci.bwd_off * self.esz
                dreads = [y.dt.data_ptr()]
                dwrites = []
                if fuse_wg:
                    nslab = x.C // 32
                    blocks = int(os.environ.get("STLPOSE_FUSED_BLOCKS", "512"))
                    npt = math.ceil(x.B * (x.H + 1) / d.TH) * math.ceil(x.W / d.TW)
                    d.wg_nsplit = max(8, min(blocks // nslab, math.ceil(npt / 8) * 8) // 8 * 8)
                    d.wg_h = self._src(x)
                    nel = y.C * 9 * x.C
                    part_off = self._slab_elems
                    self._slab_elems += (d.wg_nsplit * nel + 3) // 4 * 4
                    self.slabs.append(dict(part_off=part_off, grad_off=ci.master_off, nsplit=d.wg_nsplit, Co=ci.Co, Ci=ci.Ci,
                                           ks=ci.ks, Cip=ci.Cik, patch=int(ci.patch), stride=0, struct=d))
                    dreads.append(x.ptr)
                    dwrites.append(id(d))
                    bk["reads"].append(id(d))
                    bk["strm"] = wstrm
                if x.kind == "plain":
                    out = self._new_grad(x)
                    if x.grads:
                        ad = x.grads.pop()
                        d.addend = ad.data_ptr()
                        dreads.append(ad.data_ptr())
                    # Residual block end z = ReLU(BN(y) + skip): when this data gradient is the LAST
                    # contribution to dz, its epilogue also applies the ReLU mask and reduces the
                    # BatchNorm-backward sums, so no separate pass over dz / z / y is needed.
                    F = producer.get(id(x))
                    same_bn = [a for a, s_, _ in F[1] if a.kind == "bn" and s_ == 0] if F else []
                    if (fuse_block_end and F is not None and F[3] and len(same_bn) == 1 and not x.grads
                            and x.bwd_seen == x.consumers):
                        ybn = same_bn[0]
                        d.mask_z = x.ptr
                        d.mask_y = ybn.ptr
                        d.mask_bn = self._src(ybn, relu=False)
                        d.red = self.rstats.data_ptr() + 8 * ybn.bn.stats_off
                        dreads += [x.ptr, ybn.ptr]
                        x.fused_du = out
                    else:
                        x.grads.append(out)
                else:
                    assert x.dt is None
                    out = self._new_grad(x)
                    x.dt = out
                    d.mask_y = x.ptr
                    d.mask_bn = self._src(x)
                    d.red = self.rstats.data_ptr() + 8 * x.bn.stats_off
                d.out = out.data_ptr()
                # ("wuse", layer): this launch is the last reader of the layer's kernel-layout weights and BatchNorm
                # parameters in the step -- what an in-program optimiser / weight re-layout of the bucket waits for
                ops.append(("stl_conv_forward", d, strm, dreads, [out.data_ptr(), ("wuse", ci.master_off)] + dwrites))
                bk["wuse"].append(("wuse", ci.master_off))
        bucket_close(force=True)
        for o in ops[n_before:]:
            self._nactive.setdefault(id(o[1]), cur_active)
        assert bk["done"] == 0 and bk["hi"] == 0, "gradient buckets do not cover the parameter buffer"
        # slab arena + reduce table
        self.slab_arena = torch.zeros(max(self._slab_elems, 1), dtype=torch.float32, device=self.dev)
        base = self.slab_arena.data_ptr()
        hb, off = self._head_bwd_args
        hb.partial = base + 4 * off
        for s in self.slabs:
            if "struct" in s:
                s["struct"].partial = base + 4 * s["part_off"]
        tab = (capi.Slab * len(self.slabs))()
        blk = 0
        for i, s in enumerate(self.slabs):
            e = tab[i]
            e.part_off, e.grad_off, e.nsplit = s["part_off"], s["grad_off"], s["nsplit"]
            e.Co, e.Ci, e.ks, e.Cip, e.patch, e.blk0, e.pad = s["Co"], s["Ci"], s["ks"], s["Cip"], s["patch"], blk, s["stride"]
            blk += math.ceil(s["Co"] * s["Ci"] * s["ks"] * s["ks"] / 1024)
        self._slab_blocks, self._slab_n = blk, len(self.slabs)
        self._slab_tab = _to_device(tab, self.dev)
        blk0 = [tab[i].blk0 for i in range(len(self.slabs))] + [blk]
        bn_off = [b_.param_off for b_ in self.bns]            # forward (= ascending offset) order
        assert bn_off == sorted(bn_off)
        import bisect
        for b in self.buckets:
            rr, br = b["rr"], b["br"]
            rr.partials, rr.grads = self.slab_arena.data_ptr(), st.grads.data_ptr()
            rr.tab = self._slab_tab.data_ptr() + b["slab0"] * C.sizeof(capi.Slab)
            rr.n, rr.blk_base, rr.nblocks = b["slab1"] - b["slab0"], blk0[b["slab0"]], blk0[b["slab1"]] - blk0[b["slab0"]]
            i0, i1 = bisect.bisect_left(bn_off, b["lo"]), bisect.bisect_left(bn_off, b["hi"])
            br.rstats, br.grads = self.rstats.data_ptr(), st.grads.data_ptr()
            br.tab, br.n = self._bn_tab.data_ptr() + i0 * C.sizeof(capi.BNRec), i1 - i0
        ops = self._lag_wgrads(ops, int(os.environ.get("STLPOSE_WGRAD_LAG", "0")))
        self.bwd_ops = self._balance_streams(ops) if self.wgrad_auto else ops
        if os.environ.get("STLPOSE_ISSUE_ORDER", "0") != "0":
            self.bwd_ops = self._issue_order(self.bwd_ops)
        for b in self.buckets:   # bucket events are addressed by op index
            b["op"] = next(i for i, o in enumerate(self.bwd_ops) if o[1] is b["br"])

    def _op_cost_us(self, op) -> float:
        """Rough duration of a backward launch for the list scheduler: a fixed launch + latency-chain part plus its
        bytes at ~2 TB/s (what these launches achieve)."""
        name, d = op[0], op[1]
        esz = self.esz
        if name == "stl_conv_forward":
            src = d.B * d.Hi * d.Wi * d.Ci * (2 if d.src.mode == capi.SRC_BNBWD else 1)
            out = d.B * d.Ho * d.Wo * d.Co * (1 + bool(d.mask_y) + bool(d.addend) + bool(d.mask_z))
            extra = d.B * d.Ho * d.Wo * d.Co if d.partial else 0
            return 12.0 + (src + out + extra) * esz / 2.0e6 + (10.0 if d.partial else 0.0)
        if name == "stl_conv_wgrad":
            by = (d.B * d.Hi * d.Wi * d.Ci + d.B * d.Ho * d.Wo * d.Co * (2 if d.g.mode == capi.SRC_BNBWD else 1)) * esz
            return 16.0 + (by + 2.0 * d.nsplit * d.Co * d.Ci * d.ks * d.ks * 4) / 2.0e6
        if name == "stl_conv_wgrad_group":
            m = d.members[0]
            by = (m.B * m.Hi * m.Wi * m.Ci + m.B * m.Ho * m.Wo * m.Co * (2 if m.g.mode == capi.SRC_BNBWD else 1)) * esz
            return 16.0 + d.n * (by + 2.0 * m.nsplit * m.Co * m.Ci * m.ks * m.ks * 4) / 2.0e6
        if name == "stl_fuse_backward":
            return 8.0 + d.B * d.H * d.W * d.C * (d.ngrads + 2 + d.nbn) * esz / 3.0e6
        if name == "stl_upsample_backward":
            return 8.0 + d.B * d.H * d.W * d.C * ((1 << (2 * d.shift)) + 2) * esz / 3.0e6
        if name == "stl_head_backward":
            return 80.0
        if name == "stl_reduce_slabs_range":
            return 20.0 + 80.0 * d.nblocks / 1100.0
        return 5.0

    def _balance_streams(self, ops):
        """Static placement of the off-chain backward launches (weight gradients, slab reductions, BatchNorm
        gradients) onto the branch streams -- one hardware queue each.  (With extra weight-gradient streams two
        streams share a queue and every switch between them costs ~6 us: 475 such gaps per step in
        profiles/r02_trace_default_summary.txt; on its own branch stream a weight gradient delays the
        data-gradient chain.)  Where the network has fewer branches than streams -- stage 3, stage 2, and the
        single-branch tail (layer1, stem), 40 % of backward -- the idle queues take the off-chain work: each such
        launch goes to the least-loaded stream among the idle ones and its own, by accumulated estimated time.
        In the tail only TWO idle queues are used: its launches stream 113 MB tensors at 3-4 TB/s each, and three
        weight gradients beside the data-gradient chain slow every one of them down by more than the overlap
        buys (round 2, tail queues 3 / 2 / 1 / 0: 16.98 / 16.86 / 16.82 / 17.33 ms per step; round 3 with ungrouped
        tail launches 3 / 2 / 1: 14.87 / 14.72 / 14.82)."""
        acc = [0.0] * self.nstreams
        out = []
        for op in ops:
            name, desc, strm, reads, writes = op
            cost = self._op_cost_us(op)
            if name in ("stl_conv_wgrad", "stl_conv_wgrad_group", "stl_reduce_slabs_range", "stl_bn_grads_range"):
                active = self._nactive.get(id(desc), self.nstreams)
                own = strm % self.nstreams
                if os.environ.get("STLPOSE_BALANCE", "idle") == "idle":
                    cands = list(range(active, self.nstreams)) + [own]
                    nq = int(os.environ.get("STLPOSE_OFFCHAIN_QUEUES", "0"))   # 0 = every idle queue
                    tq = int(os.environ.get("STLPOSE_TAIL_QUEUES", "2"))       # same, for the single-branch tail only (0 = all three); with ungrouped tail launches 1 / 2 / 3: 14.82 / 14.72 / 14.87
                    if active == 1 and tq:
                        cands = list(range(1, min(1 + tq, self.nstreams))) + [own]
                    elif nq:
                        cands = list(range(active, min(active + nq, self.nstreams))) + [own]
                else:   # any stream: a foreign weight gradient is issued behind that branch's ops of the module and
                    cands = range(self.nstreams)   # only waits for a data gradient the next exchange needs anyway
                strm = min(cands, key=lambda s_: (acc[s_], s_ != own))
            acc[strm] += cost
            out.append((name, desc, strm, reads, writes))
        self.sched_estimate_us = list(acc)
        return out

    @staticmethod
    def _lag_wgrads(ops, lag: int):
        """Issue every stand-alone weight gradient `lag` launches LATER than the reverse walk emits it.
        A weight gradient is emitted right behind the data gradient that produces its input, i.e. while
        that producer is still running; its event wait then parks at the head of a hardware queue that it
        shares with another branch's data-gradient stream (8 streams -> 4 queues) and blocks that branch.
        Issued a few launches later the wait is already satisfied when the packet reaches the queue head.
        A weight gradient never moves past the slab reduction that reads it."""
        if lag <= 0:
            return ops
        out, pending = [], []   # pending: [remaining, op]
        for op in ops:
            if op[0] == "stl_conv_wgrad":
                pending.append([lag, op])
                continue
            if op[0] == "stl_reduce_slabs_range":   # its reads include the ids of the bucket's weight gradients
                need = {r for r in op[3] if not isinstance(r, tuple)}
                keep = []
                for item in pending:
                    if any(w in need for w in item[1][4]):
                        out.append(item[1])
                    else:
                        keep.append(item)
                pending = keep
            out.append(op)
            keep = []
            for item in pending:
                item[0] -= 1
                if item[0] <= 0:
                    out.append(item[1])
                else:
                    keep.append(item)
            pending = keep
        out += [item[1] for item in pending]
        return out

    def _emit_wgrad(self, ops, bk, x: Act, y: Act, ci: ConvInfo, g, kks: int, kstride: int, strm: int, wstrm: int):
        """Stand-alone weight-gradient launch of one convolution (split-K slabs) on its own stream: weight
        gradients are off the critical path (only the data-gradient chain is), so they overlap with the chain."""
        wg = capi.Wgrad()
        wg.dtype = self.dtype
        wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co = x.B, x.H, x.W, x.C, y.H, y.W, y.C
        wg.ks, wg.stride = kks, kstride
        ctile = capi.lib().stl_wgrad_chunk(C.byref(wg))   # 32, or 64 for the wide-channel variant
        want256 = os.environ.get("STLPOSE_WGRAD_TILE", "128") == "256"
        if ctile == 64 and kks == 1:
            want256 = os.environ.get("STLPOSE_WGRAD_K1_TILE", "128") == "256"
        big = (self.esz == 2 and kstride == 1 and want256
               and x.B * y.H * y.W >= 256 * 64 and (ctile == 32 or kks == 1))
        if big:   # 256-pixel tiles: fewer barriers per pixel, but one block per CU
            wg.TH, wg.TW = choose_tile(x.B, y.H, y.W, kstride, kks, self.esz, bn_cols=32, maxpx=256,
                                       maxhalo=352 if ctile == 64 else 384)
        else:
            wg.TH, wg.TW = choose_tile(x.B, y.H, y.W, kstride, kks, self.esz, bn_cols=32,
                                       maxhalo=192 if ctile == 64 else 576)
        npt = math.ceil(x.B * (y.H + 1) / wg.TH) * math.ceil(y.W / wg.TW)
        chunks = math.ceil(y.C / ctile) * math.ceil(x.C / ctile)
        # Block budget per launch: one block per CU.  (End to end, budgets of 128..256 measure the same within
        # run-to-run noise on MI355X, before and after the kernel was rebuilt: fewer blocks mean fewer slab bytes
        # but a longer launch; 96 and 384 are clearly worse.)
        budget = int(os.environ.get("STLPOSE_WGRAD_BLOCKS", "512"))   # round 3, grouped launches: 128 / 256 / 512 blocks = 17.65 / 16.03 / 15.36 ms per step (two 8-wave blocks per CU hide each other's tile latency)
        if ctile == 64:
            budget = int(os.environ.get("STLPOSE_WGRAD_BLOCKS64", "512"))
        elif kks == 1 and self.esz == 2:
            budget = int(os.environ.get("STLPOSE_WGRAD_BLOCKS_K1", str(budget)))
        # The single-branch tail of backward (layer1, stem, transition1) is one serial data-gradient chain beside three idle
        # queues: a group there fills only when the chain has walked through ALL its members (every layer1 group completes
        # at the first block, i.e. at the very end of the step), so grouped weight gradients pile up behind the chain.
        tail = self._active_of(ci.key) == 1
        gmax = capi.WGRAD_GROUP_MAX
        if tail:
            budget = int(os.environ.get("STLPOSE_TAIL_BLOCKS", str(budget)))
            gmax = int(os.environ.get("STLPOSE_TAIL_GROUP", "1"))   # 4 / 2 / 1: 15.02 / 14.81 / 14.82 ms per step; with two tail queues 14.73 / 14.72
        top = max(1, min(npt, budget // chunks if chunks <= budget else 1))
        wg.nsplit = min(range(1, top + 1), key=lambda ns: (math.ceil(npt / ns) + 0.004 * ns * chunks / 8, ns))
        wg.h = self._src(x)
        wg.g = g
        # Grouped launches (stl_conv_wgrad_group): weight gradients of one shape -- the 3x3 convolutions of a branch --
        # wait until `gsize` of them are ready and go out as ONE launch that shares the block budget: the four hardware
        # queues carry one off-chain launch instead of gsize (in stages 3 / 4 every queue is busy with a data-gradient
        # chain and each stand-alone weight gradient costs its chain a full launch latency, whatever its size), every
        # block walks gsize times as many pixel tiles, and gsize times fewer split-K slabs are written and reduced.
        gsize = 1
        if (ctile == 32 or os.environ.get("STLPOSE_WGRAD_GROUP64", "1") != "0") and not big and os.environ.get("STLPOSE_SKIP_WGRAD", "0") == "0":
            gsize = max(1, min(int(os.environ.get("STLPOSE_WGRAD_GROUP", "4")), gmax, max(1, budget // chunks),
                               self._wg_count.get((x.C, y.C, kks, kstride, x.H, x.W), 1) if os.environ.get("STLPOSE_WGRAD_COUNT", "0") != "0" else 99))
        if gsize > 1:
            bg = budget // gsize
            top = max(1, min(npt, bg // chunks if chunks <= bg else 1))
            wg.nsplit = min(range(1, top + 1), key=lambda ns: (math.ceil(npt / ns) + 0.004 * ns * chunks / 8, ns))
        nel = y.C * kks * kks * x.C
        part_off = self._slab_elems
        self._slab_elems += (wg.nsplit * nel + 3) // 4 * 4
        self.slabs.append(dict(part_off=part_off, grad_off=ci.master_off, nsplit=wg.nsplit, Co=ci.Co, Ci=ci.Ci,
                               ks=ci.ks, Cip=ci.Cik, patch=int(ci.patch), stride=0, struct=wg))
        bk["reads"].append(id(wg))
        bk["strm"] = wstrm
        if os.environ.get("STLPOSE_SKIP_WGRAD", "0") != "0":   # calibration only: no weight-gradient launches, slabs stay zero
            return
        if gsize == 1:
            ops.append(("stl_conv_wgrad", wg, wstrm, [y.dt.data_ptr(), x.ptr], [id(wg)]))
            return
        key = (x.C, y.C, kks, kstride, x.H, x.W, wg.TH, wg.TW, wg.nsplit, int(g.mode), gsize)
        pend = self._wg_pending.setdefault(key, [])
        pend.append((wg, [y.dt.data_ptr(), x.ptr], wstrm))
        if len(pend) >= gsize:
            self._flush_wgrad_group(ops, key)

    def _flush_ready_groups(self, ops):
        ready, self._wg_ready = self._wg_ready, []
        for key, pend in ready:
            self._wg_pending[key] = pend
            self._flush_wgrad_group(ops, key, now=True)

    def _flush_wgrad_group(self, ops, key, now: bool = False):
        if self._wg_defer and not now:
            self._wg_ready.append((key, self._wg_pending.pop(key)))
            return
        pend = self._wg_pending.pop(key, [])
        if not pend:
            return
        if len(pend) == 1:
            wg, reads, wstrm = pend[0]
            ops.append(("stl_conv_wgrad", wg, wstrm, reads, [id(wg)]))
            return
        grp = capi.WgradGroup()
        grp.n = len(pend)
        for i, (wg, _r, _s) in enumerate(pend):
            grp.p[i] = C.pointer(wg)
        self._keep.append([wg for wg, _r, _s in pend])
        grp.members = [wg for wg, _r, _s in pend]   # python-side view (bench, tools)
        ops.append(("stl_conv_wgrad_group", grp, pend[-1][2], [r for _w, rs, _s in pend for r in rs], [id(wg) for wg, _r, _s in pend]))

    def _build_tables(self):
        tab = (capi.BNRec * len(self.bns))()
        for i, b in enumerate(self.bns):
            e = tab[i]
            e.stats_off, e.param_off, e.buf_off, e.C, e.inv_count = b.stats_off, b.param_off, b.buf_off, b.C, b.inv_count
        self._bn_tab = _to_device(tab, self.dev)
        # num_batches_tracked entries are in registry (state_dict) order == self.bns order
        assert len(self.bns) == self.store.nnbt

    # ------------------------------------------------------------------ execution
    def _schedule(self, ops):
        """Cross-stream RAW dependencies: op index -> indices it must wait for / whether it records.  A wait is
        dropped when the consumer's stream already knows the producer to be complete -- directly (an earlier wait on
        the same or a later op of that stream) or transitively (vector clocks: 41 of 183 waits of the W32 backward)."""
        last, waits, need = {}, [], set()
        ns = max(o[2] for o in ops) + 1 if ops else 1
        clock = [[-1] * ns for _ in range(ns)]   # clock[s][t]: latest op of stream t known complete at this point of stream s
        snap = {}
        prune = os.environ.get("STLPOSE_PRUNE_WAITS", "1") != "0"
        for i, (_, _, st_, reads, writes) in enumerate(ops):
            w = set()
            for r in reads:
                j = last.get(r)
                if j is not None and ops[j][2] != st_:
                    w.add(j)
            latest = {}
            for j in w:                      # streams are in-order: the latest producer per stream covers the others
                latest[ops[j][2]] = max(latest.get(ops[j][2], -1), j)
            w = set()
            for t, j in latest.items():
                if prune and clock[st_][t] >= j:
                    continue                 # already ordered behind it
                w.add(j)
                for u in range(ns):
                    clock[st_][u] = max(clock[st_][u], snap[j][u])
            clock[st_][st_] = i
            snap[i] = list(clock[st_])
            need.update(w)
            waits.append(sorted(w))
            for t in writes:
                last[t] = i
        need.update(b["op"] for b in getattr(self, "buckets", []) if ops is self.bwd_ops)
        return waits, need

    def _issue_order(self, ops):
        """Host issue order = the order in which the launches can START on the device: an in-order replay of the plan's
        streams with estimated durations (dependencies as in _schedule), launches sorted by their simulated start time.
        The planner emits a module's branches one after the other (forward: branch 0 first; backward: branch 3 first), so
        a host that is not far ahead of the device feeds one queue while the others wait for their first launch of the
        module.  Per-stream order and every read-after-write dependency are preserved (tensors are written once per pass)."""
        last, sfin, fin, start = {}, {}, [], []
        for i, op in enumerate(ops):
            name, desc, st_, reads, writes = op
            t0 = sfin.get(st_, 0.0)
            for r in reads:
                j = last.get(r)
                if j is not None:
                    t0 = max(t0, fin[j])
            d = self._op_cost_us(op) if name != "stl_conv_forward" or True else 0.0
            start.append(t0)
            fin.append(t0 + d)
            sfin[st_] = t0 + d
            for w in writes:
                last[w] = i
        order = sorted(range(len(ops)), key=lambda i: (start[i], i))
        # safety: a dependency must never be issued after its consumer
        pos = {i: k for k, i in enumerate(order)}
        last = {}
        for i, (_, _, st_, reads, writes) in enumerate(ops):
            for r in reads:
                j = last.get(r)
                assert j is None or pos[j] < pos[i], "issue order breaks a dependency"
            for w in writes:
                last[w] = i
        return [ops[i] for i in order]

    def _program(self, ops):
        """Compile an op list into a native program (csrc/program.hip), once."""
        key = id(ops)
        prog = self._progs.get(key)
        if prog is None:
            waits, need = self._schedule(ops)
            arr = (capi.Op * max(len(ops), 1))()
            for i, (name, desc, st_, _, _) in enumerate(ops):
                o = arr[i]
                o.kind, o.stream, o.desc = capi.OP_KIND[name], st_, C.addressof(desc)
                assert len(waits[i]) <= 8, "op waits on more than 8 producers"
                o.nwait = len(waits[i])
                for j, wv in enumerate(waits[i]):
                    o.wait[j] = wv
                o.record = int(i in need)
            if self._chain_mask:   # every op on a masked stream of its own; streams[0] (the caller's) only forks / joins
                for i in range(len(ops)):
                    arr[i].stream += 1
            h = C.c_void_p()
            capi.call("stl_program_create", arr, len(ops), self.total_streams + (1 if self._chain_mask else 0), C.byref(h))
            prog = self._progs[key] = (h, arr)
        return prog[0]

    @property
    def total_streams(self) -> int:
        return self.nstreams + self.nwstreams

    def _run(self, ops, stream: int):
        """Replay a program natively.  With several streams the independent branches of each
        exchange module (and, in backward, the weight gradients) run concurrently; fork/join and
        cross-stream dependencies are HIP events inside stl_program_run."""
        h = self._program(ops)
        if self._side is None:
            self._make_streams()
        self._stream_arr[0] = stream
        rc = self.lib.stl_program_run(h, self._stream_arr)
        if rc != 0:
            raise RuntimeError(f"stl_program_run: {self.lib.stl_last_error().decode()}")

    @staticmethod
    def _cu_range(spec: str):
        """'lo:hi' -> ctypes uint32[8] with bits lo..hi-1 set (256 CUs), or None for an empty spec."""
        if not spec:
            return None
        lo, hi = (int(v) for v in spec.split(":"))
        assert 0 <= lo < hi <= 256, f"CU range {spec!r} outside 0..256"
        words = (C.c_uint32 * 8)()
        for i in range(lo, hi):
            words[i >> 5] |= 1 << (i & 31)
        return words

    def _make_streams(self):
        """HIP streams of the program: index 0 is the caller's stream; 1 .. nstreams-1 the other branch streams;
        nstreams .. the off-chain (weight-gradient) streams when STLPOSE_WGRAD_STREAMS asks for them.
        STLPOSE_CUMASK_OFF='lo:hi' restricts the off-chain streams to those compute units, STLPOSE_CUMASK_CHAIN='lo:hi'
        the branch streams (then ALL ops run on masked streams and the caller's stream only forks / joins)."""
        off = 1 if self._chain_mask else 0
        n = self.total_streams + off
        self._stream_arr = (C.c_void_p * n)()
        self._side = []
        m_off = self._cu_range(os.environ.get("STLPOSE_CUMASK_OFF", ""))
        m_chain = self._cu_range(os.environ.get("STLPOSE_CUMASK_CHAIN", ""))
        # ONE set of side streams per device, shared by every engine (engines never run concurrently): each new HIP
        # stream is another hardware queue, queues are spread round-robin over the four compute pipes, and two ACTIVE
        # queues on one pipe are time-sliced -- a second engine with streams of its own ran its plan at half speed
        # (W32 256x192 as the second plan of a process: 20.8 ms per step instead of 10.2).
        pool = _STREAM_POOL.setdefault((self.dev.index, os.environ.get("STLPOSE_CUMASK_OFF", ""), os.environ.get("STLPOSE_CUMASK_CHAIN", "")), {})
        with torch.cuda.device(self.dev):
            for i in range(1, n):
                op_stream = i - off                     # index in the planner's numbering
                mask = m_chain if op_stream < self.nstreams else m_off
                key = (i, mask is not None)
                if key not in pool:
                    if mask is not None:
                        h = C.c_void_p()
                        capi.call("stl_stream_create_masked", mask, 8, C.byref(h))
                        pool[key] = (h, h.value)        # lives as long as the process
                    else:
                        # STLPOSE_STREAM_PRIO="p1,p2,p3": HIP priority of side stream 1, 2, 3 (0 normal, -1 high): experiment
                        pr = [int(v) for v in os.environ.get("STLPOSE_STREAM_PRIO", "").split(",") if v]
                        s_ = torch.cuda.Stream(device=self.dev, priority=pr[i - 1] if i - 1 < len(pr) else 0)
                        pool[key] = (s_, s_.cuda_stream)
                self._side.append(pool[key][0])
                self._stream_arr[i] = pool[key][1]

    def prep_weights(self, stream: int):
        st = self.store
        capi.call("stl_weight_prep", self.dtype, st.master.data_ptr(), self.wk.data_ptr(), self._wprep_tab.data_ptr(),
                  self._wprep_n, self._wprep_blocks, stream)

    def prep_weights_range(self, i: int, stream: int):
        """weights -> kernel layout for the convolutions whose parameters lie in gradient bucket i."""
        b = self.buckets[i]
        if "conv0" not in b:
            import bisect
            offs = [c.master_off for c in self.convs]          # forward order == ascending offset
            assert offs == sorted(offs)
            b["conv0"], b["conv1"] = bisect.bisect_left(offs, b["lo"]), bisect.bisect_left(offs, b["hi"])
        i0, i1 = b["conv0"], b["conv1"]
        if i1 > i0:
            st = self.store
            capi.call("stl_weight_prep_range", self.dtype, st.master.data_ptr(), self.wk.data_ptr(),
                      self._wprep_tab.data_ptr() + i0 * C.sizeof(capi.WPrep), i1 - i0, self._wprep_blk0[i0],
                      self._wprep_blk0[i1] - self._wprep_blk0[i0], stream)

    def forward(self, stream: int, update_running: bool = True):
        """weights -> kernel layout, zero statistics, forward program, running-stat update."""
        self.generation += 1   # every pass overwrites the plan's activations (hrnet._Fn stale-backward check)
        if not self.weights_ready:
            self.prep_weights(stream)
        self.weights_ready = False
        if self.training:
            self.stats.zero_()
        self._run(self.fwd_ops, stream)
        if self.training and update_running:
            st = self.store
            capi.call("stl_bn_running_update", self.stats.data_ptr(), st.bufs.data_ptr(), st.nbt.data_ptr(),
                      self._bn_tab.data_ptr(), len(self.bns), MOMENTUM, stream)

    def backward(self, stream: int, fused_optim: bool = False):
        """expects self.dout filled; leaves dL/dparam in store.grads (overwrites).  fused_optim: the program of
        attach_optimizer (optimiser + next step's weight layouts inside backward)."""
        assert self.training
        st = self.store
        self.rstats.zero_()
        self._run(self.bwd_ops_opt if fused_optim else self.bwd_ops, stream)   # includes the per-bucket slab reductions and BatchNorm gradients

    def attach_optimizer(self, kind: int, p: int, g: int, m: int, v: int, hyper: int, step: int):
        """Single-process training: the optimiser slice and the next step's kernel-layout weights of every gradient bucket
        become ops of the backward program (``backward(stream, fused_optim=True)``), issued one bucket late on the stream
        of that later bucket's reductions -- by then the data gradients that still read the bucket's weights / BatchNorm
        parameters (the ("wuse", layer) tokens both ops wait for) have long finished.  The step then has no serial
        optimiser (0.13 ms) and weight re-layout (0.10 ms at the start of the next forward) section; no fifth stream is
        involved (a fifth active hardware queue is time-sliced).  kind: 0 Adam, 1 SGD."""
        ops = list(self.bwd_ops)
        at = sorted((b["op"], i) for i, b in enumerate(self.buckets))
        self._optim_descs = []
        new_ops, prev = [], None

        def emit(i, strm):
            b = self.buckets[i]
            o = capi.OptimSlice()
            lo, n = b["lo"], b["hi"] - b["lo"]
            o.kind, o.p, o.g, o.m, o.v, o.n, o.hyper, o.step = kind, p + 4 * lo, g + 4 * lo, m + 4 * lo, (v + 4 * lo) if v else 0, n, hyper, step
            new_ops.append(("stl_optim_slice", o, strm, [("bucketbn", i)] + b["wuse"], [("optim", i)]))
            self._optim_descs.append(o)
            if "conv0" not in b:
                import bisect
                offs = [c.master_off for c in self.convs]
                b["conv0"], b["conv1"] = bisect.bisect_left(offs, b["lo"]), bisect.bisect_left(offs, b["hi"])
            i0, i1 = b["conv0"], b["conv1"]
            if i1 > i0:
                w = capi.WPrepRange()
                w.dtype, w.n, w.blk_base, w.nblocks = self.dtype, i1 - i0, self._wprep_blk0[i0], self._wprep_blk0[i1] - self._wprep_blk0[i0]
                w.master, w.wk, w.tab = self.store.master.data_ptr(), self.wk.data_ptr(), self._wprep_tab.data_ptr() + i0 * C.sizeof(capi.WPrep)
                new_ops.append(("stl_wprep_range", w, strm, [("optim", i)], [("wprep", i)]))
                self._optim_descs.append(w)

        nxt = dict(at)
        # STLPOSE_OPTIM_STREAM: stream of the optimiser ops; default = the stream of the later bucket's reductions; the last
        # branch stream is the least loaded queue of the plan (busy 4.4 of 15 ms) and idle in the single-branch tail
        fixed = os.environ.get("STLPOSE_OPTIM_STREAM", "")
        for idx, op in enumerate(ops):
            new_ops.append(op)
            if idx in nxt:
                if prev is not None:
                    emit(prev, int(fixed) if fixed else op[2])
                prev = nxt[idx]
        if prev is not None:
            emit(prev, int(fixed) if fixed else ops[self.buckets[prev]["op"]][2])
        self.bwd_ops_opt = new_ops

    def bucket_wait(self, i: int, stream: int):
        """Make `stream` wait until gradient bucket i (self.buckets[i]: flat slice [lo, hi)) of the
        backward pass enqueued last is final."""
        capi.call("stl_program_wait_op", self._program(self.bwd_ops), self.buckets[i]["op"], stream)


_STREAM_POOL: Dict[Tuple, Dict] = {}   # (device, mask specs) -> {(stream index, masked): (owner object, hipStream_t)}


def _is_bn_weight(key: str, reg: Registry) -> bool:
    return key in _bn_weight_keys(reg)


_bn_cache: Dict[int, set] = {}


def _bn_weight_keys(reg: Registry) -> set:
    s = _bn_cache.get(id(reg))
    if s is None:
        s = {k[: -len("running_mean")] + "weight" for k, _ in reg.buffers if k.endswith("running_mean")}
        _bn_cache[id(reg)] = s
    return s


def _to_device(ctab, device) -> torch.Tensor:
    raw = bytes(ctab)
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
    return t.to(device)
