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
    return 4 if (dtype & 0xff) == capi.F32 else 2


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
        # dtype: capi.F32, capi.BF16 or capi.MIXED (= dt2(BF16, F16)).  self.dtype = element type of the GRADIENT tensors (and, in
        # the pure modes, of everything), self.fdtype = of the FORWARD tensors (raw conv outputs, sums, forward kernel-layout
        # weights): f16 in the mixed mode -- BatchNorm bounds their range, and 10 mantissa bits instead of 7 cut the distance
        # from the fp32 reference; gradients have no such bound and stay bf16.
        self.dtype_code = dtype
        self.dtype, self.fdtype = dtype & 0xff, ((dtype >> 8) & 0xff) or (dtype & 0xff)
        self.ydtype = self.fdtype if self.fdtype != self.dtype else 0   # what the backward descriptors carry (0 = same)
        self.training = training
        self.dev = store.device
        self.esz = _esz(dtype)
        self.tdtype = torch.float32 if self.dtype == capi.F32 else torch.bfloat16   # kernel-layout weights: just a byte container in the 16-bit modes
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
        nstat = bn_stat_elems(store.reg)
        self.stats = torch.zeros(max(nstat, 1), dtype=torch.float64, device=self.dev)
        self.rstats = torch.zeros(max(nstat, 1), dtype=torch.float64, device=self.dev) if training else None
        self.overflow = torch.full((1,), 2 ** 31 - 1, dtype=torch.int32, device=self.dev)   # range guard, see check_forward_range
        # ---- the planner's knobs (all of them; INTEGRATION.md lists what each is for and what was measured)
        env = os.environ.get
        self.nstreams = int(env("STLPOSE_STREAMS", "4"))          # HIP streams = hardware queues of the plan (4 compute pipes)
        # block-end sums z = ReLU(BN(y) + skip) formed by the consuming conv1 (STL_SRC_BNADD) for layers of at least this many
        # channels.  Bit-identical to the two-launch form and time-neutral on MI355X (round 3, all 69 eligible sums merged:
        # 15.50 vs 15.33-15.43 ms per step; C <= 32 / C <= 64 / C >= 64 / C >= 128 only: 15.53 / 15.51 / 15.45 / 15.36): the conv
        # re-forms the sum for every halo pixel and every output-channel block, which costs what the saved launch bought.
        self.merge_minc = int(env("STLPOSE_MERGE_MINC", "128"))
        self.bucket_mb = float(env("STLPOSE_BUCKET_MB", "64"))    # gradient bucket size (16 -> 32 MB: 16.79 -> 16.66 ms per step in round 2; 32 -> 64: 13.50 -> 13.42 in round 5)
        # block budget of a weight-gradient launch (a group shares it): 512 = two 8-wave blocks per CU, which hide each other's
        # tile latency (round 3, grouped launches: 128 / 256 / 512 / 768 blocks = 17.65 / 16.03 / 15.36 / 15.57 ms per step)
        # Weight-gradient blocks per launch.  Round 5: 256 / 128 instead of 512 / 256 -- ONE 8-wave block (128 VGPRs) per CU, resp.
        # 16-wave blocks (a whole CU's registers each) on HALF the CUs, so that the data-gradient chain always finds room: a
        # persistent weight-gradient block holds its CU for the whole launch (45 - 70 us), and beside 500 / 256 of them a data
        # gradient ran 2.3 - 4.5 x its own time while the weight gradient lost 3 - 17 % (tools/pair_probe.py,
        # profiles/r05_pair_*.txt); beside 256 / 128 it runs 1.45 - 1.6 x.  Step: 13.30 -> 13.12 ms.
        # (16-bit plans; the fp32 kernels' blocks are half as many waves per CU to begin with: 512 stays -- 48.6 vs 50.1 ms per step)
        self.wgrad_blocks = int(env("STLPOSE_WGRAD_BLOCKS", "256" if self.esz == 2 else "512"))
        self.wgrad_blocks_wide = int(env("STLPOSE_WGRAD_BLOCKS_WIDE", str(self.wgrad_blocks // 2)))
        # members per grouped launch (round 3 at 256 blocks: 1 / 2 / 4 / 8 = 15.87 / 15.64 / 16.03 / 17.51; 4 at 512 blocks: 15.36)
        self.wgrad_group = int(env("STLPOSE_WGRAD_GROUP", "4"))
        # the 64-channel 3x3 blocks (16 waves, one per CU: 256 per launch): eight layers per launch keep the split-K slabs at the
        # 32-channel kernel's size (a slab is the whole [Co][9][Ci] filter bank; 256 blocks over four layers would double them)
        self.wgrad_group_wide = int(env("STLPOSE_WGRAD_GROUP_WIDE", "8"))
        self.skip_wgrad = env("STLPOSE_SKIP_WGRAD", "0") != "0"   # calibration only (wrong numerics): no weight-gradient launches
        # STLPOSE_GRAPH=1: replay each program as ONE explicit HIP graph (csrc/program.hip: kernel nodes + the planner's
        # dependencies) instead of launches and events on four streams
        # Replay as ONE explicit HIP graph (stl_program_graph_build) instead of launches + events: slower for training plans (16.78 vs
        # 14.53 ms per step, round 4) and for inference plans (4.37 vs 4.20 ms per W32 384x288 bs 32 forward, round 5) -- opt-in.
        self.graph_mode = env("STLPOSE_GRAPH", "0") != "0"
        self._graphs = set()
        self._poison = env("STLPOSE_POISON", "0") != "0"          # debug: planned buffers start as NaNs (see _alloc)
        self._stream = 0
        self._side = None
        self._stats_used = 0
        self._wk_elems = 0
        self._wk_fix: List[Tuple] = []
        walk(self, arch)
        self._check_stats_arena(nstat)
        self._finalize_weights()
        self._build_tables()
        if training:
            self._build_backward()

    def _check_stats_arena(self, nstat: int):
        """every layer's [NSHARD][2C] slice must lie inside the arenas: kernels add to them through raw pointers"""
        if self._stats_used != nstat:
            raise RuntimeError(f"engine: the plan uses {self._stats_used} statistics elements, the arenas hold {nstat}")

    # ------------------------------------------------------------------ allocation helpers
    def _alloc(self, nbytes: int) -> torch.Tensor:
        """Every planned buffer lives as long as the engine: kernels hold raw pointers, so a tensor
        that merely lost its last Python reference must never go back to the caching allocator."""
        self.act_bytes += nbytes
        t = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        if self._poison:   # debug (STLPOSE_POISON=1): every planned buffer starts as NaNs, so that a kernel which reads a location
            t.fill_(0xFF)  # nobody wrote -- and lets it reach a result -- shows up deterministically (0xFFFF / 0xFFFFFFFF = NaN)
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

    def stem_input(self) -> Act:
        B, H, W = self.B, self.H, self.W
        Ho, Wo = H // 2, W // 2
        t = self._act_tensor(B, Ho, Wo, 32)
        pd = capi.Patch()
        pd.dtype, pd.B, pd.H, pd.W, pd.stride = self.fdtype, B, H, W, 2
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
        p.dtype = self.fdtype
        p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co = x.B, x.H, x.W, x.C, Ho, Wo, cout
        p.ks, p.stride, p.stuff = kks, kstride, 0
        p.TH, p.TW, p.shape = 0, 0, -1
        capi.call("stl_conv_plan", C.byref(p))  # block shape + pixel tile, searched once
        reads, writes = [x.ptr], [y.ptr]
        pend = x.pending
        if (pend is not None and kstride == 1 and kks in (1, 3) and pend[0][2] == self._stream and self.merge_minc <= x.C
                and capi.lib().stl_conv_bnadd_ok(C.byref(p)) == 1):
            # Residual block end z = ReLU(BN(y2) + skip) whose FIRST consumer is this convolution (the next unit's conv1: 3x3
            # in the branches, 1x1 in layer1 -- round 5): the sum is formed while the conv stages its tiles and written out once
            # (STL_SRC_BNADD + src_out) -- the stand-alone sum launch and one pass over the tensor go (HRnet.py:58-59, 88-100).
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
        self.fwd_ops.append(("stl_conv_forward", p, self._stream, reads, writes))
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
        p.dtype, p.B, p.H, p.W, p.C, p.nterms, p.relu = self.fdtype, B, H, W, Cc, len(terms), int(relu)
        for i, (a, s, tr) in enumerate(terms):
            assert a.C == Cc and a.H << s == H and a.W << s == W, "fuse: term shape mismatch"
            p.t[i].src = self._src(a, relu=tr)
            p.t[i].shift = s
            a.consumers += 1
        p.out = z.ptr
        op = ("stl_fuse_forward", p, self._stream, [a.ptr for a, _, _ in terms], [z.ptr])
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
        hd.dtype, hd.B, hd.H, hd.W, hd.Ci, hd.J = self.fdtype, x.B, x.H, x.W, x.C, joints
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
        # ---- gradient buckets: contiguous suffixes of the flat gradient buffer, closed as soon as every
        # parameter in them has its slabs / BatchNorm reductions complete (backward finishes the last
        # layers first).  Each bucket gets one ranged slab reduction + BN-gradient launch inside the
        # program, so the step has no serial tail, and an event a data-parallel all-reduce can wait on.
        self.buckets: List[dict] = []
        bucket_min = int(self.bucket_mb * (1 << 20) / 4)
        bk = dict(done=0, lo=st.nparam, hi=st.nparam, slab0=0, reads=[], strm=0, wuse=[])
        # The serial tail of backward (layer1 + stem: one branch, 113 MB tensors) finishes last.  Close a bucket
        # where it begins, whatever its size, so that the final slab reduction (the only work left after the last
        # weight gradient, in front of the optimiser) covers just the stem / layer1 slabs instead of every layer
        # since the last 16 MB boundary.
        tail_keys = ["transition1.0.0.weight", "layer1.1.conv1.weight"]
        force_at = {st.param_off[k] for k in tail_keys if k in st.param_off}

        def bucket_add(off: int, size: int):
            bk["done"] += size
            bk["lo"] = min(bk["lo"], off)

        self._wg_pending: Dict[Tuple, List] = {}   # grouped weight gradients waiting for their group to fill
        def bucket_close(force: bool = False):
            complete = bk["done"] == bk["hi"] - bk["lo"]          # suffix [lo, hi) fully covered
            force = force or (complete and bk["lo"] in force_at)
            if not complete or bk["done"] == 0 or (bk["done"] < bucket_min and not force):
                return
            assert complete
            for key in list(self._wg_pending):                     # the bucket's slab reduction reads every member's slabs
                self._flush_wgrad_group(ops, key)
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
                nblk = max(1, min(256, math.ceil(x.B * x.H * x.W / 256)))   # <= one 256-pixel chunk per block
                dx = self._new_grad(x)
                nel = joints * x.C + joints
                part_off = self._slab_elems
                self._slab_elems += (nblk * nel + 3) // 4 * 4   # keep every entry 16-byte aligned
                hb = capi.HeadBwd()
                hb.dtype, hb.B, hb.H, hb.W, hb.Ci, hb.J, hb.nblk = capi.dt2(self.dtype, self.fdtype), x.B, x.H, x.W, x.C, joints, nblk
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
                p.ydtype = self.ydtype
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
                        u.ydtype = self.ydtype
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
                g = self._gsrc(y)
                self._emit_wgrad(ops, bk, x, y, ci, g, kks, kstride, strm)
                bucket_add(ci.master_off, ci.Co * ci.Ci * ci.ks * ci.ks)
                bucket_add(y.bn.param_off, 2 * y.bn.C)   # gamma, beta of the BatchNorm behind this conv
                # ---- data gradient
                if not x.needs_grad:
                    continue
                d = capi.Conv()
                d.dtype, d.ydtype = self.dtype, self.ydtype
                d.B, d.Hi, d.Wi, d.Ci = y.B, y.H, y.W, y.C
                d.Ho, d.Wo, d.Co = x.H, x.W, x.C
                d.ks, d.stride, d.stuff = kks, 1, int(kstride == 2)
                d.TH, d.TW, d.shape = 0, 0, -1
                d.src = g   # (before the plan: data gradients get a block shape of their own)
                capi.call("stl_conv_plan", C.byref(d))
                d.w = self.wk.data_ptr() + ci.bwd_off * self.esz
                dreads = [y.dt.data_ptr()]
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
                    if (F is not None and F[3] and len(same_bn) == 1 and not x.grads
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
                ops.append(("stl_conv_forward", d, strm, dreads, [out.data_ptr(), ("wuse", ci.master_off)]))
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
        self.bwd_ops = self._balance_streams(ops)
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
            return 12.0 + (src + out) * esz / 2.0e6
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
                cands = list(range(active, self.nstreams)) + [own]
                if active == 1:   # the single-branch tail: TWO of its three idle queues (see above)
                    cands = list(range(1, min(3, self.nstreams))) + [own]
                strm = min(cands, key=lambda s_: (acc[s_], s_ != own))
            acc[strm] += cost
            out.append((name, desc, strm, reads, writes))
        self.sched_estimate_us = list(acc)
        return out

    def _emit_wgrad(self, ops, bk, x: Act, y: Act, ci: ConvInfo, g, kks: int, kstride: int, strm: int):
        """Weight-gradient launch of one convolution (split-K slabs): off the critical path (only the data-gradient chain
        is on it); _balance_streams places it on an idle queue where there is one."""
        wg = capi.Wgrad()
        wg.dtype, wg.ydtype = self.dtype, self.ydtype
        wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co = x.B, x.H, x.W, x.C, y.H, y.W, y.C
        wg.ks, wg.stride = kks, kstride
        ctile = capi.lib().stl_wgrad_chunk(C.byref(wg))   # 32, or 64: the 1x1 layers' wide kernel and the 16-wave 3x3 blocks (C >= 64)
        wide3 = ctile == 64 and kks == 3   # 64 x 64 channels per 1024-thread block: ONE block per CU, up to eight layers per launch
        wg.TH, wg.TW = choose_tile(x.B, y.H, y.W, kstride, kks, self.esz, bn_cols=32, maxhalo=(256 if wide3 else 192) if ctile == 64 else 576)   # (256 halo pixels: two staging vectors per thread of the 16-wave block)
        npt = math.ceil(x.B * (y.H + 1) / wg.TH) * math.ceil(y.W / wg.TW)
        chunks = math.ceil(y.C / ctile) * math.ceil(x.C / ctile)
        budget = self.wgrad_blocks_wide if wide3 else self.wgrad_blocks
        # The single-branch tail of backward (layer1, stem, transition1) is one serial data-gradient chain beside three idle
        # queues: a group there fills only when the chain has walked through ALL its members (every layer1 group completes
        # at the first block, i.e. at the very end of the step), so grouped weight gradients pile up behind the chain:
        # tail launches are not grouped (round 3, groups of 4 / 2 / 1: 15.02 / 14.81 / 14.82 ms per step).
        tail = self._active_of(ci.key) == 1
        gmax = self.wgrad_group_wide if wide3 else self.wgrad_group
        gsize = 1 if (tail or self.skip_wgrad) else max(1, min(gmax, capi.WGRAD_GROUP_MAX, max(1, budget // chunks)))
        # Grouped launches (stl_conv_wgrad_group): weight gradients of one shape -- the 3x3 convolutions of a branch --
        # wait until `gsize` of them are ready and go out as ONE launch that shares the block budget: the four hardware
        # queues carry one off-chain launch instead of gsize (in stages 3 / 4 every queue is busy with a data-gradient
        # chain and each stand-alone weight gradient costs its chain a full launch latency, whatever its size), every
        # block walks gsize times as many pixel tiles, and gsize times fewer split-K slabs are written and reduced.
        bg = budget // gsize
        top = max(1, min(npt, bg // chunks if chunks <= bg else 1))
        wg.nsplit = min(range(1, top + 1), key=lambda ns: (math.ceil(npt / ns) + 0.004 * ns * chunks / 8, ns))
        wg.h = self._src(x)
        wg.g = g
        nel = y.C * kks * kks * x.C
        part_off = self._slab_elems
        self._slab_elems += (wg.nsplit * nel + 3) // 4 * 4
        self.slabs.append(dict(part_off=part_off, grad_off=ci.master_off, nsplit=wg.nsplit, Co=ci.Co, Ci=ci.Ci,
                               ks=ci.ks, Cip=ci.Cik, patch=int(ci.patch), stride=0, struct=wg))
        bk["reads"].append(id(wg))
        bk["strm"] = strm
        if self.skip_wgrad:
            return
        if gsize == 1:
            ops.append(("stl_conv_wgrad", wg, strm, [y.dt.data_ptr(), x.ptr], [id(wg)]))
            return
        key = (x.C, y.C, kks, kstride, x.H, x.W, wg.TH, wg.TW, wg.nsplit, int(g.mode), gsize)
        pend = self._wg_pending.setdefault(key, [])
        pend.append((wg, [y.dt.data_ptr(), x.ptr], strm))
        if len(pend) >= gsize:
            self._flush_wgrad_group(ops, key)

    def _flush_wgrad_group(self, ops, key):
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
                if clock[st_][t] >= j:
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
            h = C.c_void_p()
            capi.call("stl_program_create", arr, len(ops), self.nstreams, C.byref(h))
            prog = self._progs[key] = (h, arr)
        return prog[0]

    def _run(self, ops, stream: int):
        """Replay a program natively.  With several streams the independent branches of each
        exchange module (and, in backward, the weight gradients) run concurrently; fork/join and
        cross-stream dependencies are HIP events inside stl_program_run."""
        h = self._program(ops)
        if self.graph_mode:
            if id(ops) not in self._graphs:
                capi.call("stl_program_graph_build", h)
                self._graphs.add(id(ops))
            capi.call("stl_program_graph_launch", h, stream)
            return
        if self._side is None:
            self._make_streams()
        self._stream_arr[0] = stream
        rc = self.lib.stl_program_run(h, self._stream_arr)
        if rc != 0:
            raise RuntimeError(f"stl_program_run: {self.lib.stl_last_error().decode()}")

    def _make_streams(self):
        """HIP streams of the program: index 0 is the caller's stream, 1 .. nstreams-1 the other branch streams.
        ONE set of side streams per device, shared by every engine (engines never run concurrently): each new HIP stream is
        another hardware queue, queues are spread round-robin over the four compute pipes, and two ACTIVE queues on one pipe
        are time-sliced -- a second engine with streams of its own ran its plan at half speed (W32 256x192 as the second
        plan of a process: 20.8 ms per step instead of 10.2).  (CU-masked streams and HIP stream priorities were measured
        in round 3 -- 27-92 ms resp. 14.64-14.76 vs 14.66 ms per step -- and removed.)"""
        n = self.nstreams
        self._stream_arr = (C.c_void_p * n)()
        self._side = []
        pool = _STREAM_POOL.setdefault(self.dev.index, {})
        with torch.cuda.device(self.dev):
            for i in range(1, n):
                if i not in pool:
                    s_ = torch.cuda.Stream(device=self.dev)
                    pool[i] = (s_, s_.cuda_stream)
                self._side.append(pool[i][0])
                self._stream_arr[i] = pool[i][1]

    def prep_weights(self, stream: int):
        st = self.store
        capi.call("stl_weight_prep", capi.dt2(self.dtype, self.fdtype), st.master.data_ptr(), self.wk.data_ptr(), self._wprep_tab.data_ptr(),
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
            capi.call("stl_weight_prep_range", capi.dt2(self.dtype, self.fdtype), st.master.data_ptr(), self.wk.data_ptr(),
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
                      self._bn_tab.data_ptr(), len(self.bns), MOMENTUM, self.overflow.data_ptr(), stream)

    NO_OVERFLOW = 2 ** 31 - 1

    def check_forward_range(self):
        """Raise if a forward pass since the last call stored a non-finite raw conv output (one 4-byte read: call it where the
        host reads the loss anyway).  Training mode only: the guard rides on the BatchNorm statistics (stl_bn_running_update)."""
        i = int(self.overflow.item())
        if i == self.NO_OVERFLOW:
            return
        self.overflow.fill_(self.NO_OVERFLOW)
        bn = self.bns[i]
        name = next((k for k, off in self.store.param_off.items() if off == bn.param_off), f"BatchNorm #{i}")
        what = {capi.F16: "f16 (|y| > 65504)", capi.BF16: "bf16", capi.F32: "fp32"}[self.fdtype]
        raise FloatingPointError(
            f"stlpose_amd: the raw output of the convolution in front of {name[:-len('.weight')] if name.endswith('.weight') else name} "
            f"left the range of its {what} storage (non-finite BatchNorm statistics; first such layer in forward order).  "
            "The optimiser skipped that step (and skips every step until this is reported): the weights are intact; the running "
            "statistics of the layers behind it took their momentum update from the poisoned pass.  Forward tensors of the default "
            "'mixed' mode are f16; for a checkpoint with badly scaled weights use compute_dtype='bf16' (same speed, bf16 range) "
            "or 'fp32'.")

    def backward(self, stream: int, fused_optim: bool = False, on_bucket=None):
        """expects self.dout filled; leaves dL/dparam in store.grads (overwrites).  fused_optim: the program of
        attach_optimizer (optimiser + next step's weight layouts inside backward).  on_bucket(i): called on the host right after
        gradient bucket i's last op has been ENQUEUED (the program is issued range by range, stl_program_run_range): the
        data-parallel path enqueues the bucket's all-reduce there, so that in every in-order hardware queue it sits directly
        behind the bucket instead of behind the rest of backward."""
        assert self.training
        self.rstats.zero_()
        if on_bucket is not None and not fused_optim and self.buckets:
            h = self._program(self.bwd_ops)
            if self._side is None:
                self._make_streams()
            self._stream_arr[0] = stream
            first = 0
            for op_idx, i in sorted((b["op"], i) for i, b in enumerate(self.buckets)):
                capi.call("stl_program_run_range", h, self._stream_arr, first, op_idx + 1)
                first = op_idx + 1
                on_bucket(i)
            capi.call("stl_program_run_range", h, self._stream_arr, first, len(self.bwd_ops))
            return
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
                w.dtype, w.n, w.blk_base, w.nblocks = capi.dt2(self.dtype, self.fdtype), i1 - i0, self._wprep_blk0[i0], self._wprep_blk0[i1] - self._wprep_blk0[i0]
                w.master, w.wk, w.tab = self.store.master.data_ptr(), self.wk.data_ptr(), self._wprep_tab.data_ptr() + i0 * C.sizeof(capi.WPrep)
                new_ops.append(("stl_wprep_range", w, strm, [("optim", i)], [("wprep", i)]))
                self._optim_descs.append(w)

        nxt = dict(at)
        for idx, op in enumerate(ops):
            new_ops.append(op)
            if idx in nxt:
                if prev is not None:
                    emit(prev, op[2])
                prev = nxt[idx]
        if prev is not None:
            emit(prev, ops[self.buckets[prev]["op"]][2])
        self.bwd_ops_opt = new_ops

    def bucket_wait(self, i: int, stream: int):
        """Make `stream` wait until gradient bucket i (self.buckets[i]: flat slice [lo, hi)) of the
        backward pass enqueued last is final."""
        capi.call("stl_program_wait_op", self._program(self.bwd_ops), self.buckets[i]["op"], stream)


_STREAM_POOL: Dict[int, Dict] = {}   # device index -> {stream index: (owner object, hipStream_t)}


def bn_weight_keys(reg: Registry) -> set:
    """Keys of the BatchNorm scale parameters of `reg` (every ``<bn>.weight`` whose ``<bn>.running_mean`` is a buffer).
    Computed from the registry itself on every call.  Rounds 2-4 cached this set in a module-level dict keyed by ``id(reg)``:
    a Registry is created per model, CPython hands a collected registry's address to the next one, and a W32 model built
    after a ``tiny`` model had been dropped got the TINY key set -- its statistics arenas came out 512 KB short, the
    producers' atomics of the layers beyond the end landed in whatever followed (and were never zeroed): the one-in-five
    "output 0.39 off, NaN gradients" failure of the 12th GPU test of round 4 (tests/test_host_cpu.py)."""
    return {k[: -len("running_mean")] + "weight" for k, _ in reg.buffers if k.endswith("running_mean")}


def bn_stat_elems(reg: Registry) -> int:
    """fp64 elements of one statistics arena: [NSHARD][2C] per BatchNorm layer."""
    keys = bn_weight_keys(reg)
    return sum(int(math.prod(s)) for k, s in reg.params if k in keys) * 2 * capi.NSHARD


def _to_device(ctab, device) -> torch.Tensor:
    raw = bytes(ctab)
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
    return t.to(device)
