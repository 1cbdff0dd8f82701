"""CPU: host logic -- the C-ABI library loads and exports every symbol the header declares, the
drop-in module has the reference's state_dict ABI, and the static planner builds a consistent
plan (no kernel is launched here)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from stlpose_amd import build
    build.build(verbose=False)
    from stlpose_amd import capi
    return capi.lib()


def test_capi_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "stlpose_hip.h")).read()
    declared = set(re.findall(r"\b(stl_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    from stlpose_amd import capi
    for name in declared:
        assert hasattr(built_lib, name), f"{name} declared in include/stlpose_hip.h but not exported"
    assert set(capi.SIGNATURES) | set(capi.STRING_FUNCS) == declared
    assert built_lib.stl_version() == 1


def test_library_was_built_from_this_tree(built_lib):
    """stl_build_id() = hash of csrc/* and include/* baked in at compile time; it must equal the hash of the sources in
    the tree, i.e. the .so that travels to the GPU box is the one these sources produce."""
    from stlpose_amd import build
    assert built_lib.stl_build_id().decode() == build.source_id()


def test_struct_sizes_match_header(built_lib):
    """ctypes mirrors must have the C layout (compile a tiny C program with gcc and compare sizeof)."""
    import subprocess
    import tempfile
    from stlpose_amd import capi
    src = '#include <stdio.h>\n#include "stlpose_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(stl_src),sizeof(stl_conv),sizeof(stl_wgrad),sizeof(stl_term),sizeof(stl_fuse),sizeof(stl_fuse_bwd),' \
          'sizeof(stl_upbwd),sizeof(stl_wprep),sizeof(stl_slab),sizeof(stl_bnrec),sizeof(stl_patch),sizeof(stl_head),' \
          'sizeof(stl_head_bwd),sizeof(stl_op),sizeof(stl_reduce_range),sizeof(stl_bn_range));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = list(map(int, subprocess.check_output([os.path.join(d, "s")]).split()))
    import ctypes
    mine = [ctypes.sizeof(c) for c in (capi.Src, capi.Conv, capi.Wgrad, capi.Term, capi.Fuse, capi.FuseBwd, capi.UpBwd,
                                       capi.WPrep, capi.Slab, capi.BNRec, capi.Patch, capi.Head, capi.HeadBwd, capi.Op,
                                       capi.ReduceRange, capi.BNRange)]
    assert mine == sizes


def test_dropin_state_dict_abi(golden_dir):
    from stlpose_amd import PoseHighResolutionNet
    m = PoseHighResolutionNet(is_train=False)  # reference call site lib/model_setup.py:38
    lines = [f"{k} {'x'.join(map(str, v.shape))}" for k, v in m.state_dict().items()]
    assert lines == open(os.path.join(golden_dir, "g8_w32_keys.txt")).read().splitlines()
    assert sum(p.numel() for p in m.parameters()) == 28536113
    # strict load of an oracle (== reference-layout) checkpoint, with and without DataParallel's prefix
    from oracle import hrnet_ref
    ref = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("w32"))
    m.load_state_dict(ref.state_dict(), strict=True)
    assert torch.equal(m.stage3[1].fuse_layers[2][0][1][0].weight, ref.stage3[1].fuse_layers[2][0][1][0].weight)
    wrapped = torch.nn.DataParallel(m)
    wrapped.load_state_dict({"module." + k: v for k, v in ref.state_dict().items()}, strict=True)
    assert m.training and not m.eval().training
    assert sum(p.numel() for p in PoseHighResolutionNet("w48").parameters()) == 63595745


def test_planner_dry_run(monkeypatch):
    """Plan construction needs no GPU: check op counts against SURVEY 8(a) (293 convs incl. head,
    28 upsample terms) and that every BN activation has a single consumer."""
    from collections import Counter
    from stlpose_amd import PoseHighResolutionNet, capi
    from stlpose_amd.engine import Engine
    m = PoseHighResolutionNet("w32", "bf16")
    m._pack(torch.device("cpu"))
    e = Engine(m.arch, m._store, 2, 256, 192, capi.BF16, True)
    f, b = Counter(o[0] for o in e.fwd_ops), Counter(o[0] for o in e.bwd_ops)
    assert f["stl_conv_forward"] + f["stl_head_forward"] == 293
    # 292 weight gradients, launched stand-alone or as members of grouped launches (round 3)
    nwg = b["stl_conv_wgrad"] + sum(o[1].n for o in e.bwd_ops if o[0] == "stl_conv_wgrad_group")
    assert b["stl_conv_forward"] == 291 and b["stl_upsample_backward"] == 28
    assert len(e.slabs) == 292 + 2   # + head weight and bias
    e0, c0 = e, b
    assert nwg == 292 and 0 < c0["stl_conv_wgrad_group"] < 80 and c0["stl_conv_wgrad"] < 80   # most layers ride in groups
    for o in e0.bwd_ops:
        if o[0] == "stl_conv_wgrad_group":
            ms = o[1].members
            # up to 4 per launch on the 32-channel blocks, up to 8 on the 64-channel 3x3 blocks (C >= 64; round 5)
            assert 2 <= o[1].n <= (8 if ms[0].Ci >= 64 and ms[0].Co >= 64 and ms[0].ks == 3 else 4)
            assert len({(m.Ci, m.Co, m.ks, m.stride, m.Hi, m.Wi, m.TH, m.TW, m.nsplit, m.g.mode) for m in ms}) == 1
    monkeypatch.setenv("STLPOSE_WGRAD_GROUP", "1")
    monkeypatch.setenv("STLPOSE_WGRAD_GROUP_WIDE", "1")
    assert Counter(o[0] for o in Engine(m.arch, m._store, 2, 256, 192, capi.BF16, True).bwd_ops)["stl_conv_wgrad"] == 292
    monkeypatch.delenv("STLPOSE_WGRAD_GROUP")
    monkeypatch.delenv("STLPOSE_WGRAD_GROUP_WIDE")
    assert len(e.bns) == 292 and e.out.shape == (2, 17, 64, 48)
    ev = Engine(m.arch, m._store, 1, 64, 64, capi.F32, False)
    assert not ev.bwd_ops
    # residual block ends formed by the consuming conv1 (STL_SRC_BNADD): at the benchmarked size the 3 inner block ends of every
    # branch with a block-end kernel variant are eligible (24 + 24 + 21 + 9 = 78: C = 32, 64, 128 and, since the deep small
    # maps run on the two-per-CU kernel -- round 5 --, C = 256), plus the two layer1 sums whose first consumer is a 1x1
    # bottleneck convolution (256 -> 64; round 5); the default merges C >= 128 only
    def merged(e):
        return Counter(o[1].Ci for o in e.fwd_ops if o[0] == "stl_conv_forward" and o[1].src.mode == capi.SRC_BNADD)
    eb = Engine(m.arch, m._store, 32, 384, 288, capi.BF16, True)
    assert merged(eb) == {128: 21, 256: 11} and Counter(o[0] for o in eb.fwd_ops)["stl_fuse_forward"] == 136 - 32
    assert sum(1 for o in eb.fwd_ops if o[0] == "stl_conv_forward" and o[1].src.mode == capi.SRC_BNADD and o[1].ks == 1) == 2
    monkeypatch.setenv("STLPOSE_MERGE_MINC", "0")
    ea = Engine(m.arch, m._store, 32, 384, 288, capi.BF16, True)
    assert merged(ea) == {32: 24, 64: 24, 128: 21, 256: 11} and Counter(o[0] for o in ea.fwd_ops)["stl_fuse_forward"] == 136 - 80
    for o in ea.fwd_ops:   # the merged conv writes the sum it consumed: src_out is the tensor the fuse launch would have produced
        if o[0] == "stl_conv_forward" and o[1].src.mode == capi.SRC_BNADD:
            assert o[1].src_out and o[1].src.y and o[1].src_out in o[4]
    assert len(ea.bwd_ops) == len(eb.bwd_ops)   # backward is untouched: the sums are still materialised


def test_choose_tile_bounds():
    from stlpose_amd.engine import choose_tile
    for (B, H, W, s, ks) in [(32, 96, 72, 1, 3), (32, 12, 9, 1, 3), (2, 8, 6, 1, 1), (32, 48, 36, 2, 3), (1, 3, 2, 1, 3)]:
        th, tw = choose_tile(B, H, W, s, ks, 2)
        assert 1 <= th and 1 <= tw <= W and th * tw <= 128


def test_torch_library_ops_are_registered_and_gpu_only():
    """north_star: 'hand-written HIP kernels through PyTorch-ROCm custom ops'.  Every op of the ``stlpose``
    namespace exists, has only a CUDA (HIP) kernel -- a CPU tensor is refused by the dispatcher, there is no CPU
    fallback -- and is traceable through its fake implementation."""
    import torch
    from stlpose_amd import ops
    for name in ops.OPS:
        assert hasattr(torch.ops.stlpose, name), name
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.stlpose.heatmap_argmax(torch.zeros(1, 17, 4, 4))
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.stlpose.person_mse(torch.zeros(1, 17, 4, 4), torch.zeros(1, 17, 4, 4), torch.ones(1, 17, 1), 1.0)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        hm = torch.empty(2, 17, 8, 6, device="cuda")
        idx, mx, preds = torch.ops.stlpose.heatmap_argmax(hm)
        assert idx.shape == (2, 17) and mx.shape == (2, 17, 1) and preds.shape == (2, 17, 2)
        loss, dout = torch.ops.stlpose.person_mse(hm, hm, torch.empty(2, 17, 1, device="cuda"), 1.0)
        assert loss.shape == () and dout.shape == hm.shape


def test_scalar_magic_division_is_exact_in_the_checked_range():
    """csrc `sdiv(n, m)` = (n * ceil(2^32 / d)) >> 32 replaces n / d on the scalar unit (conv_core.hip, wgrad.hip) for tile
    decoding.  The hosts check d < 2^11 (conv: tiles per row, virtual row pitches) resp. d < 2^12 (weight gradient) and
    n < 2^21 resp. 2^20: exactness needs n * (m * d - 2^32) < 2^32, which those ranges imply (m * d - 2^32 < d)."""
    rng = np.random.default_rng(0)
    for dmax, nmax in ((1 << 11, 1 << 21), (1 << 12, 1 << 20)):
        ds = np.unique(np.concatenate([np.arange(1, 64), rng.integers(1, dmax, 400), [dmax - 1]])).astype(np.uint64)
        for d in ds:
            m = (np.uint64(1 << 32) + d - np.uint64(1)) // d
            n = np.unique(np.concatenate([np.arange(0, 4096), rng.integers(0, nmax, 4000), [nmax - 1],
                                          (np.arange(1, 300) * int(d)) % nmax, (np.arange(1, 300) * int(d) - 1) % nmax])).astype(np.uint64)
            q = (n * m) >> np.uint64(32)
            assert np.array_equal(q, n // d), f"d={int(d)}"


def test_in_program_optimizer_ops_are_ordered_behind_the_last_readers(monkeypatch):
    """engine.attach_optimizer (STLPOSE_FUSED_OPTIM=1): the optimiser slice of a gradient bucket may run only when (a) the bucket's
    reductions are done and (b) every data gradient that still reads the bucket's weights / BatchNorm parameters has run -- the
    ("wuse", layer) tokens.  Checked on the planned op list (no GPU): one optimiser op per bucket, placed behind the bucket's own
    reductions, waiting for a token of every convolution in the bucket that has a data gradient; the weight re-layout follows its
    optimiser op; the scheduler turns the tokens into waits on the right producers."""
    monkeypatch.setenv("STLPOSE_BUCKET_MB", "8")
    from stlpose_amd import PoseHighResolutionNet, capi
    from stlpose_amd.engine import Engine
    m = PoseHighResolutionNet("w32", "bf16")
    m._pack(torch.device("cpu"))
    e = Engine(m.arch, m._store, 2, 128, 96, capi.BF16, True)
    e.attach_optimizer(0, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 5 << 20, 6 << 20)   # dummy addresses: nothing is launched
    ops = e.bwd_ops_opt
    names = [o[0] for o in ops]
    nb = len(e.buckets)
    assert nb >= 8 and names.count("stl_optim_slice") == nb
    assert [o for o in ops if o[0] not in ("stl_optim_slice", "stl_wprep_range")] == list(e.bwd_ops)   # the backward ops themselves are untouched
    pos_bn = {i: next(k for k, o in enumerate(ops) if o[1] is b["br"]) for i, b in enumerate(e.buckets)}
    producers = {}
    for k, o in enumerate(ops):
        for w in o[4]:
            if isinstance(w, tuple) and w[0] == "wuse":
                producers[w] = k
    dgrad_layers = {w[1] for w in producers}
    covered = 0
    for k, o in enumerate(ops):
        if o[0] != "stl_optim_slice":
            continue
        i = next(r[1] for r in o[3] if isinstance(r, tuple) and r[0] == "bucketbn")
        b = e.buckets[i]
        assert k > pos_bn[i], "optimiser slice in front of its bucket's reductions"
        assert o[1].n == b["hi"] - b["lo"] and o[1].p == (1 << 20) + 4 * b["lo"]
        toks = {r for r in o[3] if isinstance(r, tuple) and r[0] == "wuse"}
        want = {("wuse", c.master_off) for c in e.convs if b["lo"] <= c.master_off < b["hi"] and c.master_off in dgrad_layers}
        assert toks == want and all(producers[t] < k for t in toks)
        covered += len(toks)
        nxt = ops[k + 1]
        if nxt[0] == "stl_wprep_range":
            assert ("optim", i) in nxt[3] and nxt[2] == o[2]
    assert covered == len(dgrad_layers) == sum(1 for o in e.bwd_ops if o[0] == "stl_conv_forward")
    waits, need = e._schedule(ops)
    # happens-before closure of the emitted waits (streams are in-order; a wait hands over everything its producer knew): the
    # scheduler prunes waits that are implied transitively, possibly through a third stream, so the check must follow them too
    ns = max(o[2] for o in ops) + 1
    known, last_on = [], {}
    for k, o in enumerate(ops):
        kn = list(known[last_on[o[2]]]) if o[2] in last_on else [-1] * ns
        for w in waits[k]:
            assert w in need and w < k
            kn = [max(a, b) for a, b in zip(kn, known[w])]
        if o[0] == "stl_optim_slice":
            for t in (r for r in o[3] if isinstance(r, tuple) and r[0] == "wuse"):
                j = producers[t]
                assert kn[ops[j][2]] >= j, (k, t)   # the last reader of the layer's weights is complete when the optimiser slice starts
        kn[o[2]] = k
        known.append(kn)
        last_on[o[2]] = k


def test_statistics_arena_is_sized_from_the_registry_itself_not_from_an_id_keyed_cache(monkeypatch):
    """Round-4's one-in-five wrong result (output 0.39 off, NaN gradients in the 12th GPU test of a process): the BatchNorm key
    set was cached per ``id(registry)``; CPython re-uses the address of a collected registry, so a W32 model built after a dropped
    ``tiny`` model sized its statistics arenas with the tiny key set (11168 of 27168 channels: 512 KB short) and the kernels'
    atomics ran past the end.  The size now comes from the registry on every call, and the planner refuses a plan whose
    statistics slices do not fill the arena exactly."""
    import gc
    import random
    from stlpose_amd import PoseHighResolutionNet, capi, engine
    from stlpose_amd.arch import ARCHS, registry
    want = {a: 2 * capi.NSHARD * sum(s[0] for k, s in registry(ARCHS[a]).buffers if k.endswith("running_mean")) for a in ARCHS}
    assert want["w32"] == 27168 * 2 * capi.NSHARD and want["tiny"] < want["w32"] < want["w48"]
    random.seed(0)
    seen, reused = {}, 0
    for _ in range(600):   # address re-use across architectures happens within a few dozen iterations
        a = random.choice(list(ARCHS))
        r = registry(ARCHS[a])
        reused += seen.get(id(r), a) != a
        seen[id(r)] = a
        assert engine.bn_stat_elems(r) == want[a]
        del r
        gc.collect(0)
    assert reused > 0, "the loop never re-used a registry address for another architecture: the test lost its point"
    # the planner's own check: tiny plan, then W32 plans in the same process
    for arch in ("tiny", "w32", "tiny", "w32"):
        m = PoseHighResolutionNet(arch, "fp32")
        m._pack(torch.device("cpu"))
        e = engine.Engine(m.arch, m._store, 2, 64, 64, capi.F32, True)
        assert e.stats.numel() == e.rstats.numel() == e._stats_used == want[arch]
        del m, e
    m = PoseHighResolutionNet("w32", "fp32")
    m._pack(torch.device("cpu"))
    monkeypatch.setattr(engine, "bn_stat_elems", lambda reg: want["tiny"])   # what the stale cache entry amounted to
    with pytest.raises(RuntimeError, match="statistics elements"):
        engine.Engine(m.arch, m._store, 2, 64, 64, capi.F32, True)
