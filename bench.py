#!/usr/bin/env python3
"""Headline benchmark: images/sec of one HRNet-W32 384x288 train step (BASELINE.json configs[1]):
fwd + masked-MSE + bwd + Adam (+ RCCL gradient all-reduce when N > 1), bf16 storage / fp32
accumulate, batch 32 per GPU, synthetic inputs already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
(N > 1: launched by torch.distributed.run, one rank per GPU.)  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_IMG = {("w32", 384, 288): 103.11e9, ("w32", 256, 192): 45.83e9, ("w48", 384, 288): 211.74e9,
                 ("w48", 256, 192): 94.11e9}  # fwd + dgrad + wgrad, convs only
MFMA_PEAK_BF16 = 2500.0  # TFLOP/s dense (MI355X_MICROARCH.md; the f16 MFMA forms take the same cycles)
DUMP_OPS = ""
DTYPE_LABEL = {"mixed": "f16/bf16", "bf16": "bf16", "fp32": "f32"}   # mixed: forward tensors + forward MFMA operands f16, gradients bf16, fp32 accumulate
HBM_PEAK = 8000.0        # GB/s


def synth_batch(B, H, W, rank, device, sigma=3.0, joints=17):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    img = torch.randn(B, 3, H, W, generator=g)
    hh, ww = H // 4, W // 4
    cx = torch.randint(0, ww, (B, joints, 1, 1), generator=g).float()
    cy = torch.randint(0, hh, (B, joints, 1, 1), generator=g).float()
    ys = torch.arange(hh).view(1, 1, hh, 1).float()
    xs = torch.arange(ww).view(1, 1, 1, ww).float()
    tgt = torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sigma ** 2))
    tw = (torch.rand(B, joints, 1, generator=g) < 0.8).float()
    return img.to(device), tgt.to(device), tw.to(device)


def time_kernel_families(ts):
    """Which kernel is the dominant one is MEASURED, not assumed: every launch of one forward + backward pass is replayed alone
    (a one-op native program on the launch stream, HIP events around it) and the durations are summed per kernel INSTANTIATION
    -- the name the library reports for the launch it just made (stl_last_kernel: template arguments in declaration order,
    what rocprofv3's kernel trace shows after demangling; tools/kernel_names.py maps one onto the other).  Per family:
    launches, summed time, algorithmic FLOPs and bytes (every tensor a launch reads or writes once, split-K
    slabs and halo re-reads are NOT algorithmic)."""
    from stlpose_amd import capi
    eng, lib = ts.eng, capi.lib()
    st = torch.cuda.current_stream().cuda_stream
    esz = eng.esz
    streams = (C.c_void_p * 1)(st)

    def cost(name, d):
        if name == "stl_conv_forward":
            px = d.B * (d.Hi * d.Wi if d.stuff else d.Ho * d.Wo)
            fl = 2.0 * px * d.Co * d.Ci * d.ks * d.ks
            by = esz * (d.B * d.Hi * d.Wi * d.Ci * (2 if d.src.mode in (capi.SRC_BNBWD, capi.SRC_BNADD) else 1)
                        + d.B * d.Ho * d.Wo * d.Co * (1 + bool(d.mask_y) + bool(d.addend) + bool(d.mask_z)) + d.B * d.Hi * d.Wi * d.Ci * bool(d.src_out))
            return fl, by
        if name in ("stl_conv_wgrad", "stl_conv_wgrad_group"):
            ms_ = d.members if name == "stl_conv_wgrad_group" else [d]
            fl = sum(2.0 * m.B * m.Ho * m.Wo * m.Co * m.Ci * m.ks * m.ks for m in ms_)
            by = sum(esz * (m.B * m.Hi * m.Wi * m.Ci + m.B * m.Ho * m.Wo * m.Co * (2 if m.g.mode == capi.SRC_BNBWD else 1)) + m.Co * m.Ci * m.ks * m.ks * 4 for m in ms_)
            return fl, by
        if name == "stl_fuse_forward":
            return 0.0, esz * d.B * d.H * d.W * d.C * (1 + sum(1.0 / (1 << (2 * d.t[i].shift)) for i in range(d.nterms)))
        if name == "stl_fuse_backward":
            return 0.0, esz * d.B * d.H * d.W * d.C * (d.ngrads + 1 + bool(d.relu) + d.nbn)
        if name == "stl_upsample_backward":
            return 0.0, esz * d.B * d.H * d.W * d.C * ((1 << (2 * d.shift)) + 2)
        return 0.0, 0.0
    fams, progs = {}, []
    for ops in (eng.fwd_ops, eng.bwd_ops):
        for name, d, *_ in ops:
            arr = (capi.Op * 1)()
            arr[0].kind, arr[0].stream, arr[0].desc, arr[0].nwait, arr[0].record = capi.OP_KIND[name], 0, C.addressof(d), 0, 0
            h = C.c_void_p()
            capi.call("stl_program_create", arr, 1, 1, C.byref(h))
            progs.append((h, arr, name, d))
    for h, *_ in progs[:16]:
        lib.stl_program_run(h, streams)
    torch.cuda.synchronize()
    evs = []
    for h, _arr, name, d in progs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.stl_program_run(h, streams)
        e1.record()
        assert rc == 0, lib.stl_last_error().decode()
        evs.append((e0, e1, lib.stl_last_kernel().decode(), name, d))
    torch.cuda.synchronize()
    for e0, e1, kern, name, d in evs:
        f = fams.setdefault(kern, dict(kernel=kern, launches=0, layers=0, ms=0.0, flops=0.0, bytes=0.0, replay=[]))
        fl, by = cost(name, d)
        f["launches"] += 1
        f["layers"] += d.n if name == "stl_conv_wgrad_group" else 1
        f["ms"] += e0.elapsed_time(e1)
        f["flops"] += fl
        f["bytes"] += by
    for (h, _arr, name, d), (_e0, _e1, kern, *_r) in zip(progs, evs):
        fams[kern]["replay"].append(h)
    # Per-family duration: the family's launches replayed BACK TO BACK between one pair of HIP events on the launch stream
    # (second of two passes).  A pair of events around every single launch adds the event / dispatch latency of an idle
    # queue (5 - 8 us) to each -- 4 % of a 50 us weight gradient, 60 % of a 15 us convolution: it ranked the short kernels
    # first and disagreed with the kernel trace (round 5).  Back to back, what is left beside the kernels is the ~1.5 us
    # dependent-launch gap per launch.
    for f in fams.values():
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for h in f["replay"]:
                lib.stl_program_run(h, streams)
            e1.record()
            torch.cuda.synchronize()
        f["ms_isolated_events"] = f["ms"]
        f["ms"] = e0.elapsed_time(e1)
    if DUMP_OPS:   # --dump-ops FILE: one line per launch (isolated duration, kernel, shape) for the analysis in DESIGN.md
        with open(DUMP_OPS, "w") as fo:
            for e0, e1, kern, name, d in evs:
                m = d.members[0] if name == "stl_conv_wgrad_group" else d
                shape = " ".join(f"{k}={getattr(m, k)}" for k in ("B", "Hi", "Wi", "Ci", "Ho", "Wo", "Co", "ks", "stride", "stuff", "H", "W", "C", "nterms", "ngrads", "shift", "nsplit") if hasattr(m, k))
                extra = f" n={d.n}" if name == "stl_conv_wgrad_group" else ""
                if name == "stl_conv_forward":
                    extra = f" src={d.src.mode} mask_y={int(bool(d.mask_y))} addend={int(bool(d.addend))} mask_z={int(bool(d.mask_z))} shape={d.shape} tile={d.TH}x{d.TW}"
                fl, by = cost(name, d)
                fo.write(f"{e0.elapsed_time(e1) * 1e3:8.1f} us  {name:24s} {kern:52s} {shape}{extra}  flops={fl:.3e} bytes={by:.3e}\n")
    return sorted(fams.values(), key=lambda f: -f["ms"]), progs


def family_roofline(f):
    """Roofline entry of one kernel family: bound = the roofline that takes longer at peak for the family's algorithmic work."""
    t_mfma, t_hbm = f["flops"] / (MFMA_PEAK_BF16 * 1e12), f["bytes"] / (HBM_PEAK * 1e9)
    sec = f["ms"] * 1e-3
    tf, gbs = f["flops"] / sec / 1e12, f["bytes"] / sec / 1e9
    out = dict(kernel=f["kernel"], launches_per_step=f["launches"], layers_per_step=f["layers"], sum_ms_per_step=round(f["ms"], 3),
               avg_launch_us=round(f["ms"] / f["launches"] * 1e3, 2), algorithmic_TFLOPs=round(tf, 2), mfma_frac=round(tf / MFMA_PEAK_BF16, 4),
               algorithmic_GBps=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK, 4), algorithmic_bytes_per_launch=round(f["bytes"] / f["launches"]),
               algorithmic_flops_per_launch=round(f["flops"] / f["launches"]),
               timing="the family's launches of one forward + backward pass replayed back to back on the launch stream between one pair of HIP events")
    if t_mfma >= t_hbm:
        out.update(bound="mfma", achieved=round(tf, 2), peak=MFMA_PEAK_BF16, unit="TFLOP/s", frac=round(tf / MFMA_PEAK_BF16, 4))
    else:
        out.update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK, unit="GB/s", frac=round(gbs / HBM_PEAK, 4))
    return out


def csrc_hash():
    from stlpose_amd import build
    return build.source_id()


def cpu_baseline(arch, H, W, batch=32, steps=6, adam=True):
    """The oracle (plain torch fp32 restatement of the reference graph) timed on the host cores at the benchmarked
    batch size (bs 32 forward + MSE + backward for a like-for-like ratio; the Adam step is in the timed
    region like in the GPU line).  Bounded sample: `steps` steps (~4-7 s each at W32 384x288 on 16 threads: one warm-up +
    five timed by default), median of all but the first."""
    from oracle import hrnet_ref, pose_ref
    torch.manual_seed(0)
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # the GPU box's CPU share for one GPU
    B = batch
    m = hrnet_ref.RefPoseNet(arch).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3) if adam else None
    img = torch.randn(B, 3, H, W)
    tgt = torch.rand(B, 17, H // 4, W // 4)
    tw = torch.ones(B, 17, 1)
    times = []
    for _ in range(max(steps, 2)):
        t0 = time.time()
        m.zero_grad(set_to_none=True)
        loss = pose_ref.person_mse_loss(m(img), tgt, tw)
        loss.backward()
        if opt is not None:
            opt.step()
        times.append(time.time() - t0)
    med = float(np.median(times[1:]))
    return dict(value=round(B / med, 3), unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle RefPoseNet({arch}) fp32 {H}x{W} bs{B} fwd+MSE+bwd{'+Adam' if adam else ''}, median of {len(times) - 1} steps after one warm-up step")


def cpu_baseline_cfg1(seconds=8.0):
    """BASELINE configs[0] exactly (W32, 256x192, bs 2, fwd + MSE + bwd; Adam timed separately) on all the host
    cores this process may use and on 8 threads (the build container's count)."""
    from oracle import hrnet_ref, pose_ref
    out = {}
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for n in sorted({min(ncpu, 16), 8}, reverse=True):   # 16 = the GPU box's CPU share for one GPU
        torch.set_num_threads(n)
        torch.manual_seed(0)
        m = hrnet_ref.RefPoseNet("w32").train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        img, tgt, tw = torch.randn(2, 3, 256, 192), torch.rand(2, 17, 64, 48), torch.ones(2, 17, 1)
        fb, ad = [], []
        t_end = time.time() + seconds / 2
        while time.time() < t_end or len(fb) < 3:
            t0 = time.time()
            opt.zero_grad()
            pose_ref.person_mse_loss(m(img), tgt, tw).backward()
            t1 = time.time()
            opt.step()
            fb.append(t1 - t0), ad.append(time.time() - t1)
            if len(fb) >= 12:
                break
        out[f"threads_{n}"] = dict(images_per_sec_fwd_bwd=round(2 / float(np.median(fb[1:])), 2),
                                   adam_step_ms=round(1e3 * float(np.median(ad[1:])), 1), steps=len(fb) - 1)
    out["host_cpus"] = ncpu
    return out


def extra_fp32_path(arch, batch, H, W, dev, steps=6, warmup=2):
    """The parity path (fp32 storage, fp32-input MFMA at 157 TFLOP/s peak) at the same configuration."""
    from stlpose_amd import PoseHighResolutionNet
    from stlpose_amd.train_step import TrainStep
    torch.manual_seed(0)
    model = PoseHighResolutionNet(arch, "fp32").to(dev)
    ts = TrainStep(model, batch, H, W, optimizer="adam", lr=1e-3, device=dev)
    ts.load_batch(*synth_batch(batch, H, W, 0, dev, sigma=3.0 if H >= 384 else 2.0))
    for _ in range(warmup):
        ts.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fl = FLOPS_PER_IMG.get((arch, H, W))
    tf = batch / dt * fl / 1e12 if fl else None
    del ts, model
    torch.cuda.empty_cache()
    return dict(metric=f"images/sec/GPU HRNet-{arch.upper()} {H}x{W} train step, fp32 parity path", value=round(batch / dt, 2),
                unit="images/sec", ms_per_step=round(dt * 1e3, 3), dtype="f32",
                roofline=dict(bound="mfma", achieved=round(tf, 2) if tf else None, peak=157.3, unit="TFLOP/s",
                              frac=round(tf / 157.3, 4) if tf else None, traffic=None,
                              note="whole step against the fp32-input MFMA peak (v_mfma_f32_16x16x4_f32)"))


def extra_train_leg(arch, batch, H, W, dev, steps=20, warmup=6, cpu_batch=8, dtype="mixed"):
    """The same train step (bf16, fwd + MSE + bwd + Adam) at another configuration of BASELINE.json's list: W32 256x192
    (north_star: "throughput on synthetic 256x192 and 384x288 batches") and W48 384x288 (configs[2]'s shape on one GPU)."""
    from stlpose_amd import PoseHighResolutionNet
    from stlpose_amd.train_step import TrainStep
    torch.manual_seed(0)
    model = PoseHighResolutionNet(arch, dtype).to(dev)
    ts = TrainStep(model, batch, H, W, optimizer="adam", lr=1e-3, device=dev)
    ts.load_batch(*synth_batch(batch, H, W, 0, dev, sigma=3.0 if H >= 384 else 2.0))
    for _ in range(warmup):
        ts.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss = float(ts.loss.item())
    fl = FLOPS_PER_IMG[(arch, H, W)]
    tf = batch / dt * fl / 1e12
    del ts, model
    torch.cuda.empty_cache()
    out = dict(metric=f"images/sec/GPU HRNet-{arch.upper()} {H}x{W} bs={batch} train step (fwd+MSE+bwd+Adam)", value=round(batch / dt, 2),
               unit="images/sec", ms_per_step=round(dt * 1e3, 3), dtype=DTYPE_LABEL[dtype], steps=steps, warmup=warmup, loss=loss,
               roofline=dict(bound="mfma", achieved=round(tf, 2), peak=MFMA_PEAK_BF16, unit="TFLOP/s", frac=round(tf / MFMA_PEAK_BF16, 4),
                             traffic=None, note="whole step: algorithmic conv FLOPs (SURVEY 8(d)) / step time"))
    if cpu_batch:
        out["cpu_baseline"] = cpu_baseline(arch, H, W, batch=cpu_batch, steps=2)
    return out


def extra_eval_path(dev, batches=10, batch=32, H=384, W=288, persons_per_image=4):
    """SURVEY 8(d)'s substitute for configs[4] (end-to-end OKS needs real data): the evaluation path of
    03_evaluate.py:132-188 on a synthetic checkpoint -- per batch flip-test forward x 2, flip_back + average on the
    device, loss, PCK, quarter-pixel decode + inverse affine; then box re-scoring and OKS-NMS over all persons.  Timed
    through stlpose_amd.evaluate.Evaluator itself, host post-processing included."""
    from stlpose_amd import PoseHighResolutionNet
    from stlpose_amd.evaluate import Evaluator
    torch.manual_seed(0)
    model = PoseHighResolutionNet("w32", "mixed").to(dev).eval()   # eval: forward only, i.e. all-f16 tensors
    img, tgt, tw = synth_batch(batch, H, W, 0, dev, sigma=3.0)
    rng = np.random.Generator(np.random.PCG64(5))

    def loader(n):
        for bi in range(n):
            ids = (bi * batch + np.arange(batch)) // persons_per_image
            meta = dict(center=rng.uniform(100, 400, (batch, 2)), scale=rng.uniform(0.8, 2.0, (batch, 2)), score=rng.uniform(0.3, 1.0, batch),
                        image_id=ids)
            yield img, tgt, tw, meta
    ev = Evaluator(model, device=dev, flip=True)
    ev.evaluate_model(loader(2))   # warm-up: plan + lazy kernel attributes
    torch.cuda.synchronize()
    prof = None
    if os.environ.get("STL_BENCH_EVAL_PROFILE"):   # where does the host spend this leg? (cProfile -> stderr)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    res = ev.evaluate_model(loader(batches))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(14)
    n = batches * batch
    tf = n / dt * 2 * 34.403e9 / 1e12     # two forward passes per image (F_fwd 34.403 GFLOP at 384x288)
    out = dict(metric="images/sec evaluation path HRNet-W32 384x288 bs=32: flip-test forward x2 + flip_merge + loss + PCK + final_preds + rescoring / OKS-NMS",
               value=round(n / dt, 2), unit="images/sec", ms_per_batch=round(dt / batches * 1e3, 3), dtype="f16", batches=batches,
               results=len(res.get("results", [])) if isinstance(res, dict) else None,
               roofline=dict(bound="mfma", achieved=round(tf, 2), peak=MFMA_PEAK_BF16, unit="TFLOP/s", frac=round(tf / MFMA_PEAK_BF16, 4), traffic=None,
                             note="2 x forward conv FLOPs per image / wall time, host post-processing included"))
    del ev, model
    torch.cuda.empty_cache()
    # CPU baseline: the oracle's flip-test forward + decode + NMS on a bounded sample (4 images)
    from oracle import hrnet_ref, pose_ref
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = hrnet_ref.RefPoseNet("w32").eval()
    x = torch.randn(4, 3, H, W)
    t0 = time.time()
    with torch.no_grad():
        o = pose_ref.forward_pass(ref, x, flip=True).numpy()
    pose_ref.final_preds(o, np.full((4, 2), 200.0), np.full((4, 2), 1.2))
    out["cpu_baseline"] = dict(value=round(4 / (time.time() - t0), 3), unit="images/sec", cores=torch.get_num_threads(), kind="port",
                               sample="oracle flip-test forward + final_preds, 4 images of 3x384x288, one evaluation")
    return out


def _he_weights(convs, seed):
    """Synthetic He-normal weights under torchvision's key names (no checkpoint, no network): (features idx, cin, cout)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for idx, cin, cout in convs:
        out[f"features.{idx}.weight"] = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        out[f"features.{idx}.bias"] = torch.randn(cout, generator=g) * 0.05
    return out


def extra_vgg_cfg4(dev, reps=5):
    """BASELINE configs[3] as far as the reference defines it (V1): VGG16 perceptual loss forward, batch 16,
    3 x 512 x 512, fp32, resize=False = 2 x 16 images through features[:23] (145.9 GFLOP per image and pass)."""
    from stlpose_amd import VGGPerceptualLoss
    from stlpose_amd.vgg import VGG16_LAYOUT
    w = _he_weights([(idx, ci, co) for _, idx, ci, co, _ in VGG16_LAYOUT], 16)
    g = torch.Generator().manual_seed(16)
    a = torch.rand(16, 3, 512, 512, generator=g)
    b = (a + 0.2 * torch.randn(16, 3, 512, 512, generator=g)).clamp_(0, 1)
    res = {}
    for dt_name, peak in (("fp32", 157.3), ("bf16", MFMA_PEAK_BF16)):
        m = VGGPerceptualLoss(resize=False, state_dict=w, compute_dtype=dt_name).to(dev)
        ad, bd = a.to(dev), b.to(dev)
        for _ in range(2):
            m(ad, bd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            loss = m(ad, bd)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tf = 2 * 16 * 145.9e9 / dt / 1e12
        res[dt_name] = dict(ms_per_loss=round(dt * 1e3, 3), pairs_per_sec=round(16 / dt, 2), loss=float(loss.item()),
                            roofline=dict(bound="mfma", achieved=round(tf, 2), peak=peak, unit="TFLOP/s", frac=round(tf / peak, 4), traffic=None))
        del m
    # CPU baseline: the oracle on 2 pairs (bounded sample of the same workload); the only use of oracle/ in this leg
    from oracle import vgg_ref
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        vgg_ref.vgg_perceptual_loss(a[:1], b[:1], w, resize=False)   # warm-up (thread pool, allocator)
        t0 = time.time()
        vgg_ref.vgg_perceptual_loss(a[:2], b[:2], w, resize=False)
    cpu = 2 / (time.time() - t0)
    return dict(metric="image pairs/sec VGG16 perceptual loss forward 512x512 bs=16 (V1 of cfg4)", unit="pairs/sec", **res,
                cpu_baseline=dict(value=round(cpu, 3), unit="pairs/sec", cores=torch.get_num_threads(), kind="port",
                                  sample="oracle.vgg_ref fp32, 2 pairs of 3x512x512, resize=False, one evaluation after one warm-up pair"))


def extra_vgg19_style(dev, reps=3):
    """BASELINE configs[3] by name: VGG19 content + Gram style loss forward, 512 x 512, batch 16 (stylised, content and
    style batches = 48 images through conv1_1 .. conv5_1 + 160 Gram launches).  V2 has no reference item: parity
    unpinned (oracle = published-method restatement)."""
    from stlpose_amd.vgg19_style import VGG19_LAYOUT, VGG19StyleLoss, vgg19_flops_per_image
    w = _he_weights([(idx, ci, co) for idx, ci, co, _ in VGG19_LAYOUT], 19)
    g = torch.Generator().manual_seed(19)
    x, c, s_ = (torch.rand(16, 3, 512, 512, generator=g) for _ in range(3))
    fl = 48 * vgg19_flops_per_image(512, 512)
    res = {}
    for dt_name, peak in (("fp32", 157.3), ("bf16", MFMA_PEAK_BF16)):
        m = VGG19StyleLoss(state_dict=w, compute_dtype=dt_name).to(dev)
        xd, cd, sd = x.to(dev), c.to(dev), s_.to(dev)
        m(xd, cd, sd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            tot, cl, sl = m(xd, cd, sd)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tf = fl / dt / 1e12
        res[dt_name] = dict(ms_per_loss=round(dt * 1e3, 3), triplets_per_sec=round(16 / dt, 2), content=float(cl.item()), style=float(sl.item()),
                            roofline=dict(bound="mfma", achieved=round(tf, 2), peak=peak, unit="TFLOP/s", frac=round(tf / peak, 4), traffic=None))
        del m
        torch.cuda.empty_cache()
    from oracle import vgg_ref   # CPU baseline only
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        vgg_ref.vgg19_style_content_loss(x[:1, :, :128, :128], c[:1, :, :128, :128], s_[:1, :, :128, :128], w)   # warm-up on a crop
        t0 = time.time()
        vgg_ref.vgg19_style_content_loss(x[:1], c[:1], s_[:1], w)
    return dict(metric="image triplets/sec VGG19 content + Gram style loss forward 512x512 bs=16 (cfg4; V2, parity unpinned)", unit="triplets/sec", **res,
                cpu_baseline=dict(value=round(1 / (time.time() - t0), 3), unit="triplets/sec", cores=torch.get_num_threads(), kind="port",
                                  sample="oracle.vgg_ref.vgg19_style_content_loss fp32, 1 triplet of 3x512x512, one evaluation after a warm-up crop"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="w32")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=288)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--dtype", default="mixed", help="mixed (forward tensors f16, gradients bf16: the default 16-bit mode), bf16 or fp32")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32-path and VGG cfg4 legs")
    ap.add_argument("--dump-ops", default="", help="write the isolated duration of every launch of one step to this file")
    a = ap.parse_args()
    global DUMP_OPS
    DUMP_OPS = a.dump_ops

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as CHILD processes (one per GPU, RCCL)
        # before this process has touched the GPU, relay their output (rank 0 prints the JSON line) and
        # exit with their code.  Never exec over a process that initialised HIP.
        import socket
        import subprocess
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    # Before the first HIP call: the plan uses four compute streams = four hardware queues, and four is all the chip runs at
    # once (one queue per compute pipe) -- a FIFTH active queue is time-sliced against the others.  With the runtime's
    # default GPU_MAX_HW_QUEUES=4 the streams of a RCCL communicator are multiplexed onto the four compute queues (one-rank
    # rehearsal of the bucketed all-reduce path, round 3: 15.39 ms per step with the collectives vs 15.44 without); with 8
    # they become queues of their own and the same step takes 23.7 ms (5 / 6 queues: 22.4 / 22.1).  Keep four.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
    # Rank 0 prints ONE JSON line on stdout and nothing else: libraries that write banners to file descriptor 1 (RCCL prints
    # its version block there when the communicator is created) are sent to stderr; the line goes out through the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)  # rehearsal on a one-GPU box: several ranks share the card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world == 1 and os.environ.get("STLPOSE_DP_FORCE", "0") == "1":   # one-rank rehearsal of the bucketed all-reduce path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(os.environ.get("STL_DIST_BACKEND", "nccl"), rank=0, world_size=1)
        pg = dist.group.WORLD
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("STL_DIST_BACKEND", "nccl")  # "nccl" == RCCL on ROCm; gloo only for rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    from stlpose_amd import PoseHighResolutionNet
    from stlpose_amd.train_step import TrainStep
    torch.manual_seed(0)  # identical random-init weights on every rank
    model = PoseHighResolutionNet(a.arch, a.dtype).to(dev)
    ts = TrainStep(model, a.batch, a.height, a.width, optimizer="adam", lr=1e-3, process_group=pg,
                   use_graph=not a.no_graph, device=dev)
    img, tgt, tw = synth_batch(a.batch, a.height, a.width, rank, dev, sigma=3.0 if a.height >= 384 else 2.0)
    ts.load_batch(img, tgt, tw)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    import contextlib
    ctx = contextlib.nullcontext()
    with ctx:
        for _ in range(a.warmup):
            ts.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ts.step()
        barrier()
        dt = time.perf_counter() - t0
    loss = float(ts.loss.item())
    comm = None
    if world > 1 or ts._force_dp:
        # Self-describing multi-GPU line: what the communicator reports, each gradient bucket's all-reduce alone
        # (HIP events around the collective, nothing else in flight), and the step with the collectives switched off
        # -- exposed communication = ms_per_step - ms_per_step_no_comm.  Every rank runs the same sequence.
        import torch.distributed as dist
        per_bucket = []
        for i, b in enumerate(ts.dp.buckets):
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ts.dp.reduce_bucket(i, force=True)
            e1.record()
            torch.cuda.synchronize()
            per_bucket.append(dict(mbytes=round(b.numel() * (2 if ts.dp.bf16 else 4) / 2 ** 20, 2), allreduce_ms=round(e0.elapsed_time(e1) / 3, 4)))
        dp_saved, force_saved = ts.dp, ts._force_dp
        ts.dp, ts._force_dp = None, False
        nc = max(5, min(a.steps, 20))
        with ctx:
            for _ in range(3):
                ts.step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(nc):
                ts.step()
            barrier()
            dt_nc = (time.perf_counter() - t1) / nc
        ts.dp, ts._force_dp = dp_saved, force_saved
        tt = torch.tensor([dt_nc], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        comm = dict(world=dist.get_world_size() if dist.is_initialized() else 1, backend=dist.get_backend() if dist.is_initialized() else None,
                    bf16_buckets=bool(ts.dp.bf16), buckets=per_bucket, gradient_mbytes=round(ts.store.nparam * 4 / 2 ** 20, 1),
                    ms_per_step_no_comm=round(float(tt.item()) * 1e3, 3),
                    exposed_comm_ms=round(dt / a.steps * 1e3 - float(tt.item()) * 1e3, 3))
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        if comm is not None:
            comm["exposed_comm_ms"] = round(dt / a.steps * 1e3 - comm["ms_per_step_no_comm"], 3)
    ms = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt

    if rank == 0:
        flops_img = FLOPS_PER_IMG.get((a.arch, a.height, a.width))
        fams, progs = time_kernel_families(ts)
        roof = others = None
        if fams:
            roof = family_roofline(fams[0])
            # HBM bytes per launch of THAT kernel from the committed rocprofv3 --pmc passes of this same command
            # (tools/pmc_traffic.py -> profiles/r*_pmc_dominant.json); used only if it was taken on these kernel sources
            # (build id) for this kernel and launch count, else null
            traffic = traffic_src = None
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_dominant.json"))):
                try:
                    rec = json.load(open(f))
                    if rec.get("kernel") == roof["kernel"] and rec.get("launches") == roof["launches_per_step"] and rec.get("build_id") == csrc_hash() \
                            and (a.arch, a.height, a.width, a.batch, a.dtype) == ("w32", 384, 288, 32, rec.get("dtype", "mixed")):
                        traffic, traffic_src = round(rec["traffic_bytes_per_launch"]), "profiles/" + os.path.basename(f)
                except Exception:
                    pass
            roof.update(traffic=traffic, traffic_unit="HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)",
                        traffic_source=(traffic_src + " (rocprofv3 --pmc passes of this command on this build id, committed; not measured in this run)") if traffic_src else None,
                        chosen_by="largest summed duration over one forward + backward pass (per kernel instantiation, its launches replayed back to back between HIP events in this run)")
            others = [family_roofline(f) for f in fams[1:6]]
            # the dominant family's launches once more, back to back, as the LAST dispatches of the process: what tools/pmc_traffic.py reads
            lib_ = __import__("stlpose_amd.capi", fromlist=["lib"]).lib()
            streams_ = (C.c_void_p * 1)(torch.cuda.current_stream().cuda_stream)
            for h in fams[0]["replay"]:
                lib_.stl_program_run(h, streams_)
            torch.cuda.synchronize()
        out = dict(metric=f"images/sec/GPU HRNet-{a.arch.upper()} {a.height}x{a.width} train step; PCKh@0.5 parity", value=round(value, 2),
                   unit="images/sec", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype=DTYPE_LABEL.get(a.dtype, a.dtype), data="synthetic",
                   config=dict(workload=f"HRNet-{a.arch.upper()} {a.height}x{a.width} bs={a.batch}/GPU train step "
                                        f"(fwd+MSE+bwd+Adam{'+RCCL allreduce' if world > 1 else ''}), random-init weights",
                               global_batch=a.batch * world, parallelism=f"dp{world}"),
                   loss=loss,
                   step_tflops=round(value * flops_img / 1e12, 2) if flops_img else None,
                   step_mfma_frac=round(value / world * flops_img / 1e12 / MFMA_PEAK_BF16, 4) if flops_img else None,
                   roofline=roof, roofline_next=others)
        if comm is not None:
            out["comm"] = comm
        if world == 1 and not a.no_extras:
            del ts, model
            torch.cuda.empty_cache()
            extras = {}
            for name, fn in (("w32_256x192", lambda: extra_train_leg("w32", a.batch, 256, 192, dev)),
                             ("w48_384x288", lambda: extra_train_leg("w48", a.batch, 384, 288, dev, steps=12, warmup=4)),
                             ("pure_bf16", lambda: extra_train_leg(a.arch, a.batch, a.height, a.width, dev, steps=12, warmup=4, cpu_batch=0, dtype="bf16")),
                             ("eval_path", lambda: extra_eval_path(dev)),
                             ("fp32_path", lambda: extra_fp32_path(a.arch, a.batch, a.height, a.width, dev)),
                             ("vgg_cfg4", lambda: extra_vgg_cfg4(dev)), ("vgg19_style_cfg4", lambda: extra_vgg19_style(dev))):
                try:
                    extras[name] = fn()
                except Exception as e:   # an extra leg must never take the headline line down
                    extras[name] = dict(error=f"{type(e).__name__}: {e}")
            out["extras"] = extras
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.arch, a.height, a.width)
            if not a.no_extras:
                out["cpu_baseline"]["cfg1"] = cpu_baseline_cfg1()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
