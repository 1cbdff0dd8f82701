"""GPU parity, round-3 additions (VERDICT r2 "Next round" item 1).

* bf16 path at the benchmarked plan (W32 384x288 B = 32) on TRAINED weights: the net is first fitted to one fixed
  synthetic batch until the oracle's PCK exceeds 0.8, so that the heat maps have real peaks; then output error, argmax
  agreement by the SURVEY section-7 margin rule, argmax displacement and PCK are checked against the fp32 oracle at those
  weights.  The output bar is CALIBRATED: the oracle itself is re-run with bf16 rounding at the HIP path's storage points
  (oracle.hrnet_ref.bf16_storage) and the HIP path must stay within a small multiple of that distance.
* fp32 path, every parameter gradient element-wise against an **fp64** oracle run, with bars relative to what torch's own
  fp32 run achieves against the same fp64 run (they do not move with tile shapes or split-K plans).
* VGG19 content + Gram style loss at BASELINE configs[3]'s size (16 x 3 x 512 x 512) against oracle.vgg_ref.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import hrnet_ref, pose_ref, vgg_ref  # noqa: E402  (checker only)
from stlpose_amd import PersonMSELoss, PoseHighResolutionNet, get_max_preds_hrnet  # noqa: E402
from stlpose_amd.pose_parsing import accuracy  # noqa: E402
from stlpose_amd.train_step import TrainStep  # noqa: E402
from tests.golden.make_golden import synth_batch  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _diag(name, lines):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name), "w") as f:
        f.write("\n".join(lines) + "\n")


def _load_synth(model):
    sd = {k: torch.from_numpy(hrnet_ref.synth_tensor(k, tuple(v.shape))) for k, v in model.state_dict().items()}
    model.load_state_dict(sd, strict=True)
    return model


# ------------------------------------------------------------------------------------------------ bf16 on peaky heat maps
def textured_batch(b, h, w, joints=17, seed=4321, sigma=3.0):
    """A batch a network CAN localise on (pure-noise images -- SURVEY 8(d)'s throughput input -- cannot be fitted in a
    few hundred steps: 1500 Adam steps reached PCK 0.27): weak noise plus, at every joint, an oriented Gabor patch whose
    orientation / frequency / colour identify the joint.  Targets as in JointsDataset.py:248-281 (unnormalised gaussian)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    hh, ww = h // 4, w // 4
    ys, xs = np.mgrid[0:hh, 0:ww].astype(np.float32)
    Y, X = np.mgrid[0:h, 0:w].astype(np.float32)
    tgt = np.zeros((b, joints, hh, ww), np.float32)
    img = 0.3 * rng.standard_normal((b, 3, h, w)).astype(np.float32)
    ang = np.pi * np.arange(joints) / joints
    freq = 0.35 + 0.25 * (np.arange(joints) % 3)
    col = rng.standard_normal((joints, 3)).astype(np.float32)
    for n in range(b):
        for j in range(joints):
            cx, cy = rng.integers(3, ww - 3), rng.integers(3, hh - 3)
            tgt[n, j] = np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sigma ** 2))
            env = np.exp(-((X - 4 * cx - 1.5) ** 2 + (Y - 4 * cy - 1.5) ** 2) / (2 * 7.0 ** 2))
            ph = freq[j] * (np.cos(ang[j]) * X + np.sin(ang[j]) * Y)
            img[n, 0] += 2 * env * np.cos(ph)
            img[n, 1] += 2 * env * np.sin(ph)
            img[n] += env[None] * col[j][:, None, None]
    return img, tgt, np.ones((b, joints, 1), np.float32)


@pytest.fixture(scope="module")
def trained_b32():
    """Fit W32 to ONE textured batch (32 x 3 x 384 x 288, sigma 3) with the fused TrainStep (bf16 path: only the
    resulting WEIGHTS matter here) until the device PCK on that batch exceeds 0.9 (at most 4000 Adam steps), and hand
    back the weights + the fp32 oracle's forward at them."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    img, tgt, tw = textured_batch(32, 384, 288)
    m = _load_synth(PoseHighResolutionNet("w32", "bf16"))
    ts = TrainStep(m, 32, 384, 288, optimizer="adam", lr=1e-3)
    ti, tt, tww = torch.from_numpy(img).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda()
    ts.load_batch(ti, tt, tww)
    hist, steps = [], 0
    while steps < 4000:
        for _ in range(100):
            loss = ts.step()
        steps += 100
        m.train()
        with torch.no_grad():
            pck = accuracy(m(ti), tt)[1]
        hist.append((steps, float(loss.item()), float(pck)))
        if pck > 0.9:
            break
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    del ts, m
    torch.cuda.empty_cache()
    ref = hrnet_ref.RefPoseNet("w32")
    ref.load_state_dict(sd, strict=True)
    ref.train()
    with torch.no_grad():
        # train-mode forward (batch statistics), like the benchmarked step; momentum 0 keeps the buffers as loaded
        for mod in ref.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.momentum = 0.0
        out = ref(torch.from_numpy(img)).numpy()
        with hrnet_ref.bf16_storage(ref):
            out_emul = ref(torch.from_numpy(img)).numpy()
        with hrnet_ref.f16_storage(ref):
            out_emul_f16 = ref(torch.from_numpy(img)).numpy()
    del ref
    return dict(img=img, tgt=tgt, tw=tw, sd=sd, out=out, out_emul=out_emul, out_emul_f16=out_emul_f16, hist=hist)


def test_w32_b32_bf16_on_trained_weights_vs_oracle(trained_b32):
    r = trained_b32
    m = PoseHighResolutionNet("w32", "bf16")
    m.load_state_dict(r["sd"], strict=True)
    m = m.cuda().train()
    with torch.no_grad():
        out = m(torch.from_numpy(r["img"]).cuda())
    torch.cuda.synchronize()
    o = out.detach().cpu().numpy()
    ref = r["out"]
    absmax = float(np.abs(ref).max())
    err = float(np.abs(o - ref).max() / absmax)
    err_emul = float(np.abs(r["out_emul"] - ref).max() / absmax)
    rms = float(np.sqrt(np.mean((o - ref) ** 2)) / absmax)
    rms_emul = float(np.sqrt(np.mean((r["out_emul"] - ref) ** 2)) / absmax)
    p, _ = get_max_preds_hrnet(o)
    pr, _ = pose_ref.get_max_preds(ref)
    pe, _ = pose_ref.get_max_preds(r["out_emul"])
    flat = ref.reshape(32, 17, -1)
    top2 = np.partition(flat, -2, axis=-1)[..., -2:]
    margin = (top2[..., 1] - top2[..., 0]) / absmax
    same = (p == pr).all(-1)
    disp = np.abs(p - pr).max(-1)                        # Chebyshev displacement of the argmax, pixels
    disp_emul = np.abs(pe - pr).max(-1)
    # per-map error: a flip is legitimate only where the oracle's top-2 margin is below twice the error ON THAT MAP
    emap = np.abs(o - ref).reshape(32, 17, -1).max(-1) / absmax
    acc_b = pose_ref.pck_accuracy(o, r["tgt"])
    acc_r = pose_ref.pck_accuracy(ref, r["tgt"])
    acc_dev = accuracy(out.detach(), torch.from_numpy(r["tgt"]).cuda())
    acc_e = pose_ref.pck_accuracy(r["out_emul"], r["tgt"])
    ae, ae_emul = np.abs(o - ref).reshape(-1) / absmax, np.abs(r["out_emul"] - ref).reshape(-1) / absmax
    q999, q999_emul = float(np.quantile(ae, 0.999)), float(np.quantile(ae_emul, 0.999))
    n_same, n_same_emul = int(same.sum()), int((disp_emul == 0).sum())
    n_far, n_far_emul = int((disp > 1).sum()), int((disp_emul > 1).sum())
    lines = [f"training history (step, loss, device PCK): {r['hist']}",
             f"out rel err: max {err:.3e} / 99.9 % {q999:.3e} / rms {rms:.3e};  bf16-storage emulation of the oracle: max {err_emul:.3e} / 99.9 % {q999_emul:.3e} / rms {rms_emul:.3e}",
             f"argmax: {n_same}/{same.size} equal, {int((~same).sum())} flipped; emulation {n_same_emul} equal",
             f"flips beyond twice the per-map error: {int((~same & (margin > 2 * emap)).sum())}",
             f"argmax displacement: max {disp.max():.0f} px, maps moved > 1 px: {n_far}; emulation max {disp_emul.max():.0f} px, {n_far_emul} maps",
             f"margin/|out|max quantiles 10/50/90 %: {np.quantile(margin, 0.1):.3e} {np.quantile(margin, 0.5):.3e} {np.quantile(margin, 0.9):.3e}",
             f"PCK bf16 {acc_b[1]:.6f} oracle {acc_r[1]:.6f} emulation {acc_e[1]:.6f} device accuracy() {acc_dev[1]:.6f}"]
    _diag("diag_w32_b32_bf16_trained.txt", lines)
    assert acc_r[1] > 0.8, f"the fitted net does not localise (oracle PCK {acc_r[1]}): {r['hist']}"
    # Every bar is a multiple of what bf16 STORAGE costs the oracle itself on this batch (the emulation above), not a
    # free constant: a trained BatchNorm net has channels of tiny variance whose 1/std amplifies the 2^-9 rounding of
    # the stored activations -- the tail of the error distribution is a property of the storage format.
    assert rms < 2.0 * rms_emul + 1e-4, f"rms error {rms:.3e} vs emulation {rms_emul:.3e}"
    assert q999 < 2.0 * q999_emul + 1e-4, f"99.9 % error {q999:.3e} vs emulation {q999_emul:.3e}"
    assert err < 3.0 * err_emul + 1e-3, f"max error {err:.3e} vs emulation {err_emul:.3e}"
    # argmax: a flip is only legitimate where the oracle's top-2 margin is below twice the error on that very map
    assert not (~same & (margin > 2 * emap)).any(), "argmax flipped on a map where the error cannot explain it"
    assert n_same >= n_same_emul - 0.1 * same.size, f"{n_same} maps keep their argmax, the emulation keeps {n_same_emul}"
    assert n_far <= n_far_emul + 0.05 * same.size, f"{n_far} maps moved their argmax by more than a pixel, the emulation {n_far_emul}"
    # PCK: within what the storage format itself moves it.  The emulation and the HIP path are two independent realisations
    # of the same rounding noise (runs of this test: HIP - oracle +0.0092 / -0.0074, emulation - oracle +0.011 / -0.0018 on
    # 544 joints), so the bar is three times the emulation's own shift or 8 joints, whichever is larger
    assert abs(acc_b[1] - acc_r[1]) <= max(3.0 * abs(acc_e[1] - acc_r[1]), 8.0 / same.size) + 1e-12, "PCK of the bf16 path vs the fp32 oracle"
    np.testing.assert_allclose(acc_dev[0], acc_b[0], rtol=0, atol=1e-12)


def test_w32_b32_mixed_on_trained_weights_vs_oracle(trained_b32):
    """The MIXED mode (forward tensors f16, gradients bf16 -- the benchmarked dtype since round 4) on the same fitted weights,
    with ABSOLUTE bars.  Round 4 located the bf16 path's distance from the fp32 oracle (tools/bf16_error_probe.py: rounding
    switched on at one class of storage points at a time): the materialised sums of the residual stream carry 5.8e-3 of its
    6.0e-3 rms error, conv inputs 2.5e-3, raw conv outputs 1.3e-3, weights 0.6e-3 -- mantissa, not range (|mean| / std of the
    raw conv outputs: median 0.22, max 2.7, so centring them buys nothing).  f16 has three more mantissa bits at the same bytes
    and MFMA rate, and BatchNorm bounds the forward tensors.  Measured: max 4.6e-2 / 99.9 % 3.9e-3 / rms 7.7e-4 of |out|max
    (bf16: 3.7e-1 / 3.2e-2 / 6.0e-3), 518 of 544 argmax kept (408), 8 maps moved by more than a pixel (39)."""
    r = trained_b32
    m = PoseHighResolutionNet("w32", "mixed")
    m.load_state_dict(r["sd"], strict=True)
    m = m.cuda().train()
    with torch.no_grad():
        out = m(torch.from_numpy(r["img"]).cuda())
    torch.cuda.synchronize()
    ref = r["out"]
    absmax = float(np.abs(ref).max())
    pr, _ = pose_ref.get_max_preds(ref)
    acc_r = pose_ref.pck_accuracy(ref, r["tgt"])

    def figures(o, decode):
        ae = np.abs(o - ref).reshape(-1) / absmax
        p, _ = decode(o)
        disp = np.abs(p - pr).max(-1)
        return dict(err=float(ae.max()), q999=float(np.quantile(ae, 0.999)), rms=float(np.sqrt(np.mean(ae ** 2))),
                    same=int((disp == 0).sum()), far=int((disp > 1).sum()), n=disp.size, pck=pose_ref.pck_accuracy(o, r["tgt"])[1])
    hip, emu = figures(out.detach().cpu().numpy(), get_max_preds_hrnet), figures(r["out_emul_f16"], pose_ref.get_max_preds)
    _diag("diag_w32_b32_mixed_trained.txt",
          [f"{tag}: out rel err max {f['err']:.3e} / 99.9 % {f['q999']:.3e} / rms {f['rms']:.3e}; argmax kept {f['same']}/{f['n']}, moved > 1 px: {f['far']}; "
           f"PCK {f['pck']:.6f} (oracle {acc_r[1]:.6f})" for tag, f in (("HIP mixed", hip), ("oracle with f16 storage", emu))])
    # Bars = what was measured in round 4 (HIP 4.6e-2 / 3.9e-3 / 7.7e-4, 518 kept, 8 moved; the f16-storage oracle 5.3e-2 / 3.9e-3 /
    # 7.7e-4, 514, 7) with ~1.3x headroom for the fitting run's own variation -- and the EMULATION must pass them too: they
    # describe what f16 storage at these points costs, not this implementation.
    for tag, f in (("HIP mixed", hip), ("oracle with f16 storage", emu)):
        assert f["rms"] < 1e-3 and f["q999"] < 5e-3 and f["err"] < 7e-2, (tag, f)
        assert f["same"] >= 505 and f["far"] <= 11, (tag, f)
        assert abs(f["pck"] - acc_r[1]) <= 8.0 / f["n"] + 1e-12, (tag, f)
    # and the HIP path is no further from the fp32 oracle than the emulation by more than a quarter
    assert hip["rms"] <= 1.25 * emu["rms"] and hip["q999"] <= 1.25 * emu["q999"], (hip, emu)


# ------------------------------------------------------------------------------------------------ fp32 gradients vs fp64
def test_w32_b32_fp32_gradients_vs_fp64_oracle():
    """Every parameter gradient of the fp32 path at B = 32 against the oracle in DOUBLE precision.  torch's own fp32
    run of the same graph is compared to the same fp64 run; the HIP bars are multiples of what torch-fp32 achieves
    (median / 90 % / worst tensor), so they follow the problem's conditioning and not this implementation's tiling."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    img, tgt, tw = synth_batch(32, 384, 288, seed=4321, sigma=3.0)
    res = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        ref = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("w32")).to(dt).train()
        out = ref(torch.from_numpy(img).to(dt))
        loss = pose_ref.person_mse_loss(out, torch.from_numpy(tgt).to(dt), torch.from_numpy(tw).to(dt))
        loss.backward()
        res[name] = {k: p.grad.double().clone() for k, p in ref.named_parameters()}
        del ref, out, loss
    m = _load_synth(PoseHighResolutionNet("w32", "fp32")).cuda().train()
    out = m(torch.from_numpy(img).cuda())
    PersonMSELoss()(out, torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda()).backward()
    torch.cuda.synchronize()
    e_hip, e_t32, names = [], [], []
    for k, prm in m.named_parameters():
        g64 = res["f64"][k].reshape(-1)
        den = max(float(g64.abs().max()), 1e-300)
        e_hip.append(float((prm.grad.detach().cpu().double().reshape(-1) - g64).abs().max() / den))
        e_t32.append(float((res["f32"][k].reshape(-1) - g64).abs().max() / den))
        names.append(k)
    e_hip, e_t32 = np.array(e_hip), np.array(e_t32)
    q = lambda a, f: float(np.quantile(a, f))   # noqa: E731
    order = np.argsort(-e_hip)[:12]
    _diag("diag_w32_b32_fp32_vs_fp64.txt",
          [f"tensors {len(names)}", f"HIP fp32 vs fp64: median {q(e_hip, .5):.3e} p90 {q(e_hip, .9):.3e} max {e_hip.max():.3e}",
           f"torch fp32 vs fp64: median {q(e_t32, .5):.3e} p90 {q(e_t32, .9):.3e} max {e_t32.max():.3e}"]
          + [f"{names[i]}: hip {e_hip[i]:.3e} torch32 {e_t32[i]:.3e}" for i in order])
    # measured (profiles/r03_b32_fp32_vs_fp64.txt): median 2.1x, 90 % 1.8x, worst tensor 3.3x torch-fp32's distance
    # from fp64 -- 221 k-term fp32 sums per weight accumulated sequentially by the matrix cores and split-K slabs, against
    # mkldnn's blocked sums, plus more ReLU-sign events.  The bars are 3x / 3x / 5x of torch-fp32's OWN
    # figures on this batch, and absolute caps that a wrong tap, slab or statistic would break by orders of magnitude.
    assert q(e_hip, .5) <= 3.0 * q(e_t32, .5) + 1e-5 and q(e_hip, .5) < 5e-3, "median gradient error vs fp64"
    assert q(e_hip, .9) <= 3.0 * q(e_t32, .9) + 1e-5 and q(e_hip, .9) < 1e-2, "90th-percentile gradient error vs fp64"
    assert e_hip.max() <= 5.0 * e_t32.max(), f"worst gradient vs fp64: {names[order[0]]} {e_hip.max():.3e} (torch fp32 worst {e_t32.max():.3e})"


# ------------------------------------------------------------------------------------------------ V2 at cfg4's size
def test_vgg19_style_cfg4_shape_16x512x512_vs_oracle():
    """BASELINE configs[3] by name: VGG19 content + Gram style loss, batch 16, 512 x 512 (48 images, 160 Gram launches,
    the full-size split-K plan).  Both losses are batch means of per-image terms, so the oracle runs 4 images at a time.
    V2 has no reference item: the oracle is the published-method restatement (PARITY UNPINNED)."""
    from stlpose_amd.vgg19_style import VGG19StyleLoss
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    w = vgg_ref.synth_vgg19_weights()
    g = torch.Generator().manual_seed(1916)
    x, c, s = (torch.rand(16, 3, 512, 512, generator=g) for _ in range(3))
    rc = rs = 0.0
    with torch.no_grad():
        for i in range(0, 16, 4):
            _, cl, sl = vgg_ref.vgg19_style_content_loss(x[i:i + 4], c[i:i + 4], s[i:i + 4], w, 1.0, 1e3)
            rc, rs = rc + cl.item() / 4, rs + sl.item() / 4
    for dt, tol in (("fp32", 2e-3), ("bf16", 6e-2)):
        m = VGG19StyleLoss(1.0, 1e3, state_dict=w, compute_dtype=dt).cuda()
        tot, cl, sl = m(x.cuda(), c.cuda(), s.cuda())
        assert abs(cl.item() - rc) <= tol * abs(rc), (dt, cl.item(), rc)
        assert abs(sl.item() - rs) <= tol * abs(rs), (dt, sl.item(), rs)
        assert abs(tot.item() - (rc + 1e3 * rs)) <= tol * abs(rc + 1e3 * rs)
        del m
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ W48 at the cfg3 resolution
def test_w48_384x288_fp32_train_step_vs_oracle():
    """BASELINE configs[2]'s network and resolution (W48, 384 x 288; batch 4 so that the oracle runs in seconds) on the fp32
    path against the oracle: the 48 / 96 / 192 / 384 widths at full-size maps go through the ragged channel chunks of the
    grouped weight gradient and, for C = 192, through the block-end source (STL_SRC_BNADD) -- output 1e-3, argmax bit-exact,
    every gradient norm 5e-3, direction 0.9995."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    img, tgt, tw = synth_batch(4, 384, 288, seed=4848, sigma=3.0)
    ref = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("w48")).train()
    ro = ref(torch.from_numpy(img))
    rl = pose_ref.person_mse_loss(ro, torch.from_numpy(tgt), torch.from_numpy(tw))
    rl.backward()
    m = _load_synth(PoseHighResolutionNet("w48", "fp32")).cuda().train()
    out = m(torch.from_numpy(img).cuda())
    loss = PersonMSELoss()(out, torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda())
    loss.backward()
    torch.cuda.synchronize()
    o, r = out.detach().cpu().numpy(), ro.detach().numpy()
    assert np.abs(o - r).max() / np.abs(r).max() < 1e-3
    p, _ = get_max_preds_hrnet(o)
    pr, _ = pose_ref.get_max_preds(r)
    assert np.array_equal(p, pr), "heatmap argmax differs from the oracle"
    assert abs(loss.item() - rl.item()) < 1e-3 * abs(rl.item())
    worst_n, worst_c = 0.0, 1.0
    for (k, prm), (kr, pr_) in zip(m.named_parameters(), ref.named_parameters()):
        assert k == kr
        g, gr = prm.grad.detach().cpu().double().reshape(-1), pr_.grad.double().reshape(-1)
        worst_n = max(worst_n, abs(float(g.norm()) - float(gr.norm())) / (float(gr.norm()) + 1e-30))
        worst_c = min(worst_c, float(torch.dot(g, gr) / (g.norm() * gr.norm() + 1e-30)))
    assert worst_n < 5e-3 and worst_c > 0.9995, (worst_n, worst_c)
    # the plan of this run did use the block-end source for the 192-channel branch
    from stlpose_amd import capi
    eng = m.engine(4, 384, 288, True)
    merged = {o_[1].Ci for o_ in eng.fwd_ops if o_[0] == "stl_conv_forward" and o_[1].src.mode == capi.SRC_BNADD}
    assert 192 in merged, merged


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_in_program_optimizer_equals_host_issued(opt, monkeypatch):
    """STLPOSE_FUSED_OPTIM=1 (optimiser slices + next step's weight layouts as ops of the backward program, behind the
    ("wuse", layer) tokens of the data gradients that still read a bucket's weights) must leave the same weights, moments and
    running statistics as the host-issued optimiser after three steps on three batches -- a slice applied while a data
    gradient still reads the old weights, or a stale layout in the next forward, shows up as a different loss / weight.
    Small buckets so that several optimiser slices sit in the middle of the backward program.  Reference: optimizer.step()
    after loss.backward(), 02_train.py:113-114."""
    monkeypatch.setenv("STLPOSE_BUCKET_MB", "8")
    kw = dict(optimizer="adam", lr=1e-3) if opt == "adam" else dict(optimizer="sgd", lr=1e-2, momentum=0.9, weight_decay=5e-4, nesterov=True)
    out = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("STLPOSE_FUSED_OPTIM", fused)
        m = _load_synth(PoseHighResolutionNet("w32", "bf16")).cuda()
        ts = TrainStep(m, 2, 128, 96, **kw)
        assert ts._fused_optim == (fused == "1")
        if fused == "1":
            names = [o[0] for o in ts.eng.bwd_ops_opt]
            assert names.count("stl_optim_slice") == len(ts.eng.buckets) >= 4 and names.count("stl_wprep_range") >= 4
        losses = []
        for step in range(3):
            img, tgt, tw = synth_batch(2, 128, 96, seed=500 + step)
            ts.load_batch(torch.from_numpy(img).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda())
            losses.append(float(ts.step().item()))
        torch.cuda.synchronize()
        out[fused] = (losses, ts.store.master.clone(), ts.m.clone(), ts.store.bufs.clone(), int(ts.step_count.item()))
    (l0, w0, m0, b0, s0), (l1, w1, m1, b1, s1) = out["0"], out["1"]
    assert s0 == s1 == 3
    # split-K slabs are reduced in a fixed order and the optimiser arithmetic is element-wise: only the fp64 atomics of the
    # BatchNorm statistics can differ in their last bits between two runs
    np.testing.assert_allclose(l1, l0, rtol=1e-5)
    for a, b, what in ((w1, w0, "weights"), (m1, m0, "first moment"), (b1, b0, "running statistics")):
        d = float((a - b).abs().max())
        ref = float(b.abs().max())
        assert d <= 2e-4 * ref, f"{what}: max |diff| {d:.3e} of {ref:.3e}"
