"""GPU: the whole HIP network (through the nn.Module drop-in) against the golden fixtures that the
REFERENCE produced (tests/golden/make_golden.py), and against the oracle on seeded inputs.

Tolerances (BASELINE north_star): fp32 path within 1e-3 relative of the reference output,
heatmap argmax indices bit-exact; bf16 path: 5e-2 relative, argmax agreement reported."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import hrnet_ref, pose_ref  # noqa: E402  (checker only)
from stlpose_amd import PersonMSELoss, PoseHighResolutionNet, forward_pass, get_max_preds_hrnet  # noqa: E402
from tests.golden.make_golden import synth_batch  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _load_synth(model):
    sd = {k: torch.from_numpy(hrnet_ref.synth_tensor(k, tuple(v.shape))) for k, v in model.state_dict().items()}
    model.load_state_dict(sd, strict=True)
    return model


def _diag(name, lines):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name), "w") as f:
        f.write("\n".join(lines) + "\n")


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_tiny_fp32_vs_reference_golden(golden_dir, mode, monkeypatch):
    g = np.load(os.path.join(golden_dir, f"g1_tiny_{mode}.npz"))
    m = _load_synth(PoseHighResolutionNet("tiny", "fp32")).cuda()
    m.train(mode == "train")
    img = torch.from_numpy(g["img"]).cuda()
    if mode == "eval":
        with torch.no_grad():
            out = m(img)
        scale = np.abs(g["output"]).max()
        err = np.abs(out.cpu().numpy() - g["output"]).max() / scale
        assert err < 1e-3, f"eval output rel err {err}"
        return
    out = m(img)
    loss = PersonMSELoss()(out, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["target_weight"]).cuda())
    loss.backward()
    torch.cuda.synchronize()
    scale = np.abs(g["output"]).max()
    err = np.abs(out.detach().cpu().numpy() - g["output"]).max() / scale
    lines = [f"output rel err {err:.3e}", f"loss {loss.item()} ref {float(g['loss'])}"]
    grads = {k: p.grad for k, p in m.named_parameters()}
    names = list(g["param_keys"])
    assert names == [k for k, _ in m.named_parameters()]
    norms = np.array([float(grads[k].double().norm()) for k in names])
    rel = np.abs(norms - g["gradnorm_all"]) / (g["gradnorm_all"] + 1e-12)
    worst = np.argsort(-rel)[:25]
    lines += [f"gradnorm {names[i]}: got {norms[i]:.6e} ref {g['gradnorm_all'][i]:.6e} rel {rel[i]:.2e}" for i in worst]
    full = []
    for k in g.files:
        if k.startswith("grad/"):
            ref = g[k]
            got = grads[k[5:]].cpu().numpy()
            full.append((np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12), k))
    full.sort(reverse=True)
    lines += [f"fullgrad {k}: rel {e:.2e}" for e, k in full[:25]]
    _diag(f"diag_tiny_{mode}.txt", lines)
    assert err < 1e-3, f"train output rel err {err}"
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
    assert full[0][0] < 5e-3, f"worst full grad {full[0]}"
    assert rel.max() < 5e-3, f"worst grad norm {names[worst[0]]} rel {rel.max()}"
    bufs = dict(m.named_buffers())
    for k in g.files:
        if k.startswith("buf/"):
            np.testing.assert_allclose(bufs[k[4:]].cpu().numpy(), g[k], rtol=1e-3, atol=1e-5, err_msg=k)
    bn = np.array([float(v.double().norm()) for _, v in m.named_buffers()])
    np.testing.assert_allclose(bn, g["buffernorm_all"], rtol=1e-3)


@pytest.mark.parametrize("tag,hw,sigma", [("256x192", (256, 192), 2.0), ("384x288", (384, 288), 3.0)])
def test_w32_fp32_argmax_bit_exact(golden_dir, tag, hw, sigma):
    """BASELINE configs[0]/[1] shapes at bs 2: train-mode fwd + MSE + bwd."""
    g = np.load(os.path.join(golden_dir, f"g3_w32_{tag}.npz"))
    img, tgt, tw = synth_batch(2, hw[0], hw[1], seed=1234, sigma=sigma)
    m = _load_synth(PoseHighResolutionNet("w32", "fp32")).cuda().train()
    out = m(torch.from_numpy(img).cuda())
    loss = PersonMSELoss()(out, torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda())
    loss.backward()
    torch.cuda.synchronize()
    o = out.detach().cpu().numpy()
    p, mv = get_max_preds_hrnet(o)
    err = np.abs(o.reshape(-1)[::64] - g["out_sample"]).max() / float(g["out_absmax"])
    gn = {}
    for k, prm in m.named_parameters():
        top = k.split(".")[0]
        gn[top] = gn.get(top, 0.0) + float((prm.grad.double() ** 2).sum())
    vals = np.array([np.sqrt(gn[k]) for k in g["gradnorm_keys"]])
    _diag(f"diag_w32_{tag}.txt", [f"out rel err {err:.3e}", f"loss {loss.item()} ref {float(g['loss'])}",
                                  f"argmax equal {np.array_equal(p, g['argmax_xy'])}",
                                  "gradnorm " + " ".join(f"{k}:{v:.4e}/{r:.4e}" for k, v, r in zip(g["gradnorm_keys"], vals, g["gradnorm_vals"]))])
    assert err < 1e-3
    assert np.array_equal(p, g["argmax_xy"]), "heatmap argmax differs from the reference"
    np.testing.assert_allclose(mv, g["maxvals"], rtol=1e-3)
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * float(g["loss"])
    np.testing.assert_allclose(vals, g["gradnorm_vals"], rtol=5e-3)
    np.testing.assert_allclose(m.bn1.running_mean.cpu().numpy(), g["rm_bn1"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(m.stage4[2].branches[0][3].bn2.running_var.cpu().numpy(), g["rv_last"], rtol=1e-3)
    # eval + flip-test on fresh weights
    m2 = _load_synth(PoseHighResolutionNet("w32", "fp32")).cuda().eval()
    with torch.no_grad():
        oe = forward_pass(m2, torch.from_numpy(img).cuda(), "HRNet", device="cuda", flip=False).cpu().numpy()
        of = forward_pass(m2, torch.from_numpy(img).cuda(), "HRNet", device="cuda", flip=True).cpu().numpy()
    pe, _ = get_max_preds_hrnet(oe)
    pf, _ = get_max_preds_hrnet(of)
    assert np.array_equal(pe, g["eval_argmax_xy"])
    assert np.array_equal(pf, g["flip_argmax_xy"])
    assert np.abs(of.reshape(-1)[::64] - g["flip_sample"]).max() < 1e-3 * np.abs(g["flip_sample"]).max()


def test_w32_bf16_close_to_fp32_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_w32_256x192.npz"))
    img, tgt, tw = synth_batch(2, 256, 192, seed=1234, sigma=2.0)
    m = _load_synth(PoseHighResolutionNet("w32", "bf16")).cuda().train()
    out = m(torch.from_numpy(img).cuda())
    loss = PersonMSELoss()(out, torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda())
    loss.backward()
    o = out.detach().cpu().numpy()
    err = np.abs(o.reshape(-1)[::64] - g["out_sample"]).max() / float(g["out_absmax"])
    p, _ = get_max_preds_hrnet(o)
    agree = float((p == g["argmax_xy"]).all(-1).mean())
    gn = {}
    for k, prm in m.named_parameters():
        top = k.split(".")[0]
        gn[top] = gn.get(top, 0.0) + float((prm.grad.double() ** 2).sum())
    vals = np.array([np.sqrt(gn[k]) for k in g["gradnorm_keys"]])
    _diag("diag_w32_bf16.txt", [f"out rel err {err:.3e}", f"argmax agreement {agree:.3f}", f"loss {loss.item()} ref {float(g['loss'])}",
                                "gradnorm " + " ".join(f"{k}:{v:.4e}/{r:.4e}" for k, v, r in zip(g["gradnorm_keys"], vals, g["gradnorm_vals"]))])
    # Calibration (round 3): the fp32 oracle re-run with bf16 rounding at the HIP path's storage points.  The bars are that
    # run's own distance from the reference fixture (x 1.5 for the output, - 10 % of the maps for the argmax), not free constants.
    from oracle import hrnet_ref
    ref = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("w32")).train()
    with torch.no_grad(), hrnet_ref.bf16_storage(ref):
        oe = ref(torch.from_numpy(img)).numpy()
    err_emul = np.abs(oe.reshape(-1)[::64] - g["out_sample"]).max() / float(g["out_absmax"])
    pe, _ = get_max_preds_hrnet(oe)
    agree_emul = float((pe == g["argmax_xy"]).all(-1).mean())
    _diag("diag_w32_bf16_emulation.txt", [f"emulation: out rel err {err_emul:.3e}, argmax agreement {agree_emul:.3f}; HIP bf16: {err:.3e}, {agree:.3f}"])
    assert err < 1.5 * err_emul + 1e-3, (err, err_emul)
    assert agree >= agree_emul - 0.1, (agree, agree_emul)   # random-weight heat maps have near-ties; bf16 storage flips some of them
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * float(g["loss"])
    np.testing.assert_allclose(vals, g["gradnorm_vals"], rtol=0.1)


@pytest.mark.parametrize("tag,hw,sigma", [("256x192", (256, 192), 2.0), ("384x288", (384, 288), 3.0)])
def test_w32_mixed_vs_reference_golden(golden_dir, tag, hw, sigma):
    """The mixed 16-bit mode (forward tensors f16, gradients bf16) against the REFERENCE's fixture with ABSOLUTE
    bars: train-mode forward + MSE + backward at BASELINE configs[0]/[1] shapes (bs 2).  Measured on MI355X: output 4.5e-3 /
    5.8e-3 of |out|max (the pure-bf16 path: 4.0e-2 / 4.3e-2), loss 1.5e-5 / 2.7e-5 relative."""
    g = np.load(os.path.join(golden_dir, f"g3_w32_{tag}.npz"))
    img, tgt, tw = synth_batch(2, hw[0], hw[1], seed=1234, sigma=sigma)
    m = _load_synth(PoseHighResolutionNet("w32", "mixed")).cuda().train()
    out = m(torch.from_numpy(img).cuda())
    loss = PersonMSELoss()(out, torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda())
    loss.backward()
    torch.cuda.synchronize()
    o = out.detach().cpu().numpy()
    err = np.abs(o.reshape(-1)[::64] - g["out_sample"]).max() / float(g["out_absmax"])
    p, _ = get_max_preds_hrnet(o)
    agree = float((p == g["argmax_xy"]).all(-1).mean())
    gn = {}
    for k, prm in m.named_parameters():
        assert not torch.isnan(prm.grad).any(), k
        top = k.split(".")[0]
        gn[top] = gn.get(top, 0.0) + float((prm.grad.double() ** 2).sum())
    vals = np.array([np.sqrt(gn[k]) for k in g["gradnorm_keys"]])
    _diag(f"diag_w32_mixed_{tag}.txt", [f"out rel err {err:.3e}", f"argmax agreement {agree:.3f}", f"loss {loss.item()} ref {float(g['loss'])}",
                                        "gradnorm " + " ".join(f"{k}:{v:.4e}/{r:.4e}" for k, v, r in zip(g["gradnorm_keys"], vals, g["gradnorm_vals"]))])
    assert err < 8e-3, f"mixed-mode output error {err:.3e}"   # measured 4.5e-3 / 5.8e-3
    assert agree >= 0.85, f"argmax agreement {agree:.3f} (random-weight heat maps have near-ties)"
    assert abs(loss.item() - float(g["loss"])) < 2e-4 * float(g["loss"])
    np.testing.assert_allclose(vals, g["gradnorm_vals"], rtol=5e-2)   # gradients are bf16 like in the pure mode
    np.testing.assert_allclose(m.bn1.running_mean.cpu().numpy(), g["rm_bn1"], rtol=2e-3, atol=1e-4)


def test_tiny_mixed_full_gradients_vs_reference_golden(golden_dir):
    """Every full gradient the tiny fixture holds, through the mixed mode: exercises every backward kernel's second element type
    (BatchNorm-backward source y, mask_y, mask_z, the weight gradient's h, the sums' z, the head's x are f16; dt, du, dx bf16)."""
    g = np.load(os.path.join(golden_dir, "g1_tiny_train.npz"))
    m = _load_synth(PoseHighResolutionNet("tiny", "mixed")).cuda().train()
    out = m(torch.from_numpy(g["img"]).cuda())
    loss = PersonMSELoss()(out, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["target_weight"]).cuda())
    loss.backward()
    torch.cuda.synchronize()
    err = np.abs(out.detach().cpu().numpy() - g["output"]).max() / np.abs(g["output"]).max()
    grads = {k: p.grad for k, p in m.named_parameters()}
    worst = []
    for k in g.files:
        if k.startswith("grad/"):
            ref = g[k]
            got = grads[k[5:]].cpu().numpy()
            den = max(np.abs(ref).max(), 1e-12)
            cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
            worst.append((np.abs(got - ref).max() / den, cos, k))
    worst.sort(reverse=True)
    _diag("diag_tiny_mixed.txt", [f"out rel err {err:.3e} loss {loss.item()} ref {float(g['loss'])}"] + [f"{k}: rel {e:.2e} cos {c:.6f}" for e, c, k in worst[:20]])
    assert err < 1e-2 and abs(loss.item() - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
    # bs 2 on a 32x24 input: a few hundred pixels per weight, so bf16 gradients are noisy here (tools/grad_probe.py on MI355X: pure
    # bf16 worst tensor 0.75 of its largest element, median 0.25, cosine >= 0.939; mixed 0.39 / 0.10 / >= 0.9917; fp32 1.7e-3)
    errs = np.array([e for e, _, _ in worst])
    assert errs.max() < 0.6 and np.median(errs) < 0.15 and min(c for _, c, _ in worst) > 0.985, worst[:3]


@pytest.mark.parametrize("dt", ["fp32", "mixed"])
def test_poisoned_plan_buffers_change_nothing(golden_dir, dt, monkeypatch):
    """STLPOSE_POISON=1: every planned activation / gradient buffer starts as NaNs.  Output, loss and every gradient must come
    out as in the clean run, bit for bit: no kernel consumes a location that nobody wrote (halo slots, tile padding, unused
    channel tails are masked AFTER they are loaded)."""
    g = np.load(os.path.join(golden_dir, "g1_tiny_train.npz"))

    def run(poison):
        monkeypatch.setenv("STLPOSE_POISON", poison)
        m = _load_synth(PoseHighResolutionNet("tiny", dt)).cuda().train()
        out = m(torch.from_numpy(g["img"]).cuda())
        loss = PersonMSELoss()(out, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["target_weight"]).cuda())
        loss.backward()
        torch.cuda.synchronize()
        return out.detach().clone(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    o0, g0 = run("0")
    o1, g1 = run("1")
    assert not torch.isnan(o1).any() and not torch.isnan(g1).any()
    assert torch.equal(o0, o1)
    # gradients: the fp64 statistics atomics arrive in a different order from run to run
    assert float((g0 - g1).abs().max()) <= 1e-5 * float(g0.abs().max())


def test_w48_eval_fp32_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_w48.npz"))
    m = _load_synth(PoseHighResolutionNet("w48", "fp32")).cuda().eval()
    assert sum(p.numel() for p in m.parameters()) == int(g["nparams"])
    img, _, _ = synth_batch(1, 128, 96, seed=5)
    with torch.no_grad():
        o = m(torch.from_numpy(img).cuda()).cpu().numpy()
    ref = g["out_sample"]
    assert np.abs(o.reshape(-1)[::16] - ref).max() < 1e-3 * np.abs(ref).max()


def test_cpu_input_fails_loudly():
    m = PoseHighResolutionNet("tiny", "fp32")
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 3, 64, 64))
