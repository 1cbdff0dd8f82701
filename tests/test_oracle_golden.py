"""CPU: the oracle (oracle/*.py) against the fixtures produced by running the reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import hrnet_ref, pose_ref

torch.set_num_threads(8)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_state_dict_abi_w32(golden_dir):
    """names + shapes of all 1754 entries equal the reference's."""
    m = hrnet_ref.RefPoseNet("w32")
    lines = [f"{k} {'x'.join(map(str, v.shape))}" for k, v in m.state_dict().items()]
    with open(os.path.join(golden_dir, "g8_w32_keys.txt")) as f:
        ref = f.read().splitlines()
    assert len(lines) == 1754
    assert lines == ref
    assert sum(p.numel() for p in m.parameters()) == 28536113


def test_state_dict_abi_w48(golden_dir):
    g = _load(golden_dir, "g8_w48.npz")
    m = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("w48")).eval()
    assert sum(p.numel() for p in m.parameters()) == int(g["nparams"]) == 63595745
    crc = np.array([zlib.crc32((k + str(tuple(v.shape))).encode()) for k, v in m.state_dict().items()])
    assert np.array_equal(crc, g["key_crc"])
    from tests.golden.make_golden import synth_batch
    img, _, _ = synth_batch(1, 128, 96, seed=5)
    with torch.no_grad():
        o = m(torch.from_numpy(img)).numpy()
    np.testing.assert_allclose(o.reshape(-1)[::16], g["out_sample"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_tiny_forward_backward(golden_dir, mode):
    g = _load(golden_dir, f"g1_tiny_{mode}.npz")
    m = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("tiny"))
    m.train(mode == "train")
    out = m(torch.from_numpy(g["img"]))
    loss = pose_ref.person_mse_loss(out, torch.from_numpy(g["target"]), torch.from_numpy(g["target_weight"]))
    loss.backward()
    scale = np.abs(g["output"]).max()
    assert np.abs(out.detach().numpy() - g["output"]).max() <= 1e-5 * scale
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    names = [k for k, _ in m.named_parameters()]
    assert names == list(g["param_keys"])
    norms = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
    np.testing.assert_allclose(norms, g["gradnorm_all"], rtol=2e-4, atol=1e-7)
    grads = dict(m.named_parameters())
    nfull = 0
    for k in g.files:
        if k.startswith("grad/"):
            ref = g[k]
            got = grads[k[5:]].grad.numpy()
            assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6), k
            nfull += 1
    assert nfull > 30
    bufs = dict(m.named_buffers())
    bn = np.array([float(v.double().norm()) for _, v in m.named_buffers()])
    np.testing.assert_allclose(bn, g["buffernorm_all"], rtol=1e-5)
    for k in g.files:
        if k.startswith("buf/"):
            np.testing.assert_allclose(bufs[k[4:]].numpy(), g[k], rtol=1e-5, atol=1e-6)


def test_w32_cfg1_argmax_and_stats(golden_dir):
    """BASELINE cfg1: W32 256x192 bs 2 fwd + MSE + bwd on CPU; argmax bit-exact."""
    from tests.golden.make_golden import synth_batch
    g = _load(golden_dir, "g3_w32_256x192.npz")
    m = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("w32")).train()
    img, tgt, tw = synth_batch(2, 256, 192, seed=1234, sigma=2.0)
    out = m(torch.from_numpy(img))
    loss = pose_ref.person_mse_loss(out, torch.from_numpy(tgt), torch.from_numpy(tw))
    loss.backward()
    o = out.detach().numpy()
    p, mv = pose_ref.get_max_preds(o)
    assert np.array_equal(p, g["argmax_xy"])
    np.testing.assert_allclose(mv, g["maxvals"], rtol=1e-4)
    np.testing.assert_allclose(o.reshape(-1)[::64], g["out_sample"], rtol=0, atol=1e-4 * float(g["out_absmax"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    gn = {}
    for k, prm in m.named_parameters():
        top = k.split(".")[0]
        gn[top] = gn.get(top, 0.0) + float((prm.grad.double() ** 2).sum())
    vals = np.array([np.sqrt(gn[k]) for k in g["gradnorm_keys"]])
    np.testing.assert_allclose(vals, g["gradnorm_vals"], rtol=1e-3)
    np.testing.assert_allclose(m.bn1.running_mean.numpy(), g["rm_bn1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m.stage4[2].branches[0][3].bn2.running_var.numpy(), g["rv_last"], rtol=1e-4)


def test_w32_flip_test_eval(golden_dir):
    from tests.golden.make_golden import synth_batch
    g = _load(golden_dir, "g3_w32_256x192.npz")
    m = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("w32")).eval()
    img, _, _ = synth_batch(2, 256, 192, seed=1234, sigma=2.0)
    with torch.no_grad():
        of = pose_ref.forward_pass(m, torch.from_numpy(img), flip=True).numpy()
    p, mv = pose_ref.get_max_preds(of)
    assert np.array_equal(p, g["flip_argmax_xy"])
    np.testing.assert_allclose(of.reshape(-1)[::64], g["flip_sample"], rtol=0, atol=2e-4 * np.abs(g["flip_sample"]).max())


def test_mse_loss_golden(golden_dir):
    g = _load(golden_dir, "g6_mse.npz")
    for name in ("b4", "b1"):
        o = torch.from_numpy(g[f"{name}_o"]).requires_grad_(True)
        l = pose_ref.person_mse_loss(o, torch.from_numpy(g[f"{name}_t"]), torch.from_numpy(g[f"{name}_w"]))
        l.backward()
        assert abs(l.item() - float(g[f"{name}_loss"])) < 1e-6 * abs(float(g[f"{name}_loss"]))
        np.testing.assert_allclose(o.grad.numpy(), g[f"{name}_grad"], rtol=1e-5, atol=1e-9)
        # closed form used by the HIP kernel
        w = torch.from_numpy(g[f"{name}_w"])[..., None]
        closed = 0.5 * (((o.detach() - torch.from_numpy(g[f"{name}_t"])) * w) ** 2).mean()
        assert abs(closed.item() - l.item()) < 1e-6 * abs(l.item())


def test_decode_golden(golden_dir):
    g = _load(golden_dir, "g7_decode.npz")
    p, mv = pose_ref.get_max_preds(g["hm"])
    assert np.array_equal(p, g["preds"]) and np.array_equal(mv, g["maxvals"])
    fp, fmv, coords = pose_ref.final_preds(g["hm"], g["center"], g["scale"])
    assert np.array_equal(coords, g["final_coords"])
    np.testing.assert_allclose(fp, g["final_preds"], rtol=1e-5, atol=1e-4)
    assert np.array_equal(pose_ref.flip_back(g["hm"]), g["flip_back"])
    kp = g["nms_kpts"].reshape(6, -1)
    for t, key in ((0.9, "keep_09"), (0.5, "keep_05")):
        assert pose_ref.oks_nms(kp, g["nms_scores"], g["nms_areas"], t) == list(g[key])


def test_pck_self_consistency():
    """accuracy is unpinned (corrupted reference line); sanity: perfect prediction -> 1.0."""
    rng = np.random.default_rng(0)
    t = np.zeros((2, 17, 24, 16), np.float32)
    for n in range(2):
        for j in range(17):
            t[n, j, rng.integers(2, 24), rng.integers(2, 16)] = 1.0
    acc, avg, cnt, _ = pose_ref.pck_accuracy(t.copy(), t)
    assert avg == 1.0 and cnt == 17


def test_g9_gaussian_targets_restatement_matches_reference(golden_dir):
    """oracle.pose_ref.gaussian_targets vs the reference's JointsDataset.generate_target
    (data/JointsDataset.py:230-286) on the G9 fixture (inside / border / outside / invisible joints)."""
    import numpy as np
    from oracle import pose_ref
    g = np.load(os.path.join(golden_dir, "g9_targets.npz"))
    for tag in ("s2", "s3"):
        sigma, wh, hh, wi, hi = g[f"{tag}_cfg"]
        for b in range(g[f"{tag}_joints"].shape[0]):
            t, w = pose_ref.gaussian_targets(g[f"{tag}_joints"][b, :, :2], g[f"{tag}_vis"][b, :, 0], (int(wh), int(hh)), (int(wi), int(hi)), float(sigma))
            assert np.array_equal(w, g[f"{tag}_tw"][b])
            assert np.array_equal(t, g[f"{tag}_target"][b])


def test_w48_train_step_oracle_vs_reference(golden_dir):
    """G8b: W48 training step produced by the reference (ragged 48/96/192/384 widths)."""
    from tests.golden.make_golden import synth_batch
    g = _load(golden_dir, "g8_w48_train.npz")
    img, tgt, tw = synth_batch(2, 128, 96, seed=48, sigma=2.0)
    m = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("w48")).train()
    out = m(torch.from_numpy(img))
    loss = pose_ref.person_mse_loss(out, torch.from_numpy(tgt), torch.from_numpy(tw))
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy().reshape(-1)[::16], g["out_sample"], rtol=1e-4, atol=1e-5 * float(g["out_absmax"]))
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    norms = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
    np.testing.assert_allclose(norms, g["gradnorm_all"], rtol=5e-4, atol=1e-7)


def test_pck_oracle_vs_reference_calc_dists(golden_dir):
    """G12: pose_ref.pck_accuracy against the reference's own calc_dists / dist_acc outputs."""
    g = _load(golden_dir, "g12_metrics.npz")
    acc, avg, cnt, pred = pose_ref.pck_accuracy(g["output"], g["target"])
    np.testing.assert_allclose(acc, g["acc"], rtol=0, atol=1e-12)
    assert abs(avg - float(g["avg_acc"])) < 1e-12 and cnt == int(g["cnt"]) and np.array_equal(pred, g["pred"])
    assert 0.0 < avg < 1.0 and (g["acc"] == -1).any()


def test_vgg_oracle_vs_reference_perceptual_loss(golden_dir):
    """G10: the reference's own VGGPerceptualLoss (its slicing / repeat / normalise / resize / L1 lines) run on a
    torchvision-free VGG16-D layer list with synthetic weights.  Pins oracle/vgg_ref.py."""
    from oracle import vgg_ref
    g = _load(golden_dir, "g10_vgg.npz")
    w = vgg_ref.synth_vgg_weights()
    for tag in ("rs_rgb", "rs_gray", "nr_rgb", "nr_odd", "nr_gray"):
        with torch.no_grad():
            l = vgg_ref.vgg_perceptual_loss(torch.from_numpy(g[f"{tag}_in"]), torch.from_numpy(g[f"{tag}_tg"]), w,
                                            resize=bool(g[f"{tag}_resize"]))
        assert abs(float(l) - float(g[f"{tag}_loss"])) <= 1e-5 * float(g[f"{tag}_loss"]), tag


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree exists in the build container only")
def test_golden_generator_reproduces_every_committed_fixture(tmp_path):
    """`python tests/golden/make_golden.py` (no argument, as documented) must reproduce all twelve .npz fixtures and the key list
    bit for bit -- the pin has to be regenerable by the documented command.  (Until round 4 the all-in-one run wrote a W32 net
    into g8_w48_train.npz: every gen_* made a new temp dir but the reference's CONFIG kept pointing at main()'s.)"""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STL_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_golden.py")], check=True, env=env, cwd=root,
                   stdout=subprocess.DEVNULL, timeout=900)
    committed = sorted(glob.glob(os.path.join(root, "tests", "golden", "*.npz")))
    assert len(committed) == 12
    for f in committed:
        a, b = np.load(f), np.load(os.path.join(str(tmp_path), os.path.basename(f)))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (f, k)
            assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), f"{os.path.basename(f)}[{k}] differs from the reference's output"
    assert open(os.path.join(root, "tests", "golden", "g8_w32_keys.txt")).read() == open(os.path.join(str(tmp_path), "g8_w32_keys.txt")).read()
    assert not glob.glob("/root/reference/**/__pycache__", recursive=True), "the generator wrote bytecode into the reference tree"
