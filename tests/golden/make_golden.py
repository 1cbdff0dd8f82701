#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box has neither it nor this
need).  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does: puts /root/reference/src on sys.path, pre-seeds an empty
``models`` package (the reference's models/__init__.py imports a file that is not in its tree),
provides a minimal dict-with-attributes object under the name ``yacs.config.CfgNode`` and empty
``torchvision`` / ``cv2`` module objects so that the reference's *own* files import unchanged,
writes the W32 / tiny architecture YAMLs to a temp dir, then calls the reference's
PoseHighResolutionNet, PersonMSELoss, forward_pass, flip_back, get_max_preds_hrnet,
get_final_preds_hrnet (cv2.getAffineTransform replaced by a 3-point linear solve) and oks_nms
on seeded inputs.  Weights come from oracle.hrnet_ref.synth_tensor (numpy PCG64 per key), so
no checkpoint is shipped.  Only inputs/outputs are stored -- no reference source text.
"""
from __future__ import annotations

import os
import re
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import hrnet_ref  # noqa: E402


# ------------------------------------------------------------------ harness-side shims
class _Node(dict):
    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = _Node(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def defrost(self):
        pass

    def freeze(self):
        pass

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def _merge(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], _Node):
                    self[k] = _Node()
                self[k]._merge(v)
            else:
                self[k] = v


def _install_shims(tmpdir):
    sys.path.insert(0, REF)
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    yacs = types.ModuleType("yacs")
    yacs_cfg = types.ModuleType("yacs.config")
    yacs_cfg.CfgNode = _Node
    yacs.config = yacs_cfg
    sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yacs_cfg
    for name in ("torchvision", "torchvision.transforms", "cv2"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    import CONFIG as ref_config
    ref_config.CONFIG["paths"]["pretrained_path"] = tmpdir


def _write_yaml(tmpdir, arch):
    import yaml
    a = hrnet_ref.ARCHS[arch]
    w, nb = a["widths"], a["blocks"]

    def st(n, mods):
        return dict(NUM_MODULES=mods, NUM_BRANCHES=n, BLOCK="BASIC", NUM_BLOCKS=[nb] * n,
                    NUM_CHANNELS=w[:n], FUSE_METHOD="SUM")
    doc = dict(MODEL=dict(NUM_JOINTS=a["joints"], EXTRA=dict(
        PRETRAINED_LAYERS=["*"], FINAL_CONV_KERNEL=1,
        STAGE2=st(2, a["modules"][0]), STAGE3=st(3, a["modules"][1]), STAGE4=st(4, a["modules"][2]))))
    os.makedirs(os.path.join(tmpdir, "HRnet"), exist_ok=True)
    with open(os.path.join(tmpdir, "HRnet", "cfg_hrnet_w32_256x192.yaml"), "w") as f:
        yaml.safe_dump(doc, f)


def _ref_model(tmpdir, arch):
    _write_yaml(tmpdir, arch)
    from models.HRnet import PoseHighResolutionNet
    torch.manual_seed(0)
    m = PoseHighResolutionNet(is_train=False)
    hrnet_ref.load_synth(m)
    return m


# ------------------------------------------------------------------ synthetic batch
def synth_batch(b, h, w, joints=17, seed=1234, sigma=2.0):
    """randn images, unnormalised gaussians at random in-bounds centres,
    Bernoulli(0.8) joint weights."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = rng.standard_normal((b, 3, h, w)).astype(np.float32)
    hh, ww = h // 4, w // 4
    ys, xs = np.mgrid[0:hh, 0:ww].astype(np.float32)
    tgt = np.zeros((b, joints, hh, ww), np.float32)
    for n in range(b):
        for j in range(joints):
            cx, cy = rng.integers(0, ww), rng.integers(0, hh)
            tgt[n, j] = np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sigma ** 2))
    tw = (rng.random((b, joints, 1)) < 0.8).astype(np.float32)
    return img, tgt, tw


FULL_GRAD_KEYS = [
    r"conv1\.weight", r"bn1\.(weight|bias)", r"conv2\.weight", r"layer1\.0\.conv[123]\.weight",
    r"layer1\.0\.downsample\.0\.weight", r"layer1\.0\.downsample\.1\.(weight|bias)", r"layer1\.3\.bn3\.(weight|bias)",
    r"transition1\.[01]\..*weight", r"transition3\.3\.0\.0\.weight",
    r"stage2\.0\.branches\.0\.0\.(conv1|conv2)\.weight", r"stage2\.0\.branches\.1\.1\.bn[12]\.(weight|bias)",
    r"stage2\.0\.fuse_layers\..*", r"stage3\.1\.fuse_layers\.2\.0\..*", r"stage3\.0\.fuse_layers\.0\.2\..*",
    r"stage4\.0\.fuse_layers\.3\.0\..*", r"stage4\.1\.fuse_layers\.0\.3\..*", r"stage4\.1\.branches\.3\.1\..*",
    r"final_layer\.(weight|bias)",
]


def _grad_norms(model):
    out = {}
    for k, p in model.named_parameters():
        top = k.split(".")[0]
        out[top] = out.get(top, 0.0) + float((p.grad.double() ** 2).sum())
    return {k: np.sqrt(v) for k, v in out.items()}


def main():
    torch.set_num_threads(8)
    tmp = tempfile.mkdtemp(prefix="stl_golden_")
    _install_shims(tmp)
    from lib.loss import PersonMSELoss
    from lib.inference import forward_pass
    from lib.pose_parsing import get_max_preds_hrnet, get_final_preds_hrnet
    import lib.transforms as ref_tf
    from lib.nms import oks_nms

    # cv2.getAffineTransform stand-in: exact solve of the 3-point system the reference sets up
    def _get_affine(src, dst):
        a = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], 1)
        return np.linalg.solve(a, np.asarray(dst, np.float64)).T
    sys.modules["cv2"].getAffineTransform = _get_affine
    crit = PersonMSELoss()

    # ---- G1/G2: tiny net, train + eval, full outputs, all grads, BN buffers after one step
    for mode in ("train", "eval"):
        m = _ref_model(tmp, "tiny")
        m.train(mode == "train")
        img, tgt, tw = synth_batch(2, 96, 64, seed=11)
        out = m(torch.from_numpy(img))
        loss = crit(out, torch.from_numpy(tgt), torch.from_numpy(tw))
        loss.backward()
        fx = dict(img=img, target=tgt, target_weight=tw, output=out.detach().numpy(),
                  loss=np.float64(loss.item()))
        # full gradients for one representative of every layer kind, L2 norms for all
        names = [k for k, _ in m.named_parameters()]
        fx["param_keys"] = np.array(names)
        fx["gradnorm_all"] = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
        for k, p in m.named_parameters():
            if any(re.fullmatch(pat, k) for pat in FULL_GRAD_KEYS):
                fx["grad/" + k] = p.grad.numpy()
        bnames = [k for k, _ in m.named_buffers()]
        fx["buffer_keys"] = np.array(bnames)
        fx["buffernorm_all"] = np.array([float(v.double().norm()) for _, v in m.named_buffers()])
        for k, v in m.named_buffers():
            if k.startswith(("bn1.", "bn2.", "stage4.1.fuse_layers.3.0.2.1", "stage3.0.branches.2.1.bn2")):
                fx["buf/" + k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, f"g1_tiny_{mode}.npz"), **fx)
        print("G1", mode, "loss", loss.item(), "out", out.shape)

    # ---- G3/G4/G5: full W32 at 256x192 (cfg1) and 384x288, train-mode fwd+loss+bwd, eval flip-test
    for tag, (h, w) in (("256x192", (256, 192)), ("384x288", (384, 288))):
        m = _ref_model(tmp, "w32")
        m.train()
        img, tgt, tw = synth_batch(2, h, w, seed=1234, sigma=2.0 if h == 256 else 3.0)
        out = m(torch.from_numpy(img))
        loss = crit(out, torch.from_numpy(tgt), torch.from_numpy(tw))
        loss.backward()
        o = out.detach().numpy()
        preds, maxvals = get_max_preds_hrnet(o)
        fx = dict(seed=1234, hw=np.array([h, w]), loss=np.float64(loss.item()),
                  argmax_xy=preds, maxvals=maxvals,
                  out_mean=o.mean(), out_std=o.std(), out_absmax=np.abs(o).max(),
                  out_sample=o.reshape(-1)[::64].copy(),
                  out_b0j0=o[0, 0].copy())
        gn = _grad_norms(m)
        fx["gradnorm_keys"] = np.array(sorted(gn))
        fx["gradnorm_vals"] = np.array([gn[k] for k in sorted(gn)])
        fx["rm_bn1"] = m.bn1.running_mean.numpy()
        fx["rv_bn1"] = m.bn1.running_var.numpy()
        fx["rm_last"] = m.stage4[2].branches[0][3].bn2.running_mean.numpy()
        fx["rv_last"] = m.stage4[2].branches[0][3].bn2.running_var.numpy()
        # eval-mode plain + flip-test (G5) on the same weights (fresh running stats)
        m2 = _ref_model(tmp, "w32")
        m2.eval()
        with torch.no_grad():
            oe = forward_pass(m2, torch.from_numpy(img), "HRNet", device="cpu", flip=False).numpy()
            of = forward_pass(m2, torch.from_numpy(img), "HRNet", device="cpu", flip=True).numpy()
        pe, me = get_max_preds_hrnet(oe)
        pf, mf = get_max_preds_hrnet(of)
        fx.update(eval_argmax_xy=pe, eval_maxvals=me, eval_sample=oe.reshape(-1)[::64].copy(),
                  flip_argmax_xy=pf, flip_maxvals=mf, flip_sample=of.reshape(-1)[::64].copy())
        np.savez_compressed(os.path.join(HERE, f"g3_w32_{tag}.npz"), **fx)
        print("G3", tag, "loss", loss.item(), "absmax", np.abs(o).max())

    # ---- G8: W48 key list / param count / output stats
    m = _ref_model(tmp, "w48")
    m.eval()
    img, _, _ = synth_batch(1, 128, 96, seed=5)
    with torch.no_grad():
        o = m(torch.from_numpy(img)).numpy()
    keys = list(m.state_dict().keys())
    shapes = [tuple(v.shape) for v in m.state_dict().values()]
    np.savez_compressed(os.path.join(HERE, "g8_w48.npz"), nparams=sum(p.numel() for p in m.parameters()),
                        nkeys=len(keys), out_sample=o.reshape(-1)[::16].copy(), out_mean=o.mean(), out_std=o.std(),
                        key_crc=np.array([__import__("zlib").crc32((k + str(s)).encode()) for k, s in zip(keys, shapes)]))
    m32 = _ref_model(tmp, "w32")
    keys32 = list(m32.state_dict().keys())
    shapes32 = [tuple(v.shape) for v in m32.state_dict().values()]
    with open(os.path.join(HERE, "g8_w32_keys.txt"), "w") as f:
        for k, s in zip(keys32, shapes32):
            f.write(f"{k} {'x'.join(map(str, s))}\n")
    print("G8 w48 params", sum(p.numel() for p in m.parameters()), "w32 keys", len(keys32))

    # ---- G6: PersonMSELoss on random tensors incl. zero-weight joints and B=1
    rng = np.random.Generator(np.random.PCG64(77))
    g6 = {}
    for name, b in (("b4", 4), ("b1", 1)):
        o = rng.standard_normal((b, 17, 16, 12)).astype(np.float32)
        t = rng.standard_normal((b, 17, 16, 12)).astype(np.float32)
        w = (rng.random((b, 17, 1)) < 0.7).astype(np.float32)
        w[:, 3] = 0
        ot = torch.from_numpy(o).requires_grad_(True)
        l = crit(ot, torch.from_numpy(t), torch.from_numpy(w))
        l.backward()
        g6.update({f"{name}_o": o, f"{name}_t": t, f"{name}_w": w, f"{name}_loss": np.float64(l.item()),
                   f"{name}_grad": ot.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "g6_mse.npz"), **g6)

    # ---- G7: get_max_preds_hrnet incl. ties and all-negative maps; final preds; flip_back; oks_nms
    hm = rng.standard_normal((3, 17, 16, 12)).astype(np.float32)
    hm[0, 0] = -1.0                      # all negative -> coordinates masked to 0
    hm[0, 1] = 0.0                       # all zero (max == 0 -> masked)
    hm[0, 2] = 0.5; hm[0, 2, 3, 4] = 2.0; hm[0, 2, 9, 7] = 2.0   # tie -> first index wins
    hm[1, 5, 15, 11] = 9.0               # last element
    hm[1, 6, 0, 0] = 9.0                 # first element
    p, mv = get_max_preds_hrnet(hm)
    center = np.array([[100.0, 120.0], [55.5, 80.25], [300.0, 10.0]])
    scale = np.array([[1.2, 1.6], [0.75, 1.0], [2.0, 2.6667]])
    fp, fmv, fcoords = get_final_preds_hrnet(hm, center, scale)
    fb = ref_tf.flip_back(torch.from_numpy(hm.copy()), [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]).numpy()
    kp = rng.random((6, 17, 3)) * np.array([200, 300, 1.0])
    kp[1] = kp[0] + rng.normal(0, 1.0, (17, 3)) * np.array([1, 1, 0])
    kp[4] = kp[3] + rng.normal(0, 30.0, (17, 3)) * np.array([1, 1, 0])
    sc = rng.random(6)
    ar = 4000 + 2000 * rng.random(6)
    db = [dict(keypoints=kp[i], score=sc[i], area=ar[i]) for i in range(6)]
    keep = {f"keep_{str(t).replace('.', '')}": np.array(oks_nms(db, t)) for t in (0.9, 0.5)}
    np.savez_compressed(os.path.join(HERE, "g7_decode.npz"), hm=hm, preds=p, maxvals=mv, center=center, scale=scale,
                        final_preds=fp, final_coords=fcoords, flip_back=fb, nms_kpts=kp, nms_scores=sc,
                        nms_areas=ar, **keep)
    print("G6/G7 done; fixtures in", HERE)


def gen_g9():
    """G9: JointsDataset.generate_target (data/JointsDataset.py:230-286) called on the reference's own
    class (file loaded directly: data/__init__.py pulls in pycocotools) with a stand-in `self`."""
    import importlib.util
    tmp = tempfile.mkdtemp()
    if "CONFIG" not in sys.modules:
        _install_shims(tmp)
    spec = importlib.util.spec_from_file_location("ref_joints_dataset", os.path.join(REF, "data", "JointsDataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.Generator(np.random.PCG64(99))
    out = {}
    for tag, sigma, hm, img in (("s2", 2, (48, 64), (192, 256)), ("s3", 3, (72, 96), (288, 384))):
        me = types.SimpleNamespace(num_joints=17, target_type="gaussian", heatmap_size=np.array(hm), image_size=np.array(img),
                                   sigma=sigma, use_different_joints_weight=False)
        B = 5
        joints = np.zeros((B, 17, 3))
        joints[..., :2] = (rng.uniform(-40, 1.15 * max(img), size=(B, 17, 2)) * 4).round() / 4
        joints[0, 0, :2] = [0.0, 0.0]
        joints[0, 1, :2] = [img[0] - 1, img[1] - 1]
        joints[0, 2, :2] = [-1000.0, 50.0]
        joints[0, 3, :2] = [img[0] + 3 * sigma * 4 + 8, 10.0]
        vis = np.zeros((B, 17, 3))
        vis[..., 0] = vis[..., 1] = (rng.uniform(size=(B, 17)) < 0.8)
        tg, tw = zip(*[mod.JointsDataset.generate_target(me, joints[b], vis[b]) for b in range(B)])
        out.update({f"{tag}_joints": joints, f"{tag}_vis": vis, f"{tag}_target": np.stack(tg), f"{tag}_tw": np.stack(tw),
                    f"{tag}_cfg": np.array([sigma, *hm, *img], np.float64)})
    np.savez_compressed(os.path.join(HERE, "g9_targets.npz"), **out)
    print("G9 done")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g9":
        gen_g9()
    else:
        main()
        gen_g9()
