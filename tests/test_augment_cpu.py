"""CPU: the host half of the GPU crop / augmentation (stlpose_amd/augment.py) against fixture G11 = the reference's
own lib/transforms.py arithmetic (get_affine_transform, affine_transform, fliplr_joints), and the warp oracle."""
import os

import numpy as np

from oracle import pose_ref
from stlpose_amd import augment


def test_affine_matrices_and_joint_transforms_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g11_affine.npz"))
    n = len(g["rots"])
    for si, size in enumerate(g["sizes"]):
        for i in range(n):
            t = augment.get_affine_transform(g["centers"][i], g["scales"][i], g["rots"][i], size)
            ti = augment.get_affine_transform(g["centers"][i], g["scales"][i], g["rots"][i], size, inv=1)
            np.testing.assert_allclose(t, g[f"trans_{si}"][i], rtol=1e-10, atol=1e-9)
            np.testing.assert_allclose(ti, g[f"trans_inv_{si}"][i], rtol=1e-10, atol=1e-9)
    for i in range(n):
        j, v = augment.fliplr_joints(g["joints"][i].copy(), g["joints_vis"][i].copy(), int(g["widths"][i]))
        np.testing.assert_array_equal(j, g["flipped_joints"][i])
        np.testing.assert_array_equal(v, g["flipped_vis"][i])
        tj = np.stack([augment.affine_transform(g["joints"][i, k, 0:2], g["trans_1"][i]) for k in range(17)])
        np.testing.assert_allclose(tj, g["transformed_joints"][i], rtol=1e-12, atol=1e-9)


def test_warp_oracle_identity_shift_and_flip():
    rng = np.random.Generator(np.random.PCG64(3))
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    np.testing.assert_array_equal(pose_ref.warp_affine_bilinear(img, ident, (30, 20)), img.astype(np.float32))
    shift = np.array([[1.0, 0, 2.0], [0, 1.0, 1.0]])          # dst(x, y) = src(x - 2, y - 1), zeros shifted in
    out = pose_ref.warp_affine_bilinear(img, shift, (30, 20))
    np.testing.assert_array_equal(out[1:, 2:], img[:-1, :-2].astype(np.float32))
    assert not out[0].any() and not out[:, :2].any()
    np.testing.assert_array_equal(pose_ref.warp_affine_bilinear(img, ident, (30, 20), flip=True), img[:, ::-1].astype(np.float32))
    half = np.array([[1.0, 0, 0.5], [0, 1.0, 0]])             # half-pixel shift: average of horizontal neighbours
    out = pose_ref.warp_affine_bilinear(img, half, (30, 20))
    np.testing.assert_allclose(out[:, 1:], 0.5 * (img[:, :-1].astype(np.float32) + img[:, 1:]), rtol=0, atol=1e-4)


def test_half_body_transform_matches_reference(golden_dir):
    """JointsDataset.half_body_transform (data/JointsDataset.py:75-130), numpy's global generator seeded like the fixture."""
    g = np.load(os.path.join(golden_dir, "g11_affine.npz"))
    for i in range(len(g["rots"])):
        np.random.seed(500 + i)
        c, s = augment.half_body_transform(g["joints"][i].copy(), g["joints_vis"][i].copy(), 192.0 / 256.0)
        assert (c is not None) == bool(g["hb_ok"][i])
        if c is not None:
            np.testing.assert_allclose(c, g["hb_center"][i], rtol=1e-7)
            np.testing.assert_allclose(s, g["hb_scale"][i], rtol=1e-7)
    vis_few = np.zeros((17, 3))
    vis_few[12, :2] = 1
    np.random.seed(77)
    assert augment.half_body_transform(g["joints"][0].copy(), vis_few, 0.75) == (None, None) and bool(g["hb_few_none"])
