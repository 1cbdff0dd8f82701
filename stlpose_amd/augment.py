"""Crop / augmentation half of the data pipeline on the GPU (reference
``data/JointsDataset.py:157-200`` + ``lib/transforms.py:167-250``).

The reference warps every sample on the host with ``cv2.warpAffine`` inside a ``num_workers=0`` loader
(``CONFIG.py:18``) -- two orders of magnitude below what the train step consumes.  Here the loader ships the
decoded uint8 image and a few scalars per person (center, scale, rotation, flip); the 2x3 matrices of the whole
batch come from one vectorised routine (``affine_matrices``: the reference's three point correspondences, solved
exactly as a batched 3x3 system, no cv2 and no per-sample loop), and one HIP launch per batch does the bilinear warp + ToTensor + Normalize into the fp32 NCHW batch the network reads.
Joint coordinates are transformed on the host (17 points per person) and turned into heatmaps by
``targets.generate_targets``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import capi, ops  # noqa: F401  (ops registers the stlpose:: custom ops)

FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]   # reference CONSTANTS.py:65
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)             # data_loaders.py:59-61


def affine_matrices(centers, scales, rots, output_size, inverse: bool = False) -> np.ndarray:
    """All B crop matrices at once (lib/transforms.py:197-231 per sample): the similarity that maps the rotated
    person box (centre c, width 200 * scale_x) onto the (W, H) crop, as the exact solution of the three point
    correspondences the reference hands to cv2.getAffineTransform -- box centre, the rotated "up" point half a box
    width above it, and that point turned by 90 degrees.  The control points are rounded to float32 exactly where
    the reference stores them in float32 arrays, so the fp64 matrices agree with its output to the last digits
    (fixture G11).  centers (B, 2), scales (B, 2) or (B,), rots (B,) degrees; returns (B, 2, 3) float64;
    inverse=True gives the crop -> image matrices (the reference's ``inv=1``)."""
    c = np.asarray(centers, np.float64).reshape(-1, 2)
    B = c.shape[0]
    sc = np.asarray(scales, np.float64)
    box_w = (np.broadcast_to(sc.reshape(B, -1), (B, max(sc.size // B, 1)))[:, 0]) * 200.0
    ang = np.pi * np.asarray(rots, np.float64).reshape(B) / 180
    up = -0.5 * box_w                                                       # "up" offset before rotation: (0, up)
    w_out, h_out = float(output_size[0]), float(output_size[1])

    def triangle(p0: np.ndarray, p1: np.ndarray) -> np.ndarray:
        p0, p1 = p0.astype(np.float32), p1.astype(np.float32)
        d = p0 - p1                                                          # float32 arithmetic, like the reference's arrays
        p2 = p1 + np.stack([-d[:, 1], d[:, 0]], 1)
        return np.stack([p0, p1, p2], 1)                                     # (B, 3, 2) float32

    img = triangle(c, c + np.stack([0.0 * np.cos(ang) - up * np.sin(ang), 0.0 * np.sin(ang) + up * np.cos(ang)], 1))
    mid = np.array([w_out * 0.5, h_out * 0.5])
    crop = triangle(np.broadcast_to(mid, (B, 2)), np.broadcast_to(mid + np.array([0.0, w_out * -0.5], np.float32), (B, 2)))
    frm, to = (crop, img) if inverse else (img, crop)
    a = np.concatenate([frm.astype(np.float64), np.ones((B, 3, 1))], 2)      # rows [x y 1]
    return np.linalg.solve(a, to.astype(np.float64)).transpose(0, 2, 1)      # batched exact 3-point solve


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=0) -> np.ndarray:
    """The reference's per-sample signature (lib/transforms.py:197) on top of `affine_matrices`; `shift` moves the
    box centre by shift * 200 * scale before the fit."""
    sc = np.broadcast_to(np.asarray(scale, np.float64).reshape(-1), (2,)) if np.ndim(scale) else np.array([scale, scale], np.float64)
    c = np.asarray(center, np.float64) + sc * 200.0 * np.asarray(shift, np.float64)
    return affine_matrices(c[None], sc[None], [rot], output_size, inverse=bool(inv))[0]


def affine_points(pts: np.ndarray, t: np.ndarray) -> np.ndarray:
    """(..., 2) points through one 2x3 matrix (lib/transforms.py:226-230, all points at once)."""
    pts = np.asarray(pts, np.float64)
    return pts[..., 0:1] * t[:, 0] + pts[..., 1:2] * t[:, 1] + t[:, 2]


def affine_transform(pt, t) -> np.ndarray:
    return affine_points(np.asarray(pt, np.float64)[:2], np.asarray(t, np.float64))


def _flip_perm(n: int, matched_parts) -> np.ndarray:
    perm = np.arange(n)
    pairs = np.asarray(matched_parts, np.int64).reshape(-1, 2)
    perm[pairs[:, 0]], perm[pairs[:, 1]] = pairs[:, 1], pairs[:, 0]
    return perm


def fliplr_joints(joints, joints_vis, width, matched_parts=FLIP_PAIRS):
    """Left-right flip of the annotations (lib/transforms.py:167-181): mirror x, exchange the left / right joints by
    one permutation gather, zero what is not visible.  Works on (J, 3) or (B, J, 3)."""
    perm = _flip_perm(joints.shape[-2], matched_parts)
    mirrored = np.array(joints, copy=True)
    mirrored[..., 0] = width - mirrored[..., 0] - 1
    vis = np.asarray(joints_vis)[..., perm, :]
    return mirrored[..., perm, :] * vis, vis


UPPER_BODY_IDS = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10)   # COCO: head, shoulders, arms (reference HRNet_Coco dataset)


def half_body_transform(joints, joints_vis, aspect_ratio: float, upper_body_ids=UPPER_BODY_IDS, pixel_std: float = 200.0,
                        rng=np.random):
    """data/JointsDataset.py:75-130: keep the visible upper- or lower-body joints (ONE ``rng.randn()`` draw, so a
    seeded generator walks the same sequence as the reference) and return (center, scale) of their bounding box,
    widened to the crop's aspect ratio and by 1.5; (None, None) when fewer than two joints are left."""
    joints = np.asarray(joints)
    visible = np.asarray(joints_vis)[:, 0] > 0
    is_upper = np.isin(np.arange(joints.shape[0]), upper_body_ids)
    n_upper = int((visible & is_upper).sum())
    keep = visible & (is_upper if (rng.randn() < 0.5 and n_upper > 2) else ~is_upper)
    if int(keep.sum()) < 2:
        return None, None
    box = joints[keep].astype(np.float32)
    extent = (box.max(0) - box.min(0))[:2]
    w, h = extent[0], extent[1]                     # float32 scalars: the widening below rounds like the reference's
    if w > aspect_ratio * h:
        h = w * 1.0 / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    return box.mean(axis=0)[:2], np.array([w * 1.0 / pixel_std, h * 1.0 / pixel_std], dtype=np.float32) * 1.5


def crop_batch(images: Sequence[torch.Tensor], centers, scales, rots, flips, image_size: Sequence[int],
               normalize: bool = True, device=None) -> Tuple[torch.Tensor, np.ndarray]:
    """JointsDataset.py:183-200 for a batch: images = uint8 HWC RGB tensors (any sizes); centers (B,2), scales (B,2)
    in the reference's units (scale * 200 px), rots in degrees, flips bool; image_size = (W, H) like the reference's
    config.  When flips[b] the image is mirrored and ``c[0] = width - c[0] - 1`` exactly as at :184-186.
    Returns the (B, 3, H, W) fp32 device batch and the (B, 2, 3) forward matrices (for the joints)."""
    dev = torch.device(device or "cuda")
    if dev.type != "cuda":
        raise RuntimeError("crop_batch (HIP) needs a GPU device; there is no CPU path")
    B = len(images)
    Wo, Ho = int(image_size[0]), int(image_size[1])
    centers = np.array(centers, np.float64).reshape(B, 2).copy()
    hw = np.zeros((B, 2), np.int32)
    offs = np.zeros(B, np.int64)
    flat: List[torch.Tensor] = []
    pos = 0
    for b, im in enumerate(images):
        assert im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3, "images must be uint8 HWC RGB"
        hw[b] = im.shape[:2]
        offs[b] = pos
        pos += im.numel()
        flat.append(im.reshape(-1))
    flipped = np.array([bool(f) for f in flips])
    centers[flipped, 0] = hw[flipped, 1] - centers[flipped, 0] - 1            # JointsDataset.py:184-186
    trans = affine_matrices(centers, np.asarray(scales, np.float64).reshape(B, -1), np.asarray(rots, np.float64), (Wo, Ho))
    full = np.concatenate([trans, np.broadcast_to(np.array([0.0, 0.0, 1.0]), (B, 1, 3))], 1)
    minv = np.linalg.inv(full)[:, :2].reshape(B, 6).astype(np.float32)       # crop pixel -> source coordinates
    src = torch.cat(flat).to(dev)
    t_off, t_hw = torch.from_numpy(offs).to(dev), torch.from_numpy(hw).to(dev)
    t_m = torch.from_numpy(minv).to(dev)
    t_f = torch.tensor([int(bool(f)) for f in flips], dtype=torch.int32, device=dev)
    mean = torch.tensor(IMAGENET_MEAN if normalize else (0.0, 0.0, 0.0), device=dev)
    std = torch.tensor(IMAGENET_STD if normalize else (1.0, 1.0, 1.0), device=dev)
    with torch.cuda.device(dev):
        out = torch.ops.stlpose.affine_crop(src, t_off, t_hw, t_m, t_f, Ho, Wo, mean, std)   # custom op -> stl_affine_crop
    return out, trans


def transform_joints(joints: np.ndarray, joints_vis: np.ndarray, trans: np.ndarray, flip: bool, width: int) -> Tuple[np.ndarray, np.ndarray]:
    """JointsDataset.py:185,195-197 for one person: optional left-right flip of the annotations, then the crop's
    affine map on every visible joint."""
    joints, joints_vis = joints.copy(), joints_vis.copy()
    if flip:
        joints, joints_vis = fliplr_joints(joints, joints_vis, width)
    seen = joints_vis[:, 0] > 0.0
    joints[seen, 0:2] = affine_points(joints[seen, 0:2], np.asarray(trans, np.float64))
    return joints, joints_vis
