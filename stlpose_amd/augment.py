"""Crop / augmentation half of the data pipeline on the GPU (reference
``data/JointsDataset.py:157-200`` + ``lib/transforms.py:167-250``).

The reference warps every sample on the host with ``cv2.warpAffine`` inside a ``num_workers=0`` loader
(``CONFIG.py:18``) -- two orders of magnitude below what the train step consumes.  Here the loader ships the
decoded uint8 image and a few scalars per person (center, scale, rotation, flip); the 2x3 matrices are the
reference's own arithmetic (``get_affine_transform``: a 3-point correspondence solved exactly, no cv2), and one
HIP launch per batch does the bilinear warp + ToTensor + Normalize into the fp32 NCHW batch the network reads.
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


def get_dir(src_point, rot_rad):
    """lib/transforms.py:240-248."""
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def get_3rd_point(a, b):
    """lib/transforms.py:234-237."""
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def _solve_affine(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """cv2.getAffineTransform for three point pairs: the exact solution of the 6x6 system (float64 result)."""
    a = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(a, np.asarray(dst, np.float64)).T


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0) -> np.ndarray:
    """lib/transforms.py:197-231, same point construction (float32 points like the reference's np.float32 casts)."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale])
    scale_tmp = np.asarray(scale, np.float64) * 200.0
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    return _solve_affine(dst, src) if inv else _solve_affine(src, dst)


def affine_transform(pt, t) -> np.ndarray:
    """lib/transforms.py:226-230."""
    return np.dot(t, np.array([pt[0], pt[1], 1.0]).T)[:2]


def fliplr_joints(joints, joints_vis, width, matched_parts=FLIP_PAIRS):
    """lib/transforms.py:167-181 (in place on copies are the caller's business, like the reference)."""
    joints[:, 0] = width - joints[:, 0] - 1
    for pair in matched_parts:
        joints[pair[0], :], joints[pair[1], :] = joints[pair[1], :], joints[pair[0], :].copy()
        joints_vis[pair[0], :], joints_vis[pair[1], :] = joints_vis[pair[1], :], joints_vis[pair[0], :].copy()
    return joints * joints_vis, joints_vis


UPPER_BODY_IDS = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10)   # COCO: head, shoulders, arms (reference HRNet_Coco dataset)


def half_body_transform(joints, joints_vis, aspect_ratio: float, upper_body_ids=UPPER_BODY_IDS, pixel_std: float = 200.0,
                        rng=np.random):
    """data/JointsDataset.py:75-130: pick the visible upper- or lower-body joints (one ``rng.randn()`` draw, like the
    reference) and return the (center, scale) of their bounding box, widened to the crop's aspect ratio and by 1.5;
    (None, None) when fewer than two joints are left."""
    upper, lower = [], []
    for jid in range(joints.shape[0]):
        if joints_vis[jid][0] > 0:
            (upper if jid in upper_body_ids else lower).append(joints[jid])
    sel = upper if (rng.randn() < 0.5 and len(upper) > 2) else lower
    if len(sel) < 2:
        return None, None
    sel = np.array(sel, dtype=np.float32)
    center = sel.mean(axis=0)[:2]
    lt, rb = np.amin(sel, axis=0), np.amax(sel, axis=0)
    w, h = rb[0] - lt[0], rb[1] - lt[1]
    if w > aspect_ratio * h:
        h = w * 1.0 / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w * 1.0 / pixel_std, h * 1.0 / pixel_std], dtype=np.float32) * 1.5
    return center, scale


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
    trans = np.zeros((B, 2, 3))
    minv = np.zeros((B, 6), np.float32)
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
        if flips[b]:
            centers[b, 0] = im.shape[1] - centers[b, 0] - 1
        trans[b] = get_affine_transform(centers[b], np.asarray(scales[b], np.float64), float(rots[b]), (Wo, Ho))
        full = np.concatenate([trans[b], [[0.0, 0.0, 1.0]]], 0)
        minv[b] = np.linalg.inv(full)[:2].reshape(-1)
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
    for i in range(joints.shape[0]):
        if joints_vis[i, 0] > 0.0:
            joints[i, 0:2] = affine_transform(joints[i, 0:2], trans)
    return joints, joints_vis
