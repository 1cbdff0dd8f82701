"""Heatmap targets on the GPU: the per-sample numpy loop of the reference's
``data/JointsDataset.py:230-286`` (``generate_target``) as one HIP launch per batch, so that a
loader only has to ship joint coordinates (a few hundred bytes per person) instead of
17 x 96 x 72 floats."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from . import capi, ops  # noqa: F401  (ops registers the stlpose:: custom ops)


def generate_targets(joints: torch.Tensor, joints_vis: torch.Tensor, heatmap_size: Sequence[int], image_size: Sequence[int],
                     sigma: float = 2.0, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """joints (B, J, 2|3) in image pixels, joints_vis (B, J) or (B, J, 3) (first column used, like the
    reference); heatmap_size / image_size are (W, H) as in the reference's config.
    Returns target (B, J, H, W) f32 and target_weight (B, J, 1) f32 on the device."""
    dev = torch.device(device or ("cuda" if not joints.is_cuda else joints.device))
    if dev.type != "cuda":
        raise RuntimeError("generate_targets (HIP) needs a GPU device; there is no CPU path")
    j = joints.to(dev, torch.float32)[..., :2].contiguous()
    v = joints_vis.to(dev, torch.float32)
    v = (v[..., 0] if v.dim() == 3 else v).contiguous()
    b, nj = j.shape[:2]
    wh, hh = int(heatmap_size[0]), int(heatmap_size[1])
    with torch.cuda.device(dev):
        target, tw = torch.ops.stlpose.gaussian_targets(j, v, hh, wh, float(image_size[0]) / wh, float(image_size[1]) / hh, float(sigma))
    return target, tw.view(b, nj, 1)
