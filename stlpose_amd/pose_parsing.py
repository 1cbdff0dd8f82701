"""Heatmap decode on device (mirror of reference ``src/lib/pose_parsing.py`` / ``lib/metrics.py``)."""
from __future__ import annotations

import numpy as np
import torch

from . import capi, ops  # noqa: F401  (ops registers the stlpose:: custom ops)


def _dev(x, device=None) -> torch.Tensor:
    if torch.is_tensor(x):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x))
    if not t.is_cuda:
        t = t.to(device or "cuda")
    return t.contiguous().float()


def max_preds_device(hm: torch.Tensor):
    """-> (idx int32 (B,J), maxvals f32 (B,J,1), preds f32 (B,J,2)) as device tensors."""
    return torch.ops.stlpose.heatmap_argmax(hm)   # custom op -> stl_heatmap_argmax (ops.py)


def get_max_preds_hrnet(scaled_heats, thr=0.1):
    """reference lib/pose_parsing.py:16-55 (numpy in, numpy out; empty batch -> ([], []))."""
    if scaled_heats.shape[0] == 0:
        return [], []
    hm = _dev(scaled_heats)
    _, mx, preds = max_preds_device(hm)
    return preds.cpu().numpy(), mx.cpu().numpy()


def get_final_preds_hrnet(heatmaps, center, scale):
    """reference lib/pose_parsing.py:58-92: argmax, +-0.25 px refinement, inverse crop affine
    (rot = 0).  Returns (preds, maxvals, coords) like the reference."""
    hm = _dev(heatmaps)
    b, j, h, w = hm.shape
    c = _dev(np.asarray(center, dtype=np.float32), hm.device)
    s = _dev(np.asarray(scale, dtype=np.float32), hm.device)
    preds, mx = torch.ops.stlpose.final_preds(hm, c, s)   # custom op -> stl_final_preds
    p = preds.cpu().numpy()
    # coords in heatmap space = forward crop transform of preds (kept for API parity)
    sc = (w / (np.asarray(scale, dtype=np.float64)[:, 0] * 200.0))[:, None]
    ctr = np.asarray(center, dtype=np.float64)
    coords = np.stack([(p[..., 0] - ctr[:, None, 0]) * sc + 0.5 * w, (p[..., 1] - ctr[:, None, 1]) * sc + 0.5 * h], -1)
    return p, mx.cpu().numpy(), coords.astype(np.float32)


def accuracy(output, target, hm_type="gaussian", thr=0.5):
    """reference lib/metrics.py:321-364 (PCK on heatmap argmaxes; the corrupted line :355-356 read
    as ``acc[i + 1] = dist_acc(dists[idx[i]])``).  Argmax runs on device; the 2*B*17 distances are
    host arithmetic."""
    o, t = _dev(output), _dev(target)
    _, _, pred = max_preds_device(o)
    _, _, tgt = max_preds_device(t)
    pred, tgt = pred.cpu().numpy(), tgt.cpu().numpy()
    b, j = pred.shape[:2]
    h, w = o.shape[2:]
    norm = np.ones((b, 2)) * np.array([h, w]) / 10
    ok = (tgt[..., 0] > 1) & (tgt[..., 1] > 1)
    d = np.linalg.norm(pred / norm[:, None] - tgt / norm[:, None], axis=-1)
    acc = np.zeros(j + 1)
    tot, cnt = 0.0, 0
    for c in range(j):
        n = ok[:, c].sum()
        a = (d[:, c][ok[:, c]] < thr).sum() / n if n > 0 else -1
        acc[c + 1] = a
        if a >= 0:
            tot, cnt = tot + a, cnt + 1
    avg = tot / cnt if cnt else 0
    if cnt:
        acc[0] = avg
    return acc, avg, cnt, pred
