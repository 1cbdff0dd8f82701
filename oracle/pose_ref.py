"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement (numpy / plain torch fp32) of the per-batch glue around the network:
masked MSE loss, flip-test forward, heatmap decode, PCK accuracy, gaussian targets.
Each function names the reference lines it follows.  Pinned by tests/golden (G5-G7, G9) except
``accuracy`` (reference metrics.py:355-356 is a corrupted line; read as
``acc[i + 1] = dist_acc(dists[idx[i]])``, consistent with :357-358) -- parity unpinned.
"""
from __future__ import annotations

import numpy as np
import torch

FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]  # CONSTANTS.py:65


def person_mse_loss(output: torch.Tensor, target: torch.Tensor, target_weight: torch.Tensor) -> torch.Tensor:
    """lib/loss.py:71-94.  Per joint: 0.5 * mean over (B, H*W) of ((o - t) * w)^2, then the
    mean over joints.  Written joint-by-joint like the reference (incl. its squeeze())."""
    b, j = output.shape[:2]
    o = output.reshape(b, j, -1)
    t = target.reshape(b, j, -1)
    total = output.new_zeros(())
    for k in range(j):
        ok = o[:, k].squeeze()
        tk = t[:, k].squeeze()
        w = target_weight[:, k, :]
        total = total + 0.5 * torch.mean((ok * w - tk * w) ** 2)
    return total / j


def flip_back(hm: np.ndarray, pairs=FLIP_PAIRS) -> np.ndarray:
    """lib/transforms.py:147-164: reverse W, swap left/right joint channels."""
    out = hm[:, :, :, ::-1].copy()
    for a, b in pairs:
        tmp = out[:, a].copy()
        out[:, a] = out[:, b]
        out[:, b] = tmp
    return out


def forward_pass(model, img: torch.Tensor, flip: bool = False) -> torch.Tensor:
    """lib/inference.py:11-32: plain forward; with flip: second forward on img.flip(3),
    flip_back, 1-px right shift (column 0 keeps its value), average."""
    out = model(img)
    if flip:
        of = model(img.flip(3))
        of = torch.from_numpy(flip_back(of.detach().cpu().numpy()))
        shifted = of.clone()
        shifted[:, :, :, 1:] = of[:, :, :, :-1]
        out = (out + shifted) * 0.5
    return out


def get_max_preds(hm: np.ndarray):
    """lib/pose_parsing.py:16-55: flat argmax (first max wins) / amax per (b, joint);
    x = idx % W, y = floor(idx / W); coordinates zeroed where max <= 0."""
    b, j, h, w = hm.shape
    flat = hm.reshape(b, j, -1)
    idx = np.argmax(flat, 2)
    mx = np.amax(flat, 2).reshape(b, j, 1)
    preds = np.zeros((b, j, 2), np.float32)
    preds[:, :, 0] = (idx % w).astype(np.float32)
    preds[:, :, 1] = np.floor(idx.astype(np.float32) / w)
    preds *= (mx > 0.0).astype(np.float32)
    return preds, mx


def refine_quarter_pixel(hm: np.ndarray, coords: np.ndarray) -> np.ndarray:
    """lib/pose_parsing.py:70-83: +-0.25 px towards the higher neighbour (strict interior)."""
    coords = coords.copy()
    _, _, h, w = hm.shape
    for n in range(coords.shape[0]):
        for p in range(coords.shape[1]):
            px = int(np.floor(coords[n, p, 0] + 0.5))
            py = int(np.floor(coords[n, p, 1] + 0.5))
            if 1 < px < w - 1 and 1 < py < h - 1:
                d = np.array([hm[n, p, py, px + 1] - hm[n, p, py, px - 1],
                              hm[n, p, py + 1, px] - hm[n, p, py - 1, px]])
                coords[n, p] += np.sign(d) * 0.25
    return coords


def affine_from_box(center, scale, out_wh, inv: bool) -> np.ndarray:
    """lib/transforms.py:197-233 with rot = 0, shift = 0.  The reference solves for the 2x3
    matrix with cv2.getAffineTransform on three point pairs; with rot = 0 the map is the
    axis-aligned scale+translate  dst = (src - center) * (dst_w / src_w) + dst_size / 2
    (both axes use dst_w / src_w because the 3rd point is the 90-degree rotation of the 2nd)."""
    src_w = float(scale[0]) * 200.0
    s = float(out_wh[0]) / src_w
    cx, cy = float(center[0]), float(center[1])
    fwd = np.array([[s, 0.0, out_wh[0] * 0.5 - s * cx], [0.0, s, out_wh[1] * 0.5 - s * cy]])
    if not inv:
        return fwd
    return np.array([[1 / s, 0.0, cx - out_wh[0] * 0.5 / s], [0.0, 1 / s, cy - out_wh[1] * 0.5 / s]])


def final_preds(hm: np.ndarray, center: np.ndarray, scale: np.ndarray):
    """lib/pose_parsing.py:58-92."""
    coords, mx = get_max_preds(hm)
    coords = refine_quarter_pixel(hm, coords)
    h, w = hm.shape[2:]
    preds = np.zeros_like(coords, dtype=np.float64)
    for i in range(coords.shape[0]):
        t = affine_from_box(center[i], scale[i], (w, h), inv=True)
        pts = np.concatenate([coords[i], np.ones((coords.shape[1], 1))], 1)
        preds[i] = pts @ t.T
    return preds, mx, coords


def pck_accuracy(output: np.ndarray, target: np.ndarray, thr: float = 0.5):
    """lib/metrics.py:268-364 (see header for the corrupted line)."""
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    b, j = pred.shape[:2]
    h, w = output.shape[2:]
    norm = np.ones((b, 2)) * np.array([h, w]) / 10
    dists = np.full((j, b), -1.0)
    for n in range(b):
        for c in range(j):
            if tgt[n, c, 0] > 1 and tgt[n, c, 1] > 1:
                dists[c, n] = np.linalg.norm(pred[n, c] / norm[n] - tgt[n, c] / norm[n])
    acc = np.zeros(j + 1)
    tot, cnt = 0.0, 0
    for c in range(j):
        valid = dists[c] != -1
        a = (dists[c][valid] < thr).sum() / valid.sum() if valid.sum() > 0 else -1
        acc[c + 1] = a
        if a >= 0:
            tot += a
            cnt += 1
    avg = tot / cnt if cnt else 0
    if cnt:
        acc[0] = avg
    return acc, avg, cnt, pred


def gaussian_targets(joints_xy: np.ndarray, vis: np.ndarray, hm_wh, img_wh, sigma: float):
    """data/JointsDataset.py:230-286 for one person: joints_xy (J,2) in image pixels,
    vis (J,) in {0,1} -> target (J, Hh, Wh) f32, target_weight (J,1) f32."""
    nj = joints_xy.shape[0]
    wh, hh = hm_wh
    target = np.zeros((nj, hh, wh), np.float32)
    tw = vis.astype(np.float32).reshape(nj, 1).copy()
    r = sigma * 3
    stride = np.array(img_wh, np.float64) / np.array(hm_wh, np.float64)
    size = int(2 * r + 1)
    ax = np.arange(0, size, 1, np.float32)
    g = np.exp(-((ax[None, :] - size // 2) ** 2 + (ax[:, None] - size // 2) ** 2) / (2 * sigma ** 2))
    for k in range(nj):
        mx = int(joints_xy[k, 0] / stride[0] + 0.5)
        my = int(joints_xy[k, 1] / stride[1] + 0.5)
        ul = [int(mx - r), int(my - r)]
        br = [int(mx + r + 1), int(my + r + 1)]
        if ul[0] >= wh or ul[1] >= hh or br[0] < 0 or br[1] < 0:
            tw[k] = 0
            continue
        gx = max(0, -ul[0]), min(br[0], wh) - ul[0]
        gy = max(0, -ul[1]), min(br[1], hh) - ul[1]
        ix = max(0, ul[0]), min(br[0], wh)
        iy = max(0, ul[1]), min(br[1], hh)
        if tw[k] > 0.5:
            target[k, iy[0]:iy[1], ix[0]:ix[1]] = g[gy[0]:gy[1], gx[0]:gx[1]]
    return target, tw


def oks_iou(g, d, a_g, a_d, sigmas=None, in_vis_thre=None):
    """lib/nms.py:48-74 (incl. its `list and list` visibility quirk: only vd is applied)."""
    if sigmas is None:
        sigmas = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62,
                           1.07, 1.07, .87, .87, .89, .89]) / 10.0
    var = (sigmas * 2) ** 2
    xg, yg = g[0::3], g[1::3]
    ious = np.zeros(d.shape[0])
    for n in range(d.shape[0]):
        dx = d[n, 0::3] - xg
        dy = d[n, 1::3] - yg
        e = (dx ** 2 + dy ** 2) / var / ((a_g + a_d[n]) / 2 + np.spacing(1)) / 2
        if in_vis_thre is not None:
            e = e[d[n, 2::3] > in_vis_thre]
        ious[n] = np.sum(np.exp(-e)) / e.shape[0] if e.shape[0] else 0.0
    return ious


def oks_nms(kpts: np.ndarray, scores: np.ndarray, areas: np.ndarray, thresh: float):
    """lib/nms.py:10-45 on arrays: kpts (n, 51), scores (n,), areas (n,) -> kept indices."""
    if len(scores) == 0:
        return []
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        ovr = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]])
        order = order[np.where(ovr <= thresh)[0] + 1]
    return keep


def warp_affine_bilinear(img: np.ndarray, trans: np.ndarray, out_wh, flip: bool = False) -> np.ndarray:
    """Float restatement of ``cv2.warpAffine(img, trans, (W, H), flags=cv2.INTER_LINEAR)`` as the reference calls
    it (data/JointsDataset.py:190-195): dst(x, y) = bilinear sample of src at trans^-1 (x, y, 1), pixel centres at
    integer coordinates, constant zero border (taps outside the image contribute 0).  cv2 itself is a third-party
    dependency absent from this image (opencv-python pinned in environment.yml); its 1/32-pixel fixed-point
    coordinate rounding is NOT restated -> PARITY UNPINNED for that rounding (differences <= 1/64 px of
    interpolation).  ``flip`` mirrors the source first (JointsDataset.py:184).  img uint8 HWC -> float HWC in 0..255."""
    if flip:
        img = img[:, ::-1, :]
    h, w = img.shape[:2]
    wo, ho = int(out_wh[0]), int(out_wh[1])
    minv = np.linalg.inv(np.concatenate([np.asarray(trans, np.float64), [[0.0, 0.0, 1.0]]], 0))[:2].astype(np.float32)
    ys, xs = np.mgrid[0:ho, 0:wo].astype(np.float32)
    sx = minv[0, 0] * xs + minv[0, 1] * ys + minv[0, 2]
    sy = minv[1, 0] * xs + minv[1, 1] * ys + minv[1, 2]
    x0, y0 = np.floor(sx), np.floor(sy)
    ax, ay = sx - x0, sy - y0
    out = np.zeros((ho, wo, 3), np.float32)
    src = img.astype(np.float32)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = (x0 + dx).astype(np.int64), (y0 + dy).astype(np.int64)
            ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
            wgt = (ax if dx else 1 - ax) * (ay if dy else 1 - ay)
            v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
            out += np.where(ok, wgt, 0.0)[..., None] * v
    return out
