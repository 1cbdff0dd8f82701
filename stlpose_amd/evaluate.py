"""Evaluation driver for the MI355X hot path: the counterpart of the reference's
``src/03_evaluate.py`` (``Evaluator.evaluate_model``, :114-216) with the post-processing it calls:

  * flip-test forward + quarter-pixel / affine decode on the device (``forward_pass(flip=True)``,
    ``get_final_preds_hrnet``);
  * box re-scoring and OKS-NMS of ``lib/metrics.py:188-262`` (``generate_submission_hrnet``) and
    ``lib/nms.py:10-74`` -- host numpy, a few hundred persons per batch;
  * keypoint AP/AR.  The reference delegates to ``pycocotools.cocoeval.COCOeval(..., "keypoints")``
    (``lib/metrics.py:154-187``), a third-party dependency (pycocotools 2.0.0, ``environment.yml:366``)
    that is neither vendored in the reference tree nor installed here.  ``oks_ap`` below restates its
    published algorithm (computeOks / evaluateImg / accumulate / summarize for iouType "keypoints":
    OKS thresholds .50:.05:.95, 101 recall points, maxDets 20, area ranges all / medium / large,
    crowd and zero-keypoint ground truth ignored) -- **parity unpinned**: there is no reference test
    or fixture for it; ``tests/test_evaluate_cpu.py`` checks it against hand-computed cases.
Dataset loading stays the caller's: loaders yield ``(imgs, target, target_weight, metadata)`` with
``metadata["center"|"scale"|"score"|"image_id"]`` like the reference's.
"""
from __future__ import annotations

import json
from collections import defaultdict
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

COCO_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


# ---------------------------------------------------------------------------------------------- NMS
def oks_iou(g: np.ndarray, d: np.ndarray, a_g: float, a_d: np.ndarray, sigmas=None, in_vis_thre=None) -> np.ndarray:
    """lib/nms.py:48-74: OKS of one pose g (51,) against poses d (n, 51); when a visibility
    threshold is given only the candidate's confidences are tested (the reference's
    ``list(vg > t) and list(vd > t)`` evaluates to the second list)."""
    var = ((COCO_SIGMAS if sigmas is None else np.asarray(sigmas)) * 2) ** 2
    xg, yg = g[0::3], g[1::3]
    out = np.zeros(d.shape[0])
    for n in range(d.shape[0]):
        e = ((d[n, 0::3] - xg) ** 2 + (d[n, 1::3] - yg) ** 2) / var / ((a_g + a_d[n]) / 2 + np.spacing(1)) / 2
        if in_vis_thre is not None:
            e = e[d[n, 2::3] > in_vis_thre]
        out[n] = np.sum(np.exp(-e)) / e.shape[0] if e.shape[0] else 0.0
    return out


def oks_nms(persons: Sequence[dict], thresh: float, sigmas=None, in_vis_thre=None) -> List[int]:
    """lib/nms.py:10-45: greedy suppression by OKS, highest score first; returns kept indices."""
    if len(persons) == 0:
        return []
    scores = np.array([p["score"] for p in persons])
    kpts = np.array([np.asarray(p["keypoints"]).flatten() for p in persons])
    areas = np.array([p["area"] for p in persons])
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        ov = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]], sigmas, in_vis_thre)
        order = order[np.where(ov <= thresh)[0] + 1]
    return keep


def rescore_and_nms(all_preds: np.ndarray, all_boxes: np.ndarray, image_ids: Sequence, in_vis_thr: float = 0.2,
                    oks_thr: float = 0.9) -> List[dict]:
    """lib/metrics.py:211-262: per person score = box score x mean confidence of the joints above
    ``in_vis_thr``; OKS-NMS per image (all persons kept if NMS keeps none); COCO result dicts."""
    per_img: Dict = defaultdict(list)
    for kp, bx, im in zip(all_preds, all_boxes, image_ids):
        conf = kp[:, 2]
        good = conf > in_vis_thr
        k = float(conf[good].mean()) if good.any() else 0.0
        per_img[im].append(dict(keypoints=np.asarray(kp, np.float64), center=bx[0:2], scale=bx[2:4], area=float(bx[4]),
                                score=k * float(bx[5]), image=im))
    results = []
    for im, people in per_img.items():
        keep = oks_nms(people, oks_thr)
        kept = people if len(keep) == 0 else [people[i] for i in keep]
        for p in kept:
            results.append(dict(image_id=int(im), category_id=1, keypoints=[float(v) for v in p["keypoints"].reshape(-1)],
                                score=float(p["score"])))
    return results


# ---------------------------------------------------------------------------------------------- AP
def _oks_matrix(dts: List[dict], gts: List[dict], sigmas: np.ndarray) -> np.ndarray:
    """COCOeval.computeOks: rows = detections, columns = ground truth."""
    if not dts or not gts:
        return np.zeros((len(dts), len(gts)))
    var = (sigmas * 2) ** 2
    k = len(sigmas)
    out = np.zeros((len(dts), len(gts)))
    for j, gt in enumerate(gts):
        g = np.asarray(gt["keypoints"], np.float64)
        xg, yg, vg = g[0::3], g[1::3], g[2::3]
        k1 = int(np.count_nonzero(vg > 0))
        bb = gt["bbox"]
        x0, x1 = bb[0] - bb[2], bb[0] + bb[2] * 2
        y0, y1 = bb[1] - bb[3], bb[1] + bb[3] * 2
        for i, dt in enumerate(dts):
            d = np.asarray(dt["keypoints"], np.float64)
            xd, yd = d[0::3], d[1::3]
            if k1 > 0:
                dx, dy = xd - xg, yd - yg
            else:  # no annotated keypoint: distance to the doubled box
                z = np.zeros(k)
                dx = np.max((z, x0 - xd), axis=0) + np.max((z, xd - x1), axis=0)
                dy = np.max((z, y0 - yd), axis=0) + np.max((z, yd - y1), axis=0)
            e = (dx ** 2 + dy ** 2) / var / (gt["area"] + np.spacing(1)) / 2
            if k1 > 0:
                e = e[vg > 0]
            out[i, j] = np.sum(np.exp(-e)) / e.shape[0]
    return out


def oks_ap(gt_annotations: Sequence[dict], results: Sequence[dict], img_ids: Optional[Sequence[int]] = None,
           sigmas: Optional[np.ndarray] = None, max_dets: int = 20) -> np.ndarray:
    """Keypoint AP/AR with COCOeval semantics.  gt_annotations: COCO person annotations (image_id,
    keypoints[51], num_keypoints, area, bbox, iscrowd); results: rescore_and_nms() output.
    Returns the 10 numbers of COCOeval.stats for keypoints:
    AP, AP50, AP75, AP(M), AP(L), AR, AR50, AR75, AR(M), AR(L)."""
    sigmas = COCO_SIGMAS if sigmas is None else np.asarray(sigmas)
    thrs = np.linspace(.5, .95, 10)
    recs = np.linspace(.0, 1.0, 101)
    ranges = [(0, 1e10), (32 ** 2, 96 ** 2), (96 ** 2, 1e10)]
    gt_by, dt_by = defaultdict(list), defaultdict(list)
    for a in gt_annotations:
        gt_by[a["image_id"]].append(a)
    for r in results:
        d = dict(r)
        if "area" not in d:  # COCO.loadRes for keypoints: area of the keypoints' bounding box
            kx, ky = np.asarray(d["keypoints"][0::3]), np.asarray(d["keypoints"][1::3])
            d["area"] = float((kx.max() - kx.min()) * (ky.max() - ky.min()))
        dt_by[d["image_id"]].append(d)
    ids = sorted(set(gt_by) | set(dt_by)) if img_ids is None else sorted(set(img_ids))
    T, R, A = len(thrs), len(recs), len(ranges)
    precision, recall = -np.ones((T, R, A)), -np.ones((T, A))
    for ai, (lo, hi) in enumerate(ranges):
        dtm_all, ig_all, sc_all, npig = [], [], [], 0
        for im in ids:
            gts, dts = gt_by.get(im, []), dt_by.get(im, [])
            if not gts and not dts:
                continue
            gig = np.array([bool(g.get("iscrowd", 0)) or g.get("num_keypoints", 1) == 0 or g["area"] < lo or g["area"] > hi for g in gts], bool)
            gorder = np.argsort(gig, kind="mergesort")
            gts = [gts[i] for i in gorder]
            gig = gig[gorder] if len(gorder) else gig
            dorder = np.argsort([-d["score"] for d in dts], kind="mergesort")[:max_dets]
            dts = [dts[i] for i in dorder]
            oks = _oks_matrix(dts, gts, sigmas)
            crowd = [bool(g.get("iscrowd", 0)) for g in gts]
            dtm = -np.ones((T, len(dts)), int)
            dig = np.zeros((T, len(dts)), bool)
            for ti, t in enumerate(thrs):
                gtm = -np.ones(len(gts), int)
                for di in range(len(dts)):
                    best, m = min(t, 1 - 1e-10), -1
                    for gi in range(len(gts)):
                        if gtm[gi] >= 0 and not crowd[gi]:
                            continue
                        if m > -1 and not gig[m] and gig[gi]:   # matched a regular gt: stop at the ignored ones
                            break
                        if oks[di, gi] < best:
                            continue
                        best, m = oks[di, gi], gi
                    if m == -1:
                        continue
                    dig[ti, di] = gig[m]
                    dtm[ti, di] = m
                    gtm[m] = di
            out_of_range = np.array([d["area"] < lo or d["area"] > hi for d in dts], bool)
            dig = dig | ((dtm < 0) & out_of_range[None, :])
            dtm_all.append(dtm), ig_all.append(dig), sc_all.append(np.array([d["score"] for d in dts]))
            npig += int(np.count_nonzero(~gig))
        if npig == 0:
            continue
        sc = np.concatenate(sc_all) if sc_all else np.zeros(0)
        order = np.argsort(-sc, kind="mergesort")
        dtm = np.concatenate(dtm_all, axis=1)[:, order] if dtm_all else np.zeros((T, 0), int)
        dig = np.concatenate(ig_all, axis=1)[:, order] if ig_all else np.zeros((T, 0), bool)
        tps = np.cumsum((dtm >= 0) & ~dig, axis=1).astype(float)
        fps = np.cumsum((dtm < 0) & ~dig, axis=1).astype(float)
        for ti in range(T):
            tp, fp = tps[ti], fps[ti]
            rc = tp / npig
            pr = tp / (fp + tp + np.spacing(1))
            recall[ti, ai] = rc[-1] if len(rc) else 0
            pr = pr.tolist()
            for i in range(len(pr) - 1, 0, -1):    # precision envelope
                if pr[i] > pr[i - 1]:
                    pr[i - 1] = pr[i]
            q = np.zeros(R)
            inds = np.searchsorted(rc, recs, side="left")
            for ri, pi in enumerate(inds):
                if pi < len(pr):
                    q[ri] = pr[pi]
            precision[ti, :, ai] = q

    def _mean(x):
        x = x[x > -1]
        return float(x.mean()) if x.size else -1.0
    t50, t75 = 0, 5
    return np.array([_mean(precision[:, :, 0]), _mean(precision[t50, :, 0]), _mean(precision[t75, :, 0]),
                     _mean(precision[:, :, 1]), _mean(precision[:, :, 2]),
                     _mean(recall[:, 0]), _mean(recall[t50:t50 + 1, 0]), _mean(recall[t75:t75 + 1, 0]),
                     _mean(recall[:, 1]), _mean(recall[:, 2])])


# ---------------------------------------------------------------------------------------------- driver
def gather_eval_shards(process_group, seq: np.ndarray, preds: np.ndarray, boxes: np.ndarray, image_ids: Sequence[int],
                       sums: Sequence[float]):
    """Data-parallel evaluation (replaces the DataParallel gather of 03_evaluate.py:100,176-188):
    every rank has run the network on its shard of the validation batches; all-gather the few KB per batch the
    host-side scoring needs -- ``preds (n,17,3)``, ``boxes (n,6)``, ``image_ids``, plus the loss / PCK sums --
    and restore the loader's order (``seq`` = (global batch index, position in batch) per person) so that
    OKS-NMS and AP see exactly what a single process would have seen."""
    import torch.distributed as dist
    payload = (np.asarray(seq, np.int64).reshape(-1, 2), preds, boxes, [int(v) for v in image_ids], [float(v) for v in sums])
    world = dist.get_world_size(process_group)
    parts = [None] * world
    dist.all_gather_object(parts, payload, group=process_group)
    seq = np.concatenate([p[0] for p in parts])
    preds = np.concatenate([p[1] for p in parts])
    boxes = np.concatenate([p[2] for p in parts])
    ids = np.concatenate([np.asarray(p[3], np.int64) for p in parts])
    order = np.lexsort((seq[:, 1], seq[:, 0]))
    sums = np.sum(np.asarray([p[4] for p in parts], np.float64), axis=0)
    return preds[order], boxes[order], [int(v) for v in ids[order]], [float(v) for v in sums]


class Evaluator:
    """03_evaluate.py:114-216 on the HIP path.  ``evaluate_model`` returns a dict with the mean
    loss, mean PCK, the COCO result list and (when ground truth is given) the 10 AP/AR numbers.

    With ``process_group`` (one process per GPU) rank r evaluates batches r, r + world, ... of the loader
    (``shard_loader=True``; pass False when the loader is already sharded by a DistributedSampler and give
    each batch's global index in ``metadata['batch_index']``), the per-person results are all-gathered
    (``gather_eval_shards``) and every rank scores the full set; only rank 0 writes ``preds_file``."""

    def __init__(self, model, device="cuda", model_name: str = "HRNet", flip: bool = True, process_group=None,
                 shard_loader: bool = True):
        from .loss import PersonMSELoss
        self.model, self.device, self.model_name, self.flip = model, torch.device(device), model_name, flip
        self.loss_function = PersonMSELoss()
        self.pg, self.shard_loader = process_group, shard_loader
        self.rank, self.world = 0, 1
        if process_group is not None:
            import torch.distributed as dist
            self.rank, self.world = dist.get_rank(process_group), dist.get_world_size(process_group)

    def _my_batches(self, loader):
        """(global batch index, batch) pairs this rank evaluates.  One process: all of them.  shard_loader=True: batches
        rank, rank + world, ...; an INDEXABLE loader (list of batches, a map-style dataset of batches: __len__ and
        __getitem__) is indexed directly, so a rank never loads or decodes the batches of the others; a plain iterable
        can only be skipped through."""
        if self.world == 1 or not self.shard_loader:
            yield from enumerate(loader)
        elif hasattr(loader, "__getitem__") and hasattr(loader, "__len__"):
            for bi in range(self.rank, len(loader), self.world):
                yield bi, loader[bi]
        else:
            for bi, batch in enumerate(loader):
                if bi % self.world == self.rank:
                    yield bi, batch

    def _batch_outputs(self, imgs, target, target_weight, centers, scales):
        """network + loss + PCK + decode for one batch (the only device work of the evaluation):
        returns (loss, avg PCK, keypoints (n,17,2) in image coordinates, max_vals (n,17,1))."""
        from .inference import forward_pass
        from .pose_parsing import accuracy, get_final_preds_hrnet
        imgs = imgs.float().to(self.device)
        target = target.float().to(self.device)
        output = forward_pass(model=self.model, img=imgs, model_name=self.model_name, device=self.device, flip=self.flip)
        loss = float(self.loss_function(output, target, target_weight.float().to(self.device)).item())
        keypoints, max_vals, _ = get_final_preds_hrnet(heatmaps=output, center=centers, scale=scales)
        if not np.isfinite(np.asarray(max_vals)).all():
            # eval mode has no batch statistics to carry the range guard of the training step (Engine.check_forward_range): a
            # forward tensor beyond its 16-bit range ends as a non-finite heat map -- say so instead of scoring NaNs
            raise FloatingPointError(
                "stlpose_amd.Evaluator: non-finite heat maps.  With compute_dtype='mixed' (f16 forward tensors, |y| <= 65504) a "
                "badly scaled checkpoint overflows; evaluate it with compute_dtype='bf16' or 'fp32'.")
        return loss, float(accuracy(output, target)[1]), keypoints, max_vals

    @torch.no_grad()
    def evaluate_model(self, loader: Iterable, gt_annotations: Optional[Sequence[dict]] = None, labels_file: Optional[str] = None,
                       preds_file: Optional[str] = None) -> dict:
        if self.model is not None:
            self.model.eval()
        loss_sum = acc_sum = 0.0
        nb = 0
        all_preds, all_boxes, image_ids, seq = [], [], [], []
        for bi, (imgs, target, target_weight, metadata) in self._my_batches(loader):
            gbi = bi if (self.world == 1 or self.shard_loader) else int(metadata.get("batch_index", bi * self.world + self.rank))
            centers, scales = np.asarray(metadata["center"], np.float64), np.asarray(metadata["scale"], np.float64)
            score = np.asarray(metadata["score"], np.float64)
            loss, acc, keypoints, max_vals = self._batch_outputs(imgs, target, target_weight, centers, scales)
            loss_sum, acc_sum, nb = loss_sum + loss, acc_sum + acc, nb + 1
            n = keypoints.shape[0]
            preds = np.zeros((n, keypoints.shape[1], 3), np.float32)
            preds[:, :, :2], preds[:, :, 2:3] = keypoints[:, :, :2], max_vals
            boxes = np.zeros((n, 6))
            boxes[:, 0:2], boxes[:, 2:4] = centers[:, 0:2], scales[:, 0:2]
            boxes[:, 4], boxes[:, 5] = np.prod(scales * 200, 1), score
            all_preds.append(preds), all_boxes.append(boxes)
            image_ids += [int(v) for v in np.asarray(metadata["image_id"]).tolist()]
            seq += [(gbi, i) for i in range(n)]
        preds = np.concatenate(all_preds) if all_preds else np.zeros((0, 17, 3), np.float32)
        boxes = np.concatenate(all_boxes) if all_boxes else np.zeros((0, 6))
        sums = [loss_sum, acc_sum, float(nb)]
        if self.world > 1:
            preds, boxes, image_ids, sums = gather_eval_shards(self.pg, np.asarray(seq, np.int64).reshape(-1, 2), preds, boxes,
                                                               image_ids, sums)
        results = rescore_and_nms(preds, boxes, image_ids) if len(preds) else []
        if preds_file and self.rank == 0:
            with open(preds_file, "w") as f:
                json.dump(results, f)
        if gt_annotations is None and labels_file is not None:
            with open(labels_file) as f:
                gt_annotations = json.load(f)["annotations"]
        stats = oks_ap(gt_annotations, results, img_ids=sorted(set(image_ids))) if gt_annotations is not None else None
        nbt = sums[2]
        return dict(loss=sums[0] / nbt if nbt else float("nan"), accuracy=sums[1] / nbt if nbt else 0.0,
                    results=results, stats=stats)
