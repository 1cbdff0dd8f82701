"""CPU: evaluation post-processing of stlpose_amd/evaluate.py (03_evaluate.py counterpart).
OKS-NMS is pinned through the oracle (fixture G7 = the reference's own lib/nms.py outputs); the
keypoint AP/AR restates pycocotools' COCOeval, which is not available -> hand-computed cases
(parity unpinned, see the module docstring)."""
import os

import numpy as np

from stlpose_amd.evaluate import COCO_SIGMAS, oks_ap, oks_nms, rescore_and_nms


def test_oks_nms_matches_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_decode.npz"))
    db = [dict(keypoints=g["nms_kpts"][i], score=g["nms_scores"][i], area=g["nms_areas"][i]) for i in range(6)]
    assert oks_nms(db, 0.9) == g["keep_09"].tolist()
    assert oks_nms(db, 0.5) == g["keep_05"].tolist()


def _person(cx, cy, img, ann_id, area=80.0 * 80.0, vis=2, crowd=0, nk=17):
    k = np.zeros((17, 3))
    k[:, 0] = cx + 20 * np.cos(np.arange(17))
    k[:, 1] = cy + 30 * np.sin(np.arange(17))
    k[:, 2] = vis
    return dict(id=ann_id, image_id=img, category_id=1, keypoints=k.reshape(-1).tolist(), num_keypoints=nk, area=area,
                bbox=[cx - 40, cy - 40, 80, 80], iscrowd=crowd)


def _det(gt, score, shift=0.0):
    k = np.array(gt["keypoints"]).reshape(17, 3).copy()
    k[:, 0] += shift
    k[:, 2] = 0.9
    return dict(image_id=gt["image_id"], category_id=1, keypoints=k.reshape(-1).tolist(), score=score)


def test_oks_ap_perfect_and_half():
    gts = [_person(100, 100, 1, 1), _person(300, 120, 1, 2), _person(150, 150, 2, 3, area=120.0 * 120.0)]
    perfect = [_det(g, 0.9 - 0.1 * i) for i, g in enumerate(gts)]
    s = oks_ap(gts, perfect)
    assert np.allclose(s[[0, 1, 2, 5, 6, 7]], 1.0)
    assert np.isclose(s[3], 1.0) and np.isclose(s[4], 1.0)       # two medium (80^2) and one large (120^2) person
    # one of two persons found, one detection far away: recall 0.5 -> precision 1 on recall points 0..0.5 (51 of 101)
    gts2 = [_person(100, 100, 1, 1), _person(300, 120, 1, 2)]
    dts2 = [_det(gts2[0], 0.9), _det(gts2[1], 0.8, shift=500.0)]
    s2 = oks_ap(gts2, dts2)
    assert np.isclose(s2[0], 51 / 101) and np.isclose(s2[5], 0.5)


def test_oks_ap_threshold_behaviour_and_ignored_gt():
    gt = _person(100, 100, 1, 1)
    # find a shift whose OKS lies between 0.5 and 0.75
    var = (COCO_SIGMAS * 2) ** 2
    shift = next(sh for sh in np.arange(1, 80, 0.5)
                 if 0.55 < np.mean(np.exp(-(sh ** 2) / var / (gt["area"] + np.spacing(1)) / 2)) < 0.7)
    s = oks_ap([gt], [_det(gt, 0.9, shift=shift)])
    assert np.isclose(s[1], 1.0) and np.isclose(s[2], 0.0) and 0.0 < s[0] < 1.0
    # a detection on a crowd / zero-keypoint person is neither a hit nor a false positive
    gts = [_person(100, 100, 1, 1), _person(300, 120, 1, 2, crowd=1), _person(500, 120, 1, 3, nk=0)]
    dts = [_det(gts[0], 0.9), _det(gts[1], 0.95), _det(gts[2], 0.97)]
    s3 = oks_ap(gts, dts)
    assert np.isclose(s3[0], 1.0) and np.isclose(s3[5], 1.0)
    # without ground truth in an area range the entry is -1 (no large person here)
    assert s3[4] == -1.0


def test_rescore_and_nms_scores_and_format():
    rng = np.random.default_rng(0)
    preds = np.zeros((3, 17, 3), np.float32)
    preds[:, :, :2] = rng.uniform(0, 200, (3, 17, 2))
    preds[0, :, 2] = 0.5
    preds[1, :, 2] = 0.1                       # nothing above the visibility threshold -> score 0
    preds[2, :8, 2], preds[2, 8:, 2] = 0.8, 0.1
    preds[2, :, :2] += 1000                    # far away from person 0: survives NMS
    boxes = np.zeros((3, 6))
    boxes[:, 4], boxes[:, 5] = 5000.0, [0.9, 0.8, 0.5]
    res = rescore_and_nms(preds, boxes, [7, 7, 7])
    assert all(r["image_id"] == 7 and r["category_id"] == 1 and len(r["keypoints"]) == 51 for r in res)
    scores = sorted(r["score"] for r in res)
    assert np.allclose(scores, sorted([0.5 * 0.9, 0.0, 0.8 * 0.5]), atol=1e-6)


# ---------------------------------------------------------------------------------------------- pinned by G12
def test_rescoring_and_nms_match_reference_generate_submission(golden_dir):
    """fixture G12: the reference's generate_submission_hrnet (lib/metrics.py:211-258) run on synthetic persons."""
    g = np.load(os.path.join(golden_dir, "g12_metrics.npz"))
    res = rescore_and_nms(g["sub_kpts"], g["sub_boxes"], g["sub_ids"].tolist())
    per_img = {}
    for r in res:
        per_img.setdefault(r["image_id"], []).append(r)
    assert [len(v) for v in per_img.values()] == g["sub_kept_n"].tolist()
    assert [r["image_id"] for r in res] == g["sub_kept_img"].tolist()
    np.testing.assert_allclose([r["score"] for r in res], g["sub_kept_scores"], rtol=1e-12)
    np.testing.assert_allclose(np.array([r["keypoints"] for r in res]).reshape(-1, 17, 3), g["sub_kept_kpts"], rtol=1e-12)


# ---------------------------------------------------------------------------------------------- sharded evaluation (gloo, world 2)
def _fake_loader(nbatch=5, per=3):
    rng = np.random.Generator(np.random.PCG64(5))
    out = []
    for b in range(nbatch):
        n = per if b != nbatch - 1 else per - 1          # ragged last batch
        meta = dict(center=rng.random((n, 2)) * 200, scale=0.5 + rng.random((n, 2)), score=rng.random(n),
                    image_id=np.array([100 + (b * per + i) // 2 for i in range(n)]))
        out.append((np.full((n, 1), b, np.float32), None, None, meta))
    return out


def _make_eval(pg):
    from stlpose_amd.evaluate import Evaluator

    class Fake(Evaluator):
        def __init__(self, pg):
            self.model, self.pg, self.shard_loader = None, pg, True
            self.rank, self.world = 0, 1
            if pg is not None:
                import torch.distributed as dist
                self.rank, self.world = dist.get_rank(pg), dist.get_world_size(pg)

        def _batch_outputs(self, imgs, target, target_weight, centers, scales):
            b = int(imgs[0, 0])
            rng = np.random.Generator(np.random.PCG64(1000 + b))
            n = len(centers)
            kp = rng.random((n, 17, 2)) * 150 + centers[:, None, :]
            if n > 1:
                kp[1] = kp[0] + 0.5             # a near-duplicate for the NMS
            return 0.1 * (b + 1), 0.05 * b, kp, rng.random((n, 17, 1))
    return Fake(pg)


def _eval_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _make_eval(dist.group.WORLD).evaluate_model(_fake_loader())
    q.put((rank, out["loss"], out["accuracy"], out["results"]))
    dist.destroy_process_group()


def test_sharded_evaluation_world2_equals_single_process():
    import socket
    import torch.multiprocessing as mp
    ref = _make_eval(None).evaluate_model(_fake_loader())
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in ps:
        p.join(60)
    assert len(ref["results"]) > 0
    for rank, loss, acc, results in got:
        assert abs(loss - ref["loss"]) < 1e-12 and abs(acc - ref["accuracy"]) < 1e-12
        assert results == ref["results"], f"rank {rank}: sharded results differ from the single-process run"
