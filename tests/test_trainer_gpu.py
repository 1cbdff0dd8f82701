"""GPU: the 02_train.py counterpart (stlpose_amd/trainer.py) -- epoch structure, training_logs.json,
checkpoint layout of the reference (module. prefix, torch.optim-shaped optimizer state), resume."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from stlpose_amd.trainer import Trainer  # noqa: E402


def _loader(n, B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    hh, ww = H // 4, W // 4
    out = []
    for _ in range(n):
        img = torch.randn(B, 3, H, W, generator=g)
        cx = torch.randint(2, ww - 2, (B, 17, 1, 1), generator=g).float()
        cy = torch.randint(2, hh - 2, (B, 17, 1, 1), generator=g).float()
        ys, xs = torch.arange(hh).view(1, 1, hh, 1).float(), torch.arange(ww).view(1, 1, 1, ww).float()
        tgt = torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / 8.0)
        tw = (torch.rand(B, 17, 1, generator=g) < 0.8).float()
        out.append((img, tgt, tw, {"perceptual_loss": torch.zeros(B)}))
    return out


def _exp_data(epochs, H, W, optimizer="adam", scheduler="step"):
    return {"training": {"num_epochs": epochs, "save_frequency": 1, "learning_rate": 1e-3, "learning_rate_factor": 0.5,
                         "patience": 1, "momentum": 0.9, "optimizer": optimizer, "nesterov": False, "scheduler": scheduler},
            "model": {"model_name": "HRNet"}, "dataset": {"image_size": [H, W], "dataset_name": "coco"}}


def test_trainer_epochs_logs_checkpoints_and_resume(tmp_path):
    B, H, W = 4, 96, 64
    train, valid = _loader(6, B, H, W, 1), _loader(10, B, H, W, 2)
    exp_a = str(tmp_path / "a")
    os.makedirs(exp_a)
    torch.manual_seed(3)
    ta = Trainer(exp_a, _exp_data(2, H, W), train, valid, B, arch="tiny", compute_dtype="fp32")
    ta.setup_model()
    ta.training_loop()
    logs = json.load(open(os.path.join(exp_a, "training_logs.json")))
    assert set(logs) == {"last_modified", "iterations", "loss", "accuracy"} and logs["iterations"] == 12
    assert len(logs["loss"]["training"]) == 2 and len(logs["accuracy"]["validation"]) == 2
    assert logs["loss"]["training"][1] < logs["loss"]["training"][0]
    assert ta.lr == pytest.approx(1e-3 * 0.5 ** 2)          # StepLR, step_size 1, two epochs
    # checkpoint layout of the reference (model_setup.py:200-205; DataParallel prefix 02_train.py:166)
    names = sorted(os.listdir(os.path.join(exp_a, "models")))
    assert names == ["checkpoint_epoch_0.pth", "checkpoint_epoch_1.pth", "checkpoint_epoch_final.pth"]
    ck = torch.load(os.path.join(exp_a, "models", "checkpoint_epoch_0.pth"), weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict"}
    assert all(k.startswith("module.") for k in ck["model_state_dict"])
    # the optimizer state is what torch.optim.Adam over the same parameter list would save
    params = [torch.nn.Parameter(torch.zeros_like(p)) for p in ta.model.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3)
    opt.load_state_dict(ck["optimizer_state_dict"])
    assert opt.state[params[0]]["exp_avg"].shape == params[0].shape
    assert float(opt.state[params[-1]]["step"]) == 6.0
    # resume from epoch 0's checkpoint: epoch 1 is replayed and ends where the straight run ended
    exp_b = str(tmp_path / "b")
    os.makedirs(exp_b)
    json.dump(json.load(open(os.path.join(exp_a, "training_logs.json"))), open(os.path.join(exp_b, "training_logs.json"), "w"))
    tb = Trainer(exp_b, _exp_data(2, H, W), train, valid, B, arch="tiny", compute_dtype="fp32",
                 checkpoint=os.path.join(exp_a, "models", "checkpoint_epoch_0.pth"), resume_training=True)
    tb.setup_model()
    assert tb.cur_epoch == 0 and int(tb.ts.step_count.item()) == 6
    tb.cur_epoch = 1                                        # the checkpoint was written AFTER epoch 0 finished
    tb.training_loop()
    wa = torch.cat([p.detach().flatten() for p in ta.model.parameters()])
    wb = torch.cat([p.detach().flatten() for p in tb.model.parameters()])
    assert float((wa - wb).abs().max()) < 2e-4 * float(wa.abs().max())


def test_trainer_plateau_scheduler_and_sgd(tmp_path):
    B, H, W = 2, 96, 64
    exp = str(tmp_path / "c")
    os.makedirs(exp)
    t = Trainer(exp, _exp_data(2, H, W, optimizer="sgd", scheduler="plateau"), _loader(3, B, H, W, 4), _loader(5, B, H, W, 5), B,
                arch="tiny", compute_dtype="bf16")
    t.setup_model()
    t.training_loop()
    assert np.isfinite(t.train_loss) and np.isfinite(t.valid_loss)
    ck = torch.load(os.path.join(exp, "models", "checkpoint_epoch_final.pth"), weights_only=False)
    assert "momentum_buffer" in ck["optimizer_state_dict"]["state"][0]
    assert ck["scheduler_state_dict"]["mode"] == "max"      # the reference steps ReduceLROnPlateau(mode="max") on the loss


def test_evaluator_end_to_end_on_synthetic_loader():
    """03_evaluate.py counterpart: flip-test forward, device decode, rescoring + OKS-NMS, AP.  With the
    ground truth built from the model's own detections the AP must be 1."""
    from stlpose_amd import PoseHighResolutionNet
    from stlpose_amd.evaluate import Evaluator, oks_ap
    B, H, W = 4, 96, 64
    torch.manual_seed(6)
    model = PoseHighResolutionNet("tiny", "fp32").cuda()
    batches = _loader(3, B, H, W, 8)
    loader = []
    for i, (img, tgt, tw, _) in enumerate(batches):
        meta = {"center": torch.tensor([[W / 2.0, H / 2.0]] * B), "scale": torch.tensor([[W / 200.0, H / 200.0]] * B),
                "score": torch.ones(B), "image_id": torch.arange(i * B, (i + 1) * B), "image": [f"im{j}.jpg" for j in range(B)]}
        loader.append((img, tgt, tw, meta))
    ev = Evaluator(model)
    out = ev.evaluate_model(loader)
    assert np.isfinite(out["loss"]) and 0.0 <= out["accuracy"] <= 1.0 and out["stats"] is None
    assert len(out["results"]) == 3 * B and all(len(r["keypoints"]) == 51 for r in out["results"])
    gts = []
    for n, r in enumerate(out["results"]):
        k = np.array(r["keypoints"]).reshape(17, 3)
        k[:, 2] = 2
        gts.append(dict(id=n + 1, image_id=r["image_id"], category_id=1, keypoints=k.reshape(-1).tolist(), num_keypoints=17,
                        area=float(W * H), bbox=[0, 0, W, H], iscrowd=0))
    stats = oks_ap(gts, out["results"])
    assert np.isclose(stats[0], 1.0) and np.isclose(stats[5], 1.0)
    out2 = ev.evaluate_model(loader, gt_annotations=gts)
    assert np.allclose(out2["stats"][[0, 5]], 1.0)


def test_checkpoints_hold_the_trained_weights_and_sgd_resume_keeps_momentum(tmp_path):
    """(high) a validation epoch before training must not re-pack the model away from the optimiser's
    flat buffers -- the saved weights are the TRAINED ones; (low) resuming SGD continues from the momentum buffer."""
    B, H, W = 2, 96, 64
    exp = str(tmp_path / "d")
    os.makedirs(exp)
    torch.manual_seed(12)
    t = Trainer(exp, _exp_data(1, H, W, optimizer="sgd", scheduler="step"), _loader(4, B, H, W, 6), _loader(5, B, H, W, 7), B,
                arch="tiny", compute_dtype="fp32")   # default device "cuda" (no index), like the reference's scripts
    t.setup_model()
    init = t.ts.store.master.clone()
    t.training_loop()
    ck = torch.load(os.path.join(exp, "models", "checkpoint_epoch_final.pth"), weights_only=False)
    flat = torch.cat([ck["model_state_dict"]["module." + k].reshape(-1) for k, _ in t.ts.store.reg.params])
    assert torch.equal(flat, t.ts.store.master.cpu()), "checkpoint does not hold the flat buffer the optimiser trained"
    assert not torch.equal(flat, init.cpu()), "checkpoint still holds the initial weights"
    assert float(t.ts.m.abs().max()) > 0
    # resume: the first step after loading must use the loaded momentum buffer (torch.optim.SGD semantics)
    t2 = Trainer(exp, _exp_data(2, H, W, optimizer="sgd", scheduler="step"), _loader(1, B, H, W, 8), _loader(5, B, H, W, 7), B,
                 arch="tiny", compute_dtype="fp32", checkpoint=os.path.join(exp, "models", "checkpoint_epoch_final.pth"),
                 resume_training=True)
    t2.setup_model()
    assert int(t2.ts.step_count.item()) >= 1
    mom = t2.ts.m.clone()
    w0 = t2.ts.store.master.clone()
    t2.train_epoch(1)
    torch.cuda.synchronize()
    g = t2.ts.store.grads
    lr, wd = 1e-3 * 0.5, 5e-4     # one StepLR decay already applied in the first run
    gg = g + wd * w0
    expect = w0 - float(t2.lr) * (0.9 * mom + gg)      # b = mu * b_loaded + g ; p -= lr * b   (no Nesterov)
    assert torch.allclose(t2.ts.store.master, expect, rtol=1e-4, atol=1e-7)


def _two_rank_worker(rank, world, port, exp, q):
    """One of two ranks sharing the box's GPU (gloo rendezvous; the collectives move GPU tensors through the host)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, H, W = 2, 64, 64
        # UNEVEN shards: rank 0 has 3 training batches, rank 1 has 5 -- every rank must take 3 steps per epoch
        train = _loader(3 + 2 * rank, B, H, W, 10 + rank)
        valid = _loader(5, B, H, W, 20 + rank)
        torch.manual_seed(100 + rank)       # different initial weights per rank: the broadcast must equalise them
        t = Trainer(exp, _exp_data(2, H, W), train, valid, B, arch="tiny", compute_dtype="fp32", process_group=dist.group.WORLD)
        t.setup_model()
        t.training_loop()
        torch.cuda.synchronize()
        w = torch.cat([p.detach().flatten() for p in t.model.parameters()]).cpu()
        bufs = t.ts.store.bufs.detach().cpu()
        # numpy, pickled by value: a torch tensor would travel as a shared-memory handle the parent has to fetch from
        # THIS process, which may already have exited
        q.put((rank, t.iterations, w.numpy().copy(), float(t.lr), float(t.train_loss), bufs[:64].numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_trainer_two_ranks_uneven_loaders_stay_in_step(tmp_path):
    """the one-process-per-GPU paths of the Trainer on MORE than one rank -- state broadcast, bucketed
    all-reduce per step, rank-averaged epoch statistics, rank-0-only checkpoints -- with loaders of different lengths
    (would deadlock without _common_steps).  Both ranks end with identical weights and learning rate."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    exp = str(tmp_path / "dp")
    os.makedirs(exp)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, exp, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda r: r[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (r0, it0, w0, lr0, tl0, b0), (r1, it1, w1, lr1, tl1, b1) = res
    assert it0 == it1 == 6, (it0, it1)                      # 2 epochs x min(3, 5) steps
    assert np.array_equal(w0, w1), f"replicas diverged: max diff {float(np.abs(w0 - w1).max())}"
    assert lr0 == lr1 and tl0 == tl1                          # epoch statistics are rank averages on both
    assert not np.array_equal(b0, b1)                         # BatchNorm statistics stay per replica (nn.DataParallel keeps replica 0's)
    names = sorted(os.listdir(os.path.join(exp, "models")))
    assert names == ["checkpoint_epoch_0.pth", "checkpoint_epoch_1.pth", "checkpoint_epoch_final.pth"]


def test_f16_overflow_of_a_badly_scaled_checkpoint_is_reported_not_trained_on():
    """Forward tensors of the default mixed mode are f16 (|y| <= 65504); lib/model_setup.py:38-42 loads arbitrary checkpoints.
    A checkpoint whose convolution weights are 1e5 x too large (BatchNorm hides the scale from an fp32 run) overflows the raw
    conv outputs: the step must not train on it silently -- the BatchNorm statistics carry the infinity, the running
    statistics and the weights stay as they were (the optimiser skips the step) and the host gets a
    FloatingPointError that names the layer and the remedy.  The same checkpoint trains in the bf16 mode."""
    from stlpose_amd import PoseHighResolutionNet
    from stlpose_amd.train_step import TrainStep
    torch.manual_seed(3)
    base = PoseHighResolutionNet("tiny", "fp32")
    sd = {k: (v * 1e5 if (v.dim() == 4 and not k.startswith("final_layer")) else v.clone()) for k, v in base.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    img, tgt, tw = torch.randn(4, 3, 64, 64, generator=g), torch.rand(4, 17, 16, 16, generator=g), torch.ones(4, 17, 1)

    def run(dtype):
        m = PoseHighResolutionNet("tiny", dtype)
        m.load_state_dict(sd, strict=True)
        ts = TrainStep(m, 4, 64, 64, optimizer="adam", lr=1e-3)
        ts.load_batch(img.cuda(), tgt.cuda(), tw.cuda())
        w0, rm0 = ts.store.master.clone(), ts.store.bufs.clone()
        loss = ts.step()
        torch.cuda.synchronize()
        return m, ts, w0, rm0, float(loss.item())

    m, ts, w0, rm0, loss = run("mixed")
    # (the loss itself may even be FINITE: ReLU(NaN) = 0 wipes the poisoned maps -- which is why the guard rides on the statistics)
    with pytest.raises(FloatingPointError, match="compute_dtype='bf16'") as ei:
        ts.check_forward_range()
    assert "conv" in str(ei.value) or "layer" in str(ei.value) or "bn" in str(ei.value), str(ei.value)
    assert torch.equal(ts.store.master, w0), "the overflowed step changed the weights"
    assert torch.isfinite(ts.store.bufs).all(), "running statistics took the overflow in"
    ts.check_forward_range()     # reported once; the flag is re-armed
    m2, ts2, _, _, loss2 = run("bf16")
    assert np.isfinite(loss2)
    ts2.check_forward_range()    # nothing to report
    assert torch.isfinite(ts2.store.master).all() and not torch.equal(ts2.store.master, w0)
