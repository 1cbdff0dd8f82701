"""CPU, world_size 2, gloo: the bucketed flat-gradient all-reduce averages like the reference's
DataParallel (sum of equal-shard mean-gradients / world == gradient of the global-batch mean)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stlpose_amd.dp import FlatAllReduce
    torch.manual_seed(0)
    w = torch.randn(1000, 7)                    # shared "model"
    x = torch.randn(8, 1000)                    # global batch of 8, sharded 4 + 4
    xs = x[rank * 4:(rank + 1) * 4]
    wl = w.clone().requires_grad_(True)
    (xs @ wl).square().mean().backward()        # local mean over the shard
    flat = wl.grad.reshape(-1).clone()
    ar = FlatAllReduce(flat, None, bucket_mb=0.004)   # ~7 buckets
    assert len(ar.buckets) > 3 and ar.bounds[0][1] == flat.numel()   # tail bucket first
    ar.launch(2)
    ar.launch()
    ar.wait()
    flat.mul_(ar.grad_scale)
    wg = w.clone().requires_grad_(True)
    (x @ wg).square().mean().backward()         # single-process global-batch gradient
    ok = torch.allclose(flat.view_as(w), wg.grad, rtol=1e-5, atol=1e-6)
    q.put((rank, bool(ok), ar.world))
    dist.destroy_process_group()


def test_flat_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
    assert res == [(0, True, 2), (1, True, 2)]


def _worker_bf16(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stlpose_amd.dp import FlatAllReduce
    torch.manual_seed(0)
    w = torch.randn(1000, 7)
    x = torch.randn(8, 1000)
    xs = x[rank * 4:(rank + 1) * 4]
    wl = w.clone().requires_grad_(True)
    (xs @ wl).square().mean().backward()
    local = wl.grad.reshape(-1).clone()
    f32, b16 = local.clone(), local.clone()
    a32 = FlatAllReduce(f32, None, bucket_mb=0.004)
    a16 = FlatAllReduce(b16, None, bucket_mb=0.004, bf16_buckets=True)
    a32.all_reduce()
    a16.launch(3)           # the overlapped form: some buckets early, the rest later, then one wait
    a16.launch()
    a16.wait()
    mean32, mean16 = f32 * a32.grad_scale, b16 * a16.grad_scale
    # averaging semantics: the bf16 buckets carry the MEAN already (grad_scale 1), the fp32 ones the sum (1/world)
    ok_scale = a16.grad_scale == 1.0 and a32.grad_scale == 0.5 and b16.dtype == torch.float32
    # each rank's contribution is rounded to bf16 once (2^-9 relative), the two-term sum once more
    tol = 2.0 ** -7 * (local.abs() / world + mean32.abs()) + 1e-12
    ok_val = bool(((mean16 - mean32).abs() <= tol).all())
    rel = float((mean16 - mean32).norm() / mean32.norm())
    a16.reduce_bucket(0)    # the per-bucket (stream-ordered) form used by the bucketed optimiser tail
    q.put((rank, ok_scale, ok_val, rel < 4e-3, len(a16.buckets) > 3))
    dist.destroy_process_group()


def test_bf16_buckets_world2_gloo_average_like_fp32():
    """bf16 gradient buckets: the collective moves bf16 copies, the result lands in the fp32
    buffer as the global-batch mean, equal to the fp32 path within bf16 rounding."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_bf16, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
    assert res == [(0, True, True, True, True), (1, True, True, True, True)], res


def _worker_steps(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stlpose_amd.trainer import Trainer
    exp = {"training": {"num_epochs": 1, "save_frequency": 1}, "dataset": {"image_size": [64, 64]}}
    import tempfile
    tr = Trainer(tempfile.mkdtemp(), exp, None, None, 2, process_group=dist.group.WORLD, device="cpu")
    got = tr._common_steps([0] * (5 + 2 * rank))        # rank 0: 5 batches, rank 1: 7
    try:
        tr._common_steps(iter([1, 2, 3]))               # no __len__ on any rank -> loud error, same on both
        err = False
    except RuntimeError:
        err = True
    q.put((rank, got, err))
    dist.destroy_process_group()


def test_trainer_agrees_on_steps_per_epoch_world2_gloo():
    """uneven per-rank loader lengths would deadlock in the per-step bucket all-reduces."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_steps, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
    assert res == [(0, 5, True), (1, 5, True)], res


def test_single_process_is_identity():
    from stlpose_amd.dp import FlatAllReduce
    g = torch.arange(10.0)
    ar = FlatAllReduce(g, None, bucket_mb=1e-5)
    ar.all_reduce()
    assert ar.world == 1 and ar.grad_scale == 1.0 and torch.equal(g, torch.arange(10.0))
