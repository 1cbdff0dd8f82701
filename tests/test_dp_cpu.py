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


def test_single_process_is_identity():
    from stlpose_amd.dp import FlatAllReduce
    g = torch.arange(10.0)
    ar = FlatAllReduce(g, None, bucket_mb=1e-5)
    ar.all_reduce()
    assert ar.world == 1 and ar.grad_scale == 1.0 and torch.equal(g, torch.arange(10.0))
