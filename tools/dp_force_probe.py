"""One-rank RCCL rehearsal of the data-parallel step (STLPOSE_DP_FORCE=1): what do the communication stream and
RCCL's own stream cost beside the four compute queues?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ["STLPOSE_DP_FORCE"] = "1"
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from stlpose_amd import PoseHighResolutionNet
from stlpose_amd.train_step import TrainStep
from bench import synth_batch
torch.manual_seed(0)
m = PoseHighResolutionNet("w32", "bf16").cuda()
for pg in (None, dist.group.WORLD):
    ts = TrainStep(m, 32, 384, 288, optimizer="adam", lr=1e-3, process_group=pg)
    ts.load_batch(*synth_batch(32, 384, 288, 0, torch.device("cuda", 0), sigma=3.0))
    for _ in range(5): ts.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ts.step()
    torch.cuda.synchronize()
    print("dp" if pg is not None else "single", f"{(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
    del ts   # the weak reference (hrnet._holder) releases the model for the next TrainStep
dist.destroy_process_group()
