"""Why is bench.py's evaluation leg half as fast inside the full run as alone?  Replays the run's pieces in one process and
times the evaluation leg after each."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stlpose_amd import PoseHighResolutionNet
from stlpose_amd.train_step import TrainStep
dev = torch.device("cuda:0")
def ev(tag):
    print(f"{tag:44s} eval leg {bench.extra_eval_path(dev)['value']:8.1f} images/s", flush=True)
ev("alone")
model = PoseHighResolutionNet("w32", "mixed").to(dev)
ts = TrainStep(model, 32, 384, 288, optimizer="adam", lr=1e-3, device=dev)
img, tgt, tw = bench.synth_batch(32, 384, 288, 0, dev, sigma=3.0)
ts.load_batch(img, tgt, tw)
for _ in range(10):
    ts.step()
torch.cuda.synchronize()
ev("after the main train steps (objects alive)")
fams, progs = bench.time_kernel_families(ts)
ev("after time_kernel_families (850 programs)")
del ts, model, fams, progs
torch.cuda.empty_cache()
ev("after deleting them")
bench.extra_train_leg("w48", 32, 384, 288, dev, steps=6, warmup=3)
ev("after the W48 leg")
bench.extra_train_leg("w32", 32, 384, 288, dev, steps=6, warmup=3, cpu_batch=0, dtype="bf16")
ev("after the pure-bf16 leg")
