"""Why is bench.py's evaluation leg sometimes half as fast inside the full run as alone?  Runs it behind a train leg with its CPU baseline
(16 torch threads), with and without resetting the host thread count in between."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
print("alone            ", bench.extra_eval_path(dev)["value"], flush=True)
bench.extra_train_leg("w32", 32, 256, 192, dev, steps=6, warmup=3)
print("after train + cpu", bench.extra_eval_path(dev)["value"], "threads", torch.get_num_threads(), flush=True)
torch.set_num_threads(4)
print("threads = 4      ", bench.extra_eval_path(dev)["value"], flush=True)
print("again            ", bench.extra_eval_path(dev)["value"], "threads", torch.get_num_threads(), flush=True)
