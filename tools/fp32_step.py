"""fp32 parity-path step time (the bench's fp32_path leg alone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
r = bench.extra_fp32_path("w32", 32, 384, 288, torch.device("cuda:0"), steps=8, warmup=3)
print(os.environ.get("STLPOSE_HIP_LIB", "current").split("/")[-1], r["ms_per_step"], "ms/step")
