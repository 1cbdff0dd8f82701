import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stlpose_amd import PoseHighResolutionNet
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = PoseHighResolutionNet("w32", "mixed").to(dev).eval()
x = torch.randn(32, 3, 384, 288, device=dev)
with torch.no_grad():
    for _ in range(3): y = model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): y = model(x)
    torch.cuda.synchronize()
    print(f"eval forward bs32: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per call (no per-call sync)")
    t0 = time.perf_counter()
    for _ in range(10):
        y = model(x); torch.cuda.synchronize()
    print(f"eval forward bs32: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per call (sync per call)")
