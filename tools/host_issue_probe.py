"""How long does the HOST need to issue one step?  If the native replay of the forward / backward programs takes about as
long as the GPU needs to run them, queues run dry and launch order decides who starts when."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
from stlpose_amd import PoseHighResolutionNet
from stlpose_amd.train_step import TrainStep
from bench import synth_batch
dev = torch.device("cuda:0")
m = PoseHighResolutionNet("w32", "bf16").to(dev)
ts = TrainStep(m, 32, 384, 288, optimizer="adam", lr=1e-3, device=dev)
img, tgt, tw = synth_batch(32, 384, 288, 0, dev, sigma=3.0)
ts.load_batch(img, tgt, tw)
for _ in range(5): ts.step()
torch.cuda.synchronize()
e = ts.eng
st = torch.cuda.current_stream().cuda_stream
res = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e.forward(st); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    e.backward(st); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    res.append((t1 - t0, t2 - t0, t3 - t2, t4 - t2))
import statistics as S
f_h, f_g, b_h, b_g = (S.median(r[i] for r in res) * 1e3 for i in range(4))
print(f"forward : host issue {f_h:.2f} ms ({len(e.fwd_ops)} ops), until the GPU is done {f_g:.2f} ms")
print(f"backward: host issue {b_h:.2f} ms ({len(e.bwd_ops)} ops), until the GPU is done {b_g:.2f} ms")
