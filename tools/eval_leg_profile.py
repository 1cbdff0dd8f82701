import sys, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
r = bench.extra_eval_path(dev)
print(json.dumps({k: r[k] for k in ("value", "ms_per_batch")}))
import cProfile, pstats
from stlpose_amd import PoseHighResolutionNet
from stlpose_amd.evaluate import Evaluator
import numpy as np
model = PoseHighResolutionNet("w32", "mixed").to(dev).eval()
img, tgt, tw = bench.synth_batch(32, 384, 288, 0, dev, sigma=3.0)
rng = np.random.Generator(np.random.PCG64(5))
def loader(n):
    for bi in range(n):
        ids = (bi * 32 + np.arange(32)) // 4
        yield img, tgt, tw, dict(center=rng.uniform(100, 400, (32, 2)), scale=rng.uniform(0.8, 2.0, (32, 2)), score=rng.uniform(0.3, 1.0, 32), image_id=ids)
ev = Evaluator(model, device=dev, flip=True)
ev.evaluate_model(loader(2))
pr = cProfile.Profile(); pr.enable()
ev.evaluate_model(loader(10))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
