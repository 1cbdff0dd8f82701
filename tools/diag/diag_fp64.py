"""Diagnostic: is the B=32 fp32 gradient deviation rounding?  Oracle in fp64 vs oracle fp32 vs HIP fp32."""
import os, sys, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hrnet_ref, pose_ref
from tests.golden.make_golden import synth_batch, FULL_GRAD_KEYS
from stlpose_amd import PersonMSELoss, PoseHighResolutionNet
torch.set_num_threads(min(16, os.cpu_count() or 1))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
img, tgt, tw = synth_batch(B, 384, 288, seed=4321, sigma=3.0)
res = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    t0 = time.time()
    ref = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("w32")).to(dt).train()
    out = ref(torch.from_numpy(img).to(dt))
    loss = pose_ref.person_mse_loss(out, torch.from_numpy(tgt).to(dt), torch.from_numpy(tw).to(dt))
    loss.backward()
    res[name] = {k: p.grad.double() for k, p in ref.named_parameters()}
    print(name, "loss", loss.item(), "time", time.time() - t0, flush=True)
    del ref, out
sd = {k: torch.from_numpy(hrnet_ref.synth_tensor(k, tuple(v.shape))) for k, v in PoseHighResolutionNet("w32", "fp32").state_dict().items()}
m = PoseHighResolutionNet("w32", "fp32"); m.load_state_dict(sd); m = m.cuda().train()
out = m(torch.from_numpy(img).cuda())
PersonMSELoss()(out, torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda()).backward()
hip = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters()}
rows = []
for k in hip:
    e = res["f64"][k]
    s = float(e.abs().max()) + 1e-30
    rows.append((float((hip[k] - e).abs().max()) / s, float((res["f32"][k] - e).abs().max()) / s, k))
rows.sort(reverse=True)
print("worst HIP-vs-f64 (max-norm rel), torch-f32-vs-f64, key")
for a, b, k in rows[:25]:
    print(f"{a:.3e} {b:.3e} {k}")
print("max hip", max(r[0] for r in rows), "max torch32", max(r[1] for r in rows))
