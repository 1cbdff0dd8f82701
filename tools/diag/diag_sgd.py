import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hrnet_ref, pose_ref
from tests.golden.make_golden import synth_batch
from stlpose_amd import PoseHighResolutionNet
from stlpose_amd.train_step import TrainStep
torch.set_num_threads(8)
ref = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("tiny")).train()
m = PoseHighResolutionNet("tiny", "fp32")
m.load_state_dict({k: torch.from_numpy(hrnet_ref.synth_tensor(k, tuple(v.shape))) for k, v in m.state_dict().items()})
m = m.cuda()
w0 = {k: v.clone() for k, v in ref.state_dict().items()}
kw = dict(lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True)
ro = torch.optim.SGD(ref.parameters(), **kw)
ts = TrainStep(m, 2, 96, 64, optimizer="sgd", **kw)
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    img, tgt, tw = synth_batch(2, 96, 64, seed=300 + step)
    ro.zero_grad()
    rl = pose_ref.person_mse_loss(ref(torch.from_numpy(img)), torch.from_numpy(tgt), torch.from_numpy(tw)); rl.backward(); ro.step()
    ts.load_batch(torch.from_numpy(img).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda())
    l = float(ts.step().item())
    print("step", step, "loss", l, rl.item())
    # gradient comparison of this step
    g = ts.store.grads.cpu(); st = ts.store
    rows = []
    for k, p in ref.named_parameters():
        a = st.param_off[k]; n = p.numel()
        d = float((g[a:a+n].view_as(p) - p.grad).abs().max() / (p.grad.abs().max() + 1e-30))
        rows.append((d, k, float(p.grad.abs().max())))
    rows.sort(reverse=True)
    print("  worst grads:", [(f"{d:.1e}", k, f"{mx:.1e}") for d, k, mx in rows[:6]])
sd = m.state_dict(); rows = []
for k, v in ref.state_dict().items():
    if "running" in k or "num_batches" in k: continue
    du_ref, du_hip = (v - w0[k]).double(), (sd[k].cpu() - w0[k]).double()
    rows.append((float((du_hip - du_ref).norm()), float(du_ref.norm()), k))
rows.sort(reverse=True)
tot_n = sum(r[0]**2 for r in rows)**0.5; tot_d = sum(r[1]**2 for r in rows)**0.5
print("total rel", tot_n / tot_d)
for n_, d_, k in rows[:10]: print(f"{k:45s} |diff| {n_:.3e} |update| {d_:.3e}")
