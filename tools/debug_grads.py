import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import hrnet_ref, pose_ref
from stlpose_amd import PersonMSELoss, PoseHighResolutionNet
g = np.load("tests/golden/g1_tiny_train.npz")
ref = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("tiny")).train()
m = PoseHighResolutionNet("tiny", "fp32"); m.load_state_dict(ref.state_dict()); m = m.cuda().train()
img = torch.from_numpy(g["img"]); tgt = torch.from_numpy(g["target"]); tw = torch.from_numpy(g["target_weight"])
out = m(img.cuda()); loss = PersonMSELoss()(out, tgt.cuda(), tw.cuda()); loss.backward(); torch.cuda.synchronize()
ro = ref(img); rl = pose_ref.person_mse_loss(ro, tgt, tw); rl.backward()
rg = dict(ref.named_parameters())
for k, p in m.named_parameters():
    a, b = p.grad.cpu(), rg[k].grad
    e = float((a - b).abs().max() / (b.abs().max() + 1e-20))
    flag = "" if e < 5e-3 else "  <<<<"
    print(f"{k:55s} |g|={float(a.norm()):.4e} ref={float(b.norm()):.4e} rel={e:.2e}{flag}")
