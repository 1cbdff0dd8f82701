"""One fresh-process forward + backward of W32 (fp32 or bf16) at bs 2 against the committed golden sample; prints the output
error, whether any gradient is NaN and the loss.  usage: flake_probe.py <package root> <tag 256x192|384x288> <dtype>"""
import os, sys
root, tag, dt = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
sys.path.insert(1, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stlpose_amd import PersonMSELoss, PoseHighResolutionNet
from oracle import hrnet_ref
from tests.golden.make_golden import synth_batch
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(here, "tests", "golden", f"g3_w32_{tag}.npz"))
h, w = (int(v) for v in tag.split("x"))
img, tgt, tw = synth_batch(2, h, w, seed=1234, sigma=3.0 if h >= 384 else 2.0)
m = PoseHighResolutionNet("w32", dt)
sd = {k: torch.from_numpy(hrnet_ref.synth_tensor(k, tuple(v.shape))) for k, v in m.state_dict().items()}
m.load_state_dict(sd, strict=True)
m = m.cuda().train()
out = m(torch.from_numpy(img).cuda())
loss = PersonMSELoss()(out, torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda())
loss.backward()
torch.cuda.synchronize()
o = out.detach().cpu().numpy()
err = np.abs(o.reshape(-1)[::64] - g["out_sample"]).max() / float(g["out_absmax"])
nan = sum(int(torch.isnan(p.grad).any()) for p in m.parameters())
print(f"{os.path.basename(os.path.abspath(root))} {tag} {dt}: out err {err:.3e} nan-grad tensors {nan} loss {loss.item():.6f} (ref {float(g['loss']):.6f})", flush=True)
