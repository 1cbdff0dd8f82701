"""Tiny-fixture gradients element-wise for several compute dtypes: worst tensors by max error / cosine.  usage: grad_probe.py [dtypes...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stlpose_amd import PersonMSELoss, PoseHighResolutionNet
from oracle import hrnet_ref
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(here, "tests", "golden", "g1_tiny_train.npz"))
for dt in (sys.argv[1:] or ["fp32", "bf16", "mixed"]):
    m = PoseHighResolutionNet("tiny", dt)
    m.load_state_dict({k: torch.from_numpy(hrnet_ref.synth_tensor(k, tuple(v.shape))) for k, v in m.state_dict().items()}, strict=True)
    m = m.cuda().train()
    out = m(torch.from_numpy(g["img"]).cuda())
    loss = PersonMSELoss()(out, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["target_weight"]).cuda())
    loss.backward()
    torch.cuda.synchronize()
    err = np.abs(out.detach().cpu().numpy() - g["output"]).max() / np.abs(g["output"]).max()
    grads = {k: p.grad for k, p in m.named_parameters()}
    rows = []
    for k in g.files:
        if k.startswith("grad/"):
            ref, got = g[k], grads[k[5:]].cpu().numpy()
            rows.append((float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)), float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30)), k[5:]))
    rows.sort(reverse=True)
    e = np.array([r[0] for r in rows]); c = np.array([r[1] for r in rows])
    print(f"{dt}: out err {err:.3e} loss {loss.item():.6f} (ref {float(g['loss']):.6f}); grads: max-rel median {np.median(e):.3e} p90 {np.quantile(e, .9):.3e} worst {e.max():.3e}; cos min {c.min():.6f} median {np.median(c):.6f}")
    for r in rows[:4]:
        print(f"    {r[2]}: rel {r[0]:.3e} cos {r[1]:.6f}")
