import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import hrnet_ref, pose_ref
from tests.golden.make_golden import synth_batch
from stlpose_amd import PoseHighResolutionNet
from stlpose_amd.train_step import TrainStep
torch.set_num_threads(8)
def load(m):
    sd = {k: torch.from_numpy(hrnet_ref.synth_tensor(k, tuple(v.shape))) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True); return m
for opt in ("sgd", "adam"):
    ref = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("tiny")).train()
    ref64 = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("tiny")).double().train()
    m = load(PoseHighResolutionNet("tiny", "fp32")).cuda()
    if opt == "adam":
        ro = torch.optim.Adam(ref.parameters(), lr=1e-3); ro64 = torch.optim.Adam(ref64.parameters(), lr=1e-3)
        ts = TrainStep(m, 2, 96, 64, optimizer="adam", lr=1e-3)
    else:
        kw = dict(lr=1e-2, momentum=0.9, weight_decay=5e-4, nesterov=True)
        ro = torch.optim.SGD(ref.parameters(), **kw); ro64 = torch.optim.SGD(ref64.parameters(), **kw)
        ts = TrainStep(m, 2, 96, 64, optimizer="sgd", **kw)
    for step in range(3):
        img, tgt, tw = synth_batch(2, 96, 64, seed=300 + step)
        w_before = {k: v.clone() for k, v in ref.state_dict().items()}
        ls = []
        for r, o, dt in ((ref, ro, torch.float32), (ref64, ro64, torch.float64)):
            o.zero_grad()
            l = pose_ref.person_mse_loss(r(torch.from_numpy(img).to(dt)), torch.from_numpy(tgt).to(dt), torch.from_numpy(tw).to(dt))
            l.backward(); o.step(); ls.append(l.item())
        ts.load_batch(torch.from_numpy(img).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda())
        l = float(ts.step().item())
        torch.cuda.synchronize()
        sd = m.state_dict(); s64 = ref64.state_dict()
        worst = (0, ""); worst32 = (0, "")
        for k, v in ref.state_dict().items():
            if v.dtype != torch.float32 or "running" in k: continue
            d = float((sd[k].cpu().double() - s64[k]).abs().max()); d32 = float((v.double() - s64[k]).abs().max())
            worst = max(worst, (d, k)); worst32 = max(worst32, (d32, k))
        print(f"{opt} step {step}: loss hip {l:.6f} torch32 {ls[0]:.6f} torch64 {ls[1]:.6f} | weight diff vs f64: hip {worst[0]:.3e} ({worst[1]}) torch32 {worst32[0]:.3e} ({worst32[1]})", flush=True)
# load_pretrained case
torch.manual_seed(0)
m = PoseHighResolutionNet("tiny", "fp32")
ref = hrnet_ref.load_synth(hrnet_ref.RefPoseNet("tiny"))
sd = dict(ref.state_dict()); sd.pop("final_layer.bias")
torch.save(sd, "/tmp/w.pth")
m.load_pretrained("/tmp/w.pth")
print("cpu equal", torch.equal(m.layer1[0].conv2.weight, sd["layer1.0.conv2.weight"]))
m = m.cuda().eval()
ref.final_layer.bias.data.zero_()
x = torch.randn(1, 3, 64, 64)
with torch.no_grad():
    o = m(x.cuda()).cpu(); r = ref.eval()(x)
print("out hip absmax", float(o.abs().max()), "ref", float(r.abs().max()), "err", float((o - r).abs().max()))
print("after fwd equal", torch.equal(m.layer1[0].conv2.weight.cpu(), sd["layer1.0.conv2.weight"]), float(m.final_layer.bias.abs().max()))
