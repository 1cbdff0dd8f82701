"""Where does the bf16 path's distance from the fp32 oracle come from?  

Fits W32 (B = 32, 384x288) to the textured batch of tests/test_parity_r3_gpu.py on the GPU, then re-runs the fp32 ORACLE on
the host with bf16 rounding switched on at ONE class of storage points at a time, and with the raw conv outputs rounded
CENTRED (bf16(y - offset_c) + offset_c, offset = batch mean or running mean of the channel):

  w      conv weights            x      conv inputs (the MFMA operand after BN + ReLU on load)
  y      raw conv outputs        s      materialised sums (block ends, exchange outputs, transitions)
  yc     raw conv outputs, centred on the batch mean      yr   ... centred on running_mean

Prints max / 99.9 % / rms error relative to |out|max, argmax kept and maps moved by more than one pixel, for every variant.
Checker-side experiment only (oracle + numpy); run on the GPU box:  python tools/bf16_error_probe.py
"""
import os
import re
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hrnet_ref, pose_ref  # noqa: E402


RT = torch.bfloat16   # rounding type of the emulation (switched to torch.float16 for the mixed mode's forward tensors)


def r16(t):
    return t.to(RT).to(t.dtype)


class Rounding:
    def __init__(self, model, what, centre=None):
        self.model, self.what, self.centre, self.handles, self.saved = model, set(what), centre, [], {}

    def __enter__(self):
        mods = dict(self.model.named_modules())
        bn_after = {}
        names = list(mods)
        for name, m in mods.items():
            if not isinstance(m, nn.Conv2d) or name == "final_layer":
                continue
            if "w" in self.what:
                self.saved[name] = m.weight.data
                m.weight.data = r16(m.weight.data)
            if "x" in self.what:
                self.handles.append(m.register_forward_pre_hook(lambda mod, args: (r16(args[0]),)))
            if "y" in self.what:
                self.handles.append(m.register_forward_hook(lambda mod, args, out: r16(out)))
            if self.centre:
                # the BatchNorm that follows this conv: next module in registration order
                bn = mods[names[names.index(name) + 1]]
                assert isinstance(bn, nn.BatchNorm2d), (name, type(bn))

                def hook(mod, args, out, bn=bn):
                    off = out.mean(dim=(0, 2, 3), keepdim=True) if self.centre == "batch" else bn.running_mean.view(1, -1, 1, 1)
                    return r16(out - off) + off
                self.handles.append(m.register_forward_hook(hook))
        if "x" in self.what and "final_layer" in mods:
            self.handles.append(mods["final_layer"].register_forward_pre_hook(lambda mod, args: (r16(args[0]),)))
        if "s" in self.what:
            def round_out(mod, args, out):
                return [r16(t) for t in out] if isinstance(out, (list, tuple)) else r16(out)
            for name, m in mods.items():
                if isinstance(m, (hrnet_ref.TwoConvUnit, hrnet_ref.ThreeConvUnit, hrnet_ref.ExchangeModule)) or re.fullmatch(r"transition\d\.\d", name):
                    self.handles.append(m.register_forward_hook(round_out))
        return self.model

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
        mods = dict(self.model.named_modules())
        for name, w in self.saved.items():
            mods[name].weight.data = w
        return False


def main():
    from tests.test_parity_r3_gpu import textured_batch, _load_synth
    from stlpose_amd import PoseHighResolutionNet
    from stlpose_amd.pose_parsing import accuracy
    from stlpose_amd.train_step import TrainStep
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    img, tgt, tw = textured_batch(32, 384, 288)
    m = _load_synth(PoseHighResolutionNet("w32", "bf16"))
    ts = TrainStep(m, 32, 384, 288, optimizer="adam", lr=1e-3)
    ti, tt, tww = torch.from_numpy(img).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda()
    ts.load_batch(ti, tt, tww)
    steps = 0
    while steps < 4000:
        for _ in range(100):
            ts.step()
        steps += 100
        with torch.no_grad():
            pck = accuracy(m(ti), tt)[1]
        if pck > 0.9:
            break
    print(f"fitted in {steps} steps, device PCK {float(pck):.4f}", flush=True)
    with torch.no_grad():
        out_hip = m(ti).cpu().numpy()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    del ts, m
    mm_ = PoseHighResolutionNet("w32", "mixed")
    mm_.load_state_dict(sd, strict=True)
    mm_ = mm_.cuda().train()
    with torch.no_grad():
        out_mixed = mm_(ti).cpu().numpy()
    del mm_
    ref = hrnet_ref.RefPoseNet("w32")
    ref.load_state_dict(sd, strict=True)
    ref.train()
    for mod in ref.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.momentum = 0.0
    x = torch.from_numpy(img)
    with torch.no_grad():
        base = ref(x).numpy()
    absmax = float(np.abs(base).max())
    pr, _ = pose_ref.get_max_preds(base)
    # |mean| / std of the raw conv outputs: how many bits centring could recover per channel
    ratios = []
    mods = dict(ref.named_modules())
    hs = []
    for name, mm in mods.items():
        if isinstance(mm, nn.Conv2d) and name != "final_layer":
            hs.append(mm.register_forward_hook(lambda mod, a, o: ratios.append((o.mean(dim=(0, 2, 3)).abs() / (o.std(dim=(0, 2, 3)) + 1e-12)).numpy())))
    with torch.no_grad():
        ref(x)
    for h in hs:
        h.remove()
    rr = np.concatenate(ratios)
    print(f"|mean|/std of raw conv-output channels: median {np.median(rr):.2f} p90 {np.quantile(rr, .9):.2f} p99 {np.quantile(rr, .99):.2f} max {rr.max():.1f}", flush=True)

    def report(tag, o):
        e = np.abs(o - base).reshape(-1) / absmax
        p, _ = pose_ref.get_max_preds(o)
        disp = np.abs(p - pr).max(-1)
        print(f"{tag:28s} max {e.max():.3e}  99.9% {np.quantile(e, .999):.3e}  rms {np.sqrt(np.mean(e ** 2)):.3e}  argmax kept {int((disp == 0).sum())}/544  "
              f"moved>1px {int((disp > 1).sum())}  PCK {pose_ref.pck_accuracy(o, tgt)[1]:.4f}", flush=True)
    print(f"oracle fp32 PCK {pose_ref.pck_accuracy(base, tgt)[1]:.4f}")
    report("HIP bf16 path", out_hip)
    report("HIP mixed path (f16 fwd)", out_mixed)
    global RT
    RT = torch.float16
    with torch.no_grad(), Rounding(ref, "wxys", None):
        report("f16 emulation w+x+y+s", ref(x).numpy())
    RT = torch.bfloat16
    for tag, what, centre in (("w only", "w", None), ("x only", "x", None), ("y only", "y", None), ("s only", "s", None),
                              ("w+x+y+s (= bf16_storage)", "wxys", None), ("w+x+s, y centred(batch)", "wxs", "batch"),
                              ("w+x+s, y centred(running)", "wxs", "running"), ("y centred(batch) only", "", "batch"), ("w+x+s, y exact", "wxs", None),
                              ("x+s, y centred(batch)", "xs", "batch")):
        with torch.no_grad(), Rounding(ref, what, centre):
            o = ref(x).numpy()
        report(tag, o)


if __name__ == "__main__":
    main()
