"""Reproduce the round-4 intermittent wrong result ON PURPOSE (GPU): the statistics arenas of a W32 plan sized with the key set
of a `tiny` registry -- what the id()-keyed cache of rounds 2-4 returned when CPython re-used a collected registry's address
(engine.bn_weight_keys).  Runs the failing test's case (W32 fp32 384x288 bs 2 against the reference fixture) with the arena as
it should be and with the short arena; prints output error, NaN count of the gradient.  `python tools/stale_bn_cache_repro.py`"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hrnet_ref  # noqa: E402  (weights generator only)
from stlpose_amd import PersonMSELoss, PoseHighResolutionNet, engine  # noqa: E402
from stlpose_amd.arch import ARCHS, registry  # noqa: E402
from tests.golden.make_golden import synth_batch  # noqa: E402


def run(tag):
    g = np.load(os.path.join(ROOT, "tests", "golden", "g3_w32_384x288.npz"))
    img, tgt, tw = synth_batch(2, 384, 288, seed=1234, sigma=3.0)
    m = PoseHighResolutionNet("w32", "fp32")
    m.load_state_dict({k: torch.from_numpy(hrnet_ref.synth_tensor(k, tuple(v.shape))) for k, v in m.state_dict().items()}, strict=True)
    m = m.cuda().train()
    res = []
    for it in range(3):   # the overflow region is never zeroed: later passes see accumulated statistics
        m.zero_grad(set_to_none=True)
        out = m(torch.from_numpy(img).cuda())
        loss = PersonMSELoss()(out, torch.from_numpy(tgt).cuda(), torch.from_numpy(tw).cuda())
        loss.backward()
        torch.cuda.synchronize()
        o = out.detach().cpu().numpy()
        err = np.abs(o.reshape(-1)[::64] - g["out_sample"]).max() / float(g["out_absmax"])
        flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        res.append((err, int((~torch.isfinite(flat)).sum()), float(loss)))
    eng = next(iter(m._engines.values()))
    print(f"{tag}: arena {eng.stats.numel()} of {eng._stats_used} fp64 elements used by the plan")
    for it, (err, nan, loss) in enumerate(res):
        print(f"   pass {it}: output error {err:.3e} (bar 1e-3), non-finite gradient elements {nan}, loss {loss:.6f} (reference {float(g['loss']):.6f})")


if __name__ == "__main__":
    run("arena sized from the W32 registry (now)")
    stale = engine.bn_weight_keys(registry(ARCHS["tiny"]))
    real = engine.bn_stat_elems
    import math
    from stlpose_amd import capi
    engine.bn_stat_elems = lambda reg: sum(int(math.prod(s)) for k, s in reg.params if k in stale) * 2 * capi.NSHARD
    engine.Engine._check_stats_arena = lambda self, nstat: None   # the planner's new size check would refuse this plan
    run("arena sized with the tiny key set (rounds 2-4 after an id() re-use)")
    engine.bn_stat_elems = real
