import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stlpose_amd import PersonMSELoss, PoseHighResolutionNet, capi
m = PoseHighResolutionNet("tiny", "fp32").cuda().train()
img = torch.randn(2, 3, 96, 64).cuda()
out = m(img); loss = out.square().mean(); loss.backward(); torch.cuda.synchronize()
e = list(m._engines.values())[0]; st = m._store
raw = e._slab_tab.cpu().numpy().tobytes()
tab = (capi.Slab * e._slab_n).from_buffer_copy(raw)
for i in [0, 2, 118, 119]:
    t = tab[i]; print(i, t.part_off, t.grad_off, t.nsplit, t.Co, t.Ci, t.ks, t.Cip, t.patch, t.blk0, t.pad)
t = tab[118]
n = t.Co * t.Ci * t.ks * t.ks
sl = e.slab_arena[t.part_off:t.part_off + t.nsplit * n].view(t.nsplit, n)
print("slab118 abs", float(sl.abs().sum()), "nparam", st.nparam, "grads numel", st.grads.numel(), st.grads.device, st.grads.dtype, st.grads.is_contiguous())
g2 = torch.zeros_like(st.grads)
capi.call("stl_reduce_slabs", e.slab_arena.data_ptr(), g2.data_ptr(), e._slab_tab.data_ptr(), e._slab_n, e._slab_blocks, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("g2 abs", float(g2.abs().sum()), "slab arena numel", e.slab_arena.numel(), e._slab_elems)
print("ops tail", [(n_, s_) for n_, a_, s_, r_, w_ in e.bwd_ops[-3:]], "nstreams", e.nstreams)
