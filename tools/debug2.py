import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stlpose_amd import PersonMSELoss, PoseHighResolutionNet, capi
m = PoseHighResolutionNet("tiny", "fp32").cuda().train()
img = torch.randn(2, 3, 96, 64).cuda()
out = m(img); loss = out.square().mean(); loss.backward(); torch.cuda.synchronize()
e = list(m._engines.values())[0]; st = m._store
print("slab arena finite/abs-sum", torch.isfinite(e.slab_arena).all().item(), float(e.slab_arena.abs().sum()))
print("grads abs-sum", float(st.grads.abs().sum()), "pub", float(m._grad_pub.abs().sum()))
print("slab_n", e._slab_n, e._slab_blocks, len(e.slabs))
st.grads.zero_()
capi.call("stl_reduce_slabs", e.slab_arena.data_ptr(), st.grads.data_ptr(), e._slab_tab.data_ptr(), e._slab_n, e._slab_blocks, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("after manual reduce", float(st.grads.abs().sum()))
print("conv1 grad norm", float(m.conv1.weight.grad.norm()), float(m.bn1.weight.grad.norm()))
