"""GPU: VGG19 content + Gram style loss (V2; no reference item -> checked against the published-method restatement in
oracle/vgg_ref.py only, PARITY UNPINNED)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vgg_ref  # noqa: E402
from stlpose_amd.vgg19_style import VGG19StyleLoss, vgg19_flops_per_image  # noqa: E402


@pytest.mark.parametrize("dt,tol", [("fp32", 2e-3), ("bf16", 6e-2)])
@pytest.mark.parametrize("shape", [(2, 64, 48), (1, 96, 80)])
def test_vgg19_style_content_loss_matches_oracle(shape, dt, tol):
    B, H, W = shape
    w = vgg_ref.synth_vgg19_weights()
    g = torch.Generator().manual_seed(19)
    x, c, s = (torch.rand(B, 3, H, W, generator=g) for _ in range(3))
    with torch.no_grad():
        ref_t, ref_c, ref_s = vgg_ref.vgg19_style_content_loss(x, c, s, w, 1.0, 1e3)
    m = VGG19StyleLoss(1.0, 1e3, state_dict=w, compute_dtype=dt).cuda()
    tot, cl, sl = m(x.cuda(), c.cuda(), s.cuda())
    assert abs(cl.item() - ref_c.item()) <= tol * abs(ref_c.item()), (cl.item(), ref_c.item())
    assert abs(sl.item() - ref_s.item()) <= tol * abs(ref_s.item()), (sl.item(), ref_s.item())
    assert abs(tot.item() - ref_t.item()) <= tol * abs(ref_t.item())
    # identical stylised / content / style images -> both terms vanish
    t0, c0, s0 = m(x.cuda(), x.cuda(), x.cuda())
    assert c0.item() == 0.0 and s0.item() < 1e-12 * max(1.0, ref_s.item())
    assert vgg19_flops_per_image(512, 512) > 1e11


def test_vgg19_style_cpu_fails_loudly():
    m = VGG19StyleLoss(state_dict=vgg_ref.synth_vgg19_weights())
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.rand(1, 3, 32, 32), torch.rand(1, 3, 32, 32), torch.rand(1, 3, 32, 32))
