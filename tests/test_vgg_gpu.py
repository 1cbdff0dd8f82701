"""GPU: VGG16 perceptual loss on the HIP kernels vs the torch-ops restatement (oracle/vgg_ref.py,
parity UNPINNED: the reference's VGG needs torchvision + downloaded weights; synthetic weights here)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vgg_ref  # noqa: E402
from stlpose_amd import VGGPerceptualLoss  # noqa: E402


@pytest.mark.parametrize("dt,tol", [("fp32", 1e-3), ("bf16", 3e-2)])
@pytest.mark.parametrize("case", [dict(B=2, C=3, H=64, W=48, resize=False), dict(B=2, C=1, H=40, W=56, resize=True),
                                  dict(B=1, C=3, H=300, W=260, resize=True)])
def test_vgg_perceptual_loss_matches_oracle(case, dt, tol):
    w = vgg_ref.synth_vgg_weights()
    g = torch.Generator().manual_seed(3)
    a = torch.rand(case["B"], case["C"], case["H"], case["W"], generator=g)
    b = torch.rand(case["B"], case["C"], case["H"], case["W"], generator=g)
    ref = vgg_ref.vgg_perceptual_loss(a, b, w, resize=case["resize"]).item()
    m = VGGPerceptualLoss(resize=case["resize"], state_dict=w, compute_dtype=dt).cuda()
    got = m(a.cuda(), b.cuda()).item()
    assert abs(got - ref) <= tol * abs(ref), (got, ref)
    assert abs(m(a.cuda(), a.cuda()).item()) < 1e-6          # identical inputs -> exactly 0
    # reference-module key names load too
    sd = {k: v for k, v in m.state_dict().items()}
    assert "blocks.2.14.weight" in sd and "mean" in sd and sd["blocks.3.21.bias"].shape == (512,)


def test_vgg_cpu_fails_loudly():
    m = VGGPerceptualLoss(resize=False, state_dict=vgg_ref.synth_vgg_weights())
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.rand(1, 3, 16, 16), torch.rand(1, 3, 16, 16))
