"""GPU: VGG16 perceptual loss on the HIP kernels vs fixture G10 (the reference's own VGGPerceptualLoss run on a
torchvision-free VGG16-D layer list, synthetic weights) and vs the torch-ops restatement (oracle/vgg_ref.py,
itself pinned by G10) at further shapes, including BASELINE configs[3]'s 16 x 3 x 512 x 512."""
import os

import numpy as np
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


def test_offline_perceptual_loss_dict(tmp_path):
    """the JSON the reference's load_perceptual_loss_dict reads (lib/loss.py:153-198):
    file name, {image name: float}, one value per (styled, original) pair = the oracle's loss of that pair."""
    import json
    from stlpose_amd.perceptual_offline import create_offline_perceptual_loss, dict_filename
    w = vgg_ref.synth_vgg_weights()
    g = torch.Generator().manual_seed(5)
    pairs = [(f"{i:012d}.jpg", torch.rand(3, 64, 48, generator=g), torch.rand(3, 64, 48, generator=g)) for i in range(3)]
    vgg = VGGPerceptualLoss(resize=True, state_dict=w, compute_dtype="fp32").cuda()
    d = create_offline_perceptual_loss(pairs, vgg, str(tmp_path), alpha=0.5, styles="all")
    path = tmp_path / dict_filename(0.5, "all")
    assert path.name == "perceptual_loss_dict_alpha_0.5_styles_all.json" and path.exists()
    loaded = json.loads(open(path).read())
    assert loaded == d and set(d) == {p[0] for p in pairs}
    for name, a, b in pairs:
        ref = vgg_ref.vgg_perceptual_loss(a[None], b[None], w, resize=True).item()
        assert abs(d[name] - ref) <= 1e-3 * abs(ref)


@pytest.mark.parametrize("tag", ["rs_rgb", "rs_gray", "nr_rgb", "nr_odd", "nr_gray"])
def test_vgg_matches_reference_fixture_g10(golden_dir, tag):
    """resize=True / False, 1- and 3-channel inputs, and a size that is not a multiple of 8 (the max-pools floor)."""
    g = np.load(os.path.join(golden_dir, "g10_vgg.npz"))
    m = VGGPerceptualLoss(resize=bool(g[f"{tag}_resize"]), state_dict=vgg_ref.synth_vgg_weights(), compute_dtype="fp32").cuda()
    got = m(torch.from_numpy(g[f"{tag}_in"]).cuda(), torch.from_numpy(g[f"{tag}_tg"]).cuda()).item()
    ref = float(g[f"{tag}_loss"])
    assert abs(got - ref) <= 1e-3 * abs(ref), (tag, got, ref)


def test_vgg_cfg4_shape_16x512x512_vs_oracle():
    """BASELINE configs[3] shape: batch 16, 512 x 512, fp32, resize=False (2 x 16 images through the VGG16 slices)."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    w = vgg_ref.synth_vgg_weights()
    g = torch.Generator().manual_seed(16)
    a = torch.rand(16, 3, 512, 512, generator=g)
    b = (a + 0.2 * torch.randn(16, 3, 512, 512, generator=g)).clamp_(0, 1)
    with torch.no_grad():
        ref = sum(vgg_ref.vgg_perceptual_loss(a[i:i + 4], b[i:i + 4], w, resize=False).item() for i in range(0, 16, 4)) / 4
    m = VGGPerceptualLoss(resize=False, state_dict=w, compute_dtype="fp32").cuda()
    got = m(a.cuda(), b.cuda()).item()
    assert abs(got - ref) <= 1e-3 * abs(ref), (got, ref)
    mb = VGGPerceptualLoss(resize=False, state_dict=w, compute_dtype="bf16").cuda()
    assert abs(mb(a.cuda(), b.cuda()).item() - ref) <= 3e-2 * abs(ref)
