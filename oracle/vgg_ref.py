"""ORACLE (test infrastructure, never imported by the product path).  Pinned by fixture G10 except for
torchvision's layer list (see below).

Torch-ops restatement of ``VGGPerceptualLoss`` (reference ``src/lib/loss.py:17-58``).  The
reference takes its convolution stack from ``torchvision.models.vgg16(pretrained=True)
.features[:23]`` (torchvision 0.4.0, ``environment.yml:297``) -- a third-party dependency that
is absent from /root/reference and from this image, and whose weights need a download.  The
layer layout below is torchvision's published VGG16 configuration "D":
conv64,conv64,M,conv128,conv128,M,conv256x3,M,conv512x3 (all 3x3 pad 1 with bias, ReLU after
each conv, 2x2/2 max-pool), sliced [0:4],[4:9],[9:16],[16:23] as at loss.py:28-31.
``tests/golden/make_golden.py g10`` runs the reference's OWN VGGPerceptualLoss class on that layer
list (plain torch.nn layers standing in for the absent torchvision, synthetic weights below) and
``tests/test_oracle_golden.py`` checks this restatement against it: the reference's slicing, channel
repeat, normalisation, bilinear resize and L1 lines are pinned; the third-party layer list and the
ImageNet weights are not (parity unpinned for those two only).
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

# (features index, cin, cout) of the ten convolutions inside features[:23]
VGG16_CONVS = [(0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256),
               (12, 256, 256), (14, 256, 256), (17, 256, 512), (19, 512, 512), (21, 512, 512)]
# slices: ops are 'c<k>' (conv k of the list above, followed by ReLU) or 'p' (max-pool)
VGG16_SLICES = [["c0", "c1"], ["p", "c2", "c3"], ["p", "c4", "c5", "c6"], ["p", "c7", "c8", "c9"]]
IMAGENET_MEAN = (0.485, 0.456, 0.406)  # loss.py:37
IMAGENET_STD = (0.229, 0.224, 0.225)   # loss.py:38


def synth_vgg_weights(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic He-normal weights under torchvision's key names (features.<idx>.weight/bias)."""
    out = {}
    for idx, cin, cout in VGG16_CONVS:
        for leaf, shape in (("weight", (cout, cin, 3, 3)), ("bias", (cout,))):
            key = f"features.{idx}.{leaf}"
            rng = np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) + 7919 * seed) & 0xFFFFFFFF))
            if leaf == "weight":
                v = rng.normal(0.0, np.sqrt(2.0 / (cin * 9)), shape)
            else:
                v = rng.normal(0.0, 0.05, shape)
            out[key] = torch.from_numpy(v.astype(np.float32))
    return out


def vgg_features(x: torch.Tensor, weights: Dict[str, torch.Tensor]):
    """Returns the four slice outputs for a normalised NCHW batch."""
    feats = []
    for ops in VGG16_SLICES:
        for op in ops:
            if op == "p":
                x = F.max_pool2d(x, 2, 2)
            else:
                idx = VGG16_CONVS[int(op[1:])][0]
                x = F.relu(F.conv2d(x, weights[f"features.{idx}.weight"], weights[f"features.{idx}.bias"], padding=1))
        feats.append(x)
    return feats


def vgg_perceptual_loss(inp: torch.Tensor, tgt: torch.Tensor, weights, resize: bool = True) -> torch.Tensor:
    """loss.py:41-58: channel repeat for non-RGB, ImageNet normalise, optional bilinear
    224x224 (align_corners=False), sum over slices of mean |x - y|."""
    if inp.shape[1] != 3:
        inp = inp.repeat(1, 3, 1, 1)
        tgt = tgt.repeat(1, 3, 1, 1)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    inp = (inp - mean) / std
    tgt = (tgt - mean) / std
    if resize:
        inp = F.interpolate(inp, mode="bilinear", size=(224, 224), align_corners=False)
        tgt = F.interpolate(tgt, mode="bilinear", size=(224, 224), align_corners=False)
    loss = inp.new_zeros(())
    for fx, fy in zip(vgg_features(inp, weights), vgg_features(tgt, weights)):
        loss = loss + F.l1_loss(fx, fy)
    return loss


# ------------------------------------------------------------------------------------------------ V2 (no reference item)
# VGG19 content + Gram style loss named by BASELINE.json configs[3]/[4].  /root/reference contains nothing of the kind
# (grep gram|vgg19|adain -> 0 hits): PARITY UNPINNED by construction.  Restated from the published
# method (Gatys et al. 2016 / Johnson et al. 2016): torchvision VGG19 configuration "E" (conv counts 2,2,4,4,4), style
# taps relu1_1, relu2_1, relu3_1, relu4_1, relu5_1, content tap relu4_2, Gram G = F F^T / (C*H*W),
# loss = content_weight * mse(F_x, F_c) + style_weight * sum_l mse(G_l(x), G_l(s)).
VGG19_CONVS = [(0, 3, 64, False), (2, 64, 64, False), (5, 64, 128, True), (7, 128, 128, False), (10, 128, 256, True),
               (12, 256, 256, False), (14, 256, 256, False), (16, 256, 256, False), (19, 256, 512, True), (21, 512, 512, False),
               (23, 512, 512, False), (25, 512, 512, False), (28, 512, 512, True)]   # (features idx, cin, cout, pool before)
VGG19_STYLE_TAPS = (0, 2, 4, 8, 12)   # positions in VGG19_CONVS: relu1_1, 2_1, 3_1, 4_1, 5_1
VGG19_CONTENT_TAP = 9                 # relu4_2


def synth_vgg19_weights(seed: int = 0) -> Dict[str, torch.Tensor]:
    out = {}
    for idx, cin, cout, _ in VGG19_CONVS:
        for leaf, shape in (("weight", (cout, cin, 3, 3)), ("bias", (cout,))):
            key = f"features.{idx}.{leaf}"
            rng = np.random.Generator(np.random.PCG64((zlib.crc32(("vgg19." + key).encode()) + 7919 * seed) & 0xFFFFFFFF))
            v = rng.normal(0.0, np.sqrt(2.0 / (cin * 9)), shape) if leaf == "weight" else rng.normal(0.0, 0.05, shape)
            out[key] = torch.from_numpy(v.astype(np.float32))
    return out


def vgg19_taps(x: torch.Tensor, weights: Dict[str, torch.Tensor]):
    """ImageNet-normalised NCHW batch -> list of the 13 post-ReLU feature maps."""
    feats = []
    for idx, _, _, pool in VGG19_CONVS:
        if pool:
            x = F.max_pool2d(x, 2, 2)
        x = F.relu(F.conv2d(x, weights[f"features.{idx}.weight"], weights[f"features.{idx}.bias"], padding=1))
        feats.append(x)
    return feats


def gram(f: torch.Tensor) -> torch.Tensor:
    b, c, h, w = f.shape
    m = f.reshape(b, c, h * w)
    return torch.bmm(m, m.transpose(1, 2)) / (c * h * w)


def vgg19_style_content_loss(x, content, style, weights, content_weight=1.0, style_weight=1e5):
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    fx, fc, fs = (vgg19_taps((t - mean) / std, weights) for t in (x, content, style))
    c_loss = F.mse_loss(fx[VGG19_CONTENT_TAP], fc[VGG19_CONTENT_TAP])
    s_loss = sum(F.mse_loss(gram(fx[i]), gram(fs[i])) for i in VGG19_STYLE_TAPS)
    return content_weight * c_loss + style_weight * s_loss, c_loss, s_loss
