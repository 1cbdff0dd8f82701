"""Offline perceptual-loss producer.

The reference trains on Styled-COCO with a per-image perceptual loss that it expects to find
precomputed in ``<dict_path>/perceptual_loss_dict_alpha_{alpha}_styles_{styles}.json``
(``lib/loss.py:153-198``: a JSON object mapping the styled image's file name to a float); the script
that writes it (``aux_create_offline_perceptual_loss.py``, named in ``loss.py:192``) is not part of the
reference tree.  This module produces that file with the HIP ``VGGPerceptualLoss``
(``lib/loss.py:17-58`` semantics: one loss per (styled, original) image pair).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, Tuple

import torch

from .vgg import VGGPerceptualLoss


def dict_filename(alpha, styles) -> str:
    """File name ``load_perceptual_loss_dict`` looks for (lib/loss.py:185-186)."""
    return f"perceptual_loss_dict_alpha_{alpha}_styles_{styles}.json"


@torch.no_grad()
def create_offline_perceptual_loss(pairs: Iterable[Tuple[str, torch.Tensor, torch.Tensor]], vgg: VGGPerceptualLoss,
                                   dict_path: str, alpha, styles, device="cuda") -> Dict[str, float]:
    """pairs: (image name, styled image (3,H,W) or (1,H,W) in [0,1], original image, same shape).
    Writes ``dict_path/perceptual_loss_dict_alpha_{alpha}_styles_{styles}.json`` and returns the dict.
    One pair per launch keeps the value per image (the module returns the batch mean)."""
    dev = torch.device(device)
    out: Dict[str, float] = {}
    pending = []
    for name, styled, content in pairs:
        loss = vgg(styled.unsqueeze(0).to(dev, torch.float32), content.unsqueeze(0).to(dev, torch.float32))
        pending.append((name, loss))
        if len(pending) >= 256:          # one host sync per 256 images
            out.update({n: float(v) for n, v in zip([p[0] for p in pending], torch.stack([p[1] for p in pending]).tolist())})
            pending = []
    if pending:
        out.update({n: float(v) for n, v in zip([p[0] for p in pending], torch.stack([p[1] for p in pending]).tolist())})
    os.makedirs(dict_path, exist_ok=True)
    with open(os.path.join(dict_path, dict_filename(alpha, styles)), "w") as f:
        json.dump(out, f)
    return out
