"""Losses of the hot path on MI355X (mirror of reference ``src/lib/loss.py``)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import capi, ops  # noqa: F401  (ops registers the stlpose:: custom ops)


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, target_weight):
        if not output.is_cuda:
            raise RuntimeError("PersonMSELoss (HIP) needs device tensors; there is no CPU path")
        loss, dout = torch.ops.stlpose.person_mse(output, target.to(output.device), target_weight.to(output.device), 1.0)
        ctx.save_for_backward(dout)   # custom op -> stl_mse_loss: loss and d(loss)/d(output) in one pass
        return loss

    @staticmethod
    def backward(ctx, g):
        (dout,) = ctx.saved_tensors
        return dout * g, None, None


class PersonMSELoss(nn.Module):
    """reference lib/loss.py:61-94: mean over joints of 0.5 * MSE(o_j * w_j, t_j * w_j), i.e.
    0.5 * mean(((o - t) * w)^2) over all B*J*H*W elements -- one fused HIP kernel that also
    produces d(loss)/d(output)."""

    def __init__(self, use_target_weight=1):
        super().__init__()
        self.use_target_weight = use_target_weight

    def forward(self, output, target, target_weight=1):
        if not torch.is_tensor(target_weight):
            target_weight = torch.ones(output.shape[0], output.shape[1], 1, device=output.device)
        return _MSEFn.apply(output, target, target_weight)


def apply_perceptual_loss(exp_data, params, loss, perceptual_loss):
    """reference lib/loss.py:97-150 (scalar re-weighting; same control flow and error behaviour)."""
    use = False
    if "perceptual_loss" not in exp_data["training"]:
        exp_data["training"]["perceptual_loss"] = False
    if getattr(params, "use_perceptual_loss", False) or exp_data["training"]["perceptual_loss"]:
        use = True
    if exp_data["dataset"]["dataset_name"] != "styled_coco" or use is False:
        return loss
    mean_perc = torch.mean(perceptual_loss).float().to(loss.device)
    if exp_data["training"].get("lambda_D") is not None and exp_data["training"].get("lambda_P") is not None:
        return loss * float(exp_data["training"]["lambda_D"]) + mean_perc * float(exp_data["training"]["lambda_P"])
    if "perceptual_weight" not in exp_data["training"]:
        exp_data["training"]["perceptual_weight"] = "add"
    if exp_data["training"]["perceptual_weight"] == "add":
        return loss + loss * mean_perc
    raise SystemExit(f"ERROR! Weighting method '{exp_data['training']['perceptual_weight']}' is not supported")


def perceptual_affine(exp_data, params, perceptual_loss):
    """The same decision tree as ``apply_perceptual_loss`` (reference lib/loss.py:97-150) expressed as
    ``loss -> scale * loss + offset`` with host floats, for the fused train step (the scale multiplies
    dL/dout inside the loss kernel): returns (scale, offset)."""
    tr = exp_data["training"]
    if "perceptual_loss" not in tr:
        tr["perceptual_loss"] = False
    use = bool(getattr(params, "use_perceptual_loss", False) or tr["perceptual_loss"])
    if exp_data["dataset"]["dataset_name"] != "styled_coco" or not use:
        return 1.0, 0.0
    if perceptual_loss is None:
        raise KeyError("perceptual loss enabled but the batch metadata has no 'perceptual_loss' entry")
    mean_perc = float(torch.as_tensor(perceptual_loss).float().mean())
    if tr.get("lambda_D") is not None and tr.get("lambda_P") is not None:
        return float(tr["lambda_D"]), float(tr["lambda_P"]) * mean_perc
    if "perceptual_weight" not in tr:
        tr["perceptual_weight"] = "add"
    if tr["perceptual_weight"] == "add":
        return 1.0 + mean_perc, 0.0
    raise SystemExit(f"ERROR! Weighting method '{tr['perceptual_weight']}' is not supported")
