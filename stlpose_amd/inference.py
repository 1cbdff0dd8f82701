"""Forward wrapper (mirror of reference ``src/lib/inference.py``), flip-test kept on device."""
from __future__ import annotations

import torch

from . import capi, ops  # noqa: F401  (ops registers the stlpose:: custom ops)

FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]  # CONSTANTS.py:65


def _perm(j: int, device) -> torch.Tensor:
    p = list(range(j))
    for a, b in FLIP_PAIRS:
        if a < j and b < j:
            p[a], p[b] = b, a
    return torch.tensor(p, dtype=torch.int32, device=device)


def forward_pass(model, img, model_name="HRNet", device=None, flip=False):
    """reference lib/inference.py:11-32.  flip=True: second forward on img.flip(3); flip_back
    (lib/transforms.py:147-164: reverse W, swap L/R joints), 1-px right shift with column 0 kept,
    average -- done by one HIP kernel instead of a device->numpy->device round trip."""
    if model_name != "HRNet":
        raise NotImplementedError("Wrong model name. Only ['HRNet'] supported")
    output = model(img)
    if flip is True:
        of = model(img.flip(3))
        output = torch.ops.stlpose.flip_merge(output, of, _perm(output.shape[1], output.device))   # custom op -> stl_flip_merge
    return output
