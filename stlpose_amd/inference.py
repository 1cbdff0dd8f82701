"""Forward wrapper (mirror of reference ``src/lib/inference.py``), flip-test kept on device."""
from __future__ import annotations

import torch

from . import capi

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
        of = model(img.flip(3)).contiguous()
        a = output.contiguous()
        out = torch.empty_like(a)
        b, j, h, w = a.shape
        capi.call("stl_flip_merge", a.data_ptr(), of.data_ptr(), out.data_ptr(), _perm(j, a.device).data_ptr(),
                  b, j, h, w, torch.cuda.current_stream().cuda_stream)
        output = out
    return output
