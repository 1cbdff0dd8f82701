"""stlpose_amd -- MI355X-native HRNet / perceptual-loss hot path (drop-in for STLPose's
``models.PoseHighResolutionNet`` and the per-batch functions of ``lib/``)."""
import os as _os

# Before the first HIP call of the process: the plan's four streams are the chip's four compute pipes; any further ACTIVE
# hardware queue (a RCCL communicator's streams at GPU_MAX_HW_QUEUES > 4) is time-sliced against them (round 3: 15.39 ms per
# step at 4 queues, 22.4 / 22.1 / 23.7 at 5 / 6 / 8).  4 is also the runtime's default; a user's explicit setting wins.
_had_queues = "GPU_MAX_HW_QUEUES" in _os.environ
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
if not _had_queues:
    import torch as _torch
    if _torch.cuda.is_initialized():   # the runtime read the variable when it started: setting it now changes nothing
        import warnings as _warnings
        _warnings.warn("stlpose_amd: HIP was initialised before this import, GPU_MAX_HW_QUEUES=4 cannot take effect any more "
                       "(harmless while the runtime's default is 4; export it before the first CUDA/HIP call to be sure)")

from .hrnet import PoseHighResolutionNet  # noqa: F401,E402
from .loss import PersonMSELoss, apply_perceptual_loss  # noqa: F401,E402
from .inference import forward_pass  # noqa: F401,E402
from .pose_parsing import get_max_preds_hrnet, get_final_preds_hrnet, accuracy  # noqa: F401,E402
from .vgg import VGGPerceptualLoss  # noqa: F401,E402
