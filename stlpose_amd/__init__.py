"""stlpose_amd -- MI355X-native HRNet / perceptual-loss hot path (drop-in for STLPose's
``models.PoseHighResolutionNet`` and the per-batch functions of ``lib/``)."""
import os as _os

# Before the first HIP call of the process: the plan's four streams are the chip's four compute pipes; any further ACTIVE
# hardware queue (a RCCL communicator's streams at GPU_MAX_HW_QUEUES > 4) is time-sliced against them (round 3: 15.39 ms per
# step at 4 queues, 22.4 / 22.1 / 23.7 at 5 / 6 / 8).  4 is also the runtime's default; a user's explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

from .hrnet import PoseHighResolutionNet  # noqa: F401,E402
from .loss import PersonMSELoss, apply_perceptual_loss  # noqa: F401,E402
from .inference import forward_pass  # noqa: F401,E402
from .pose_parsing import get_max_preds_hrnet, get_final_preds_hrnet, accuracy  # noqa: F401,E402
from .vgg import VGGPerceptualLoss  # noqa: F401,E402
