"""stlpose_amd -- MI355X-native HRNet / perceptual-loss hot path (drop-in for STLPose's
``models.PoseHighResolutionNet`` and the per-batch functions of ``lib/``)."""
from .hrnet import PoseHighResolutionNet  # noqa: F401
from .loss import PersonMSELoss, apply_perceptual_loss  # noqa: F401
from .inference import forward_pass  # noqa: F401
from .pose_parsing import get_max_preds_hrnet, get_final_preds_hrnet, accuracy  # noqa: F401
from .vgg import VGGPerceptualLoss  # noqa: F401
