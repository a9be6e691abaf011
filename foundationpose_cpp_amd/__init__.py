"""foundationpose_cpp_amd -- MI355X-native FoundationPose Register/Track hot path."""
from .api import FoundationPose, FoundationPoseError  # noqa: F401
