"""foundationpose_cpp_amd -- MI355X-native FoundationPose Register/Track hot path."""
from .api import FoundationPose, FoundationPoseError, load_mesh  # noqa: F401
