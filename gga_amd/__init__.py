"""gga_amd — MI355X-native implementation of the GGA training hot path.

Host side mirrors the reference's mmdet3d plugin surface (registry names,
constructor arguments, forward signatures); the data-movement-bound stages
run as hand-written HIP kernels for gfx950 behind a C-ABI shared library
(include/gga_hip.h, gga_amd/csrc/). See DESIGN.md.
"""
__version__ = '0.1.0'

from . import registry  # noqa: E402,F401
from . import losses, bbox_coders, voxel_encoders, middle_encoders, sparse, sparse_encoder, backbones, dense_heads, detectors  # noqa: E402,F401
from .config import Config  # noqa: E402,F401
from .registry import build_detector, build_model  # noqa: E402,F401
from .pseudo_labels import pseudo_label_matching_kitti  # noqa: E402,F401
