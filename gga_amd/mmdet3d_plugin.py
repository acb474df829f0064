"""Register the MI355X classes into the REFERENCE's registry.

Only meaningful inside a checkout of the reference (mmdet3d importable):
``import gga_amd.mmdet3d_plugin`` before ``build_model`` replaces the CUDA-backed classes by the
HIP-backed ones under the same names, so ``configs/gga/*.py`` run unchanged."""
import gga_amd
from gga_amd.registry import MODELS as _OURS

try:
    from mmdet3d.models.builder import MODELS as _THEIRS
except ImportError as e:        # not inside the reference: nothing to plug into
    raise ImportError('gga_amd.mmdet3d_plugin needs the reference (mmdet3d) to be importable') from e

NAMES = ('GGA', 'MVXTwoStageDetector_GGA', 'CenterHead_GGA', 'SeparateHead', 'SECOND', 'SECONDFPN',
         'PointPillarsScatter', 'SparseEncoder', 'HardSimpleVFE', 'PillarFeatureNet',
         'FCOSMono3D', 'PGDHead')        # configs/gga/gga_pdg.py
for _name in NAMES:
    _THEIRS.register_module(name=_name, force=True, module=_OURS.get(_name))

try:                            # the towers' deformable convolution is built through mmcv's CONV_LAYERS
    from mmcv.cnn import CONV_LAYERS as _CONV
    from gga_amd.dcn import ModulatedDeformConv2dPack as _DCN
    _CONV.register_module(name='DCNv2', force=True, module=_DCN)
except ImportError:
    pass
