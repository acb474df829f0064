"""Plugin registry + ``build_*`` helpers — the drop-in boundary of the path.

The reference sits behind mmcv's ``Registry`` (third-party, not in the tree):
one ``MODELS`` registry that every ``BACKBONES/NECKS/HEADS/LOSSES/DETECTORS/
VOXEL_ENCODERS/MIDDLE_ENCODERS`` name aliases
(mmdet3d/models/builder.py:16-28), classes register with
``@X.register_module()`` and are instantiated as ``cls(**cfg_without_type)``
(mmdet3d/models/builder.py:31-137). This file keeps that surface — same
names, same ``build(cfg, default_args=...)`` semantics, same ``KeyError``
on an unknown ``type`` — so the ``model=dict(...)`` of
``configs/gga/gga_kitti_config.py`` builds unchanged.
"""
import inspect
import warnings


class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}
        self.build_func = build_func or build_from_cfg

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{type(self).__name__}(name={self._name}, items={sorted(self._module_dict)})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register_module(self, module, module_name=None, force=False):
        if not inspect.isclass(module) and not inspect.isfunction(module):
            raise TypeError(f'module must be a class or a function, but got {type(module)}')
        if module_name is None:
            module_name = module.__name__
        names = [module_name] if isinstance(module_name, str) else module_name
        for name in names:
            if not force and name in self._module_dict:
                raise KeyError(f'{name} is already registered in {self.name}')
            self._module_dict[name] = module

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, but got {type(force)}')
        if not (name is None or isinstance(name, str) or
                (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError(f'name must be None, a str or a sequence of str, but got {type(name)}')
        if module is not None:
            self._register_module(module, name, force)
            return module

        def _register(cls):
            self._register_module(cls, name, force)
            return cls

        return _register


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg and (default_args is None or 'type' not in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry object, but got {type(registry)}')
    if not (isinstance(default_args, dict) or default_args is None):
        raise TypeError(f'default_args must be a dict or None, but got {type(default_args)}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type) or inspect.isfunction(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)(f'{obj_cls.__name__}: {e}')


MODELS = Registry('models')
BACKBONES = NECKS = HEADS = LOSSES = DETECTORS = MODELS
VOXEL_ENCODERS = MIDDLE_ENCODERS = FUSION_LAYERS = MODELS
BBOX_CODERS = Registry('bbox_coder')
CONV_LAYERS = Registry('conv layer')
PIPELINES = Registry('pipeline')             # mmdet.datasets.builder.PIPELINES (mmdet3d/datasets/builder.py)
OBJECTSAMPLERS = Registry('Object sampler')  # mmdet3d/datasets/builder.py:13
DATASETS = Registry('dataset')               # mmdet.datasets.builder.DATASETS (mmdet3d/datasets/builder.py:14)


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_voxel_encoder(cfg):
    return VOXEL_ENCODERS.build(cfg)


def build_middle_encoder(cfg):
    return MIDDLE_ENCODERS.build(cfg)


def build_bbox_coder(cfg, **default_args):
    return BBOX_CODERS.build(cfg, default_args=default_args or None)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    if train_cfg is not None or test_cfg is not None:
        warnings.warn('train_cfg and test_cfg is deprecated, please specify them in model', UserWarning)
    assert cfg.get('train_cfg') is None or train_cfg is None, \
        'train_cfg specified in both outer field and model field '
    assert cfg.get('test_cfg') is None or test_cfg is None, \
        'test_cfg specified in both outer field and model field '
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_model(cfg, train_cfg=None, test_cfg=None):
    return build_detector(cfg, train_cfg=train_cfg, test_cfg=test_cfg)
