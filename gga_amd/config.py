"""Python-dict config files with ``_base_`` inheritance and ``_delete_`` — the subset of
``mmcv.Config`` the reference's entry points use (tools/train.py:122-124 loads
``configs/gga/*.py`` with ``Config.fromfile``; ``_delete_`` appears in
configs/gga/gga_pdg.py). Keys are reachable as attributes (``cfg.model.type``)."""
import copy
import os

BASE_KEY, DELETE_KEY = '_base_', '_delete_'


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


def _merge(a, b):
    """Merge dict ``a`` (child) into ``b`` (base)."""
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and isinstance(b[k], dict) and not v.get(DELETE_KEY, False):
            b[k] = _merge(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = v
    return b


def _load(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(f'file "{filename}" does not exist')
    if not filename.endswith('.py'):
        raise IOError('Only py type is supported')
    scope = {'__file__': filename}
    with open(filename) as f:
        exec(compile(f.read(), filename, 'exec'), scope)
    cfg = {k: v for k, v in scope.items()
           if not k.startswith('__') and not callable(v) and type(v).__name__ != 'module'}
    bases = cfg.pop(BASE_KEY, None)
    if bases is not None:
        bases = bases if isinstance(bases, list) else [bases]
        base_cfg = {}
        for bname in bases:
            sub = _load(os.path.join(os.path.dirname(filename), bname))
            dup = base_cfg.keys() & sub.keys()
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {sorted(dup)}')
            base_cfg.update(sub)
        cfg = _merge(cfg, base_cfg)
    return cfg


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    __setitem__ = __setattr__

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def merge_from_dict(self, options):
        """``--cfg-options a.b=1`` style overrides (tools/train.py:81-90)."""
        nested = {}
        for full, v in options.items():
            d = nested
            keys = full.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        object.__setattr__(self, '_cfg_dict', _wrap(_merge(nested, self._cfg_dict)))
