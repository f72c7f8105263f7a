"""Minimal `_base_`-aware Python-dict config loader.

Re-states the subset of third_party/mmcv/mmcv/utils/config.py the model configs use:
``Config.fromfile`` executes a .py config, merges the files listed in ``_base_`` (child keys
override, dicts merge recursively, ``_delete_=True`` replaces) and exposes the result with
attribute access.
"""
import copy
import os
import runpy

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'


class ConfigDict(dict):
    """dict with attribute access and recursive conversion (addict.Dict subset)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _hook(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._hook(e) for e in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._hook(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{self.__class__.__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def setdefault(self, k, d=None):
        if k not in self:
            self[k] = d
        return self[k]

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __deepcopy__(self, memo):
        other = self.__class__()
        memo[id(self)] = other
        for k, v in self.items():
            other[copy.deepcopy(k, memo)] = copy.deepcopy(v, memo)
        return other

    def to_dict(self):
        def conv(v):
            if isinstance(v, ConfigDict):
                return {k: conv(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(conv(e) for e in v)
            return v
        return conv(self)


def _merge_a_into_b(a, b):
    b = copy.copy(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and not v.get(DELETE_KEY, False):
            if not isinstance(b[k], dict):
                raise TypeError(f'{k}={v} in child config cannot inherit from base because {k} '
                                f'is a dict in the child config but is of type {type(b[k])} in base')
            b[k] = _merge_a_into_b(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = v
    return b


class Config:

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', ConfigDict(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def _file2dict(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        if not filename.endswith('.py'):
            raise IOError('Only py type is supported')
        ns = runpy.run_path(filename)
        cfg = {k: v for k, v in ns.items()
               if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            base = base if isinstance(base, list) else [base]
            merged = {}
            for f in base:
                b = Config._file2dict(os.path.join(os.path.dirname(filename), f))
                dup = merged.keys() & b.keys()
                if dup:
                    raise KeyError(f'Duplicate key is not allowed among bases: {dup}')
                merged.update(b)
            cfg = _merge_a_into_b(cfg, merged)
        return cfg

    @staticmethod
    def fromfile(filename):
        return Config(Config._file2dict(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, k, d=None):
        return self._cfg_dict.get(k, d)

    def __repr__(self):
        return f'Config (path: {self._filename}): {dict(self._cfg_dict)!r}'
