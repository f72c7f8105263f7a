"""Import shim for the zgspose/PAVENet reference -- TEST INFRASTRUCTURE ONLY.

This file is used *only* in the build container (where /root/reference is
mounted) by ``oracle/gen_golden.py`` to produce the small golden fixtures under
``tests/golden/``.  Nothing in ``pavenet_amd/`` imports it, and nothing on the
GPU box can (the reference does not travel).

What it does:
  1. puts the reference's vendored mmcv / mmdet / opera on ``sys.path``;
  2. supplies a minimal ``addict.Dict`` and MagicMock stand-ins for packages
     that are import-time-only on the inference path (cv2, yapf, torchvision,
     pycocotools, ... and the compiled ``mmcv._ext``);
  3. routes ``MultiScaleDeformableAttnFunction.apply`` to the reference's own
     pure-PyTorch sampler ``multi_scale_deformable_attn_pytorch``
     (third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py:92-149) and makes the
     modules take their "CUDA" branch, because the reference's CPU branches are
     broken (6-arg call of a 4-arg function, OT:1843-1851, MO:1561-1569);
  4. turns the debug visualisation left in the forward path into a no-op
     (OT:1818-1830).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

REF = os.environ.get('PAVENET_REFERENCE', '/root/reference')

_MOCKED = ('cv2', 'yapf', 'torchvision', 'easydict', 'pycocotools',
           'xtcocotools', 'terminaltables', 'shapely', 'termcolor',
           'motmetrics', 'imagecorruptions', 'mmcv._ext', 'matplotlib',
           'seaborn', 'scipy.optimize', 'clip', 'lap', 'tqdm_', 'PIL',
           'cityscapesscripts', 'lvis', 'onnx', 'onnxruntime', 'tensorrt',
           'pandas_', 'sklearn_', 'json_tricks', 'munkres', 'poseval',
           'ftfy', 'timm', 'mmpose', 'mmtrack', 'mmcls', 'albumentations',
           'instaboostfast', 'panopticapi', 'skimage', 'tifffile', 'turbojpeg',
           'lmdb', 'transformers', 'petrel_client', 'mc', 'ruamel', 'regex_', 'IPython')


class _Dict(dict):
    """Minimal addict.Dict: attribute access + recursive dict conversion."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for a in args:
            if not a:
                continue
            items = a.items() if isinstance(a, dict) else a
            for k, v in items:
                self[k] = self._hook(v)
        for k, v in kwargs.items():
            self[k] = self._hook(v)

    @classmethod
    def _hook(cls, item):
        if isinstance(item, dict):
            return cls(item)
        if isinstance(item, (list, tuple)):
            return type(item)(cls._hook(e) for e in item)
        return item

    def __setattr__(self, name, value):
        self[name] = value

    def __setitem__(self, name, value):
        super().__setitem__(name, value)

    def __getattr__(self, item):
        return self.__getitem__(item)

    def __missing__(self, name):
        raise KeyError(name)

    def __delattr__(self, name):
        del self[name]

    def to_dict(self):
        base = {}
        for k, v in self.items():
            if isinstance(v, type(self)):
                base[k] = v.to_dict()
            elif isinstance(v, (list, tuple)):
                base[k] = type(v)(
                    e.to_dict() if isinstance(e, type(self)) else e for e in v)
            else:
                base[k] = v
        return base

    def copy(self):
        import copy
        return copy.copy(self)

    def deepcopy(self):
        import copy
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        import copy
        other = self.__class__()
        memo[id(self)] = other
        for k, v in self.items():
            other[copy.deepcopy(k, memo)] = copy.deepcopy(v, memo)
        return other

    def update(self, *args, **kwargs):
        other = {}
        if args:
            other.update(args[0])
        other.update(kwargs)
        for k, v in other.items():
            if (k not in self or not isinstance(self[k], dict)
                    or not isinstance(v, dict)):
                self[k] = self._hook(v)
            else:
                self[k].update(v)

    def __getstate__(self):
        return self.to_dict()

    def __setstate__(self, state):
        self.update(state)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        self[key] = default
        return default


class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):

    def find_spec(self, fullname, path, target=None):
        root = fullname.split('.')[0]
        if fullname in _MOCKED or root in _MOCKED:
            return importlib.machinery.ModuleSpec(fullname, self,
                                                  is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        m.__version__ = '0.0.0'
        return m

    def exec_module(self, module):
        pass


_INSTALLED = False


def install():
    """Make ``import mmcv, mmdet, opera`` resolve to the reference tree."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not os.path.isdir(REF):
        raise RuntimeError(f'reference tree not found at {REF}')
    sys.dont_write_bytecode = True
    addict = types.ModuleType('addict')
    addict.Dict = _Dict
    sys.modules['addict'] = addict
    sys.meta_path.insert(0, _MockFinder())
    for p in (REF, os.path.join(REF, 'third_party', 'mmdetection'),
              os.path.join(REF, 'third_party', 'mmcv')):
        if p not in sys.path:
            sys.path.insert(0, p)

    import torch
    # every module must take its "CUDA" branch (the CPU branches are broken)
    torch.cuda.is_available = lambda: True
    try:
        torch.Tensor.is_cuda = property(lambda self: True)
    except Exception:  # pragma: no cover
        pass

    import mmcv.ops.multi_scale_deform_attn as MO  # noqa: N812

    class _ApplyViaPytorch:
        """Stand-in: same call signature as the autograd Function."""

        @staticmethod
        def apply(value, shapes, lsi, loc, w, step):
            return MO.multi_scale_deformable_attn_pytorch(value, shapes, loc, w)

    MO.MultiScaleDeformableAttnFunction = _ApplyViaPytorch
    import opera.models.utils.transformer as OT  # noqa: N812
    OT.MultiScaleDeformableAttnFunction = _ApplyViaPytorch
    for name in dir(OT):
        cls = getattr(OT, name)
        if isinstance(cls, type) and hasattr(cls, 'vis_attention'):
            cls.vis_attention = lambda self, *a, **k: None
    import builtins
    _print = builtins.print

    def _quiet_print(*a, **k):  # swallow the debug prints left in forward
        if a and isinstance(a[0], str) and (
                a[0] == 'end' or a[0].startswith('image h') or
                a[0].startswith('Inference time')):
            return
        _print(*a, **k)

    builtins.print = _quiet_print
    _INSTALLED = True
    return MO, OT


def build_reference_model(cfg_path, seed=0, cfg_overrides=None):
    """Build a reference detector from one of its own config files."""
    install()
    import torch
    from mmcv import Config
    from opera.models import build_model
    cfg = Config.fromfile(os.path.join(REF, cfg_path))
    cfg.model['init_cfg'] = None
    if 'backbone' in cfg.model and isinstance(cfg.model['backbone'], dict):
        cfg.model['backbone']['init_cfg'] = None
        cfg.model['backbone'].pop('pretrained', None)
    cfg.model.pop('pretrained', None)
    if cfg_overrides:
        cfg_overrides(cfg)
    torch.manual_seed(seed)
    model = build_model(cfg.model)
    model.init_weights()
    model.eval()
    return model, cfg
