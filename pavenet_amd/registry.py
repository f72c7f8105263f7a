"""Scope-aware registry: the type-name surface of the reference, re-stated.

Behaviour follows third_party/mmcv/mmcv/utils/registry.py:12-76 (build_from_cfg),
:174-193 (scope split / infer) and :211-234 (get: own scope, else child of that
scope, else walk to the root and retry), so that the reference's model dicts --
``type='opera.VideoPoseV1'``, ``'mmdet.ResNet'``, ``'mmcv.MultiScaleDeformableAttention'``,
bare ``'DetrTransformerDecoderLayer'`` -- resolve unchanged.

A class's scope is the registry it is *registered into* (e.g. the fork's
``DeformableDetrTransformerDecoderV1`` is defined in mmdet but registered in
mmcv's TRANSFORMER_LAYER_SEQUENCE, hence ``'mmcv.DeformableDetrTransformerDecoderV1'``).
"""
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        if default_args is None or 'type' not in default_args:
            raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry object, but got {type(registry)}')
    args = dict(cfg)
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
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


class Registry:

    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}
        self._children = {}
        self._scope = scope if scope is not None else 'mmcv'
        if build_func is None:
            self.build_func = parent.build_func if parent is not None else build_from_cfg
        else:
            self.build_func = build_func
        self.parent = None
        if parent is not None:
            parent._add_children(self)
            self.parent = parent

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{self.__class__.__name__}(name={self._name}, scope={self._scope}, ' \
               f'items={sorted(self._module_dict)})'

    @staticmethod
    def split_scope_key(key):
        split_index = key.find('.')
        if split_index != -1:
            return key[:split_index], key[split_index + 1:]
        return None, key

    @property
    def name(self):
        return self._name

    @property
    def scope(self):
        return self._scope

    @property
    def module_dict(self):
        return self._module_dict

    @property
    def children(self):
        return self._children

    def get(self, key):
        scope, real_key = self.split_scope_key(key)
        if scope is None or scope == self._scope:
            if real_key in self._module_dict:
                return self._module_dict[real_key]
        else:
            if scope in self._children:
                return self._children[scope].get(real_key)
            parent = self.parent
            while parent is not None and parent.parent is not None:
                parent = parent.parent
            if parent is not None:
                return parent.get(key)
        # superset of the reference: a bare / own-scope miss also tries the parent chain
        # (the reference returns None here), so bare mmcv names work from a child registry
        if self.parent is not None and (scope is None or scope == self._scope):
            return self.parent.get(real_key if scope is None else key)
        return None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _add_children(self, registry):
        assert registry.scope not in self.children, \
            f'scope {registry.scope} exists in {self.name} registry'
        self.children[registry.scope] = registry

    def _register_module(self, module_class, module_name=None, force=False):
        if not inspect.isclass(module_class):
            raise TypeError(f'module must be a class, but got {type(module_class)}')
        if module_name is None:
            module_name = module_class.__name__
        if isinstance(module_name, str):
            module_name = [module_name]
        for name in module_name:
            if not force and name in self._module_dict:
                raise KeyError(f'{name} is already registered in {self.name}')
            self._module_dict[name] = module_class

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register_module(module_class=module, module_name=name, force=force)
            return module

        def _register(cls):
            self._register_module(module_class=cls, module_name=name, force=force)
            return cls

        return _register


# ---- registries, laid out as in the reference ------------------------------
# mmcv roots (mmcv/cnn/builder.py, mmcv/cnn/bricks/registry.py)
MMCV_MODELS = Registry('model', scope='mmcv')
MMCV_ATTENTION = Registry('attention', scope='mmcv')
MMCV_FEEDFORWARD_NETWORK = Registry('feed-forward Network', scope='mmcv')
MMCV_POSITIONAL_ENCODING = Registry('position encoding', scope='mmcv')
MMCV_TRANSFORMER_LAYER = Registry('transformerLayer', scope='mmcv')
MMCV_TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence', scope='mmcv')
MMCV_TRANSFORMER = Registry('Transformer', scope='mmcv')
# mmdet children (mmdet/models/builder.py:5-15)
MMDET_MODELS = Registry('models', parent=MMCV_MODELS, scope='mmdet')
# opera children (opera/models/builder.py:7-15, opera/models/utils/builder.py:11-17)
MODELS = Registry('models', parent=MMCV_MODELS, scope='opera')
BACKBONES = MODELS
NECKS = MODELS
ROI_EXTRACTORS = MODELS
SHARED_HEADS = MODELS
HEADS = MODELS
LOSSES = MODELS
DETECTORS = MODELS
ATTENTION = Registry('attention', parent=MMCV_ATTENTION, scope='opera')
POSITIONAL_ENCODING = Registry('Position encoding', parent=MMCV_POSITIONAL_ENCODING, scope='opera')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence',
                                      parent=MMCV_TRANSFORMER_LAYER_SEQUENCE, scope='opera')
TRANSFORMER = Registry('Transformer', parent=MMCV_TRANSFORMER, scope='opera')
TRANSFORMER_LAYER = Registry('transformerLayer', parent=MMCV_TRANSFORMER_LAYER, scope='opera')
FEEDFORWARD_NETWORK = Registry('feed-forward Network', parent=MMCV_FEEDFORWARD_NETWORK,
                               scope='opera')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_model(cfg, train_cfg=None, test_cfg=None):
    """opera/models/builder.py:49-59."""
    return build_detector(cfg, train_cfg=train_cfg, test_cfg=test_cfg)


def build_attention(cfg, default_args=None):
    return ATTENTION.build(cfg, default_args=default_args)


def build_feedforward_network(cfg, default_args=None):
    return FEEDFORWARD_NETWORK.build(cfg, default_args=default_args)


def build_positional_encoding(cfg, default_args=None):
    return POSITIONAL_ENCODING.build(cfg, default_args=default_args)


def build_transformer_layer(cfg, default_args=None):
    return TRANSFORMER_LAYER.build(cfg, default_args=default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return TRANSFORMER_LAYER_SEQUENCE.build(cfg, default_args=default_args)


def build_transformer(cfg, default_args=None):
    return TRANSFORMER.build(cfg, default_args=default_args)
