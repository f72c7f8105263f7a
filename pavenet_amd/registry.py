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
    """``cfg['type']`` (a registered name, possibly scope-qualified, or a class / function) called
    with the remaining keys of ``cfg``; keys of ``default_args`` fill in what ``cfg`` leaves out.
    Same contract as the reference's helper (registry.py:12-76): TypeError for a non-dict cfg or a
    non-Registry registry, KeyError for a missing or unknown type; a failure inside the constructor
    is re-raised as the same exception type with the class name in front."""
    if not isinstance(cfg, dict):
        raise TypeError(f'a config must be a dict, not {type(cfg).__name__}')
    if not isinstance(registry, Registry):
        raise TypeError(f'expected a Registry to build from, not {type(registry).__name__}')
    kwargs = {**(default_args or {}), **cfg}
    if 'type' not in kwargs:
        raise KeyError(f'neither the config nor its defaults name a "type": {cfg} / {default_args}')
    target = kwargs.pop('type')
    if isinstance(target, str):
        factory = registry.get(target)
        if factory is None:
            raise KeyError(f'no "{target}" among the {registry.name} types (scope {registry.scope})')
    elif inspect.isclass(target) or inspect.isfunction(target):
        factory = target
    else:
        raise TypeError(f'"type" must be a name, a class or a function, not {type(target).__name__}')
    try:
        return factory(**kwargs)
    except Exception as err:
        raise type(err)(f'{factory.__name__}: {err}')


class Registry:
    """name -> class table with a scope; child registries (one per scope) hang off a root."""

    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._scope = 'mmcv' if scope is None else scope
        self._table = {}
        self._by_scope = {}     # child registries of this one, keyed by their scope
        self.parent = parent
        self.build_func = build_func or (parent.build_func if parent is not None else build_from_cfg)
        if parent is not None:
            if self._scope in parent._by_scope:
                raise AssertionError(f'{parent.name} already has a child registry for scope {self._scope}')
            parent._by_scope[self._scope] = self

    name = property(lambda self: self._name)
    scope = property(lambda self: self._scope)
    module_dict = property(lambda self: self._table)
    children = property(lambda self: self._by_scope)

    def __len__(self):
        return len(self._table)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{type(self).__name__}(name={self._name}, scope={self._scope}, items={sorted(self._table)})'

    @staticmethod
    def split_scope_key(key):
        """'scope.Name' -> ('scope', 'Name') at the FIRST dot; 'Name' -> (None, 'Name')."""
        scope, dot, rest = key.partition('.')
        return (scope, rest) if dot else (None, key)

    def _root(self):
        node = self
        while node.parent is not None:
            node = node.parent
        return node

    def get(self, key):
        """Resolution order of the reference (registry.py:211-234): a name of this registry's own
        scope is looked up here; a foreign scope goes to the child registry of that scope, else to
        the root, which retries.  Superset: an own-scope or bare miss also asks the parent, so bare
        mmcv names resolve from a child registry (the reference returns None there)."""
        scope, bare = self.split_scope_key(key)
        mine = scope is None or scope == self._scope
        if mine:
            hit = self._table.get(bare)
            if hit is not None or self.parent is None:
                return hit
            return self.parent.get(bare if scope is None else key)
        child = self._by_scope.get(scope)
        if child is not None:
            return child.get(bare)
        return self._root().get(key) if self.parent is not None else None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _put(self, cls, names, force):
        if not inspect.isclass(cls):
            raise TypeError(f'only classes can be registered, got {type(cls).__name__}')
        for n in ([cls.__name__] if names is None else [names] if isinstance(names, str) else list(names)):
            if n in self._table and not force:
                raise KeyError(f'{self.name} already has a type called {n} (pass force=True to replace it)')
            self._table[n] = cls
        return cls

    def register_module(self, name=None, force=False, module=None):
        """Decorator ``@R.register_module()`` / ``@R.register_module(name=...)``, or a direct call
        with ``module=cls``; ``name`` may be one name or several."""
        if module is not None:
            return self._put(module, name, force)
        return lambda cls: self._put(cls, name, force)


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
