"""What a forward really launched besides this package's own kernels: a census of the ATen operators executed
on device tensors, taken with a TorchDispatchMode around the call ("make fallbacks loud").

The package's kernels are C-ABI calls on raw pointers and never pass through ATen, so everything the census sees
is a torch / vendor-library launch:

* ``fallback_ops`` -- dense compute that has a kernel of this package on the hot path: GEMMs (``mm, addmm, bmm,
  baddbmm, _addmm_activation, linear``: hipBLASLt / rocBLAS), convolutions (MIOpen), attention (AOTriton),
  Layer / Group / BatchNorm (and their ``var_mean`` statistics), pooling, interpolation, softmax, top-k / sort,
  GELU, grid_sample.  One of these on a
  device tensor under ``no_grad`` means a module-level gate (``bricks.split_gemm_ok``, ``_fusable``, the shape rules
  of backbones / necks / swin / deform_attn) dropped off the hand-written path -- silently, until now.
* ``aten_launches`` -- every other ATen operator that launches a kernel (elementwise adds, ``stack``, copies,
  fills): not a fallback, but each one is a dependent launch on the latency-bound decoder tail.
* ``host_syncs`` -- operators that read a device value on the host (``_local_scalar_dense``, ``nonzero``).
* ``slow_paths`` -- this package's own generic kernels taken where a specialised form exists
  (``note_slow_path``: the first-generation stem for W % 4 != 0, the direct-gather encoder sampler, the un-fused
  T-frame attention, ...).

``detectors.VideoPoseV1.forward_device(strict=True)`` runs the forward under a census (after one un-counted
warm-up of the same shapes, which builds the per-shape constant tables with torch operators) and raises
``FallbackError`` when ``fallback_ops`` is non-zero.  ``bench.py`` prints the counts per step for the headline and
every ``extra`` workload; ``tests/test_model_gpu.py::test_no_fallback_ops_*`` asserts zero.
"""
from collections import Counter

import torch
from torch.utils._python_dispatch import TorchDispatchMode

# operator packets (aten::<name>) that are library / torch COMPUTE kernels with a counterpart in this package
FALLBACK_PACKETS = frozenset((
    'mm', 'addmm', 'bmm', 'baddbmm', '_addmm_activation', 'addmv', 'mv', 'matmul', 'linear', 'addbmm', 'dot',
    'convolution', '_convolution', 'convolution_overrideable', 'cudnn_convolution', 'miopen_convolution',
    'miopen_convolution_add_relu', 'miopen_convolution_relu', 'miopen_depthwise_convolution', 'conv2d',
    '_scaled_dot_product_flash_attention', '_scaled_dot_product_efficient_attention',
    '_scaled_dot_product_attention_math', 'scaled_dot_product_attention', '_scaled_dot_product_cudnn_attention',
    '_native_multi_head_attention', '_transform_bias_rescale_qkv', '_flash_attention_forward',
    '_efficient_attention_forward',
    'native_layer_norm', 'layer_norm', 'native_group_norm', 'group_norm', 'native_batch_norm', 'batch_norm',
    '_native_batch_norm_legit', '_native_batch_norm_legit_no_training', 'cudnn_batch_norm', 'miopen_batch_norm',
    'max_pool2d_with_indices', 'max_pool2d', 'avg_pool2d', 'adaptive_avg_pool2d', '_adaptive_avg_pool2d',
    'upsample_nearest2d', 'upsample_bilinear2d', '_upsample_nearest_exact2d',
    '_softmax', 'softmax', '_log_softmax', 'topk', 'sort', 'argsort', 'gelu', 'grid_sampler_2d', 'grid_sampler',
    'var_mean', 'std_mean', 'var', 'std',
))

HOST_SYNC_PACKETS = frozenset(('_local_scalar_dense', 'nonzero', 'item', 'equal', 'is_nonzero'))

# no kernel behind these (allocation, aliasing, metadata)
_NO_LAUNCH = frozenset((
    'empty', 'empty_like', 'empty_strided', 'new_empty', 'new_empty_strided', 'detach', 'alias', 'lift_fresh',
    '_unsafe_view', 'view', 'reshape', '_reshape_alias', 'set_', 'resize_', 'is_pinned', 'pin_memory', '_pin_memory',
    'record_stream', 'sym_size', 'sym_stride', 'sym_numel', 'sym_storage_offset', 'stride', 'size',
    'split', 'split_with_sizes', 'chunk', 'unbind', 'tensor_split', 'expand_as', 'view_as', 'contiguous',
    'result_type', 'can_cast', '_has_compatible_shallow_copy_type',
))

_ACTIVE = []          # stack of running censuses (note_slow_path reports to all of them)


class FallbackError(RuntimeError):
    """A strict forward executed a torch / vendor compute operator on a device tensor."""


def note_slow_path(name):
    """Called by the module layer where it takes one of this package's GENERIC kernels although a specialised
    form exists for the common shapes (nothing is counted unless a census is running)."""
    for c in _ACTIVE:
        c.slow_paths[name] += 1


def _on_device(x, device_types):
    if isinstance(x, torch.Tensor):
        return x.device.type in device_types
    if isinstance(x, (list, tuple)):
        return any(_on_device(v, device_types) for v in x)
    return False


class LaunchCensus(TorchDispatchMode):
    """with LaunchCensus() as c: model.forward_device(...)  ->  c.fallback_ops, c.aten_launches, c.host_syncs,
    c.slow_paths (Counters keyed by operator name), c.summary()."""

    def __init__(self, device_types=('cuda',), where=False):
        """where=True: also record the innermost source line of THIS package (file:line) every counted operator was
        called from (`.sites`: Counter keyed by (operator, 'file.py:line')) -- costs a stack walk per operator."""
        super().__init__()
        self.device_types = tuple(device_types)
        self.where = bool(where)
        self.fallback_ops = Counter()
        self.aten_launches = Counter()
        self.host_syncs = Counter()
        self.slow_paths = Counter()
        self.sites = Counter()

    def _site(self):
        import os
        import sys
        pkg = os.path.dirname(os.path.abspath(__file__))
        f = sys._getframe(2)
        while f is not None:
            fn = f.f_code.co_filename
            if fn.startswith(pkg) and not fn.endswith('census.py'):
                return f'{os.path.basename(fn)}:{f.f_lineno}'
            f = f.f_back
        return '(outside pavenet_amd)'

    def __enter__(self):
        _ACTIVE.append(self)
        return super().__enter__()

    def __exit__(self, *exc):
        try:
            return super().__exit__(*exc)
        finally:
            _ACTIVE.remove(self)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        packet = getattr(func, 'overloadpacket', None)
        name = getattr(packet, '__name__', None) or str(func)
        if name in _NO_LAUNCH or getattr(func, 'is_view', False):
            return out
        if not (_on_device(args, self.device_types) or _on_device(tuple(kwargs.values()), self.device_types)
                or _on_device(out, self.device_types)):
            return out
        if name in FALLBACK_PACKETS:
            self.fallback_ops[name] += 1
        elif name in HOST_SYNC_PACKETS:
            self.host_syncs[name] += 1
        else:
            self.aten_launches[name] += 1
        if self.where:
            self.sites[(name, self._site())] += 1
        return out

    def summary(self):
        return dict(fallback_ops=int(sum(self.fallback_ops.values())),
                    fallback_op_names=dict(self.fallback_ops),
                    aten_launches=int(sum(self.aten_launches.values())),
                    aten_launch_names=dict(self.aten_launches),
                    host_syncs=dict(self.host_syncs), slow_paths=dict(self.slow_paths),
                    **({'sites': {f'{k[0]} @ {k[1]}': v for k, v in sorted(self.sites.items())}} if self.where else {}))

    def raise_on_fallback(self, what='forward'):
        if self.fallback_ops:
            raise FallbackError(
                f'{what}: {sum(self.fallback_ops.values())} torch / vendor compute operator(s) ran on device '
                f'tensors -- a gate dropped off the hand-written path: {dict(self.fallback_ops)} '
                f'(slow paths: {dict(self.slow_paths)})')
