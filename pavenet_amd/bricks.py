"""Transformer building blocks with the reference's names, ctor kwargs, forward kwargs and
state-dict keys (restated from third_party/mmcv/mmcv/cnn/bricks/transformer.py and
third_party/mmdetection/mmdet/models/utils/{transformer,positional_encoding}.py).

On the device, under ``no_grad`` and in the default GEMM mode (`set_gemm_mode('bf16x3')`) every dense op here --
Linear, Linear + identity + LayerNorm, the FFN pair, the decoders' self-attention -- is a launch of this package's
kernels (`ops.gemm_bf16x3*`, `ops.mha_core`); the torch expressions beside them serve CPU tensors, autograd and the
`'native'` (vendor fp32-MFMA) mode.  Which side a forward took is measurable: `census.LaunchCensus`,
`forward_device(strict=True)`.  Everything is inference-mode: dropout layers exist only as identities so that
configs with ``dropout=0.1`` build unchanged.

Layout note: sequence-first tensors ``[n, bs, C]`` handed between modules are kept as
*views* of batch-first contiguous storage wherever possible (``seq_first_view``), so the
``permute(1, 0, 2)`` the reference performs in every attention module is free and the
token-major ``[bs*n, C]`` matrices feed the GEMMs and HIP kernels without copies.
"""
import copy
import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import (MMCV_ATTENTION, MMCV_FEEDFORWARD_NETWORK, MMCV_POSITIONAL_ENCODING,
                       MMCV_TRANSFORMER_LAYER, MMCV_TRANSFORMER_LAYER_SEQUENCE, build_attention,
                       build_feedforward_network, build_transformer_layer)


class BaseModule(nn.Module):
    """mmcv.runner.BaseModule subset: carries init_cfg, provides init_weights()."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()
        self._is_init = True


Linear = nn.Linear


def build_norm_layer(cfg, num_features):
    """mmcv/cnn/bricks/norm.py: returns (name, layer) for LN / BN / GN."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    if t == 'LN':
        name, layer = 'ln', nn.LayerNorm(num_features, **cfg)
    elif t in ('BN', 'BN2d', 'SyncBN'):
        name, layer = 'bn', nn.BatchNorm2d(num_features, **cfg)
    elif t == 'GN':
        name, layer = 'gn', nn.GroupNorm(num_channels=num_features, **cfg)
    else:
        raise KeyError(f'Unrecognized norm type {t}')
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name, layer


def build_activation_layer(cfg):
    cfg = dict(cfg)
    t = cfg.pop('type')
    table = dict(ReLU=nn.ReLU, GELU=nn.GELU, LeakyReLU=nn.LeakyReLU, Sigmoid=nn.Sigmoid,
                 Tanh=nn.Tanh, PReLU=nn.PReLU, ELU=nn.ELU, ReLU6=nn.ReLU6)
    if t not in table:
        raise KeyError(f'Unrecognized activation type {t}')
    if t == 'GELU':
        cfg.pop('inplace', None)
    return table[t](**cfg)


def inverse_sigmoid(x, eps=1e-5):
    """mmdet/models/utils/transformer.py:390-406."""
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


def seq_first_view(x_batch_first):
    """[bs, n, C] contiguous -> [n, bs, C] view (no copy)."""
    return x_batch_first.transpose(0, 1)


def batch_first(x_seq_first):
    """[n, bs, C] -> [bs, n, C], contiguous; free when x is a seq_first_view."""
    y = x_seq_first.transpose(0, 1)
    return y if y.is_contiguous() else y.contiguous()


def layer_norm_any_layout(norm, x):
    """LayerNorm over the last dim without forcing a layout change of dims 0/1."""
    if x.dim() == 3 and not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
        return norm(x.transpose(0, 1)).transpose(0, 1)
    return norm(x)


def _fusable(*tensors):
    return all(t is not None and t.is_cuda and t.dtype == torch.float32 for t in tensors)


# Dense projections (nn.Linear on [M, K] rows).  'native': hipBLASLt fp32 (v_mfma_f32_*_f32,
# 157 TFLOP/s peak).  'bf16x3' (the default of the library and of bench.py, the headline mode): the
# hand-written split GEMM (pave_gemm_bf16x3_f32): both operands split EXACTLY into three bf16 terms,
# six bf16 MFMAs per product tile, fp32 accumulate -- fp32-level accuracy
# (tests/test_ops_gpu.py::test_gemm_bf16x3_accuracy_vs_fp64) at 1.25-1.7x hipBLASLt's fp32 rate on
# the shapes of this model (profiles/r02_gemm_shapes.txt); the whole golden / oracle GPU suite runs
# in both modes at the same tolerances.
# Layer i's closing LayerNorm can also emit `out + query_pos` for layer i+1 (one pass less);
# measured SLOWER on the bench workload (the extra 640 MB store costs more than the broadcast add
# it saves) and superseded by folding the positional term into the merged projection GEMM
# (deform_attn._forward_merged), so it is off (set the attribute for an A/B run).
FP16_ACTIVATIONS = True  # 'fp16' mode: the FFN hidden stored as fp16 between its two launches (A/B switch)
FUSE_QUERY_POS = False   # A/B switch (tools/ab_switch.py sets the attribute; never read from the environment)
_GEMM = {'mode': 'bf16x3', 'min_rows': 8192, 'ln_fused': True, 'small': True}


_PLANES = {'bf16x3': 3, 'bf16x2': 2, 'bf16': 1, 'fp16': 16}   # 16 = ops.PLANES_FP16


def set_gemm_mode(mode):
    """'native' (hipBLASLt fp32) | 'bf16x3' (exact 3-term split, fp32-level accuracy) |
    'bf16x2' (2 terms, ~2^-16) | 'bf16' / 'fp16' (plain 16-bit operands, fp32 accumulate; 'fp16' is
    BASELINE config 5's "fp16 MFMA projections": keypoints stay within 0.5 px of fp32 in the
    parity test; fp16 operands overflow above 65504)."""
    assert mode == 'native' or mode in _PLANES
    _GEMM['mode'] = mode


def get_gemm_mode():
    return _GEMM['mode']


# The modes of the LDS-DMA kernel generation: the exact 3-plane split and fp16 operands (one plane).  Every
# fused form of the package (LayerNorm / encoder-projection / chain / grouped / small-row launches, padded
# planes) exists for both; the 1- / 2-plane bf16 modes keep the first-generation row GEMM and 3x3 only.
_QMODES = ('bf16x3', 'fp16')


def fused_mode():
    return _GEMM['mode'] in _QMODES


def mode_planes():
    """The `planes` argument of ops.split_* for the current GEMM mode."""
    return _PLANES[_GEMM['mode']]


_CACHE_EPOCH = [0]


def invalidate_caches():
    """Drop every derived-operand cache of the package (split weight planes, merged / stacked /
    paired projection weights, folded BatchNorm, epilogue tables): they are keyed on the identity
    and `Tensor._version` of their sources, which in-place updates made through `.data`
    (`p.data.copy_()`, `p.data = ...`, EMA helpers) do NOT bump.  Call this after such an update;
    `load_state_dict` / `formats.load_checkpoint` and ordinary in-place ops need no call."""
    _CACHE_EPOCH[0] += 1


class SourceKey:
    """Identity + version of the tensors a derived cache was built from.  Holds the tensors
    themselves (an address is no identity: the caching allocator hands a freed parameter's address
    to its replacement with _version 0 again), so `key == SourceKey(srcs)` is true only for the
    very same, unmodified tensor objects (and the same `invalidate_caches` epoch)."""

    def __init__(self, tensors, extra=None):
        self.tensors = tuple(tensors)
        self.versions = tuple(t._version for t in self.tensors)
        self.extra = extra
        self.epoch = _CACHE_EPOCH[0]

    def __eq__(self, other):
        return (isinstance(other, SourceKey) and self.extra == other.extra
                and self.epoch == other.epoch
                and len(self.tensors) == len(other.tensors)
                and all(a is b for a, b in zip(self.tensors, other.tensors))
                and self.versions == other.versions)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


def module_tensors(module):
    """Every parameter and buffer of a module tree, in a fixed order -- what `list(m.parameters()) +
    list(m.buffers())` holds (without de-duplication), but by walking `_modules / _parameters /
    _buffers` directly: the generator chain of nn.Module costs ~0.5 ms on a ResNet-50, which sits on
    the critical path before a step's first kernel launch."""
    out = []
    stack = [module]
    while stack:
        m = stack.pop()
        out.extend(t for t in m._parameters.values() if t is not None)
        out.extend(t for t in m._buffers.values() if t is not None)
        stack.extend(c for c in m._modules.values() if c is not None)
    return out


def _split_slots(weight):
    """Cache dict that LIVES ON the tensor owning the storage (the Parameter, or the folded-BN
    weight kept in ResNet._folded / on the conv module): it dies with that tensor.  An address is
    no identity -- the caching allocator hands a freed folded weight's address to the next
    same-shaped one with _version 0 again -- so nothing here is keyed on data_ptr()."""
    owner = weight._base if weight._base is not None else weight
    slots = owner.__dict__.get('_pave_split')
    if slots is None:
        slots = owner.__dict__['_pave_split'] = {}
    return slots


def _split_cached(weight, kind, make, planes=None):
    planes = planes or _PLANES[_GEMM['mode']]
    slots = _split_slots(weight)
    slot = (tuple(weight.shape), tuple(weight.stride()), weight.storage_offset(), planes, kind)
    hit = slots.get(slot)
    stamp = (weight._version, _CACHE_EPOCH[0])
    if hit is None or hit[0] != stamp:
        with torch.no_grad():
            hit = (stamp, make(planes))
        slots[slot] = hit
        _SPLIT_STATS['made'] += 1
    return hit[1]


_SPLIT_STATS = {'made': 0}   # number of weight splits performed (tests: the path was exercised)


def exact_planes(rows):
    """'fp16' mode = BASELINE configs[4]'s "fp16 MFMA projections": the throughput-bound launches (projections,
    FFNs and convolutions over >= min_rows rows) take fp16 operands; the decoders' and heads' few-hundred-row
    Linears -- latency-bound, and the ones whose outputs are logits, key points and sigmas -- stay on the exact
    3-plane split.  Returns the `planes` override for _split_cached (None: the mode's own)."""
    return 3 if (_GEMM['mode'] == 'fp16' and rows < _GEMM['min_rows']) else None


def _split_weight(weight, rows=None):
    """Weight [N, K] -> cached slab-major bf16x3 planes (re-split when the tensor changes).  rows: the
    row count of the launch (see exact_planes)."""
    from . import ops
    return _split_cached(weight, 'gemm',
                         lambda planes: ops.split_weight_bf16x3(weight.detach().contiguous(), planes),
                         exact_planes(rows) if rows is not None else None)


def split_conv_weight(weight):
    """Folded 3x3 conv weight [Cout, Cin, 3, 3] -> cached operand of ops.conv3x3_split for the
    current GEMM mode, or None when the mode / shape does not take the split kernel."""
    if _GEMM['mode'] not in _PLANES or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) \
            or not weight.is_cuda or weight.dtype != torch.float32:
        return None
    if _GEMM['mode'] in _QMODES:       # 3 planes / fp16: zero-padded planes (HRNet's 48 / 96 channels)
        if weight.shape[0] % 4 or weight.shape[1] % 16:
            return None
    elif weight.shape[0] % 64 or weight.shape[1] % 64:
        return None
    from . import ops
    return _split_cached(weight, 'conv',
                         lambda planes: ops.split_conv3x3_weight(weight.detach(), planes))


def split_gemm_ok(x2, weight):
    """Shapes / dtypes the split GEMM takes (and where it beats the library: every K % 64 == 0,
    N % 128 == 0 shape of the model, down to the K = 64 Bottleneck tails -- tools/
    bench_gemm_shapes.py, profiles/r02_gemm_shapes.txt)."""
    return (_GEMM['mode'] in _PLANES and x2.is_cuda and x2.dtype == torch.float32
            and x2.dim() == 2 and x2.is_contiguous() and weight.dtype == torch.float32
            and weight.shape[1] % 64 == 0
            and (weight.shape[0] % 128 == 0 or (weight.shape[0] == 64 and _GEMM['mode'] in _QMODES))
            and x2.shape[0] >= _GEMM['min_rows']
            and not (torch.is_grad_enabled() and (x2.requires_grad or weight.requires_grad)))


def small_split_ok(x2, weight):
    """Shapes below split_gemm_ok's row threshold or off its column grid that still take the
    3-plane kernel (zero-padded planes): inference only, exact split mode only."""
    return (_GEMM['mode'] in _QMODES and _GEMM['small'] and x2.is_cuda and x2.dtype == torch.float32
            and x2.dim() == 2 and x2.is_contiguous() and x2.shape[0] >= 1
            and weight.dtype == torch.float32 and weight.dim() == 2
            and weight.shape[1] % 32 == 0 and weight.shape[1] >= 64
            and not torch.is_grad_enabled())


def mlp_rows(module, x, act=None):
    """A branch of the heads on [..., K] rows: nn.Linear, heads.Linear_with_norm(norm=False) or an
    nn.Sequential of those and nn.ReLU, every Linear through linear_rows (ReLU in its epilogue)
    -- on the device in the exact split mode these are launches of this package's GEMM; anything
    else (other layer types, training) is the module's own forward.  act='sigmoid': the caller's
    `.sigmoid()` of the branch output, in the last Linear's epilogue where that is a kernel of ours."""
    assert act in (None, 'sigmoid')
    post = (lambda t: t.sigmoid()) if act else (lambda t: t)
    if not (x.is_cuda and x.dtype == torch.float32 and _GEMM['mode'] in _QMODES
            and not torch.is_grad_enabled()):
        return post(module(x))
    layers = list(module) if isinstance(module, nn.Sequential) else [module]
    plan = []
    for i, m in enumerate(layers):
        if isinstance(m, nn.ReLU):
            if not plan or plan[-1][2]:
                return post(module(x))
            plan[-1][2] = True
            continue
        lin = m if isinstance(m, nn.Linear) else getattr(m, 'linear', None)
        if not isinstance(lin, nn.Linear) or (lin is not m and getattr(m, 'norm', True)):
            return post(module(x))
        use_bias = lin.bias is not None and (lin is m or bool(getattr(m, 'bias', True)))
        plan.append([lin.weight, lin.bias if use_bias else None, False])
    if act and plan and not plan[-1][2]:
        plan[-1][2] = act
        post = lambda t: t    # noqa: E731
    rows = x.reshape(-1, x.shape[-1])
    if not rows.is_contiguous():
        rows = rows.contiguous()
    for i, (w, b, relu) in enumerate(plan):
        rows = linear_rows(rows, w, b, relu=relu, exact=True)     # (head branches: never 16-bit operands)
        if i + 1 < len(plan) and not rows.is_contiguous():
            rows = rows.contiguous()
    # (an output width off the 4-column grid comes back as a column slice of a padded matrix: kept
    # as a strided view, the consumers are elementwise)
    return post(rows.unflatten(0, tuple(x.shape[:-1])))


def linear_rows(x2, weight, bias=None, relu=False, residual=None, inplace_residual=False,
                a_bias=None, exact=False):
    """act(A' @ weight^T + bias + residual) on rows [M, K]: the split GEMM when enabled and
    applicable, else hipBLASLt with the same fusions (bias / ReLU / residual in the epilogue).
    A' = relu(x2 + a_bias) when a_bias is given.  exact: never 16-bit operands ('fp16' mode: the
    3-plane split for this launch)."""
    if split_gemm_ok(x2, weight):
        from . import ops
        out = residual if (residual is not None and inplace_residual) else None
        return ops.gemm_bf16x3(x2, _split_weight(weight, 0 if exact else x2.shape[0]), bias, residual,
                               relu=relu, out=out, a_bias=a_bias)
    if small_split_ok(x2, weight) and a_bias is None:
        # every other Linear of the path -- the decoders' and heads' few-hundred-row projections,
        # FFNs and branch MLPs, any output width: the same 3-plane kernel with planes zero-padded
        # to N % 64 == 0 (and the output to N % 4 == 0), so that no library GEMM is left in the step
        from . import ops
        N = weight.shape[0]
        N4 = (N + 3) // 4 * 4
        wp = _split_cached(weight, 'gemm_pad', lambda planes: ops.split_weight_bf16x3(
            weight.detach().contiguous(), planes, pad=True), exact_planes(0))
        if N4 == N:
            out = residual if (residual is not None and inplace_residual) else None
            return ops.gemm_bf16x3(x2, wp, bias, residual, relu=relu, out=out, n_out=N)
        if residual is None:    # 1, 2, 30 outputs (class logit, refine offsets, sigmas)
            b4 = None if bias is None else _split_cached(
                bias, 'bias_pad4', lambda planes: F.pad(bias.detach(), (0, N4 - N)).contiguous())
            return ops.gemm_bf16x3(x2, wp, b4, None, relu=relu, n_out=N4)[:, :N]
    if isinstance(relu, str):     # 'gelu' / 'sigmoid' outside the split kernels: the plain torch expression
        t = linear_rows(x2, weight, bias, False, residual, inplace_residual, a_bias, exact)
        return F.gelu(t) if relu == 'gelu' else torch.sigmoid(t)
    if a_bias is not None:
        x2 = torch.relu(x2 + a_bias)
    if residual is not None:
        t = residual.addmm_(x2, weight.t()) if inplace_residual else \
            torch.addmm(residual, x2, weight.t())
        if bias is not None:
            t = t + bias if not inplace_residual else t.add_(bias)
        return torch.relu_(t) if relu else t
    if relu and bias is not None and x2.is_cuda:
        return torch._addmm_activation(bias, x2, weight.t())
    y = F.linear(x2, weight, bias)
    return torch.relu_(y) if relu else y


def linear_residual_norm(x_bf, linear, identity_bf, post_norm=None, inplace=False, pos_rows=None, out=None):
    """(x @ W^T + b + identity) [-> LayerNorm], batch-first tensors [..., C].

    Device fp32: the residual rides the GEMM (beta = 1, C = identity) and bias + LayerNorm are
    one hand-written pass (pave_bias_add_layernorm_f32) -- 2 passes over the activation instead
    of the 5 of Linear / add / LayerNorm run separately.  Elsewhere: plain torch ops."""
    C_out = linear.out_features
    if _fusable(x_bf, identity_bf) and identity_bf.is_contiguous() and x_bf.is_contiguous():
        from . import ops
        idt2 = identity_bf.reshape(-1, C_out)
        x2 = x_bf.reshape(-1, x_bf.shape[-1])
        if (_GEMM['ln_fused'] and _GEMM['mode'] in _QMODES and post_norm is not None
                and pos_rows is None and C_out == 256 and tuple(post_norm.normalized_shape) == (256,)
                and post_norm.weight is not None and post_norm.bias is not None
                and linear.weight.shape[1] % 64 == 0
                and (split_gemm_ok(x2, linear.weight) or small_split_ok(x2, linear.weight))):
            # Linear + bias + residual + LayerNorm as ONE launch (the block tile owns whole rows)
            # (out: a dense [.., 256] buffer of the caller's -- a level of a decoder's preallocated stack of
            # intermediate states -- written by this launch instead of a fresh tensor; other paths ignore it)
            dst = out.view(-1, C_out) if (out is not None and out.is_contiguous() and out.numel() == idt2.numel()
                                          and out.dtype == torch.float32 and out.device == idt2.device) else None
            t = ops.gemm_bf16x3_ln(x2, _split_weight(linear.weight, x2.shape[0]), linear.bias, idt2,
                                   post_norm.weight, post_norm.bias, post_norm.eps,
                                   out=dst if dst is not None else (idt2 if inplace else None))
            return t.view(identity_bf.shape)
        if split_gemm_ok(x2, linear.weight):
            t = linear_rows(x2, linear.weight, None, residual=idt2, inplace_residual=inplace)
        elif inplace:  # caller guarantees nobody else reads identity: no copy of C into D
            t = idt2.addmm_(x2, linear.weight.t())
        else:
            t = torch.addmm(idt2, x2, linear.weight.t())
        if post_norm is not None and pos_rows is not None:
            # second output: LayerNorm(..) + pos, the next layer's `query + query_pos`
            t, tp = ops.bias_add_layernorm(t, linear.bias, None, post_norm.weight, post_norm.bias,
                                           post_norm.eps, pos=pos_rows)
            return t.view(identity_bf.shape), tp.view(identity_bf.shape)
        if post_norm is not None:
            t = ops.bias_add_layernorm(t, linear.bias, None, post_norm.weight, post_norm.bias,
                                       post_norm.eps)
        elif linear.bias is not None:
            ops.bias_act_rows_(t, linear.bias, None, relu=False)
        return t.view(identity_bf.shape)
    out = linear(x_bf) + identity_bf
    out = post_norm(out) if post_norm is not None else out
    return (out, None) if pos_rows is not None else out


def linear_norm_fused_ok(x2, linear, norm):
    """linear_norm's one-launch form applies to these rows."""
    return (_GEMM['ln_fused'] and _GEMM['mode'] in _QMODES and linear.out_features == 256
            and isinstance(norm, nn.LayerNorm) and tuple(norm.normalized_shape) == (256,)
            and norm.weight is not None and norm.bias is not None and x2.dim() == 2 and x2.is_contiguous()
            and not torch.is_grad_enabled() and linear.weight.shape[1] % 64 == 0
            and (split_gemm_ok(x2, linear.weight) or small_split_ok(x2, linear.weight)))


def linear_norm(x, linear, norm, out=None):
    """LayerNorm(Linear(x)) on [..., C] rows: ONE launch of the Linear + LayerNorm kernel in the
    exact split mode (256-wide rows, device fp32, no grad), plain torch modules otherwise.  out: a
    [rows, 256] fp32 buffer for the one-launch form (callers check linear_norm_fused_ok first)."""
    x2 = x.reshape(-1, x.shape[-1])
    if linear_norm_fused_ok(x2, linear, norm):
        from . import ops
        t = ops.gemm_bf16x3_ln(x2, _split_weight(linear.weight, x2.shape[0]), linear.bias, None, norm.weight,
                               norm.bias, norm.eps, out=out)
        return t.view(*x.shape[:-1], 256)
    assert out is None, 'linear_norm: out= only with the one-launch form'
    return norm(linear(x))


class ConvModule(nn.Module):
    """conv -> norm -> act with mmcv's attribute names (``conv``, ``gn`` / ``bn``)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), inplace=True):
        super().__init__()
        assert conv_cfg is None or conv_cfg.get('type', 'Conv2d') in ('Conv2d', 'Conv')
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation,
                              groups, bias)
        self.norm_name = None
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            act_cfg = dict(act_cfg)
            if act_cfg['type'] not in ('Tanh', 'PReLU', 'Sigmoid', 'GELU'):
                act_cfg.setdefault('inplace', inplace)
            self.activate = build_activation_layer(act_cfg)

    # True: run-to-run bit-reproducible convolution (MIOpen's fp32 kernels are not, see
    # tools/debug_determinism.py): 1x1 as a hipBLASLt row GEMM, 3x3 / pad 1 through the
    # hand-written MFMA implicit GEMM.  Set per instance by `set_deterministic`.
    deterministic = False

    def _conv_deterministic(self, x):
        c = self.conv
        k, cin, cout = c.kernel_size, c.in_channels, c.out_channels
        if not (x.is_cuda and x.dtype == torch.float32 and c.groups == 1 and c.dilation == (1, 1)):
            return None
        x = x.contiguous(memory_format=torch.channels_last)
        if k == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0):
            n, _, h, w = x.shape
            rows = x.permute(0, 2, 3, 1).reshape(-1, cin)
            wt = c.weight.flatten(1).t()
            y = rows @ wt if c.bias is None else torch.addmm(c.bias, rows, wt)
            return y.view(n, h, w, cout).permute(0, 3, 1, 2)
        if k == (3, 3) and c.padding == (1, 1) and c.stride[0] == c.stride[1] and \
                c.stride[0] in (1, 2) and cin % 32 == 0 and cout % 64 == 0:
            from . import ops
            return ops.conv3x3_nhwc(x, c.weight.permute(2, 3, 1, 0).contiguous(), c.bias,
                                    stride=c.stride[0], relu=False)
        return None

    def forward(self, x):
        y = self._conv_deterministic(x) if self.deterministic and not self.training else None
        x = self.conv(x) if y is None else y
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def set_deterministic(model, flag=True):
    """Bit-reproducible forward: every ConvModule and the ResNet 3x3 convolutions leave MIOpen
    (whose searched fp32 kernels accumulate in a run-dependent order) for hipBLASLt row GEMMs /
    the hand-written MFMA convolution.  ~5 % slower on the bench workload."""
    for mod in model.modules():
        if isinstance(mod, ConvModule):
            mod.deterministic = bool(flag)
        if hasattr(mod, 'deterministic_conv3x3'):
            mod.deterministic_conv3x3 = bool(flag)
    return model


@MMCV_FEEDFORWARD_NETWORK.register_module()
class FFN(BaseModule):
    """mmcv/cnn/bricks/transformer.py:1046-1120 (keys ``layers.0.0``, ``layers.1``)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        self.act_cfg = act_cfg
        self.activate = build_activation_layer(act_cfg)
        layers = []
        in_channels = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(Linear(in_channels, feedforward_channels), self.activate,
                                        nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    supports_post_norm = True

    def _fast_ok(self, x):
        return (self.num_fcs == 2 and self.add_identity and isinstance(self.activate, nn.ReLU)
                and _fusable(x) and x.dim() == 3)

    def forward(self, x, identity=None, post_norm=None, inplace_residual=False, carry=None,
                query_pos=None, out=None):
        """carry (dict with carry['emit'] = True) + query_pos: also produce out + query_pos in the
        LayerNorm pass and leave it in carry['q_plus'] for the next layer's attention."""
        if identity is None:
            identity = x
        if self._fast_ok(x):
            xb, ib = batch_first(x), batch_first(identity)
            fc1, fc2 = self.layers[0][0], self.layers[1]
            x2 = xb.reshape(-1, xb.shape[-1])
            if (_GEMM['mode'] == 'fp16' and FP16_ACTIVATIONS and x2.shape[0] >= 65536 and x2.is_contiguous()
                    and ib.is_contiguous() and post_norm is not None and fc2.out_features == 256
                    and fc1.out_features % 256 == 0 and fc1.in_features % 32 == 0 and fc1.in_features >= 64
                    and tuple(post_norm.normalized_shape) == (256,) and post_norm.weight is not None
                    and post_norm.bias is not None and not torch.is_grad_enabled()
                    and not (carry is not None and carry.get('emit'))):
                # fp16 mode: the hidden activation only ever feeds fc2's MFMA, which rounds it to fp16 at operand
                # fetch -- stored AS fp16 it is the same values at half the bytes of the two launches that are
                # bound by them (fc1 writes 4 x the layer input, fc2 reads it back)
                from . import ops
                h16 = ops.gemm_fp16_act(x2, _split_weight(fc1.weight), fc1.bias, relu=True, out_half=True)
                idt2 = ib.reshape(-1, 256)
                out = ops.gemm_fp16_act(h16, _split_weight(fc2.weight), fc2.bias, residual=idt2,
                                        ln=(post_norm.weight, post_norm.bias, post_norm.eps),
                                        out=idt2 if inplace_residual else None)
                return seq_first_view(out.view(ib.shape))
            h = linear_rows(x2, fc1.weight, fc1.bias, relu=True)
            pos_rows = None
            if carry is not None and carry.get('emit') and post_norm is not None \
                    and query_pos is not None and query_pos.dtype == torch.float32:
                pb = batch_first(query_pos)
                if pb.dim() == 3 and pb.stride(0) == 0 and pb[0].is_contiguous():
                    pos_rows = pb[0]                      # one [S, C] table shared by all frames
                elif pb.is_contiguous() and pb.shape == xb.shape:
                    pos_rows = pb.reshape(-1, pb.shape[-1])
            out = linear_residual_norm(h.view(xb.shape[0], xb.shape[1], -1), fc2, ib, post_norm,
                                       inplace=inplace_residual, pos_rows=pos_rows, out=out)
            if pos_rows is not None:
                out, plus = out
                out = seq_first_view(out)
                if plus is not None:
                    carry['q_plus'], carry['q_for'] = seq_first_view(plus), out
                return out
            return seq_first_view(out)
        if x.dim() == 3 and not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
            out = self.layers(x.transpose(0, 1)).transpose(0, 1)  # keep the token-major storage
        else:
            out = self.layers(x)
        if self.add_identity:
            out = identity + out
        if post_norm is not None:
            out = layer_norm_any_layout(post_norm, out)
        return out


@MMCV_ATTENTION.register_module()
class MultiheadAttention(BaseModule):
    """Wrapper of nn.MultiheadAttention, mmcv/cnn/bricks/transformer.py:406-551."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=dict(type='Dropout', drop_prob=0.), init_cfg=None,
                 batch_first=False, **kwargs):
        super().__init__(init_cfg)
        if 'dropout' in kwargs:
            attn_drop = kwargs.pop('dropout')
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Identity()
        self.dropout_layer = nn.Identity()

    supports_post_norm = True

    def _self_attention_split(self, x, pos, identity, post_norm):
        """The decoders' self-attention on this package's own kernels, three launches: ONE split
        GEMM for q | k | v -- (x + pos) W^T = x W^T + pos W^T, and the positional term of the
        decoders is a parameter-derived constant per query row ([L, E], the same for every
        sequence of the batch), so it rides the GEMM epilogue as a row-periodic table
        [pos W_qk^T + b_qk | b_v] cached per weights --, the scaled-dot-product core
        (pave_mha_core_f32: K and V of a head in LDS, quad-owned queries, online softmax), and
        out_proj + identity + LayerNorm as one launch of the LayerNorm-epilogue GEMM.
        Returns None when the shapes are not the ones this path takes."""
        from . import ops
        a = self.attn
        L, N, E = x.shape
        H = self.num_heads
        if E != 256 or E // H != 32 or pos is None or post_norm is None or L > 568:
            return None
        pb = pos.transpose(0, 1)                      # [N, L, E]
        if not (pb.stride(0) == 0 or N == 1) or pb.stride(2) != 1:
            return None                               # a per-sequence positional term
        pos_rows = pb[0]                              # [L, E] view of the embedding parameter
        base = pos_rows._base if pos_rows._base is not None else pos_rows
        w, b = a.in_proj_weight, a.in_proj_bias
        key = SourceKey((base, w, b), extra=(tuple(pos_rows.shape), tuple(pos_rows.stride()),
                                             pos_rows.storage_offset()))
        hit = self.__dict__.get('_pave_qkv')
        if hit is None or hit[0] != key:
            with torch.no_grad():
                t = pos_rows.double() @ w[:2 * E].double().t() + b[:2 * E].double()
                tab = torch.cat([t.float(), b[2 * E:].float().expand(L, E)], 1).contiguous()
            hit = self.__dict__['_pave_qkv'] = (key, tab)
        xb = batch_first(x).reshape(N * L, E)
        qkv, _ = ops.gemm_bf16x3_ex(xb, _split_weight(w, N * L), None, hit[1], residual_rows=L)
        o = ops.mha_core(qkv, N, L, H)
        t = ops.gemm_bf16x3_ln(o, _split_weight(a.out_proj.weight, N * L), a.out_proj.bias,
                               batch_first(identity).reshape(N * L, E), post_norm.weight,
                               post_norm.bias, post_norm.eps)
        return seq_first_view(t.view(N, L, E))

    def _self_attention_fast(self, x, pos, identity, post_norm):
        """Self-attention of the decoders on the device (query = key = x + pos, value = x, no
        masks), seq-first [L, N, E]: one add, the q|k and v projections as two GEMMs, fused SDPA,
        out_proj with the identity riding the GEMM, bias + LayerNorm as one pass -- 7 launches
        instead of ~16 through nn.MultiheadAttention.  Same arithmetic per element."""
        from . import ops
        a = self.attn
        L, N, E = x.shape
        H, d = self.num_heads, E // self.num_heads
        w, b = a.in_proj_weight, a.in_proj_bias
        # token-major rows of the batch-first storage the seq-first tensors are views of: the
        # projections, the residual GEMM and the LayerNorm then run without layout copies
        xb = batch_first(x).reshape(N * L, E)
        qk_in = xb + batch_first(pos).reshape(N * L, E) if pos is not None else xb
        qk = F.linear(qk_in, w[:2 * E], b[:2 * E]).view(N, L, 2, H, d)
        v = F.linear(xb, w[2 * E:], b[2 * E:]).view(N, L, H, d)
        o = F.scaled_dot_product_attention(qk[:, :, 0].transpose(1, 2), qk[:, :, 1].transpose(1, 2),
                                           v.transpose(1, 2))                        # [N, H, L, d]
        o = o.transpose(1, 2).reshape(N * L, E)
        t = torch.addmm(batch_first(identity).reshape(N * L, E), o, a.out_proj.weight.t())
        if post_norm is not None:
            t = ops.bias_add_layernorm(t, a.out_proj.bias, None, post_norm.weight, post_norm.bias,
                                       post_norm.eps)
        else:
            t = t + a.out_proj.bias
        return seq_first_view(t.view(N, L, E))

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, post_norm=None, **kwargs):
        a = self.attn
        # (train() under no_grad -- a validation pass without eval(), MC dropout -- keeps the
        # reference's attention dropout: the fast path has none)
        if (query.is_cuda and query.dtype == torch.float32 and not torch.is_grad_enabled()
                and (not self.training or a.dropout == 0)
                and not self.batch_first and query.dim() == 3 and attn_mask is None
                and key_padding_mask is None and (key is None or key is query)
                and (value is None or value is query)
                and (key_pos is None or key_pos is query_pos)
                and (query_pos is None or query_pos.shape == query.shape)
                and a._qkv_same_embed_dim and a.in_proj_bias is not None
                and a.bias_k is None and not a.add_zero_attn
                and (post_norm is None or (isinstance(post_norm, nn.LayerNorm)
                                           and post_norm.elementwise_affine))):
            if _GEMM['mode'] in _QMODES:
                out = self._self_attention_split(query, query_pos,
                                                 query if identity is None else identity, post_norm)
                if out is not None:
                    return out
            return self._self_attention_fast(query, query_pos,
                                             query if identity is None else identity, post_norm)
        out = self._forward_reference(query, key, value, identity, query_pos, key_pos, attn_mask,
                                      key_padding_mask)
        return layer_norm_any_layout(post_norm, out) if post_norm is not None else out

    def _forward_reference(self, query, key=None, value=None, identity=None, query_pos=None,
                           key_pos=None, attn_mask=None, key_padding_mask=None):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None:
            if query_pos is not None:
                if query_pos.shape == key.shape:
                    key_pos = query_pos
                else:
                    warnings.warn(f'position encoding of key is missing in '
                                  f'{self.__class__.__name__}.')
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask, need_weights=False)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + out


@MMCV_TRANSFORMER_LAYER.register_module()
class BaseTransformerLayer(BaseModule):
    """mmcv/cnn/bricks/transformer.py:1123-1353 incl. the fork's ``query_time_pos`` kwarg."""

    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024, num_fcs=2,
                               ffn_drop=0., act_cfg=dict(type='ReLU', inplace=True)),
                 operation_order=None, norm_cfg=dict(type='LN'), init_cfg=None,
                 batch_first=False, **kwargs):
        ffn_cfgs = copy.deepcopy(ffn_cfgs)
        deprecated_args = dict(feedforward_channels='feedforward_channels', ffn_dropout='ffn_drop',
                               ffn_num_fcs='num_fcs')
        for ori_name, new_name in deprecated_args.items():
            if ori_name in kwargs:
                ffn_cfgs[new_name] = kwargs[ori_name]
        super().__init__(init_cfg)
        self.batch_first = batch_first
        assert set(operation_order) & {'self_attn', 'norm', 'ffn', 'cross_attn'} == \
            set(operation_order)
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            attn_cfgs = [copy.deepcopy(dict(c)) for c in attn_cfgs]
            assert num_attn == len(attn_cfgs)
        self.num_attn = num_attn
        self.operation_order = tuple(operation_order)
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = nn.ModuleList()
        index = 0
        for operation_name in operation_order:
            if operation_name in ['self_attn', 'cross_attn']:
                if 'batch_first' in attn_cfgs[index]:
                    assert self.batch_first == attn_cfgs[index]['batch_first']
                else:
                    attn_cfgs[index]['batch_first'] = self.batch_first
                attention = build_attention(attn_cfgs[index])
                attention.operation_name = operation_name
                self.attentions.append(attention)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        num_ffns = operation_order.count('ffn')
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [copy.deepcopy(dict(ffn_cfgs)) for _ in range(num_ffns)]
        assert len(ffn_cfgs) == num_ffns
        for ffn_index in range(num_ffns):
            if 'embed_dims' not in ffn_cfgs[ffn_index]:
                ffn_cfgs[ffn_index]['embed_dims'] = self.embed_dims
            else:
                assert ffn_cfgs[ffn_index]['embed_dims'] == self.embed_dims
            self.ffns.append(build_feedforward_network(ffn_cfgs[ffn_index], dict(type='FFN')))
        self.norms = nn.ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(build_norm_layer(norm_cfg, self.embed_dims)[1])

    def forward(self, query, key=None, value=None, query_pos=None, query_time_pos=None,
                key_pos=None, attn_masks=None, query_key_padding_mask=None,
                key_padding_mask=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
        else:
            assert len(attn_masks) == self.num_attn
        order = self.operation_order
        skip_norm = False
        carry = kwargs.pop('fusion_carry', None)
        # layer_out: a dense batch-first [bs, n, C] buffer the layer's LAST launch may write its result into (the
        # decoders' preallocated stack of intermediate states: no torch.stack copy afterwards); a layer that does
        # not end in a fused FFN + LayerNorm ignores it -- the caller checks where the result lives
        layer_out = kwargs.pop('layer_out', None)
        for pos, layer in enumerate(order):
            # post-norm layers: hand the following LayerNorm to a module that can fuse it with
            # its own bias + residual epilogue (one pass instead of three)
            fuse = {}
            if (not self.pre_norm and pos + 1 < len(order) and order[pos + 1] == 'norm'
                    and layer != 'norm'):
                mod = self.ffns[ffn_index] if layer == 'ffn' else self.attentions[attn_index]
                if getattr(mod, 'supports_post_norm', False) and query.is_cuda:
                    fuse = dict(post_norm=self.norms[norm_index])
            if layer == 'self_attn':
                temp_key = temp_value = query
                plus = {}
                if carry is not None and carry.get('q_for') is query and \
                        getattr(self.attentions[attn_index], 'supports_query_plus_pos', False):
                    plus = dict(query_plus_pos=carry.pop('q_plus'))  # made by the previous LayerNorm
                    carry.pop('q_for')
                query = self.attentions[attn_index](
                    query, temp_key, temp_value, identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=query_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=query_key_padding_mask, **plus, **fuse, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'norm':
                if skip_norm:
                    skip_norm = False
                else:
                    query = layer_norm_any_layout(self.norms[norm_index], query)
                norm_index += 1
                continue
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    query_time_pos=query_time_pos, key_pos=key_pos,
                    attn_mask=attn_masks[attn_index], key_padding_mask=key_padding_mask,
                    **fuse, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'ffn':
                if kwargs.get('inplace_residual', False):
                    fuse = dict(fuse, inplace_residual=True)
                if carry is not None and fuse.get('post_norm') is not None and \
                        pos + 2 == len(order) and isinstance(self.ffns[ffn_index], FFN):
                    fuse = dict(fuse, carry=carry, query_pos=query_pos)
                if layer_out is not None and fuse.get('post_norm') is not None and pos + 2 == len(order) \
                        and isinstance(self.ffns[ffn_index], FFN):
                    fuse = dict(fuse, out=layer_out)
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None, **fuse)
                ffn_index += 1
            skip_norm = bool(fuse)
            if fuse:
                identity = query
        return query


@MMCV_TRANSFORMER_LAYER.register_module()
class DetrTransformerDecoderLayer(BaseTransformerLayer):
    """mmdet/models/utils/transformer.py:409-453."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'),
                 ffn_num_fcs=2, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order,
                         act_cfg=act_cfg, norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)


@MMCV_TRANSFORMER_LAYER_SEQUENCE.register_module()
class TransformerLayerSequence(BaseModule):
    """mmcv/cnn/bricks/transformer.py:1606-1687."""

    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        for i in range(num_layers):
            self.layers.append(build_transformer_layer(transformerlayers[i]))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm

    def forward(self, query, key, value, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        # layer i's closing LayerNorm can also write `out + query_pos`, which layer i+1's
        # self-attention starts with (one elementwise pass less per layer)
        carry = {} if (query.is_cuda and query_pos is not None and key is None
                       and not torch.is_grad_enabled() and FUSE_QUERY_POS) else None
        # input_is_shared: the caller still needs `query` afterwards (it aliases the neck output),
        # so the first attention must not accumulate its residual GEMM into it
        shared = kwargs.pop('input_is_shared', False)
        inplace = kwargs.get('inplace_residual', False)
        for i, layer in enumerate(self.layers):
            if shared and inplace:
                kwargs['inplace_residual'] = 'ffn_only' if i == 0 else inplace
            if carry is not None:
                carry['emit'] = i + 1 < len(self.layers)
                kwargs['fusion_carry'] = carry
            query = layer(query, key, value, query_pos=query_pos, key_pos=key_pos,
                          attn_masks=attn_masks, query_key_padding_mask=query_key_padding_mask,
                          key_padding_mask=key_padding_mask, **kwargs)
        return query


@MMCV_TRANSFORMER_LAYER_SEQUENCE.register_module()
class DetrTransformerEncoder(TransformerLayerSequence):
    """mmdet/models/utils/transformer.py:501-531."""

    def __init__(self, *args, post_norm_cfg=dict(type='LN'), **kwargs):
        super().__init__(*args, **kwargs)
        if post_norm_cfg is not None:
            self.post_norm = build_norm_layer(post_norm_cfg, self.embed_dims)[1] \
                if self.pre_norm else None
        else:
            assert not self.pre_norm
            self.post_norm = None

    def forward(self, *args, **kwargs):
        x = super().forward(*args, **kwargs)
        if self.post_norm is not None:
            x = layer_norm_any_layout(self.post_norm, x)
        return x


@MMCV_POSITIONAL_ENCODING.register_module()
class SinePositionalEncoding(BaseModule):
    """mmdet/models/utils/positional_encoding.py:11-93."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi,
                 eps=1e-6, offset=0., init_cfg=None):
        super().__init__(init_cfg)
        if normalize:
            assert isinstance(scale, (float, int))
        self.num_feats = num_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = scale
        self.eps = eps
        self.offset = offset

    def forward(self, mask):
        mask = mask.to(torch.int)
        not_mask = 1 - mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / (x_embed[:, :, -1:] + self.eps) * self.scale
        # dim_t on the host: fully-padded columns give arguments of ~1e6 rad (offset / eps),
        # where a 1-ulp difference in a device powf changes sin / cos completely
        dim_t = torch.arange(self.num_feats, dtype=torch.float32)
        dim_t = (self.temperature**(2 * (dim_t // 2) / self.num_feats)).to(mask.device)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        B, H, W = mask.size()
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()),
                            dim=4).view(B, H, W, -1)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()),
                            dim=4).view(B, H, W, -1)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-math.log((1 - prior_prob) / prior_prob))
