"""Torch-facing wrappers of the HIP kernels (device tensors in, device tensors out).

Mirrors the reference operator surface for this path:

* ``ms_deform_attn_forward`` -- same name and keyword names as the pybind export
  ``mmcv._ext.ms_deform_attn_forward``
  (third_party/mmcv/mmcv/ops/csrc/pytorch/pybind.cpp:737-741);
* ``MultiScaleDeformableAttnFunction`` -- same ``apply`` signature as
  third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py:20-57;
* ``deform_attn_grid_fused`` / ``deform_attn_pose_fused`` -- the fused T-frame
  launches that replace the per-frame softmax / location / sampling / fusion
  chain of MO:1388-1587 and OT:1644-1863.

PyTorch is used here only for device memory and the current HIP stream.  There
is no CPU path: a non-device tensor raises, exactly as the reference's
``AT_ASSERTM(value.is_cuda())`` does (ms_deform_attn_cuda.cu:221-230).
"""
import ctypes

import torch

from . import native


# When set to a list, launches whose tag is in KERNEL_EVENT_TAGS append (tag, start_event,
# end_event, flops) recorded on the stream they launch on (bench.py times the dominant kernel this way
# inside its timed region; every other launch records nothing).
KERNEL_EVENTS = None
KERNEL_EVENT_TAGS = ('enc_tile', 'enc_grid_T1')
KERNEL_EVENT_SHAPES = None   # tools/gemm_census.py: a list that receives the shape note of every recorded launch


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


class _Timed:
    def __init__(self, tag, flops=0, shape=None):
        self.tag = tag if (KERNEL_EVENTS is not None and tag in KERNEL_EVENT_TAGS) else None
        self.flops = flops
        if self.tag is not None and KERNEL_EVENT_SHAPES is not None:
            KERNEL_EVENT_SHAPES.append(shape)

    def __enter__(self):
        if self.tag is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if self.tag is not None:
            self.e.record()
            KERNEL_EVENTS.append((self.tag, self.s, self.e, self.flops))


SMALL_ROWS = 8192   # GEMM launches with fewer rows are latency-bound by shape (the decoders' few hundred
#                     query rows): timed under their own tag, reported beside the dominant class


def _gemm_tag(tag, M):
    return tag if M >= SMALL_ROWS else tag + '_small'


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _dev(t, name, dtype=None):
    _require(isinstance(t, torch.Tensor), f'{name} must be a tensor')
    _require(t.is_cuda, f'{name} must be a HIP device tensor (pavenet_amd has no CPU path)')
    _require(t.is_contiguous(), f'{name} tensor has to be contiguous')
    if dtype is not None:
        _require(t.dtype == dtype, f'{name} must be {dtype}, got {t.dtype}')
    return t


PLANES_FP16 = 16   # include/pave_hip.h PAVE_PLANES_FP16: one plane of fp16 operands


def _planes(w_planes, name='w_planes', fp16=False, only=None):
    """Checks a split-weight operand and returns the C ABI's `nplanes` for it: int16 tensors hold
    shape[1] bf16 planes, float16 tensors ONE plane of fp16 operands (split_weight_bf16x3(..., PLANES_FP16));
    `fp16=True` with an int16 single plane = the same bits under the old dtype.  only: allowed values."""
    # (one combined test on the way every launch takes; the messages are built on failure only)
    dt = w_planes.dtype if isinstance(w_planes, torch.Tensor) else None
    if not (dt in (torch.int16, torch.float16) and w_planes.is_cuda and w_planes.is_contiguous()
            and w_planes.dim() == 4 and w_planes.shape[3] == 16):
        _dev(w_planes, name)
        _require(dt in (torch.int16, torch.float16), f'{name} must be int16 (bf16 planes) or float16 (fp16 plane)')
        _require(False, f'{name}: [K/16, planes, N, 16] (split_weight_bf16x3)')
    n = w_planes.shape[1]
    if dt == torch.float16 or fp16:
        _require(n == 1, f'{name}: fp16 operands are a single plane')
        n = PLANES_FP16
    if only is not None and n not in only:
        _require(False, f'{name}: this entry point takes 3 bf16 planes or one fp16 plane')
    return n


_Q_PLANES = (3, PLANES_FP16)   # the modes of the LDS-DMA generation (every fused form)


def ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                           sampling_locations, attention_weights, im2col_step=64):
    """[R1] value [bs,S,M,D], shapes [L,2] i64, lsi [L] i64, loc [bs,Lq,M,L,P,2],
    weights [bs,Lq,M,L,P] -> [bs, Lq, M*D]."""
    lib = native.load()
    _require(value.dtype in (torch.float32, torch.float64),
             'ms_deform_attn_forward: value must be float32 or float64')
    dt = value.dtype
    _dev(value, 'value', dt)
    _dev(value_spatial_shapes, 'spatial_shapes', torch.int64)
    _dev(value_level_start_index, 'level_start_index', torch.int64)
    _dev(sampling_locations, 'sampling_loc', dt)
    _dev(attention_weights, 'attn_weight', dt)
    _require(value.dim() == 4 and sampling_locations.dim() == 6 and attention_weights.dim() == 5,
             'ms_deform_attn_forward: bad tensor ranks')
    bs, S, M, D = value.shape
    _, Lq, M2, L, P, two = sampling_locations.shape
    _require(M2 == M and two == 2 and sampling_locations.shape[0] == bs,
             'ms_deform_attn_forward: sampling_loc shape mismatch')
    _require(tuple(attention_weights.shape) == (bs, Lq, M, L, P),
             'ms_deform_attn_forward: attn_weight shape mismatch')
    _require(tuple(value_spatial_shapes.shape) == (L, 2) and
             tuple(value_level_start_index.shape) == (L,),
             'ms_deform_attn_forward: spatial_shapes / level_start_index shape mismatch')
    out = torch.empty((bs, Lq, M * D), dtype=dt, device=value.device)
    if out.numel() == 0:   # no queries (or empty batch): empty result, as the PyTorch formulation
        return out         # (MO:92-149) gives; the C entry point rejects non-positive sizes
    fn = (lib.pave_ms_deform_attn_forward_f32 if dt == torch.float32
          else lib.pave_ms_deform_attn_forward_f64)
    with torch.cuda.device(value.device):
        st = fn(value.data_ptr(), value_spatial_shapes.data_ptr(),
                value_level_start_index.data_ptr(), sampling_locations.data_ptr(),
                attention_weights.data_ptr(), out.data_ptr(), bs, S, M, D, L, Lq, P,
                int(im2col_step), _stream_ptr())
    native.check(st, 'ms_deform_attn_forward')
    return out


def ms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index,
                            sampling_locations, attention_weights, grad_output, grad_value,
                            grad_sampling_loc, grad_attn_weight, im2col_step=64):
    """Same positional / keyword surface as ``mmcv._ext.ms_deform_attn_backward``
    (pybind.cpp:743-748): grad_value is accumulated into, the other two are overwritten."""
    lib = native.load()
    dt = value.dtype
    _require(dt in (torch.float32, torch.float64), 'ms_deform_attn_backward: fp32 / fp64 only')
    for t, n in ((value, 'value'), (sampling_locations, 'sampling_loc'),
                 (attention_weights, 'attn_weight'), (grad_output, 'grad_output'),
                 (grad_value, 'grad_value'), (grad_sampling_loc, 'grad_sampling_loc'),
                 (grad_attn_weight, 'grad_attn_weight')):
        _dev(t, n, dt)
    _dev(value_spatial_shapes, 'spatial_shapes', torch.int64)
    _dev(value_level_start_index, 'level_start_index', torch.int64)
    bs, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    _require(grad_output.numel() == bs * Lq * M * D and grad_value.shape == value.shape and
             grad_sampling_loc.shape == sampling_locations.shape and
             grad_attn_weight.shape == attention_weights.shape,
             'ms_deform_attn_backward: gradient shapes mismatch')
    fn = (lib.pave_ms_deform_attn_backward_f32 if dt == torch.float32
          else lib.pave_ms_deform_attn_backward_f64)
    with torch.cuda.device(value.device):
        st = fn(value.data_ptr(), value_spatial_shapes.data_ptr(),
                value_level_start_index.data_ptr(), sampling_locations.data_ptr(),
                attention_weights.data_ptr(), grad_output.data_ptr(), grad_value.data_ptr(),
                grad_sampling_loc.data_ptr(), grad_attn_weight.data_ptr(), bs, S, M, D, L, Lq, P,
                int(im2col_step), _stream_ptr())
    native.check(st, 'ms_deform_attn_backward')


class MultiScaleDeformableAttnFunction(torch.autograd.Function):
    """Same ``apply(value, shapes, lsi, loc, weights, im2col_step)`` and gradients as MO:20-89."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        out = ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index,
            sampling_locations, attention_weights, im2col_step=im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        grad_value = torch.zeros_like(value)
        grad_loc = torch.zeros_like(loc)
        grad_aw = torch.zeros_like(aw)
        ms_deform_attn_backward(value, shapes, lsi, loc, aw, grad_output.contiguous(), grad_value,
                                grad_loc, grad_aw, im2col_step=ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_aw, None


def _level_rows(ref, L, name):
    """(dense tensor, ref_levels) of a reference tensor [.., L, c]: a level axis that is a broadcast
    (`reference_points[:, :, None].expand(..)`, stride 0: un-padded batches) is passed as ONE row per entry
    (ref_levels = 1, no materialised copy); anything else must be dense with L rows."""
    if ref.dim() >= 2 and ref.shape[-2] == L and L > 1 and ref.stride(-2) == 0:
        base = ref.select(-2, 0)
        if base.is_contiguous():
            return base, 1
        ref = ref.contiguous()
    _dev(ref, name, torch.float32)
    return ref, L


def deform_attn_grid_fused(value, spatial_shapes, level_start_index, proj, ref, *, T,
                           n_clips, units_per_clip, unit_clip=None, order=None,
                           return_stats=False, frame_table=None):
    """Fused grid-offset deformable attention over T frames ([R2] with T=1, [R4]).

    value [n_clips*T, S, 8, 32]; proj [n_units, >= T*8*16*3]; ref [T, n_units, 4, 2]
    -> out [n_units, 256] (+ (max, sum) [n_units, 8] if return_stats).
    Differentiable in value / proj / ref (pavenet_amd/fused_autograd.py).
    """
    if torch.is_grad_enabled() and not return_stats and \
            (value.requires_grad or proj.requires_grad or ref.requires_grad):
        from .fused_autograd import GridFusedFunction
        _require(frame_table is None, 'deform_attn_grid_fused: frame_table is an inference-only option')
        return GridFusedFunction.apply(value, spatial_shapes, level_start_index, proj, ref, T,
                                       n_clips, units_per_clip, unit_clip, order)
    lib = native.load()
    f32 = torch.float32
    _dev(value, 'value', f32)
    _dev(spatial_shapes, 'spatial_shapes', torch.int64)
    _dev(level_start_index, 'level_start_index', torch.int64)
    _dev(proj, 'proj', f32)
    _require(ref.is_cuda and ref.dtype == f32, 'deform_attn_grid_fused: ref fp32 on the device')
    _require(value.dim() == 4 and value.shape[2] == 8 and value.shape[3] == 32,
             'deform_attn_grid_fused: value must be [frames, S, 8, 32]')
    if frame_table is not None:   # value = per-frame cache, slab of (clip, t) = frame_table[clip*T + t]
        # (entries must be < value.shape[0]: checked where the table is built, streaming.decode; the
        # kernel clamps every slab index into the value tensor, so a bad table cannot fault)
        _dev(frame_table, 'frame_table', torch.int32)
        _require(frame_table.numel() == n_clips * T, 'deform_attn_grid_fused: frame_table [n_clips*T]')
    else:
        _require(value.shape[0] == n_clips * T, 'deform_attn_grid_fused: value frames != n_clips*T')
    S = value.shape[1]
    L = spatial_shapes.shape[0]
    _require(proj.dim() == 2, 'deform_attn_grid_fused: proj must be 2-D')
    n_units = proj.shape[0]
    _require(tuple(ref.shape) == (T, n_units, L, 2),
             f'deform_attn_grid_fused: ref must be [T, n_units, L, 2], got {tuple(ref.shape)}')
    ref, ref_levels = _level_rows(ref, L, 'ref')
    if unit_clip is not None:
        _dev(unit_clip, 'unit_clip', torch.int32)
        _require(unit_clip.numel() == n_units, 'deform_attn_grid_fused: unit_clip length')
    if order is not None:
        _dev(order, 'order', torch.int32)
        _require(order.numel() == n_units, 'deform_attn_grid_fused: order length')
    out = torch.empty((n_units, 256), dtype=f32, device=value.device)
    smax = ssum = None
    if return_stats:
        smax = torch.empty((n_units, 8), dtype=f32, device=value.device)
        ssum = torch.empty((n_units, 8), dtype=f32, device=value.device)
    tag = 'enc_grid_T1' if (T == 1 and unit_clip is None) else f'grid_T{T}'
    with torch.cuda.device(value.device), _Timed(tag):
        st = lib.pave_deform_attn_grid_fused_f32(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            proj.data_ptr(), ref.data_ptr(),
            unit_clip.data_ptr() if unit_clip is not None else None,
            order.data_ptr() if order is not None else None, out.data_ptr(),
            smax.data_ptr() if return_stats else None,
            ssum.data_ptr() if return_stats else None, n_units, int(units_per_clip),
            int(n_clips), int(T), S, L, 4, proj.stride(0),
            frame_table.data_ptr() if frame_table is not None else None, int(value.shape[0]), ref_levels,
            _stream_ptr())
    native.check(st, 'deform_attn_grid_fused')
    if return_stats:
        return out, smax, ssum
    return out


def deform_attn_pose_fused(value, spatial_shapes, level_start_index, proj, ref, *, T,
                           n_clips, num_query, num_keypoints, return_stats=False, frame_table=None):
    """Fused pose-aware deformable attention over T frames ([R3]).

    value [n_clips*T, S, 8, 32]; proj [n_clips*Q, >= T*8*L*K*3];
    ref [n_clips, T*Q, L, 2K] -> out [n_clips*Q, 256].
    Differentiable in value / proj / ref (pavenet_amd/fused_autograd.py).
    """
    if torch.is_grad_enabled() and not return_stats and \
            (value.requires_grad or proj.requires_grad or ref.requires_grad):
        from .fused_autograd import PoseFusedFunction
        _require(frame_table is None, 'deform_attn_pose_fused: frame_table is an inference-only option')
        return PoseFusedFunction.apply(value, spatial_shapes, level_start_index, proj, ref, T,
                                       n_clips, int(num_query), int(num_keypoints))
    lib = native.load()
    f32 = torch.float32
    _dev(value, 'value', f32)
    _dev(spatial_shapes, 'spatial_shapes', torch.int64)
    _dev(level_start_index, 'level_start_index', torch.int64)
    _dev(proj, 'proj', f32)
    _require(ref.is_cuda and ref.dtype == f32, 'deform_attn_pose_fused: ref fp32 on the device')
    _require(value.dim() == 4 and value.shape[2] == 8 and value.shape[3] == 32,
             'deform_attn_pose_fused: value must be [frames, S, 8, 32]')
    if frame_table is not None:
        _dev(frame_table, 'frame_table', torch.int32)
        _require(frame_table.numel() == n_clips * T, 'deform_attn_pose_fused: frame_table [n_clips*T]')
    else:
        _require(value.shape[0] == n_clips * T, 'deform_attn_pose_fused: value frames != n_clips*T')
    S = value.shape[1]
    L = spatial_shapes.shape[0]
    Q, K = int(num_query), int(num_keypoints)
    _require(proj.dim() == 2 and proj.shape[0] == n_clips * Q,
             'deform_attn_pose_fused: proj must be [n_clips*Q, cols]')
    _require(ref.numel() == n_clips * T * Q * L * 2 * K and ref.shape[-2:] == (L, 2 * K),
             'deform_attn_pose_fused: ref must be [n_clips, T*Q, L, 2K]')
    ref, ref_levels = _level_rows(ref, L, 'ref')
    out = torch.empty((n_clips * Q, 256), dtype=f32, device=value.device)
    smax = ssum = None
    if return_stats:
        smax = torch.empty((n_clips * Q, 8), dtype=f32, device=value.device)
        ssum = torch.empty((n_clips * Q, 8), dtype=f32, device=value.device)
    with torch.cuda.device(value.device), _Timed(f'pose_T{T}'):
        st = lib.pave_deform_attn_pose_fused_f32(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            proj.data_ptr(), ref.data_ptr(), out.data_ptr(),
            smax.data_ptr() if return_stats else None,
            ssum.data_ptr() if return_stats else None, int(n_clips), Q, int(T), S, L, K,
            proj.stride(0), frame_table.data_ptr() if frame_table is not None else None,
            int(value.shape[0]), ref_levels, _stream_ptr())
    native.check(st, 'deform_attn_pose_fused')
    if return_stats:
        return out, smax, ssum
    return out


def oks_nms(kpts, scores, sigmas, thresh):
    """Device OKS-NMS (HEAD:1624-1665).  kpts [n_clips, N, K, 3], scores [n_clips, N],
    sigmas [K] float64 -> (keep [n_clips, N] int32, order [n_clips, N] int32)."""
    lib = native.load()
    _dev(kpts, 'kpts', torch.float32)
    _dev(scores, 'scores', torch.float32)
    _dev(sigmas, 'sigmas', torch.float64)
    _require(kpts.dim() == 4 and kpts.shape[-1] == 3, 'oks_nms: kpts must be [n_clips, N, K, 3]')
    n_clips, N, K, _ = kpts.shape
    _require(tuple(scores.shape) == (n_clips, N) and sigmas.numel() == K,
             'oks_nms: scores / sigmas shape mismatch')
    keep = torch.empty((n_clips, N), dtype=torch.int32, device=kpts.device)
    order = torch.empty((n_clips, N), dtype=torch.int32, device=kpts.device)
    if keep.numel() == 0:  # nothing to suppress (the NumPy loop of HEAD:1624-1665 returns [])
        return keep, order
    with torch.cuda.device(kpts.device):
        st = lib.pave_oks_nms_f32(kpts.data_ptr(), scores.data_ptr(), sigmas.data_ptr(),
                                  float(thresh), keep.data_ptr(), order.data_ptr(), n_clips, N,
                                  K, _stream_ptr())
    native.check(st, 'oks_nms')
    return keep, order


def fuse_sum_nhwc(terms, relu=True):
    """HRNet fuse layer in one pass (pave_fuse_sum_nhwc_f32): terms = [(map, shift), ...] (1..4), map
    [N, C, H >> shift, W >> shift] fp32 channels_last; returns relu(sum of the maps, the coarser ones
    read through a nearest-neighbour up-sampling by 2^shift) [N, C, H, W] channels_last, summed in
    the order given (hrnet.py:197-214)."""
    lib = native.load()
    _require(1 <= len(terms) <= 4, 'fuse_sum_nhwc: 1..4 terms')
    t0, s0 = terms[0]
    N, C = t0.shape[0], t0.shape[1]
    H, W = t0.shape[2] << s0, t0.shape[3] << s0
    args = []
    for t, sh in terms:
        _require(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4
                 and t.is_contiguous(memory_format=torch.channels_last)
                 and tuple(t.shape) == (N, C, H >> sh, W >> sh) and (H >> sh) << sh == H
                 and (W >> sh) << sh == W,
                 'fuse_sum_nhwc: terms [N, C, H >> s, W >> s] fp32 channels_last')
        args += [t.data_ptr(), int(sh)]
    args += [None, 0] * (4 - len(terms))
    y = torch.empty((N, H, W, C), dtype=torch.float32, device=t0.device)
    with torch.cuda.device(t0.device), _Timed('fuse_sum'):
        st = lib.pave_fuse_sum_nhwc_f32(*args, y.data_ptr(), N, H, W, C, int(bool(relu)), _stream_ptr())
    native.check(st, 'fuse_sum_nhwc')
    return y.permute(0, 3, 1, 2)


def bias_act_rows_(x, bias=None, res=None, relu=True):
    """In place: x[r, c] = act(x[r, c] + bias[c] + res[r, c]) over the last (channel) dim.
    `x` must be dense with channels innermost (token matrix, or an NHWC / channels_last map)."""
    lib = native.load()
    _require(x.is_cuda and x.dtype == torch.float32, 'bias_act_rows_: fp32 device tensor')
    nhwc = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) \
        and not x.is_contiguous()
    C = bias.numel() if bias is not None else (x.shape[1] if nhwc else x.shape[-1])
    if nhwc:
        _require(x.shape[1] == C, 'bias_act_rows_: channel mismatch')
        if res is not None:
            _require(res.shape == x.shape and res.is_contiguous(memory_format=torch.channels_last),
                     'bias_act_rows_: residual layout mismatch')
    else:
        _require(x.is_contiguous() and x.shape[-1] == C, 'bias_act_rows_: channels must be innermost')
        if res is not None:
            _require(res.shape == x.shape and res.is_contiguous(), 'bias_act_rows_: residual layout')
    rows = x.numel() // C
    with torch.cuda.device(x.device), _Timed('bias_act_rows'):
        st = lib.pave_bias_act_rows_f32(
            x.data_ptr(), bias.data_ptr() if bias is not None else None,
            res.data_ptr() if res is not None else None, x.data_ptr(), rows, C, int(bool(relu)),
            _stream_ptr())
    native.check(st, 'bias_act_rows')
    return x


def fill_rows_(x, rows, values=None):
    """x[rows] = values (zeros when None), in place: x [R, C] fp32 (a row-strided view is fine), rows int32
    indices on the device, values [C] -- pave_fill_rows_f32."""
    lib = native.load()
    _require(x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1,
             'fill_rows_: x [R, C] fp32 on the device, unit column stride')
    _dev(rows, 'rows', torch.int32)
    if values is not None:
        _dev(values, 'values', torch.float32)
        _require(values.numel() == x.shape[1], 'fill_rows_: values [C]')
    with torch.cuda.device(x.device):
        st = lib.pave_fill_rows_f32(x.data_ptr(), x.stride(0), x.shape[0], rows.data_ptr(), rows.numel(),
                                    values.data_ptr() if values is not None else None, x.shape[1],
                                    _stream_ptr())
    native.check(st, 'fill_rows_')
    return x


def bias_add_layernorm(x, bias, res, gamma, beta, eps=1e-5, pos=None):
    """LayerNorm(x + bias + res) over the last dim; x / res dense row-major [..., C].
    With pos ([P, C] rows, P dividing the row count pattern r % P): returns (y, y + pos) from the
    same pass (the next layer's `query + query_pos`)."""
    lib = native.load()
    _dev(x, 'x', torch.float32)
    C = x.shape[-1]
    if res is not None:
        _require(res.shape == x.shape and res.is_contiguous() and res.dtype == torch.float32,
                 'bias_add_layernorm: residual must match x')
    out = torch.empty_like(x)
    rows = x.numel() // C
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    if pos is None:
        with torch.cuda.device(x.device), _Timed('bias_add_layernorm'):
            st = lib.pave_bias_add_layernorm_f32(x.data_ptr(), ptr(bias), ptr(res), gamma.data_ptr(),
                                                 beta.data_ptr(), out.data_ptr(), rows, C,
                                                 float(eps), _stream_ptr())
        native.check(st, 'bias_add_layernorm')
        return out
    _dev(pos, 'pos', torch.float32)
    _require(pos.shape[-1] == C and rows % (pos.numel() // C) == 0,
             'bias_add_layernorm: pos [P, C] with P dividing the number of rows')
    out_plus = torch.empty_like(x)
    with torch.cuda.device(x.device), _Timed('bias_add_layernorm'):
        st = lib.pave_bias_add_layernorm_pos_f32(
            x.data_ptr(), ptr(bias), ptr(res), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
            pos.data_ptr(), pos.numel() // C, out_plus.data_ptr(), rows, C, float(eps),
            _stream_ptr())
    native.check(st, 'bias_add_layernorm')
    return out, out_plus


def enc_tile_supported(levels_hw):
    """The LDS-tile encoder kernel covers 4-level halving pyramids: every token of level l sits
    in exactly one 8 x 8-pixel image tile (H_l <= (8 >> l) * ceil(H_0 / 8), same for W)."""
    if len(levels_hw) != 4:
        return False
    ny, nx = (int(levels_hw[0][0]) + 7) // 8, (int(levels_hw[0][1]) + 7) // 8
    return all(int(h) <= (8 >> l) * ny and int(w) <= (8 >> l) * nx
               for l, (h, w) in enumerate(levels_hw))


def enc_tile_window_shift(offset_bias):
    """Window shift table for `deform_attn_enc_tile` from a `sampling_offsets.bias` [8*4*4*2]: the
    mean offset of each (head, level) over its 4 points, rounded -> tuple of 64 ints (host values:
    one device read, cache it per bias version)."""
    m = offset_bias.detach().float().view(8, 4, 4, 2).mean(2).round().clamp_(-16, 16)
    return tuple(int(v) for v in m.flatten().tolist())


def gemm_bf16x3_encproj(a, w_planes, table, ref, levels_hw, value_bias=None, out=None):
    """The encoder layer's merged N = 640 projection with the sampler's softmax / location
    arithmetic in the GEMM epilogue (pave_gemm_bf16x3_encproj_f32): a [M, K], w_planes = 3-plane
    split of the [640, K] weight, table [rows, 640] (row m adds table[m % rows]), ref [M, 4, 2]
    -> (value [M, 256], samp [M, 384]): samp = level pixel coordinates + attention weights, the
    `prepared=True` input of deform_attn_enc_tile.  value_bias [256]: added to the value columns
    instead of table[:, :256] (not read then: value_proj has no positional term)."""
    lib = native.load()
    for t, nm in ((a, 'a'), (table, 'table'), (ref, 'ref')):
        _dev(t, nm, torch.float32)
    npl = _planes(w_planes, only=_Q_PLANES)
    M, K = a.shape
    _require(w_planes.shape[2] == 640 and w_planes.shape[0] * 16 == K,
             'gemm_bf16x3_encproj: w_planes [K/16, planes, 640, 16]')
    _require(table.dim() == 2 and table.shape[1] == 640 and ref.numel() == M * 8,
             'gemm_bf16x3_encproj: table [rows, 640], ref [M, 4, 2]')
    _require(len(levels_hw) == 4, 'gemm_bf16x3_encproj: four (h, w) levels')
    if value_bias is not None:
        _dev(value_bias, 'value_bias', torch.float32)
        _require(value_bias.numel() == 256 and value_bias.is_contiguous(), 'gemm_bf16x3_encproj: value_bias [256]')
    import ctypes
    hw_arr = (ctypes.c_int * 8)(*[int(v) for hw in levels_hw for v in hw])
    if out is not None:     # (value [M, 256], samp [M, 384]) given: row slices of larger dense matrices
        value, samp = out
        _dev(value, 'out value', torch.float32)
        _dev(samp, 'out samp', torch.float32)
        _require(tuple(value.shape) == (M, 256) and tuple(samp.shape) == (M, 384), 'gemm_bf16x3_encproj: out shapes')
    else:
        value = torch.empty((M, 256), dtype=torch.float32, device=a.device)
        samp = torch.empty((M, 384), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device), _Timed('gemm_bf16x3', 2 * M * K * 640, (M, K, 640, 'encproj')):
        st = lib.pave_gemm_bf16x3_encproj_f32(a.data_ptr(), w_planes.data_ptr(), table.data_ptr(),
                                              table.shape[0],
                                              value_bias.data_ptr() if value_bias is not None else None,
                                              ref.data_ptr(),
                                              ctypes.cast(hw_arr, ctypes.c_void_p), value.data_ptr(),
                                              samp.data_ptr(), M, K, npl, _stream_ptr())
    native.check(st, 'gemm_bf16x3_encproj')
    return value, samp


def deform_attn_enc_tile(value, proj, ref, *, levels_hw, variant=0, window_shift=None, prepared=False,
                         out_half=False):
    """Encoder deformable attention ([R2], T = 1) with LDS-staged value windows per image tile.
    value [F, S, 8, 32]; proj [F*S, >= 384]; ref [.., F*S, 4, 2] -> out [F*S, 256].
    Same results as ``deform_attn_grid_fused(..., T=1)``.  Differentiable (fused_autograd.py).
    window_shift: optional 64 host ints [8 heads][4 levels][(dx, dy)] (`enc_tile_window_shift`):
    moves each head's LDS window onto its mean sampling offset -- speed only, same results.
    prepared=True: `proj` comes from gemm_bf16x3_encproj (level pixel coordinates + attention
    weights instead of offsets + logits; ref is not read and may be None) -- same bits."""
    if not prepared and torch.is_grad_enabled() and \
            (value.requires_grad or proj.requires_grad or ref.requires_grad):
        from .fused_autograd import EncTileFunction
        return EncTileFunction.apply(value, proj, ref, levels_hw, variant)
    lib = native.load()
    f32 = torch.float32
    _dev(value, 'value', f32)
    _dev(proj, 'proj', f32)
    if prepared:
        _require(variant == 0, 'deform_attn_enc_tile: prepared input with the default windows only')
        variant = 4
    if out_half:     # fp16 output rows (fp16 operand mode: they only feed output_proj's MFMA)
        variant |= 8
    if ref is not None or not prepared:
        _dev(ref, 'ref', f32)
    _require(value.dim() == 4 and value.shape[2] == 8 and value.shape[3] == 32,
             'deform_attn_enc_tile: value must be [frames, S, 8, 32]')
    F_, S = value.shape[0], value.shape[1]
    _require(proj.dim() == 2 and proj.shape[0] == F_ * S, 'deform_attn_enc_tile: proj rows')
    _require(ref is None or ref.numel() == F_ * S * 8, 'deform_attn_enc_tile: ref must be [F*S, 4, 2]')
    _require(enc_tile_supported(levels_hw), 'deform_attn_enc_tile: needs a 4-level halving pyramid')
    import ctypes
    flat = [int(v) for hw in levels_hw for v in hw]
    hw_arr = (ctypes.c_int * 8)(*flat)
    sh_arr = None
    if window_shift is not None:
        _require(len(window_shift) == 64, 'deform_attn_enc_tile: window_shift has 64 entries')
        sh_arr = (ctypes.c_int * 64)(*[int(v) for v in window_shift])
    out = torch.empty((F_ * S, 256), dtype=torch.float16 if out_half else f32, device=value.device)
    with torch.cuda.device(value.device), _Timed('enc_tile'):
        st = lib.pave_enc_deform_attn_tile_f32(
            value.data_ptr(), proj.data_ptr(), ref.data_ptr() if ref is not None else None,
            out.data_ptr(), F_, S, ctypes.cast(hw_arr, ctypes.c_void_p), proj.stride(0), int(variant),
            ctypes.cast(sh_arr, ctypes.c_void_p) if sh_arr is not None else None, _stream_ptr())
    native.check(st, 'deform_attn_enc_tile')
    return out


def conv3x3_nhwc(x, w_taps, bias, stride=1, relu=False):
    """3x3 / pad 1 convolution of a channels_last map on the fp32 MFMA, bias (+ReLU) fused.
    x [N, Cin, H, W] in channels_last storage; w_taps [3, 3, Cin, Cout] contiguous;
    -> [N, Cout, Ho, Wo] channels_last."""
    lib = native.load()
    _require(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4, 'conv3x3_nhwc: fp32 4-D')
    _require(x.is_contiguous(memory_format=torch.channels_last), 'conv3x3_nhwc: channels_last input')
    _dev(w_taps, 'w_taps', torch.float32)
    N, Cin, H, W = x.shape
    _require(tuple(w_taps.shape[:3]) == (3, 3, Cin), 'conv3x3_nhwc: weight must be [3,3,Cin,Cout]')
    Cout = w_taps.shape[3]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _Timed('conv3x3'):
        st = lib.pave_conv3x3_nhwc_f32(x.data_ptr(), w_taps.data_ptr(),
                                       bias.data_ptr() if bias is not None else None,
                                       y.data_ptr(), N, H, W, Cin, Cout, int(stride),
                                       int(bool(relu)), _stream_ptr())
    native.check(st, 'conv3x3_nhwc')
    return y.permute(0, 3, 1, 2)


def rows_gemm_bias_res_act(a, w_kn, bias=None, residual=None, relu=False, out=None, a_bias=None,
                           a2=None):
    """out[M, N] = act([A1 | a2] @ w_kn + bias + residual) on the fp32 MFMA in one pass, with
    A1 = relu(a + a_bias) if a_bias is given else a  (the Bottleneck tail `bn2 -> relu -> conv3 ->
    bn3 -> += identity | downsample(x) -> relu`, resnet.py:264-283).  w_kn is [K (+ K2), N].
    `residual` may be the tensor given as `out` (in-place accumulate into the identity)."""
    lib = native.load()
    _dev(a, 'a', torch.float32)
    _dev(w_kn, 'w_kn', torch.float32)
    _require(a.dim() == 2 and w_kn.dim() == 2, 'rows_gemm: a [M,K], w [K,N]')
    M, K = a.shape
    K2 = 0
    if a2 is not None:
        _dev(a2, 'a2', torch.float32)
        _require(a2.dim() == 2 and a2.shape[0] == M, 'rows_gemm: a2 [M,K2]')
        K2 = a2.shape[1]
    _require(w_kn.shape[0] == K + K2, 'rows_gemm: w must be [K + K2, N]')
    N = w_kn.shape[1]
    for t, nm, n in ((bias, 'bias', N), (a_bias, 'a_bias', K)):
        if t is not None:
            _dev(t, nm, torch.float32)
            _require(t.numel() == n, f'rows_gemm: {nm} has {t.numel()} elements, expected {n}')
    if residual is not None:
        _dev(residual, 'residual', torch.float32)
        _require(tuple(residual.shape) == (M, N), 'rows_gemm: residual [M,N]')
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    else:
        _dev(out, 'out', torch.float32)
        _require(tuple(out.shape) == (M, N), 'rows_gemm: out [M,N]')
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    with torch.cuda.device(a.device), _Timed('rows_gemm'):
        st = lib.pave_rows_gemm_bias_res_act_f32(
            a.data_ptr(), ptr(a_bias), ptr(a2), w_kn.data_ptr(), ptr(bias), ptr(residual),
            out.data_ptr(), M, K, K2, N, int(bool(relu)), _stream_ptr())
    native.check(st, 'rows_gemm_bias_res_act')
    return out


def bias_relu_maxpool_nhwc(x, bias):
    """maxpool3x3/s2/p1(relu(x + bias)) of a channels_last map in one pass (ResNet stem tail)."""
    lib = native.load()
    _require(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4, 'bias_relu_maxpool: fp32 4-D')
    _require(x.is_contiguous(memory_format=torch.channels_last),
             'bias_relu_maxpool: channels_last input')
    _dev(bias, 'bias', torch.float32)
    N, C, H, W = x.shape
    _require(bias.numel() == C, 'bias_relu_maxpool: bias [C]')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _Timed('stem_pool'):
        st = lib.pave_bias_relu_maxpool_nhwc_f32(x.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                 N, H, W, C, _stream_ptr())
    native.check(st, 'bias_relu_maxpool_nhwc')
    return y.permute(0, 3, 1, 2)


def split_bf16x3(x, planes=3):
    """fp32 tensor -> int16 tensor [planes, *x.shape] of bf16 bit patterns: truncation terms, the
    last rounded to nearest (planes = 3: x == p0 + p1 + p2 exactly).  planes = PLANES_FP16: one
    plane of fp16 bit patterns."""
    lib = native.load()
    _dev(x, 'x', torch.float32)
    _require(planes in (1, 2, 3, PLANES_FP16), 'split_bf16x3: planes must be 1, 2, 3 or PLANES_FP16')
    n_out = 1 if planes == PLANES_FP16 else planes
    out = torch.empty((n_out,) + tuple(x.shape), dtype=torch.int16, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.pave_split_bf16x3_f32(x.data_ptr(), out.data_ptr(), x.numel(), planes,
                                       _stream_ptr())
    native.check(st, 'split_bf16x3')
    return out.view(torch.float16) if planes == PLANES_FP16 else out


def split_weight_bf16x3(weight, planes=3, pad=False):
    """nn.Linear weight [N, K] -> the W operand of `gemm_bf16x3`: int16 [K/16, planes, N, 16],
    i.e. the bf16 planes cut into 16-wide K slabs, slab-major, so that one slab of a column tile
    is contiguous in memory (every 128-byte line is fetched once).  pad=True (3-plane kernels):
    zero rows / columns up to N % 64 == 0, K % 32 == 0 (the kernels store the real N columns)."""
    _require(weight.dim() == 2, 'split_weight_bf16x3: weight [N, K]')
    N, K = weight.shape
    if pad and (N % 64 or K % 32):
        Np, Kp = (N + 63) // 64 * 64, (K + 31) // 32 * 32
        wpad = torch.zeros((Np, Kp), dtype=weight.dtype, device=weight.device)
        wpad[:N, :K] = weight
        weight, N, K = wpad, Np, Kp
    _require((K % 64 == 0 and N % 64 == 0) or (pad and K % 32 == 0 and N % 64 == 0),
             'split_weight_bf16x3: weight [N % 64 == 0, K % 64 == 0] (pad=True: any N, K)')
    pl = split_bf16x3(weight.contiguous(), planes)
    return pl.view(pl.shape[0], N, K // 16, 16).permute(2, 0, 1, 3).contiguous()


BLOCK_SLOTS = 512         # resident GEMM blocks of the chip: 256 CUs x 2 blocks of the wide / LayerNorm forms
ROUND_SPLIT = True        # A/B switch of round_split_rows (tools/ab_switch.py)


def round_split_rows(M, tiles_per_row_tile):
    """Block-slot quantisation: a launch whose tile count lands just above a multiple of the chip's 512 block slots
    runs a whole extra round for a handful of tiles (3 x 22 323 rows = 524 row tiles: every LayerNorm-epilogue launch
    of a one-clip T = 3 step ran a second round of 12 tiles -- 227 us where one round takes 110).  Returns the row
    count M1 of the FULL rounds when the last round would be less than an eighth full and its rows (at most 2 048)
    fit the small-row forms -- the caller launches rows [0, M1) and [M1, M) separately (GEMM rows are independent:
    the same values up to the summation order of the form each part takes) -- else None."""
    if not ROUND_SPLIT:
        return None
    rt = (M + 127) // 128
    per_round = BLOCK_SLOTS // max(1, tiles_per_row_tile)      # row tiles per round
    if per_round < 1 or rt <= per_round:
        return None
    r = rt % per_round
    if r == 0 or r * tiles_per_row_tile > BLOCK_SLOTS // 8:
        return None
    M1 = (rt - r) * 128
    return M1 if 0 < M - M1 <= 2048 else None


_ACT_CODE = {'gelu': 2, 'sigmoid': 3}     # pave_gemm_bf16x3_f32's `relu` argument beyond 0 / 1
SPLITK_ROWS = True     # plain row GEMMs with a split-K plan take pave_gemm_bf16x3_splitk_f32 (tools: A/B switch)


def gemm_bf16x3(a, w_planes, bias=None, residual=None, relu=False, out=None, a_bias=None,
                fp16=False, n_out=None):
    """out[M, N] = act(A' @ W^T + bias + residual), A' = relu(a + a_bias) if a_bias is given, on
    the bf16 matrix cores with both operands split into P = w_planes.shape[1] bf16 terms
    (relu: False | True | 'gelu' = exact GELU | 'sigmoid'; the last two: 3 planes / fp16 only)
    (P = 3: exact split, 6 MFMA products, fp32-level accuracy;  2: 3 products, ~2^-16;
    1: plain bf16 operands, or fp16 operands with fp16=True and a PLANES_FP16 weight), fp32
    accumulate, fp32 in / out.
    w_planes = split_weight_bf16x3(weight [N, K]).  `residual` may be the tensor given as `out`.
    n_out (3 planes): the real output width when the planes were zero-padded to N % 64 == 0
    (split_weight_bf16x3(..., pad=True)); out / bias / residual then have n_out columns."""
    lib = native.load()
    _dev(a, 'a', torch.float32)
    npl = _planes(w_planes, fp16=fp16)
    _require(a.dim() == 2 and w_planes.shape[1] in (1, 2, 3) and w_planes.shape[0] * 16 == a.shape[1],
             'gemm_bf16x3: a [M,K], w_planes [K/16,3,N,16] (split_weight_bf16x3)')
    M, K = a.shape
    N = w_planes.shape[2]
    if n_out is not None and n_out != N:
        _require(npl in _Q_PLANES and n_out % 4 == 0 and (n_out + 63) // 64 * 64 == N,
                 'gemm_bf16x3: n_out % 4 == 0 with 3 planes / fp16 padded to roundup(n_out, 64) rows')
        N = int(n_out)
    else:
        _require(N % 128 == 0 or (N % 64 == 0 and npl in _Q_PLANES),
                 'gemm_bf16x3: N % 128 == 0 (or N % 64 == 0 with 3 planes / fp16)')
    for t, nm, n in ((bias, 'bias', N), (a_bias, 'a_bias', K)):
        if t is not None:
            _dev(t, nm, torch.float32)
            _require(t.numel() == n, f'gemm_bf16x3: {nm} has {t.numel()} elements, expected {n}')
    if residual is not None:
        _dev(residual, 'residual', torch.float32)
        _require(tuple(residual.shape) == (M, N), 'gemm_bf16x3: residual [M,N]')
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    else:
        _dev(out, 'out', torch.float32)
        _require(tuple(out.shape) == (M, N), 'gemm_bf16x3: out [M,N]')
    if N % 256 == 0 and npl in _Q_PLANES and a_bias is None and N == w_planes.shape[2]:
        M1 = round_split_rows(M, N // 256)          # (the wide form's tiles: N / 256 per row tile)
        if M1 is not None:
            gemm_bf16x3(a[:M1], w_planes, bias, residual[:M1] if residual is not None else None, relu=relu,
                        out=out[:M1], fp16=fp16)
            gemm_bf16x3(a[M1:], w_planes, bias, residual[M1:] if residual is not None else None, relu=relu,
                        out=out[M1:], fp16=fp16)
            return out
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    # few row tiles and a long K (a one-clip batch's layer4 1x1 reductions, the neck's C5 lateral): the split-K
    # form, as for the 3x3 convolutions (workspace from torch's stream-aware caching allocator)
    # (a plan exists only for K >= 2048 on fewer than 200 row tiles -- pave_internal_splitk_plan --, so the ~170
    # K = 256 / 1024 launches of the decoder tail do not pay a library call to learn that)
    ws_bytes = lib.pave_gemm_splitk_workspace_bytes(M, K, N) \
        if (SPLITK_ROWS and K >= 2048 and M < 200 * 128 and npl in _Q_PLANES and a_bias is None
            and relu in (False, True, 0, 1)) else 0
    if ws_bytes > 0:
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device), _Timed(_gemm_tag('gemm_bf16x3', M), 2 * M * K * N, (M, K, N, 'relu' if relu else '', 'res' if residual is not None else '', 'split-K')):
            st = lib.pave_gemm_bf16x3_splitk_f32(a.data_ptr(), w_planes.data_ptr(), ptr(bias), ptr(residual),
                                                 out.data_ptr(), M, K, N, int(bool(relu)), npl, ws.data_ptr(),
                                                 ws_bytes, _stream_ptr())
        native.check(st, 'gemm_bf16x3_splitk')
        return out
    with torch.cuda.device(a.device), _Timed(_gemm_tag('gemm_bf16x3', M), 2 * M * K * N, (M, K, N, 'relu' if relu else '', 'res' if residual is not None else '', 'abias' if a_bias is not None else '')):
        st = lib.pave_gemm_bf16x3_f32(a.data_ptr(), ptr(a_bias), w_planes.data_ptr(), ptr(bias),
                                      ptr(residual), out.data_ptr(), M, K, N,
                                      _ACT_CODE.get(relu, int(bool(relu))), npl, _stream_ptr())
    native.check(st, 'gemm_bf16x3')
    return out


def gemm_bf16x3_cat(a, a2, w_planes, bias=None, residual=None, relu=False, out=None):
    """out[M, N] = act([a | a2] @ W^T + bias + residual): two row matrices share the K axis and one
    accumulator (the Bottleneck tail with a stride-1 downsample: conv3 and the downsample
    convolution in one launch, no concatenation copy).  w_planes = split_weight_bf16x3 of the
    [N, K1 + K2] row-concatenated weight."""
    lib = native.load()
    _dev(a, 'a', torch.float32)
    _dev(a2, 'a2', torch.float32)
    npl = _planes(w_planes, only=_Q_PLANES)
    _require(a.dim() == 2 and a2.dim() == 2 and a.shape[0] == a2.shape[0],
             'gemm_bf16x3_cat: a [M, K1], a2 [M, K2]')
    M, K1 = a.shape
    K = K1 + a2.shape[1]
    _require(w_planes.shape[0] * 16 == K, 'gemm_bf16x3_cat: w_planes [(K1+K2)/16, 3, N, 16]')
    N = w_planes.shape[2]
    _require(K % 32 == 0 and N % 64 == 0 and K1 % 16 == 0, 'gemm_bf16x3_cat: K % 32, N % 64, K1 % 16')
    if bias is not None:
        _dev(bias, 'bias', torch.float32)
        _require(bias.numel() == N, 'gemm_bf16x3_cat: bias [N]')
    if residual is not None:
        _dev(residual, 'residual', torch.float32)
        _require(tuple(residual.shape) == (M, N), 'gemm_bf16x3_cat: residual [M, N]')
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    else:
        _dev(out, 'out', torch.float32)
        _require(tuple(out.shape) == (M, N), 'gemm_bf16x3_cat: out [M, N]')
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    with torch.cuda.device(a.device), _Timed(_gemm_tag('gemm_bf16x3', M), 2 * M * K * N, (M, K, N, 'cat', f'k1={K1}')):
        st = lib.pave_gemm_bf16x3_cat_f32(a.data_ptr(), K1, a2.data_ptr(), w_planes.data_ptr(),
                                          ptr(bias), ptr(residual), out.data_ptr(), M, K, N,
                                          int(bool(relu)), npl, _stream_ptr())
    native.check(st, 'gemm_bf16x3_cat')
    return out


def gemm_bf16x3_grouped(a, w_planes, bias, group_n, relu=False):
    """Grouped row GEMM: a [M, G*K] (group i = columns [i K, (i+1) K)), w_planes = planes of the
    [G*group_n, K] row-concatenated per-group weights -> out [M, G*group_n] with
    out[:, i*group_n:(i+1)*group_n] = act(a_i @ W_i^T + bias_i).  One launch for G Linears."""
    lib = native.load()
    _dev(a, 'a', torch.float32)
    npl = _planes(w_planes, only=_Q_PLANES)
    _require(a.dim() == 2, 'gemm_bf16x3_grouped: a [M, G*K], w_planes [K/16, 3, N, 16]')
    M, lda = a.shape
    K, N = w_planes.shape[0] * 16, w_planes.shape[2]
    G = N // int(group_n)
    _require(G * group_n == N and G * K == lda, 'gemm_bf16x3_grouped: a has G*K columns, N = G*group_n')
    if bias is not None:
        _dev(bias, 'bias', torch.float32)
        _require(bias.numel() == N, 'gemm_bf16x3_grouped: bias [N]')
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device), _Timed(_gemm_tag('gemm_bf16x3', M), 2 * M * K * N, (M, K, N, 'grouped', f'g{G}')):
        st = lib.pave_gemm_bf16x3_grouped_f32(a.data_ptr(), lda, w_planes.data_ptr(),
                                              bias.data_ptr() if bias is not None else None,
                                              out.data_ptr(), M, K, N, int(group_n), int(bool(relu)),
                                              npl, _stream_ptr())
    native.check(st, 'gemm_bf16x3_grouped')
    return out


def conv1x1_strided_split(x, w_planes, bias=None, stride=2, relu=False):
    """1x1 convolution with a stride on a channels_last map through the 3-plane split GEMM (the A
    rows are the strided input pixels: no slice copy).  x [N, Cin, H, W] channels_last;
    w_planes = split_weight_bf16x3(weight [Cout, Cin]) -> [N, Cout, Ho, Wo] channels_last."""
    lib = native.load()
    _require(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4, 'conv1x1_strided_split: fp32 4-D')
    _require(x.is_contiguous(memory_format=torch.channels_last),
             'conv1x1_strided_split: channels_last input')
    npl = _planes(w_planes, only=_Q_PLANES)
    N, Cin, H, W = x.shape
    _require(w_planes.shape[0] * 16 == Cin, 'conv1x1_strided_split: w_planes [Cin/16, 3, Cout, 16]')
    Cout = w_planes.shape[2]
    if bias is not None:
        _dev(bias, 'bias', torch.float32)
        _require(bias.numel() == Cout, 'conv1x1_strided_split: bias [Cout]')
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _Timed('conv1x1_strided', 2 * N * Ho * Wo * Cin * Cout, (N * Ho * Wo, Cin, Cout, f's{stride}')):
        st = lib.pave_conv1x1_strided_split_f32(
            x.data_ptr(), w_planes.data_ptr(), bias.data_ptr() if bias is not None else None,
            y.data_ptr(), N, H, W, Cin, Cout, int(stride), int(bool(relu)), npl, _stream_ptr())
    native.check(st, 'conv1x1_strided_split')
    return y.permute(0, 3, 1, 2)


def conv3x3s2_c3_nchw(x, w_taps, bias=None, relu=False):
    """HRNet stem conv1: 3x3 / stride 2 / pad 1, 3 -> 64 channels, read straight from the NCHW batch
    x [N, 3, H, W] (contiguous) -> [N, 64, Ho, Wo] channels_last; w_taps [27, 64] =
    weight.permute(1, 2, 3, 0).reshape(27, 64) (pave_conv3x3s2_c3_nchw_f32)."""
    lib = native.load()
    _dev(x, 'x', torch.float32)
    _dev(w_taps, 'w_taps', torch.float32)
    _require(x.dim() == 4 and x.shape[1] == 3 and tuple(w_taps.shape) == (27, 64),
             'conv3x3s2_c3_nchw: x [N, 3, H, W], w_taps [27, 64]')
    if bias is not None:
        _dev(bias, 'bias', torch.float32)
        _require(bias.numel() == 64, 'conv3x3s2_c3_nchw: bias [64]')
    N, _, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, Ho, Wo, 64), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.pave_conv3x3s2_c3_nchw_f32(x.data_ptr(), w_taps.data_ptr(),
                                            bias.data_ptr() if bias is not None else None,
                                            y.data_ptr(), N, H, W, int(bool(relu)), _stream_ptr())
    native.check(st, 'conv3x3s2_c3_nchw')
    return y.permute(0, 3, 1, 2)


def split_stem7x7_weight(weight, planes=3):
    """Stem weight [64, 3, 7, 7] -> operand of conv7x7s2_nchw_split, int16 [23, 3, 64, 16]: the 3
    bf16 planes in two K layouts (include/pave_hip.h): 12 slabs of (c, ky, kx | pad) (K = 192)
    followed by 11 slabs of (c, ky, kx + 1) with a zero 22nd row (K = 176).  planes = PLANES_FP16:
    float16 [23, 1, 64, 16], one plane of fp16 operands in the same two layouts."""
    _require(tuple(weight.shape[1:]) == (3, 7, 7), 'split_stem7x7_weight: weight [Cout, 3, 7, 7]')
    _require(planes in _Q_PLANES, 'split_stem7x7_weight: 3 bf16 planes or PLANES_FP16')
    co = weight.shape[0]
    w = torch.zeros((co, 24, 8), dtype=torch.float32, device=weight.device)
    w[:, :21, :7] = weight.reshape(co, 21, 7)
    a = split_weight_bf16x3(w.reshape(co, 192).contiguous(), planes)
    w2 = torch.zeros((co, 22, 8), dtype=torch.float32, device=weight.device)
    w2[:, :21, 1:] = weight.reshape(co, 21, 7)
    pl = split_bf16x3(w2.reshape(co, 176).contiguous(), planes)          # [planes, co, 176]
    b = pl.view(pl.shape[0], co, 11, 16).permute(2, 0, 1, 3).contiguous()
    return torch.cat([a, b], 0)


def repitch_rows(x):
    """x [..., W] dense fp32 -> the same rows at a pitch of roundup(W, 4) elements with zero pad columns
    (pave_repitch_rows_f32): 16-byte aligned rows for the LDS-DMA stem when W % 4 != 0 (the 750 x 1333 frames of
    the reference's PoseTrack pipeline).  Returns the padded tensor [..., roundup(W, 4)]."""
    lib = native.load()
    _dev(x, 'x', torch.float32)
    W = x.shape[-1]
    pitch = (W + 3) // 4 * 4
    out = torch.empty(tuple(x.shape[:-1]) + (pitch,), dtype=torch.float32, device=x.device)
    rows = x.numel() // W
    with torch.cuda.device(x.device):
        st = lib.pave_repitch_rows_f32(x.data_ptr(), out.data_ptr(), rows, W, pitch, _stream_ptr())
    native.check(st, 'repitch_rows')
    return out


def conv7x7s2_nchw_split(x, w_planes, bias=None, relu=False, valid_w=None):
    """7x7 / stride 2 / pad 3 stem convolution of the NCHW image batch x [N, 3, H, W] through the
    3-plane split kernel -> [N, 64, Ho, Wo] channels_last.  valid_w: x is a `repitch_rows` result whose real
    width is valid_w (columns valid_w .. W - 1 are zero): the output is that of the valid_w-wide image."""
    lib = native.load()
    _dev(x, 'x', torch.float32)
    npl = _planes(w_planes, only=_Q_PLANES)
    _require(x.dim() == 4 and x.shape[1] == 3, 'conv7x7s2_nchw_split: x [N, 3, H, W] (NCHW, dense)')
    _require(w_planes.shape[0] == 23, 'conv7x7s2_nchw_split: w_planes from split_stem7x7_weight')
    N, _, H, pitch = x.shape
    W = pitch if valid_w is None else int(valid_w)
    _require(0 < W <= pitch and (W == pitch or pitch % 4 == 0), 'conv7x7s2_nchw_split: valid_w <= W, W % 4 == 0')
    Cout = w_planes.shape[2]
    if bias is not None:
        _dev(bias, 'bias', torch.float32)
        _require(bias.numel() == Cout, 'conv7x7s2_nchw_split: bias [Cout]')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _Timed('conv7x7_stem', 2 * N * Ho * Wo * Cout * 147, (N * Ho * Wo, 147, Cout)):
        st = lib.pave_conv7x7s2_nchw_split_f32(
            x.data_ptr(), w_planes.data_ptr(), bias.data_ptr() if bias is not None else None,
            y.data_ptr(), N, H, W, pitch, Cout, int(bool(relu)), npl, _stream_ptr())
    native.check(st, 'conv7x7s2_nchw_split')
    return y.permute(0, 3, 1, 2)


def ref_update(tmp, ref, eps=1e-5):
    """sigmoid(tmp + inverse_sigmoid(ref)) in one launch (decoder reference-point update)."""
    lib = native.load()
    _require(tmp.is_cuda and tmp.dtype == torch.float32 and ref.dtype == torch.float32
             and tuple(tmp.shape) == tuple(ref.shape), 'ref_update: fp32 device tensors, same shape')
    if (tmp.dim() >= 2 and not tmp.is_contiguous() and tmp.stride(-1) == 1 and ref.is_contiguous()
            and tmp.numel() > 0 and tmp.stride(-2) >= tmp.shape[-1]
            and all(tmp.stride(i) == tmp.stride(i + 1) * tmp.shape[i + 1] for i in range(tmp.dim() - 2))):
        # a column slice of a wider matrix (a branch output padded to the 4-column grid): read in place
        # by the strided form of the same kernel (T = 1: rows [R, ld], the first o columns)
        R, o = tmp.numel() // tmp.shape[-1], tmp.shape[-1]
        out = torch.empty_like(ref)
        with torch.cuda.device(tmp.device), _Timed('ref_update'):
            st = lib.pave_ref_update_frames_f32(tmp.data_ptr(), ref.data_ptr(), out.data_ptr(), R, 1,
                                                tmp.stride(-2), o, R, float(eps), _stream_ptr())
        native.check(st, 'ref_update')
        return out
    tmp, ref = tmp.contiguous(), ref.contiguous()
    out = torch.empty_like(tmp)
    if tmp.numel() == 0:
        return out
    with torch.cuda.device(tmp.device), _Timed('ref_update'):
        st = lib.pave_ref_update_f32(tmp.data_ptr(), ref.data_ptr(), out.data_ptr(), tmp.numel(),
                                     float(eps), _stream_ptr())
    native.check(st, 'ref_update')
    return out


def ref_update_frames(y, ref, T, o, group, eps=1e-5, out=None):
    """sigmoid(y_t + inverse_sigmoid(ref)) read straight from the grouped per-frame MLP output
    y [R, T * op] (frame t's `o` outputs at columns [t op, t op + o)); ref [..., o] with R * T rows
    ordered (r // group, t, r % group) -> same shape as ref.  One launch, no layout copy."""
    lib = native.load()
    _dev(y, 'y', torch.float32)
    _dev(ref, 'ref', torch.float32)
    R = y.shape[0]
    op = y.shape[1] // int(T)
    _require(y.dim() == 2 and op * T == y.shape[1] and op >= o and ref.shape[-1] == o
             and ref.numel() == R * T * o and R % int(group) == 0,
             'ref_update_frames: y [R, T*op], ref [R*T rows, o], R % group == 0')
    if out is None:
        out = torch.empty_like(ref)
    else:       # (a level of the decoder's preallocated [levels, ...] stack of intermediate references)
        _dev(out, 'out', torch.float32)
        _require(tuple(out.shape) == tuple(ref.shape), 'ref_update_frames: out shaped like ref')
    with torch.cuda.device(y.device), _Timed('ref_update'):
        st = lib.pave_ref_update_frames_f32(y.data_ptr(), ref.data_ptr(), out.data_ptr(), R, int(T),
                                            op, int(o), int(group), float(eps), _stream_ptr())
    native.check(st, 'ref_update_frames')
    return out


def groupnorm_nhwc_into(x_rows, gamma, beta, num_groups, eps, dst):
    """GroupNorm of an NHWC map given as rows x_rows [N, HW, C] (dense), written into dst
    [N, HW, C] whose batch stride may be larger than HW * C (a slice of a [N, S, C] token buffer).
    fp64 statistics in a fixed order."""
    lib = native.load()
    _dev(x_rows, 'x', torch.float32)
    _require(x_rows.dim() == 3, 'groupnorm_nhwc_into: x [N, HW, C]')
    N, HW, C = x_rows.shape
    _dev(gamma, 'gamma', torch.float32)
    _dev(beta, 'beta', torch.float32)
    _require(gamma.numel() == C and beta.numel() == C, 'groupnorm_nhwc_into: gamma / beta [C]')
    _require(dst.is_cuda and dst.dtype == torch.float32 and tuple(dst.shape) == (N, HW, C)
             and dst.stride(2) == 1 and dst.stride(1) == C and (N == 1 or dst.stride(0) >= HW * C),
             'groupnorm_nhwc_into: dst [N, HW, C] with dense rows')
    nchunks = max(1, min(128, (HW + 63) // 64))
    partial = torch.empty((N, nchunks, num_groups, 2), dtype=torch.float64, device=x_rows.device)
    ab = torch.empty((N, 2, C), dtype=torch.float32, device=x_rows.device)
    with torch.cuda.device(x_rows.device), _Timed('groupnorm'):
        st = lib.pave_groupnorm_nhwc_f32(x_rows.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                         dst.data_ptr(), dst.stride(0) if N > 1 else HW * C, N, HW,
                                         C, int(num_groups), float(eps), partial.data_ptr(), nchunks,
                                         ab.data_ptr(), _stream_ptr())
    native.check(st, 'groupnorm_nhwc_into')
    return dst


def groupnorm_levels_into(levels, num_groups):
    """GroupNorm of several NHWC maps of one N and C at once -- the neck's levels -- as three launches in all
    (statistics, scale / shift, apply) instead of three per level; per level the values of `groupnorm_nhwc_into`,
    bit for bit.  levels: [(x_rows [N, HW_l, C], gamma, beta, eps, dst [N, HW_l, C])]."""
    lib = native.load()
    _require(len(levels) >= 1, 'groupnorm_levels_into: at least one level')
    N, _, C = levels[0][0].shape
    dev = levels[0][0].device
    for i0 in range(0, len(levels), 4):       # (the entry takes up to four maps per call)
        part = levels[i0:i0 + 4]
        arr = (native.GnLevel * len(part))()
        total_chunks = 0
        for j, (x_rows, gamma, beta, eps, dst) in enumerate(part):
            _dev(x_rows, 'x', torch.float32)
            _dev(gamma, 'gamma', torch.float32)
            _dev(beta, 'beta', torch.float32)
            _require(x_rows.dim() == 3 and x_rows.shape[0] == N and x_rows.shape[2] == C and x_rows.device == dev,
                     'groupnorm_levels_into: every x [N, HW_l, C] on one device')
            HW = x_rows.shape[1]
            _require(gamma.numel() == C and beta.numel() == C, 'groupnorm_levels_into: gamma / beta [C]')
            _require(dst.is_cuda and dst.dtype == torch.float32 and tuple(dst.shape) == (N, HW, C)
                     and dst.stride(2) == 1 and dst.stride(1) == C and (N == 1 or dst.stride(0) >= HW * C),
                     'groupnorm_levels_into: dst [N, HW, C] with dense rows')
            nchunks = max(1, min(128, (HW + 63) // 64))       # (as groupnorm_nhwc_into)
            arr[j] = native.GnLevel(x_rows.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dst.data_ptr(),
                                    dst.stride(0) if N > 1 else HW * C, HW, nchunks, float(eps))
            total_chunks += nchunks
        partial = torch.empty((N * total_chunks * num_groups * 2,), dtype=torch.float64, device=dev)
        ab = torch.empty((len(part), N, 2, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev), _Timed('groupnorm'):
            st = lib.pave_groupnorm_levels_nhwc_f32(ctypes.cast(arr, ctypes.c_void_p), len(part), N, C,
                                                    int(num_groups), partial.data_ptr(), ab.data_ptr(), _stream_ptr())
        native.check(st, 'groupnorm_levels_into')
    return [lv[4] for lv in levels]


def gemm_bf16x3_ln(a, w_planes, bias, residual, gamma, beta, eps, out=None):
    """out[M, 256] = LayerNorm(a @ W^T + bias + residual) * gamma + beta in ONE launch (3 bf16
    planes; the 128 x 256 block tile owns whole rows).  `residual` may be None or the tensor given
    as `out`."""
    lib = native.load()
    _dev(a, 'a', torch.float32)
    npl = _planes(w_planes, only=_Q_PLANES)
    _require(a.dim() == 2 and w_planes.shape[0] * 16 == a.shape[1],
             'gemm_bf16x3_ln: a [M,K], w_planes [K/16,3,N,16] (split_weight_bf16x3, 3 planes or fp16)')
    M, K = a.shape
    N = w_planes.shape[2]
    _require(N == 256, 'gemm_bf16x3_ln: N == 256')
    for t, nm in ((bias, 'bias'), (gamma, 'gamma'), (beta, 'beta')):
        if t is not None:
            _dev(t, nm, torch.float32)
            _require(t.numel() == N, f'gemm_bf16x3_ln: {nm} has {t.numel()} elements, expected {N}')
    _require(gamma is not None and beta is not None, 'gemm_bf16x3_ln: gamma and beta are required')
    if residual is not None:
        _dev(residual, 'residual', torch.float32)
        _require(tuple(residual.shape) == (M, N), 'gemm_bf16x3_ln: residual [M,N]')
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    else:
        _dev(out, 'out', torch.float32)
        _require(tuple(out.shape) == (M, N), 'gemm_bf16x3_ln: out [M,N]')
    M1 = round_split_rows(M, 1)
    if M1 is not None:      # (the last, nearly empty round of block slots as a launch of the small-row form)
        gemm_bf16x3_ln(a[:M1], w_planes, bias, residual[:M1] if residual is not None else None, gamma, beta, eps,
                       out=out[:M1])
        gemm_bf16x3_ln(a[M1:], w_planes, bias, residual[M1:] if residual is not None else None, gamma, beta, eps,
                       out=out[M1:])
        return out
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    with torch.cuda.device(a.device), _Timed(_gemm_tag('gemm_bf16x3_ln', M), 2 * M * K * N, (M, K, N, 'ln', 'res' if residual is not None else '')):
        st = lib.pave_gemm_bf16x3_ln_f32(a.data_ptr(), w_planes.data_ptr(), ptr(bias), ptr(residual),
                                         gamma.data_ptr(), beta.data_ptr(), float(eps),
                                         out.data_ptr(), M, K, N, npl, _stream_ptr())
    native.check(st, 'gemm_bf16x3_ln')
    return out


def gemm_fp16_act(a, w_plane, bias=None, relu=False, out_half=False, residual=None, ln=None, out=None):
    """fp16 operand mode with fp16 activations around the launch (pave_gemm_fp16_act_f32; wide tile forms):
    a [M, K] fp32 or float16, w_plane = the ONE fp16 plane of split_weight_bf16x3(weight, PLANES_FP16), N % 256 == 0.
    ln = None: act(a W^T + bias) -> [M, N] fp32, or float16 with out_half (an activation that only feeds the next
    GEMM).  ln = (gamma, beta, eps): LayerNorm(a W^T + bias + residual) gamma + beta -> [M, 256] fp32 (`residual`
    may be the tensor given as `out`)."""
    lib = native.load()
    _require(isinstance(a, torch.Tensor) and a.is_cuda and a.is_contiguous() and a.dim() == 2
             and a.dtype in (torch.float32, torch.float16), 'gemm_fp16_act: a [M, K] fp32 or float16 on the device')
    _require(_planes(w_plane) == PLANES_FP16, 'gemm_fp16_act: one fp16 plane (split_weight_bf16x3(w, PLANES_FP16))')
    M, K = a.shape
    N = w_plane.shape[2]
    _require(w_plane.shape[0] * 16 == K and N % 256 == 0, 'gemm_fp16_act: w_plane [K/16, 1, N % 256 == 0, 16]')
    if bias is not None:
        _dev(bias, 'bias', torch.float32)
        _require(bias.numel() == N, 'gemm_fp16_act: bias [N]')
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    if ln is not None:
        gamma, beta, eps = ln
        _dev(gamma, 'gamma', torch.float32)
        _dev(beta, 'beta', torch.float32)
        _require(N == 256 and gamma.numel() == 256 and beta.numel() == 256 and not out_half and not relu,
                 'gemm_fp16_act: the LayerNorm form has N == 256 and an fp32 output')
        if residual is not None:
            _dev(residual, 'residual', torch.float32)
            _require(tuple(residual.shape) == (M, N), 'gemm_fp16_act: residual [M, N]')
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        _require(out.dtype == torch.float32 and tuple(out.shape) == (M, N) and out.is_contiguous(), 'gemm_fp16_act: out')
        args = (ptr(residual), gamma.data_ptr(), beta.data_ptr(), float(eps))
        tag = _gemm_tag('gemm_bf16x3_ln', M)
    else:
        _require(residual is None and out is None, 'gemm_fp16_act: the plain form takes bias + activation only')
        out = torch.empty((M, N), dtype=torch.float16 if out_half else torch.float32, device=a.device)
        args = (None, None, None, 0.0)
        tag = _gemm_tag('gemm_bf16x3', M)
    note = (M, K, N, 'f16act', 'a16' if a.dtype == torch.float16 else '', 'o16' if out_half else '',
            'ln' if ln is not None else ('relu' if relu else ''), 'res' if residual is not None else '')
    with torch.cuda.device(a.device), _Timed(tag, 2 * M * K * N, note):
        st = lib.pave_gemm_fp16_act_f32(a.data_ptr(), int(a.dtype == torch.float16), w_plane.data_ptr(), ptr(bias),
                                        *args, out.data_ptr(), int(bool(out_half)), M, K, N,
                                        2 if relu == 'gelu' else int(bool(relu)), _stream_ptr())
    native.check(st, 'gemm_fp16_act')
    return out


def gemm_bf16x3_ex(a, w_planes, bias=None, residual=None, residual_rows=0, n_split=0, relu=False,
                   a_bias=None, fp16=False, _out=None):
    """gemm_bf16x3 with a row-periodic residual table (`residual` [residual_rows, N], row m adds
    residual[m % residual_rows]) and / or the output cut at column `n_split` into two dense
    matrices -> out [M, n_split], out2 [M, N - n_split] (n_split = 0: one output, out2 = None)."""
    lib = native.load()
    _dev(a, 'a', torch.float32)
    npl = _planes(w_planes, fp16=fp16)
    _require(a.dim() == 2 and w_planes.shape[1] in (1, 2, 3) and w_planes.shape[0] * 16 == a.shape[1],
             'gemm_bf16x3_ex: a [M,K], w_planes [K/16,3,N,16] (split_weight_bf16x3)')
    M, K = a.shape
    N = w_planes.shape[2]
    _require(N % 128 == 0, 'gemm_bf16x3_ex: N % 128 == 0')
    for t, nm, n in ((bias, 'bias', N), (a_bias, 'a_bias', K)):
        if t is not None:
            _dev(t, nm, torch.float32)
            _require(t.numel() == n, f'gemm_bf16x3_ex: {nm} has {t.numel()} elements, expected {n}')
    rr = int(residual_rows)
    if residual is not None:
        _dev(residual, 'residual', torch.float32)
        _require(tuple(residual.shape) == ((rr if rr > 0 else M), N),
                 'gemm_bf16x3_ex: residual [residual_rows or M, N]')
    else:
        _require(rr == 0, 'gemm_bf16x3_ex: residual_rows without a residual')
    n_split = int(n_split)
    _require(n_split == 0 or (0 < n_split < N and n_split % 128 == 0),
             'gemm_bf16x3_ex: 0 < n_split < N, n_split % 128 == 0')
    if N % 256 == 0 and npl in _Q_PLANES and a_bias is None and rr == 0 and (n_split == 0 or n_split % 256 == 0) \
            and _out is None:
        M1 = round_split_rows(M, N // 256)      # (a nearly empty last round of block slots: see round_split_rows)
        if M1 is not None:
            out = torch.empty((M, n_split or N), dtype=torch.float32, device=a.device)
            out2 = torch.empty((M, N - n_split), dtype=torch.float32, device=a.device) if n_split else None
            for r0, r1 in ((0, M1), (M1, M)):
                gemm_bf16x3_ex(a[r0:r1], w_planes, bias, residual[r0:r1] if residual is not None else None, 0, n_split,
                               relu, None, fp16, _out=(out[r0:r1], out2[r0:r1] if out2 is not None else None))
            return out, out2
    if _out is not None:
        out, out2 = _out
    else:
        out = torch.empty((M, n_split or N), dtype=torch.float32, device=a.device)
        out2 = torch.empty((M, N - n_split), dtype=torch.float32, device=a.device) if n_split else None
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    with torch.cuda.device(a.device), _Timed(_gemm_tag('gemm_bf16x3', M), 2 * M * K * N, (M, K, N, 'ex', f'rows{rr}', f'split{n_split}')):
        st = lib.pave_gemm_bf16x3_ex_f32(a.data_ptr(), ptr(a_bias), w_planes.data_ptr(), ptr(bias),
                                         ptr(residual), rr, out.data_ptr(), ptr(out2), n_split,
                                         M, K, N, int(bool(relu)), npl, _stream_ptr())
    native.check(st, 'gemm_bf16x3_ex')
    return out, out2


def swin_window_attn(qkv, bias_t, pad_qkv, heads, window, shift, scale):
    """The (shifted-)window attention core of a Swin block on the un-partitioned token map
    (pave_swin_window_attn_f32): qkv [B, H, W, 3C], bias_t [heads, ws^2, ws^2] (relative-position bias, key-major),
    pad_qkv [3C] (the qkv Linear's bias) -> [B, H, W, C]."""
    lib = native.load()
    for t, nm in ((qkv, 'qkv'), (bias_t, 'bias_t'), (pad_qkv, 'pad_qkv')):
        _dev(t, nm, torch.float32)
    _require(qkv.dim() == 4 and qkv.shape[3] % 3 == 0, 'swin_window_attn: qkv [B, H, W, 3C]')
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    n = window * window
    _require(tuple(bias_t.shape) == (heads, n, n) and pad_qkv.numel() == C3 and C == heads * 32,
             'swin_window_attn: bias_t [heads, ws^2, ws^2], pad_qkv [3C], head dim 32')
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        st = lib.pave_swin_window_attn_f32(qkv.data_ptr(), bias_t.data_ptr(), pad_qkv.data_ptr(), out.data_ptr(),
                                           B, H, W, C, int(heads), int(window), int(shift), float(scale),
                                           _stream_ptr())
    native.check(st, 'swin_window_attn')
    return out


def merge_softmax_partials(parts, C, H):
    """parts [G, U, C + 2 H] (all-gathered partial rows | per-head max | per-head sum-exp) -> [U, C]: the
    exact full-softmax row (pave_merge_softmax_partials_f32), one launch."""
    lib = native.load()
    _dev(parts, 'parts', torch.float32)
    _require(parts.dim() == 3 and parts.shape[2] == C + 2 * H, 'merge_softmax_partials: parts [G, U, C + 2 H]')
    G, U = parts.shape[0], parts.shape[1]
    out = torch.empty((U, C), dtype=torch.float32, device=parts.device)
    with torch.cuda.device(parts.device):
        st = lib.pave_merge_softmax_partials_f32(parts.data_ptr(), out.data_ptr(), G, U, int(C), int(H),
                                                 _stream_ptr())
    native.check(st, 'merge_softmax_partials')
    return out


def mha_core(qkv, n_seq, seq_len, num_heads):
    """Scaled-dot-product core of the decoders' self-attention: qkv [n_seq * seq_len, >= 3 * E]
    (q | k | v columns, E = num_heads * 32; row = (sequence, position)) -> [n_seq * seq_len, E] =
    softmax(q k^T / sqrt(32)) v per (sequence, head).  pave_mha_core_f32 (csrc/pave_decoder.hip)."""
    lib = native.load()
    _dev(qkv, 'qkv', torch.float32)
    E = int(num_heads) * 32
    _require(qkv.dim() == 2 and qkv.shape[0] == n_seq * seq_len and qkv.shape[1] >= 3 * E,
             'mha_core: qkv [n_seq * seq_len, >= 3 * num_heads * 32]')
    out = torch.empty((qkv.shape[0], E), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        st = lib.pave_mha_core_f32(qkv.data_ptr(), out.data_ptr(), int(n_seq), int(seq_len),
                                   int(num_heads), qkv.stride(0), _stream_ptr())
    native.check(st, 'mha_core')
    return out


def topk_rows(x, k):
    """torch.topk(x, k, dim=1) for a 2-D fp32 device tensor as ONE launch (pave_topk_rows_f32):
    -> (values [rows, k], indices [rows, k] int64), sorted by value descending, ties by the
    lower index.  x may be any strided 2-D view (no copy)."""
    lib = native.load()
    _require(isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
             and x.shape[0] > 0 and x.stride(1) > 0 and (x.shape[0] == 1 or x.stride(0) > 0),
             'topk_rows: x [rows, n] fp32 on the device, positive strides')
    rows, n = x.shape
    k = int(k)
    _require(0 < k <= n and n <= 32768 and k <= 1024, 'topk_rows: 0 < k <= n <= 32768, k <= 1024')
    values = torch.empty((rows, k), dtype=torch.float32, device=x.device)
    index = torch.empty((rows, k), dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.pave_topk_rows_f32(x.data_ptr(), values.data_ptr(), index.data_ptr(), rows, n,
                                    x.stride(0), x.stride(1), k, _stream_ptr())
    native.check(st, 'topk_rows')
    return values, index


def gather_frame_poses(poses, index, T):
    """poses [B, T*Q, C] (frame-major rows), index [B, N] int64 -> [T, B*N, C]: the selected
    queries of every frame, frame-major (pave_gather_frame_poses_f32)."""
    lib = native.load()
    _dev(poses, 'poses', torch.float32)
    _dev(index, 'index', torch.int64)
    _require(poses.dim() == 3 and poses.shape[1] % T == 0 and index.dim() == 2
             and index.shape[0] == poses.shape[0], 'gather_frame_poses: poses [B, T*Q, C], index [B, N]')
    B, TQ, C = poses.shape
    N = index.shape[1]
    out = torch.empty((T, B * N, C), dtype=torch.float32, device=poses.device)
    with torch.cuda.device(poses.device):
        st = lib.pave_gather_frame_poses_f32(poses.data_ptr(), index.data_ptr(), out.data_ptr(), B,
                                             int(T), TQ // T, N, C, _stream_ptr())
    native.check(st, 'gather_frame_poses')
    return out


def gather_rows_add(src, index, add=None):
    """src [n, S, C] fp32 (rows dense, any batch stride), index [n, Q] int64 -> rows [n, Q, C] = src[b, index[b, q]]
    and, with add [Q, C], (rows, rows + add): the proposal top-k's `tgt` and `query = tgt + query`
    (OT:21386, 21417) in one launch -- pave_gather_rows_add_f32."""
    lib = native.load()
    _require(src.is_cuda and src.dtype == torch.float32 and src.dim() == 3 and src.stride(2) == 1
             and src.stride(1) == src.shape[2], 'gather_rows_add: src [n, S, C] fp32 on the device, dense rows')
    _dev(index, 'index', torch.int64)
    n, S, C = src.shape
    _require(index.dim() == 2 and index.shape[0] == n, 'gather_rows_add: index [n, Q]')
    Q = index.shape[1]
    rows = torch.empty((n, Q, C), dtype=torch.float32, device=src.device)
    total = None
    if add is not None:
        _dev(add, 'add', torch.float32)
        _require(tuple(add.shape) == (Q, C), 'gather_rows_add: add [Q, C]')
        total = torch.empty_like(rows)
    with torch.cuda.device(src.device):
        st = lib.pave_gather_rows_add_f32(src.data_ptr(), src.stride(0) if n > 1 else 0, index.data_ptr(),
                                          add.data_ptr() if add is not None else None, rows.data_ptr(),
                                          total.data_ptr() if total is not None else None, n, Q, S, C,
                                          _stream_ptr())
    native.check(st, 'gather_rows_add')
    return rows if add is None else (rows, total)


def proposal_refs_(kpt, props, index, T):
    """kpt [n, Q, 2K] fp32 (unit column stride; a column slice of a padded matrix is fine), in place:
    kpt[..., 0::2] += props[b, index, 0], kpt[..., 1::2] += props[b, index, 1] (props [n or 1, S, 2]); returns
    refs [n, T*Q, 2K] = sigmoid(kpt) repeated for the T frames (OT:21390-21391, 21412) -- pave_proposal_refs_f32."""
    lib = native.load()
    _require(kpt.is_cuda and kpt.dtype == torch.float32 and kpt.dim() == 3 and kpt.stride(2) == 1
             and kpt.stride(0) == kpt.shape[1] * kpt.stride(1), 'proposal_refs_: kpt [n, Q, 2K] fp32, rows evenly strided')
    _dev(index, 'index', torch.int64)
    n, Q, K2 = kpt.shape
    _require(props.is_cuda and props.dtype == torch.float32 and props.dim() == 3 and props.shape[2] == 2
             and props.shape[0] in (1, n) and props[0].is_contiguous(), 'proposal_refs_: props [n | 1, S, 2] fp32')
    _require(tuple(index.shape) == (n, Q), 'proposal_refs_: index [n, Q]')
    refs = torch.empty((n, int(T) * Q, K2), dtype=torch.float32, device=kpt.device)
    with torch.cuda.device(kpt.device):
        st = lib.pave_proposal_refs_f32(kpt.data_ptr(), kpt.stride(1), props.data_ptr(),
                                        props.stride(0) if props.shape[0] > 1 else 0, index.data_ptr(),
                                        refs.data_ptr(), n, Q, props.shape[1], K2, int(T), _stream_ptr())
    native.check(st, 'proposal_refs_')
    return refs


def pose_finalize(kpts, sigmas, scores, wh, sf=None):
    """Post-processing of the refined poses (HEAD:1440-1490 + get_p) in one launch:
    kpts, sigmas [B, N, K, 2], scores [B, N], wh [B, 2] (image w, h), sf [B, 2] or None (rescale)
    -> (det_kpts [B, N, K, 3], det_bboxes [B, N, 5])."""
    lib = native.load()
    for t, nm in ((kpts, 'kpts'), (scores, 'scores'), (wh, 'wh')):
        _dev(t, nm, torch.float32)
    # sigmas: dense, or evenly strided (x, y) rows -- a 2-column slice of the sigma branch's padded output
    _require(sigmas.is_cuda and sigmas.dtype == torch.float32 and sigmas.dim() == 4 and sigmas.stride(3) == 1
             and sigmas.stride(1) == sigmas.shape[2] * sigmas.stride(2)
             and sigmas.stride(0) == sigmas.shape[1] * sigmas.stride(1) and sigmas.stride(2) >= 2,
             'pose_finalize: sigmas [B, N, K, 2] fp32 on the device, rows evenly strided')
    sigma_ld = sigmas.stride(2)
    _require(kpts.dim() == 4 and kpts.shape[-1] == 2 and kpts.shape == sigmas.shape
             and tuple(scores.shape) == tuple(kpts.shape[:2]) and wh.numel() == 2 * kpts.shape[0],
             'pose_finalize: kpts / sigmas [B, N, K, 2], scores [B, N], wh [B, 2]')
    B, N, K, _ = kpts.shape
    if sf is not None:
        _dev(sf, 'sf', torch.float32)
        _require(sf.numel() == 2 * B, 'pose_finalize: sf [B, 2]')
    det_kpts = torch.empty((B, N, K, 3), dtype=torch.float32, device=kpts.device)
    det_bboxes = torch.empty((B, N, 5), dtype=torch.float32, device=kpts.device)
    with torch.cuda.device(kpts.device):
        st = lib.pave_pose_finalize_f32(kpts.data_ptr(), sigmas.data_ptr(), scores.data_ptr(),
                                        wh.data_ptr(), sf.data_ptr() if sf is not None else None,
                                        det_kpts.data_ptr(), det_bboxes.data_ptr(), B, N, K,
                                        int(sf is not None), sigma_ld, _stream_ptr())
    native.check(st, 'pose_finalize')
    return det_kpts, det_bboxes


def split_conv3x3_weight(weight, planes=3):
    """Conv2d weight [Cout, Cin, 3, 3] -> the operand of `conv3x3_split`: rows [Cout, (ky, kx, cin)]
    split and re-laid by `split_weight_bf16x3`."""
    _require(weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3), 'split_conv3x3_weight: [Cout,Cin,3,3]')
    cout, cin = weight.shape[:2]
    # 3 planes: rows / columns zero-padded to Cout % 64 == 0, 9 Cin % 32 == 0 (HRNet's 48 / 96 channels)
    return split_weight_bf16x3(weight.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous(), planes,
                               pad=planes in _Q_PLANES)


def bottleneck_chain(c1, w2_planes, b2, w3_planes, b3, residual=None, a2=None, w1n_planes=None,
                     b1n=None, out=None, c2=None):
    """One launch for a 64-channel ResNet Bottleneck from its 3x3 convolution on, chained with the
    next block's conv1 (pave_bottleneck_chain_f32):  c1 [N, 64, H, W] channels_last ->
    out = relu(conv3(relu(conv3x3(c1) + b2)) (+ downsample(a2)) + b3 + residual) [N, 256, H, W] and
    c1n = relu(conv1_next(out) + b1n) [N, cn, H, W] (None without w1n_planes).  residual
    [N, 256, H, W] channels_last (may be given as `out`: in place) or a2 [N, k2, H, W] channels_last
    with w3_planes = the planes of the [256, 64 + k2] concatenated weight.
    c1 = w2_planes = None with c2 [N, 64, H, W] channels_last given: the launch starts at conv3
    (the 3x3 was run by conv3x3_split)."""
    lib = native.load()
    tail_only = c1 is None
    src = c2 if tail_only else c1
    _require(src is not None and src.is_cuda and src.dtype == torch.float32 and src.dim() == 4
             and src.shape[1] == 64 and src.is_contiguous(memory_format=torch.channels_last),
             'bottleneck_chain: c1 (or c2) [N, 64, H, W] fp32 channels_last')
    _require((w2_planes is None) == tail_only, 'bottleneck_chain: c1 and w2_planes go together')
    N, _, H, W = src.shape
    M = N * H * W
    npl = _planes(w3_planes, 'w3_planes', only=_Q_PLANES)
    k2 = 0
    if a2 is not None:
        _require(residual is None and a2.is_cuda and a2.dtype == torch.float32 and a2.dim() == 4
                 and tuple(a2.shape[0:1] + a2.shape[2:]) == (N, H, W)
                 and a2.is_contiguous(memory_format=torch.channels_last),
                 'bottleneck_chain: a2 [N, k2, H, W] channels_last, no residual')
        k2 = a2.shape[1]
    if not tail_only:
        _require(_planes(w2_planes, 'w2_planes') == npl, 'bottleneck_chain: one operand mode for all weights')
        _require(tuple(w2_planes.shape) == (36, w3_planes.shape[1], 64, 16), 'bottleneck_chain: w2_planes [36, 3, 64, 16]')
    _require(tuple(w3_planes.shape) == ((64 + k2) // 16, w3_planes.shape[1], 256, 16),
             'bottleneck_chain: w3_planes [(64 + k2)/16, 3, 256, 16]')
    if residual is not None:
        _require(residual.is_cuda and residual.dtype == torch.float32
                 and tuple(residual.shape) == (N, 256, H, W)
                 and residual.is_contiguous(memory_format=torch.channels_last),
                 'bottleneck_chain: residual [N, 256, H, W] channels_last')
    cn = 0
    if w1n_planes is not None:
        _require(_planes(w1n_planes, 'w1n_planes') == npl, 'bottleneck_chain: one operand mode for all weights')
        cn = w1n_planes.shape[2]
        _require(tuple(w1n_planes.shape) == (16, w3_planes.shape[1], cn, 16) and cn in (64, 128),
                 'bottleneck_chain: w1n_planes [16, 3, 64 | 128, 16]')
    for b, n in ((b2, 64), (b3, 256), (b1n, cn)):
        if b is not None:
            _dev(b, 'bias', torch.float32)
            _require(b.numel() == n, 'bottleneck_chain: bias sizes 64 / 256 / cn')
    if out is None:
        out = torch.empty((N, H, W, 256), dtype=torch.float32, device=src.device).permute(0, 3, 1, 2)
    else:
        _require(out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (N, 256, H, W)
                 and out.is_contiguous(memory_format=torch.channels_last),
                 'bottleneck_chain: out [N, 256, H, W] channels_last')
    if not tail_only:
        c2 = torch.empty((M, 64), dtype=torch.float32, device=src.device)
    c1n = torch.empty((N, H, W, cn), dtype=torch.float32, device=src.device) if cn else None
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    flops = 2 * M * ((0 if tail_only else 576 * 64) + (64 + k2) * 256 + 256 * cn)
    with torch.cuda.device(src.device), _Timed('bottleneck_chain', flops,
                                               (M, 64, 256, cn, f'k2={k2}', 'tail' if tail_only else '')):
        st = lib.pave_bottleneck_chain_f32(
            ptr(c1), ptr(w2_planes), ptr(b2), c2.data_ptr(), w3_planes.data_ptr(), ptr(b3),
            ptr(residual), ptr(a2), k2, out.data_ptr(), ptr(w1n_planes), ptr(b1n), ptr(c1n), cn,
            N, H, W, npl, _stream_ptr())
    native.check(st, 'bottleneck_chain')
    return out, (c1n.permute(0, 3, 1, 2) if cn else None)


def conv3x3_split(x, w_planes, bias=None, stride=1, relu=False, fp16=False, residual=None, cout=None):
    """3x3 / pad 1 convolution of a channels_last map through the split-operand GEMM kernel
    (implicit GEMM), bias (+ residual) (+ReLU) fused.  x [N, Cin, H, W] channels_last; w_planes =
    split_conv3x3_weight(weight) -> [N, Cout, Ho, Wo] channels_last.  `cout` = the real number of
    output channels when the planes were zero-padded (3 planes: Cin % 16 == 0, Cout % 4 == 0);
    residual [N, Cout, Ho, Wo] channels_last is added before the ReLU (3 planes)."""
    lib = native.load()
    _require(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4, 'conv3x3_split: fp32 4-D')
    _require(x.is_contiguous(memory_format=torch.channels_last), 'conv3x3_split: channels_last input')
    npl = _planes(w_planes, fp16=fp16)
    N, Cin, H, W = x.shape
    kp = 9 * Cin if npl not in _Q_PLANES else (9 * Cin + 31) // 32 * 32
    _require(w_planes.shape[0] * 16 == kp, 'conv3x3_split: w_planes [9*Cin/16, P, Cout, 16] (split_conv3x3_weight)')
    Cout = int(cout) if cout is not None else w_planes.shape[2]
    _require((Cout + 63) // 64 * 64 == w_planes.shape[2], 'conv3x3_split: cout does not match the planes')
    if bias is not None:
        _dev(bias, 'bias', torch.float32)
        _require(bias.numel() == Cout, 'conv3x3_split: bias [Cout]')
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if residual is not None:
        _require(residual.is_cuda and residual.dtype == torch.float32
                 and tuple(residual.shape) == (N, Cout, Ho, Wo)
                 and residual.is_contiguous(memory_format=torch.channels_last),
                 'conv3x3_split: residual [N, Cout, Ho, Wo] channels_last')
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    # few output pixels, many input channels: the split-K form (parts of the K axis on their own
    # workgroups, ordered sum in a second launch); the workspace comes from torch's stream-aware
    # caching allocator
    ws_bytes = lib.pave_conv3x3_splitk_workspace_bytes(N, H, W, Cin, Cout, int(stride)) \
        if npl in _Q_PLANES else 0
    if ws_bytes > 0:
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device), _Timed('conv3x3_split', 2 * N * Ho * Wo * Cout * 9 * Cin,
                                                 (N * Ho * Wo, 9 * Cin, Cout, f'3x3 s{stride} split-K', 'res' if residual is not None else '')):
            st = lib.pave_conv3x3_splitk_f32(
                x.data_ptr(), w_planes.data_ptr(), bias.data_ptr() if bias is not None else None,
                residual.data_ptr() if residual is not None else None,
                y.data_ptr(), N, H, W, Cin, Cout, int(stride), int(bool(relu)), ws.data_ptr(), ws_bytes,
                npl, _stream_ptr())
        native.check(st, 'conv3x3_splitk')
        return y.permute(0, 3, 1, 2)
    with torch.cuda.device(x.device), _Timed('conv3x3_split', 2 * N * Ho * Wo * Cout * 9 * Cin, (N * Ho * Wo, 9 * Cin, Cout, f'3x3 s{stride}', 'res' if residual is not None else '')):
        st = lib.pave_conv3x3_split_f32(
            x.data_ptr(), w_planes.data_ptr(), bias.data_ptr() if bias is not None else None,
            residual.data_ptr() if residual is not None else None,
            y.data_ptr(), N, H, W, Cin, Cout, int(stride), int(bool(relu)), npl, _stream_ptr())
    native.check(st, 'conv3x3_split')
    return y.permute(0, 3, 1, 2)
