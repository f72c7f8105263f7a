"""Deformable-attention modules of the PAVE-Net forward path, on the fused HIP kernels.

Same registry names, ctor kwargs, forward kwargs and state-dict keys as the reference:

* ``mmcv.MultiScaleDeformableAttention``                      MO:207-412   (encoder, a2)
* ``opera.MultiScaleDeformablePoseAttention``                 OT:251-427   (PETR decoder, a15)
* ``opera.MulFramesMultiScaleDeformablePoseAttentionNumFrames3/5``  OT:1543-1863 / 2738-3117 (a5)
* ``mmcv.MulFramesMultiScaleDeformableAttentionNumFrames3/5``       MO:1268-1587 / 1590-1981 (a7)

plus ``...MulFrames...Attention`` generalisations to any odd ``num_frames`` (the reference
hard-codes T = 3 / 5 with one named Linear pair per frame; frame offset k from the centre
gets the prefix 'pre_' * (-k) / '' / 'next_' * k, which reproduces the reference's names).

What changes relative to the reference is *how* the forward runs on the device:
one concatenated GEMM produces the offsets and logits of all frames, and ONE HIP launch
(pavenet_amd/csrc/pave_kernels.hip) does the joint softmax over T*L*P logits, the sampling
location arithmetic, the bilinear gathers of all frames and the cross-frame fusion
(the reference: 2T Linears, T softmaxes, T exp-sums, T sampler launches, 3T elementwise ops).
The joint softmax is stabilised; it equals the reference's Z_t re-weighting whenever that
does not overflow (|logit| < ~88), see DESIGN.md "Intentional divergences".
"""
import math
import warnings

import torch
import torch.nn as nn

from . import ops
from .bricks import (BaseModule, SourceKey, batch_first, constant_init, linear_residual_norm,
                     linear_rows, seq_first_view, xavier_init)
from .census import note_slow_path
from .registry import ATTENTION, MMCV_ATTENTION


def frame_prefixes(num_frames):
    assert num_frames % 2 == 1, 'num_frames must be odd (centre frame + symmetric neighbours)'
    c = num_frames // 2
    return ['pre_' * (c - t) if t < c else 'next_' * (t - c) for t in range(num_frames)]


def _check_heads(embed_dims, num_heads):
    if embed_dims % num_heads != 0:
        raise ValueError(f'embed_dims must be divisible by num_heads, '
                         f'but got {embed_dims} and {num_heads}')
    d = embed_dims // num_heads
    if not (isinstance(d, int) and d > 0 and (d & (d - 1)) == 0):
        warnings.warn("You'd better set embed_dims in MultiScaleDeformAttention to make the "
                      'dimension of each attention head a power of 2 which is more efficient '
                      'in our implementation.')


class _CatProj:
    """Caches the row-concatenation [offsets of all frames ; logits of all frames] of the
    per-frame ``sampling_offsets`` / ``attention_weights`` Linears, so one GEMM feeds the
    fused kernel.  Rebuilt whenever a source parameter changes (version / storage)."""

    def _cat_sources(self):
        raise NotImplementedError

    def _cat_proj(self, frames=None):
        """frames: optional tuple of frame indices (frame-sharded multi-GPU: local frames)."""
        offs, logits = self._cat_sources()
        if frames is not None:
            offs, logits = [offs[t] for t in frames], [logits[t] for t in frames]
        srcs = [m.weight for m in offs + logits] + [m.bias for m in offs + logits]
        key = SourceKey(srcs, frames)
        if getattr(self, '_cat_key', None) != key:
            with torch.no_grad():
                self._cat_w = torch.cat([m.weight for m in offs + logits], 0).contiguous()
                self._cat_b = torch.cat([m.bias for m in offs + logits], 0).contiguous()
            self._cat_key = key
        return self._cat_w, self._cat_b


FOLD_QUERY_POS = True   # A/B switch: `query + query_pos` of the T-frame attentions folded into the projection GEMM


def cat_proj_rows(mod, query, query_pos, frames=None):
    """The offsets | logits projection of the T-frame attentions on [bs * Q, C] rows: proj = (query + query_pos)
    W_cat^T + b_cat.  The decoders' positional term is a parameter-derived constant per query ([Q, C], the same for
    every clip / pose: a stride-0 expand of an embedding slice), so on the device it rides the GEMM epilogue as a
    row-periodic table pos W_cat^T + b_cat (cached per weights, built in fp64) -- no `query + query_pos` launch in
    front of the GEMM (5 elementwise launches per step on the latency-bound decoder tail; the decoders'
    self-attention does the same with its q | k | v projection, bricks.MultiheadAttention).  The weight rows are
    zero-padded to N % 128 == 0 (the row stride of the result; the fused kernels take it as `proj_ld`).
    query, query_pos: sequence-first [Q, bs, C].  Returns proj [bs * Q, >= N]; anything else: the plain expression."""
    from .bricks import fused_mode
    q = batch_first(query)
    bs, Q, C = q.shape
    pos_rows = None
    if (FOLD_QUERY_POS and query_pos is not None and q.is_cuda and q.dtype == torch.float32 and fused_mode()
            and not torch.is_grad_enabled() and query_pos.shape == query.shape and C % 64 == 0 and q.is_contiguous()):
        pb = query_pos.transpose(0, 1)                       # [bs, Q, C]
        if (pb.stride(0) == 0 or bs == 1) and pb.stride(2) == 1:
            pos_rows = pb[0]                                 # [Q, C] view of the embedding parameter
    if pos_rows is None:
        w, b = mod._cat_proj(frames)
        if query_pos is not None:
            q = batch_first(query + query_pos)
        return linear_rows(q.reshape(bs * Q, C), w, b)
    w, b = mod._cat_proj(frames)
    base = pos_rows._base if pos_rows._base is not None else pos_rows
    from .bricks import get_gemm_mode
    key = SourceKey((base, w, b), extra=(tuple(pos_rows.shape), tuple(pos_rows.stride()), pos_rows.storage_offset(),
                                        frames, get_gemm_mode()))
    hit = mod.__dict__.get('_pave_pos_proj')
    if hit is None or hit[0] != key:
        with torch.no_grad():
            N = w.shape[0]
            Np = (N + 127) // 128 * 128
            wp = torch.zeros((Np, w.shape[1]), dtype=torch.float32, device=w.device)
            wp[:N] = w
            tab = torch.zeros((Q, Np), dtype=torch.float32, device=w.device)
            tab[:, :N] = (pos_rows.double() @ w.double().t() + b.double()).float()
            planes = ops.split_weight_bf16x3(wp, 3)          # (a few hundred query rows: always the exact split)
        hit = mod.__dict__['_pave_pos_proj'] = (key, planes, tab)
    return ops.gemm_bf16x3_ex(q.reshape(bs * Q, C), hit[1], None, hit[2], residual_rows=Q)[0]


def _proj(linear, x):
    """nn.Linear on [..., K] through bricks.linear_rows (split GEMM when enabled)."""
    if x.is_cuda and x.is_contiguous():
        y = linear_rows(x.reshape(-1, x.shape[-1]), linear.weight, linear.bias)
        return y.view(x.shape[:-1] + (y.shape[-1],))
    return linear(x)


PADDED_FAST_PATH = True   # A/B switch (tools/debug_padded.py): False = the masked_fill / separate-GEMM formulation


def masked_row_index(mask):
    """int32 indices of the True entries of a (cached) padding mask, flattened row-major -- built ONCE per mask
    tensor (one device read-back when a padded batch shape is first seen) and kept on the tensor that owns the
    storage; the masks themselves are cached per batch shape by the head."""
    base = mask._base if mask._base is not None else mask
    key = SourceKey((base,), (tuple(mask.shape), tuple(mask.stride()), mask.storage_offset()))
    slots = base.__dict__.setdefault('_pave_mask_rows', [])
    for k, idx in slots:
        if k == key:
            return idx
    with torch.no_grad():
        idx = mask.reshape(-1).nonzero().flatten().to(torch.int32)
    if len(slots) > 8:
        del slots[:]
    slots.append((key, idx))
    return idx


def project_values_hoisted(attns, value_bf, key_padding_mask=None, masked_rows=None):
    """value_proj of several decoder layers over the same (constant) memory, as the modules'
    own `project_value` would give them: [B*T, S, 8, 32] per layer.  In the split GEMM modes two
    layers share one launch (N = 512, two dense outputs), so the memory is read once per pair.
    With a padding mask (both T-frame attentions mask the memory BEFORE value_proj, OT:1706-1711 /
    MO:1454-1458: a masked token's value is value_proj.bias) the launches are the un-masked ones and
    the masked rows -- ~1 % of the tokens -- are overwritten with the layer's bias afterwards
    (pave_fill_rows_f32), instead of a masked_fill copy of the memory per layer."""
    from .bricks import _split_weight, get_gemm_mode, split_gemm_ok
    x = value_bf
    fill = None
    if PADDED_FAST_PATH and key_padding_mask is not None and x.is_cuda and x.is_contiguous() and len(attns) >= 2 \
            and not torch.is_grad_enabled() and key_padding_mask.dtype == torch.bool \
            and all(type(a).project_value in (MulFramesMultiScaleDeformablePoseAttention.project_value,
                                              MulFramesMultiScaleDeformableAttention.project_value)
                    and a.value_proj.bias is not None for a in attns) \
            and split_gemm_ok(x.reshape(-1, x.shape[-1]), attns[0].value_proj.weight):
        # (masked_rows: the caller's cached index of the mask's True entries, heads.MaskList.masked_rows)
        fill = masked_rows if masked_rows is not None else masked_row_index(key_padding_mask)
        key_padding_mask = None
    if key_padding_mask is not None or not (x.is_cuda and x.is_contiguous()) or len(attns) < 2:
        return [a.project_value(value_bf, key_padding_mask) for a in attns]
    rows = x.reshape(-1, x.shape[-1])
    outs = []
    i = 0
    while i < len(attns):
        a = attns[i]
        b = attns[i + 1] if i + 1 < len(attns) else None
        if b is not None and a.value_proj.out_features == b.value_proj.out_features \
                and a.value_proj.out_features % 128 == 0 \
                and split_gemm_ok(rows, a.value_proj.weight):
            srcs = (a.value_proj.weight, a.value_proj.bias, b.value_proj.weight, b.value_proj.bias)
            key = SourceKey(srcs)
            if getattr(a, '_pair_key', None) != key:
                with torch.no_grad():
                    a._pair_w = torch.cat([a.value_proj.weight, b.value_proj.weight], 0).contiguous()
                    a._pair_b = torch.cat([a.value_proj.bias, b.value_proj.bias], 0).contiguous()
                a._pair_key = key
            nv = a.value_proj.out_features
            v1, v2 = ops.gemm_bf16x3_ex(rows, _split_weight(a._pair_w), a._pair_b, n_split=nv,
                                        fp16=get_gemm_mode() == 'fp16')
            for m, v in ((a, v1), (b, v2)):
                if fill is not None:
                    ops.fill_rows_(v, fill, m.value_proj.bias.detach())
                outs.append(v.view(-1, x.shape[-2], m.num_heads, nv // m.num_heads))
            i += 2
        else:
            v = a.project_value(value_bf, None)
            if fill is not None:
                ops.fill_rows_(v.view(-1, v.shape[-2] * v.shape[-1]), fill, a.value_proj.bias.detach())
            outs.append(v)
            i += 1
    return outs


def _fused_ok(mod, *tensors):
    """The fused kernels are built for 8 heads x 32 channels, fp32, on the device."""
    return (mod.embed_dims == 256 and mod.num_heads == 8
            and all(t.is_cuda and t.dtype == torch.float32 for t in tensors))


def _host_levels(spatial_shapes, kwargs):
    """Level sizes as host ints without a device sync when the caller provides them."""
    hs = kwargs.get('spatial_shapes_host')
    if hs is not None:
        return [(int(h), int(w)) for h, w in hs]
    return [(int(h), int(w)) for h, w in spatial_shapes.tolist()]


# ---------------------------------------------------------------------------
@MMCV_ATTENTION.register_module()
class MultiScaleDeformableAttention(BaseModule, _CatProj):
    """Encoder self-attention (MO:207-412)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        _check_heads(embed_dims, num_heads)
        self.norm_cfg = norm_cfg
        self.dropout = nn.Identity()  # inference path
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.prepare_in_gemm = True   # A/B switch: softmax / locations in the merged GEMM's epilogue
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        constant_init(self.sampling_offsets, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid_init.view(-1))
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    def _cat_sources(self):
        return [self.sampling_offsets], [self.attention_weights]

    supports_post_norm = True
    supports_query_plus_pos = True

    def _tile_shift(self):
        """Per-(head, level) LDS-window shift of the tile kernel = the rounded mean of the
        sampling-offset bias over the 4 points (the reference initialises that bias on a ray,
        MO:227-240); host ints, re-read only when the bias changes."""
        b = self.sampling_offsets.bias
        key = SourceKey((b,))
        if getattr(self, '_tile_shift_key', None) != key:
            self._tile_shift_val = ops.enc_tile_window_shift(b)
            self._tile_shift_key = key
        return self._tile_shift_val

    def _merged_proj(self):
        """[value_proj ; sampling_offsets ; attention_weights] as one [640, 256] operand (and the
        [384, 256] offsets / logits part alone), rebuilt when a source parameter changes."""
        w_cat, b_cat = self._cat_proj()
        srcs = (self.value_proj.weight, self.value_proj.bias, w_cat, b_cat)
        key = SourceKey(srcs)
        if getattr(self, '_merged_key', None) != key:
            with torch.no_grad():
                self._merged_w = torch.cat([self.value_proj.weight, w_cat], 0).contiguous()
            self._merged_key = key
        return self._merged_w, w_cat, b_cat

    def _forward_merged(self, q, pos_row, reference_points, tile_levels, out_half=False):
        """Self-attention over a frame batch whose positional encoding is the same for every
        frame (no padding): (q + pos) W^T = q W^T + (pos W^T), so value_proj and the offsets /
        logits Linears read the layer input ONCE, as one N = 640 GEMM whose epilogue adds the
        per-token table [b_v | pos W_cat^T + b_cat] (row m -> token m % S) and writes value and
        projections as two dense matrices.  No `query + query_pos` pass, no second read of q."""
        from .bricks import _split_weight, fused_mode, get_gemm_mode
        bs, S, C = q.shape
        w_all, w_cat, b_cat = self._merged_proj()
        nv = self.value_proj.out_features
        # the per-token epilogue table depends on the positional table and this layer's weights only
        src = pos_row._base if pos_row._base is not None else pos_row
        # (two same-shaped slices of one base tensor are different tables: the view's offset and
        # strides are part of the key)
        tkey = SourceKey([src, self.value_proj.bias, w_cat, b_cat],
                         (tuple(pos_row.shape), pos_row.storage_offset(), tuple(pos_row.stride()), S))
        if getattr(self, '_table_key', None) == tkey:
            table = self._table
        else:
            table = torch.empty((S, w_all.shape[0]), dtype=torch.float32, device=q.device)
            table[:, :nv] = self.value_proj.bias
            torch.addmm(b_cat, pos_row, w_cat.t(), out=table[:, nv:])
            self._table_key, self._table = tkey, table
        ref = reference_points.reshape(1, bs * S, self.num_levels, 2)
        if not ref.is_contiguous():
            ref = ref.contiguous()
        if fused_mode() and self.prepare_in_gemm and nv == 256 and w_all.shape[0] == 640:
            # the sampler's softmax / location arithmetic runs in this GEMM's epilogue (its waves are
            # ~25 % VALU-active, the sampler is VALU-bound): same code, same bits (pave_enc_math.h)
            v, samp = ops.gemm_bf16x3_encproj(q.reshape(bs * S, C), _split_weight(w_all), table, ref,
                                              tile_levels, value_bias=self.value_proj.bias.detach())
            out = ops.deform_attn_enc_tile(v.view(bs, S, self.num_heads, -1), samp, None,
                                           levels_hw=tile_levels, window_shift=self._tile_shift(),
                                           prepared=True, out_half=out_half)
            return out.view(bs, S, self.embed_dims)
        v, proj = ops.gemm_bf16x3_ex(q.reshape(bs * S, C), _split_weight(w_all), None, table,
                                     residual_rows=S, n_split=nv,
                                     fp16=get_gemm_mode() == 'fp16')
        out = ops.deform_attn_enc_tile(v.view(bs, S, self.num_heads, -1), proj, ref,
                                       levels_hw=tile_levels, window_shift=self._tile_shift())
        return out.view(bs, S, self.embed_dims)

    def _forward_merged_groups(self, q, pos_bf, mask, reference_points, tile_levels, groups, masked_rows=None):
        """`_forward_merged` for a PADDED batch: the positional encoding and the padding pattern are
        the same for the frames of a run (`groups`: (first frame, count); the T frames of a clip share one
        valid size, HEAD:429-445), so each run is ONE launch of the merged projection GEMM with the run's own
        epilogue table, writing its row slice of the batch's value / prepared-projection matrices; the value
        rows of masked tokens are zeroed afterwards (the mask FOLLOWS value_proj here, MO:369-371;
        pave_fill_rows_f32 on the ~1 % masked rows) and ONE sampler launch serves the whole batch.  The
        reference points carry the per-frame valid ratios (transformer.get_reference_points)."""
        from .bricks import _split_weight
        bs, S, C = q.shape
        w_all, w_cat, b_cat = self._merged_proj()
        nv = self.value_proj.out_features
        src = pos_bf._base if pos_bf._base is not None else pos_bf
        tabs = self.__dict__.setdefault('_pave_group_tables', {})
        value = torch.empty((bs * S, nv), dtype=torch.float32, device=q.device)
        samp = torch.empty((bs * S, w_all.shape[0] - nv), dtype=torch.float32, device=q.device)
        ref = reference_points.reshape(bs * S, self.num_levels, 2)
        if not ref.is_contiguous():
            ref = ref.contiguous()
        rows = q.reshape(bs * S, C)
        wp = _split_weight(w_all)
        vb = self.value_proj.bias.detach()
        for f0, n in groups:
            pos_row = pos_bf[f0]
            tkey = SourceKey([src, self.value_proj.bias, w_cat, b_cat],
                             (tuple(pos_row.shape), pos_row.storage_offset(), tuple(pos_row.stride()), S))
            hit = tabs.get(f0)
            if hit is None or hit[0] != tkey:
                table = torch.empty((S, w_all.shape[0]), dtype=torch.float32, device=q.device)
                table[:, :nv] = self.value_proj.bias
                torch.addmm(b_cat, pos_row, w_cat.t(), out=table[:, nv:])
                if len(tabs) > 64:
                    tabs.clear()
                hit = tabs[f0] = (tkey, table)
            r0, r1 = f0 * S, (f0 + n) * S
            ops.gemm_bf16x3_encproj(rows[r0:r1], wp, hit[1], ref[r0:r1], tile_levels, value_bias=vb,
                                    out=(value[r0:r1], samp[r0:r1]))
        ops.fill_rows_(value, masked_rows if masked_rows is not None else masked_row_index(mask), None)
        out = ops.deform_attn_enc_tile(value.view(bs, S, self.num_heads, -1), samp, None,
                                       levels_hw=tile_levels, window_shift=self._tile_shift(), prepared=True)
        return out.view(bs, S, self.embed_dims)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, post_norm=None, query_plus_pos=None, **kwargs):
        self_value = value is None or value is query
        if value is None:
            value = query
        if identity is None:
            identity = query
        tile_levels = kwargs.get('tile_levels')
        groups = kwargs.get('frame_groups')
        if (PADDED_FAST_PATH and groups is not None and self_value and query_pos is not None and query_plus_pos is None
                and not self.batch_first and key_padding_mask is not None and tile_levels is not None
                and kwargs.get('memory_clip_index') is None and reference_points.shape[-1] == 2
                and self.num_levels == 4 and self.num_points == 4 and query_pos.dim() == 3
                and query_pos.shape == query.shape and key_padding_mask.dtype == torch.bool
                and self.prepare_in_gemm and not torch.is_grad_enabled()):
            from .bricks import fused_mode, split_gemm_ok
            q = batch_first(query)
            pos_bf = batch_first(query_pos)
            if (fused_mode() and _fused_ok(self, q) and self.value_proj.out_features == 256
                    and self.value_proj.bias is not None and pos_bf.stride(2) == 1 and q.is_contiguous()
                    and all(split_gemm_ok(q[f0:f0 + n].reshape(-1, q.shape[-1]), self.sampling_offsets.weight)
                            for f0, n in groups)):
                out = self._forward_merged_groups(q, pos_bf, key_padding_mask, reference_points,
                                                  tile_levels, groups, kwargs.get('masked_rows'))
                idt = batch_first(identity)
                out = linear_residual_norm(out, self.output_proj, idt, post_norm,
                                           inplace=kwargs.get('inplace_residual', False) is True)
                return seq_first_view(out)
        if (self_value and query_pos is not None and query_plus_pos is None and not self.batch_first
                and key_padding_mask is None and tile_levels is not None
                and kwargs.get('memory_clip_index') is None and reference_points.shape[-1] == 2
                and self.num_levels == 4 and self.num_points == 4 and query_pos.dim() == 3
                and (query_pos.stride(1) == 0 or query_pos.shape[1] == 1)
                and not torch.is_grad_enabled()):
            # query_pos [S, bs, C] expanded over the frame axis: one positional table for all frames
            from .bricks import split_gemm_ok
            q = batch_first(query)
            if _fused_ok(self, q) and split_gemm_ok(q.reshape(-1, q.shape[-1]),
                                                    self.sampling_offsets.weight):
                from . import bricks
                idt = batch_first(identity)
                # fp16 mode: the sampled rows only ever feed output_proj's MFMA, which rounds them to fp16 at operand
                # fetch -- written AS fp16 by the sampler they are the same values at half the bytes (bricks.FFN does
                # the same with its hidden activation)
                half = (bricks.get_gemm_mode() == 'fp16' and bricks.FP16_ACTIVATIONS and self.prepare_in_gemm
                        and q.shape[0] * q.shape[1] >= 65536 and post_norm is not None and idt.is_contiguous()
                        and tuple(post_norm.normalized_shape) == (256,) and post_norm.weight is not None
                        and post_norm.bias is not None and self.value_proj.out_features == 256)
                out = self._forward_merged(q, query_pos[:, 0], reference_points, tile_levels, out_half=half)
                if half and out.dtype == torch.float16:
                    idt2 = idt.reshape(-1, 256)
                    inplace = kwargs.get('inplace_residual', False) is True
                    t = ops.gemm_fp16_act(out.reshape(-1, 256), bricks._split_weight(self.output_proj.weight),
                                          self.output_proj.bias, residual=idt2,
                                          ln=(post_norm.weight, post_norm.bias, post_norm.eps),
                                          out=idt2 if inplace else None)
                    return seq_first_view(t.view(idt.shape))
                out = linear_residual_norm(out, self.output_proj, idt, post_norm,
                                           inplace=kwargs.get('inplace_residual', False) is True)
                return seq_first_view(out)
        if query_plus_pos is not None:   # `query + query_pos`, already made by the producer
            query = query_plus_pos
        elif query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            q, v = batch_first(query), batch_first(value)
        else:
            q, v = query, value
        bs, num_query, _ = q.shape
        num_value = v.shape[1]
        v = _proj(self.value_proj, v)
        if key_padding_mask is not None:
            v = v.masked_fill(key_padding_mask[..., None], 0.0)
        v = v.view(v.shape[0], num_value, self.num_heads, -1)
        if self_value and q.is_cuda and not torch.is_grad_enabled() and num_query >= 4096:
            note_slow_path('encoder layer off the merged-projection path (separate value / offset GEMMs)')
        fused = (_fused_ok(self, q, v) and self.num_levels == 4 and self.num_points == 4
                 and reference_points.shape[-1] == 2)
        if fused:
            w, b = self._cat_proj()
            proj = linear_rows(q.reshape(bs * num_query, self.embed_dims), w, b)
            ref = reference_points.reshape(1, bs * num_query, self.num_levels, 2)
            if not ref.is_contiguous():
                ref = ref.contiguous()
            clip_index = kwargs.get('memory_clip_index')
            tile_levels = kwargs.get('tile_levels')
            unit_clip = None
            if clip_index is not None:  # queries of pose n read the memory of image clip_index[n]
                unit_clip = clip_index.to(torch.int32).repeat_interleave(num_query)
            if tile_levels is not None and clip_index is None and num_query == num_value:
                # encoder self-attention over a halving pyramid: LDS-tile kernel
                out = ops.deform_attn_enc_tile(v, proj, ref, levels_hw=tile_levels,
                                               window_shift=self._tile_shift())
            else:
                if clip_index is None and num_query == num_value and num_query >= 4096:
                    note_slow_path('encoder self-attention on the direct-gather sampler (no LDS-tile launch)')
                out = ops.deform_attn_grid_fused(
                    v, spatial_shapes, level_start_index, proj, ref, T=1, n_clips=v.shape[0],
                    units_per_clip=num_query, unit_clip=unit_clip,
                    order=kwargs.get('unit_order'))
            out = out.view(bs, num_query, self.embed_dims)
        else:
            if q.is_cuda:
                note_slow_path('MultiScaleDeformableAttention: un-fused softmax / locations + generic sampler')
            off = self.sampling_offsets(q).view(bs, num_query, self.num_heads, self.num_levels,
                                                self.num_points, 2)
            aw = self.attention_weights(q).view(bs, num_query, self.num_heads,
                                                self.num_levels * self.num_points).softmax(-1)
            aw = aw.view(bs, num_query, self.num_heads, self.num_levels, self.num_points)
            if kwargs.get('memory_clip_index') is not None:
                v = v[kwargs['memory_clip_index']]  # generic path: one slab per pose
            if reference_points.shape[-1] == 2:
                norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
                loc = reference_points[:, :, None, :, None, :] + \
                    off / norm[None, None, None, :, None, :]
            elif reference_points.shape[-1] == 4:
                loc = reference_points[:, :, None, :, None, :2] + \
                    off / self.num_points * reference_points[:, :, None, :, None, 2:] * 0.5
            else:
                raise ValueError(f'Last dim of reference_points must be 2 or 4, but get '
                                 f'{reference_points.shape[-1]} instead.')
            out = ops.MultiScaleDeformableAttnFunction.apply(
                v.contiguous(), spatial_shapes, level_start_index, loc.contiguous(),
                aw.contiguous(), self.im2col_step)
        idt = identity if self.batch_first else batch_first(identity)
        out = linear_residual_norm(out, self.output_proj, idt, post_norm,
                                   inplace=kwargs.get('inplace_residual', False) is True)
        return out if self.batch_first else seq_first_view(out)


# ---------------------------------------------------------------------------
class MulFramesMultiScaleDeformablePoseAttention(BaseModule, _CatProj):
    """Pose-aware T-frame cross-attention, any odd T (generalises OT:1543-1863, 2738-3117).

    forward: query [Q, B, C]; value [S, B*T, C] frame-interleaved (clip b, frame t at b*T+t);
    key_padding_mask [B*T, S] applied BEFORE value_proj (OT:1706-1711);
    reference_points [B, T*Q, L, 2K] frame-major along dim 1.
    """

    def __init__(self, num_frames=3, embed_dims=256, num_heads=8, num_levels=4, num_points=17,
                 im2col_step=64, dropout=0.1, norm_cfg=None, init_cfg=None, batch_first=False):
        super().__init__(init_cfg)
        _check_heads(embed_dims, num_heads)
        self.norm_cfg = norm_cfg
        self.dropout = nn.Identity()
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.num_frames = num_frames
        self.frame_prefixes = frame_prefixes(num_frames)
        for fp in self.frame_prefixes:
            setattr(self, fp + 'sampling_offsets',
                    nn.Linear(embed_dims, num_heads * num_levels * num_points * 2))
            setattr(self, fp + 'attention_weights',
                    nn.Linear(embed_dims, num_heads * num_levels * num_points))
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        for fp in self.frame_prefixes:
            constant_init(getattr(self, fp + 'sampling_offsets'), 0.)
            constant_init(getattr(self, fp + 'attention_weights'), val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)

    def _cat_sources(self):
        return ([getattr(self, fp + 'sampling_offsets') for fp in self.frame_prefixes],
                [getattr(self, fp + 'attention_weights') for fp in self.frame_prefixes])

    def project_value(self, value_bf, key_padding_mask=None):
        """value_bf [B*T, S, C] -> [B*T, S, 8, 32]: padding mask, then value_proj."""
        if key_padding_mask is not None:
            value_bf = value_bf.masked_fill(key_padding_mask[..., None], 0.0)
        v = _proj(self.value_proj, value_bf)
        return v.view(v.shape[0], v.shape[1], self.num_heads, -1)

    supports_post_norm = True

    def forward(self, query, key=None, value=None, residual=None, query_pos=None,
                query_time_pos=None, key_padding_mask=None, reference_points=None,
                spatial_shapes=None, level_start_index=None, post_norm=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        inp_residual = query if residual is None else residual
        query_nopos = query
        if query_pos is not None and (self.batch_first or kwargs.get('frame_shard') is not None
                                      or not query.is_cuda or torch.is_grad_enabled()):
            query, query_nopos = query + query_pos, None
        T, M, L, K = self.num_frames, self.num_heads, self.num_levels, self.num_points
        q = batch_first(query) if not self.batch_first else query
        bs, num_query, _ = q.shape
        if reference_points.shape[-1] != K * 2:
            raise ValueError(f'Last dim of reference_points must be 2K, but get '
                             f'{reference_points.shape[-1]} instead.')
        projected = kwargs.get('value_projected')
        if projected is not None:
            v = projected  # [B*T, S, 8, 32], hoisted by the transformer
        else:
            vb = batch_first(value) if not self.batch_first else value
            v = self.project_value(vb, key_padding_mask)
        shard = kwargs.get('frame_shard')
        if shard is not None:
            out = self._forward_sharded(q, v, reference_points, spatial_shapes,
                                        level_start_index, shard)
            idt = inp_residual if self.batch_first else batch_first(inp_residual)
            out = linear_residual_norm(out, self.output_proj, idt, post_norm)
            return out if self.batch_first else seq_first_view(out)
        # streaming: `v` is a per-frame cache of projected values and frame t of clip b is slab
        # table[b * T + t] (pavenet_amd/streaming.py) -- no per-window copy or re-projection
        table = kwargs.get('value_frame_table')
        assert table is not None or v.shape[0] == bs * T, 'value must hold num_frames slabs per clip'
        if query_nopos is not None:     # (+ query_pos in the GEMM's epilogue table where it is a per-query constant)
            proj = cat_proj_rows(self, query_nopos, query_pos)
        else:
            w, b = self._cat_proj()
            proj = linear_rows(q.reshape(bs * num_query, self.embed_dims), w, b)
        if _fused_ok(self, q, v) and L <= 4 and K <= 24:
            # (inference: a level axis that is a broadcast is passed as such, ops._level_rows)
            ref = reference_points.contiguous() if torch.is_grad_enabled() else reference_points
            stats = kwargs.get('return_softmax_stats', False)
            res = ops.deform_attn_pose_fused(
                v if v.is_contiguous() else v.contiguous(), spatial_shapes, level_start_index,
                proj, ref, T=T, n_clips=bs, num_query=num_query, num_keypoints=K,
                return_stats=stats, frame_table=table)
            if stats:  # frame-sharded multi-GPU: caller merges partial rows, then projects
                return res
            out = res.view(bs, num_query, self.embed_dims)
        else:
            if q.is_cuda:
                note_slow_path('pose T-frame attention: un-fused per-frame sampler launches')
            if table is not None:
                v = v[table.long()]
            out = self._unfused(v, proj, reference_points, spatial_shapes, level_start_index, bs,
                                num_query)
        idt = inp_residual if self.batch_first else batch_first(inp_residual)
        out = linear_residual_norm(out, self.output_proj, idt, post_norm)
        return out if self.batch_first else seq_first_view(out)

    def _forward_sharded(self, q, v, reference_points, spatial_shapes, level_start_index, shard):
        """Frame-sharded multi-GPU: v holds only this rank's frames [bs*T_loc, S, 8, 32]; the
        partial row + per-head (max, sum-exp) of the local frames are merged across ranks with
        one all-gather (pavenet_amd/dist.py)."""
        from . import dist as pdist
        T, L, K = self.num_frames, self.num_levels, self.num_points
        bs, num_query, _ = q.shape
        Tl = shard.n_local
        assert _fused_ok(self, q, v) and L <= 4 and K <= 24, 'sharded path needs the fused kernel'
        if Tl > 0:
            assert v.shape[0] == bs * Tl
            w, b = self._cat_proj(frames=tuple(shard.local))
            proj = linear_rows(q.reshape(bs * num_query, self.embed_dims), w, b)
            ref = reference_points.view(bs, T, num_query, L, 2 * K)[:, shard.local].reshape(
                bs, Tl * num_query, L, 2 * K).contiguous()
            row, smax, ssum = ops.deform_attn_pose_fused(
                v if v.is_contiguous() else v.contiguous(), spatial_shapes, level_start_index,
                proj, ref, T=Tl, n_clips=bs, num_query=num_query, num_keypoints=K,
                return_stats=True)
        else:
            row = q.new_zeros(bs * num_query, self.embed_dims)
            smax = q.new_full((bs * num_query, self.num_heads), float('-inf'))
            ssum = q.new_zeros(bs * num_query, self.num_heads)
        out = pdist.all_gather_merge(row, smax, ssum, group=shard.group)
        return out.view(bs, num_query, self.embed_dims)

    def _unfused(self, v, proj, reference_points, spatial_shapes, level_start_index, bs, nq):
        """Shapes the fused kernel does not cover: per-frame launches of the generic sampler
        with the stabilised joint softmax computed in torch."""
        T, M, L, K = self.num_frames, self.num_heads, self.num_levels, self.num_points
        n_off = T * M * L * K * 2
        off = proj[:, :n_off].view(bs, nq, T, M, L, K, 2)
        lg = proj[:, n_off:n_off + n_off // 2].view(bs, nq, T, M, L * K)   # (proj may be wider: zero-padded columns)
        aw = lg.permute(0, 1, 3, 2, 4).reshape(bs, nq, M, T * L * K).softmax(-1)
        aw = aw.view(bs, nq, M, T, L, K)
        out = 0
        for t in range(T):
            rp_t = reference_points[:, t * nq:(t + 1) * nq]
            rp = rp_t.reshape(bs, nq, L, -1, 2).unsqueeze(2)
            x1 = rp_t[..., 0::2].min(-1, keepdim=True)[0]
            y1 = rp_t[..., 1::2].min(-1, keepdim=True)[0]
            x2 = rp_t[..., 0::2].max(-1, keepdim=True)[0]
            y2 = rp_t[..., 1::2].max(-1, keepdim=True)[0]
            wh = torch.cat([torch.clamp(x2 - x1, min=1e-4), torch.clamp(y2 - y1, min=1e-4)],
                           -1)[:, :, None, :, None, :]
            loc = rp + off[:, :, t] * wh * 0.5
            out = out + ops.MultiScaleDeformableAttnFunction.apply(
                v[t::T].contiguous(), spatial_shapes, level_start_index, loc.contiguous(),
                aw[:, :, :, t].contiguous(), self.im2col_step)
        return out


@ATTENTION.register_module()
class MulFramesMultiScaleDeformablePoseAttentionNumFrames3(MulFramesMultiScaleDeformablePoseAttention):
    """OT:1543-1863."""

    def __init__(self, num_frames=3, **kwargs):
        assert num_frames == 3
        super().__init__(num_frames=3, **kwargs)


@ATTENTION.register_module()
class MulFramesMultiScaleDeformablePoseAttentionNumFrames5(MulFramesMultiScaleDeformablePoseAttention):
    """OT:2738-3117 (its ctor has no num_frames argument)."""

    def __init__(self, **kwargs):
        kwargs.pop('num_frames', None)
        super().__init__(num_frames=5, **kwargs)


ATTENTION.register_module(name='MulFramesMultiScaleDeformablePoseAttention',
                          module=MulFramesMultiScaleDeformablePoseAttention)


@ATTENTION.register_module()
class MultiScaleDeformablePoseAttention(MulFramesMultiScaleDeformablePoseAttention):
    """Single-frame pose attention of PETR (OT:251-427): the T = 1 case, except that the
    padding mask is applied AFTER value_proj (OT:386-388)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=17, im2col_step=64,
                 dropout=0.1, norm_cfg=None, init_cfg=None, batch_first=False):
        super().__init__(num_frames=1, embed_dims=embed_dims, num_heads=num_heads,
                         num_levels=num_levels, num_points=num_points, im2col_step=im2col_step,
                         dropout=dropout, norm_cfg=norm_cfg, init_cfg=init_cfg,
                         batch_first=batch_first)

    def project_value(self, value_bf, key_padding_mask=None):
        v = _proj(self.value_proj, value_bf)
        if key_padding_mask is not None:
            v = v.masked_fill(key_padding_mask[..., None], 0.0)
        return v.view(v.shape[0], v.shape[1], self.num_heads, -1)


# ---------------------------------------------------------------------------
class MulFramesMultiScaleDeformableAttention(BaseModule, _CatProj):
    """Joint-decoder T-frame cross-attention, any odd T (generalises MO:1268-1587, 1590-1981).

    Reference calling convention: query [K, N, C]; value [S, N, T, C] with the clip memory
    replicated once per pose (OT:21498) and re-projected N times; key_padding_mask [N, T, S]
    applied BEFORE value_proj (MO:1454-1458); reference_points [T*N, K, L, 2] frame-major.

    Native convention (no replication): pass the clip memory once as value [S, B, T, C] (B clips)
    plus ``memory_clip_index`` int tensor [N] (clip of each pose) and key_padding_mask [B, T, S];
    or pass ``value_projected`` [B*T, S, 8, 32].  A replicated value whose pose dimension is a
    stride-0 expand is de-duplicated automatically.
    """

    def __init__(self, num_frames=3, embed_dims=256, num_heads=8, num_levels=4, num_points=4,
                 im2col_step=64, dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
        _check_heads(embed_dims, num_heads)
        self.norm_cfg = norm_cfg
        self.dropout = nn.Identity()
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.num_frames = num_frames
        self.frame_prefixes = frame_prefixes(num_frames)
        for fp in self.frame_prefixes:
            setattr(self, fp + 'sampling_offsets',
                    nn.Linear(embed_dims, num_heads * num_levels * num_points * 2))
            setattr(self, fp + 'attention_weights',
                    nn.Linear(embed_dims, num_heads * num_levels * num_points))
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid_init[:, :, i, :] *= i + 1
        for fp in self.frame_prefixes:
            constant_init(getattr(self, fp + 'sampling_offsets'), 0.)
            with torch.no_grad():  # own storage per frame (the reference aliases one tensor)
                getattr(self, fp + 'sampling_offsets').bias.copy_(grid_init.view(-1))
            constant_init(getattr(self, fp + 'attention_weights'), val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    def _cat_sources(self):
        return ([getattr(self, fp + 'sampling_offsets') for fp in self.frame_prefixes],
                [getattr(self, fp + 'attention_weights') for fp in self.frame_prefixes])

    def project_value(self, memory_bt, key_padding_mask=None):
        """memory_bt [B, T, S, C] (+ mask [B, T, S]) -> [B*T, S, 8, 32]."""
        if key_padding_mask is not None:
            memory_bt = memory_bt.masked_fill(key_padding_mask[..., None], 0.0)
        v = _proj(self.value_proj, memory_bt)
        B, T, S, _ = v.shape
        return v.view(B * T, S, self.num_heads, -1)

    supports_post_norm = True

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                query_time_pos=None, key_padding_mask=None, reference_points=None,
                spatial_shapes=None, level_start_index=None, post_norm=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        query_nopos = query
        if query_pos is not None and (self.batch_first or kwargs.get('frame_shard') is not None
                                      or not query.is_cuda or torch.is_grad_enabled()):
            query, query_nopos = query + query_pos, None
        T, M, L, P = self.num_frames, self.num_heads, self.num_levels, self.num_points
        q = batch_first(query) if not self.batch_first else query
        N, num_query, _ = q.shape
        clip_index = kwargs.get('memory_clip_index')
        projected = kwargs.get('value_projected')
        shard = kwargs.get('frame_shard')
        Tv = T if shard is None else max(shard.n_local, 1)  # frames present in `value`
        table = kwargs.get('value_frame_table')   # streaming: per-frame cache + frame table
        if projected is not None:
            v = projected
            n_clips = (v.shape[0] if table is None else table.numel()) // Tv
            if clip_index is None:
                assert n_clips == 1 or n_clips == N
                clip_index = torch.arange(N, device=q.device) if n_clips == N and N > 1 else \
                    torch.zeros(N, dtype=torch.long, device=q.device)
        else:
            # value arrives [S, n, T, C] (seq-first) or [n, S, T, C] (batch_first)
            vb = value.permute(1, 2, 0, 3) if not self.batch_first else value.permute(0, 2, 1, 3)
            n = vb.shape[0]                                     # vb: [n, T, S, C]
            mask = key_padding_mask
            if clip_index is None and n == N and N > 1 and vb.stride(0) == 0:
                vb = vb[:1]                                     # stride-0 replica -> one clip
                mask = mask[:1] if mask is not None else None
                n = 1
            if clip_index is None:
                assert n in (1, N), 'value must carry one slab set per pose or per clip'
                clip_index = torch.arange(N, device=q.device) if (n == N and N > 1) else \
                    torch.zeros(N, dtype=torch.long, device=q.device)
            assert vb.shape[1] == Tv
            v = self.project_value(vb, mask)
            n_clips = n
        if reference_points.shape[-1] != 2:
            raise ValueError('MulFrames joint attention is built for 2-d reference points '
                             f'(got last dim {reference_points.shape[-1]})')
        if shard is not None:
            from . import dist as pdist
            Tl = shard.n_local
            assert _fused_ok(self, q, v) and L == 4 and P == 4
            if Tl > 0:
                w, b = self._cat_proj(frames=tuple(shard.local))
                proj = linear_rows(q.reshape(N * num_query, self.embed_dims), w, b)
                ref = reference_points.reshape(T, N * num_query, L, 2)[shard.local].contiguous()
                unit_clip = clip_index.to(torch.int32).repeat_interleave(num_query)
                row, smax, ssum = ops.deform_attn_grid_fused(
                    v if v.is_contiguous() else v.contiguous(), spatial_shapes,
                    level_start_index, proj, ref, T=Tl, n_clips=v.shape[0] // Tl,
                    units_per_clip=num_query, unit_clip=unit_clip, return_stats=True)
            else:
                row = q.new_zeros(N * num_query, self.embed_dims)
                smax = q.new_full((N * num_query, M), float('-inf'))
                ssum = q.new_zeros(N * num_query, M)
            out = pdist.all_gather_merge(row, smax, ssum, group=shard.group)
            out = out.view(N, num_query, self.embed_dims)
            idt = identity if self.batch_first else batch_first(identity)
            out = linear_residual_norm(out, self.output_proj, idt, post_norm)
            return out if self.batch_first else seq_first_view(out)
        if query_nopos is not None:     # (+ query_pos in the GEMM's epilogue table where it is a per-query constant)
            proj = cat_proj_rows(self, query_nopos, query_pos)
        else:
            w, b = self._cat_proj()
            proj = linear_rows(q.reshape(N * num_query, self.embed_dims), w, b)
        ref = reference_points.reshape(T, N * num_query, L, 2)
        if _fused_ok(self, q, v) and L == 4 and P == 4:
            hit = self.__dict__.get('_pave_unit_clip')   # (the head passes one cached index tensor)
            ukey = SourceKey((clip_index,), num_query)   # (identity AND version: an index rewritten in place)
            if hit is None or hit[0] != ukey:
                hit = (ukey, clip_index.to(torch.int32).repeat_interleave(num_query))
                self.__dict__['_pave_unit_clip'] = hit
            unit_clip = hit[1]
            out = ops.deform_attn_grid_fused(
                v if v.is_contiguous() else v.contiguous(), spatial_shapes, level_start_index,
                proj, ref.contiguous() if torch.is_grad_enabled() else ref, T=T, n_clips=n_clips,
                units_per_clip=num_query, unit_clip=unit_clip, frame_table=table)
            out = out.view(N, num_query, self.embed_dims)
        else:
            if q.is_cuda:
                note_slow_path('joint T-frame attention: un-fused per-frame sampler launches')
            if table is not None:
                v = v[table.long()]
            out = self._unfused(v, proj, ref, clip_index, spatial_shapes, level_start_index, N,
                                num_query)
        idt = identity if self.batch_first else batch_first(identity)
        out = linear_residual_norm(out, self.output_proj, idt, post_norm)
        return out if self.batch_first else seq_first_view(out)

    def _unfused(self, v, proj, ref, clip_index, spatial_shapes, level_start_index, N, nq):
        T, M, L, P = self.num_frames, self.num_heads, self.num_levels, self.num_points
        n_off = T * M * L * P * 2
        off = proj[:, :n_off].view(N, nq, T, M, L, P, 2)
        lg = proj[:, n_off:n_off + n_off // 2].view(N, nq, T, M, L * P)    # (proj may be wider: zero-padded columns)
        aw = lg.permute(0, 1, 3, 2, 4).reshape(N, nq, M, T * L * P).softmax(-1)
        aw = aw.view(N, nq, M, T, L, P)
        norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        out = 0
        for t in range(T):
            rp = ref[t].view(N, nq, L, 2)
            loc = rp[:, :, None, :, None, :] + off[:, :, t] / norm[None, None, None, :, None, :]
            vt = v[clip_index * T + t]
            out = out + ops.MultiScaleDeformableAttnFunction.apply(
                vt.contiguous(), spatial_shapes, level_start_index, loc.contiguous(),
                aw[:, :, :, t].contiguous(), self.im2col_step)
        return out


@MMCV_ATTENTION.register_module()
class MulFramesMultiScaleDeformableAttentionNumFrames3(MulFramesMultiScaleDeformableAttention):
    """MO:1268-1587."""

    def __init__(self, num_frames=3, **kwargs):
        assert num_frames == 3
        super().__init__(num_frames=3, **kwargs)


@MMCV_ATTENTION.register_module()
class MulFramesMultiScaleDeformableAttentionNumFrames5(MulFramesMultiScaleDeformableAttention):
    """MO:1590-1981 (its ctor has no num_frames argument)."""

    def __init__(self, **kwargs):
        kwargs.pop('num_frames', None)
        super().__init__(num_frames=5, **kwargs)


MMCV_ATTENTION.register_module(name='MulFramesMultiScaleDeformableAttention',
                               module=MulFramesMultiScaleDeformableAttention)
