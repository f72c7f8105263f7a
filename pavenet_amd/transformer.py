"""Decoders and transformers of the PAVE-Net forward path (same registry names / kwargs /
state-dict keys as the reference).

* ``opera.VideoPoseTransformerDecoderV2`` / ``V2_1``           OT:6661-6753 / 6757-6852   (a6)
* ``mmcv.DeformableDetrTransformerDecoderV1`` / ``V1_2``       MT:794-886 / 889-986       (a8)
* ``opera.VideoPoseTransformerMulFrames``                     OT:20986-21536             (a4, a9)
* ``opera.PetrTransformerDecoder``, ``mmcv.DeformableDetrTransformerDecoder``  (a15)

MT = third_party/mmdetection/mmdet/models/utils/transformer.py, OT = opera/models/utils/transformer.py.

Device-side differences from the reference (results identical): tensors are kept batch-first
contiguous under sequence-first views; constant-per-shape tensors (reference grids, level
start indices, XCD unit order) are cached; the clip memory is never replicated per pose; the
value projections of all decoder layers are hoisted out of the layer loop when asked.
"""

import torch
import torch.nn as nn

from .bricks import (BaseModule, SourceKey, TransformerLayerSequence, batch_first, inverse_sigmoid,
                     linear_norm, mlp_rows,
                     seq_first_view, xavier_init)
from .deform_attn import (MulFramesMultiScaleDeformableAttention,
                          MulFramesMultiScaleDeformablePoseAttention,
                          MultiScaleDeformableAttention, frame_prefixes,
                          project_values_hoisted)
from .locality import encoder_unit_order
from .registry import (MMCV_TRANSFORMER, MMCV_TRANSFORMER_LAYER_SEQUENCE, TRANSFORMER,
                       TRANSFORMER_LAYER_SEQUENCE, build_transformer_layer_sequence)

_REF_BRANCH_KW = ('pre_pre_', 'pre_', '', 'next_', 'next_next_')


def _collect_frame_branches(T, kwargs, suffix):
    """Per-frame branch lists from the reference's kwarg names (pre_pre_X, pre_X, X, next_X,
    next_next_X) or from the generic ``frame_<suffix>`` list (any T)."""
    generic = kwargs.pop('frame_' + suffix, None)
    named = {p: kwargs.pop(p + suffix, None) for p in _REF_BRANCH_KW}
    if generic is not None:
        assert len(generic) == T
        return list(generic)
    if named[''] is None:
        return None
    out = [named.get(p) for p in frame_prefixes(T)]
    if any(b is None for b in out):
        return None
    return out


def _ref_update(tmp, ref):
    """(tmp + inverse_sigmoid(ref)).sigmoid(): one HIP launch on the device."""
    if tmp.is_cuda and tmp.dtype == torch.float32 and not torch.is_grad_enabled() \
            and tmp.shape == ref.shape:
        from . import ops
        return ops.ref_update(tmp, ref)
    return (tmp + inverse_sigmoid(ref)).sigmoid()


def _stack_buffers(query, reference_points, n_layers, return_intermediate):
    """Preallocated [levels, ...] stacks of a decoder's intermediate states / reference points (device inference):
    every layer's last launch writes its level in place, so `torch.stack` (a copy launch per stack on the
    latency-bound tail) is not needed.  states: batch-first storage [levels, bs, Q, C], handed out as the
    sequence-first view the reference's stack has; -> (states | None, refs | None)."""
    if not (return_intermediate and query.is_cuda and query.dtype == torch.float32 and not torch.is_grad_enabled()
            and reference_points.dtype == torch.float32):
        return None, None
    Q, bs, C = query.shape
    states = torch.empty((n_layers, bs, Q, C), dtype=torch.float32, device=query.device)
    refs = torch.empty((n_layers,) + tuple(reference_points.shape), dtype=torch.float32, device=query.device)
    return states, refs


def _into_level(buf, lid, value_seq_first=None, value=None):
    """Level `lid` of a preallocated stack holds `value` (copied only if the producing launch did not write there)."""
    if value_seq_first is not None:
        v = value_seq_first.transpose(0, 1)                  # batch-first view
        if v.data_ptr() != buf[lid].data_ptr() or v.stride() != buf[lid].stride():
            buf[lid].copy_(v)
        return buf[lid].transpose(0, 1)
    if value.data_ptr() != buf[lid].data_ptr() or value.stride() != buf[lid].stride():
        buf[lid].copy_(value)
    return buf[lid]


def _frame_branches(branches, lid, x, cat_dim, update_ref=None, out=None):
    """torch.cat([b[lid](x) for b in branches], dim=cat_dim) for T per-frame MLPs of identical
    structure (Linear / ReLU chains, OT:6728-6732, MT:861-864).  On the device the T first Linears
    run as ONE GEMM over the row-concatenated weights and the following per-frame Linears as
    batched GEMMs (T x fewer launches in the launch-bound decoders); same arithmetic per element."""
    mods = [b[lid] for b in branches]
    T = len(mods)
    m0 = mods[0]
    ok = (T > 1 and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
          and x.dim() == 3 and isinstance(m0, nn.Sequential) and len(m0) >= 3 and len(m0) % 2 == 1
          and all(isinstance(m, nn.Sequential) and len(m) == len(m0) for m in mods)
          and all(isinstance(m[i], nn.Linear if i % 2 == 0 else nn.ReLU)
                  and (i % 2 == 1 or (m[i].weight.shape == m0[i].weight.shape
                                      and m[i].bias is not None))
                  for m in mods for i in range(len(m0))))
    if not ok:
        y = torch.cat([m(x) for m in mods], dim=cat_dim)
        return y if update_ref is None else _ref_update(y, update_ref)
    lins = [[m[i] for m in mods] for i in range(0, len(m0), 2)]      # [layer][frame]
    srcs = [p for layer in lins for l in layer for p in (l.weight, l.bias)]
    key = SourceKey(srcs)
    cache = m0.__dict__.get('_pave_stacked')
    if cache is None or cache[0] != key:
        with torch.no_grad():
            w1 = torch.cat([l.weight for l in lins[0]], 0).t().contiguous()          # [C, T*h]
            b1 = torch.cat([l.bias for l in lins[0]], 0).contiguous()
            rest = [(torch.stack([l.weight.t() for l in layer], 0).contiguous(),   # [T, in, out]
                     torch.stack([l.bias for l in layer], 0)[:, None].contiguous())
                    for layer in lins[1:]]
        cache = (key, w1, b1, rest)
        m0.__dict__['_pave_stacked'] = cache
    _, w1, b1, rest = cache
    lead = x.shape[:-1]
    rows = x.reshape(-1, x.shape[-1])
    R = rows.shape[0]
    from .bricks import _GEMM, fused_mode, get_gemm_mode
    mode_planes = lambda: 3     # noqa: E731  (per-frame branch MLPs on a few hundred query rows: always exact)
    dims = [l[0].weight.shape for l in lins]                      # [(out, in)] per layer
    # (any number of rows: below a few thousand frame-rows -- a one-clip batch -- this used to fall to batched
    # vendor GEMMs, the `Cijk_*` rows of a T = 3 one-clip step's trace)
    if (fused_mode() and rows.is_contiguous()
            and dims[0][1] % 64 == 0 and (T * dims[0][0]) % 64 == 0
            and all(d[1] % 64 == 0 for d in dims[1:])      # split_weight_bf16x3 (un-padded): K % 64 == 0
            and all(d[0] % 64 == 0 for d in dims[1:-1]) and dims[-1][0] % 2 == 0):
        # the exact 3-plane split kernels: ONE GEMM for the T first Linears, then one GROUPED
        # launch per following layer (group t = frame t's Linear on its own column block)
        from . import ops
        gp = m0.__dict__.get('_pave_grouped')
        gmode = get_gemm_mode()
        if gp is None or gp[0] != key or gp[3] != gmode:
            with torch.no_grad():
                planes = [ops.split_weight_bf16x3(torch.cat([l.weight for l in lins[0]], 0).contiguous(),
                                                  mode_planes())]
                biases = [b1]
                for li, layer in enumerate(lins[1:]):
                    o = layer[0].weight.shape[0]
                    op = (o + 63) // 64 * 64          # the last layer (2K outputs) is padded per group
                    wcat = torch.zeros((T, op, layer[0].weight.shape[1]), device=x.device)
                    bcat = torch.zeros((T, op), device=x.device)
                    for t, l in enumerate(layer):
                        wcat[t, :o] = l.weight
                        bcat[t, :o] = l.bias
                    planes.append(ops.split_weight_bf16x3(wcat.flatten(0, 1).contiguous(), mode_planes()))
                    biases.append(bcat.flatten().contiguous())
            gp = (key, planes, biases, gmode)
            m0.__dict__['_pave_grouped'] = gp
        _, planes, biases, _ = gp
        y = ops.gemm_bf16x3(rows, planes[0], biases[0], None, relu=True)          # [R, T*h]
        for li in range(1, len(lins)):
            last = li + 1 == len(lins)
            o = dims[li][0]
            y = ops.gemm_bf16x3_grouped(y, planes[li], biases[li], (o + 63) // 64 * 64, relu=not last)
        o = dims[-1][0]
        if update_ref is not None and update_ref.is_contiguous() and update_ref.dtype == torch.float32 \
                and update_ref.shape[-1] == o and update_ref.numel() == R * T * o:
            # update_ref given: return sigmoid(cat_t(branch_t(x)) + inverse_sigmoid(update_ref)), read
            # straight from the grouped output (no layout copy, one launch)
            return ops.ref_update_frames(y, update_ref, T, o, lead[1] if cat_dim == 1 else R, out=out)
        y = y.view(R, T, -1)[:, :, :o].permute(1, 0, 2)            # [T, R, out]
        y = y.reshape((T,) + tuple(lead) + (o,))
        if cat_dim == 0:
            y = y.reshape((T * lead[0],) + tuple(lead[1:]) + (o,))
        else:
            assert cat_dim == 1
            y = y.permute(1, 0, 2, 3).reshape(lead[0], T * lead[1], o)
        return y if update_ref is None else _ref_update(y, update_ref)
    y = torch._addmm_activation(b1, rows, w1)                    # relu(x W1^T + b1), all frames
    y = y.view(R, T, -1).transpose(0, 1)                          # [T, R, h]
    for li, (w, b) in enumerate(rest):
        y = torch.baddbmm(b, y, w)
        if li + 1 < len(rest):
            y = torch.relu_(y)
    y = y.view((T,) + tuple(lead) + (y.shape[-1],))               # [T, d0, d1, out]
    if cat_dim == 0:
        y = y.reshape((T * lead[0],) + tuple(lead[1:]) + (y.shape[-1],))
    else:
        assert cat_dim == 1
        y = y.permute(1, 0, 2, 3).reshape(lead[0], T * lead[1], y.shape[-1])
    return y if update_ref is None else _ref_update(y, update_ref)


# ---------------------------------------------------------------------------
class VideoPoseTransformerDecoderMulFrames(TransformerLayerSequence):
    """Pose decoder, any odd T (generalises OT:6661-6753 and 6757-6852)."""

    def __init__(self, *args, return_intermediate=False, num_keypoints=17, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.num_keypoints = num_keypoints

    def forward(self, query, *args, reference_points=None, valid_ratios=None, **kwargs):
        K = self.num_keypoints
        T = getattr(self.layers[0].attentions[-1], 'num_frames', 1)
        branches = _collect_frame_branches(T, kwargs, 'kpt_branches')
        projected = kwargs.pop('values_projected', None)  # optional: one per layer
        # un-padded clips: every valid ratio is exactly 1.0 and x * 1.0 == x, so the scaled
        # reference is a broadcast view instead of two elementwise launches per layer
        unit_ratios = kwargs.pop('unit_valid_ratios', False)
        output = query
        intermediate, intermediate_reference_points = [], []
        states, refs = _stack_buffers(query, reference_points, len(self.layers), self.return_intermediate)
        for lid, layer in enumerate(self.layers):
            if states is not None:
                kwargs['layer_out'] = states[lid]
            if reference_points.shape[-1] == K * 2 and unit_ratios:
                reference_points_input = reference_points[:, :, None].expand(
                    -1, -1, valid_ratios.shape[1], -1)
            elif reference_points.shape[-1] == K * 2:
                reference_points_input = reference_points[:, :, None] * \
                    valid_ratios.repeat(1, 1, K)[:, None]
            else:
                assert reference_points.shape[-1] == 2
                reference_points_input = reference_points[:, :, None] * valid_ratios[:, None]
            if projected is not None:
                kwargs['value_projected'] = projected[lid]
            output = layer(output, *args, reference_points=reference_points_input, **kwargs)
            output = output.permute(1, 0, 2)
            if branches is not None:
                if reference_points.shape[-1] != K * 2:
                    raise NotImplementedError
                reference_points = _frame_branches(branches, lid, output, 1,       # OT:6728-6735
                                                   update_ref=reference_points,
                                                   out=refs[lid] if refs is not None else None)
            output = output.permute(1, 0, 2)
            if states is not None:
                output = _into_level(states, lid, value_seq_first=output)
                reference_points = _into_level(refs, lid, value=reference_points)
            elif self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if states is not None:
            return states.transpose(1, 2), refs
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points


TRANSFORMER_LAYER_SEQUENCE.register_module(name='VideoPoseTransformerDecoderV2',
                                           module=VideoPoseTransformerDecoderMulFrames)
TRANSFORMER_LAYER_SEQUENCE.register_module(name='VideoPoseTransformerDecoderV2_1',
                                           module=VideoPoseTransformerDecoderMulFrames, force=True)
TRANSFORMER_LAYER_SEQUENCE.register_module(name='VideoPoseTransformerDecoderMulFrames',
                                           module=VideoPoseTransformerDecoderMulFrames, force=True)


class DeformableDetrTransformerDecoderMulFrames(TransformerLayerSequence):
    """Joint decoder, any odd T (generalises MT:794-886 and 889-986).

    reference_points [T*N, K, 2] frame-major on dim 0; valid_ratios [N*T, L, 2]."""

    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate

    def forward(self, query, *args, reference_points=None, valid_ratios=None, **kwargs):
        T = getattr(self.layers[0].attentions[-1], 'num_frames', 1)
        branches = _collect_frame_branches(T, kwargs, 'reg_branches')
        projected = kwargs.pop('values_projected', None)
        unit_ratios = kwargs.pop('unit_valid_ratios', False)   # (see the pose decoder)
        output = query
        intermediate, intermediate_reference_points = [], []
        states, refs = _stack_buffers(query, reference_points, len(self.layers), self.return_intermediate)
        for lid, layer in enumerate(self.layers):
            if states is not None:
                kwargs['layer_out'] = states[lid]
            if unit_ratios and reference_points.shape[-1] == 2:
                reference_points_input = reference_points[:, :, None].expand(
                    -1, -1, valid_ratios.shape[1], -1)
            elif reference_points.shape[-1] == 4:
                reference_points_input = reference_points[:, :, None] * \
                    torch.cat([valid_ratios, valid_ratios], -1)[:, None]
            else:
                assert reference_points.shape[-1] == 2
                reference_points_input = reference_points[:, :, None] * valid_ratios[:, None]
            if projected is not None:
                kwargs['value_projected'] = projected[lid]
            output = layer(output, *args, reference_points=reference_points_input, **kwargs)
            output = output.permute(1, 0, 2)
            if branches is not None:
                assert reference_points.shape[-1] == 2
                reference_points = _frame_branches(branches, lid, output, 0,       # MT:861-866
                                                   update_ref=reference_points,
                                                   out=refs[lid] if refs is not None else None)
            output = output.permute(1, 0, 2)
            if states is not None:
                output = _into_level(states, lid, value_seq_first=output)
                reference_points = _into_level(refs, lid, value=reference_points)
            elif self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if states is not None:
            return states.transpose(1, 2), refs
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points


MMCV_TRANSFORMER_LAYER_SEQUENCE.register_module(
    name='DeformableDetrTransformerDecoderV1', module=DeformableDetrTransformerDecoderMulFrames)
MMCV_TRANSFORMER_LAYER_SEQUENCE.register_module(
    name='DeformableDetrTransformerDecoderV1_2', module=DeformableDetrTransformerDecoderMulFrames,
    force=True)
MMCV_TRANSFORMER_LAYER_SEQUENCE.register_module(
    name='DeformableDetrTransformerDecoderMulFrames',
    module=DeformableDetrTransformerDecoderMulFrames, force=True)


@MMCV_TRANSFORMER_LAYER_SEQUENCE.register_module()
class DeformableDetrTransformerDecoder(TransformerLayerSequence):
    """Stock single-frame refine decoder (MT:705-791); reference points are detached there,
    which is a no-op at inference."""

    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate

    def forward(self, query, *args, reference_points=None, valid_ratios=None, reg_branches=None,
                **kwargs):
        output = query
        intermediate, intermediate_reference_points = [], []
        for lid, layer in enumerate(self.layers):
            if reference_points.shape[-1] == 4:
                reference_points_input = reference_points[:, :, None] * \
                    torch.cat([valid_ratios, valid_ratios], -1)[:, None]
            else:
                assert reference_points.shape[-1] == 2
                reference_points_input = reference_points[:, :, None] * valid_ratios[:, None]
            output = layer(output, *args, reference_points=reference_points_input, **kwargs)
            output = output.permute(1, 0, 2)
            if reg_branches is not None:
                tmp = mlp_rows(reg_branches[lid], output)
                if reference_points.shape[-1] == 4:
                    new_reference_points = (tmp + inverse_sigmoid(reference_points)).sigmoid()
                else:
                    new_reference_points = tmp.clone()
                    new_reference_points[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points)
                    new_reference_points = new_reference_points.sigmoid()
                reference_points = new_reference_points.detach()
            output = output.permute(1, 0, 2)
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class PetrTransformerDecoder(TransformerLayerSequence):
    """PETR pose decoder (OT:4148-4231): single frame, references detached."""

    def __init__(self, *args, return_intermediate=False, num_keypoints=17, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.num_keypoints = num_keypoints

    def forward(self, query, *args, reference_points=None, valid_ratios=None, kpt_branches=None,
                **kwargs):
        K = self.num_keypoints
        output = query
        intermediate, intermediate_reference_points = [], []
        for lid, layer in enumerate(self.layers):
            if reference_points.shape[-1] == K * 2:
                reference_points_input = reference_points[:, :, None] * \
                    valid_ratios.repeat(1, 1, K)[:, None]
            else:
                assert reference_points.shape[-1] == 2
                reference_points_input = reference_points[:, :, None] * valid_ratios[:, None]
            output = layer(output, *args, reference_points=reference_points_input, **kwargs)
            output = output.permute(1, 0, 2)
            if kpt_branches is not None:
                tmp = mlp_rows(kpt_branches[lid], output)
                if reference_points.shape[-1] == K * 2:
                    reference_points = (tmp + inverse_sigmoid(reference_points)).sigmoid().detach()
                else:
                    raise NotImplementedError
            output = output.permute(1, 0, 2)
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points


# ---------------------------------------------------------------------------
@MMCV_TRANSFORMER.register_module()
class Transformer(BaseModule):
    """mmdet Transformer base (MT:533-576): builds ``encoder`` and ``decoder``."""

    def __init__(self, encoder=None, decoder=None, init_cfg=None):
        super().__init__(init_cfg=init_cfg)
        self.encoder = build_transformer_layer_sequence(encoder)
        self.decoder = build_transformer_layer_sequence(decoder)
        self.embed_dims = self.encoder.embed_dims

    def init_weights(self):
        for m in self.modules():
            if hasattr(m, 'weight') and m.weight is not None and m.weight.dim() > 1:
                xavier_init(m, distribution='uniform')
        self._is_init = True


# processing order of encoder tokens (pavenet_amd/locality.py)
UNIT_ORDER_MODE = 'band'   # module attribute (tests / tools set it); not read from the environment


class _LevelGeometry:
    """Host-side description of the flattened multi-level token sequence (cached per shape)."""

    def __init__(self, hw_list, device):
        self.hw = [(int(h), int(w)) for h, w in hw_list]
        self.spatial_shapes = torch.as_tensor(self.hw, dtype=torch.long, device=device)
        starts = [0]
        for h, w in self.hw[:-1]:
            starts.append(starts[-1] + h * w)
        self.starts = starts
        self.level_start_index = torch.as_tensor(starts, dtype=torch.long, device=device)
        self.S = sum(h * w for h, w in self.hw)
        self._order = {}
        # un-padded batches: tensors that depend only on the level sizes and the batch size
        # (valid ratios = 1, encoder reference grid, two-stage proposal grid), built once
        self.unpadded = {}

    def unit_order(self, n_frames, device):
        if n_frames not in self._order:
            self._order[n_frames] = encoder_unit_order(self.hw, n_frames, UNIT_ORDER_MODE).to(device)
        return self._order[n_frames]

    def tile_levels(self):
        """Level sizes if the LDS-tile encoder kernel covers this pyramid, else None."""
        from .ops import enc_tile_supported
        return self.hw if enc_tile_supported(self.hw) else None


@TRANSFORMER.register_module()
class VideoPoseTransformerMulFrames(Transformer):
    """OT:20986-21536."""

    def __init__(self, hm_encoder=None, refine_decoder=None, as_two_stage=True,
                 num_feature_levels=4, two_stage_num_proposals=300, num_keypoints=17,
                 num_frames=3, **kwargs):
        super().__init__(**kwargs)
        self.as_two_stage = as_two_stage
        self.num_feature_levels = num_feature_levels
        self.two_stage_num_proposals = two_stage_num_proposals
        self.embed_dims = self.encoder.embed_dims
        self.num_keypoints = num_keypoints
        self.num_frames = num_frames
        # hm_encoder is accepted and ignored, as in the reference (OT:21041 commented out)
        self.refine_decoder = build_transformer_layer_sequence(refine_decoder)
        self.init_layers()
        self._geom = {}
        self.hoist_value_proj = True
        self.xcd_unit_order = True
        # encoder sampling through the LDS-tile kernel (pave_enc_tile.hip); False = head-major
        # direct-gather kernel (also the path for pyramids the tile kernel does not cover)
        self.enc_lds_tile = True
        # proposal stage on the fused launches (ops.gather_rows_add / proposal_refs_, the per-clip Linear +
        # LayerNorm with filled border rows); False = the tensor expressions of OT:21204-21418 (tools/ab_switch.py)
        self.fused_proposal_stage = True

    def init_layers(self):
        self.level_embeds = nn.Parameter(torch.Tensor(self.num_feature_levels, self.embed_dims))
        if self.as_two_stage:
            self.enc_output = nn.Linear(self.embed_dims, self.embed_dims)
            self.enc_output_norm = nn.LayerNorm(self.embed_dims)
            self.refine_query_embedding = nn.Embedding(self.num_keypoints, self.embed_dims * 2)
        else:
            self.reference_points = nn.Linear(self.embed_dims, 2 * self.num_keypoints)

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MultiScaleDeformableAttention):  # OT:21084-21089
                m.init_weights()
        if not self.as_two_stage:
            xavier_init(self.reference_points, distribution='uniform', bias=0.)
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.refine_query_embedding.weight)
        self._is_init = True

    # -- helpers (OT:21095-21216) -------------------------------------------
    def geometry(self, hw_list, device):
        key = (tuple((int(h), int(w)) for h, w in hw_list), str(device))
        if key not in self._geom:
            self._geom[key] = _LevelGeometry(hw_list, device)
        return self._geom[key]

    def gen_encoder_output_proposals(self, memory, memory_padding_mask, geom, mask_key=None):
        """mask_key: identity of a PADDED batch's (cached) mask set -- the proposal grid, its validity and the rows to
        blank are functions of the masks alone, built once per mask set instead of ~60 tensor launches per step."""
        N, S, C = memory.shape
        dev = memory.device
        cached = None
        if memory_padding_mask is None and memory.is_cuda and not torch.is_grad_enabled():
            cached = geom.unpadded.get(('proposals', N))
            if cached is not None:   # the grid depends on the level sizes only
                output_proposals, valid = cached
                filled = self._output_memory_filled(memory, valid, geom) if self.fused_proposal_stage else None
                if filled is not None:
                    return filled, output_proposals
                output_memory = memory.masked_fill(~valid, float(0))
                output_memory = linear_norm(output_memory, self.enc_output, self.enc_output_norm)
                return output_memory, output_proposals
        if memory_padding_mask is not None and mask_key is not None and memory.is_cuda \
                and not torch.is_grad_enabled() and self.fused_proposal_stage:
            hit = geom.unpadded.get(('proposals_padded', N))
            if hit is not None and hit[0] == mask_key:
                _, output_proposals, keep = hit
                filled = self._output_memory_filled(memory, keep, geom, tag=('padded', N), key_extra=mask_key)
                if filled is not None:
                    return filled, output_proposals
        proposals = []
        _cur = 0
        for lvl, (H, W) in enumerate(geom.hw):
            if memory_padding_mask is not None:
                m = memory_padding_mask[:, _cur:(_cur + H * W)].view(N, H, W, 1)
                valid_H = torch.sum(~m[:, :, 0, 0], 1)
                valid_W = torch.sum(~m[:, 0, :, 0], 1)
            else:
                valid_H = torch.full((N,), H, device=dev)
                valid_W = torch.full((N,), W, device=dev)
            grid_y, grid_x = torch.meshgrid(
                torch.linspace(0, H - 1, H, dtype=torch.float32, device=dev),
                torch.linspace(0, W - 1, W, dtype=torch.float32, device=dev), indexing='ij')
            grid = torch.cat([grid_x.unsqueeze(-1), grid_y.unsqueeze(-1)], -1)
            scale = torch.cat([valid_W.unsqueeze(-1), valid_H.unsqueeze(-1)], 1).view(N, 1, 1, 2)
            grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
            proposals.append(grid.view(N, -1, 2))
            _cur += H * W
        output_proposals = torch.cat(proposals, 1)
        valid = ((output_proposals > 0.01) & (output_proposals < 0.99)).all(-1, keepdim=True)
        output_proposals = torch.log(output_proposals / (1 - output_proposals))
        if memory_padding_mask is not None:
            output_proposals = output_proposals.masked_fill(
                memory_padding_mask.unsqueeze(-1), float('inf'))
        output_proposals = output_proposals.masked_fill(~valid, float('inf'))
        if memory_padding_mask is None and memory.is_cuda and not torch.is_grad_enabled():
            geom.unpadded[('proposals', N)] = (output_proposals, valid)
            # (the very first call takes the same launches as every later one: the per-clip Linear + LayerNorm and the
            # batch-wide one may pick different LayerNorm-epilogue forms, which differ in the last bits)
            filled = self._output_memory_filled(memory, valid, geom) if self.fused_proposal_stage else None
            if filled is not None:
                return filled, output_proposals
        if memory_padding_mask is not None and mask_key is not None and memory.is_cuda \
                and not torch.is_grad_enabled() and self.fused_proposal_stage:
            # rows that stay: valid proposal AND not padding (OT:21204-21214 blanks the memory rows of both)
            keep = valid & ~memory_padding_mask.unsqueeze(-1)
            geom.unpadded[('proposals_padded', N)] = (mask_key, output_proposals, keep)
            filled = self._output_memory_filled(memory, keep, geom, tag=('padded', N), key_extra=mask_key)
            if filled is not None:
                return filled, output_proposals
        output_memory = memory
        if memory_padding_mask is not None:
            output_memory = output_memory.masked_fill(memory_padding_mask.unsqueeze(-1), float(0))
        output_memory = output_memory.masked_fill(~valid, float(0))
        output_memory = linear_norm(output_memory, self.enc_output, self.enc_output_norm)
        return output_memory, output_proposals

    def _dense_const(self, name, source, view, rows=None):
        """A dense copy of `view` (a slice / broadcast of the parameter `source`), made once per version of the
        parameter instead of once per step; rows: the leading broadcast size of the copy."""
        if torch.is_grad_enabled():      # (training: the copy is part of the graph, nothing is cached)
            return view.contiguous()
        key = SourceKey((source,), extra=rows)
        hit = self.__dict__.get('_pave_' + name)
        if hit is None or hit[0] != key:
            hit = (key, view.detach().contiguous())
            self.__dict__['_pave_' + name] = hit
        return hit[1]

    def _output_memory_filled(self, memory, valid, geom, tag=None, key_extra=None):
        """enc_output_norm(enc_output(memory.masked_fill(~valid, 0))) (OT:21206-21214) without the masked copy
        of the memory: Linear + LayerNorm per clip straight from the (strided) centre-frame rows, then the rows of
        the invalid proposals -- the level borders, ~5 % of the tokens -- overwritten with the value a zero row
        gets (the same for all of them: LayerNorm(bias)), computed by the same kernel from a block of zero rows."""
        from .bricks import linear_norm_fused_ok
        N, S, C = memory.shape
        lin, norm = self.enc_output, self.enc_output_norm
        if not (memory.dtype == torch.float32 and memory.stride(2) == 1 and memory.stride(1) == C
                and linear_norm_fused_ok(memory[0], lin, norm)):
            return None
        from . import ops
        from .bricks import get_gemm_mode
        # (key_extra: the mask set of a padded batch -- `valid` then also excludes its padded tokens)
        key = SourceKey([lin.weight, lin.bias, norm.weight, norm.bias], extra=get_gemm_mode())
        slot = ('proposal_fill', N) if tag is None else ('proposal_fill',) + tuple(tag)
        const = geom.unpadded.get(slot)
        if const is None or const[0] != key or (key_extra is not None and const[3] != key_extra):
            zrows = linear_norm(memory.new_zeros((max(S, 1), C)), lin, norm)   # (the same launch form as a clip)
            rows = (~valid.reshape(N * S)).nonzero().flatten().to(torch.int32)
            const = (key, zrows[0].clone(), rows, key_extra)
            geom.unpadded[slot] = const
        out = torch.empty((N, S, norm.normalized_shape[0]), dtype=torch.float32, device=memory.device)
        for b in range(N):
            linear_norm(memory[b], lin, norm, out=out[b])
        if const[2].numel():
            ops.fill_rows_(out.view(N * S, -1), const[2], const[1])
        return out

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        reference_points_list = []
        for lvl, (H, W) in enumerate(spatial_shapes):
            H, W = int(H), int(W)
            ref_y, ref_x = torch.meshgrid(
                torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device),
                torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device),
                indexing='ij')
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
            reference_points_list.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(reference_points_list, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    @staticmethod
    def get_valid_ratio(mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    def _frame_branch_lists(self, kwargs, suffix):
        """Reference kwarg names -> per-frame list (or generic frame_<suffix>)."""
        return _collect_frame_branches(self.num_frames, kwargs, suffix)

    # -- encoder over independent frames (OT:21277-21322) ----------------------
    @staticmethod
    def _flat_view(mlvl_feats):
        """[n, S, C] tensor aliasing the levels when they already are consecutive slices of one
        token-major buffer (necks.ChannelMapper._forward_flat), else None."""
        f0 = mlvl_feats[0]
        n, C = f0.shape[:2]
        S = sum(f.shape[2] * f.shape[3] for f in mlvl_feats)
        off = 0
        for f in mlvl_feats:
            h, w = f.shape[2:]
            if f.dtype != f0.dtype or f.stride() != (S * C, 1, w * C, C) or \
                    f.data_ptr() != f0.data_ptr() + off * C * f0.element_size():
                return None
            off += h * w
        return f0.as_strided((n, S, C), (S * C, C, 1))

    def encode_frames(self, mlvl_feats, mlvl_masks, mlvl_pos_embeds, has_padding=True):
        """Flatten levels and run the encoder.  Frames are independent here, so the result can
        be cached per frame (pavenet_amd.streaming) or computed on another rank (frame sharding).
        -> (memory [n_frames, S, C], mask_flatten, valid_ratios [n_frames, L, 2], geometry)."""
        dev = mlvl_feats[0].device
        geom = self.geometry([f.shape[-2:] for f in mlvl_feats], dev)
        feat_flatten = [feat.flatten(2).transpose(1, 2) for feat in mlvl_feats]
        flat = self._flat_view(mlvl_feats) if feat_flatten[0].is_cuda else None
        feat_flatten = flat if flat is not None else torch.cat(feat_flatten, 1)  # [n, S, C]
        # flattened masks and `pos + level_embed`: functions of the (cached) mask / sine tensors and
        # one parameter -- rebuilt only when one of those objects changes
        key = SourceKey(list(mlvl_masks) + list(mlvl_pos_embeds) + [self.level_embeds])
        const = geom.unpadded.get('flat_pos') if (feat_flatten.is_cuda
                                                   and not torch.is_grad_enabled()) else None
        if const is not None and const[0] == key:
            mask_flatten, lvl_pos_embed_flatten = const[1], const[2]
        else:
            mask_flatten, lvl_pos_embed_flatten = [], []
            for lvl, (mask, pos_embed) in enumerate(zip(mlvl_masks, mlvl_pos_embeds)):
                mask_flatten.append(mask.flatten(1))
                pos_embed = pos_embed.flatten(2).transpose(1, 2)
                lvl_pos_embed_flatten.append(pos_embed + self.level_embeds[lvl].view(1, 1, -1))
            mask_flatten = torch.cat(mask_flatten, 1)
            lvl_pos_embed_flatten = torch.cat(lvl_pos_embed_flatten, 1)
            if feat_flatten.is_cuda and not torch.is_grad_enabled():
                geom.unpadded['flat_pos'] = (key, mask_flatten, lvl_pos_embed_flatten)
        if lvl_pos_embed_flatten.shape[0] != feat_flatten.shape[0]:    # shared across frames
            lvl_pos_embed_flatten = lvl_pos_embed_flatten.expand(feat_flatten.shape[0], -1, -1)
        spatial_shapes, level_start_index = geom.spatial_shapes, geom.level_start_index
        nfr = feat_flatten.shape[0]
        const = geom.unpadded.get(('refs', nfr)) if (not has_padding and feat_flatten.is_cuda) else None
        if const is not None:       # no padding: valid ratios are exactly 1, the grid is constant
            valid_ratios, reference_points = const
        else:
            pkey = SourceKey(list(mlvl_masks), extra=nfr)
            phit = geom.unpadded.get(('refs_padded', nfr)) if (feat_flatten.is_cuda
                                                               and not torch.is_grad_enabled()) else None
            if phit is not None and phit[0] == pkey:    # (a padded batch's masks are cached per batch shape)
                valid_ratios, reference_points = phit[1], phit[2]
            else:
                valid_ratios = torch.stack([self.get_valid_ratio(m) for m in mlvl_masks], 1)
                if valid_ratios.shape[0] != nfr:
                    valid_ratios = valid_ratios.expand(nfr, -1, -1)
                reference_points = self.get_reference_points(geom.hw, valid_ratios, device=dev)
                if feat_flatten.is_cuda and not torch.is_grad_enabled():
                    if has_padding:
                        geom.unpadded[('refs_padded', nfr)] = (pkey, valid_ratios, reference_points)
                    else:
                        geom.unpadded[('refs', nfr)] = (valid_ratios.contiguous(), reference_points)
        attn_mask = mask_flatten if has_padding else None
        if attn_mask is not None and attn_mask.shape[0] != feat_flatten.shape[0]:
            attn_mask = attn_mask.expand(feat_flatten.shape[0], -1)
        bs = feat_flatten.shape[0]
        extra = {}
        if self.enc_lds_tile and feat_flatten.is_cuda and geom.tile_levels() is not None:
            extra['tile_levels'] = geom.tile_levels()
            groups = getattr(mlvl_masks, 'frame_groups', None)
            if attn_mask is not None and groups is not None and sum(n for _, n in groups) == bs:
                # padded batch: runs of frames with one positional table / padding pattern each
                extra['frame_groups'] = groups
                extra['masked_rows'] = mlvl_masks.masked_rows()
        elif self.xcd_unit_order and feat_flatten.is_cuda:
            extra['unit_order'] = geom.unit_order(bs, dev)
        # every encoder layer input is a temporary owned by this function, so the residual
        # GEMMs may accumulate into it (saves a copy of the 0.6 GB activation per GEMM)
        extra['inplace_residual'] = True
        extra['input_is_shared'] = flat is not None   # aliases the neck output: keep it intact
        memory = self.encoder(
            query=seq_first_view(feat_flatten), key=None, value=None,
            query_pos=seq_first_view(lvl_pos_embed_flatten), query_key_padding_mask=attn_mask,
            spatial_shapes=spatial_shapes, reference_points=reference_points,
            level_start_index=level_start_index, valid_ratios=valid_ratios, **extra)
        return batch_first(memory), mask_flatten, valid_ratios, geom

    # -- a4: forward (OT:21218-21456) ----------------------------------------
    def forward(self, mlvl_feats, mlvl_masks, query_embed, mlvl_pos_embeds, cls_branches=None,
                sigma_branches=None, has_padding=True, frame_shard=None, **kwargs):
        """frame_shard (pavenet_amd.dist.FrameShard): mlvl_feats hold only this rank's frames
        ([B*T_loc, C, h, w]); the encoder runs on them, the centre-frame proposals are broadcast
        from their owner, and the T-frame attentions merge per-rank partial rows."""
        assert self.as_two_stage or query_embed is not None
        T = self.num_frames
        Tl = T if frame_shard is None else frame_shard.n_local
        branches = self._frame_branch_lists(kwargs, 'kpt_branches')
        kpt_branches = branches[T // 2] if branches is not None else None
        encoded = kwargs.pop('encoded', None)
        if encoded is None:
            encoded = self.encode_frames(mlvl_feats, mlvl_masks, mlvl_pos_embeds, has_padding)
        memory, mask_flatten, valid_ratios, geom = encoded
        dev = memory.device
        bs = memory.shape[0]
        spatial_shapes, level_start_index = geom.spatial_shapes, geom.level_start_index
        attn_mask = mask_flatten if has_padding else None
        if attn_mask is not None and attn_mask.shape[0] != bs:
            attn_mask = attn_mask.expand(bs, -1)
        c = memory.shape[-1]
        n_clips = bs // Tl
        if frame_shard is None:
            ctr = slice(T // 2, None, T)
        elif frame_shard.owns_center():
            ctr = slice(frame_shard.local_index_of_center(), None, Tl)
        else:
            ctr = slice(0, None, Tl)  # stand-in rows; the owner's proposals are broadcast below
        now_frame_memory = memory[ctr]
        now_frame_mask_flatten = mask_flatten[ctr] if mask_flatten.shape[0] == bs else mask_flatten
        now_frame_valid_ratios = valid_ratios[ctr]  # equal for every frame of a clip
        if self.as_two_stage:
            mask_key = SourceKey((mask_flatten,), extra=(ctr.start, ctr.step, now_frame_memory.shape[0])) \
                if (has_padding and frame_shard is None) else None
            output_memory, output_proposals = self.gen_encoder_output_proposals(
                now_frame_memory, now_frame_mask_flatten if has_padding else None, geom, mask_key=mask_key)
            enc_outputs_class = mlp_rows(cls_branches[self.decoder.num_layers], output_memory)
            topk = self.two_stage_num_proposals
            logits = enc_outputs_class[..., 0]
            if (logits.is_cuda and logits.dtype == torch.float32 and logits.shape[1] <= 32768
                    and topk <= 1024 and not torch.is_grad_enabled()):
                from . import ops
                topk_proposals = ops.topk_rows(logits, topk)[1]     # one launch (torch.topk: ~22)
            else:
                topk_proposals = torch.topk(logits, topk, dim=1)[1]
            forced = kwargs.pop('force_topk_proposals', None)
            if forced is not None:  # parity harness: follow the reference's selection
                topk_proposals = forced
            self.last_topk_proposals = topk_proposals
            self.last_enc_cls = enc_outputs_class  # [B, S, 1] (parity harness)
            # The reference runs the keypoint / sigma branches on all S tokens and gathers the
            # top-k rows afterwards (OT:21372-21389); the branches are row-wise, so gathering
            # first is identical and 74x less work (300 of 22 323 rows).  The all-token
            # outputs only feed training losses.
            query_pos, query = torch.split(query_embed, c, dim=1)
            # device fast path: the gather / repeat / strided add / sigmoid / repeat sequence below as two
            # launches (ops.gather_rows_add: `tgt` and `tgt + query`; ops.proposal_refs_: the in-place proposal
            # offset and the T-fold reference points)
            fused = (self.fused_proposal_stage and output_memory.is_cuda
                     and output_memory.dtype == torch.float32 and frame_shard is None
                     and not torch.is_grad_enabled() and output_memory.is_contiguous() and c % 4 == 0
                     and output_proposals.is_contiguous())
            fused_query = None
            if fused:
                from . import ops
                topk_proposals = topk_proposals.contiguous()
                tgt, fused_query = ops.gather_rows_add(output_memory, topk_proposals,
                                                       self._dense_const('query_half', query_embed, query))
            else:
                tgt = torch.gather(output_memory, 1,
                                   topk_proposals.unsqueeze(-1).repeat(1, 1, self.embed_dims))
            topk_kpts_unact = mlp_rows(kpt_branches[self.decoder.num_layers], tgt)
            fused_refs = None
            if fused and topk_kpts_unact.stride(-1) == 1:
                fused_refs = ops.proposal_refs_(topk_kpts_unact, output_proposals, topk_proposals, T)
            else:
                top_props = torch.gather(output_proposals, 1,
                                         topk_proposals.unsqueeze(-1).repeat(1, 1, 2))
                topk_kpts_unact[..., 0::2] += top_props[..., 0:1]
                topk_kpts_unact[..., 1::2] += top_props[..., 1:2]
            enc_outputs_kpt_unact = topk_kpts_unact
            enc_outputs_sigma_unact = mlp_rows(sigma_branches[self.decoder.num_layers], tgt)
            if frame_shard is not None:
                from . import dist as pdist
                # (branch outputs off the 4-column grid are column slices of a padded matrix)
                tgt, topk_kpts_unact = tgt.contiguous(), topk_kpts_unact.contiguous()
                enc_outputs_kpt_unact = topk_kpts_unact
                enc_outputs_sigma_unact = enc_outputs_sigma_unact.contiguous()
                for t_ in (tgt, topk_kpts_unact, enc_outputs_sigma_unact):
                    pdist.broadcast_from(t_, frame_shard.center_owner, frame_shard.group)
            reference_points = fused_refs if fused_refs is not None else \
                topk_kpts_unact.sigmoid().repeat(1, T, 1)
            init_reference_out = reference_points
            query_pos = query_pos.unsqueeze(0).expand(n_clips, -1, -1)
            query = fused_query if fused_query is not None else \
                tgt + query.unsqueeze(0).expand(n_clips, -1, -1)
        else:
            query_pos, query = torch.split(query_embed, c, dim=1)
            query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
            query = query.unsqueeze(0).expand(bs, -1, -1)
            reference_points = self.reference_points(query_pos).sigmoid()
            init_reference_out = reference_points
            enc_outputs_class = enc_outputs_kpt_unact = enc_outputs_sigma_unact = None
        dec_kwargs = {}
        if branches is not None:
            dec_kwargs['frame_kpt_branches'] = branches
        if frame_shard is not None:
            dec_kwargs['frame_shard'] = frame_shard
        if not has_padding and isinstance(self.decoder, VideoPoseTransformerDecoderMulFrames):
            dec_kwargs['unit_valid_ratios'] = True
        cached = kwargs.pop('values_projected', None)   # streaming: per-frame caches + frame table
        if cached is not None:
            dec_kwargs['values_projected'] = cached
            dec_kwargs['value_frame_table'] = kwargs.pop('value_frame_table')
        elif self.hoist_value_proj and all(
                isinstance(l.attentions[-1], MulFramesMultiScaleDeformablePoseAttention)
                for l in self.decoder.layers):
            dec_kwargs['values_projected'] = project_values_hoisted(
                [l.attentions[-1] for l in self.decoder.layers], memory, attn_mask,
                masked_rows=mlvl_masks.masked_rows() if (attn_mask is not None
                                                         and hasattr(mlvl_masks, 'masked_rows')) else None)
        inter_states, inter_references = self.decoder(
            query=seq_first_view(query.contiguous()), key=None, value=seq_first_view(memory),
            query_pos=seq_first_view(query_pos), key_padding_mask=attn_mask,
            reference_points=reference_points, spatial_shapes=spatial_shapes,
            level_start_index=level_start_index, valid_ratios=now_frame_valid_ratios,
            **dec_kwargs)
        memory_out = seq_first_view(memory)
        if self.as_two_stage:
            return inter_states, init_reference_out, inter_references, enc_outputs_class, \
                enc_outputs_kpt_unact, enc_outputs_sigma_unact, None, memory_out
        return inter_states, init_reference_out, inter_references, None, None, None, None, None, None

    # -- a9: forward_refine (OT:21458-21536) ---------------------------------
    def forward_refine(self, mlvl_masks, memory, reference_points_pose, img_inds,
                       has_padding=True, frame_shard=None, **kwargs):
        """memory [S, B, T, C] (view of [B, T, S, C]); reference_points_pose [T*N, 2K]
        frame-major; img_inds [N] clip index of each pose.  With frame_shard, memory and masks
        hold only the local frames ([S, B, T_loc, C])."""
        T = self.num_frames
        Tl = T if frame_shard is None else frame_shard.n_local
        branches = self._frame_branch_lists(kwargs, 'kpt_branches')
        dev = memory.device
        geom = self.geometry([m.shape[-2:] for m in mlvl_masks], dev)
        spatial_shapes, level_start_index = geom.spatial_shapes, geom.level_start_index
        const = geom.unpadded.get(('refine', mlvl_masks[0].shape[0])) \
            if (not has_padding and memory.is_cuda) else None
        pkey = SourceKey(list(mlvl_masks)) if (has_padding and memory.is_cuda and not torch.is_grad_enabled()) else None
        phit = geom.unpadded.get(('refine_padded', mlvl_masks[0].shape[0])) if pkey is not None else None
        if const is not None:            # no padding: all-False masks, valid ratios exactly 1
            mask_flatten, valid_ratios = const
        elif phit is not None and phit[0] == pkey:     # (a padded batch's masks are cached per batch shape)
            mask_flatten, valid_ratios = phit[1], phit[2]
        else:
            mask_flatten = torch.cat([m.flatten(1) for m in mlvl_masks], 1)
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in mlvl_masks], 1)
            if not has_padding and memory.is_cuda and not torch.is_grad_enabled():
                geom.unpadded[('refine', mlvl_masks[0].shape[0])] = (mask_flatten, valid_ratios)
            elif pkey is not None:
                geom.unpadded[('refine_padded', mlvl_masks[0].shape[0])] = (pkey, mask_flatten, valid_ratios)
        B = memory.shape[1]
        if valid_ratios.shape[0] != B * Tl:
            valid_ratios = valid_ratios.expand(B * Tl, -1, -1)
            mask_flatten = mask_flatten.expand(B * Tl, -1)
        rq = self.refine_query_embedding.weight
        query_pos, query = torch.split(rq, rq.size(1) // 2, dim=1)
        pos_num = reference_points_pose.size(0) // T
        query_pos = query_pos.unsqueeze(0).expand(pos_num, -1, -1)
        query = query.unsqueeze(0).expand(pos_num, -1, -1)
        if query.is_cuda and not torch.is_grad_enabled():
            # (the decoder reads its first query from a dense tensor: the same rows every step)
            query = self._dense_const('refine_query', rq, query, rows=pos_num)
        reference_points = reference_points_pose.reshape(-1, reference_points_pose.size(1) // 2, 2)
        mask_bt = mask_flatten.reshape(-1, Tl, mask_flatten.size(-1))         # [B, T_loc, S]
        if has_padding:
            vkey = SourceKey((valid_ratios, img_inds), extra=(Tl, T, frame_shard is not None)) if pkey is not None else None
            vhit = self.__dict__.get('_pave_vr_rows')
            if vkey is not None and vhit is not None and vhit[0] == vkey:
                vr_rows = vhit[1]          # (the gather by the head's cached clip-of-pose index, once per mask set)
            else:
                vr = valid_ratios.reshape(-1, Tl, valid_ratios.size(-2), valid_ratios.size(-1))[img_inds]
                if frame_shard is not None:
                    # every frame of a clip has the same valid ratios: rebuild the [N, T, L, 2] table
                    vr = vr[:, :1].expand(-1, T, -1, -1)
                # FRAME-major rows, as the reference points they scale ([T * N, K, 2]: `[pre..., now..., next...]`,
                # HEAD:610).  The reference flattens this table pose-major (`flatten(0, 1)`, OT:21511) -- the same
                # thing only while every pose belongs to ONE clip (it asserts batch size 1); in a batch of clips
                # with different valid sizes that order would hand a pose another clip's ratios (found in round 6:
                # poses of a padded two-clip batch were sampled at 750/800 of their y coordinate)
                vr_rows = vr.transpose(0, 1).flatten(0, 1)
                if vkey is not None:
                    self.__dict__['_pave_vr_rows'] = (vkey, vr_rows)
        else:   # every valid ratio is exactly 1: a broadcast row, no gather / copy per step
            vr_rows = valid_ratios[:1].expand(img_inds.shape[0] * T, -1, -1)
        dec_kwargs = {}
        if branches is not None:
            dec_kwargs['frame_reg_branches'] = branches
        if frame_shard is not None:
            dec_kwargs['frame_shard'] = frame_shard
        if not has_padding and isinstance(self.refine_decoder, DeformableDetrTransformerDecoderMulFrames):
            dec_kwargs['unit_valid_ratios'] = True
        mem_bt = memory.permute(1, 2, 0, 3)                                   # [B, T, S, C]
        attn_mask = mask_bt if has_padding else None
        cached = kwargs.pop('values_projected', None)   # streaming: per-frame caches + frame table
        if cached is not None:
            dec_kwargs['values_projected'] = cached
            dec_kwargs['value_frame_table'] = kwargs.pop('value_frame_table')
        elif self.hoist_value_proj and all(
                isinstance(l.attentions[-1], MulFramesMultiScaleDeformableAttention)
                for l in self.refine_decoder.layers):
            dec_kwargs['values_projected'] = project_values_hoisted(
                [l.attentions[-1] for l in self.refine_decoder.layers], mem_bt, attn_mask,
                masked_rows=mlvl_masks.masked_rows() if (attn_mask is not None
                                                         and hasattr(mlvl_masks, 'masked_rows')) else None)
        inter_states, inter_references = self.refine_decoder(
            query=seq_first_view(query.contiguous()), key=None, value=memory,
            query_pos=seq_first_view(query_pos), key_padding_mask=attn_mask,
            reference_points=reference_points, spatial_shapes=spatial_shapes,
            level_start_index=level_start_index, valid_ratios=vr_rows,
            memory_clip_index=img_inds, **dec_kwargs)
        return inter_states, reference_points, inter_references
