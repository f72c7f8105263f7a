"""Single-frame PETR / vedpose family on the same kernels:

* ``opera.PETRTransformer``   OT:4234-4693
* ``opera.PETRHead``          opera/models/dense_heads/petr_head.py:26-300, 956-1037
* ``opera.VedPoseHeadV2``     opera/models/dense_heads/vedpose_head_v2.py (PETRHead + RLE sigma
                              branches + ``get_p`` confidence re-scaling, no NMS: :1066-1180)
* ``opera.PETR``              opera/models/detectors/petr.py:84-116

Same ctor kwargs / state-dict keys.  Native differences: B >= 1 images per call; the image
memory is not replicated per pose in the refine decoder (OT:4673 ``memory[:, img_inds]``);
the keypoint branch of the two-stage proposals runs on the top-k rows only.
"""
import copy

import numpy as np
import torch
import torch.nn as nn

from .bricks import (Linear, batch_first, bias_init_with_prob, constant_init, inverse_sigmoid,
                     mlp_rows, seq_first_view, xavier_init)
from .detectors import VideoPoseV1
from .heads import (RealNVP, VideoPoseHeadMulFrames, _clones, _kpt_branch, _refine_kpt_branch,
                    _sigma_branch, _TrainingOnlyLoss)
from .registry import (DETECTORS, HEADS, LOSSES, TRANSFORMER, build_positional_encoding,
                       build_transformer, build_transformer_layer_sequence)
from .transformer import VideoPoseTransformerMulFrames
from .bricks import BaseModule, build_activation_layer
from .deform_attn import MultiScaleDeformableAttention, MultiScaleDeformablePoseAttention


@TRANSFORMER.register_module()
class PETRTransformer(VideoPoseTransformerMulFrames):
    """OT:4234-4693.  Reuses the geometry / proposal helpers of the video transformer."""

    def __init__(self, hm_encoder=None, refine_decoder=None, as_two_stage=True,
                 num_feature_levels=4, two_stage_num_proposals=300, num_keypoints=17, **kwargs):
        self._hm_encoder_cfg = hm_encoder
        super().__init__(hm_encoder=None, refine_decoder=refine_decoder,
                         as_two_stage=as_two_stage, num_feature_levels=num_feature_levels,
                         two_stage_num_proposals=two_stage_num_proposals,
                         num_keypoints=num_keypoints, num_frames=1, **kwargs)
        if hm_encoder is None:
            hm_encoder = dict(
                type='mmcv.DetrTransformerEncoder', num_layers=1,
                transformerlayers=dict(
                    type='mmcv.BaseTransformerLayer',
                    attn_cfgs=dict(type='mmcv.MultiScaleDeformableAttention', embed_dims=256,
                                   num_levels=1),
                    feedforward_channels=1024, ffn_dropout=0.1,
                    operation_order=('self_attn', 'norm', 'ffn', 'norm')))
        # heat-map encoder: training-only branch (OT:4551-4573), kept for the state dict
        self.hm_encoder = build_transformer_layer_sequence(hm_encoder)

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MultiScaleDeformableAttention):
                m.init_weights()
        for m in self.modules():
            if isinstance(m, MultiScaleDeformablePoseAttention):
                m.init_weights()
        if not self.as_two_stage:
            xavier_init(self.reference_points, distribution='uniform', bias=0.)
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.refine_query_embedding.weight)
        self._is_init = True

    def forward(self, mlvl_feats, mlvl_masks, query_embed, mlvl_pos_embeds, kpt_branches=None,
                cls_branches=None, has_padding=True, **kwargs):
        assert self.as_two_stage or query_embed is not None
        # flatten + encoder: the video transformer's frame encoder (every image is a one-frame clip) -- the same
        # launches, and its per-shape caches (flattened masks / positional table, valid ratios, reference grid), so a
        # warm step rebuilds none of them (round 6: the merged projection's epilogue table used to miss its cache
        # every step here, six vendor GEMMs per forward)
        memory, mask_flatten, valid_ratios, geom = self.encode_frames(mlvl_feats, mlvl_masks, mlvl_pos_embeds,
                                                                      has_padding)
        bs = memory.shape[0]
        if mask_flatten.shape[0] != bs:
            mask_flatten = mask_flatten.expand(bs, -1)
        if valid_ratios.shape[0] != bs:
            valid_ratios = valid_ratios.expand(bs, -1, -1)
        spatial_shapes, level_start_index = geom.spatial_shapes, geom.level_start_index
        attn_mask = mask_flatten if has_padding else None
        c = memory.shape[-1]
        hm_proto = None  # training only (OT:4551)
        if self.as_two_stage:
            output_memory, output_proposals = self.gen_encoder_output_proposals(
                memory, attn_mask, geom)
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
            if forced is not None:
                topk_proposals = forced
            self.last_topk_proposals = topk_proposals
            self.last_enc_cls = enc_outputs_class  # [B, S, 1] (parity harness)
            rows = torch.gather(output_memory, 1,
                                topk_proposals.unsqueeze(-1).repeat(1, 1, self.embed_dims))
            props = torch.gather(output_proposals, 1, topk_proposals.unsqueeze(-1).repeat(1, 1, 2))
            topk_kpts_unact = mlp_rows(kpt_branches[self.decoder.num_layers], rows)  # top-k rows only
            topk_kpts_unact[..., 0::2] += props[..., 0:1]
            topk_kpts_unact[..., 1::2] += props[..., 1:2]
            enc_outputs_kpt_unact = topk_kpts_unact
            reference_points = topk_kpts_unact.sigmoid()
            init_reference_out = reference_points
            query_pos, query = torch.split(query_embed, c, dim=1)
            query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
            query = query.unsqueeze(0).expand(bs, -1, -1)  # NOT added to the memory rows (OT:4596)
        else:
            query_pos, query = torch.split(query_embed, c, dim=1)
            query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
            query = query.unsqueeze(0).expand(bs, -1, -1)
            reference_points = self.reference_points(query_pos).sigmoid()
            init_reference_out = reference_points
            enc_outputs_class = enc_outputs_kpt_unact = None
        inter_states, inter_references = self.decoder(
            query=seq_first_view(query.contiguous()), key=None, value=seq_first_view(memory),
            query_pos=seq_first_view(query_pos), key_padding_mask=attn_mask,
            reference_points=reference_points, spatial_shapes=spatial_shapes,
            level_start_index=level_start_index, valid_ratios=valid_ratios,
            kpt_branches=kpt_branches)
        return inter_states, init_reference_out, inter_references, enc_outputs_class, \
            enc_outputs_kpt_unact, hm_proto, seq_first_view(memory)

    def forward_refine(self, mlvl_masks, memory, reference_points_pose, img_inds,
                       kpt_branches=None, has_padding=True, **kwargs):
        """memory [S, B, C]; reference_points_pose [N, 2K]; img_inds [N]."""
        dev = memory.device
        geom = self.geometry([m.shape[-2:] for m in mlvl_masks], dev)
        mask_flatten = torch.cat([m.flatten(1) for m in mlvl_masks], 1)
        valid_ratios = torch.stack([self.get_valid_ratio(m) for m in mlvl_masks], 1)
        B = memory.shape[1]
        if valid_ratios.shape[0] != B:
            valid_ratios = valid_ratios.expand(B, -1, -1)
            mask_flatten = mask_flatten.expand(B, -1)
        rq = self.refine_query_embedding.weight
        query_pos, query = torch.split(rq, rq.size(1) // 2, dim=1)
        pos_num = reference_points_pose.size(0)
        query_pos = query_pos.unsqueeze(0).expand(pos_num, -1, -1)
        query = query.unsqueeze(0).expand(pos_num, -1, -1)
        reference_points = reference_points_pose.reshape(pos_num,
                                                         reference_points_pose.size(1) // 2, 2)
        inter_states, inter_references = self.refine_decoder(
            query=seq_first_view(query.contiguous()), key=None, value=memory,
            query_pos=seq_first_view(query_pos),
            key_padding_mask=mask_flatten if has_padding else None,
            reference_points=reference_points, spatial_shapes=geom.spatial_shapes,
            level_start_index=geom.level_start_index, valid_ratios=valid_ratios[img_inds],
            reg_branches=kpt_branches, memory_clip_index=img_inds)
        return inter_states, reference_points, inter_references


@HEADS.register_module()
class PETRHead(BaseModule):
    """petr_head.py:26-300 (forward), :956-1037 (_get_bboxes_single)."""

    with_sigma = False

    def __init__(self, num_classes, in_channels, num_query=100, num_kpt_fcs=2, num_keypoints=17,
                 transformer=None, sync_cls_avg_factor=True,
                 positional_encoding=dict(type='SinePositionalEncoding', num_feats=128,
                                          normalize=True),
                 loss_cls=dict(type='mmdet.FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                               loss_weight=2.0),
                 loss_kpt=None, loss_oks=None, loss_hm=None, as_two_stage=True,
                 with_kpt_refine=True, train_cfg=None, loss_kpt_rpn=None, loss_kpt_refine=None,
                 loss_oks_refine=None, test_cfg=dict(max_per_img=100), init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.num_query = num_query
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.num_kpt_fcs = num_kpt_fcs
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg if test_cfg is not None else dict(max_per_img=100)
        self.as_two_stage = as_two_stage
        self.with_kpt_refine = with_kpt_refine
        self.num_keypoints = num_keypoints
        self.num_frames = 1
        if not as_two_stage:
            raise RuntimeError('only "as_two_stage=True" is supported.')
        transformer = copy.deepcopy(dict(transformer))
        transformer['as_two_stage'] = as_two_stage
        self.loss_cls = LOSSES.build(dict(loss_cls))
        self.cls_out_channels = num_classes if self.loss_cls.use_sigmoid else num_classes + 1
        self.act_cfg = transformer.get('act_cfg', dict(type='ReLU', inplace=True))
        self.activate = build_activation_layer(self.act_cfg)
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.transformer = build_transformer(transformer)
        self.embed_dims = self.transformer.embed_dims
        assert positional_encoding['num_feats'] * 2 == self.embed_dims
        self._init_layers()
        self._consts = {}

    def _init_layers(self):
        E, K, F_ = self.embed_dims, self.num_keypoints, self.num_kpt_fcs
        num_pred = self.transformer.decoder.num_layers + 1
        self.cls_branches = _clones(Linear(E, self.cls_out_channels), num_pred)
        self.kpt_branches = _clones(_kpt_branch(E, F_, K), num_pred)
        if self.with_sigma:
            self.dec_fc_sigma_branches = _clones(_sigma_branch(E, F_, 2 * K), num_pred - 1)
        self.query_embedding = nn.Embedding(self.num_query, E * 2)
        n_ref = self.transformer.refine_decoder.num_layers
        self.refine_kpt_branches = _clones(_refine_kpt_branch(E, F_), n_ref)
        if self.with_sigma:
            self.refine_fc_sigma_branches = _clones(_sigma_branch(E, F_, 2), n_ref)
        self.fc_hm = Linear(E, K)
        if self.with_sigma:
            masks = torch.from_numpy(np.array([[0, 1], [1, 0]] * 3).astype(np.float32))
            self.dec_flow = RealNVP(masks.clone())
            self.flow = RealNVP(masks.clone())

    def init_weights(self):
        self.transformer.init_weights()
        if self.loss_cls.use_sigmoid:
            for m in self.cls_branches:
                nn.init.constant_(m.bias, bias_init_with_prob(0.01))
        for m in self.kpt_branches:
            constant_init(m[-1], 0, bias=0)
        for m in self.refine_kpt_branches:
            constant_init(m[-1], 0, bias=0)
        nn.init.normal_(self.fc_hm.weight, std=0.01)
        nn.init.constant_(self.fc_hm.bias, bias_init_with_prob(0.1))
        self._is_init = True

    make_masks = VideoPoseHeadMulFrames.make_masks
    make_masks_from_shapes = VideoPoseHeadMulFrames.make_masks_from_shapes
    _meta_scales = VideoPoseHeadMulFrames._meta_scales
    get_p = staticmethod(VideoPoseHeadMulFrames.get_p)
    results_to_list = staticmethod(VideoPoseHeadMulFrames.results_to_list)

    def forward(self, mlvl_feats, img_metas, **tr_kwargs):
        mlvl_masks, mlvl_pos, has_padding = self.make_masks(mlvl_feats, img_metas, 1)
        hs, init_reference, inter_references, enc_outputs_class, enc_outputs_kpt, hm_proto, \
            memory = self.transformer(mlvl_feats, mlvl_masks, self.query_embedding.weight,
                                      mlvl_pos, kpt_branches=self.kpt_branches,
                                      cls_branches=self.cls_branches, has_padding=has_padding,
                                      **tr_kwargs)
        hs = hs.permute(0, 2, 1, 3)
        outputs_classes, outputs_kpts = [], []
        for lvl in range(hs.shape[0]):
            reference = init_reference if lvl == 0 else inter_references[lvl - 1]
            reference = inverse_sigmoid(reference)
            outputs_classes.append(mlp_rows(self.cls_branches[lvl], hs[lvl]))
            outputs_kpts.append((mlp_rows(self.kpt_branches[lvl], hs[lvl]) + reference).sigmoid())
        return dict(all_cls_scores=torch.stack(outputs_classes),
                    all_kpt_preds=torch.stack(outputs_kpts), enc_cls_scores=enc_outputs_class,
                    enc_kpt_preds=enc_outputs_kpt.sigmoid(), memory=memory,
                    mlvl_masks=mlvl_masks, has_padding=has_padding, hs=hs,
                    init_reference=init_reference, inter_references=inter_references)

    def forward_refine(self, memory, mlvl_masks, kpt_preds, img_inds, has_padding=True):
        hs, init_reference, inter_references = self.transformer.forward_refine(
            mlvl_masks, memory, kpt_preds.detach(), img_inds,
            kpt_branches=self.refine_kpt_branches, has_padding=has_padding)
        hs = hs.permute(0, 2, 1, 3)
        outs_kpt, outs_sigma = [], []
        for lvl in range(hs.shape[0]):
            reference = init_reference if lvl == 0 else inter_references[lvl - 1]
            reference = inverse_sigmoid(reference)
            outs_kpt.append((mlp_rows(self.refine_kpt_branches[lvl], hs[lvl]) + reference).sigmoid())
            if self.with_sigma:
                outs_sigma.append(mlp_rows(self.refine_fc_sigma_branches[lvl], hs[lvl]).sigmoid())
        return torch.stack(outs_kpt), (torch.stack(outs_sigma) if outs_sigma else None), hs

    def get_bboxes(self, outs, img_metas, rescale=False, force_score_topk=None, taps=None):
        cls_scores = outs['all_cls_scores'][-1]
        kpt_preds = outs['all_kpt_preds'][-1]
        B = cls_scores.shape[0]
        K = self.num_keypoints
        N = self.test_cfg.get('max_per_img', self.num_query)
        cls_score = cls_scores.sigmoid().view(B, -1)
        if (cls_score.is_cuda and cls_score.dtype == torch.float32 and not torch.is_grad_enabled()
                and cls_score.shape[1] <= 32768 and N <= 1024):
            from . import ops
            scores, indexs = ops.topk_rows(cls_score, N)          # one launch
        else:
            scores, indexs = cls_score.topk(N, dim=1)
        if force_score_topk is not None:
            indexs = force_score_topk
            scores = torch.gather(cls_score, 1, indexs)
        det_labels = indexs % self.num_classes
        bbox_index = indexs // self.num_classes
        sel = torch.gather(kpt_preds, 1, bbox_index.unsqueeze(-1).expand(-1, -1, 2 * K))
        img_inds = torch.arange(B, device=cls_scores.device).repeat_interleave(N)
        r_kpts, r_sigmas, r_hs = self.forward_refine(
            outs['memory'], outs['mlvl_masks'], sel.reshape(B * N, 2 * K), img_inds,
            has_padding=outs['has_padding'])
        det_kpts = r_kpts[-1].view(B, N, K, 2)
        if taps is not None:
            taps.update(score_topk=indexs, refine_hs=r_hs, refine_kpts=det_kpts.clone())
        dev = det_kpts.device
        wh, sf = self._meta_scales(img_metas, dev)
        det_kpts = torch.minimum((det_kpts * wh).clamp(min=0), wh)
        if rescale:
            det_kpts = det_kpts / sf
        x1 = det_kpts[..., 0].min(dim=2, keepdim=True)[0]
        y1 = det_kpts[..., 1].min(dim=2, keepdim=True)[0]
        x2 = det_kpts[..., 0].max(dim=2, keepdim=True)[0]
        y2 = det_kpts[..., 1].max(dim=2, keepdim=True)[0]
        det_bboxes = torch.cat([x1, y1, x2, y2, scores.unsqueeze(-1)], dim=2)
        if self.with_sigma:  # vedpose_head_v2.py:1158-1178
            p = self.get_p(r_sigmas[-1].view(B * N, K, 2)).view(B, N, K, 1)
            p5 = p**5
            det_kpts = (det_kpts * p5) / (p5 + 1e-10)
            kscore = scores[:, :, None, None] * p
        else:              # petr_head.py:1033-1035
            kscore = det_kpts.new_ones(det_kpts[..., :1].shape)
        det_kpts = torch.cat((det_kpts, kscore), dim=3)
        keep = torch.ones((B, N), dtype=torch.int32, device=dev)  # no NMS in this family
        order = torch.arange(N, device=dev, dtype=torch.int32).expand(B, N)
        return dict(bboxes=det_bboxes, labels=det_labels, kpts=det_kpts, keep=keep, order=order,
                    scores=scores)

    def simple_test_bboxes(self, feats, img_metas, rescale=False):
        outs = self.forward(feats, img_metas)
        return self.results_to_list(self.get_bboxes(outs, img_metas, rescale=rescale))

    simple_test = simple_test_bboxes


@HEADS.register_module()
class VedPoseHeadV2(PETRHead):
    """vedpose_head_v2.py: PETRHead + RLE sigma branches; inference re-scales coordinates and
    scores with get_p(sigma) (:1158-1178)."""
    with_sigma = True


@DETECTORS.register_module()
class PETR(VideoPoseV1):
    """opera/models/detectors/petr.py:84-116: img [B, 3, H, W]."""
