"""``opera.VideoPoseHeadMulFrames`` (a11, a12): branches, query embedding, decode, device OKS-NMS.

Restated from opera/models/dense_heads/videopose_head_mul_frames.py (HEAD): ctor :73-160,
_init_layers :162-353, forward :403-567, forward_refine (inference branch) :569-674,
_get_bboxes_single :1371-1505, get_p :1531-1535, RealNVP :1538-1601, Linear_with_norm
:1605-1622.  Same ctor kwargs and state-dict keys (incl. the training-only fc_hm / RealNVP
flows, kept so reference checkpoints load with strict=True).

Native differences: any odd num_frames; B >= 1 clips per call (the reference asserts B = 1 and
replicates the clip memory per pose); results stay on the device as fixed-shape tensors with a
keep mask from the HIP OKS-NMS kernel -- no host sync inside the path.
"""
import copy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .bricks import (BaseModule, Linear, bias_init_with_prob, build_activation_layer,
                     constant_init, mlp_rows)
from .deform_attn import frame_prefixes
from .transformer import _frame_branches, _ref_update
from .registry import HEADS, LOSSES, MMDET_MODELS, build_positional_encoding, build_transformer

OKS_SIGMAS_POSETRACK15 = [.26, .79, .79, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87,
                          .89, .89]


class _TrainingOnlyLoss(nn.Module):
    """Placeholder for the reference's training losses (out of scope):
    keeps the config surface (``use_sigmoid`` etc.) and fails loudly if called."""

    def __init__(self, use_sigmoid=False, **kwargs):
        super().__init__()
        self.use_sigmoid = use_sigmoid
        self.cfg = kwargs

    def forward(self, *args, **kwargs):
        raise NotImplementedError('pavenet_amd is a forward (inference) path: training losses '
                                  'of the reference are not built')


for _scope, _names in ((MMDET_MODELS, ('FocalLoss', 'L1Loss', 'CrossEntropyLoss', 'GIoULoss',
                                       'SmoothL1Loss', 'MSELoss')),
                       (LOSSES, ('OKSLoss', 'RLELoss', 'CenterFocalLoss'))):
    for _n in _names:
        _scope.register_module(name=_n, module=type(_n, (_TrainingOnlyLoss,), {}))


class Linear_with_norm(nn.Module):  # noqa: N801  (reference class name, HEAD:1605-1622)

    def __init__(self, in_channel, out_channel, bias=True, norm=True):
        super().__init__()
        self.bias = bias
        self.norm = norm
        self.linear = nn.Linear(in_channel, out_channel, bias)
        self.constructor_init()

    def constructor_init(self):
        """The reference's constructor-time initialisation (HEAD:1611: xavier_uniform, gain 0.01), which
        `init_weights` never redoes.  weights.init_random_weights(reference_sigma_init=True) re-applies it after
        re-drawing the defaults (the bench / parity recipe leaves it off: see weights.py)."""
        nn.init.xavier_uniform_(self.linear.weight, gain=0.01)

    def forward(self, x):
        y = x.matmul(self.linear.weight.t())
        if self.norm:
            y = y / torch.norm(x, dim=-1, keepdim=True)
        if self.bias:
            y = y + self.linear.bias
        return y


def _nets():
    return nn.Sequential(nn.Linear(2, 64), nn.LeakyReLU(), nn.Linear(64, 64), nn.LeakyReLU(),
                         nn.Linear(64, 2), nn.Tanh())


def _nett():
    return nn.Sequential(nn.Linear(2, 64), nn.LeakyReLU(), nn.Linear(64, 64), nn.LeakyReLU(),
                         nn.Linear(64, 2))


class RealNVP(nn.Module):
    """Parameter container of the RLE flow (HEAD:1538-1601); used by training losses only."""

    def __init__(self, mask):
        super().__init__()
        self.register_buffer('mask', mask)
        self.t = nn.ModuleList([_nett() for _ in range(len(mask))])
        self.s = nn.ModuleList([_nets() for _ in range(len(mask))])


def _kpt_branch(embed_dims, num_kpt_fcs, num_keypoints):
    layers = [Linear(embed_dims, 512), nn.ReLU()]
    for _ in range(num_kpt_fcs):
        layers += [Linear(512, 512), nn.ReLU()]
    layers.append(Linear(512, 2 * num_keypoints))
    return nn.Sequential(*layers)


def _sigma_branch(embed_dims, num_kpt_fcs, out):
    layers = [Linear(embed_dims, embed_dims) for _ in range(num_kpt_fcs)]
    layers.append(Linear_with_norm(embed_dims, out, norm=False))
    return nn.Sequential(*layers)


def _refine_kpt_branch(embed_dims, num_kpt_fcs):
    layers = []
    for _ in range(num_kpt_fcs):
        layers += [Linear(embed_dims, embed_dims), nn.ReLU()]
    layers.append(Linear(embed_dims, 2))
    return nn.Sequential(*layers)


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class MaskList(list):
    """The per-level padding masks of a batch, with what the host knows about them: `frame_groups` = runs
    (first frame, count) of consecutive frames sharing one valid size (padded batches only) -- lets the encoder
    take its merged-projection path per run without reading the masks back from the device."""
    frame_groups = None
    _rows = None

    def masked_rows(self):
        """int32 indices of the masked tokens in the flattened [n_frames * S] token order (levels concatenated
        per frame, as the transformer flattens them): built once per cached mask set -- one device read-back
        when a padded batch shape is first seen, none afterwards (and none inside a hipGraph capture)."""
        if self._rows is None:
            with torch.no_grad():
                flat = torch.cat([m.flatten(1) for m in self], 1)
                self._rows = flat.reshape(-1).nonzero().flatten().to(torch.int32)
        return self._rows


@HEADS.register_module()
class VideoPoseHeadMulFrames(BaseModule):

    def __init__(self, num_classes, in_channels, num_frames=3, num_query=100, num_kpt_fcs=2,
                 num_keypoints=17, transformer=None, sync_cls_avg_factor=True,
                 positional_encoding=dict(type='SinePositionalEncoding', num_feats=128,
                                          normalize=True),
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                               loss_weight=2.0),
                 loss_kpt=None, loss_oks=None, loss_hm=None, as_two_stage=True,
                 with_kpt_refine=True, train_cfg=None, loss_kpt_rpn=None, loss_kpt_refine=None,
                 loss_oks_refine=None, test_cfg=dict(max_per_img=100), init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.bg_cls_weight = 0
        self.sync_cls_avg_factor = sync_cls_avg_factor
        self.num_query = num_query
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.num_kpt_fcs = num_kpt_fcs
        self.train_cfg = train_cfg  # assigner / sampler are training-only: not built
        self.test_cfg = test_cfg if test_cfg is not None else dict(max_per_img=100)
        self.fp16_enabled = False
        self.as_two_stage = as_two_stage
        self.with_kpt_refine = with_kpt_refine
        self.num_keypoints = num_keypoints
        self.num_frames = num_frames
        self.frame_prefixes = frame_prefixes(num_frames)
        # False (default) = the reference's contract: `forward` returns full [levels, ...] stacks of
        # class / key-point / sigma predictions (HEAD:486-545).  The inference entry points
        # (`simple_test_bboxes`, the detector's `forward_device`) only read [-1] (HEAD:1304-1330) and
        # ask `forward(..., last_level_only=True)` for one-level stacks instead.
        self.eval_last_level_only = False
        if not self.as_two_stage:
            raise RuntimeError('only "as_two_stage=True" is supported.')
        transformer = copy.deepcopy(dict(transformer))
        transformer['as_two_stage'] = self.as_two_stage
        loss_cls = dict(loss_cls)
        self.loss_cls = LOSSES.build(loss_cls) if 'type' in loss_cls else _TrainingOnlyLoss(**loss_cls)
        self.cls_out_channels = num_classes if self.loss_cls.use_sigmoid else num_classes + 1
        self.act_cfg = transformer.get('act_cfg', dict(type='ReLU', inplace=True))
        self.activate = build_activation_layer(self.act_cfg)
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.transformer = build_transformer(transformer)
        self.embed_dims = self.transformer.embed_dims
        num_feats = positional_encoding['num_feats']
        assert num_feats * 2 == self.embed_dims
        self._init_layers()
        self.oks_thresh = 0.45  # HEAD:1399
        self._consts = {}

    # HEAD:162-353
    def _init_layers(self):
        E, K, F_ = self.embed_dims, self.num_keypoints, self.num_kpt_fcs
        fc_cls = Linear(E, self.cls_out_channels)
        num_pred = self.transformer.decoder.num_layers + 1
        c = self.num_frames // 2
        # auxiliary-frame branches, num_pred-1 each (the centre frame has num_pred)
        for t, fp in enumerate(self.frame_prefixes):
            if t != c:
                setattr(self, fp + 'kpt_branches',
                        nn.ModuleList([_kpt_branch(E, F_, K) for _ in range(num_pred - 1)]))
        self.cls_branches = _clones(fc_cls, num_pred)
        self.kpt_branches = _clones(_kpt_branch(E, F_, K), num_pred)
        self.dec_fc_sigma_branches = _clones(_sigma_branch(E, F_, 2 * K), num_pred)
        self.query_embedding = nn.Embedding(self.num_query, E * 2)
        n_ref = self.transformer.refine_decoder.num_layers
        for t, fp in enumerate(self.frame_prefixes):
            if t != c:
                setattr(self, fp + 'refine_kpt_branches',
                        nn.ModuleList([_refine_kpt_branch(E, F_) for _ in range(n_ref)]))
        self.refine_kpt_branches = _clones(_refine_kpt_branch(E, F_), n_ref)
        self.refine_fc_sigma_branches = _clones(_sigma_branch(E, F_, 2), n_ref)
        self.fc_hm = Linear(E, K)
        masks = torch.from_numpy(np.array([[0, 1], [1, 0]] * 3).astype(np.float32))
        self.enc_flow = RealNVP(masks.clone())
        self.dec_flow = RealNVP(masks.clone())
        self.flow = RealNVP(masks.clone())

    def init_weights(self):
        self.transformer.init_weights()
        if self.loss_cls.use_sigmoid:
            bias_init = bias_init_with_prob(0.01)
            for m in self.cls_branches:
                nn.init.constant_(m.bias, bias_init)
        for m in self.kpt_branches:
            constant_init(m[-1], 0, bias=0)
        pre = getattr(self, 'pre_refine_kpt_branches', None)
        if pre is not None:
            for m in pre:
                constant_init(m[-1], 0, bias=0)
        nn.init.normal_(self.fc_hm.weight, std=0.01)
        nn.init.constant_(self.fc_hm.bias, bias_init_with_prob(0.1))
        self._is_init = True

    def _branches(self, suffix):
        return [getattr(self, fp + suffix) for fp in self.frame_prefixes]

    # HEAD:429-445 ----------------------------------------------------------
    def make_masks(self, mlvl_feats, img_metas, frames_per_clip=None):
        """Padding masks + sine encodings per level.  When no clip is padded they are built
        once for a single frame (they are identical for every frame) and broadcast."""
        return self.make_masks_from_shapes(mlvl_feats[0].size(0),
                                           [tuple(f.shape[-2:]) for f in mlvl_feats],
                                           mlvl_feats[0].device, img_metas, frames_per_clip)

    def make_masks_from_shapes(self, n, level_hw, dev, img_metas, frames_per_clip=None):
        T = frames_per_clip or self.num_frames
        H, W = img_metas[0]['batch_input_shape']
        shapes = [tuple(int(v) for v in img_metas[i // T]['img_shape'][:2]) for i in range(n)]
        has_padding = any(s != (H, W) for s in shapes)
        key = (H, W, tuple(shapes) if has_padding else None,
               tuple(tuple(int(v) for v in hw) for hw in level_hw), str(dev))
        if key not in self._consts:
            nm = n if has_padding else 1
            img_masks = torch.ones((nm, H, W), device=dev)
            for i in range(nm):
                h, w = shapes[i]
                img_masks[i, :h, :w] = 0
            masks, pos = [], []
            for hw in level_hw:
                m = F.interpolate(img_masks[None], size=tuple(hw)).to(torch.bool).squeeze(0)
                masks.append(m)
                pos.append(self.positional_encoding(m))
            if len(self._consts) > 8:
                self._consts.clear()
            masks = MaskList(masks)
            if has_padding:
                # runs of consecutive frames with one valid size (the T frames of a clip always; neighbouring
                # clips of equal size merge): one positional table / padding pattern per run
                runs, f0 = [], 0
                for i in range(1, n + 1):
                    if i == n or shapes[i] != shapes[f0]:
                        runs.append((f0, i - f0))
                        f0 = i
                masks.frame_groups = tuple(runs)
            self._consts[key] = (masks, pos)
        masks, pos = self._consts[key]
        return masks, pos, has_padding

    # HEAD:403-567 ----------------------------------------------------------
    def forward(self, mlvl_feats, img_metas, last_level_only=None, **tr_kwargs):
        """HEAD:403-567.  `last_level_only` (eval mode only; default `self.eval_last_level_only`):
        run the class / key-point / sigma branches on the last decoder level only -- the returned
        `all_*` stacks then have leading dim 1 instead of num_dec_layers (`[-1]` is unchanged)."""
        T, Q = self.num_frames, self.num_query
        c = T // 2
        shard = tr_kwargs.get('frame_shard')
        refine_value_cache = tr_kwargs.pop('refine_value_cache', None)   # streaming (see forward_refine)
        pre = tr_kwargs.pop('precomputed', None)
        if pre is not None:  # streaming: encoder memory comes from the per-frame cache
            mlvl_masks, mlvl_pos, has_padding, tr_kwargs['encoded'] = pre
        else:
            mlvl_masks, mlvl_pos, has_padding = self.make_masks(
                mlvl_feats, img_metas, shard.n_local if shard is not None else None)
        hs, init_reference, inter_references, enc_outputs_class, enc_outputs_kpt, \
            enc_outputs_sigma, hm_proto, memory = self.transformer(
                mlvl_feats, mlvl_masks, self.query_embedding.weight, mlvl_pos,
                frame_kpt_branches=self._branches('kpt_branches'),
                cls_branches=self.cls_branches, sigma_branches=self.dec_fc_sigma_branches,
                has_padding=has_padding, **tr_kwargs)
        hs = hs.permute(0, 2, 1, 3)
        outputs_classes, outputs_kpts, output_sigmas = [], [], []
        aux_poses = all_frame_poses = None
        n_lvl = hs.shape[0]
        # The class / key-point / sigma branches of the earlier decoder levels only feed training
        # losses (HEAD:1304-1330 reads [-1]); in eval mode they are skipped unless asked for.
        only_last = self.eval_last_level_only if last_level_only is None else bool(last_level_only)
        levels = [n_lvl - 1] if (only_last and not self.training) else range(n_lvl)
        for lvl in levels:
            reference = init_reference if lvl == 0 else inter_references[lvl - 1]
            if lvl == n_lvl - 1:
                # all T frames' key-point branches on hs[lvl] at once (one GEMM + batched GEMMs,
                # transformer._frame_branches), then ONE reference update over [B, T*Q, 2K]
                brs = [getattr(self, ('next_' if (T == 5 and t == 4) else fp) + 'kpt_branches')
                       for t, fp in enumerate(self.frame_prefixes)]   # HEAD:503 quirk kept
                poses = _frame_branches(brs, lvl, hs[lvl], 1, update_ref=reference)
                all_frame_poses = poses
                aux_poses = [None if t == c else poses[:, t * Q:(t + 1) * Q] for t in range(T)]
                outputs_kpt = poses[:, c * Q:(c + 1) * Q]
            else:
                # (tmp + inverse_sigmoid(reference)).sigmoid(): one launch on the device
                outputs_kpt = _ref_update(mlp_rows(self.kpt_branches[lvl], hs[lvl]),
                                          reference[:, c * Q:(c + 1) * Q])
            outputs_class = mlp_rows(self.cls_branches[lvl], hs[lvl])
            output_sigma = mlp_rows(self.dec_fc_sigma_branches[lvl], hs[lvl], act='sigmoid')
            outputs_classes.append(outputs_class)
            outputs_kpts.append(outputs_kpt)
            output_sigmas.append(output_sigma)
        one = len(outputs_classes) == 1        # a one-level stack is a view, not a copy
        stack = (lambda ts: ts[0].unsqueeze(0)) if one else torch.stack
        # (the encoder-side predictions only feed training losses: left un-activated at inference)
        lazy = only_last and not self.training
        return dict(all_cls_scores=stack(outputs_classes),
                    all_kpt_preds=stack(outputs_kpts),
                    all_sigma_preds=stack(output_sigmas),
                    enc_cls_scores=enc_outputs_class,
                    enc_kpt_preds=None if lazy else enc_outputs_kpt.sigmoid(),
                    enc_sigma_preds=None if lazy else enc_outputs_sigma.sigmoid(), memory=memory,
                    mlvl_masks=mlvl_masks, has_padding=has_padding, aux_poses=aux_poses,
                    all_frame_poses=all_frame_poses, last_level_only=lazy,
                    frame_shard=shard, refine_value_cache=refine_value_cache,
                    hs=hs, init_reference=init_reference, inter_references=inter_references)

    # HEAD:569-674 (inference branch) ----------------------------------------
    def forward_refine(self, memory, mlvl_masks, frame_poses, img_inds, has_padding=True,
                       frame_shard=None, value_cache=None, last_level_only=None):
        """frame_poses: list of T tensors [Ntot, 2K] (centre = selected kpt preds).
        Returns (kpts [Ntot, K, 2] normalised, score [Ntot, K, 1], sigma [Ntot, K, 2]) of the last
        refine layer plus all intermediates."""
        T = self.num_frames
        c = T // 2
        # frame-major, HEAD:610 (a [T, Ntot, 2K] tensor: already concatenated by the gather)
        pos_kpt_preds = frame_poses.flatten(0, 1) if torch.is_tensor(frame_poses) else \
            torch.cat(frame_poses, dim=0)
        S = memory.size(0)
        Tl = T if frame_shard is None else frame_shard.n_local
        mem4 = memory.reshape(S, -1, Tl, memory.size(-1))  # [S, B, T(_loc), C] view
        extra = {}
        if value_cache is not None:   # streaming: (per-layer projected-value caches, frame table)
            extra = dict(values_projected=value_cache[0], value_frame_table=value_cache[1])
        hs, init_reference, inter_references = self.transformer.forward_refine(
            mlvl_masks, mem4, pos_kpt_preds.detach(), img_inds,
            frame_kpt_branches=self._branches('refine_kpt_branches'), has_padding=has_padding,
            frame_shard=frame_shard, **extra)
        hs = hs.permute(0, 2, 1, 3)
        outs_kpt, outs_sigma, outs_score = [], [], []
        # (only the last refine level is read outside training, HEAD:1430-1438: one-level stacks)
        n_lvl = hs.shape[0]
        only_last = self.eval_last_level_only if last_level_only is None else bool(last_level_only)
        lazy = only_last and not self.training
        levels = [n_lvl - 1] if lazy else range(n_lvl)
        for lvl in levels:
            reference = init_reference if lvl == 0 else inter_references[lvl - 1]
            n = reference.shape[0] // T
            h = hs[lvl] if hs[lvl].is_contiguous() else hs[lvl].contiguous()   # (one layout copy for both branches)
            tmp_kpt = mlp_rows(self.refine_kpt_branches[lvl], h)
            tmp_sigma = mlp_rows(self.refine_fc_sigma_branches[lvl], h, act='sigmoid')
            if not lazy:    # (the refine score only feeds the training loss, HEAD:640-674: get_bboxes never reads it)
                outs_score.append(torch.mean(1 - tmp_sigma, dim=2, keepdim=True))
            outs_kpt.append(_ref_update(tmp_kpt, reference[c * n:(c + 1) * n]))
            outs_sigma.append(tmp_sigma)
        stack = (lambda ts: ts[0].unsqueeze(0)) if len(outs_kpt) == 1 else torch.stack
        return stack(outs_kpt), (None if lazy else stack(outs_score)), stack(outs_sigma), hs

    @staticmethod
    def get_p(output_regression_sigma, p_x=0.2):
        """HEAD:1531-1535."""
        p = 1 - torch.exp(-(p_x / output_regression_sigma))
        p = p[:, :, 0] * p[:, :, 1]
        return p[:, :, None] * 0.7

    def _cached(self, key, make):
        """Shape-derived constant tensors (no dependence on the inputs), built once per key."""
        if key not in self._consts:
            self._consts[key] = make()
        return self._consts[key]

    def _meta_scales(self, img_metas, device):
        """[B,1,1,2] (w, h) and scale-factor tensors of the clips, cached on the device (a
        host->device copy per call would also break hipGraph capture)."""
        key = ('scales', str(device),
               tuple((tuple(m['img_shape'][:2]), tuple(float(v) for v in m['scale_factor'][:2]))
                     for m in img_metas))
        if key not in self._consts:
            B = len(img_metas)
            wh = torch.tensor([[m['img_shape'][1], m['img_shape'][0]] for m in img_metas],
                              dtype=torch.float32, device=device).view(B, 1, 1, 2)
            sf = torch.tensor([list(m['scale_factor'][:2]) for m in img_metas],
                              dtype=torch.float32, device=device).view(B, 1, 1, 2)
            self._consts[key] = (wh, sf)
        return self._consts[key]

    def _sigmas(self, device):
        key = ('oks_sigmas', str(device))
        if key not in self._consts:
            K = self.num_keypoints
            if K != len(OKS_SIGMAS_POSETRACK15):
                raise ValueError('the reference hard-codes 15 PoseTrack OKS sigmas (HEAD:1400); '
                                 f'num_keypoints={K} has no NMS sigmas')
            self._consts[key] = torch.tensor(OKS_SIGMAS_POSETRACK15, dtype=torch.float64,
                                             device=device) / 10.0
        return self._consts[key]

    # HEAD:1304-1505, batched over clips ------------------------------------
    def get_bboxes(self, outs, img_metas, rescale=False, force_score_topk=None, taps=None):
        """-> dict of fixed-shape device tensors: bboxes [B,N,5], labels [B,N], kpts [B,N,K,3],
        keep [B,N] (int32, OKS-NMS survivors), all ordered by descending score."""
        cls_scores = outs['all_cls_scores'][-1]   # [B, Q, C]
        kpt_preds = outs['all_kpt_preds'][-1]     # [B, Q, 2K]
        B, Q = cls_scores.shape[:2]
        T, K = self.num_frames, self.num_keypoints
        c = T // 2
        N = self.test_cfg.get('max_per_img', self.num_query)
        assert self.loss_cls.use_sigmoid
        cls_score = cls_scores.sigmoid().view(B, -1)
        fused = cls_score.is_cuda and cls_score.dtype == torch.float32 and not torch.is_grad_enabled()
        if fused and cls_score.shape[1] <= 32768 and N <= 1024:
            scores, indexs = ops.topk_rows(cls_score, N)          # one launch
        else:
            scores, indexs = cls_score.topk(N, dim=1)
        if force_score_topk is not None:
            indexs = force_score_topk
            scores = torch.gather(cls_score, 1, indexs)
        shard = outs.get('frame_shard')
        if shard is not None:
            # a discrete choice made from replicated arithmetic: take rank 0's, so that a
            # last-bit difference between ranks can never make them decode different poses
            from . import dist as pdist
            indexs = pdist.broadcast_from(indexs.contiguous(), 0, shard.group)
            scores = torch.gather(cls_score, 1, indexs)
        if self.num_classes == 1:
            bbox_index = indexs                                      # [B, N]
            det_labels = self._cached(('labels0', B, N, str(indexs.device)),
                                      lambda: torch.zeros((B, N), dtype=indexs.dtype, device=indexs.device))
        else:
            det_labels = indexs % self.num_classes
            bbox_index = indexs // self.num_classes
        all_poses = outs.get('all_frame_poses')     # [B, T*Q, 2K]: every frame's poses, frame-major
        if fused and all_poses is not None and all_poses.is_contiguous():
            # the N selected queries of all T frames in one gather, already frame-major
            frame_poses = ops.gather_frame_poses(all_poses, bbox_index.contiguous(), T)
        else:
            gidx = bbox_index.unsqueeze(-1).expand(-1, -1, 2 * K)
            frame_poses = []
            for t in range(T):
                src = kpt_preds if t == c else outs['aux_poses'][t]
                frame_poses.append(torch.gather(src, 1, gidx).reshape(B * N, 2 * K))
        img_inds = self._cached(('img_inds', B, N, str(cls_scores.device)),
                                lambda: torch.arange(B, device=cls_scores.device).repeat_interleave(N))
        r_kpts, r_scores, r_sigmas, r_hs = self.forward_refine(
            outs['memory'], outs['mlvl_masks'], frame_poses, img_inds,
            has_padding=outs['has_padding'], frame_shard=outs.get('frame_shard'),
            value_cache=outs.get('refine_value_cache'), last_level_only=outs.get('last_level_only'))
        det_kpts = r_kpts[-1].view(B, N, K, 2)
        det_sigmas = r_sigmas[-1].view(B, N, K, 2)
        if taps is not None:
            taps.update(score_topk=indexs, refine_hs=r_hs, refine_kpts=det_kpts.clone(),
                        refine_sigma=det_sigmas)
        dev = det_kpts.device
        wh, sf = self._meta_scales(img_metas, dev)
        # (det_sigmas may be the 2-column slice of the sigma branch's padded GEMM output: evenly strided rows)
        sig_rows = det_sigmas.stride(3) == 1 and det_sigmas.stride(1) == K * det_sigmas.stride(2) \
            and det_sigmas.stride(0) == N * det_sigmas.stride(1)
        if fused and K <= 64 and det_kpts.is_contiguous() and sig_rows:
            # pixels / clamp / rescale / box / RLE confidence / key-point scores: one launch
            scores = scores.contiguous()
            det_kpts, det_bboxes = ops.pose_finalize(det_kpts, det_sigmas, scores, wh.view(B, 2),
                                                     sf.view(B, 2) if rescale else None)
            keep, order = ops.oks_nms(det_kpts, scores, self._sigmas(dev), self.oks_thresh)
            return dict(bboxes=det_bboxes, labels=det_labels, kpts=det_kpts, keep=keep, order=order,
                        scores=scores, score_index=indexs)
        det_kpts = det_kpts * wh
        det_kpts = torch.minimum(det_kpts.clamp(min=0), wh)
        if rescale:
            det_kpts = det_kpts / sf
        x1 = det_kpts[..., 0].min(dim=2, keepdim=True)[0]
        y1 = det_kpts[..., 1].min(dim=2, keepdim=True)[0]
        x2 = det_kpts[..., 0].max(dim=2, keepdim=True)[0]
        y2 = det_kpts[..., 1].max(dim=2, keepdim=True)[0]
        det_bboxes = torch.cat([x1, y1, x2, y2, scores.unsqueeze(-1)], dim=2)
        p = self.get_p(det_sigmas.view(B * N, K, 2)).view(B, N, K, 1)
        p5 = p**5
        det_kpts = (det_kpts * p5) / (p5 + 1e-10)
        kpt_scores = scores[:, :, None, None] * p
        det_kpts = torch.cat((det_kpts, kpt_scores), dim=3).contiguous()
        keep, order = ops.oks_nms(det_kpts, scores.contiguous(), self._sigmas(dev), self.oks_thresh)
        return dict(bboxes=det_bboxes, labels=det_labels, kpts=det_kpts, keep=keep, order=order,
                    scores=scores, score_index=indexs)

    def simple_test_bboxes(self, feats, img_metas, rescale=False):
        """HEAD:1507-1529 -> list (per clip) of (det_bboxes [n,5], det_labels [n], det_kpts [n,K,3])."""
        outs = self.forward(feats, img_metas, last_level_only=True)
        res = self.get_bboxes(outs, img_metas, rescale=rescale)
        return self.results_to_list(res)

    simple_test = simple_test_bboxes

    @staticmethod
    def results_to_list(res):
        """Fixed-shape device results -> the reference's per-clip variable-length tuples
        (one device->host sync, at the very end of the path)."""
        keep = res['keep'].bool()
        out = []
        for b in range(keep.shape[0]):
            k = keep[b]
            out.append((res['bboxes'][b][k], res['labels'][b][k], res['kpts'][b][k]))
        return out
