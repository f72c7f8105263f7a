"""``opera.VideoPoseV1`` (L6): backbone -> neck -> head, ``simple_test`` restated from
opera/models/detectors/videoposev1.py:18-190 and mmdet SingleStageDetector.extract_feat
(third_party/mmdetection/mmdet/models/detectors/single_stage.py:47).

Native additions: ``forward_device`` keeps the whole clip batch on the device and returns
fixed-shape result tensors (no host sync); B >= 1 clips per call (the reference asserts B = 1,
videoposev1.py:175-177).
"""
import numpy as np
import torch

from .bricks import BaseModule
from .registry import DETECTORS, MMDET_MODELS


def bbox_kpt2result(bboxes, labels, kpts, num_classes):
    """opera/core/keypoint/transforms.py:132-154."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)], \
            [np.zeros((0, kpts.size(1), 3), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
        kpts = kpts.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)], \
        [kpts[labels == i, :, :] for i in range(num_classes)]


@DETECTORS.register_module()
class VideoPoseV1(BaseModule):

    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        backbone = dict(backbone)
        if pretrained:
            backbone['pretrained'] = pretrained
        # the reference detectors derive from mmdet's SingleStageDetector, whose __init__ builds
        # its parts through mmdet's registry (single_stage.py:29-38): bare names such as
        # 'HRNet' resolve in the mmdet scope, 'opera.X' walks up to the root and down again
        self.backbone = MMDET_MODELS.build(backbone)
        self.neck = MMDET_MODELS.build(neck) if neck is not None else None
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg)
        bbox_head.update(test_cfg=test_cfg)
        self.bbox_head = MMDET_MODELS.build(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg

    @property
    def with_neck(self):
        return self.neck is not None

    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    @torch.no_grad()
    def forward_device(self, img, img_metas, rescale=False, force_score_topk=None, strict=False,
                       **head_kwargs):
        """img [B, T, 3, H, W] on the device; img_metas: one dict per clip.  Returns the
        head's fixed-shape device result dict (see VideoPoseHeadMulFrames.get_bboxes).

        Frame-sharded multi-GPU: pass ``frame_shard=FrameShard(T, rank, world)`` and only the
        rank's frames, img [B, T_loc, 3, H, W] (frames t with t % world == rank, in order).

        strict=True: the forward runs under ``census.LaunchCensus`` (after one un-counted warm-up of these
        shapes, which builds the per-shape constant tables) and raises ``census.FallbackError`` if a torch /
        vendor compute operator (GEMM, convolution, attention, normalisation, pooling, ...) ran on a device
        tensor, i.e. if a module-level gate dropped off the hand-written path; the census of the last strict
        call stays in ``self.last_census``."""
        if strict:
            from .census import LaunchCensus
            key = (tuple(img.shape), str(img.device),
                   tuple((tuple(m['batch_input_shape']), tuple(m['img_shape'][:2])) for m in img_metas),
                   bool(rescale), force_score_topk is not None, tuple(sorted(head_kwargs)))
            warm = self.__dict__.setdefault('_strict_warm', set())
            if key not in warm:
                self.forward_device(img, img_metas, rescale=rescale, force_score_topk=force_score_topk,
                                    **head_kwargs)
                warm.add(key)
            with LaunchCensus(where=strict == 'where') as census:    # strict='where': record the call sites too
                res = self.forward_device(img, img_metas, rescale=rescale, force_score_topk=force_score_topk,
                                          **head_kwargs)
            self.last_census = census
            census.raise_on_fallback(f'{type(self).__name__}.forward_device(strict=True)')
            return res
        feat = self.extract_feat(img)
        head_kwargs.setdefault('last_level_only', True)   # get_bboxes reads [-1] only
        outs = self.bbox_head(feat, img_metas, **head_kwargs)
        return self.bbox_head.get_bboxes(outs, img_metas, rescale=rescale,
                                         force_score_topk=force_score_topk)

    @torch.no_grad()
    def simple_test(self, img, img_metas, rescale=False):
        """videoposev1.py:159-190 -> per clip (bbox_results, kpt_results) lists of numpy arrays."""
        res = self.forward_device(img, img_metas, rescale=rescale)
        results_list = self.bbox_head.results_to_list(res)
        return [bbox_kpt2result(b, l, k, self.bbox_head.num_classes) for b, l, k in results_list]

    def forward(self, img, img_metas, return_loss=False, rescale=False, **kwargs):
        if return_loss:
            raise NotImplementedError('pavenet_amd is a forward (inference) path')
        if isinstance(img, (list, tuple)):  # mmdet forward_test nests one level per augmentation
            img, img_metas = img[0], img_metas[0]
        return self.simple_test(img, img_metas, rescale=rescale)
