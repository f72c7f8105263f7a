"""Importing this module registers every class of the PAVE-Net forward path under the
reference's type names, and offers the re-stated model dicts (the reference's config files do
not travel to the GPU box; the dicts below keep their type names and kwargs)."""
from . import (backbones, bricks, deform_attn, detectors, heads, necks, petr,  # noqa: F401
               swin, transformer)
from .registry import build_model  # noqa: F401


def _decoder_layer(attn_type, **attn_kwargs):
    return dict(type='mmcv.DetrTransformerDecoderLayer',
                attn_cfgs=[dict(type='mmcv.MultiheadAttention', embed_dims=256, num_heads=8,
                                dropout=0.1),
                           dict(type=attn_type, embed_dims=256, **attn_kwargs)],
                feedforward_channels=1024, ffn_dropout=0.1,
                operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm'))


def videopose_r50_cfg(num_frames=3, num_keypoints=15, num_query=300, max_per_img=20,
                      enc_layers=6, dec_layers=3, refine_layers=2, depth=50):
    """Model dict equivalent to configs/videopose/2025-5-11/
    2025_5_11_res50_num_frames_3_posetrack17_layer_num_3.py:8-138 (T = 3) and
    configs/videopose/2025-2-7/2025_2_7_res50_num_frames_5_posetrack17.py (T = 5); other odd T
    use the generalised ``...MulFrames...`` type names."""
    T = num_frames
    if T == 3:
        pose_attn = 'opera.MulFramesMultiScaleDeformablePoseAttentionNumFrames3'
        joint_attn = 'mmcv.MulFramesMultiScaleDeformableAttentionNumFrames3'
        dec_type, ref_type = 'opera.VideoPoseTransformerDecoderV2', 'mmcv.DeformableDetrTransformerDecoderV1'
        pk, jk = dict(num_frames=3), dict(num_frames=3)
    elif T == 5:
        pose_attn = 'opera.MulFramesMultiScaleDeformablePoseAttentionNumFrames5'
        joint_attn = 'mmcv.MulFramesMultiScaleDeformableAttentionNumFrames5'
        dec_type, ref_type = 'opera.VideoPoseTransformerDecoderV2_1', 'mmcv.DeformableDetrTransformerDecoderV1_2'
        pk, jk = {}, {}
    else:
        pose_attn = 'opera.MulFramesMultiScaleDeformablePoseAttention'
        joint_attn = 'mmcv.MulFramesMultiScaleDeformableAttention'
        dec_type = 'opera.VideoPoseTransformerDecoderMulFrames'
        ref_type = 'mmcv.DeformableDetrTransformerDecoderMulFrames'
        pk, jk = dict(num_frames=T), dict(num_frames=T)
    return dict(
        type='opera.VideoPoseV1',
        backbone=dict(type='mmdet.ResNet', input_type='mul_frames', depth=depth, num_stages=4,
                      out_indices=(1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True,
                      style='pytorch'),
        neck=dict(type='mmdet.ChannelMapper', in_channels=[512, 1024, 2048], kernel_size=1,
                  out_channels=256, act_cfg=None, norm_cfg=dict(type='GN', num_groups=32),
                  num_outs=4),
        bbox_head=dict(
            type='opera.VideoPoseHeadMulFrames', num_frames=T, num_keypoints=num_keypoints,
            num_query=num_query, num_classes=1, in_channels=2048, sync_cls_avg_factor=True,
            with_kpt_refine=True, as_two_stage=True,
            transformer=dict(
                type='opera.VideoPoseTransformerMulFrames', num_keypoints=num_keypoints,
                num_frames=T,
                encoder=dict(
                    type='mmcv.DetrTransformerEncoder', num_layers=enc_layers,
                    transformerlayers=dict(
                        type='mmcv.BaseTransformerLayer',
                        attn_cfgs=dict(type='mmcv.MultiScaleDeformableAttention', embed_dims=256),
                        feedforward_channels=1024, ffn_dropout=0.1,
                        operation_order=('self_attn', 'norm', 'ffn', 'norm'))),
                decoder=dict(type=dec_type, num_keypoints=num_keypoints, num_layers=dec_layers,
                             return_intermediate=True,
                             transformerlayers=_decoder_layer(pose_attn, num_points=num_keypoints,
                                                              **pk)),
                refine_decoder=dict(type=ref_type, num_layers=refine_layers,
                                    return_intermediate=True,
                                    transformerlayers=_decoder_layer(joint_attn, im2col_step=128,
                                                                     **jk))),
            positional_encoding=dict(type='mmcv.SinePositionalEncoding', num_feats=128,
                                     normalize=True, offset=-0.5),
            loss_cls=dict(type='mmdet.FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                          loss_weight=0.5),
            loss_kpt=dict(type='opera.RLELoss', loss_weight=1.0),
            loss_kpt_rpn=dict(type='opera.RLELoss', loss_weight=1.0),
            loss_oks=dict(type='opera.OKSLoss', num_keypoints=num_keypoints, loss_weight=0.0),
            loss_hm=dict(type='opera.CenterFocalLoss', loss_weight=0.0),
            loss_kpt_refine=dict(type='opera.RLELoss', loss_weight=1.0),
            loss_oks_refine=dict(type='opera.OKSLoss', num_keypoints=num_keypoints,
                                 loss_weight=0.0)),
        train_cfg=None,
        test_cfg=dict(max_per_img=max_per_img))


def petr_r50_cfg(num_keypoints=17, num_query=300, max_per_img=100, head='opera.PETRHead',
                 depth=50):
    """Model dict equivalent to configs/petr/petr_r50_16x2_100e_coco.py:4-115 (PETRHead, K = 17)
    and configs/vedpose/single_frame_posetrack_resnet50_inference.py (VedPoseHeadV2, K = 15)."""
    K = num_keypoints
    enc = lambda levels: dict(  # noqa: E731
        type='mmcv.DetrTransformerEncoder', num_layers=6 if levels == 4 else 1,
        transformerlayers=dict(
            type='mmcv.BaseTransformerLayer',
            attn_cfgs=dict(type='mmcv.MultiScaleDeformableAttention', embed_dims=256,
                           num_levels=levels),
            feedforward_channels=1024, ffn_dropout=0.1,
            operation_order=('self_attn', 'norm', 'ffn', 'norm')))
    return dict(
        type='opera.PETR',
        backbone=dict(type='mmdet.ResNet', depth=depth, num_stages=4, out_indices=(1, 2, 3),
                      frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=False),
                      norm_eval=True, style='pytorch'),
        neck=dict(type='mmdet.ChannelMapper', in_channels=[512, 1024, 2048], kernel_size=1,
                  out_channels=256, act_cfg=None, norm_cfg=dict(type='GN', num_groups=32),
                  num_outs=4),
        bbox_head=dict(
            type=head, num_keypoints=K, num_query=num_query, num_classes=1, in_channels=2048,
            sync_cls_avg_factor=True, with_kpt_refine=True, as_two_stage=True,
            transformer=dict(
                type='opera.PETRTransformer', num_keypoints=K, encoder=enc(4),
                decoder=dict(type='opera.PetrTransformerDecoder', num_keypoints=K, num_layers=3,
                             return_intermediate=True,
                             transformerlayers=_decoder_layer(
                                 'opera.MultiScaleDeformablePoseAttention', num_points=K)),
                hm_encoder=enc(1),
                refine_decoder=dict(type='mmcv.DeformableDetrTransformerDecoder', num_layers=2,
                                    return_intermediate=True,
                                    transformerlayers=_decoder_layer(
                                        'mmcv.MultiScaleDeformableAttention'))),
            positional_encoding=dict(type='mmcv.SinePositionalEncoding', num_feats=128,
                                     normalize=True, offset=-0.5),
            loss_cls=dict(type='mmdet.FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                          loss_weight=2.0)),
        train_cfg=None,
        test_cfg=dict(max_per_img=max_per_img))


HRNET_W48_EXTRA = dict(
    stage1=dict(num_modules=1, num_branches=1, block='BOTTLENECK', num_blocks=(4,),
                num_channels=(64,)),
    stage2=dict(num_modules=1, num_branches=2, block='BASIC', num_blocks=(4, 4),
                num_channels=(48, 96)),
    stage3=dict(num_modules=4, num_branches=3, block='BASIC', num_blocks=(4, 4, 4),
                num_channels=(48, 96, 192)),
    stage4=dict(num_modules=3, num_branches=4, block='BASIC', num_blocks=(4, 4, 4, 4),
                num_channels=(48, 96, 192, 384)))


def with_hrnet_w48(model_cfg):
    """Swap the backbone / neck of a model dict for HRNet-w48 as in
    configs/petr/petr_hrnetw48_16x2_100e_coco.py:7-45 (BASELINE configs[3]: the reference has
    no HRNet PAVE-Net config; this composes its HRNet backbone dict with the MulFrames head)."""
    import copy
    cfg = copy.deepcopy(model_cfg)
    cfg['backbone'] = dict(type='HRNet', in_channels=3, extra=copy.deepcopy(HRNET_W48_EXTRA))
    cfg['neck']['in_channels'] = [96, 192, 384]
    return cfg


def with_swin_l(model_cfg, num_frames=None):
    """Swin-L backbone / neck as in configs/videopose/2025-2-7/
    2025_2_7_swin_num_frames_3_posetrack17.py:12-35."""
    import copy
    cfg = copy.deepcopy(model_cfg)
    cfg['backbone'] = dict(type='mmdet.SwinTransformer', num_frames=num_frames, embed_dims=192,
                           depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=7,
                           mlp_ratio=4, qkv_bias=True, qk_scale=None, drop_rate=0.,
                           attn_drop_rate=0., drop_path_rate=0.3, patch_norm=True,
                           out_indices=(1, 2, 3), with_cp=False)
    cfg['neck']['in_channels'] = [384, 768, 1536]
    return cfg
