"""oracle/gen_golden.py -- TEST INFRASTRUCTURE, runs ONLY in the build container.

Imports the real reference (zgspose/PAVENet at /root/reference) through
oracle/ref_shim.py and writes small golden input/output vectors to
tests/golden/*.npz.  The reference itself never travels; these vectors do.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [op|modules|e2e|all]

Parameters are never stored: both sides regenerate them with the name-seeded
recipe in oracle/seeded.py from the (key -> shape) table kept in each file.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from seeded import seeded_array, seeded_state_dict  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
LEVELS = [(12, 20), (6, 10), (3, 5), (2, 3)]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v)
                                 for k, v in arrs.items()})
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def _load_seeded(module, salt=0):
    sd = module.state_dict()
    shapes = {k: list(v.shape) for k, v in sd.items()}
    module.load_state_dict(seeded_state_dict(shapes, salt, like=sd))
    return json.dumps(shapes)


def _shapes_lsi(levels):
    shapes = torch.as_tensor(levels, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return shapes, lsi


def _pad_mask(n, levels, frac_w):
    """Right-padding mask per level like HEAD:429-445 produces: [n, S] bool."""
    ms = []
    for (h, w) in levels:
        m = torch.zeros(n, h, w, dtype=torch.bool)
        for i in range(n):
            vw = max(1, int(round(w * frac_w[i])))
            m[i, :, vw:] = True
        ms.append(m.flatten(1))
    return torch.cat(ms, 1)


# ---------------------------------------------------------------------------
def gen_op():
    """Golden vectors for the sampler from the reference's own PyTorch implementation
    (MO:92-149), incl. mmcv's seed-3 known-answer inputs (test_ms_deformable_attn.py:54-66)."""
    MO, _ = ref_shim.install() or (None, None)
    import mmcv.ops.multi_scale_deform_attn as MO  # noqa: N812
    ref = MO.multi_scale_deformable_attn_pytorch
    out = {}
    # (1) mmcv seed-3 case, float and double
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = sum((H * W).item() for H, W in shapes)
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    aw = torch.rand(N, Lq, M, L, P) + 1e-5
    aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    out.update(s3_shapes=shapes, s3_value=value, s3_loc=loc, s3_aw=aw,
               s3_out_f32=ref(value, shapes, loc, aw),
               s3_out_f64=ref(value.double(), shapes, loc.double(), aw.double()))
    # (2) the three call shapes of the hot path at reduced S, with out-of-range locations
    shapes, lsi = _shapes_lsi(LEVELS)
    S = int(shapes.prod(1).sum())
    cases = dict(enc=(1, 96, 8, 32, 4, 4), pose=(1, 10, 8, 32, 4, 15), joint=(3, 15, 8, 32, 4, 4),
                 odd=(2, 7, 3, 20, 4, 3), d71=(1, 5, 2, 71, 4, 2))
    for name, (bs, Lq, M, D, L, P) in cases.items():
        value = _t(seeded_array(f'op.{name}.value', (bs, S, M, D)))
        # locations in [-0.3, 1.3]: a good share falls outside / on the border
        loc = _t(seeded_array(f'op.{name}.loc', (bs, Lq, M, L, P, 2), 0.45)) + 0.5
        aw = _t(seeded_array(f'op.{name}.aw', (bs, Lq, M, L * P))).softmax(-1).view(bs, Lq, M, L, P)
        out[f'{name}_value'], out[f'{name}_loc'], out[f'{name}_aw'] = value, loc, aw
        out[f'{name}_out'] = ref(value, shapes, loc, aw)
    out['levels'] = shapes
    _save('op_msda', **out)


# ---------------------------------------------------------------------------
def gen_modules():
    ref_shim.install()
    import mmcv.ops.multi_scale_deform_attn as MO  # noqa: N812
    import opera.models.utils.transformer as OT  # noqa: N812
    shapes, lsi = _shapes_lsi(LEVELS)
    S = int(shapes.prod(1).sum())
    C = 256

    with torch.no_grad():
        # a2: encoder self-attention
        m = MO.MultiScaleDeformableAttention(embed_dims=C).eval()
        keys = _load_seeded(m)
        bs = 2
        query = _t(seeded_array('enc.query', (S, bs, C)))
        pos = _t(seeded_array('enc.pos', (S, bs, C)))
        mask = _pad_mask(bs, LEVELS, [1.0, 0.8])
        vr = _t(np.array([[[1.0, 1.0]] * 4, [[0.8, 1.0]] * 4], dtype=np.float32))
        refp = OT.VideoPoseTransformerMulFrames.get_reference_points(shapes, vr, 'cpu')
        out = m(query, None, None, query_pos=pos, key_padding_mask=mask, reference_points=refp,
                spatial_shapes=shapes, level_start_index=lsi)
        _save('mod_enc_msda', keys=keys, query=query, pos=pos, mask=mask, ref=refp, out=out,
              levels=shapes)

        # a15: single-frame pose attention (PETR, K=17)
        K = 17
        m = OT.MultiScaleDeformablePoseAttention(embed_dims=C, num_points=K).eval()
        keys = _load_seeded(m)
        bs, Q = 2, 9
        query = _t(seeded_array('pose1.query', (Q, bs, C)))
        pos = _t(seeded_array('pose1.pos', (Q, bs, C)))
        value = _t(seeded_array('pose1.value', (S, bs, C)))
        mask = _pad_mask(bs, LEVELS, [1.0, 0.7])
        refp = torch.sigmoid(_t(seeded_array('pose1.ref', (bs, Q, 4, 2 * K), 1.0)))
        out = m(query, None, value, query_pos=pos, key_padding_mask=mask, reference_points=refp,
                spatial_shapes=shapes, level_start_index=lsi)
        _save('mod_pose_single', keys=keys, query=query, pos=pos, value=value, mask=mask,
              ref=refp, out=out, levels=shapes)

        # a5: pose-aware T-frame attention, T = 3 and 5
        K = 15
        for T, cls, B, Q in ((3, OT.MulFramesMultiScaleDeformablePoseAttentionNumFrames3, 2, 10),
                             (5, OT.MulFramesMultiScaleDeformablePoseAttentionNumFrames5, 1, 7)):
            kw = dict(num_frames=T) if T == 3 else {}  # the T=5 classes take no num_frames
            m = cls(embed_dims=C, num_points=K, **kw).eval()
            keys = _load_seeded(m)
            query = _t(seeded_array(f'pose{T}.query', (Q, B, C)))
            pos = _t(seeded_array(f'pose{T}.pos', (Q, B, C)))
            value = _t(seeded_array(f'pose{T}.value', (S, B * T, C)))
            mask = _pad_mask(B * T, LEVELS, [1.0 if (i // T) == 0 else 0.75 for i in range(B * T)])
            refp = torch.sigmoid(_t(seeded_array(f'pose{T}.ref', (B, T * Q, 4, 2 * K), 1.0)))
            out = m(query, None, value, query_pos=pos, key_padding_mask=mask,
                    reference_points=refp, spatial_shapes=shapes, level_start_index=lsi)
            _save(f'mod_pose_t{T}', keys=keys, query=query, pos=pos, value=value, mask=mask,
                  ref=refp, out=out, levels=shapes)

        # a7: joint-decoder T-frame attention, T = 3 and 5
        for T, cls, N in ((3, MO.MulFramesMultiScaleDeformableAttentionNumFrames3, 5),
                          (5, MO.MulFramesMultiScaleDeformableAttentionNumFrames5, 3)):
            kw = dict(num_frames=T) if T == 3 else {}
            m = cls(embed_dims=C, im2col_step=128, **kw).eval()
            keys = _load_seeded(m)
            Kq = 15
            query = _t(seeded_array(f'joint{T}.query', (Kq, N, C)))
            pos = _t(seeded_array(f'joint{T}.pos', (Kq, N, C)))
            mem = _t(seeded_array(f'joint{T}.memory', (S, 1, T, C)))
            value = mem[:, [0] * N]  # replicated per pose, as OT:21498 does
            mask1 = _pad_mask(T, LEVELS, [0.85] * T)  # [T, S]
            mask = mask1[None].expand(N, -1, -1).contiguous()
            refp = torch.sigmoid(_t(seeded_array(f'joint{T}.ref', (T * N, Kq, 4, 2), 1.0)))
            out = m(query, None, value, query_pos=pos, key_padding_mask=mask,
                    reference_points=refp, spatial_shapes=shapes, level_start_index=lsi)
            _save(f'mod_joint_t{T}', keys=keys, query=query, pos=pos, memory=mem, mask=mask1,
                  ref=refp, out=out, levels=shapes)


# ---------------------------------------------------------------------------
E2E = {
    'e2e_videopose_r50_t3': ('configs/videopose/2025-5-11/'
                             '2025_5_11_res50_num_frames_3_posetrack17_layer_num_3.py', 3),
    'e2e_videopose_r50_t5': ('configs/videopose/2025-2-7/'
                             '2025_2_7_res50_num_frames_5_posetrack17.py', 5),
    'e2e_videopose_swinl_t3': ('configs/videopose/2025-2-7/'
                               '2025_2_7_swin_num_frames_3_posetrack17.py', 3),
}


def gen_e2e(which=None):
    for name, (cfg_path, T) in E2E.items():
        if which and which != name:
            continue
        model, cfg = ref_shim.build_reference_model(cfg_path)
        keys = _load_seeded(model)
        H, W = 128, 160
        img = _t(seeded_array(f'{name}.img', (1, T, 3, H, W)))
        # padded clip: valid area 120 x 150 inside a 128 x 160 batch canvas
        meta = [dict(batch_input_shape=(H, W), img_shape=(120, 150, 3),
                     scale_factor=(1., 1., 1., 1.))]
        taps = {}
        tr = model.bbox_head.transformer

        def enc_hook(mod, args, kwargs, out):
            taps['memory'] = out.permute(1, 0, 2).detach().clone()  # [B*T, S, C]

        def dec_hook(mod, args, kwargs, out):
            taps['hs'], taps['inter_references'] = out[0].detach().clone(), out[1].detach().clone()

        def ref_hook(mod, args, kwargs, out):
            taps['refine_hs'] = out[0].detach().clone()
            taps['refine_refs'] = out[1].detach().clone()
            taps['refine_init_ref'] = kwargs['reference_points'].detach().clone()

        head_forward = model.bbox_head.forward

        def tapped_forward(*a, **k):  # simple_test_bboxes calls self.forward directly
            out = head_forward(*a, **k)
            taps['cls_all'] = out[0].detach().clone()
            taps['kpt_all'] = out[1].detach().clone()
            taps['enc_cls'] = out[3].detach().clone()
            return out

        model.bbox_head.forward = tapped_forward
        hs = [tr.encoder.register_forward_hook(enc_hook, with_kwargs=True),
              tr.decoder.register_forward_hook(dec_hook, with_kwargs=True),
              tr.refine_decoder.register_forward_hook(ref_hook, with_kwargs=True)]
        with torch.no_grad():
            feats = model.extract_feat(img)
            res = model.bbox_head.simple_test(feats, meta, rescale=False)
        for h in hs:
            h.remove()
        det_bboxes, det_labels, det_kpts = res[0]
        Q = model.bbox_head.num_query
        enc_topk = torch.topk(taps['enc_cls'][..., 0], Q, dim=1)[1]
        N = model.bbox_head.test_cfg['max_per_img']
        score_topk = taps['cls_all'][-1][0].sigmoid().view(-1).topk(N)[1]
        full = (name == 'e2e_videopose_r50_t3')  # others: keep the fixture small (centre-frame memory, no refine taps)
        extra = dict(memory=taps['memory'], neck3=feats[3], refine_hs=taps['refine_hs'][-1],
                     refine_refs=taps['refine_refs'], refine_init_ref=taps['refine_init_ref'],
                     kpt_last=taps['kpt_all'][-1]) if full else \
            dict(memory_center=taps['memory'][T // 2::T])
        _save(name, keys=keys, img=img, img_shape=np.array([120, 150, 3]),
              enc_topk=enc_topk, hs=taps['hs'], inter_references=taps['inter_references'],
              cls_last=taps['cls_all'][-1], score_topk=score_topk,
              det_bboxes=det_bboxes, det_labels=det_labels, det_kpts=det_kpts, **extra)


# ---------------------------------------------------------------------------
def _stats(t):
    t = t.detach().double()
    return [float(t.mean()), float(t.abs().max()), float(t.std())]


def gen_fullsize():
    """the BENCHMARK-size artefacts.  T = 3 R-50 PAVE-Net at 800 x 1344 (un-padded,
    as bench.py feeds it) run through the real reference: per-stage mean / abs-max / std, a few
    hundred sampled rows of `memory`, all proposal logits, `hs`, `inter_references`, final
    detections.  Neither image nor weights are stored (both name-seeded)."""
    name = 'full_videopose_r50_t3'
    cfg_path, T = E2E['e2e_videopose_r50_t3']

    def small(cfg):
        cfg.model['test_cfg'] = dict(max_per_img=20)

    model, cfg = ref_shim.build_reference_model(cfg_path, cfg_overrides=small)
    keys = _load_seeded(model)
    H, W = 800, 1344
    img = _t(seeded_array(f'{name}.img', (1, T, 3, H, W)))
    meta = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=(1., 1., 1., 1.))]
    taps, stats = {}, {}
    tr = model.bbox_head.transformer

    def bb_hook(mod, args, out):
        for i, o in enumerate(out):
            stats[f'backbone{i}'] = _stats(o)
        taps['c5_rows'] = out[-1][:, :, ::6, ::10].detach().clone()

    def layer_hook(i):
        def h(mod, args, kwargs, out):
            stats[f'enc_layer{i}'] = _stats(out)
        return h

    def enc_hook(mod, args, kwargs, out):
        taps['memory'] = out.permute(1, 0, 2).detach().clone()  # [B*T, S, C]

    def dec_hook(mod, args, kwargs, out):
        taps['hs'], taps['inter_references'] = out[0].detach().clone(), out[1].detach().clone()

    def ref_hook(mod, args, kwargs, out):
        taps['refine_hs'] = out[0].detach().clone()
        taps['refine_refs'] = out[1].detach().clone()

    head_forward = model.bbox_head.forward

    def tapped_forward(*a, **k):
        out = head_forward(*a, **k)
        taps['cls_all'] = out[0].detach().clone()
        taps['enc_cls'] = out[3].detach().clone()
        return out

    model.bbox_head.forward = tapped_forward
    hs = [model.backbone.register_forward_hook(bb_hook),
          tr.encoder.register_forward_hook(enc_hook, with_kwargs=True),
          tr.decoder.register_forward_hook(dec_hook, with_kwargs=True),
          tr.refine_decoder.register_forward_hook(ref_hook, with_kwargs=True)]
    hs += [l.register_forward_hook(layer_hook(i), with_kwargs=True)
           for i, l in enumerate(tr.encoder.layers)]
    with torch.no_grad():
        feats = model.extract_feat(img)
        res = model.bbox_head.simple_test(feats, meta, rescale=False)
    for h in hs:
        h.remove()
    for i, f in enumerate(feats):
        stats[f'neck{i}'] = _stats(f)
    stats['memory'] = _stats(taps['memory'])
    stats['hs'] = _stats(taps['hs'])
    stats['refine_hs'] = _stats(taps['refine_hs'])
    det_bboxes, det_labels, det_kpts = res[0]
    S = taps['memory'].shape[1]
    rows = np.sort(np.random.default_rng(0).choice(S, 160, replace=False))
    Q = model.bbox_head.num_query
    enc_topk = torch.topk(taps['enc_cls'][..., 0], Q, dim=1)[1]
    N = model.bbox_head.test_cfg['max_per_img']
    score_topk = taps['cls_all'][-1][0].sigmoid().view(-1).topk(N)[1]
    _save(name, keys=keys, stats=json.dumps(stats), rows=rows,
          memory_rows=taps['memory'][:, rows], neck3=feats[3], c5_rows=taps['c5_rows'],
          enc_cls=taps['enc_cls'][0, :, 0], enc_topk=enc_topk, hs_last=taps['hs'][-1],
          inter_references=taps['inter_references'], cls_last=taps['cls_all'][-1],
          score_topk=score_topk, refine_hs_last=taps['refine_hs'][-1],
          det_bboxes=det_bboxes, det_labels=det_labels, det_kpts=det_kpts)
    print(json.dumps(stats, indent=1))



def gen_rescale():
    """f3: the PUBLIC entry VideoPoseV1.simple_test(img, img_metas, rescale=True) with a non-unit
    scale_factor, through bbox_kpt2result (opera/core/keypoint/transforms.py:132-154): the
    per-class lists the reference returns for the T = 3 R-50 clip of `e2e_videopose_r50_t3`."""
    name = 'e2e_videopose_r50_t3'
    cfg_path, T = E2E[name]
    model, cfg = ref_shim.build_reference_model(cfg_path)
    keys = _load_seeded(model)
    H, W = 128, 160
    img = _t(seeded_array(f'{name}.img', (1, T, 3, H, W)))
    sf = (0.375, 0.4, 0.375, 0.4)  # (w, h, w, h) as mmdet's Resize writes it
    meta = [dict(batch_input_shape=(H, W), img_shape=(120, 150, 3), scale_factor=sf)]
    with torch.no_grad():
        (bbox_results, kpt_results), = model.simple_test(img, meta, rescale=True)
    assert len(bbox_results) == 1 and len(kpt_results) == 1  # one class (person)
    _save('e2e_videopose_r50_t3_rescale', keys=keys, scale_factor=np.array(sf, dtype=np.float32),
          bbox_results=bbox_results[0], kpt_results=kpt_results[0])



def gen_pipeline_shapes():
    """f4: the pure-Python parts of the reference's test pipeline that CAN run here (cv2 is absent,
    so mmcv.imresize / imnormalize / impad themselves cannot): mmcv.rescale_size (the size
    Resize(keep_ratio=True) asks cv2 for), the scale_factor mmdet's Resize records and the shape
    Pad(size_divisor) pads to (mmdet/datasets/pipelines/transforms.py Resize._resize_img /
    Pad._pad_img; configs/_base_/datasets/posetrack17_video_keypoint.py:71-84)."""
    ref_shim.install()
    from mmcv.image.geometric import rescale_size
    cases = []
    for (w, h) in [(1920, 1080), (1280, 720), (640, 480), (480, 640), (333, 500), (1333, 800),
                   (1344, 800), (1000, 1000), (427, 640), (3, 2000)]:
        for scale in [(1333, 800), (800, 1333), (1000, 600)]:
            new_size, factor = rescale_size((w, h), scale, return_scale=True)
            nw, nh = new_size
            # Resize._resize_img: w_scale = new_w / w, h_scale = new_h / h (after mmcv.imrescale)
            for div in (1, 32):
                ph = int(np.ceil(nh / div)) * div      # mmcv.impad_to_multiple
                pw = int(np.ceil(nw / div)) * div
                cases.append(dict(src_wh=[w, h], scale=list(scale), new_wh=[nw, nh],
                                  scale_factor=[nw / w, nh / h, nw / w, nh / h], divisor=div,
                                  pad_hw=[ph, pw], factor=float(factor)))
    path = os.path.join(OUT, 'pipeline_shapes.json')
    open(path, 'w').write('[\n' + ',\n'.join(json.dumps(c) for c in cases) + '\n]\n')
    print('wrote', path, len(cases), 'cases')


PETR_E2E = {
    'e2e_petr_r50': 'configs/petr/petr_r50_16x2_100e_coco.py',
    'e2e_vedpose_r50': 'configs/vedpose/single_frame_posetrack_resnet50_inference.py',
    'e2e_petr_hrnetw48': 'configs/petr/petr_hrnetw48_16x2_100e_coco.py',
}


def gen_petr(which=None):
    """Single-image PETR (BASELINE configs[0]) and the vedpose single-frame head."""
    for name, cfg_path in PETR_E2E.items():
        if which and which != name:
            continue

        def small(cfg):
            cfg.model['test_cfg'] = dict(max_per_img=20)

        model, cfg = ref_shim.build_reference_model(cfg_path, cfg_overrides=small)
        keys = _load_seeded(model)
        H, W = 128, 160
        img = _t(seeded_array(f'{name}.img', (1, 3, H, W)))
        meta = [dict(batch_input_shape=(H, W), img_shape=(120, 150, 3),
                     scale_factor=(1., 1., 1., 1.))]
        taps = {}
        tr = model.bbox_head.transformer

        def enc_hook(mod, args, kwargs, out):
            taps['memory'] = out.permute(1, 0, 2).detach().clone()

        def dec_hook(mod, args, kwargs, out):
            taps['hs'], taps['inter_references'] = out[0].detach().clone(), out[1].detach().clone()

        head_forward = model.bbox_head.forward

        def tapped_forward(*a, **k):
            out = head_forward(*a, **k)
            taps['cls_all'] = out[0].detach().clone()
            # enc_outputs_class [B, S, 1]: out[2] for PETRHead, out[3] for VedPoseHeadV2
            taps['enc_cls'] = next(o for o in out[2:4] if o.dim() == 3 and o.shape[-1] == 1
                                   ).detach().clone()
            return out

        model.bbox_head.forward = tapped_forward
        hs = [tr.encoder.register_forward_hook(enc_hook, with_kwargs=True),
              tr.decoder.register_forward_hook(dec_hook, with_kwargs=True)]
        with torch.no_grad():
            feats = model.extract_feat(img)
            res = model.bbox_head.simple_test(feats, meta, rescale=False)
        for h in hs:
            h.remove()
        det_bboxes, det_labels, det_kpts = res[0]
        N = model.bbox_head.test_cfg['max_per_img']
        score_topk = taps['cls_all'][-1][0].sigmoid().view(-1).topk(N)[1]
        enc_topk = torch.topk(taps['enc_cls'][..., 0], model.bbox_head.num_query, dim=1)[1]
        _save(name, keys=keys, img=img, img_shape=np.array([120, 150, 3]), enc_topk=enc_topk,
              memory=taps['memory'], hs=taps['hs'], inter_references=taps['inter_references'],
              cls_last=taps['cls_all'][-1], score_topk=score_topk, det_bboxes=det_bboxes,
              det_labels=det_labels, det_kpts=det_kpts)


if __name__ == '__main__':
    import logging
    logging.disable(logging.INFO)
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('op', 'all'):
        gen_op()
    if what in ('modules', 'all'):
        gen_modules()
    if what in ('e2e', 'all'):
        gen_e2e(sys.argv[2] if len(sys.argv) > 2 else None)
    if what in ('pipeline', 'all'):
        gen_pipeline_shapes()
    if what in ('rescale', 'all'):
        gen_rescale()
    if what in ('full', 'all'):
        gen_fullsize()
    if what in ('petr', 'all'):
        gen_petr(sys.argv[2] if len(sys.argv) > 2 else None)
