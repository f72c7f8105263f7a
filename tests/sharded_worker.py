"""Worker of tests/test_model_gpu.py::test_frame_sharded_two_ranks: 2 processes share the one
GPU of the box (gloo backend: RCCL refuses two ranks on one device), each runs the
frame-sharded model on its frames; rank 0 compares with the un-sharded model."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def vs_oracle(T, gemm_mode, tol_px):
    """BASELINE configs[4] shape (long clip, frame-sharded, optional fp16 projections) against
    the ORACLE: rank 0 runs oracle/pavenet_ref.py on the whole clip, its top-k selections are
    forced on every rank (near-ties under random weights), the sharded product's keypoints must
    be within `tol_px` of the oracle's."""
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from oracle import pavenet_ref as R
    from oracle.seeded import seeded_array, seeded_state_dict
    from pavenet_amd import bricks
    from pavenet_amd.dist import FrameShard, broadcast_from
    from pavenet_amd.models import build_model, videopose_r50_cfg
    N = 12
    m = build_model(videopose_r50_cfg(num_frames=T, max_per_img=N))
    shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(seeded_state_dict(shapes, like=m.state_dict()))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    H, W = 128, 160
    img = torch.from_numpy(seeded_array(f'sharded.oracle.{T}', (1, T, 3, H, W)))
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=(1., 1., 1., 1.))]
    sel_p = torch.zeros(1, 300, dtype=torch.long, device='cuda')
    sel_s = torch.zeros(1, N, dtype=torch.long, device='cuda')
    exp = None
    if rank == 0:
        taps = {}
        with torch.no_grad():
            exp = R.videopose_simple_test(sd, dict(num_frames=T, num_keypoints=15, num_query=300,
                                                   max_per_img=N), img, taps=taps)
        sel_p.copy_(taps['topk_idx'])
        sel_s.copy_(taps['score_topk_idx'].view(1, -1))
    broadcast_from(sel_p, 0)
    broadcast_from(sel_s, 0)
    shard = FrameShard(T, rank, world)
    bricks.set_gemm_mode(gemm_mode)
    bricks._GEMM['min_rows'] = 1   # exercise the hand-written GEMM at test sizes
    with torch.no_grad():
        feat = m.extract_feat(img[:, shard.local].contiguous().cuda())
        outs = m.bbox_head(feat, metas, frame_shard=shard, force_topk_proposals=sel_p)
        res = m.bbox_head.get_bboxes(outs, metas, force_score_topk=sel_s)
        (gb, gl, gk), = m.bbox_head.results_to_list(res)
    torch.cuda.synchronize()
    ok = True
    if rank == 0:
        eb, el, ek = exp
        if gk.shape != ek.shape:
            print(f'MISMATCH keep set: {tuple(gk.shape)} vs oracle {tuple(ek.shape)}', flush=True)
            ok = False
        else:
            d = (gk.cpu()[..., :2] - ek[..., :2]).abs().max().item()
            ds = (gk.cpu()[..., 2] - ek[..., 2]).abs().max().item()
            print(f'T={T} gemm={gemm_mode} world={world}: max |kpt - oracle| = {d:.4g} px, '
                  f'score {ds:.3g}', flush=True)
            if not (d <= tol_px and ds <= 1e-2):
                print(f'MISMATCH kpts: {d} px > {tol_px}', flush=True)
                ok = False
        print('sharded == oracle:', ok, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def main():
    T = int(sys.argv[1])
    if len(sys.argv) > 2:
        return vs_oracle(T, sys.argv[2], float(sys.argv[3]))
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from oracle.seeded import seeded_array, seeded_state_dict
    from pavenet_amd.dist import FrameShard, broadcast_from
    from pavenet_amd.models import build_model, videopose_r50_cfg
    m = build_model(videopose_r50_cfg(num_frames=T, max_per_img=12))
    shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(seeded_state_dict(shapes, like=m.state_dict()))
    m = m.cuda().eval()
    head = m.bbox_head
    B, H, W = 2, 128, 160
    img = torch.from_numpy(seeded_array(f'sharded.{T}', (B, T, 3, H, W))).cuda()
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=(1., 1., 1., 1.))
             for _ in range(B)]
    shard = FrameShard(T, rank, world)
    with torch.no_grad():
        # un-sharded run first, on every rank; its proposal / score selections (rank 0's) are
        # then forced in the sharded run, because with random weights both top-k sit on
        # near-ties
        outs_f = head(m.extract_feat(img), metas)
        full = head.get_bboxes(outs_f, metas)
        sel_p = head.transformer.last_topk_proposals.clone()
        sel_s = full['score_index'].clone()
        broadcast_from(sel_p, 0)
        broadcast_from(sel_s, 0)
        outs_f = head(m.extract_feat(img), metas, force_topk_proposals=sel_p)
        full = head.get_bboxes(outs_f, metas, force_score_topk=sel_s)
        outs_s = head(m.extract_feat(img[:, shard.local].contiguous()), metas, frame_shard=shard,
                      force_topk_proposals=sel_p)
        res = head.get_bboxes(outs_s, metas, force_score_topk=sel_s)
    torch.cuda.synchronize()
    ok = True
    for k in ('hs', 'inter_references', 'all_cls_scores', 'all_kpt_preds'):
        d = (outs_s[k] - outs_f[k]).abs().max().item()
        print(f'rank {rank}: {k} sharded-vs-full max abs {d:.3e}', flush=True)
    if rank == 0:
        for k in ('bboxes', 'kpts'):
            a, b = res[k].cpu().numpy(), full[k].cpu().numpy()
            if not np.allclose(a, b, rtol=1e-4, atol=1e-2):
                print(f'MISMATCH {k}: max abs {np.abs(a - b).max()}', flush=True)
                ok = False
        if not torch.equal(res['keep'], full['keep']):
            print('MISMATCH keep', flush=True)
            ok = False
        print('sharded == unsharded:', ok, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
