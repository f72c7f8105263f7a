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


def main():
    T = int(sys.argv[1])
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
