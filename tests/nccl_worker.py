"""Worker of tests/test_dist_gpu.py: ONE fresh process, WORLD_SIZE = 1, backend "nccl" (= RCCL on
ROCm), every collective on DEVICE tensors -- the exact calls the N > 1 paths of bench.py and
pavenet_amd/dist.py make (the reference's counterpart: tools/dist_test.sh:8-10 ->
opera/apis/test.py:247-276).  A one-GPU box cannot hold two RCCL ranks, so the multi-rank tests
run on gloo; this one makes sure the first 8-GPU job is not the first time an RCCL communicator
is created, a device all-gather is enqueued behind this package's kernels or the group is torn
down."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    assert int(os.environ['WORLD_SIZE']) == 1 and int(os.environ['RANK']) == 0
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', device_id=dev)
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    from pavenet_amd import dist as pd
    from pavenet_amd import ops

    g = torch.Generator(device=dev).manual_seed(7)
    B, N, K = 4, 20, 15
    res = dict(bboxes=torch.randn(B, N, 5, device=dev, generator=g),
               kpts=torch.randn(B, N, K, 3, device=dev, generator=g),
               keep=torch.rand(B, N, device=dev, generator=g) > 0.5)
    # (1) clip-parallel result exchange: all_gather_into_tensor of the packed results
    got = pd.all_gather_results(res)
    assert got.is_cuda and tuple(got.shape) == (1, B, N * (5 + 3 * K + 1))
    assert torch.equal(got[0], pd.pack_results(res))
    back = pd.unpack_results(got[0], N, K)
    assert torch.equal(back['kpts'], res['kpts']) and torch.equal(back['keep'], res['keep'])
    # (2) frame-sharded merge: the partial rows come from THIS package's fused kernel on the same
    # stream the collective is enqueued on (stream ordering between our launches and RCCL's)
    levels = [(12, 20), (6, 10), (3, 5), (2, 3)]
    shapes = torch.as_tensor(levels, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    T, U = 3, 40
    value = torch.randn(T, S, 8, 32, device=dev, generator=g)
    proj = torch.randn(U, T * 8 * 16 * 3, device=dev, generator=g)
    ref = torch.rand(T, U, 4, 2, device=dev, generator=g)
    full = ops.deform_attn_grid_fused(value, shapes, lsi, proj, ref, T=T, n_clips=1, units_per_clip=U)
    row, smax, ssum = ops.deform_attn_grid_fused(value, shapes, lsi, proj, ref, T=T, n_clips=1,
                                                 units_per_clip=U, return_stats=True)
    merged = pd.all_gather_merge(row.reshape(U, 256), smax.reshape(U, 8), ssum.reshape(U, 8))
    assert merged.is_cuda
    torch.testing.assert_close(merged, full.reshape(U, 256), rtol=1e-5, atol=1e-6)
    # (3) broadcast of the proposal selection, (4) object all-gather, (5) the bench's timing reduce
    sel = torch.arange(300, device=dev).view(1, 300)
    assert torch.equal(pd.broadcast_from(sel.clone(), 0), sel)
    names = [None]
    dist.all_gather_object(names, f'cuda:0 {torch.cuda.get_device_name(0)}')
    assert names[0].startswith('cuda:0')
    tt = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert float(tt.item()) == 1.25
    ids = torch.full((1,), 0, dtype=torch.int64, device=dev)
    seen = torch.empty((1,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(seen, ids)
    assert seen.tolist() == [0]
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print('nccl world-size-1 collectives on device tensors: ok', flush=True)


if __name__ == '__main__':
    main()
