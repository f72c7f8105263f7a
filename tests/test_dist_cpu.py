"""Multi-process (world_size 2, gloo, CPU) tests of the N > 1 host logic: clip-parallel result
exchange and the exact merge of frame-sharded partial softmax rows."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partial(logits, vals, frames):
    """Reference partial attention of a frame subset: (row normalised by its own sum, max, sum).
    logits [U, T, H, P], vals [U, T, H, P, D]."""
    lg = logits[:, frames].permute(0, 2, 1, 3).reshape(logits.shape[0], logits.shape[2], -1)
    v = vals[:, frames].permute(0, 2, 1, 3, 4).reshape(lg.shape[0], lg.shape[1], lg.shape[2], -1)
    m = lg.max(-1)[0]
    e = torch.exp(lg - m[..., None])
    s = e.sum(-1)
    row = (e[..., None] * v).sum(2) / s[..., None]
    return row.flatten(1), m, s


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pavenet_amd import dist as pd
        # --- clip-parallel exchange
        B, N, K = 2, 5, 15
        g = torch.Generator().manual_seed(100 + rank)
        res = dict(bboxes=torch.randn(B, N, 5, generator=g), kpts=torch.randn(B, N, K, 3, generator=g),
                   keep=(torch.rand(B, N, generator=g) > 0.5).int())
        allr = pd.all_gather_results(res)
        assert allr.shape == (world, B, N * (5 + 3 * K + 1))
        mine = pd.unpack_results(allr[rank], N, K)
        assert torch.equal(mine['bboxes'], res['bboxes']) and torch.equal(mine['kpts'], res['kpts'])
        assert torch.equal(mine['keep'], res['keep'].bool())
        for r in range(world):  # every rank sees every rank's rows
            gr = torch.Generator().manual_seed(100 + r)
            assert torch.equal(pd.unpack_results(allr[r], N, K)['bboxes'],
                               torch.randn(B, N, 5, generator=gr))
        # --- frame-sharded softmax merge, T = 5 frames over 2 ranks (rank 1 owns frames 1, 3)
        T, U, H, P, D = 5, 7, 8, 16, 32
        gs = torch.Generator().manual_seed(7)  # same data on both ranks
        logits = torch.randn(U, T, H, P, generator=gs) * 4
        logits[0, 1] += 60.0   # one frame dominates: exercises the max rescale
        vals = torch.randn(U, T, H, P, D, generator=gs)
        shard = pd.FrameShard(T, rank, world)
        assert shard.local == [t for t in range(T) if t % world == rank]
        assert shard.center == 2 and shard.center_owner == 0
        row, m, s = _partial(logits, vals, shard.local)
        merged = pd.all_gather_merge(row, m, s)
        full, _, _ = _partial(logits, vals, list(range(T)))
        np.testing.assert_allclose(merged.numpy(), full.numpy(), rtol=1e-5, atol=1e-5)
        # a rank with no frames (world > T) drops out of the merge
        empty_row = torch.zeros(U, H * D)
        rows = torch.stack([full, empty_row])
        mm = torch.stack([m, torch.full_like(m, float('-inf'))])
        ss = torch.stack([s, torch.zeros_like(s)])
        np.testing.assert_allclose(pd.merge_softmax_partials(rows, mm, ss).numpy(), full.numpy(),
                                   rtol=1e-6)
        # broadcast of the centre-frame proposals
        t = torch.full((3,), float(rank))
        pd.broadcast_from(t, shard.center_owner)
        assert t.tolist() == [0.0, 0.0, 0.0]
        open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def _worker8(rank, world, port, tmp):
    """World size 8, the driver's first multi-rank shape: BASELINE configs[4]'s T = 15 over 8 ranks gives
    2, 2, 2, 2, 2, 2, 2, 1 frames per rank with the centre frame (7) on rank 7 -- the one rank that owns a
    single frame; every collective of the frame-sharded forward and the clip-parallel result exchange with
    those shapes."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pavenet_amd import dist as pd
        T, U, H, P, D = 15, 6, 8, 60, 32
        shard = pd.FrameShard(T, rank, world)
        assert shard.local == [t for t in range(T) if t % 8 == rank]
        assert shard.n_local == (1 if rank == 7 else 2)
        assert shard.center == 7 and shard.center_owner == 7
        assert shard.owns_center() == (rank == 7)
        if rank == 7:
            assert shard.local_index_of_center() == 0
        # the proposals of the centre frame: broadcast from the single-frame rank
        sel = torch.arange(300, dtype=torch.float32) * (1.0 if rank == 7 else -1.0)
        pd.broadcast_from(sel, shard.center_owner)
        assert torch.equal(sel, torch.arange(300, dtype=torch.float32))
        # the T-frame attention: every rank's partial row over ITS frames, one all-gather, exact merge
        gs = torch.Generator().manual_seed(11)       # the same logits / values on every rank
        logits = torch.randn(U, T, H, P, generator=gs) * 4
        logits[1, 7] += 70.0                         # the single-frame rank's frame dominates one query
        logits[2, 0] += 70.0
        vals = torch.randn(U, T, H, P, D, generator=gs)
        row, m, s = _partial(logits, vals, shard.local)
        merged = pd.all_gather_merge(row, m, s)
        full, _, _ = _partial(logits, vals, list(range(T)))
        np.testing.assert_allclose(merged.numpy(), full.numpy(), rtol=2e-5, atol=2e-5)
        # five such layers in a row (3 pose-decoder + 2 joint-decoder layers), different unit counts
        for units in (300, 300, 300, 20 * 15, 20 * 15):
            r_ = torch.randn(units, H * D, generator=gs)
            m_ = torch.randn(units, H, generator=gs)
            s_ = torch.rand(units, H, generator=gs) + 0.1
            out = pd.all_gather_merge(r_, m_, s_)      # (identical inputs on every rank: the merge of equals)
            np.testing.assert_allclose(out.numpy(), r_.numpy(), rtol=1e-5, atol=1e-6)
        # clip-parallel result exchange, 4 clips per rank, 20 poses, 15 key points (the bench's shapes)
        B, N, K = 4, 20, 15
        g = torch.Generator().manual_seed(500 + rank)
        res = dict(bboxes=torch.randn(B, N, 5, generator=g), kpts=torch.randn(B, N, K, 3, generator=g),
                   keep=(torch.rand(B, N, generator=g) > 0.5).int())
        allr = pd.all_gather_results(res)
        assert allr.shape == (8, B, N * (5 + 3 * K + 1))
        for r in range(8):
            gr = torch.Generator().manual_seed(500 + r)
            assert torch.equal(pd.unpack_results(allr[r], N, K)['bboxes'], torch.randn(B, N, 5, generator=gr))
        # T = 7 on 8 ranks: refused on every rank before any collective is entered
        with pytest.raises(AssertionError):
            pd.FrameShard(7, rank, world)
        dist.barrier()
        open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_world_size_8_gloo_t15_layout(tmp_path):
    """No rank count other than 2 had executed before the driver's 8-GPU run: the collectives of both shard
    modes at world size 8 (gloo, CPU), with the T = 15 frame layout of BASELINE configs[4]."""
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def test_merge_is_exact_for_any_split():
    from pavenet_amd import dist as pd
    T, U, H, P, D = 7, 4, 8, 15, 32
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(U, T, H, P, generator=g) * 3
    vals = torch.randn(U, T, H, P, D, generator=g)
    full, _, _ = _partial(logits, vals, list(range(T)))
    for G in (2, 3, 4, 7):
        parts = [_partial(logits, vals, pd.FrameShard(T, r, G).local) for r in range(G)]
        merged = pd.merge_softmax_partials(torch.stack([p[0] for p in parts]),
                                           torch.stack([p[1] for p in parts]),
                                           torch.stack([p[2] for p in parts]))
        np.testing.assert_allclose(merged.numpy(), full.numpy(), rtol=1e-5, atol=1e-5)


def test_frame_shard_rejects_more_ranks_than_frames():
    """world > T would leave a rank without a frame (zero-frame memory reshape, peers hanging in
    the collective): refused up front."""
    import pytest
    from pavenet_amd.dist import FrameShard
    with pytest.raises(AssertionError):
        FrameShard(3, 0, 4)
    assert FrameShard(3, 2, 3).local == [2] and FrameShard(15, 3, 4).local == [3, 7, 11]


def test_bench_gpus_n_spawns_its_own_ranks_and_reports_failure():
    """`python bench.py --gpus 2` with no launcher (how the driver runs it) starts the two ranks
    itself as a child `torch.distributed.run` (never an exec).  This host has no GPU, so the ranks
    die in `torch.cuda.set_device`: the parent must relay that as a non-zero return code and say
    so -- not fall back to one rank, not retry."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip('covered on the GPU box by test_bench_multi_rank_code_path_on_one_gpu')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1',
                        '--warmup', '0', '--height', '128', '--width', '160', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert 'the 2-rank child exited with code' in r.stderr
    assert 'nproc-per-node' not in r.stdout and not any(l.startswith('{') for l in r.stdout.splitlines())
