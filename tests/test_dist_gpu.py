"""RCCL executed on the one-GPU box: a world-size-1 "nccl" process group running, on DEVICE
tensors, the very collectives the N > 1 paths make (pavenet_amd/dist.py, bench.py).  Multi-rank
semantics are covered on gloo (tests/test_dist_cpu.py, the two-process GPU tests); what only RCCL
can show -- communicator creation on this image, device all-gather / broadcast / all-reduce
enqueued behind this package's kernels, clean teardown -- is covered here.  No scaling number
comes out of this; DESIGN.md section 6 says so."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port))
    # the host driver of this pool only supports dmabuf IPC; RCCL needs it for any peer mapping
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return env


def test_nccl_world_size_1_collectives_on_device_tensors():
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'nccl_worker.py')],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, (r.stdout[-1500:] + '\n---\n' + r.stderr[-6000:])
    assert 'nccl world-size-1 collectives on device tensors: ok' in r.stdout


@pytest.mark.parametrize('shard', ['clips', 'frames'])
def test_bench_world_branches_on_nccl_with_one_rank(shard):
    """`bench.py --gpus 1 --force-dist nccl`: the `world > 1` branches of the bench (rank census,
    device all-gather of the packed results / frame-sharded forward with its all-gather merges,
    max-over-ranks all-reduce, barrier, teardown) with world = 1 on RCCL."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', 'nccl',
           '--steps', '2', '--warmup', '1', '--height', '128', '--width', '160', '--shard', shard,
           '--frames', '5' if shard == 'frames' else '3', '--no-cpu-baseline', '--no-native-side',
           '--no-secondary', '--gemm-select', 'default']
    env = _env()
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k)      # bench.py sets up its own one-rank rendezvous
    torch.cuda.empty_cache()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:] + '\n---\n' + r.stderr[-4000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['backend'] == 'nccl'
    assert line['ranks_seen'] == [0] and len(line['devices']) == 1
