"""How the CPU oracle scales with torch's intra-op thread count on the host (the GPU suite's full-size tests and
bench.py's cpu_baseline leg spend their time in it): one frame through backbone + neck + one encoder pass.
python tests/oracle_threads.py [threads ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (lives under tests/: the oracle is test infrastructure)
import torch  # noqa: E402
from oracle import pavenet_ref as R  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402

m = init_random_weights(build_model(videopose_r50_cfg(num_frames=3, max_per_img=20)), seed=0).eval()
sd = {k: v.detach().float() for k, v in m.state_dict().items()}
cfg = dict(num_frames=3, num_keypoints=15, num_query=300, max_per_img=20)
img = torch.randn(1, 3, 3, 800, 1344, generator=torch.Generator().manual_seed(0))
R.SAMPLER = 'torch'
print('host cpus', os.cpu_count(), 'default threads', torch.get_num_threads(), flush=True)
for n in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
    torch.set_num_threads(n)
    with torch.no_grad():
        t0 = time.time()
        R.videopose_simple_test(sd, cfg, img)
        print(f'threads {n:4d}: T = 3 clip {time.time() - t0:6.1f} s', flush=True)
