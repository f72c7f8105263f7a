"""Whole-video inference with the per-frame encoder-memory cache at full size:
every frame of an N-frame 800x1344 video gets its T-frame window result; frames/s against
running simple_test on every window.   python tools/bench_streaming.py [n_frames=28] [T=7] [gemm=bf16x3] [decode_chunk=14]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd.bricks import set_gemm_mode  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.streaming import VideoPoseStream  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    gemm = sys.argv[3] if len(sys.argv) > 3 else 'bf16x3'
    set_gemm_mode(gemm)
    torch.backends.cudnn.benchmark = True
    m = init_random_weights(build_model(videopose_r50_cfg(num_frames=T, max_per_img=20)), seed=0)
    m = m.cuda().eval()
    meta = dict(batch_input_shape=(800, 1344), img_shape=(800, 1344, 3), scale_factor=(1., 1., 1., 1.))
    video = torch.randn(n, 3, 800, 1344, device='cuda')
    dchunk = int(sys.argv[4]) if len(sys.argv) > 4 else 14
    stream = VideoPoseStream(m, meta, encode_chunk=14, decode_chunk=dchunk)
    for _ in range(2):
        out = stream.infer_video(video)
    torch.cuda.synchronize()
    t0 = time.time()
    reps = 3
    for _ in range(reps):
        out = stream.infer_video(video)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    print(f'[--gemm {gemm}] streaming: {n} frames (= {n} T={T} windows) in {dt * 1e3:.1f} ms -> {n / dt:.1f} windows/s')
    wins = stream.window_indices(n, T)
    clips = torch.stack([video[w] for w in wins[:4]], 0)
    for _ in range(2):
        m.forward_device(clips, [meta] * 4)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        m.forward_device(clips, [meta] * 4)
    torch.cuda.synchronize()
    dc = (time.time() - t0) / reps
    print(f'per-window simple_test: 4 windows in {dc * 1e3:.1f} ms -> {4 / dc:.1f} windows/s; '
          f'streaming speed-up {n / dt / (4 / dc):.2f}x')


if __name__ == '__main__':
    main()
