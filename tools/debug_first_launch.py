"""Host-side timeline between the end of one bench step (results on the host) and the first kernel
launch of the next: where the GPU's ~0.8 ms of idle time at a step boundary goes.
    python tools/debug_first_launch.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
from pavenet_amd import native, ops, tuning  # noqa: E402
from pavenet_amd.bricks import set_gemm_mode  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402


def main():
    T, B, H, W = 7, 4, 800, 1344
    m = build_model(videopose_r50_cfg(num_frames=T, max_per_img=20))
    init_random_weights(m, seed=0)
    m = m.cuda().eval()
    set_gemm_mode('bf16x3')
    tuning.use_tuned_gemms()
    img = torch.randn(B, T, 3, H, W, device='cuda')
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=(1., 1., 1., 1.))] * B
    marks = {}
    lib = native.load()
    orig_wrapper = ops.conv7x7s2_nchw_split
    orig_entry = lib.pave_conv7x7s2_nchw_split_f32

    def wrapper(*a, **k):
        marks['wrapper'] = time.perf_counter()
        return orig_wrapper(*a, **k)

    class Entry:
        def __call__(self, *a):
            marks['entry'] = time.perf_counter()
            r = orig_entry(*a)
            marks['launched'] = time.perf_counter()
            return r
    ops.conv7x7s2_nchw_split = wrapper
    lib.pave_conv7x7s2_nchw_split_f32 = Entry()
    buf = None
    t_done = None
    for it in range(6):
        t_call = time.perf_counter()
        res = m.forward_device(img, metas)
        packed = torch.cat([res['bboxes'].flatten(1), res['kpts'].flatten(1), res['keep'].float()], dim=1)
        if buf is None:
            buf = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
        buf.copy_(packed, non_blocking=True)
        t_enq = time.perf_counter()
        torch.cuda.current_stream().synchronize()
        t_sync = time.perf_counter()
        if t_done is not None and it >= 3:
            print(f'step {it}: results on host -> forward_device called {1e3 * (t_call - t_done):.3f} ms, '
                  f'-> stem wrapper {1e3 * (marks["wrapper"] - t_done):.3f}, -> C entry '
                  f'{1e3 * (marks["entry"] - t_done):.3f}, -> launch returned {1e3 * (marks["launched"] - t_done):.3f}; '
                  f'whole enqueue {1e3 * (t_enq - t_call):.2f} ms, step {1e3 * (t_sync - t_call):.2f} ms')
        t_done = t_sync


if __name__ == '__main__':
    main()
