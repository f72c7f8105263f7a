"""hipGraph capture of the whole forward (backbone .. OKS-NMS) for a fixed input shape.

The decoders are launch-bound (300 pose queries, N*K joint queries: hundreds of sub-10-us
kernels per clip); replaying one captured graph removes the per-launch host cost.  Everything
the path needs is capture-safe by construction: the HIP entry points neither allocate nor
synchronise nor read device memory from the host, constant-per-shape tensors are cached on the
device during the eager warm-up, and the only device->host copy (results) happens outside.

Measured (MI355X, R-50, 800x1344, fp32): T = 3, B = 1: 19.96 ms eager -> 18.07 ms replayed.
Validated up to 3-frame batches.  With the 28-frame headline batch (T = 7, B = 4) one capture +
replay run ended in a GPU memory-access fault inside the replay (not reproduced eagerly, where the
same launches run clean, and of no use there: that batch is GPU-bound, 128.0 vs 129.5 ms), so
bench.py keeps graphs opt-in (--graph 1) until that is understood.
"""
import torch


class GraphedForward:
    """``g = GraphedForward(model, example_img, img_metas); res = g(img)``.

    KNOWN LIMIT: at 14+ frames of 800x1344 a replay that follows an explicit
    ``torch.cuda.synchronize()`` faults on this ROCm stack; small batches (T = 3, B = 1) are fine.

    `res` is the head's fixed-shape result dict; its tensors are static buffers that the next
    call overwrites (clone what must survive)."""

    def __init__(self, model, example_img, img_metas, rescale=False, warmup=3, **head_kwargs):
        assert example_img.is_cuda
        from . import ops
        self.model = model
        self.img_metas = img_metas
        self.static_in = example_img.clone()
        saved, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None  # no event records inside a capture
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.no_grad():
                for _ in range(warmup):  # builds every cache, lets MIOpen pick its solvers
                    model.forward_device(self.static_in, img_metas, rescale=rescale, **head_kwargs)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph), torch.no_grad():
                self.static_out = model.forward_device(self.static_in, img_metas, rescale=rescale,
                                                       **head_kwargs)
        finally:
            ops.KERNEL_EVENTS = saved

    def __call__(self, img):
        self.static_in.copy_(img, non_blocking=True)
        self.graph.replay()
        return self.static_out
