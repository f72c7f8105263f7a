"""Can the decoders' value-projection GEMMs run under the launch-bound decoder stages?  A stream
restricted to a subset of the CUs (hipExtStreamCreateWithCUMask) for the big GEMM, the default stream
for a chain of small launches; each alone and both together.   python tools/cu_mask_probe.py [free_every=8]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops  # noqa: E402

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(free_every, n_cu=256):
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(n_cu):
        if free_every == 0 or i % free_every != 0:
            mask[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value), sum(bin(w).count('1') for w in mask)


def main():
    free_every = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = 'cuda'
    torch.zeros(1, device=dev)
    M = 625044
    a = torch.randn(M, 256, device=dev)
    wv = ops.split_weight_bf16x3(torch.randn(512, 256, device=dev) * 0.05)
    x = torch.randn(1200, 256, device=dev)
    ws = ops.split_weight_bf16x3(torch.randn(256, 256, device=dev) * 0.05)
    big = lambda: ops.gemm_bf16x3_ex(a, wv, None, None, n_split=256)    # noqa: E731

    def small_chain(n=150):
        y = x
        for _ in range(n):
            y = ops.gemm_bf16x3(y, ws)
        return y

    def wall(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    print(f'3 big GEMMs, default stream:            {wall(lambda: [big() for _ in range(3)]):7.3f} ms')
    print(f'150 small launches, default stream:     {wall(small_chain):7.3f} ms')
    for fe in (0, free_every, 4):
        side, n_on = masked_stream(fe)
        main_s = torch.cuda.current_stream()

        def big_side():
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                for _ in range(3):
                    big()

        def both():
            big_side()
            small_chain()
            main_s.wait_stream(side)

        def big_only():
            big_side()
            main_s.wait_stream(side)
        print(f'side stream with {n_on} CUs: 3 big GEMMs alone {wall(big_only):7.3f} ms, '
              f'with the 150 small launches on the default stream {wall(both):7.3f} ms')


if __name__ == '__main__':
    main()
