import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from pavenet_amd import ops
torch.backends.cudnn.benchmark = True
def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
n = 28
shapes = [(n, 200, 336, 64, 64, 1), (n, 200, 336, 128, 128, 2), (n, 100, 168, 128, 128, 1),
          (n, 50, 84, 256, 256, 1), (n, 25, 42, 512, 512, 1)]
for (N, H, W, Cin, Cout, s) in shapes:
    x = torch.randn(N, Cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device='cuda') / (3 * Cin**0.5)).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device='cuda')
    t0 = timeit(lambda: F.conv2d(x, w, None, s, 1))
    t1 = timeit(lambda: ops.bias_act_rows_(F.conv2d(x, w, None, s, 1), b, None, relu=True))
    try:
        t2 = timeit(lambda: torch.miopen_convolution_relu(x, w, b, [s, s], [1, 1], [1, 1], 1))
        y2 = torch.miopen_convolution_relu(x, w, b, [s, s], [1, 1], [1, 1], 1)
        err = (y2 - torch.relu(F.conv2d(x, w, b, s, 1))).abs().max().item()
        cl = y2.is_contiguous(memory_format=torch.channels_last)
    except Exception as ex:
        t2, err, cl = float('nan'), str(ex)[:80], None
    print(f'{H}x{W} {Cin}->{Cout} s{s}: conv {t0:.3f}  conv+pass {t1:.3f}  miopen_conv_relu {t2:.3f} err {err} cl={cl}')
# fp16-in / fp32-out GEMM probe
M = 625044
a = torch.randn(M, 256, device='cuda'); w1 = torch.randn(1024, 256, device='cuda') * 0.05
a16, w16 = a.half(), w1.half()
try:
    y = torch.mm(a16, w16.t(), out_dtype=torch.float32)
    t = timeit(lambda: torch.mm(a16, w16.t(), out_dtype=torch.float32))
    print('mm fp16->fp32 out_dtype OK', y.dtype, f'{t:.3f} ms', (y - a @ w1.t()).abs().max().item())
except Exception as ex:
    print('mm out_dtype failed:', str(ex)[:200])
t = timeit(lambda: torch.mm(a16, w16.t()))
print(f'mm fp16->fp16 {t:.3f} ms')
t = timeit(lambda: a.half())
print(f'cast fp32->fp16 [M,256] {t:.3f} ms')
ab, wb = a.bfloat16(), w1.bfloat16()
t = timeit(lambda: torch.mm(ab, wb.t()))
print(f'mm bf16->bf16 {t:.3f} ms')
