"""Runs conv3x3_split on one shape given on the command line and checks it against F.conv2d:
python tools/debug_conv_shapes.py N H W Cin Cout stride planes(1|2|3|16)"""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops
N, H, W, Cin, Cout, s, P = map(int, sys.argv[1:8])
x = torch.randn(N, Cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
w = torch.randn(Cout, Cin, 3, 3, device='cuda') / (3 * Cin ** 0.5)
b = torch.randn(Cout, device='cuda')
wp = ops.split_conv3x3_weight(w, P)
y = ops.conv3x3_split(x, wp, b, stride=s, relu=True, fp16=(P == 16))
torch.cuda.synchronize()
ref = torch.relu(F.conv2d(x, w, b, s, 1))
print('shape', sys.argv[1:8], 'max|d|', float((y - ref).abs().max()), 'ref max', float(ref.abs().max()))
