import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd.backbones import ResNet
from pavenet_amd import ops
torch.manual_seed(0)
net = ResNet(depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch')
net.init_weights()
for m in net.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
net = net.cuda().eval()
x = torch.randn(4, 3, 128, 160, device='cuda')
with torch.no_grad():
    a = net(x); b = net(x)
    print('determinism', [float((p - q).abs().max()) for p, q in zip(a, b)])
    c = net(x[:3].contiguous())
    print('batch 3 vs 4', [float((p[:3] - q).abs().max()) for p, q in zip(a, c)])
    net.fused_tail_max_k = 0; net._folded = None
    d = net(x)
    print('fused tail vs lib', [float((p - q).abs().max() / q.abs().max()) for p, q in zip(a, d)])
    net.channels_last = False; net._folded = None
    e = net(x)
    print('lib vs plain conv path', [float((p - q).abs().max() / q.abs().max()) for p, q in zip(d, e)])
    print('fused vs plain conv path', [float((p - q).abs().max() / q.abs().max()) for p, q in zip(a, e)])
    for k, cl in ((0, True), (256, True), (0, False)):
        net.fused_tail_max_k = k; net.channels_last = cl; net._folded = None
        outs = [net(x) for _ in range(4)]
        print('det k=%d cl=%s' % (k, cl), [float(max((outs[0][i] - o[i]).abs().max() for o in outs[1:])) for i in range(3)],
              [float(outs[0][i].abs().max()) for i in range(3)])
    M, K, N = 3840, 64, 256
    a = torch.randn(M, K, device='cuda'); w = torch.randn(K, N, device='cuda'); b = torch.randn(N, device='cuda'); ab = torch.randn(K, device='cuda')
    r = torch.randn(M, N, device='cuda')
    ref = None
    for i in range(6):
        idt = r.clone()
        o = ops.rows_gemm_bias_res_act(a, w, b, idt, relu=True, out=idt, a_bias=ab)
        if ref is None: ref = o.clone()
        else: print('rows_gemm rerun diff', float((o - ref).abs().max()))
