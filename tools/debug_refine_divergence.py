"""Inside the refine (joint) decoder: clip c of a batch against the same clip alone, both top-k selections pinned
to the alone run's -- every sub-module output of every refine layer (self-attention, T-frame cross-attention, FFN),
the hoisted projected values and the reference points going in.
    python tools/debug_refine_divergence.py [T=7] [padded=1] [clips=2]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
import bench  # noqa: E402
from pavenet_amd import bricks  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 7
padded = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) == 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
H, W = 800, 1344
m = init_random_weights(build_model(videopose_r50_cfg(num_frames=T, max_per_img=20)), seed=0).cuda().eval()
bricks.set_gemm_mode('bf16x3')


class A:
    height, width = H, W


img = torch.randn(B, T, 3, H, W, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1234))
img[0].copy_(bench.clip0_image(A, T)[0])
sizes = [(800, 1333), (750, 1333)] if padded else [(H, W)]
shapes = [sizes[i % len(sizes)] + (3,) for i in range(B)]
metas = [dict(batch_input_shape=(H, W), img_shape=s, scale_factor=(1., 1., 1., 1.)) for s in shapes]
head = m.bbox_head
rd = head.transformer.refine_decoder
taps = {}


def hook(name):
    def f(mod, args, kwargs, out):
        taps[name] = out.detach().clone()
        if name.endswith('cross'):
            taps[name + '.ref'] = kwargs['reference_points'].detach().clone()
            vp = kwargs.get('value_projected')
            if vp is not None:
                taps[name + '.value'] = vp.detach().clone()
    return f


for li, layer in enumerate(rd.layers):
    layer.attentions[0].register_forward_hook(hook(f'L{li}.self'), with_kwargs=True)
    layer.attentions[1].register_forward_hook(hook(f'L{li}.cross'), with_kwargs=True)
    layer.ffns[0].register_forward_hook(hook(f'L{li}.ffn'), with_kwargs=True)


def run(imgs, mt, sel=None):
    taps.clear()
    with torch.no_grad():
        feat = m.extract_feat(imgs)
        kw = {} if sel is None else dict(force_topk_proposals=sel[0])
        outs = head(feat, mt, last_level_only=True, **kw)
        res = head.get_bboxes(outs, mt, force_score_topk=None if sel is None else sel[1])
    tp = head.transformer.last_topk_proposals.clone()
    return dict(taps), tp, res['score_index'].clone(), res['kpts'].clone()


def d(a, b):
    return float((a - b).abs().max())


N = 20
for c in range(B):
    _, tp1, si1, _ = run(img[c:c + 1], metas[c:c + 1])
    alone, _, _, ka = run(img[c:c + 1], metas[c:c + 1], (tp1, si1))
    _, tpb, sib, _ = run(img, metas)
    tpb[c].copy_(tp1[0])
    sib[c].copy_(si1[0])
    batch, _, _, kb = run(img, metas, (tpb, sib))
    print(f'--- clip {c} (valid {shapes[c][:2]}) in a batch of {B} vs alone; final kpts {d(kb[c, ..., :2], ka[0, ..., :2]):.3e} px')
    for name in sorted(alone):
        a, b = alone[name], batch[name]
        if name.endswith('.value'):            # [B*T, S, 8, 32]
            b = b[c * T:(c + 1) * T]
        elif name.endswith('.ref'):            # [T*Ntot, K, L, 2] frame-major -> the clip's poses of every frame
            b = b.view(T, B * N, *b.shape[1:])[:, c * N:(c + 1) * N].reshape(a.shape)
        else:                                   # [K, Ntot, C] sequence first
            b = b[:, c * N:(c + 1) * N]
        print(f'  {name:14s} {tuple(a.shape)}  max|d| {d(a, b):.3e}   max|a| {float(a.abs().max()):.3f}')
