"""Where two batch compositions of the same clip start to differ: clip c of a (padded or un-padded) batch against
the same clip alone, BOTH top-k selections pinned to the alone run's, stage by stage -- encoder memory, pose-decoder
states, key-point predictions, refine-decoder states, refined key points, sigmas, the RLE confidence p, the
p^5 / (p^5 + 1e-10) rescale factor, final key points (pixels).
    python tools/debug_batch_vs_alone.py [T=7] [padded=1] [clips=2]"""
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


def run(imgs, mt, sel=None):
    with torch.no_grad():
        feat = m.extract_feat(imgs)
        kw = {} if sel is None else dict(force_topk_proposals=sel[0])
        outs = head(feat, mt, last_level_only=True, **kw)
        taps = {}
        res = head.get_bboxes(outs, mt, force_score_topk=None if sel is None else sel[1], taps=taps)
    tp = head.transformer.last_topk_proposals.clone()
    return dict(memory=outs['memory'].clone(), hs=outs['hs'].clone(), kpt=outs['all_kpt_preds'].clone(),
                cls=outs['all_cls_scores'].clone(), rhs=taps['refine_hs'].clone(), rk=taps['refine_kpts'].clone(),
                rs=taps['refine_sigma'].clone(), kpts=res['kpts'].clone(), keep=res['keep'].clone()), tp, \
        res['score_index'].clone()


def d(a, b):
    return float((a - b).abs().max())


for c in range(B):
    alone, tp1, si1 = run(img[c:c + 1], metas[c:c + 1])
    alone, _, _ = run(img[c:c + 1], metas[c:c + 1], (tp1, si1))
    _, tpb, sib = run(img, metas)
    tpb[c].copy_(tp1[0])
    sib[c].copy_(si1[0])
    batch, _, _ = run(img, metas, (tpb, sib))
    Wc = shapes[c][1]
    print(f'--- clip {c} (valid {shapes[c][:2]}) in a batch of {B} vs alone, selections pinned to the alone run')
    print(f'memory            {d(batch["memory"][:, c * T:(c + 1) * T], alone["memory"]):.3e}   (max |x| {float(alone["memory"].abs().max()):.2f})')
    print(f'pose-decoder hs   {d(batch["hs"][:, c:c + 1], alone["hs"]):.3e}')
    print(f'cls logits        {d(batch["cls"][:, c:c + 1], alone["cls"]):.3e}')
    print(f'kpt preds (px)    {d(batch["kpt"][:, c:c + 1], alone["kpt"]) * Wc:.3e}')
    N = alone['rk'].shape[1]
    rb = batch['rhs'][:, c * N:(c + 1) * N] if batch['rhs'].shape[1] != N else batch['rhs']
    print(f'refine hs         {d(rb, alone["rhs"]):.3e}')
    print(f'refine kpts (px)  {d(batch["rk"][c], alone["rk"][0]) * Wc:.3e}')
    sb, sa = batch['rs'][c].float(), alone['rs'][0].float()
    print(f'refine sigma      {d(sb, sa):.3e}   range [{float(sa.min()):.4f}, {float(sa.max()):.4f}]')
    pa = 0.7 * (1 - torch.exp(-0.2 / sa[..., 0])) * (1 - torch.exp(-0.2 / sa[..., 1]))
    pb = 0.7 * (1 - torch.exp(-0.2 / sb[..., 0])) * (1 - torch.exp(-0.2 / sb[..., 1]))
    fa, fb = pa**5 / (pa**5 + 1e-10), pb**5 / (pb**5 + 1e-10)
    print(f'p                 {d(pa, pb):.3e}   range [{float(pa.min()):.3e}, {float(pa.max()):.3e}]  p^5 min {float((pa**5).min()):.3e}')
    print(f'rescale factor    {d(fa, fb):.3e}   range [{float(fa.min()):.6f}, {float(fa.max()):.6f}]')
    print(f'final kpts (px)   {d(batch["kpts"][c, ..., :2], alone["kpts"][0, ..., :2]):.3e}   '
          f'keep equal {bool(torch.equal(batch["keep"][c], alone["keep"][0]))}  kept {int(alone["keep"].sum())}')
    worst = (batch['kpts'][c, ..., :2] - alone['kpts'][0, ..., :2]).abs().amax(-1)      # [N, K]
    n_, k_ = divmod(int(worst.argmax()), worst.shape[1])
    print(f'  worst pose {n_} joint {k_}: batch {batch["kpts"][c, n_, k_].tolist()} alone {alone["kpts"][0, n_, k_].tolist()}'
          f' sigma {sa[n_, k_].tolist()} refined (norm) batch {batch["rk"][c, n_, k_].tolist()} alone {alone["rk"][0, n_, k_].tolist()}')
