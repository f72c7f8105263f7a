"""Random-weight initialisation for benchmarks and demos (no checkpoints are published for
PAVE-Net, README.md:14-15 of the reference)."""
import torch


@torch.no_grad()
def init_random_weights(model, seed=0):
    """Reference-equivalent ``init_weights`` under a fixed seed, then re-randomise the
    sampling-offset / attention-logit Linears: their default init is
    all-zero weights (OT:1631-1642), which makes every query sample the same points."""
    torch.manual_seed(seed)
    # PyTorch draws Linear / Conv / Embedding defaults from the global generator at CONSTRUCTION time,
    # and init_weights leaves many of them (every bias of the Linears, the attention in-projections) as
    # they are: re-draw them under the seed first, so that the same seed gives the same weights in
    # every process (bench parity, detection counts and the full-size tests are then reproducible)
    for mod in model.modules():
        reset = getattr(mod, 'reset_parameters', None) or getattr(mod, '_reset_parameters', None)
        if callable(reset) and not list(mod.children()) or isinstance(mod, torch.nn.MultiheadAttention):
            if callable(reset):
                reset()
    # NOT re-applied on purpose: heads.Linear_with_norm's constructor-time xavier_uniform(gain = 0.01) on the last
    # layer of the sigma / RLE branches (HEAD:1611; `init_weights` never redoes it, so the reset above leaves the
    # default Linear init there -- advisor finding, round 4).  Tried in round 5 (`constructor_init()` after the
    # reset): every sigma then sits at ~0.5, the twenty best poses of a random-weight clip become near copies of
    # each other and OKS-NMS keeps ONE (bench parity: oracle_poses 1) -- the full-size parity tests need >= 5
    # surviving poses to mean anything.  The recipe of rounds 1-4 stays; what it costs: the RLE rescale
    # kpt p^5 / (p^5 + 1e-10) is ill-conditioned for some poses (p^5 ~ 1e-10), so rounding-level differences
    # between two batch compositions show as up to ~0.4 px in FINAL key points while the decoder states agree to
    # 1e-4 (tests compare across batch compositions on the decoder states, against the oracle on one composition).
    for m in (model.backbone, model.neck, model.bbox_head):
        if m is not None:
            m.init_weights()
    g = torch.Generator().manual_seed(seed + 1)
    for name, p in model.named_parameters():
        if 'sampling_offsets' in name or 'attention_weights' in name:
            std = 0.05 if name.endswith('weight') else 0.5
            p.copy_(torch.randn(p.shape, generator=g) * std)
        elif 'cls_branches' in name and name.endswith('weight'):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)  # spread the proposal scores
        elif ('kpt_branches' in name) and name.endswith('.6.weight'):
            p.copy_(torch.randn(p.shape, generator=g) * 0.01)
    # BN running stats as a trained net would have (non-trivial but well conditioned)
    for name, b in model.named_buffers():
        if name.endswith('running_var'):
            b.copy_(1.0 + 0.1 * torch.rand(b.shape, generator=g))
        elif name.endswith('running_mean'):
            b.copy_(0.1 * torch.randn(b.shape, generator=g))
    for m in model.modules():
        if m.__class__.__name__ == 'Bottleneck':
            m.bn3.weight.fill_(0.3)
    return model
