"""Backward of the fused T-frame deformable-attention launches.

Forward = the fused HIP kernel (joint softmax over T*L*P logits, location arithmetic, bilinear
gathers of all frames, cross-frame fusion in one launch).  Backward re-expresses the same function
as the reference's un-fused chain -- softmax / location arithmetic in PyTorch, one
``MultiScaleDeformableAttnFunction`` per frame (MO:1466-1578, OT:1722-1858) -- under autograd, so
the gathers' gradients come from the hand-written col2im kernel (``pave_ms_deform_attn_backward``,
ms_deform_attn_cuda_kernel.cuh:256-801) and the softmax / offset / reference-point gradients from
autograd.  The joint softmax is algebraically the reference's per-frame softmax x Z_t / sum Z
re-weighting, so these are the gradients of the reference formulation
(tests/test_ops_gpu.py::test_fused_backward_vs_autograd_of_reference_formulation).
"""
import torch

from . import ops


def _norm_wh(shapes, dtype):
    return torch.stack([shapes[..., 1], shapes[..., 0]], -1).to(dtype)


def _joint_softmax(lg):
    """lg [U, T, M, LP] -> softmax over (T, LP) jointly per (U, M), same layout."""
    U, T, M, LP = lg.shape
    return lg.permute(0, 2, 1, 3).reshape(U, M, T * LP).softmax(-1).view(U, M, T, LP).permute(0, 2, 1, 3)


def grid_unfused(value, shapes, lsi, proj, ref, T, n_clips, units_per_clip, unit_clip):
    """value [n_clips*T, S, 8, 32]; proj [U, >= T*8*16*3]; ref [T, U, 4, 2] -> [U, 256]."""
    U = proj.shape[0]
    M, L, P = 8, shapes.shape[0], 4
    n_off = T * M * L * P * 2
    off = proj[:, :n_off].reshape(U, T, M, L, P, 2)
    aw = _joint_softmax(proj[:, n_off:n_off + n_off // 2].reshape(U, T, M, L * P))
    norm = _norm_wh(shapes, proj.dtype)
    if unit_clip is None:
        groups = [(c, slice(c * units_per_clip, (c + 1) * units_per_clip)) for c in range(n_clips)]
    else:
        uc = unit_clip.long()
        groups = [(c, torch.nonzero(uc == c).flatten()) for c in range(n_clips)]
    parts = []
    for c, idx in groups:
        n = (idx.stop - idx.start) if isinstance(idx, slice) else idx.numel()
        if n == 0:
            continue
        acc = 0
        for t in range(T):
            loc = ref[t][idx][:, None, :, None, :] + off[idx, t] / norm[None, None, :, None, :]
            o = ops.MultiScaleDeformableAttnFunction.apply(
                value[c * T + t][None], shapes, lsi, loc[None].contiguous(),
                aw[idx, t].reshape(1, n, M, L, P).contiguous(), 64)
            acc = acc + o[0]
        parts.append((idx, acc))
    if unit_clip is None:
        return torch.cat([a for _, a in parts], 0)        # clip-major units, in order
    out = proj.new_zeros((U, M * value.shape[-1]))
    for idx, a in parts:
        out = out.index_copy(0, idx, a)
    return out


def pose_unfused(value, shapes, lsi, proj, ref, T, n_clips, Q, K):
    """value [n_clips*T, S, 8, 32]; proj [n_clips*Q, >= T*8*L*K*3]; ref [n_clips, T*Q, L, 2K]."""
    M, L = 8, shapes.shape[0]
    n_off = T * M * L * K * 2
    off = proj[:, :n_off].reshape(n_clips, Q, T, M, L, K, 2)
    aw = _joint_softmax(proj[:, n_off:n_off + n_off // 2].reshape(n_clips * Q, T, M, L * K))
    aw = aw.reshape(n_clips, Q, T, M, L, K)
    ref = ref.reshape(n_clips, T * Q, L, 2 * K)
    acc = 0
    for t in range(T):
        rp_t = ref[:, t * Q:(t + 1) * Q]
        rp = rp_t.reshape(n_clips, Q, L, K, 2).unsqueeze(2)
        xs, ys = rp_t[..., 0::2], rp_t[..., 1::2]
        wh = torch.cat([torch.clamp(xs.max(-1, keepdim=True)[0] - xs.min(-1, keepdim=True)[0], min=1e-4),
                        torch.clamp(ys.max(-1, keepdim=True)[0] - ys.min(-1, keepdim=True)[0], min=1e-4)],
                       -1)[:, :, None, :, None, :]
        loc = rp + off[:, :, t] * wh * 0.5
        acc = acc + ops.MultiScaleDeformableAttnFunction.apply(
            value[t::T].contiguous(), shapes, lsi, loc.contiguous(), aw[:, :, t].contiguous(), 64)
    return acc.reshape(n_clips * Q, -1)


def _grads(fn, tensors, needs, gout):
    leaves = [t.detach().requires_grad_(n) for t, n in zip(tensors, needs)]
    with torch.enable_grad():
        out = fn(*leaves)
    wanted = [l for l, n in zip(leaves, needs) if n]
    got = iter(torch.autograd.grad(out, wanted, gout.contiguous()))
    return [next(got) if n else None for n in needs]


class GridFusedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, shapes, lsi, proj, ref, T, n_clips, units_per_clip, unit_clip, order):
        ctx.save_for_backward(value, shapes, lsi, proj, ref, unit_clip)
        ctx.cfg = (T, n_clips, units_per_clip)
        return ops.deform_attn_grid_fused(value, shapes, lsi, proj, ref, T=T, n_clips=n_clips,
                                          units_per_clip=units_per_clip, unit_clip=unit_clip,
                                          order=order)

    @staticmethod
    def backward(ctx, gout):
        value, shapes, lsi, proj, ref, unit_clip = ctx.saved_tensors
        T, n_clips, upc = ctx.cfg
        gv, gp, gr = _grads(
            lambda v, p, r: grid_unfused(v, shapes, lsi, p, r, T, n_clips, upc, unit_clip),
            (value, proj, ref), ctx.needs_input_grad[0:1] + ctx.needs_input_grad[3:5], gout)
        return gv, None, None, gp, gr, None, None, None, None, None


class PoseFusedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, shapes, lsi, proj, ref, T, n_clips, Q, K):
        ctx.save_for_backward(value, shapes, lsi, proj, ref)
        ctx.cfg = (T, n_clips, Q, K)
        return ops.deform_attn_pose_fused(value, shapes, lsi, proj, ref, T=T, n_clips=n_clips,
                                          num_query=Q, num_keypoints=K)

    @staticmethod
    def backward(ctx, gout):
        value, shapes, lsi, proj, ref = ctx.saved_tensors
        T, n_clips, Q, K = ctx.cfg
        gv, gp, gr = _grads(
            lambda v, p, r: pose_unfused(v, shapes, lsi, p, r, T, n_clips, Q, K),
            (value, proj, ref), ctx.needs_input_grad[0:1] + ctx.needs_input_grad[3:5], gout)
        return gv, None, None, gp, gr, None, None, None, None


class EncTileFunction(torch.autograd.Function):
    """The LDS-tile encoder kernel (T = 1): same function as the grid form, frame-major units."""

    @staticmethod
    def forward(ctx, value, proj, ref, levels_hw, variant):
        ctx.save_for_backward(value, proj, ref)
        ctx.levels_hw = tuple((int(h), int(w)) for h, w in levels_hw)
        return ops.deform_attn_enc_tile(value, proj, ref, levels_hw=levels_hw, variant=variant)

    @staticmethod
    def backward(ctx, gout):
        value, proj, ref = ctx.saved_tensors
        F_, S = value.shape[:2]
        shapes = torch.as_tensor(ctx.levels_hw, dtype=torch.long, device=value.device)
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        gv, gp, gr = _grads(
            lambda v, p, r: grid_unfused(v, shapes, lsi, p, r.reshape(1, F_ * S, -1, 2), 1, F_, S,
                                         None),
            (value, proj, ref), ctx.needs_input_grad[0:3], gout)
        return gv, gp, gr, None, None
