"""Expected values of the fused T-frame ops, composed from the ORACLE's un-fused arithmetic
(per-frame softmax, Z_t re-weighting, one sampler call per frame), i.e. what
MO:1484-1578 / OT:1737-1858 compute between the Linears.  Test helper (CPU)."""
import torch

from oracle import pavenet_ref as R


def grid_expected(value, shapes, lsi, proj, ref, T, unit_clip):
    """value [n_clips*T, S, 8, 32]; proj [U, T*8*16*3]; ref [T, U, 4, 2]; unit_clip [U]."""
    U = proj.shape[0]
    M, L, P = 8, 4, 4
    off = proj[:, :T * M * L * P * 2].view(U, T, M, L, P, 2)
    lg = proj[:, T * M * L * P * 2:T * M * L * P * 3].view(U, T, M, L * P)
    norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1).to(proj.dtype)
    outs, zs = [], []
    for t in range(T):
        z = torch.exp(lg[:, t]).sum(-1, keepdim=True)
        aw = lg[:, t].softmax(-1).view(U, 1, M, L, P)
        loc = ref[t][:, None, None, :, None, :] + off[:, t][:, None] / norm[None, None, None, :, None, :]
        v = value[unit_clip * T + t]  # [U, S, 8, 32] one slab per unit
        outs.append(R.msda(v, shapes, lsi, loc, aw).view(U, M, -1))
        zs.append(z)
    z_all = sum(zs)
    return sum(o * (z / z_all) for o, z in zip(outs, zs)).flatten(-2, -1)


def pose_expected(value, shapes, lsi, proj, ref, T, n_clips, Q, K):
    """value [n_clips*T, S, 8, 32]; proj [n_clips*Q, T*8*L*K*3]; ref [n_clips, T*Q, L, 2K]."""
    M, L = 8, shapes.shape[0]
    U = n_clips * Q
    off = proj[:, :T * M * L * K * 2].view(n_clips, Q, T, M, L, K, 2)
    lg = proj[:, T * M * L * K * 2:T * M * L * K * 3].view(n_clips, Q, T, M, L * K)
    outs, zs = [], []
    for t in range(T):
        z = torch.exp(lg[:, :, t]).sum(-1, keepdim=True)
        aw = lg[:, :, t].softmax(-1).view(n_clips, Q, M, L, K)
        rp_t = ref[:, t * Q:(t + 1) * Q]
        rp = rp_t.reshape(n_clips, Q, L, -1, 2).unsqueeze(2)
        x1 = rp_t[..., 0::2].min(-1, keepdim=True)[0]
        y1 = rp_t[..., 1::2].min(-1, keepdim=True)[0]
        x2 = rp_t[..., 0::2].max(-1, keepdim=True)[0]
        y2 = rp_t[..., 1::2].max(-1, keepdim=True)[0]
        wh = torch.cat([torch.clamp(x2 - x1, min=1e-4), torch.clamp(y2 - y1, min=1e-4)],
                       -1)[:, :, None, :, None, :]
        loc = rp + off[:, :, t] * wh * 0.5
        v = value[t::T]
        outs.append(R.msda(v, shapes, lsi, loc, aw).view(n_clips, Q, M, -1))
        zs.append(z)
    z_all = sum(zs)
    return sum(o * (z / z_all) for o, z in zip(outs, zs)).flatten(-2, -1).view(U, -1)
