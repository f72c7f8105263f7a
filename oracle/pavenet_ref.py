"""oracle/pavenet_ref.py -- TEST INFRASTRUCTURE (the checker, never the product).

CPU restatement (PyTorch fp32/fp64 on CPU tensors + the plain-C sampler in
msda_ref.c) of the reference's PAVE-Net forward path, written functionally over
a state dict that uses the reference's own parameter names.  Each function cites
the reference lines it follows (paths relative to zgspose/PAVENet;
OT = opera/models/utils/transformer.py,
MO = third_party/mmcv/mmcv/ops/multi_scale_deform_attn.py,
HEAD = opera/models/dense_heads/videopose_head_mul_frames.py,
MT = third_party/mmdetection/mmdet/models/utils/transformer.py,
BT = third_party/mmcv/mmcv/cnn/bricks/transformer.py).

It deliberately keeps the reference's *un-fused* arithmetic (per-frame softmax,
Z_t = sum exp(logit) re-weighting with an un-stabilised exp, T separate sampler
calls, memory replicated per pose) so that comparing the fused HIP path against
it is a real test.  The T-frame modules are generalised from the reference's
hard-coded T = 3 / T = 5 to any odd T with the same per-frame pattern; parameter
prefix of frame offset k from the centre: 'pre_' * (-k) for k < 0, '' for k = 0,
'next_' * k for k > 0 (this reproduces pre_pre_/pre_/''/next_/next_next_).

Pinned against golden vectors generated from the real reference
(oracle/gen_golden.py -> tests/golden/*.npz): see tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def _clib():
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, '_build', 'libmsda_oracle.so')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} missing: run `make -C oracle`')
        _CLIB = ctypes.CDLL(path)
    return _CLIB


def frame_prefixes(T):
    """Parameter-name prefix per frame (OT:1607-1624, MO:1337-1354 for T=3)."""
    assert T % 2 == 1
    c = T // 2
    return ['pre_' * (c - t) if t < c else 'next_' * (t - c) for t in range(T)]


# --------------------------------------------------------------------------
# a1: the sampler
# --------------------------------------------------------------------------
def msda_forward_c(value, shapes, lsi, loc, attw):
    """ms_deform_attn_cuda_kernel.cuh:200-254 via oracle/msda_ref.c (numpy in/out)."""
    value = np.ascontiguousarray(value)
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attw = np.ascontiguousarray(attw, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    bs, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.empty((bs, Lq, M * D), dtype=dt)
    fn = (_clib().oracle_msda_forward_f32 if dt == np.float32
          else _clib().oracle_msda_forward_f64)
    fn.restype = None
    fn(value.ctypes.data_as(ctypes.c_void_p), shapes.ctypes.data_as(ctypes.c_void_p),
       lsi.ctypes.data_as(ctypes.c_void_p), loc.ctypes.data_as(ctypes.c_void_p),
       attw.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
       ctypes.c_int(bs), ctypes.c_int(S), ctypes.c_int(M), ctypes.c_int(D),
       ctypes.c_int(L), ctypes.c_int(Lq), ctypes.c_int(P))
    return out


def msda_forward_torch(value, shapes, loc, attw):
    """MO:92-149 (grid_sample formulation), restated."""
    bs, _, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    hw = [(int(h), int(w)) for h, w in shapes]
    parts = value.split([h * w for h, w in hw], dim=1)
    grids = 2 * loc - 1
    sampled = []
    for lvl, (h, w) in enumerate(hw):
        v = parts[lvl].flatten(2).transpose(1, 2).reshape(bs * M, D, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros',
                                     align_corners=False))
    a = attw.transpose(1, 2).reshape(bs * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(bs, M * D, Lq)
    return out.transpose(1, 2).contiguous()


SAMPLER = 'c'  # 'torch' = the reference's own CPU formulation (multi-threaded grid_sample)


def msda(value, shapes, lsi, loc, attw):
    """Sampler used inside the restated modules: the C restatement (or, for timing the CPU
    baseline, the reference's grid_sample formulation MO:92-149)."""
    if SAMPLER == 'torch':
        return msda_forward_torch(value, shapes, loc, attw)
    out = msda_forward_c(value.detach().cpu().numpy(), shapes.cpu().numpy(),
                         lsi.cpu().numpy(), loc.detach().cpu().numpy(),
                         attw.detach().cpu().numpy())
    return torch.from_numpy(out)


# --------------------------------------------------------------------------
# small helpers over the state dict
# --------------------------------------------------------------------------
def linear(sd, pre, x):
    return F.linear(x, sd[pre + '.weight'], sd.get(pre + '.bias'))


def layer_norm(sd, pre, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + '.weight'], sd[pre + '.bias'], 1e-5)


def inverse_sigmoid(x, eps=1e-5):
    """MT:390-406."""
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


def mlp(sd, pre, x, idxs, relu=True):
    """nn.Sequential of Linear(+ReLU) with Linear modules at indices `idxs`."""
    for n, i in enumerate(idxs):
        x = linear(sd, f'{pre}.{i}', x)
        if relu and n < len(idxs) - 1:
            x = F.relu(x)
    return x


def kpt_branch(sd, pre, x):
    """HEAD:172-180: Linear-ReLU x3 + Linear."""
    return mlp(sd, pre, x, (0, 2, 4, 6))


def sigma_branch(sd, pre, x):
    """HEAD:182-189 + Linear_with_norm(norm=False) HEAD:1605-1622: three affine maps."""
    x = linear(sd, pre + '.0', x)
    x = linear(sd, pre + '.1', x)
    return x.matmul(sd[pre + '.2.linear.weight'].t()) + sd[pre + '.2.linear.bias']


def refine_kpt_branch(sd, pre, x):
    """HEAD:249-254: Linear-ReLU x2 + Linear(256->2)."""
    return mlp(sd, pre, x, (0, 2, 4))


# --------------------------------------------------------------------------
# a10 / a3: MultiheadAttention wrapper, FFN
# --------------------------------------------------------------------------
def mha(sd, pre, query, query_pos, num_heads=8):
    """BT:461-551 with key = value-source = query, key_pos = query_pos, + identity."""
    q = query + query_pos
    out = F.multi_head_attention_forward(
        q, q, query, query.shape[-1], num_heads,
        sd[pre + '.attn.in_proj_weight'], sd[pre + '.attn.in_proj_bias'],
        None, None, False, 0.0, sd[pre + '.attn.out_proj.weight'],
        sd[pre + '.attn.out_proj.bias'], training=False, need_weights=False)[0]
    return query + out


def ffn(sd, pre, x):
    """BT:1046-1120: Linear-ReLU-Linear + identity."""
    y = linear(sd, pre + '.layers.1', F.relu(linear(sd, pre + '.layers.0.0', x)))
    return x + y


# --------------------------------------------------------------------------
# a2: MultiScaleDeformableAttention (encoder self-attention)  MO:305-412
# --------------------------------------------------------------------------
def msda_module(sd, pre, query, query_pos, key_padding_mask, reference_points, shapes, lsi,
                M=8, L=4, P=4, value=None):
    """value=None: self-attention over `query` (encoder); else cross-attention (refine decoder)."""
    identity = query
    if value is None:
        value = query
    q = query + query_pos if query_pos is not None else query
    q = q.permute(1, 0, 2)
    v = value.permute(1, 0, 2)
    bs, nq, _ = q.shape
    nv = v.shape[1]
    v = linear(sd, pre + '.value_proj', v)
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask[..., None], 0.0)  # MO:369-371 (mask AFTER proj)
    v = v.view(bs, nv, M, -1)
    off = linear(sd, pre + '.sampling_offsets', q).view(bs, nq, M, L, P, 2)
    aw = linear(sd, pre + '.attention_weights', q).view(bs, nq, M, L * P).softmax(-1)
    aw = aw.view(bs, nq, M, L, P)
    if reference_points.shape[-1] == 2:
        norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1).to(q.dtype)
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = reference_points[:, :, None, :, None, :2] + \
            off / P * reference_points[:, :, None, :, None, 2:] * 0.5
    out = msda(v, shapes, lsi, loc, aw)
    out = linear(sd, pre + '.output_proj', out).permute(1, 0, 2)
    return out + identity


# --------------------------------------------------------------------------
# a15: single-frame MultiScaleDeformablePoseAttention  OT:328-427
# --------------------------------------------------------------------------
def pose_attn_single(sd, pre, query, value, query_pos, key_padding_mask, reference_points,
                     shapes, lsi, M=8, L=4, K=17):
    residual = query
    q = (query + query_pos).permute(1, 0, 2)
    v = value.permute(1, 0, 2)
    bs, nq, _ = q.shape
    nv = v.shape[1]
    v = linear(sd, pre + '.value_proj', v)
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask[..., None], 0.0)  # OT:386-388 (mask after proj)
    v = v.view(bs, nv, M, -1)
    off = linear(sd, pre + '.sampling_offsets', q).view(bs, nq, M, L, K, 2)
    aw = linear(sd, pre + '.attention_weights', q).view(bs, nq, M, L * K).softmax(-1)
    aw = aw.view(bs, nq, M, L, K)
    rp = reference_points.reshape(bs, nq, L, -1, 2).unsqueeze(2)
    x1 = reference_points[:, :, :, 0::2].min(dim=-1, keepdim=True)[0]
    y1 = reference_points[:, :, :, 1::2].min(dim=-1, keepdim=True)[0]
    x2 = reference_points[:, :, :, 0::2].max(dim=-1, keepdim=True)[0]
    y2 = reference_points[:, :, :, 1::2].max(dim=-1, keepdim=True)[0]
    w = torch.clamp(x2 - x1, min=1e-4)
    h = torch.clamp(y2 - y1, min=1e-4)
    wh = torch.cat([w, h], dim=-1)[:, :, None, :, None, :]
    loc = rp + off * wh * 0.5
    out = msda(v, shapes, lsi, loc, aw)
    out = linear(sd, pre + '.output_proj', out).permute(1, 0, 2)
    return out + residual


# --------------------------------------------------------------------------
# a5: pose-aware T-frame cross-attention  OT:1644-1863 (T=3) / 2738-3117 (T=5)
# --------------------------------------------------------------------------
def pose_attn_mulframes(sd, pre, T, query, value, query_pos, key_padding_mask,
                        reference_points, shapes, lsi, M=8, L=4, K=15):
    """query [Q, B, C]; value [S, B*T, C] frame-interleaved; reference_points
    [B, T*Q, L, 2K] frame-major on dim 1; key_padding_mask [B*T, S]."""
    residual = query
    q = (query + query_pos).permute(1, 0, 2)
    v = value.permute(1, 0, 2)
    bs, nq, _ = q.shape
    nk = v.shape[1]
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask[..., None], 0.0)  # OT:1706-1707 (mask BEFORE proj)
    v = linear(sd, pre + '.value_proj', v)
    outs, zs = [], []
    for t, fp in enumerate(frame_prefixes(T)):
        vt = v[t::T].reshape(bs, nk, M, -1).contiguous()
        off = linear(sd, f'{pre}.{fp}sampling_offsets', q).view(bs, nq, M, L, K, 2)
        lg = linear(sd, f'{pre}.{fp}attention_weights', q).view(bs, nq, M, L * K)
        zs.append(torch.exp(lg).sum(-1, keepdim=True))  # OT:1737-1739 (un-stabilised)
        aw = lg.softmax(-1).view(bs, nq, M, L, K)
        rp_t = reference_points[:, t * nq:(t + 1) * nq]
        rp = rp_t.reshape(bs, nq, L, -1, 2).unsqueeze(2)
        x1 = rp_t[:, :, :, 0::2].min(dim=-1, keepdim=True)[0]
        y1 = rp_t[:, :, :, 1::2].min(dim=-1, keepdim=True)[0]
        x2 = rp_t[:, :, :, 0::2].max(dim=-1, keepdim=True)[0]
        y2 = rp_t[:, :, :, 1::2].max(dim=-1, keepdim=True)[0]
        w = torch.clamp(x2 - x1, min=1e-4)
        h = torch.clamp(y2 - y1, min=1e-4)
        wh = torch.cat([w, h], dim=-1)[:, :, None, :, None, :]
        loc = rp + off * wh * 0.5  # OT:1803-1811
        outs.append(msda(vt, shapes, lsi, loc, aw).reshape(bs, nq, M, -1))
    z_all = sum(zs)
    out = sum(o * (z / z_all) for o, z in zip(outs, zs)).flatten(-2, -1)  # OT:1854-1858
    out = linear(sd, pre + '.output_proj', out).permute(1, 0, 2)
    return out + residual


# --------------------------------------------------------------------------
# a7: joint-decoder T-frame cross-attention  MO:1388-1587 (T=3) / 1590-1981 (T=5)
# --------------------------------------------------------------------------
def joint_attn_mulframes(sd, pre, T, query, value, query_pos, key_padding_mask,
                         reference_points, shapes, lsi, M=8, L=4, P=4):
    """query [K, N, C]; value [S, N, T, C] (memory replicated per pose, OT:21498);
    key_padding_mask [N, T, S]; reference_points [T*N, K, L, 2] frame-major."""
    identity = query
    q = (query + query_pos).permute(1, 0, 2)
    v = value.permute(1, 0, 2, 3)
    bs, nq, _ = q.shape
    nv = v.shape[1]
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask.transpose(1, 2)[..., None], 0.0)  # MO:1454-1455
    v = linear(sd, pre + '.value_proj', v)
    norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1).to(q.dtype)
    outs, zs = [], []
    for t, fp in enumerate(frame_prefixes(T)):
        vt = v[:, :, t].reshape(bs, nv, M, -1).contiguous()
        off = linear(sd, f'{pre}.{fp}sampling_offsets', q).view(bs, nq, M, L, P, 2)
        lg = linear(sd, f'{pre}.{fp}attention_weights', q).view(bs, nq, M, L * P)
        zs.append(torch.exp(lg).sum(-1, keepdim=True))
        aw = lg.softmax(-1).view(bs, nq, M, L, P)
        rp = reference_points[t * bs:(t + 1) * bs]
        loc = rp[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        outs.append(msda(vt, shapes, lsi, loc, aw).reshape(bs, nq, M, -1))
    z_all = sum(zs)
    out = sum(o * (z / z_all) for o, z in zip(outs, zs)).flatten(-2, -1)
    out = linear(sd, pre + '.output_proj', out).permute(1, 0, 2)
    return out + identity


# --------------------------------------------------------------------------
# a13: SinePositionalEncoding  positional_encoding.py:56-93
# --------------------------------------------------------------------------
def sine_pos_enc(mask, num_feats=128, temperature=10000, normalize=True,
                 scale=2 * math.pi, eps=1e-6, offset=0.0):
    not_mask = 1 - mask.to(torch.int)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    if normalize:
        y_embed = (y_embed + offset) / (y_embed[:, -1:, :] + eps) * scale
        x_embed = (x_embed + offset) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature**(2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    B, H, W = mask.size()
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()),
                        dim=4).view(B, H, W, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()),
                        dim=4).view(B, H, W, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------
# a14: ResNet-50 (mmdet, style='pytorch', BN eval) + ChannelMapper
# --------------------------------------------------------------------------
def _bn(sd, pre, x):
    return F.batch_norm(x, sd[pre + '.running_mean'], sd[pre + '.running_var'],
                        sd[pre + '.weight'], sd[pre + '.bias'], False, 0.0, 1e-5)


def resnet_forward(sd, pre, x, depth=50, out_indices=(1, 2, 3)):
    """third_party/mmdetection/mmdet/models/backbones/resnet.py:632-654 (Bottleneck, pytorch style)."""
    assert depth in (50, 101)
    blocks = (3, 4, 6, 3) if depth == 50 else (3, 4, 23, 3)
    if x.dim() == 5:
        x = x.flatten(0, 1)  # input_type='mul_frames' resnet.py:634-639
    x = F.relu(_bn(sd, pre + '.bn1', F.conv2d(x, sd[pre + '.conv1.weight'], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for i, nb in enumerate(blocks):
        for b in range(nb):
            bp = f'{pre}.layer{i + 1}.{b}'
            stride = 2 if (b == 0 and i > 0) else 1
            idt = x
            y = F.relu(_bn(sd, bp + '.bn1', F.conv2d(x, sd[bp + '.conv1.weight'])))
            y = F.relu(_bn(sd, bp + '.bn2', F.conv2d(y, sd[bp + '.conv2.weight'], None, stride, 1)))
            y = _bn(sd, bp + '.bn3', F.conv2d(y, sd[bp + '.conv3.weight']))
            if bp + '.downsample.0.weight' in sd:
                idt = _bn(sd, bp + '.downsample.1',
                          F.conv2d(x, sd[bp + '.downsample.0.weight'], None, stride))
            x = F.relu(y + idt)
        if i in out_indices:
            outs.append(x)
    return outs


# --------------------------------------------------------------------------
# a14: HRNet (mmdet)  third_party/mmdetection/mmdet/models/backbones/hrnet.py
# The module tree is read off the state-dict keys, so one function serves every `extra` config.
# --------------------------------------------------------------------------
def _conv_bn(sd, conv, bn, x, stride=1, pad=0):
    return _bn(sd, bn, F.conv2d(x, sd[conv + '.weight'], None, stride, pad))


def _hr_basic_block(sd, bp, x):
    """resnet.py BasicBlock.forward (3x3 - 3x3, identity shortcut; HRNet branches never stride)."""
    y = F.relu(_conv_bn(sd, bp + '.conv1', bp + '.bn1', x, 1, 1))
    y = _conv_bn(sd, bp + '.conv2', bp + '.bn2', y, 1, 1)
    return F.relu(y + x)


def _hr_bottleneck(sd, bp, x):
    """resnet.py Bottleneck.forward, style='pytorch', stride 1 (HRNet layer1, hrnet.py:462-507)."""
    y = F.relu(_conv_bn(sd, bp + '.conv1', bp + '.bn1', x))
    y = F.relu(_conv_bn(sd, bp + '.conv2', bp + '.bn2', y, 1, 1))
    y = _conv_bn(sd, bp + '.conv3', bp + '.bn3', y)
    idt = x
    if bp + '.downsample.0.weight' in sd:
        idt = _conv_bn(sd, bp + '.downsample.0', bp + '.downsample.1', x)
    return F.relu(y + idt)


def _hr_blocks(sd, pre, x):
    b = 0
    while f'{pre}.{b}.conv1.weight' in sd:
        bp = f'{pre}.{b}'
        x = _hr_bottleneck(sd, bp, x) if bp + '.conv3.weight' in sd else _hr_basic_block(sd, bp, x)
        b += 1
    return x


def _has_prefix(sd, pre):
    return any(k.startswith(pre) for k in sd)


def hr_module(sd, mp, xs):
    """HRModule.forward hrnet.py:183-205 (+ fuse layers :121-181)."""
    nb = len(xs)
    xs = [_hr_blocks(sd, f'{mp}.branches.{i}', xs[i]) for i in range(nb)]
    if nb == 1:
        return xs
    outs = []
    i = 0
    while _has_prefix(sd, f'{mp}.fuse_layers.{i}.'):
        y = 0
        for j in range(nb):
            fp = f'{mp}.fuse_layers.{i}.{j}'
            if j == i:
                y = y + xs[j]
            elif j > i:  # 1x1 conv + BN + nearest upsample by 2^(j-i)
                t = _conv_bn(sd, fp + '.0', fp + '.1', xs[j])
                y = y + F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:        # (i - j) stride-2 3x3 conv + BN, ReLU between them but not after the last
                t = xs[j]
                for k in range(i - j):
                    t = _conv_bn(sd, f'{fp}.{k}.0', f'{fp}.{k}.1', t, 2, 1)
                    if k != i - j - 1:
                        t = F.relu(t)
                y = y + t
        outs.append(F.relu(y))
        i += 1
    return outs


def _hr_transition(sd, tp, i, pre_list):
    """HRNet._make_transition_layer hrnet.py:416-460: None / 3x3 conv / chain of stride-2 convs."""
    if f'{tp}.{i}.0.weight' in sd:
        return F.relu(_conv_bn(sd, f'{tp}.{i}.0', f'{tp}.{i}.1', pre_list[i], 1, 1))
    if f'{tp}.{i}.0.0.weight' in sd:
        t = pre_list[-1]
        j = 0
        while f'{tp}.{i}.{j}.0.weight' in sd:
            t = F.relu(_conv_bn(sd, f'{tp}.{i}.{j}.0', f'{tp}.{i}.{j}.1', t, 2, 1))
            j += 1
        return t
    return pre_list[i]


def hrnet_forward(sd, pre, x, stage_branches=(2, 3, 4)):
    """HRNet.forward hrnet.py:549-583; the reference returns y_list[1:] (three coarser branches).
    5-d video input is flattened to frames first (the reference HRNet has no such branch; the
    build's HRNet under the MulFrames head does exactly this)."""
    if x.dim() == 5:
        x = x.flatten(0, 1)
    x = F.relu(_conv_bn(sd, pre + '.conv1', pre + '.bn1', x, 2, 1))
    x = F.relu(_conv_bn(sd, pre + '.conv2', pre + '.bn2', x, 2, 1))
    x = _hr_blocks(sd, pre + '.layer1', x)
    y_list = [x]
    for s, nbr in enumerate(stage_branches):
        tp = f'{pre}.transition{s + 1}'
        x_list = [_hr_transition(sd, tp, i, y_list) for i in range(nbr)]
        m = 0
        while _has_prefix(sd, f'{pre}.stage{s + 2}.{m}.'):
            x_list = hr_module(sd, f'{pre}.stage{s + 2}.{m}', x_list)
            m += 1
        y_list = x_list
    return y_list[1:]


def backbone_forward(sd, cfg, img):
    if cfg.get('backbone', 'resnet') == 'hrnet':
        return hrnet_forward(sd, 'backbone', img)
    return resnet_forward(sd, 'backbone', img, depth=cfg.get('depth', 50))


def channel_mapper(sd, pre, feats, num_groups=32):
    """third_party/mmdetection/mmdet/models/necks/channel_mapper.py:90-100 (conv + GN, no act)."""
    outs = []
    for i, f in enumerate(feats):
        y = F.conv2d(f, sd[f'{pre}.convs.{i}.conv.weight'], sd.get(f'{pre}.convs.{i}.conv.bias'))
        outs.append(F.group_norm(y, num_groups, sd[f'{pre}.convs.{i}.gn.weight'],
                                 sd[f'{pre}.convs.{i}.gn.bias'], 1e-5))
    i = 0
    while f'{pre}.extra_convs.{i}.conv.weight' in sd:
        src = feats[-1] if i == 0 else outs[-1]
        y = F.conv2d(src, sd[f'{pre}.extra_convs.{i}.conv.weight'],
                     sd.get(f'{pre}.extra_convs.{i}.conv.bias'), 2, 1)
        outs.append(F.group_norm(y, num_groups, sd[f'{pre}.extra_convs.{i}.gn.weight'],
                                 sd[f'{pre}.extra_convs.{i}.gn.bias'], 1e-5))
        i += 1
    return outs


# --------------------------------------------------------------------------
# a4 helpers  OT:21095-21216
# --------------------------------------------------------------------------
def get_valid_ratio(mask):
    _, H, W = mask.shape
    valid_H = torch.sum(~mask[:, :, 0], 1)
    valid_W = torch.sum(~mask[:, 0, :], 1)
    return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)


def get_reference_points(spatial_shapes, valid_ratios):
    refs = []
    for lvl, (H, W) in enumerate(spatial_shapes):
        H, W = int(H), int(W)
        ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float32),
                                      torch.linspace(0.5, W - 0.5, W, dtype=torch.float32),
                                      indexing='ij')
        ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        refs.append(torch.stack((ref_x, ref_y), -1))
    reference_points = torch.cat(refs, 1)
    return reference_points[:, :, None] * valid_ratios[:, None]


def gen_encoder_output_proposals(sd, pre, memory, memory_padding_mask, spatial_shapes):
    N, S, C = memory.shape
    proposals = []
    _cur = 0
    for lvl, (H, W) in enumerate(spatial_shapes):
        H, W = int(H), int(W)
        m = memory_padding_mask[:, _cur:(_cur + H * W)].view(N, H, W, 1)
        valid_H = torch.sum(~m[:, :, 0, 0], 1)
        valid_W = torch.sum(~m[:, 0, :, 0], 1)
        grid_y, grid_x = torch.meshgrid(torch.linspace(0, H - 1, H, dtype=torch.float32),
                                        torch.linspace(0, W - 1, W, dtype=torch.float32),
                                        indexing='ij')
        grid = torch.cat([grid_x.unsqueeze(-1), grid_y.unsqueeze(-1)], -1)
        scale = torch.cat([valid_W.unsqueeze(-1), valid_H.unsqueeze(-1)], 1).view(N, 1, 1, 2)
        grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
        proposals.append(grid.view(N, -1, 2))
        _cur += H * W
    output_proposals = torch.cat(proposals, 1)
    valid = ((output_proposals > 0.01) & (output_proposals < 0.99)).all(-1, keepdim=True)
    output_proposals = torch.log(output_proposals / (1 - output_proposals))
    output_proposals = output_proposals.masked_fill(memory_padding_mask.unsqueeze(-1), float('inf'))
    output_proposals = output_proposals.masked_fill(~valid, float('inf'))
    output_memory = memory.masked_fill(memory_padding_mask.unsqueeze(-1), float(0))
    output_memory = output_memory.masked_fill(~valid, float(0))
    output_memory = layer_norm(sd, pre + '.enc_output_norm', linear(sd, pre + '.enc_output', output_memory))
    return output_memory, output_proposals


# --------------------------------------------------------------------------
# encoder (hot loop #1): DetrTransformerEncoder of BaseTransformerLayer
# ('self_attn','norm','ffn','norm')  MT:501-531, BT:1253-1353
# --------------------------------------------------------------------------
def encoder_forward(sd, pre, num_layers, feat, pos, mask, reference_points, shapes, lsi):
    x = feat
    for i in range(num_layers):
        lp = f'{pre}.layers.{i}'
        x = msda_module(sd, lp + '.attentions.0', x, pos, mask, reference_points, shapes, lsi)
        x = layer_norm(sd, lp + '.norms.0', x)
        x = ffn(sd, lp + '.ffns.0', x)
        x = layer_norm(sd, lp + '.norms.1', x)
    return x


def flatten_levels(sd, tpre, mlvl_feats, mlvl_masks, mlvl_pos):
    """OT:21277-21310."""
    feat_f, mask_f, pos_f, shapes = [], [], [], []
    for lvl, (feat, mask, pos) in enumerate(zip(mlvl_feats, mlvl_masks, mlvl_pos)):
        bs, c, h, w = feat.shape
        shapes.append((h, w))
        feat_f.append(feat.flatten(2).transpose(1, 2))
        mask_f.append(mask.flatten(1))
        pos_f.append(pos.flatten(2).transpose(1, 2) + sd[tpre + '.level_embeds'][lvl].view(1, 1, -1))
    feat_f = torch.cat(feat_f, 1)
    mask_f = torch.cat(mask_f, 1)
    pos_f = torch.cat(pos_f, 1)
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    valid_ratios = torch.stack([get_valid_ratio(m) for m in mlvl_masks], 1)
    return feat_f, mask_f, pos_f, shapes, lsi, valid_ratios


# --------------------------------------------------------------------------
# a4 + a6: VideoPoseTransformerMulFrames.forward  OT:21218-21456,
#          VideoPoseTransformerDecoderV2  OT:6681-6753 (generalised T)
# --------------------------------------------------------------------------
def videopose_transformer_forward(sd, hpre, cfg, mlvl_feats, mlvl_masks, mlvl_pos, taps=None):
    T, K, Q = cfg['num_frames'], cfg['num_keypoints'], cfg['num_query']
    n_enc, n_dec = cfg.get('enc_layers', 6), cfg.get('dec_layers', 3)
    tpre = hpre + '.transformer'
    feat_f, mask_f, pos_f, shapes, lsi, valid_ratios = flatten_levels(
        sd, tpre, mlvl_feats, mlvl_masks, mlvl_pos)
    reference_points = get_reference_points(shapes, valid_ratios)
    memory = encoder_forward(sd, tpre + '.encoder', n_enc, feat_f.permute(1, 0, 2),
                             pos_f.permute(1, 0, 2), mask_f, reference_points, shapes, lsi)
    memory = memory.permute(1, 0, 2)  # [B*T, S, C]
    bs, _, c = memory.shape
    if taps is not None:
        taps['memory'] = memory
    now = slice(T // 2, None, T)
    now_memory, now_mask, now_vr = memory[now], mask_f[now], valid_ratios[now]
    output_memory, output_proposals = gen_encoder_output_proposals(
        sd, tpre, now_memory, now_mask, shapes)
    enc_cls = linear(sd, f'{hpre}.cls_branches.{n_dec}', output_memory)
    enc_kpt = kpt_branch(sd, f'{hpre}.kpt_branches.{n_dec}', output_memory)
    enc_kpt[..., 0::2] += output_proposals[..., 0:1]
    enc_kpt[..., 1::2] += output_proposals[..., 1:2]
    enc_sigma = sigma_branch(sd, f'{hpre}.dec_fc_sigma_branches.{n_dec}', output_memory)
    topk_idx = torch.topk(enc_cls[..., 0], Q, dim=1)[1]
    if taps is not None:
        taps['enc_cls'] = enc_cls
        taps['topk_idx'] = topk_idx
        if 'force_topk_idx' in taps:
            topk_idx = taps['force_topk_idx']
    topk_kpts = torch.gather(enc_kpt, 1, topk_idx.unsqueeze(-1).repeat(1, 1, enc_kpt.size(-1)))
    tgt = torch.gather(output_memory, 1, topk_idx.unsqueeze(-1).repeat(1, 1, c))
    reference_points = topk_kpts.sigmoid().repeat(1, T, 1)  # [B, T*Q, 2K]
    init_reference = reference_points
    qe = sd[hpre + '.query_embedding.weight']
    query_pos, query = torch.split(qe, c, dim=1)
    B = bs // T
    query_pos = query_pos.unsqueeze(0).expand(B, -1, -1).permute(1, 0, 2)
    query = (tgt + query.unsqueeze(0).expand(B, -1, -1)).permute(1, 0, 2)
    value = memory.permute(1, 0, 2)  # [S, B*T, C]

    prefixes = frame_prefixes(T)
    inter, inter_refs = [], []
    out = query
    for lid in range(n_dec):
        lp = f'{tpre}.decoder.layers.{lid}'
        ref_in = reference_points[:, :, None] * now_vr.repeat(1, 1, K)[:, None]  # OT:6712-6715
        out = mha(sd, lp + '.attentions.0', out, query_pos)
        out = layer_norm(sd, lp + '.norms.0', out)
        out = pose_attn_mulframes(sd, lp + '.attentions.1', T, out, value, query_pos, mask_f,
                                  ref_in, shapes, lsi, K=K)
        out = layer_norm(sd, lp + '.norms.1', out)
        out = ffn(sd, lp + '.ffns.0', out)
        out = layer_norm(sd, lp + '.norms.2', out)
        o = out.permute(1, 0, 2)
        tmps = torch.cat([kpt_branch(sd, f'{hpre}.{fp}kpt_branches.{lid}', o) for fp in prefixes],
                         dim=1)  # OT:6728-6732
        reference_points = (tmps + inverse_sigmoid(reference_points)).sigmoid()
        inter.append(out)
        inter_refs.append(reference_points)
    hs = torch.stack(inter)
    inter_refs = torch.stack(inter_refs)
    return dict(hs=hs, init_reference=init_reference, inter_references=inter_refs,
                enc_cls=enc_cls, enc_kpt=enc_kpt, enc_sigma=enc_sigma, memory=value,
                shapes=shapes, lsi=lsi, mask_flatten=mask_f, valid_ratios=valid_ratios)


# --------------------------------------------------------------------------
# a9 + a8: forward_refine  OT:21458-21536, DeformableDetrTransformerDecoderV1 MT:809-886
# --------------------------------------------------------------------------
def videopose_transformer_refine(sd, hpre, cfg, mlvl_masks, memory, ref_pose, img_inds):
    """memory [S, B, T, C]; ref_pose [T*N, 2K] ordered frame-major (pre..., now..., next...)."""
    T, K = cfg['num_frames'], cfg['num_keypoints']
    n_ref = cfg.get('refine_layers', 2)
    tpre = hpre + '.transformer'
    mask_f, shapes = [], []
    for mask in mlvl_masks:
        bs, h, w = mask.shape
        shapes.append((h, w))
        mask_f.append(mask.flatten(1))
    mask_f = torch.cat(mask_f, 1)
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    valid_ratios = torch.stack([get_valid_ratio(m) for m in mlvl_masks], 1)
    rq = sd[tpre + '.refine_query_embedding.weight']
    query_pos, query = torch.split(rq, rq.size(1) // 2, dim=1)
    N = ref_pose.size(0) // T
    query_pos = query_pos.unsqueeze(0).expand(N, -1, -1).permute(1, 0, 2)
    query = query.unsqueeze(0).expand(N, -1, -1).permute(1, 0, 2)
    reference_points = ref_pose.reshape(-1, ref_pose.size(1) // 2, 2)  # [T*N, K, 2]
    pos_memory = memory[:, img_inds, :, :]  # OT:21498 replicate per pose
    mask_f = mask_f.reshape(-1, T, mask_f.size(-1))[img_inds, :]
    vr = valid_ratios.reshape(-1, T, valid_ratios.size(-2), valid_ratios.size(-1))[img_inds, ...]
    vr = vr.flatten(0, 1)  # pose-major [N*T, L, 2] (reference quirk a8; harmless: equal per clip)
    prefixes = frame_prefixes(T)
    out = query
    inter, inter_refs = [], []
    init_reference = reference_points
    for lid in range(n_ref):
        lp = f'{tpre}.refine_decoder.layers.{lid}'
        ref_in = reference_points[:, :, None] * vr[:, None]  # MT:848-850
        out = mha(sd, lp + '.attentions.0', out, query_pos)
        out = layer_norm(sd, lp + '.norms.0', out)
        out = joint_attn_mulframes(sd, lp + '.attentions.1', T, out, pos_memory, query_pos,
                                   mask_f, ref_in, shapes, lsi)
        out = layer_norm(sd, lp + '.norms.1', out)
        out = ffn(sd, lp + '.ffns.0', out)
        out = layer_norm(sd, lp + '.norms.2', out)
        o = out.permute(1, 0, 2)
        tmps = torch.cat([refine_kpt_branch(sd, f'{hpre}.{fp}refine_kpt_branches.{lid}', o)
                          for fp in prefixes], dim=0)  # MT:861-864
        reference_points = (tmps + inverse_sigmoid(reference_points)).sigmoid()
        inter.append(out)
        inter_refs.append(reference_points)
    return torch.stack(inter), init_reference, torch.stack(inter_refs)


# --------------------------------------------------------------------------
# a12: OKS-NMS (NumPy, as the reference)  HEAD:1624-1665
# --------------------------------------------------------------------------
def oks_iou(g, d, a_g, a_d, sigmas):
    vars_ = (sigmas * 2)**2
    xg, yg = g[0::3], g[1::3]
    ious = np.zeros((d.shape[0]))
    for n_d in range(0, d.shape[0]):
        xd, yd = d[n_d, 0::3], d[n_d, 1::3]
        dx, dy = xd - xg, yd - yg
        e = (dx**2 + dy**2) / vars_ / ((a_g + a_d[n_d]) / 2 + np.spacing(1)) / 2
        ious[n_d] = np.sum(np.exp(-e)) / e.shape[0] if e.shape[0] != 0 else 0.0
    return ious


def oks_nms(poses, scores, thresh, sigmas):
    poses = poses.cpu().numpy()
    scores = scores.cpu().numpy()
    if len(poses) == 0:
        return []
    areas = (np.max(poses[:, :, 0], axis=1) - np.min(poses[:, :, 0], axis=1)) * \
            (np.max(poses[:, :, 1], axis=1) - np.min(poses[:, :, 1], axis=1))
    poses = poses.reshape(poses.shape[0], -1)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        ovr = oks_iou(poses[i], poses[order[1:]], areas[i], areas[order[1:]], sigmas)
        inds = np.where(ovr <= thresh)[0]
        order = order[inds + 1]
    return keep


OKS_SIGMAS_15 = np.array([.26, .79, .79, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87,
                          .89, .89]) / 10.0


def get_p(sigma, p_x=0.2):
    """HEAD:1531-1535."""
    p = 1 - torch.exp(-(p_x / sigma))
    p = p[:, :, 0] * p[:, :, 1]
    return p[:, :, None] * 0.7


# --------------------------------------------------------------------------
# a11 + a12: head forward / get_bboxes  HEAD:403-567, 569-674, 1371-1505
# --------------------------------------------------------------------------
def make_masks_and_pos(feats, batch_input_shape, img_shape, pos_offset=-0.5):
    """HEAD:429-445 (single clip: every frame shares img_shape)."""
    n = feats[0].size(0)
    H, W = batch_input_shape
    img_masks = feats[0].new_ones((n, H, W))
    img_masks[:, :img_shape[0], :img_shape[1]] = 0
    masks, poss = [], []
    for f in feats:
        m = F.interpolate(img_masks[None], size=f.shape[-2:]).to(torch.bool).squeeze(0)
        masks.append(m)
        poss.append(sine_pos_enc(m, offset=pos_offset))
    return masks, poss


def videopose_simple_test(sd, cfg, img, img_shape=None, rescale_factor=None, taps=None):
    """VideoPoseV1.simple_test for ONE clip: img [1, T, 3, H, W] -> (bboxes[n,5], labels[n],
    kpts[n,K,3]).  videoposev1.py:159-190, HEAD:1507-1529."""
    T, K, Q = cfg['num_frames'], cfg['num_keypoints'], cfg['num_query']
    N = cfg.get('max_per_img', 20)
    n_dec, n_ref = cfg.get('dec_layers', 3), cfg.get('refine_layers', 2)
    hpre = 'bbox_head'
    H, W = img.shape[-2:]
    if img_shape is None:
        img_shape = (H, W, 3)
    feats = channel_mapper(sd, 'neck', backbone_forward(sd, cfg, img))
    if taps is not None:
        taps['neck'] = feats
    masks, poss = make_masks_and_pos(feats, (H, W), img_shape)
    tr = videopose_transformer_forward(sd, hpre, cfg, feats, masks, poss, taps=taps)
    hs = tr['hs'].permute(0, 2, 1, 3)  # [n_dec, B, Q, C]
    prefixes = frame_prefixes(T)
    c = T // 2
    # HEAD:478-549 -- last decoder layer only matters at inference
    lvl = n_dec - 1
    reference = tr['init_reference'] if lvl == 0 else tr['inter_references'][lvl - 1]
    poses_t = []
    for t, fp in enumerate(prefixes):
        ref_t = inverse_sigmoid(reference[:, t * Q:(t + 1) * Q])
        br = fp
        if T == 5 and t == 4 and not cfg.get('fix_next_next_typo', False):
            br = 'next_'  # HEAD:503 decodes next_next with next_kpt_branches (reference quirk)
        poses_t.append((kpt_branch(sd, f'{hpre}.{br}kpt_branches.{lvl}', hs[lvl]) + ref_t).sigmoid())
    cls = linear(sd, f'{hpre}.cls_branches.{lvl}', hs[lvl])
    if taps is not None:
        taps.update(hs=tr['hs'], inter_references=tr['inter_references'],
                    init_reference=tr['init_reference'], cls_last=cls, kpt_last=poses_t[c])
    # _get_bboxes_single HEAD:1371-1505 (B = 1)
    cls_score = cls[0].sigmoid()
    scores, indexs = cls_score.view(-1).topk(N)
    if taps is not None:
        taps['score_topk_idx'] = indexs
        if 'force_score_topk_idx' in taps:
            indexs = taps['force_score_topk_idx']
            scores = cls_score.view(-1)[indexs]
    det_labels = indexs % 1
    bbox_index = indexs // 1
    sel = [p.flatten(0, 1)[bbox_index] for p in poses_t]
    ref_pose = torch.cat(sel, dim=0)  # [T*N, 2K] frame-major HEAD:610
    img_inds = (torch.arange(N)[:, None] / Q).squeeze(1).to(torch.int64)  # HEAD:612
    S = tr['memory'].size(0)
    mem4 = tr['memory'].reshape(S, -1, T, tr['memory'].size(-1))
    rhs, rinit, rrefs = videopose_transformer_refine(sd, hpre, cfg, masks, mem4, ref_pose, img_inds)
    rhs = rhs.permute(0, 2, 1, 3)  # [n_ref, N, K, C]
    rl = n_ref - 1
    reference = rinit if rl == 0 else rrefs[rl - 1]
    reference = inverse_sigmoid(reference[c * N:(c + 1) * N])
    det_kpts = (refine_kpt_branch(sd, f'{hpre}.refine_kpt_branches.{rl}', rhs[rl]) + reference).sigmoid()
    det_sigma = sigma_branch(sd, f'{hpre}.refine_fc_sigma_branches.{rl}', rhs[rl]).sigmoid()
    if taps is not None:
        taps.update(refine_hs=rhs, refine_kpts=det_kpts.clone(), refine_sigma=det_sigma)
    det_kpts[..., 0] = det_kpts[..., 0] * img_shape[1]
    det_kpts[..., 1] = det_kpts[..., 1] * img_shape[0]
    det_kpts[..., 0].clamp_(min=0, max=img_shape[1])
    det_kpts[..., 1].clamp_(min=0, max=img_shape[0])
    if rescale_factor is not None:
        det_kpts /= det_kpts.new_tensor(rescale_factor[:2]).unsqueeze(0).unsqueeze(0)
    x1 = det_kpts[..., 0].min(dim=1, keepdim=True)[0]
    y1 = det_kpts[..., 1].min(dim=1, keepdim=True)[0]
    x2 = det_kpts[..., 0].max(dim=1, keepdim=True)[0]
    y2 = det_kpts[..., 1].max(dim=1, keepdim=True)[0]
    det_bboxes = torch.cat([x1, y1, x2, y2], dim=1)
    p = get_p(det_sigma)
    det_kpts = (det_kpts * p**5) / (p**5 + 1e-10)
    det_bboxes = torch.cat((det_bboxes, scores.unsqueeze(1)), -1)
    kpt_scores = scores[:, None, None] * p
    det_kpts = torch.cat((det_kpts, kpt_scores), dim=2)
    sig = OKS_SIGMAS_15 if K == 15 else np.full((K,), 0.079)
    keep = oks_nms(det_kpts, scores, 0.45, sig)
    keep = torch.as_tensor(np.asarray(keep, dtype=np.int64))
    return det_bboxes[keep], det_labels[keep], det_kpts[keep]


# --------------------------------------------------------------------------
# a15 / a16: single-image PETR and the vedpose single-frame head
#   PETRTransformer.forward / forward_refine  OT:4448-4693
#   PetrTransformerDecoder                    OT:4148-4231
#   DeformableDetrTransformerDecoder          MT:705-791
#   PETRHead.forward / _get_bboxes_single     petr_head.py:213-300, 956-1037
#   VedPoseHeadV2._get_bboxes_single          vedpose_head_v2.py:1066-1180
# --------------------------------------------------------------------------
def petr_simple_test(sd, cfg, img, img_shape=None, taps=None):
    """img [1, 3, H, W] -> (bboxes [N,5], labels [N], kpts [N,K,3]).  cfg: num_keypoints,
    num_query, max_per_img, head in {'petr', 'vedpose'}."""
    K, Q, N = cfg['num_keypoints'], cfg['num_query'], cfg.get('max_per_img', 100)
    n_dec, n_ref = cfg.get('dec_layers', 3), cfg.get('refine_layers', 2)
    vedpose = cfg.get('head', 'petr') == 'vedpose'
    hpre, tpre = 'bbox_head', 'bbox_head.transformer'
    H, W = img.shape[-2:]
    if img_shape is None:
        img_shape = (H, W, 3)
    feats = channel_mapper(sd, 'neck', backbone_forward(sd, cfg, img))
    masks, poss = make_masks_and_pos(feats, (H, W), img_shape)
    feat_f, mask_f, pos_f, shapes, lsi, valid_ratios = flatten_levels(sd, tpre, feats, masks, poss)
    reference_points = get_reference_points(shapes, valid_ratios)
    memory = encoder_forward(sd, tpre + '.encoder', cfg.get('enc_layers', 6),
                             feat_f.permute(1, 0, 2), pos_f.permute(1, 0, 2), mask_f,
                             reference_points, shapes, lsi).permute(1, 0, 2)
    bs, _, c = memory.shape
    if taps is not None:
        taps['memory'] = memory
    output_memory, output_proposals = gen_encoder_output_proposals(sd, tpre, memory, mask_f, shapes)
    enc_cls = linear(sd, f'{hpre}.cls_branches.{n_dec}', output_memory)
    enc_kpt = kpt_branch(sd, f'{hpre}.kpt_branches.{n_dec}', output_memory)
    enc_kpt[..., 0::2] += output_proposals[..., 0:1]
    enc_kpt[..., 1::2] += output_proposals[..., 1:2]
    topk_idx = torch.topk(enc_cls[..., 0], Q, dim=1)[1]
    if taps is not None:
        taps['topk_idx'] = topk_idx
    topk_kpts = torch.gather(enc_kpt, 1, topk_idx.unsqueeze(-1).repeat(1, 1, enc_kpt.size(-1)))
    reference_points = topk_kpts.sigmoid()
    init_reference = reference_points
    query_pos, query = torch.split(sd[hpre + '.query_embedding.weight'], c, dim=1)
    query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1).permute(1, 0, 2)
    out = query.unsqueeze(0).expand(bs, -1, -1).permute(1, 0, 2)  # no memory rows added (OT:4596)
    value = memory.permute(1, 0, 2)
    inter, inter_refs = [], []
    for lid in range(n_dec):
        lp = f'{tpre}.decoder.layers.{lid}'
        ref_in = reference_points[:, :, None] * valid_ratios.repeat(1, 1, K)[:, None]
        out = mha(sd, lp + '.attentions.0', out, query_pos)
        out = layer_norm(sd, lp + '.norms.0', out)
        out = pose_attn_single(sd, lp + '.attentions.1', out, value, query_pos, mask_f, ref_in,
                               shapes, lsi, K=K)
        out = layer_norm(sd, lp + '.norms.1', out)
        out = ffn(sd, lp + '.ffns.0', out)
        out = layer_norm(sd, lp + '.norms.2', out)
        tmp = kpt_branch(sd, f'{hpre}.kpt_branches.{lid}', out.permute(1, 0, 2))
        reference_points = (tmp + inverse_sigmoid(reference_points)).sigmoid()
        inter.append(out)
        inter_refs.append(reference_points)
    hs = torch.stack(inter).permute(0, 2, 1, 3)
    inter_refs = torch.stack(inter_refs)
    lvl = n_dec - 1
    reference = init_reference if lvl == 0 else inter_refs[lvl - 1]
    cls = linear(sd, f'{hpre}.cls_branches.{lvl}', hs[lvl])
    kpt = (kpt_branch(sd, f'{hpre}.kpt_branches.{lvl}', hs[lvl]) + inverse_sigmoid(reference)).sigmoid()
    if taps is not None:
        taps.update(hs=torch.stack(inter), inter_references=inter_refs, cls_last=cls)
    cls_score = cls[0].sigmoid()
    scores, indexs = cls_score.view(-1).topk(N)
    if taps is not None:
        taps['score_topk_idx'] = indexs
    det_labels, bbox_index = indexs % 1, indexs // 1
    kpt_pred = kpt[0][bbox_index]
    # forward_refine OT:4638-4693 (memory replicated per pose)
    img_inds = (torch.arange(N)[:, None] / Q).squeeze(1).to(torch.int64)
    rq = sd[tpre + '.refine_query_embedding.weight']
    rpos, rquery = torch.split(rq, rq.size(1) // 2, dim=1)
    rpos = rpos.unsqueeze(0).expand(N, -1, -1).permute(1, 0, 2)
    rout = rquery.unsqueeze(0).expand(N, -1, -1).permute(1, 0, 2)
    rref = kpt_pred.reshape(N, kpt_pred.size(1) // 2, 2)
    pos_memory = value[:, img_inds, :]
    rmask = mask_f[img_inds, :]
    rvr = valid_ratios[img_inds, ...]
    rinit = rref
    rinter, rrefs = [], []
    for lid in range(n_ref):
        lp = f'{tpre}.refine_decoder.layers.{lid}'
        ref_in = rref[:, :, None] * rvr[:, None]
        rout = mha(sd, lp + '.attentions.0', rout, rpos)
        rout = layer_norm(sd, lp + '.norms.0', rout)
        rout = msda_module(sd, lp + '.attentions.1', rout, rpos, rmask, ref_in, shapes, lsi,
                           value=pos_memory)
        rout = layer_norm(sd, lp + '.norms.1', rout)
        rout = ffn(sd, lp + '.ffns.0', rout)
        rout = layer_norm(sd, lp + '.norms.2', rout)
        tmp = refine_kpt_branch(sd, f'{hpre}.refine_kpt_branches.{lid}', rout.permute(1, 0, 2))
        rref = (tmp + inverse_sigmoid(rref)).sigmoid()  # MT:764-770 with 2-d references
        rinter.append(rout)
        rrefs.append(rref)
    rhs = torch.stack(rinter).permute(0, 2, 1, 3)
    rl = n_ref - 1
    reference = inverse_sigmoid(rinit if rl == 0 else rrefs[rl - 1])
    det_kpts = (refine_kpt_branch(sd, f'{hpre}.refine_kpt_branches.{rl}', rhs[rl]) + reference).sigmoid()
    det_kpts[..., 0] = det_kpts[..., 0] * img_shape[1]
    det_kpts[..., 1] = det_kpts[..., 1] * img_shape[0]
    det_kpts[..., 0].clamp_(min=0, max=img_shape[1])
    det_kpts[..., 1].clamp_(min=0, max=img_shape[0])
    x1 = det_kpts[..., 0].min(dim=1, keepdim=True)[0]
    y1 = det_kpts[..., 1].min(dim=1, keepdim=True)[0]
    x2 = det_kpts[..., 0].max(dim=1, keepdim=True)[0]
    y2 = det_kpts[..., 1].max(dim=1, keepdim=True)[0]
    det_bboxes = torch.cat([x1, y1, x2, y2, scores.unsqueeze(1)], dim=1)
    if vedpose:
        sigma = sigma_branch(sd, f'{hpre}.refine_fc_sigma_branches.{rl}', rhs[rl]).sigmoid()
        p = get_p(sigma)
        det_kpts = (det_kpts * p**5) / (p**5 + 1e-10)
        kscore = scores[:, None, None] * p
    else:
        kscore = det_kpts.new_ones(det_kpts[..., :1].shape)
    det_kpts = torch.cat((det_kpts, kscore), dim=2)
    return det_bboxes, det_labels, det_kpts
