"""``mmdet.SwinTransformer`` incl. the fork's ``num_frames`` flatten (a14), restated from
third_party/mmdetection/mmdet/models/backbones/swin.py (WindowMSA :22-126, ShiftWindowMSA
:128-286, SwinBlock :288-379, SwinBlockSequence :381-465, SwinTransformer :467-768) and
mmdet/models/utils/transformer.py (AdaptivePadding :56-133, PatchEmbed :136-259, PatchMerging
:262-387).  Same ctor kwargs and state-dict keys; inference only (DropPath = identity).
On the device (fp32, no grad, the exact-split or fp16 GEMM mode) the forward runs on this package's kernels
(`SwinTransformer._forward_device`): every Linear -- patch embedding as a row GEMM over 4 x 4 patches, qkv, proj,
the FFN (exact GELU in the GEMM epilogue), the patch-merging reduction -- is a launch of the split GEMM with
bias / identity in its epilogue, LayerNorm is `pave_bias_add_layernorm_f32`, and the (shifted-)window attention
is ONE kernel on the un-partitioned token map (`pave_swin_window_attn_f32`: pad / roll / partition / mask /
reverse are index arithmetic) -- no library GEMM, no batched 49 x 49 products.  Elsewhere (CPU, autograd): the
plain torch formulation below, bit-identical to the reference backbone.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .bricks import FFN, BaseModule, build_norm_layer
from .registry import MMDET_MODELS


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class AdaptivePadding(nn.Module):

    def __init__(self, kernel_size=1, stride=1, dilation=1, padding='corner'):
        super().__init__()
        assert padding in ('same', 'corner')
        self.padding = padding
        self.kernel_size, self.stride, self.dilation = _pair(kernel_size), _pair(stride), _pair(dilation)

    def get_pad_shape(self, input_shape):
        ih, iw = input_shape
        kh, kw = self.kernel_size
        sh, sw = self.stride
        oh, ow = math.ceil(ih / sh), math.ceil(iw / sw)
        return (max((oh - 1) * sh + (kh - 1) * self.dilation[0] + 1 - ih, 0),
                max((ow - 1) * sw + (kw - 1) * self.dilation[1] + 1 - iw, 0))

    def forward(self, x):
        pad_h, pad_w = self.get_pad_shape(x.size()[-2:])
        if pad_h > 0 or pad_w > 0:
            if self.padding == 'corner':
                x = F.pad(x, [0, pad_w, 0, pad_h])
            else:
                x = F.pad(x, [pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2])
        return x


class PatchEmbed(BaseModule):

    def __init__(self, in_channels=3, embed_dims=768, conv_type='Conv2d', kernel_size=16,
                 stride=16, padding='corner', dilation=1, bias=True, norm_cfg=None,
                 input_size=None, init_cfg=None):
        super().__init__(init_cfg)
        self.embed_dims = embed_dims
        stride = stride if stride is not None else kernel_size
        if isinstance(padding, str):
            self.adap_padding = AdaptivePadding(kernel_size, stride, dilation, padding)
            padding = 0
        else:
            self.adap_padding = None
        self.projection = nn.Conv2d(in_channels, embed_dims, _pair(kernel_size), _pair(stride),
                                    _pair(padding), _pair(dilation), bias=bias)
        self.norm = build_norm_layer(norm_cfg, embed_dims)[1] if norm_cfg is not None else None

    def forward(self, x):
        if self.adap_padding:
            x = self.adap_padding(x)
        x = self.projection(x)
        out_size = (x.shape[2], x.shape[3])
        x = x.flatten(2).transpose(1, 2)
        if self.norm is not None:
            x = self.norm(x)
        return x, out_size


class PatchMerging(BaseModule):

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=None, padding='corner',
                 dilation=1, bias=False, norm_cfg=dict(type='LN'), init_cfg=None):
        super().__init__(init_cfg)
        self.in_channels, self.out_channels = in_channels, out_channels
        stride = stride if stride else kernel_size
        kernel_size, stride, dilation = _pair(kernel_size), _pair(stride), _pair(dilation)
        if isinstance(padding, str):
            self.adap_padding = AdaptivePadding(kernel_size, stride, dilation, padding)
            padding = 0
        else:
            self.adap_padding = None
        self.sampler = nn.Unfold(kernel_size=kernel_size, dilation=dilation, padding=_pair(padding),
                                 stride=stride)
        sample_dim = kernel_size[0] * kernel_size[1] * in_channels
        self.norm = build_norm_layer(norm_cfg, sample_dim)[1] if norm_cfg is not None else None
        self.reduction = nn.Linear(sample_dim, out_channels, bias=bias)

    def forward(self, x, input_size):
        B, L, C = x.shape
        H, W = input_size
        assert L == H * W, 'input feature has wrong size'
        x = x.view(B, H, W, C).permute([0, 3, 1, 2])
        if self.adap_padding:
            x = self.adap_padding(x)
            H, W = x.shape[-2:]
        x = self.sampler(x)
        s = self.sampler
        out_h = (H + 2 * s.padding[0] - s.dilation[0] * (s.kernel_size[0] - 1) - 1) // s.stride[0] + 1
        out_w = (W + 2 * s.padding[1] - s.dilation[1] * (s.kernel_size[1] - 1) - 1) // s.stride[1] + 1
        x = x.transpose(1, 2)
        x = self.norm(x) if self.norm else x
        return self.reduction(x), (out_h, out_w)


class WindowMSA(BaseModule):

    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True, qk_scale=None,
                 attn_drop_rate=0., proj_drop_rate=0., init_cfg=None):
        super().__init__(init_cfg)
        self.embed_dims = embed_dims
        self.window_size = window_size
        self.num_heads = num_heads
        self.scale = qk_scale or (embed_dims // num_heads)**-0.5
        Wh, Ww = window_size
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        rel = self.double_step_seq(2 * Ww - 1, Wh, 1, Ww)
        rel = (rel + rel.T).flip(1).contiguous()
        self.register_buffer('relative_position_index', rel)
        self.qkv = nn.Linear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.proj = nn.Linear(embed_dims, embed_dims)

    def init_weights(self):
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x, mask=None):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q * self.scale) @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(
            N, N, -1).permute(2, 0, 1).contiguous()
        attn = attn + bias.unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = attn.view(B // nW, nW, self.num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.num_heads, N, N)
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)

    @staticmethod
    def double_step_seq(step1, len1, step2, len2):
        seq1 = torch.arange(0, step1 * len1, step1)
        seq2 = torch.arange(0, step2 * len2, step2)
        return (seq1[:, None] + seq2[None, :]).reshape(1, -1)


class ShiftWindowMSA(BaseModule):

    def __init__(self, embed_dims, num_heads, window_size, shift_size=0, qkv_bias=True,
                 qk_scale=None, attn_drop_rate=0, proj_drop_rate=0, dropout_layer=None,
                 init_cfg=None):
        super().__init__(init_cfg)
        self.window_size = window_size
        self.shift_size = shift_size
        assert 0 <= shift_size < window_size
        self.w_msa = WindowMSA(embed_dims, num_heads, _pair(window_size), qkv_bias, qk_scale,
                               attn_drop_rate, proj_drop_rate)
        self._mask_cache = {}

    def _attn_mask(self, H_pad, W_pad, device):
        key = (H_pad, W_pad, str(device))
        if key not in self._mask_cache:
            ws, ss = self.window_size, self.shift_size
            img_mask = torch.zeros((1, H_pad, W_pad, 1), device=device)
            cnt = 0
            for h in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                for w in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mw = self.window_partition(img_mask).view(-1, ws * ws)
            am = mw.unsqueeze(1) - mw.unsqueeze(2)
            am = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
            if len(self._mask_cache) > 8:
                self._mask_cache.clear()
            self._mask_cache[key] = am
        return self._mask_cache[key]

    def forward(self, query, hw_shape):
        B, L, C = query.shape
        H, W = hw_shape
        assert L == H * W, 'input feature has wrong size'
        ws = self.window_size
        query = query.view(B, H, W, C)
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        query = F.pad(query, (0, 0, 0, pad_r, 0, pad_b))
        H_pad, W_pad = query.shape[1], query.shape[2]
        if self.shift_size > 0:
            shifted = torch.roll(query, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
            attn_mask = self._attn_mask(H_pad, W_pad, query.device)
        else:
            shifted, attn_mask = query, None
        windows = self.window_partition(shifted).view(-1, ws * ws, C)
        attn_windows = self.w_msa(windows, mask=attn_mask).view(-1, ws, ws, C)
        x = self.window_reverse(attn_windows, H_pad, W_pad)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        if pad_r > 0 or pad_b:
            x = x[:, :H, :W, :].contiguous()
        return x.view(B, H * W, C)

    def window_reverse(self, windows, H, W):
        ws = self.window_size
        B = int(windows.shape[0] / (H * W / ws / ws))
        x = windows.view(B, H // ws, W // ws, ws, ws, -1)
        return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)

    def window_partition(self, x):
        B, H, W, C = x.shape
        ws = self.window_size
        x = x.view(B, H // ws, ws, W // ws, ws, C)
        return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


class SwinBlock(BaseModule):

    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size=7, shift=False,
                 qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN'), with_cp=False,
                 init_cfg=None):
        super().__init__(init_cfg)
        self.norm1 = build_norm_layer(norm_cfg, embed_dims)[1]
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size,
                                   window_size // 2 if shift else 0, qkv_bias, qk_scale,
                                   attn_drop_rate, drop_rate)
        self.norm2 = build_norm_layer(norm_cfg, embed_dims)[1]
        self.ffn = FFN(embed_dims=embed_dims, feedforward_channels=feedforward_channels,
                       num_fcs=2, ffn_drop=drop_rate, act_cfg=act_cfg, add_identity=True)

    def forward(self, x, hw_shape):
        identity = x
        x = self.attn(self.norm1(x), hw_shape) + identity
        identity = x
        return self.ffn(self.norm2(x), identity=identity)


class SwinBlockSequence(BaseModule):

    def __init__(self, embed_dims, num_heads, feedforward_channels, depth, window_size=7,
                 qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., downsample=None, act_cfg=dict(type='GELU'),
                 norm_cfg=dict(type='LN'), with_cp=False, init_cfg=None):
        super().__init__(init_cfg)
        self.blocks = nn.ModuleList([
            SwinBlock(embed_dims, num_heads, feedforward_channels, window_size,
                      shift=(i % 2 == 1), qkv_bias=qkv_bias, qk_scale=qk_scale,
                      drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, act_cfg=act_cfg,
                      norm_cfg=norm_cfg) for i in range(depth)])
        self.downsample = downsample

    def forward(self, x, hw_shape):
        for block in self.blocks:
            x = block(x, hw_shape)
        if self.downsample:
            x_down, down_hw = self.downsample(x, hw_shape)
            return x_down, down_hw, x, hw_shape
        return x, hw_shape, x, hw_shape


@MMDET_MODELS.register_module()
class SwinTransformer(BaseModule):

    def __init__(self, num_frames=None, pretrain_img_size=224, in_channels=3, embed_dims=96,
                 patch_size=4, window_size=7, mlp_ratio=4, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.1, use_abs_pos_embed=False, act_cfg=dict(type='GELU'),
                 norm_cfg=dict(type='LN'), with_cp=False, pretrained=None, convert_weights=False,
                 frozen_stages=-1, init_cfg=None):
        super().__init__(init_cfg)
        self.num_frames = num_frames
        self.out_indices = out_indices
        self.use_abs_pos_embed = use_abs_pos_embed
        assert strides[0] == patch_size, 'Use non-overlapping patch embed.'
        self.patch_embed = PatchEmbed(in_channels, embed_dims, 'Conv2d', patch_size, strides[0],
                                      norm_cfg=norm_cfg if patch_norm else None)
        if use_abs_pos_embed:
            p = _pair(pretrain_img_size)
            self.absolute_pos_embed = nn.Parameter(
                torch.zeros((1, (p[0] // patch_size) * (p[1] // patch_size), embed_dims)))
        self.stages = nn.ModuleList()
        ic = embed_dims
        for i in range(len(depths)):
            downsample = PatchMerging(ic, 2 * ic, stride=strides[i + 1],
                                      norm_cfg=norm_cfg if patch_norm else None) \
                if i < len(depths) - 1 else None
            self.stages.append(SwinBlockSequence(
                ic, num_heads[i], mlp_ratio * ic, depths[i], window_size, qkv_bias, qk_scale,
                drop_rate, attn_drop_rate, downsample=downsample, act_cfg=act_cfg,
                norm_cfg=norm_cfg))
            if downsample:
                ic = downsample.out_channels
        self.num_features = [int(embed_dims * 2**i) for i in range(len(depths))]
        for i in out_indices:
            self.add_module(f'norm{i}', build_norm_layer(norm_cfg, self.num_features[i])[1])

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
            elif isinstance(m, WindowMSA):
                m.init_weights()
        self._is_init = True

    # ---- the device path: this package's kernels -------------------------------------------------
    def _device_ok(self, x):
        from .bricks import fused_mode
        return (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and fused_mode()
                and not self.use_abs_pos_embed and self.patch_embed.adap_padding is not None
                and self.patch_embed.projection.kernel_size == self.patch_embed.projection.stride
                and all(blk.attn.window_size == 7 and blk.attn.w_msa.embed_dims == 32 * blk.attn.w_msa.num_heads
                        and isinstance(blk.ffn.activate, nn.GELU) and blk.ffn.num_fcs == 2
                        for st in self.stages for blk in st.blocks))

    @staticmethod
    def _ln(rows, norm):
        from . import ops
        return ops.bias_add_layernorm(rows, None, None, norm.weight, norm.bias, norm.eps)

    @staticmethod
    def _lin(rows, weight, bias=None, act=False, residual=None):
        """Linear on rows through the split GEMM (planes zero-padded to N % 64 == 0), bias / GELU / identity in
        its epilogue."""
        from . import ops
        from .bricks import _split_cached
        wp = _split_cached(weight, 'gemm_pad', lambda planes: ops.split_weight_bf16x3(
            weight.detach().contiguous(), planes, pad=True))
        return ops.gemm_bf16x3(rows, wp, bias, residual, relu=act, n_out=weight.shape[0],
                               out=residual if residual is not None else None)

    def _block_device(self, blk, x, B, H, W):
        """SwinBlock.forward (swin.py:347-365) on token rows x [B*H*W, C]; x is overwritten (it is a temporary)."""
        from . import ops
        from .bricks import SourceKey
        msa = blk.attn.w_msa
        y = self._ln(x, blk.norm1)
        qkv = self._lin(y, msa.qkv.weight, msa.qkv.bias)
        hit = msa.__dict__.get('_pave_bias_t')
        key = SourceKey((msa.relative_position_bias_table,))
        if hit is None or hit[0] != key:
            n = msa.window_size[0] * msa.window_size[1]
            with torch.no_grad():      # [heads, key j, query i]: bias[i, j] of the reference, transposed per head
                bt = msa.relative_position_bias_table[msa.relative_position_index.view(-1)].view(n, n, -1)
                bt = bt.permute(2, 1, 0).contiguous()
            hit = msa.__dict__['_pave_bias_t'] = (key, bt)
        pad = msa.qkv.bias if msa.qkv.bias is not None else torch.zeros(qkv.shape[1], device=x.device)
        a = ops.swin_window_attn(qkv.view(B, H, W, -1), hit[1], pad.detach(), msa.num_heads, blk.attn.window_size,
                                 blk.attn.shift_size, msa.scale)
        x = self._lin(a.view(-1, a.shape[-1]), msa.proj.weight, msa.proj.bias, residual=x)       # + identity
        y = self._ln(x, blk.norm2)
        fc1, fc2 = blk.ffn.layers[0][0], blk.ffn.layers[1]
        h = self._lin(y, fc1.weight, fc1.bias, act='gelu')
        return self._lin(h, fc2.weight, fc2.bias, residual=x)                                    # + identity

    def _merge_device(self, pm, x, B, H, W):
        """PatchMerging.forward (mmdet/models/utils/transformer.py:336-387): 2 x 2 neighbourhoods in nn.Unfold's
        channel order (c, ky, kx), LayerNorm, the bias-free reduction Linear."""
        C = x.shape[1]
        m = x.view(B, H, W, C)
        if H % 2 or W % 2:                                   # 'corner' padding to even sizes
            m = F.pad(m, (0, 0, 0, W % 2, 0, H % 2))
        Ho, Wo = m.shape[1] // 2, m.shape[2] // 2
        rows = m.view(B, Ho, 2, Wo, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B * Ho * Wo, 4 * C)
        if pm.norm is not None:
            rows = self._ln(rows, pm.norm)
        return self._lin(rows, pm.reduction.weight, pm.reduction.bias), Ho, Wo

    def _forward_device(self, x):
        pe = self.patch_embed
        x = pe.adap_padding(x)
        B, Cin, Hi, Wi = x.shape
        p = pe.projection.kernel_size[0]
        H, W = Hi // p, Wi // p
        K = Cin * p * p
        Kp = max(64, (K + 31) // 32 * 32)
        # the patch embedding as a row GEMM: one row per p x p patch in the convolution's (c, ky, kx) order,
        # zero-padded to the kernels' K % 32 == 0
        rows = x.new_zeros((B * H * W, Kp))
        rows[:, :K].view(B, H, W, Cin, p, p).copy_(x.view(B, Cin, H, p, W, p).permute(0, 2, 4, 1, 3, 5))
        w = pe.projection.weight
        hit = pe.__dict__.get('_pave_w')
        from .bricks import SourceKey
        key = SourceKey((w,))
        if hit is None or hit[0] != key:
            with torch.no_grad():
                wk = w.new_zeros((w.shape[0], Kp))
                wk[:, :K] = w.flatten(1)
            hit = pe.__dict__['_pave_w'] = (key, wk)
        t = self._lin(rows, hit[1], pe.projection.bias)
        if pe.norm is not None:
            t = self._ln(t, pe.norm)
        outs = []
        for i, stage in enumerate(self.stages):
            for blk in stage.blocks:
                t = self._block_device(blk, t, B, H, W)
            if i in self.out_indices:
                o = self._ln(t, getattr(self, f'norm{i}'))
                outs.append(o.view(B, H, W, -1).permute(0, 3, 1, 2))     # channels_last [B, C, H, W]
            if stage.downsample:
                t, H, W = self._merge_device(stage.downsample, t, B, H, W)
        return tuple(outs)

    def forward(self, x):
        if x.dim() == 5:  # swin.py:747-749 (`num_frames` flatten)
            x = x.flatten(0, 1)
        if self._device_ok(x):
            return self._forward_device(x)
        x, hw_shape = self.patch_embed(x)
        if self.use_abs_pos_embed:
            x = x + self.absolute_pos_embed
        outs = []
        for i, stage in enumerate(self.stages):
            x, hw_shape, out, out_hw = stage(x, hw_shape)
            if i in self.out_indices:
                out = getattr(self, f'norm{i}')(out)
                outs.append(out.view(-1, *out_hw, self.num_features[i]).permute(0, 3, 1, 2).contiguous())
        return tuple(outs)
