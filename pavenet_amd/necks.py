"""``mmdet.ChannelMapper`` (a14), restated from
third_party/mmdetection/mmdet/models/necks/channel_mapper.py:50-100."""

import torch
import torch.nn as nn

from .bricks import BaseModule, ConvModule
from .registry import MMDET_MODELS


@MMDET_MODELS.register_module()
class ChannelMapper(BaseModule):

    def __init__(self, in_channels, out_channels, kernel_size=3, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), num_outs=None,
                 init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform')):
        super().__init__(init_cfg)
        assert isinstance(in_channels, (list, tuple))
        # device eval: levels are written into one flattened buffer (attribute = A/B switch)
        self.flat_output = True
        self.extra_convs = None
        if num_outs is None:
            num_outs = len(in_channels)
        self.convs = nn.ModuleList()
        for in_channel in in_channels:
            self.convs.append(ConvModule(in_channel, out_channels, kernel_size,
                                         padding=(kernel_size - 1) // 2, conv_cfg=conv_cfg,
                                         norm_cfg=norm_cfg, act_cfg=act_cfg))
        if num_outs > len(in_channels):
            self.extra_convs = nn.ModuleList()
            for i in range(len(in_channels), num_outs):
                in_channel = in_channels[-1] if i == len(in_channels) else out_channels
                self.extra_convs.append(ConvModule(in_channel, out_channels, 3, stride=2, padding=1,
                                                   conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                                   act_cfg=act_cfg))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self._is_init = True

    def _flat_ok(self, inputs):
        mods = list(self.convs) + list(self.extra_convs or [])
        # the flat path writes through raw pointers / `out=` (no grad_fn): inference only.  With
        # grad mode on (frozen-neck fine-tuning, saliency) the differentiable ConvModule path runs.
        return (not self.training and not torch.is_grad_enabled()
                and all(x.is_cuda and x.dtype == torch.float32 for x in inputs)
                and (self.extra_convs is None or len(self.extra_convs) == 1)
                and all(m.with_norm and not m.with_activation and m.conv.bias is None
                        and isinstance(getattr(m, m.norm_name), nn.GroupNorm) for m in mods))

    @staticmethod
    def _conv_split(conv, x):
        """1x1 convolution as a row GEMM on the NHWC map / 3x3 pad-1 convolution as an implicit
        GEMM through the hand-written split-operand kernel, when the GEMM mode and the shape take
        it (bricks.split_gemm_ok / split_conv_weight); None otherwise (library convolution)."""
        from . import ops
        from .bricks import (fused_mode, get_gemm_mode, linear_rows, small_split_ok, split_conv_weight,
                             split_gemm_ok)
        if conv.groups != 1 or conv.dilation != (1, 1) or conv.bias is not None \
                or torch.is_grad_enabled():
            return None
        n, c, h, w = x.shape
        if conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0):
            xc = x.contiguous(memory_format=torch.channels_last)
            rows = xc.permute(0, 2, 3, 1).reshape(-1, c)
            w2 = conv.weight.flatten(1)
            if not (split_gemm_ok(rows, w2) or small_split_ok(rows, w2)):    # (HRNet's K = 96 level:
                return None                                                 #  zero-padded planes)
            y = linear_rows(rows, w2)
            return y.view(n, h, w, -1).permute(0, 3, 1, 2)
        if conv.kernel_size == (3, 3) and conv.padding == (1, 1) and \
                conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2):
            s_ = conv.stride[0]
            if n * ((h - 1) // s_ + 1) * ((w - 1) // s_ + 1) < 16384 and \
                    (not fused_mode() or 9 * c < 2048):
                # too few 128-row tiles to fill 256 CUs: the library's split-K wins (the 3-plane / fp16
                # kernel has its own split-K form from K = 2048 on: ops.conv3x3_split)
                return None
            wsplit = split_conv_weight(conv.weight)
            if wsplit is None:
                return None
            xc = x.contiguous(memory_format=torch.channels_last)
            return ops.conv3x3_split(xc, wsplit, None, stride=conv.stride[0], relu=False,
                                     fp16=get_gemm_mode() == 'fp16')
        return None

    def _forward_flat(self, inputs):
        """Same values as `forward`, but every level's GroupNorm writes straight into ONE
        [n, sum(h*w), C] buffer -- the transformer's flattened multi-level feature
        (OT:21312-21331) -- and the returned maps are views of it: no concatenation copy."""
        mods = list(self.convs) + list(self.extra_convs or [])
        xs = list(inputs) + ([inputs[-1]] if self.extra_convs else [])
        ys = []
        for m, x in zip(mods, xs):
            y = m._conv_deterministic(x) if m.deterministic else None
            if y is None:
                y = self._conv_split(m.conv, x)     # split / 16-bit GEMM modes: in-tree kernels
            y = m.conv(x) if y is None else y
            ys.append(y.contiguous(memory_format=torch.channels_last))
        n, C = ys[0].shape[0], ys[0].shape[1]
        S = sum(y.shape[2] * y.shape[3] for y in ys)
        buf = torch.empty((n, S, C), dtype=torch.float32, device=ys[0].device)
        outs, st = [], 0
        gns = [getattr(m, m.norm_name) for m in mods]
        G = gns[0].num_groups
        # (the kernel's own shape conditions, pave_groupnorm_nhwc_f32: anything else -> torch below)
        hip_ok = [C % 4 == 0 and C <= 1024 and C % gn.num_groups == 0 and (C // gn.num_groups) % 4 == 0
                  and gn.num_groups <= 256 and 256 % (C // 4) == 0 and gn.affine for gn in gns]
        if all(hip_ok) and all(gn.num_groups == G for gn in gns):
            # every level in the same three launches (statistics, scale / shift, apply)
            from . import ops
            levels = []
            for gn, y in zip(gns, ys):
                h, w = y.shape[2:]
                rows = y.permute(0, 2, 3, 1).reshape(n, h * w, C)      # view of the NHWC storage
                dst = buf[:, st:st + h * w]
                levels.append((rows, gn.weight, gn.bias, gn.eps, dst))
                outs.append(dst.view(n, h, w, C).permute(0, 3, 1, 2))
                st += h * w
            ops.groupnorm_levels_into(levels, G)
            return tuple(outs)
        for m, y, gn, ok in zip(mods, ys, gns, hip_ok):
            G = gn.num_groups
            h, w = y.shape[2:]
            rows = y.permute(0, 2, 3, 1).reshape(n, h * w, C)          # view of the NHWC storage
            dst = buf[:, st:st + h * w]
            if ok:
                from . import ops
                ops.groupnorm_nhwc_into(rows, gn.weight, gn.bias, G, gn.eps, dst)   # HIP, 3 launches
                outs.append(dst.view(n, h, w, C).permute(0, 3, 1, 2))
                st += h * w
                continue
            var, mean = torch.var_mean(rows.view(n, h * w, G, C // G), dim=(1, 3), unbiased=False)
            rstd = torch.rsqrt(var + gn.eps)[:, :, None]                # [n, G, 1]
            a = rstd * gn.weight.view(1, G, -1)
            b = gn.bias.view(1, G, -1) - mean[:, :, None] * a
            torch.addcmul(b.reshape(n, 1, C), rows, a.reshape(n, 1, C), out=dst)
            outs.append(dst.view(n, h, w, C).permute(0, 3, 1, 2))
            st += h * w
        return tuple(outs)

    def forward(self, inputs):
        assert len(inputs) == len(self.convs)
        if self.flat_output and self._flat_ok(inputs):
            return self._forward_flat(inputs)
        outs = [self.convs[i](inputs[i]) for i in range(len(inputs))]
        if self.extra_convs:
            for i in range(len(self.extra_convs)):
                outs.append(self.extra_convs[i](inputs[-1] if i == 0 else outs[-1]))
        return tuple(outs)
