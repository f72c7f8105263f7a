"""``mmdet.ResNet`` incl. the fork's ``input_type='mul_frames'`` (a14), restated from
third_party/mmdetection/mmdet/models/backbones/resnet.py (Bottleneck :99-305, ResNet :308-672,
forward :632-654).  Same ctor kwargs (the ones the PAVE-Net configs use) and state-dict keys.

Dense convolutions are true contractions: in the default GEMM mode every one of them (7x7 stem, 1x1, 3x3, the
strided downsample, the 64-channel stage as one chained launch per Bottleneck) is a split-operand MFMA launch of this
package (ops.conv7x7s2_nchw_split / gemm_bf16x3 / conv3x3_split / conv1x1_strided_split / bottleneck_chain); MIOpen /
hipBLASLt through PyTorch-ROCm serve the `'native'` mode, CPU tensors and autograd.
Inference-time: frozen BatchNorm is folded into the preceding convolution (cached; the
reference offers the same transformation as ``--fuse-conv-bn``, tools/test.py:227-228), and
the whole trunk runs channels-last.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .bricks import BaseModule, build_norm_layer
from .registry import MMDET_MODELS


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch',
                 norm_cfg=dict(type='BN')):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.pairs = (('conv1', 'bn1', True), ('conv2', 'bn2', False))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch',
                 norm_cfg=dict(type='BN')):
        super().__init__()
        assert style in ['pytorch', 'caffe']
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, s1, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.conv2 = nn.Conv2d(planes, planes, 3, s2, dilation, dilation, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = build_norm_layer(norm_cfg, planes * 4)[1]
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.pairs = (('conv1', 'bn1', True), ('conv2', 'bn2', True), ('conv3', 'bn3', False))


def _fold(conv, bn):
    """conv (no bias) followed by eval-mode BN -> (weight, bias) of one convolution."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = conv.weight * scale.view(-1, 1, 1, 1)
    b = bn.bias - bn.running_mean * scale
    if conv.bias is not None:
        b = b + conv.bias * scale
    return w, b


@MMDET_MODELS.register_module()
class ResNet(BaseModule):
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)),
                     50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)),
                     152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3),
                 style='pytorch', deep_stem=False, avg_down=False, frozen_stages=-1,
                 conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True,
                 dcn=None, stage_with_dcn=(False, False, False, False), plugins=None,
                 with_cp=False, zero_init_residual=True, pretrained=None, init_cfg=None,
                 input_type='single_frame'):
        super().__init__(init_cfg)
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        assert not deep_stem and not avg_down and dcn is None and plugins is None, \
            'pavenet_amd.ResNet builds the plain stem / no DCN / no plugins used by PAVE-Net'
        self.depth = depth
        self.input_type = input_type
        stem_channels = stem_channels or base_channels
        self.num_stages = num_stages
        self.out_indices = out_indices
        self.frozen_stages = frozen_stages
        self.norm_eval = norm_eval
        self.style = style
        block, stage_blocks = self.arch_settings[depth]
        stage_blocks = stage_blocks[:num_stages]
        self.conv1 = nn.Conv2d(in_channels, stem_channels, 7, 2, 3, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, stem_channels)[1]
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        inplanes = stem_channels
        for i, nb in enumerate(stage_blocks):
            planes = base_channels * 2**i
            layers = []
            for b in range(nb):
                stride = strides[i] if b == 0 else 1
                downsample = None
                if b == 0 and (stride != 1 or inplanes != planes * block.expansion):
                    downsample = nn.Sequential(
                        nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                        build_norm_layer(norm_cfg, planes * block.expansion)[1])
                layers.append(block(inplanes, planes, stride, dilations[i], downsample, style,
                                    norm_cfg))
                inplanes = planes * block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self.feat_dim = inplanes
        self._folded = None
        self.channels_last = True
        # Bottleneck tails (conv3 + bn3 + identity + ReLU) with K = planes up to this run as ONE
        # hand-written MFMA kernel; above it hipBLASLt + a bias/ReLU pass is faster (measured,
        # tools/bench_gemm_shapes.py: K=64 1.14 vs 1.84 ms, 128: 0.81 vs 1.04, 256: 0.68 vs 0.73,
        # 512: 0.64 vs 0.58)
        self.fused_tail_max_k = 256
        self.fused_ds_max_k = 128     # conv3 + downsample in one kernel up to this K1 + K2 (layer1.0)
        # 64-channel stage (layer1) in the exact split mode: one launch per Bottleneck from its 3x3
        # on, chained with the next block's conv1 (ops.bottleneck_chain); attribute = A/B switch
        self.chain_stage64 = True
        # ... with the 3x3 as a launch of its own (three blocks per CU) and the chain from conv3 on
        # ('tail'), or the 3x3 as the chain's first phase ('full'); measured: see DESIGN.md 4.2
        self.chain_mode = 'full'
        # 3x3 convolutions: MIOpen's searched fp32 kernels are 10-30 % faster than the
        # hand-written MFMA implicit GEMM (tools/bench_conv.py) but NOT run-to-run deterministic
        # (tools/debug_determinism.py); True routes them through pave_conv3x3_nhwc_f32
        # (bit-reproducible, bn2 + ReLU in its epilogue).
        self.deterministic_conv3x3 = False

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():
            if isinstance(m, Bottleneck):
                nn.init.constant_(m.bn3.weight, 0)
            elif isinstance(m, BasicBlock):
                nn.init.constant_(m.bn2.weight, 0)
        self._is_init = True

    # -- folded inference path ----------------------------------------------
    def _params_key(self):
        from .bricks import SourceKey, module_tensors
        return SourceKey(module_tensors(self))

    def _stem_folded(self):
        """The folded stem (weight, bias), validated on its own five tensors: a step's first kernel
        launch does not wait for the walk over the whole backbone (_build_folded runs after it)."""
        from .bricks import SourceKey
        bn = self.bn1
        key = SourceKey([self.conv1.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var])
        hit = self.__dict__.get('_stem_cache')
        if hit is None or hit[0] != key:
            mf = torch.channels_last if self.channels_last else torch.contiguous_format
            with torch.no_grad():
                hit = (key, tuple(t.contiguous(memory_format=mf) if t.dim() == 4 else t
                                  for t in _fold(self.conv1, self.bn1)))
            self.__dict__['_stem_cache'] = hit
        return hit[1]

    def _build_folded(self):
        key = self._params_key()
        if self._folded is not None and self._folded[0] == key:
            return self._folded[1]
        mf = torch.channels_last if self.channels_last else torch.contiguous_format
        with torch.no_grad():
            f = {'stem': tuple(t.contiguous(memory_format=mf) if t.dim() == 4 else t
                               for t in _fold(self.conv1, self.bn1))}
            for name in self.res_layers:
                for bi, blk in enumerate(getattr(self, name)):
                    for cn, bn, _ in blk.pairs:
                        w, b = _fold(getattr(blk, cn), getattr(blk, bn))
                        f[(name, bi, cn)] = (w.contiguous(memory_format=mf), b)
                        if cn == 'conv2' and w.shape[2:] == (3, 3) and w.shape[1] % 32 == 0 \
                                and w.shape[0] % 64 == 0:
                            f[(name, bi, 'conv2_taps')] = w.permute(2, 3, 1, 0).contiguous()
                        if cn == 'conv3' and w.shape[1] <= self.fused_tail_max_k \
                                and w.shape[1] % 32 == 0:
                            # [K, N] operand of pave_rows_gemm_bias_res_act_f32
                            f[(name, bi, 'conv3_kn')] = w.flatten(1).t().contiguous()
                    if blk.downsample is not None:
                        w, b = _fold(blk.downsample[0], blk.downsample[1])
                        f[(name, bi, 'ds')] = (w.contiguous(memory_format=mf), b)
                        w3_kn = f.get((name, bi, 'conv3_kn'))
                        if w3_kn is not None and w.shape[1] % 32 == 0 and \
                                w3_kn.shape[0] + w.shape[1] <= self.fused_ds_max_k:
                            # conv3 and the downsample convolution share one accumulator:
                            # [y | x] @ [W3; Wd] + (b3 + bd)
                            f[(name, bi, 'tail_ds_kn')] = (
                                torch.cat([w3_kn, w.flatten(1).t()], 0).contiguous(),
                                f[(name, bi, 'conv3')][1] + b)
        self._folded = (key, f)
        return f

    @staticmethod
    def _stem_split(x_nchw, w):
        """7x7 / stride 2 stem through the hand-written 3-plane split kernel, read straight from
        the NCHW image batch (bf16x3 GEMM mode); None when the mode / shape does not take it."""
        from . import ops
        from .bricks import _split_cached, fused_mode, get_gemm_mode
        if not fused_mode() or torch.is_grad_enabled() or not x_nchw.is_cuda \
                or x_nchw.dtype != torch.float32 or tuple(w.shape) != (64, 3, 7, 7) \
                or x_nchw.shape[1] != 3 or x_nchw.shape[3] < 8 or not x_nchw.is_contiguous():
            return None
        wp = _split_cached(w, 'stem7x7', lambda planes: ops.split_stem7x7_weight(w.detach(), planes))
        if x_nchw.shape[3] % 4 or x_nchw.data_ptr() % 16:
            # a width off the 4-pixel grid (the reference's PoseTrack pipeline pads with size_divisor = 1: 750 x 1333
            # frames) or an unaligned view: the rows are re-laid once at a 16-byte aligned pitch with zero pad
            # columns -- the convolution's own right-hand padding -- and the LDS-window kernel reads that (one copy
            # of the 3-channel image instead of the per-lane window-load kernel: the stem is 60 x the image's FLOPs)
            return ops.conv7x7s2_nchw_split(ops.repitch_rows(x_nchw), wp, valid_w=x_nchw.shape[3])
        return ops.conv7x7s2_nchw_split(x_nchw, wp)

    @staticmethod
    def _ds_bias(f, name, bi):
        """b3 + bd of a downsample block, summed once per folded parameter set."""
        key = (name, bi, 'ds_b3')
        if key not in f:
            f[key] = f[(name, bi, 'ds')][1] + f[(name, bi, 'conv3')][1]
        return f[key]

    @staticmethod
    def _as_rows(x):
        """channels_last [N, C, H, W] -> ([N*H*W, C] view, (N, H, W))."""
        n, c, h, w = x.shape
        return x.permute(0, 2, 3, 1).reshape(-1, c), (n, h, w)

    @staticmethod
    def _as_map(rows, nhw):
        n, h, w = nhw
        return rows.view(n, h, w, rows.shape[-1]).permute(0, 3, 1, 2)  # channels_last 4-D

    def _stage64_chain_ok(self, name, x, f):
        from .bricks import _GEMM, fused_mode
        blocks = list(getattr(self, name))
        if blocks[0].downsample is not None and (name, 0, 'tail_ds_kn') not in f:
            return False
        if not (self.chain_stage64 and fused_mode() and not torch.is_grad_enabled()
                and x.is_contiguous(memory_format=torch.channels_last)
                and x.shape[0] * x.shape[2] * x.shape[3] >= _GEMM['min_rows']
                and all(isinstance(b, Bottleneck) for b in blocks)):
            return False
        for bi, blk in enumerate(blocks):
            c1, c2, c3 = blk.conv1, blk.conv2, blk.conv3
            if not (c1.out_channels == 64 and c1.stride == (1, 1)
                    and tuple(c2.weight.shape) == (64, 64, 3, 3) and c2.stride == (1, 1)
                    and c2.dilation == (1, 1) and c2.padding == (1, 1) and c2.groups == 1
                    and tuple(c3.weight.shape[:2]) == (256, 64)):
                return False
            if blk.downsample is not None and not (
                    bi == 0 and blk.downsample[0].stride == (1, 1)
                    and blk.downsample[0].in_channels == 64):
                return False
            if blk.downsample is None and c1.in_channels != 256:
                return False
        # pave_bottleneck_chain_f32's own size guards: larger batches take the per-block path
        pixels = x.shape[0] * x.shape[2] * x.shape[3]
        if pixels >= 2 ** 31 or (self.chain_mode != 'tail' and pixels * 256 >= 2 ** 32 - 65536):
            return False
        return blocks[0].conv1.in_channels == x.shape[1] and x.shape[1] % 64 == 0

    def _stage64_chain(self, name, x, f, next_conv1):
        """The 64-channel stage with ONE launch per Bottleneck from its 3x3 convolution on
        (conv2 -> conv3 + identity | downsample -> ReLU -> the NEXT block's conv1,
        pave_bottleneck_chain_f32); only the first conv1 is a launch of its own.  next_conv1 =
        the folded (weight, bias) of the following stage's first conv1 (64 | 128 outputs) or None.
        Returns (stage output, that conv1's output or None)."""
        from . import ops
        from .bricks import _split_cached, _split_weight, linear_rows, split_conv_weight
        blocks = list(getattr(self, name))
        rows, nhw = self._as_rows(x)
        w1, b1 = f[(name, 0, 'conv1')]
        c1 = self._as_map(linear_rows(rows, w1.flatten(1), b1, relu=True), nhw)
        out = None
        for bi, blk in enumerate(blocks):
            w2, b2 = f[(name, bi, 'conv2')]
            w3, b3 = f[(name, bi, 'conv3')]
            nxt = f[(name, bi + 1, 'conv1')] if bi + 1 < len(blocks) else next_conv1
            wnp = _split_weight(nxt[0].flatten(1)) if nxt is not None else None
            bnx = nxt[1] if nxt is not None else None
            if self.chain_mode == 'tail':
                y = ops.conv3x3_split(c1, split_conv_weight(w2), b2, stride=1, relu=True)
                head = dict(c1=None, w2_planes=None, b2=None, c2=y)
            else:
                head = dict(c1=c1, w2_planes=split_conv_weight(w2), b2=b2)
            if blk.downsample is not None:
                tail = f[(name, bi, 'tail_ds_kn')]     # [W3; Wd] as [K, N], b3 + bd
                w3p = _split_cached(tail[0], 'tail_cat', lambda planes: ops.split_weight_bf16x3(
                    tail[0].t().contiguous(), planes))
                out, c1 = ops.bottleneck_chain(w3_planes=w3p, b3=tail[1], a2=x, w1n_planes=wnp,
                                               b1n=bnx, **head)
            else:
                # blocks after the first overwrite the previous block's output (a temporary)
                res = out if out is not None else x
                out, c1 = ops.bottleneck_chain(w3_planes=_split_weight(w3.flatten(1)), b3=b3,
                                               residual=res, w1n_planes=wnp, b1n=bnx,
                                               out=res if out is not None else None, **head)
        return out, c1

    def _bottleneck_gemm(self, blk, x, f, name, bi, inplace_identity=False, c1=None):
        """Bottleneck on the NHWC map: conv1 as a hipBLASLt row GEMM (bias + ReLU epilogue), the
        3x3 through MIOpen, and the tail `bn2 -> relu -> conv3 -> bn3 -> + identity |
        downsample(x) -> relu` as ONE hand-written MFMA kernel (pave_rows_gemm_bias_res_act_f32)
        where that wins (K <= fused_tail_max_k); above it hipBLASLt GEMMs with the residual in
        the epilogue plus one fused bias+ReLU pass each (pave_bias_act_rows_f32)."""
        from . import ops
        from .bricks import linear_rows, split_conv_weight, split_gemm_ok
        rows, nhw = self._as_rows(x)
        if c1 is not None:
            y = c1                          # computed by the previous stage's last chained launch
        else:
            w1, b1 = f[(name, bi, 'conv1')]
            y = linear_rows(rows, w1.flatten(1), b1, relu=True)           # conv1 + bn1 + relu
            y = self._as_map(y, nhw)
        w2, b2 = f[(name, bi, 'conv2')]
        c2 = blk.conv2
        w3, b3 = f[(name, bi, 'conv3')]
        w3_kn = f.get((name, bi, 'conv3_kn'))
        taps = f.get((name, bi, 'conv2_taps')) if self.deterministic_conv3x3 else None
        wsplit = None
        if c2.dilation[0] == 1 and c2.padding[0] == 1 and c2.stride[0] == c2.stride[1] \
                and not torch.is_grad_enabled():
            wsplit = split_conv_weight(w2)     # split / 16-bit GEMM modes: same kernel family
        from .bricks import fused_mode, small_split_ok
        # (below min_rows too when the 3x3 runs on the split kernel -- its epilogue applies bn2 + relu, so the tail
        # needs no A-side prologue: the same kernels for a one-clip batch as for a large one)
        split3 = blk.downsample is None and (
            split_gemm_ok(rows, w3.flatten(1))
            or (wsplit is not None and fused_mode() and small_split_ok(rows, w3.flatten(1))))
        if split3:
            w3_kn = None  # the split GEMM takes the tail (same prologue / epilogue) for K >= 256
        if wsplit is not None:
            from .bricks import get_gemm_mode
            y = ops.conv3x3_split(y, wsplit, b2, stride=c2.stride[0], relu=True,
                                  fp16=get_gemm_mode() == 'fp16')
            b2 = None
        elif taps is not None and c2.dilation[0] == 1 and c2.padding[0] == 1:
            y = ops.conv3x3_nhwc(y, taps, b2, stride=c2.stride[0], relu=True)  # MFMA, bn2+relu fused
            b2 = None
        else:
            y = F.conv2d(y, w2, None, c2.stride, c2.padding, c2.dilation)  # MIOpen 3x3
            if w3_kn is None and not split3:
                ops.bias_act_rows_(y, b2, None, relu=True)                # bn2 + relu, one pass
                b2 = None
        # else: bn2 + relu are applied by the tail kernel while it loads its A operand
        yrows, onhw = self._as_rows(y)
        if blk.downsample is not None:
            wd, bd = f[(name, bi, 'ds')]
            s = blk.downsample[0].stride[0]
            tail = f.get((name, bi, 'tail_ds_kn'))
            from .bricks import _GEMM, _split_weight, fused_mode, small_split_ok
            # (any number of rows: below min_rows -- layer4 of a one-clip batch -- these were vendor GEMMs until
            # round 5, so a clip's values depended on the batch it came in and a library kernel was on the path)
            w3_split = b2 is None and (split_gemm_ok(yrows, w3.flatten(1)) or small_split_ok(yrows, w3.flatten(1)))
            if (s > 1 and tail is None and fused_mode() and not torch.is_grad_enabled()
                    and wd.shape[0] % 128 == 0 and wd.shape[1] % 64 == 0
                    and (yrows.shape[0] >= _GEMM['min_rows'] or w3_split)
                    and (w3_kn is not None or w3_split or split_gemm_ok(yrows, w3.flatten(1)))):
                # stride-2 downsample: the GEMM reads the strided pixels itself (no slice copy)
                idt = ops.conv1x1_strided_split(x, _split_weight(wd.flatten(1)),
                                                self._ds_bias(f, name, bi), stride=s)
                idt, _ = self._as_rows(idt)
                if w3_split or split_gemm_ok(yrows, w3.flatten(1)):
                    out = linear_rows(yrows, w3.flatten(1), None, relu=True, residual=idt,
                                      inplace_residual=True, a_bias=b2)
                else:
                    out = ops.rows_gemm_bias_res_act(yrows, w3_kn, None, idt, relu=True, out=idt,
                                                     a_bias=b2)
                return self._as_map(out, onhw)
            xs = x if s == 1 else x[:, :, ::s, ::s].contiguous(memory_format=torch.channels_last)
            xrows, _ = self._as_rows(xs)
            from .bricks import _split_cached
            if tail is not None and b2 is None and fused_mode() \
                    and not torch.is_grad_enabled() \
                    and tail[0].shape[0] % 32 == 0 and tail[0].shape[1] % 64 == 0 \
                    and yrows.shape[1] % 16 == 0:
                # relu([y | x] @ [W3 | Wd]^T + b3 + bd) on the split kernel: two A sources, one
                # accumulator, one launch (bn2 + relu were applied by the 3x3 kernel's epilogue)
                wcat = _split_cached(tail[0], 'tail_cat', lambda planes: ops.split_weight_bf16x3(
                    tail[0].t().contiguous(), planes))
                out = ops.gemm_bf16x3_cat(yrows, xrows, wcat, tail[1], None, relu=True)
            elif tail is not None:
                # relu([relu(y + b2) | x] @ [W3; Wd] + b3 + bd): the whole tail in one kernel
                out = ops.rows_gemm_bias_res_act(yrows, tail[0], tail[1], None, relu=True,
                                                 a_bias=b2, a2=xrows)
            elif split_gemm_ok(xrows, wd.flatten(1)) and \
                    (w3_kn is not None or split_gemm_ok(yrows, w3.flatten(1))):
                # split / 16-bit GEMM modes: downsample GEMM (both biases), then conv3 accumulates
                # into it with the ReLU in its epilogue
                idt = linear_rows(xrows, wd.flatten(1), self._ds_bias(f, name, bi))
                if split_gemm_ok(yrows, w3.flatten(1)):
                    out = linear_rows(yrows, w3.flatten(1), None, relu=True, residual=idt,
                                      inplace_residual=True, a_bias=b2)
                else:
                    out = ops.rows_gemm_bias_res_act(yrows, w3_kn, None, idt, relu=True, out=idt,
                                                     a_bias=b2)
            else:
                idt = torch.addmm(bd + b3, xrows, wd.flatten(1).t())      # both biases here
                if w3_kn is not None:                                     # += y @ W3, ReLU: one pass
                    out = ops.rows_gemm_bias_res_act(yrows, w3_kn, None, idt, relu=True, out=idt,
                                                     a_bias=b2)
                else:
                    out = torch.addmm(idt, yrows, w3.flatten(1).t())
                    ops.bias_act_rows_(out, None, None, relu=True)
        elif split3:
            out = linear_rows(yrows, w3.flatten(1), b3, relu=True, residual=rows,
                              inplace_residual=inplace_identity, a_bias=b2)
        elif w3_kn is not None:
            # bn2 + relu + conv3 + bn3 + identity + ReLU as one MFMA kernel; a temporary identity
            # is overwritten in place
            out = ops.rows_gemm_bias_res_act(yrows, w3_kn, b3, rows, relu=True,
                                             out=rows if inplace_identity else None, a_bias=b2)
        elif inplace_identity:
            out = rows.addmm_(yrows, w3.flatten(1).t())                   # identity += y @ W3
            ops.bias_act_rows_(out, b3, None, relu=True)
        else:
            out = torch.addmm(rows, yrows, w3.flatten(1).t())             # + identity in the GEMM
            ops.bias_act_rows_(out, b3, None, relu=True)
        return self._as_map(out, onhw)

    def forward(self, x):
        if self.input_type == 'mul_frames' and x.dim() == 5:
            x = x.flatten(0, 1)  # [B, T, C, H, W] -> [B*T, C, H, W]  (resnet.py:634-639)
        assert not self.training, 'pavenet_amd.ResNet is an inference (frozen BN) backbone'
        gemm_path = (self.channels_last and x.is_cuda and x.dtype == torch.float32
                     and self.style == 'pytorch')
        w, b = self._stem_folded()
        y = self._stem_split(x, w) if gemm_path else None             # reads the NCHW batch as is
        if y is None and self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        if gemm_path:
            from . import ops
            if y is None:
                y = F.conv2d(x, w, None, 2, 3)                       # MIOpen
            # bn1 + relu + maxpool in one pass over the stem map
            x = ops.bias_relu_maxpool_nhwc(y, b)
        else:
            x = self.maxpool(F.relu_(F.conv2d(x, w, b, 2, 3)))
        f = self._build_folded()     # (validated while the stem runs)
        outs = []
        pre_c1 = None       # the coming block's conv1 output, when the previous launch chained it
        for i, name in enumerate(self.res_layers):
            if gemm_path and self._stage64_chain_ok(name, x, f):
                nxt = None
                if i + 1 < len(self.res_layers):
                    nb = getattr(self, self.res_layers[i + 1])[0]
                    if isinstance(nb, Bottleneck) and nb.conv1.stride == (1, 1) and \
                            nb.conv1.in_channels == 256 and nb.conv1.out_channels in (64, 128):
                        nxt = f[(self.res_layers[i + 1], 0, 'conv1')]
                x, pre_c1 = self._stage64_chain(name, x, f, nxt)
                if i in self.out_indices:
                    outs.append(x)
                continue
            for bi, blk in enumerate(getattr(self, name)):
                if gemm_path and isinstance(blk, Bottleneck):
                    # blocks after the first of a stage read a temporary (the previous
                    # block's output, never a stage output): accumulate into it
                    x = self._bottleneck_gemm(blk, x, f, name, bi, inplace_identity=bi > 0,
                                              c1=pre_c1)
                    pre_c1 = None
                    continue
                identity = x
                y = x
                for cn, _, act in blk.pairs:
                    conv = getattr(blk, cn)
                    w, b = f[(name, bi, cn)]
                    y = F.conv2d(y, w, b, conv.stride, conv.padding, conv.dilation)
                    if act:
                        y = F.relu_(y)
                if blk.downsample is not None:
                    w, b = f[(name, bi, 'ds')]
                    identity = F.conv2d(x, w, b, blk.downsample[0].stride)
                x = F.relu_(y.add_(identity))
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


# ---------------------------------------------------------------------------
# HRNet (a14): third_party/mmdetection/mmdet/models/backbones/hrnet.py, incl. the fork's
# `return y_list[1:]` (:583).  Same ctor kwargs (`extra`) and state-dict keys.
# ---------------------------------------------------------------------------
class _Folded:
    """Device inference path shared by the HRNet pieces: eval BatchNorm folded into its
    convolution (cached until a parameter changes), channels-last maps, MIOpen convolution
    without bias + ONE hand-written pass for bias / residual / ReLU (pave_bias_act_rows_f32)."""
    @classmethod
    def weights(cls, conv, bn):
        # the folded pair lives ON the conv module (dies with it; Python reuses id()s, so a
        # class-level dict keyed on id(conv) could serve another model's weights)
        from .bricks import _CACHE_EPOCH
        key = tuple((id(t), t.data_ptr(), t._version) for t in
                    (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)) + \
            (_CACHE_EPOCH[0],)
        hit = conv.__dict__.get('_pave_folded')
        if hit is None or hit[0] != key:
            with torch.no_grad():
                w, b = _fold(conv, bn)
                hit = (key, w.contiguous(memory_format=torch.channels_last), b.contiguous())
            conv.__dict__['_pave_folded'] = hit
        return hit[1], hit[2]

    @classmethod
    def conv_bn(cls, x, conv, bn, relu=False, residual=None):
        """act(bn(conv(x)) + residual); x, residual channels_last; the result is a fresh tensor."""
        from . import ops
        from .bricks import fused_mode, get_gemm_mode, split_conv_weight
        w, b = cls.weights(conv, bn)
        if conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.dilation == (1, 1) \
                and conv.groups == 1 and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2):
            wsplit = split_conv_weight(w)   # split / 16-bit GEMM modes
            if wsplit is not None and fused_mode():
                # 3-plane / fp16 kernel: any Cin % 16, Cout % 4 (zero-padded planes); bias, identity
                # and ReLU in its epilogue (no separate pass)
                return ops.conv3x3_split(x, wsplit, b, stride=conv.stride[0], relu=relu,
                                         residual=residual, cout=w.shape[0])
            if wsplit is not None:
                y = ops.conv3x3_split(x, wsplit, b, stride=conv.stride[0],
                                      relu=relu and residual is None,
                                      fp16=get_gemm_mode() == 'fp16')
                return y if residual is None else ops.bias_act_rows_(y, None, residual, relu=relu)
        if conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) \
                and conv.groups == 1 and fused_mode() and w.shape[1] % 32 == 0 \
                and w.shape[1] >= 64 and w.shape[0] % 4 == 0 \
                and x.is_contiguous(memory_format=torch.channels_last):
            # 1x1 convolution = row GEMM on the NHWC map through the exact 3-plane kernel (planes
            # zero-padded to Cout % 64 == 0), bias / identity / ReLU in its epilogue
            from .bricks import _split_cached
            n, cin, h, wd = x.shape
            wp = _split_cached(w, 'gemm_pad', lambda planes: ops.split_weight_bf16x3(
                w.detach().flatten(1).contiguous(), planes, pad=True))
            rows = x.permute(0, 2, 3, 1).reshape(-1, cin)
            res = residual.permute(0, 2, 3, 1).reshape(-1, w.shape[0]) if residual is not None else None
            out = ops.gemm_bf16x3(rows, wp, b, res, relu=relu, n_out=w.shape[0])
            return out.view(n, h, wd, -1).permute(0, 3, 1, 2)
        y = F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        return ops.bias_act_rows_(y, b, residual, relu=relu)

    @staticmethod
    def ok(x):
        return x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()

    @classmethod
    def seq(cls, mods, x):
        """nn.Sequential of [Conv2d, BN, (ReLU)] (+ Upsample) groups (transitions, fuse layers)."""
        mods = list(mods)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Sequential):
                x = cls.seq(m, x)
                i += 1
            elif isinstance(m, nn.Conv2d):
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                x = cls.conv_bn(x, m, mods[i + 1], relu=relu)
                i += 3 if relu else 2
            elif isinstance(m, nn.Upsample):
                x = F.interpolate(x, scale_factor=m.scale_factor, mode=m.mode)
                x = x.contiguous(memory_format=torch.channels_last)
                i += 1
            else:
                raise TypeError(f'unexpected module in a folded sequence: {type(m).__name__}')
        return x


class _Block(nn.Module):
    """Runs a BasicBlock / Bottleneck parameter container (module forward, eval BN)."""

    @staticmethod
    def run_folded(blk, x):
        identity = x
        if blk.downsample is not None:
            identity = _Folded.conv_bn(x, blk.downsample[0], blk.downsample[1])
        y = x
        n = len(blk.pairs)
        for k, (cn, bn, act) in enumerate(blk.pairs):
            last = k + 1 == n   # the last conv: + identity, then the block's final ReLU
            y = _Folded.conv_bn(y, getattr(blk, cn), getattr(blk, bn), relu=act or last,
                                residual=identity if last else None)
        return y

    @staticmethod
    def run(blk, x):
        identity = x
        y = x
        for cn, bn, act in blk.pairs:
            y = getattr(blk, bn)(getattr(blk, cn)(y))
            if act:
                y = F.relu(y)
        if blk.downsample is not None:
            identity = blk.downsample(x)
        return F.relu(y + identity)


class _BlockSeq(nn.Sequential):

    def forward(self, x):
        fast = _Folded.ok(x) and not self.training
        for blk in self:
            x = _Block.run_folded(blk, x) if fast else _Block.run(blk, x)
        return x


class HRModule(nn.Module):
    """hrnet.py:13-214."""

    def __init__(self, num_branches, block, num_blocks, in_channels, num_channels,
                 multiscale_output=True, norm_cfg=dict(type='BN')):
        super().__init__()
        self.in_channels = in_channels
        self.num_branches = num_branches
        self.multiscale_output = multiscale_output
        self.norm_cfg = norm_cfg
        self.branches = nn.ModuleList([
            self._make_one_branch(i, block, num_blocks, num_channels) for i in range(num_branches)])
        self.fuse_layers = self._make_fuse_layers()

    def _make_one_branch(self, i, block, num_blocks, num_channels, stride=1):
        downsample = None
        if stride != 1 or self.in_channels[i] != num_channels[i] * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.in_channels[i], num_channels[i] * block.expansion, 1, stride,
                          bias=False),
                build_norm_layer(self.norm_cfg, num_channels[i] * block.expansion)[1])
        layers = [block(self.in_channels[i], num_channels[i], stride, downsample=downsample,
                        norm_cfg=self.norm_cfg)]
        self.in_channels[i] = num_channels[i] * block.expansion
        for _ in range(1, num_blocks[i]):
            layers.append(block(self.in_channels[i], num_channels[i], norm_cfg=self.norm_cfg))
        return _BlockSeq(*layers)

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        nb, ic = self.num_branches, self.in_channels
        fuse_layers = []
        for i in range(nb if self.multiscale_output else 1):
            fuse_layer = []
            for j in range(nb):
                if j > i:
                    fuse_layer.append(nn.Sequential(
                        nn.Conv2d(ic[j], ic[i], 1, 1, 0, bias=False),
                        build_norm_layer(self.norm_cfg, ic[i])[1],
                        nn.Upsample(scale_factor=2**(j - i), mode='nearest')))
                elif j == i:
                    fuse_layer.append(None)
                else:
                    convs = []
                    for k in range(i - j):
                        if k == i - j - 1:
                            convs.append(nn.Sequential(
                                nn.Conv2d(ic[j], ic[i], 3, 2, 1, bias=False),
                                build_norm_layer(self.norm_cfg, ic[i])[1]))
                        else:
                            convs.append(nn.Sequential(
                                nn.Conv2d(ic[j], ic[j], 3, 2, 1, bias=False),
                                build_norm_layer(self.norm_cfg, ic[j])[1], nn.ReLU(inplace=False)))
                    fuse_layer.append(nn.Sequential(*convs))
            fuse_layers.append(nn.ModuleList(fuse_layer))
        return nn.ModuleList(fuse_layers)

    def forward(self, x):
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        x = [self.branches[i](x[i]) for i in range(self.num_branches)]
        if _Folded.ok(x[0]) and not self.training:
            from . import ops
            x_fuse = []
            for i in range(len(self.fuse_layers)):
                terms = []      # (map, log2 of the nearest-neighbour up-sampling still to apply)
                for j in range(self.num_branches):
                    fl = self.fuse_layers[i][j]
                    if j == i:
                        terms.append((x[i], 0))
                    elif j > i and len(fl) == 3 and isinstance(fl[2], nn.Upsample) \
                            and fl[2].mode == 'nearest' and fl[2].scale_factor == 2 ** (j - i):
                        # 1x1 conv + BN at the coarse resolution; the up-sampling is folded
                        # into the fused sum's read
                        terms.append((_Folded.conv_bn(x[j], fl[0], fl[1]), j - i))
                    else:
                        terms.append((_Folded.seq(fl, x[j]), 0))
                if len(terms) <= 4 and all(
                        t.is_contiguous(memory_format=torch.channels_last)
                        and t.shape[2] << sh == x[i].shape[2] and t.shape[3] << sh == x[i].shape[3]
                        and t.shape[1] % 4 == 0 for t, sh in terms):
                    # sum over the branches in the reference's order + ReLU: one pass
                    x_fuse.append(ops.fuse_sum_nhwc(terms, relu=True))
                    continue
                acc = None
                for (t, sh), j in zip(terms, range(self.num_branches)):
                    if sh:
                        t = F.interpolate(t, scale_factor=2 ** sh, mode='nearest')
                    acc = t if acc is None else acc + t
                x_fuse.append(F.relu(acc))
            return x_fuse
        x_fuse = []
        for i in range(len(self.fuse_layers)):
            y = 0
            for j in range(self.num_branches):
                y = y + (x[j] if i == j else self.fuse_layers[i][j](x[j]))
            x_fuse.append(F.relu(y))
        return x_fuse


class _ModuleSeq(nn.Sequential):

    def forward(self, x):
        for m in self:
            x = m(x)
        return x


@MMDET_MODELS.register_module()
class HRNet(BaseModule):
    blocks_dict = {'BASIC': BasicBlock, 'BOTTLENECK': Bottleneck}

    def __init__(self, extra, in_channels=3, conv_cfg=None, norm_cfg=dict(type='BN'),
                 norm_eval=True, with_cp=False, zero_init_residual=False,
                 multiscale_output=True, pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        assert all(f'stage{i + 1}' in extra for i in range(4))
        self.extra = extra
        self.norm_cfg = norm_cfg
        self.norm_eval = norm_eval
        self.conv1 = nn.Conv2d(in_channels, 64, 3, 2, 1, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, 64)[1]
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, 64)[1]
        cfg1 = extra['stage1']
        block = self.blocks_dict[cfg1['block']]
        nc = cfg1['num_channels'][0]
        self.layer1 = self._make_layer(block, 64, nc, cfg1['num_blocks'][0])
        pre = [nc * block.expansion]
        for s in (2, 3, 4):
            cfg = extra[f'stage{s}']
            block = self.blocks_dict[cfg['block']]
            chans = [c * block.expansion for c in cfg['num_channels']]
            setattr(self, f'transition{s - 1}', self._make_transition_layer(pre, chans))
            stage, pre = self._make_stage(cfg, chans,
                                          multiscale_output if s == 4 else True)
            setattr(self, f'stage{s}', stage)
            setattr(self, f'stage{s}_cfg', cfg)

    def _make_transition_layer(self, pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                if cur[i] != pre[i]:
                    layers.append(nn.Sequential(
                        nn.Conv2d(pre[i], cur[i], 3, 1, 1, bias=False),
                        build_norm_layer(self.norm_cfg, cur[i])[1], nn.ReLU(inplace=True)))
                else:
                    layers.append(None)
            else:
                convs = []
                for j in range(i + 1 - len(pre)):
                    ic = pre[-1]
                    oc = cur[i] if j == i - len(pre) else ic
                    convs.append(nn.Sequential(
                        nn.Conv2d(ic, oc, 3, 2, 1, bias=False),
                        build_norm_layer(self.norm_cfg, oc)[1], nn.ReLU(inplace=True)))
                layers.append(nn.Sequential(*convs))
        return nn.ModuleList(layers)

    def _make_layer(self, block, inplanes, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                build_norm_layer(self.norm_cfg, planes * block.expansion)[1])
        layers = [block(inplanes, planes, stride, downsample=downsample, norm_cfg=self.norm_cfg)]
        inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(inplanes, planes, norm_cfg=self.norm_cfg))
        return _BlockSeq(*layers)

    def _make_stage(self, cfg, in_channels, multiscale_output=True):
        block = self.blocks_dict[cfg['block']]
        mods = []
        for i in range(cfg['num_modules']):
            ms = not (not multiscale_output and i == cfg['num_modules'] - 1)
            mods.append(HRModule(cfg['num_branches'], block, cfg['num_blocks'], in_channels,
                                 cfg['num_channels'], ms, norm_cfg=self.norm_cfg))
        return _ModuleSeq(*mods), in_channels

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._is_init = True

    def forward(self, x):
        if x.dim() == 5:  # [B, T, C, H, W]: the reference HRNet has no frame flatten
            x = x.flatten(0, 1)
        assert not self.training
        fast = _Folded.ok(x)
        if fast and x.is_contiguous() and x.shape[1] == 3 and self.conv1.out_channels == 64 and \
                self.conv1.stride == (2, 2) and self.conv1.padding == (1, 1) and self.conv1.groups == 1:
            # conv1 reads the NCHW batch as it is (no channels_last copy of the images, no library
            # convolution): K = 27 is too thin for the matrix cores, the launch is HBM-bound
            from . import ops
            w, b = _Folded.weights(self.conv1, self.bn1)     # (cached on the conv, re-made when it changes)
            hit = self.conv1.__dict__.get('_pave_taps')
            if hit is None or hit[0] is not w:
                hit = (w, w.detach().permute(1, 2, 3, 0).reshape(27, 64).contiguous())
                self.conv1.__dict__['_pave_taps'] = hit
            x = ops.conv3x3s2_c3_nchw(x, hit[1], b, relu=True)
            x = _Folded.conv_bn(x, self.conv2, self.bn2, relu=True)
        elif fast:
            x = x.contiguous(memory_format=torch.channels_last)
            x = _Folded.conv_bn(x, self.conv1, self.bn1, relu=True)
            x = _Folded.conv_bn(x, self.conv2, self.bn2, relu=True)
        else:
            x = F.relu(self.bn1(self.conv1(x)))
            x = F.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        y_list = [x]
        for s in (2, 3, 4):
            cfg = getattr(self, f'stage{s}_cfg')
            tr = getattr(self, f'transition{s - 1}')
            x_list = []
            for i in range(cfg['num_branches']):
                if tr[i] is not None and fast:
                    x_list.append(_Folded.seq(tr[i], y_list[-1] if s > 2 else x))
                elif tr[i] is not None:
                    x_list.append(tr[i](y_list[-1] if s > 2 else x))
                else:
                    x_list.append(y_list[i] if s > 2 else x)
            y_list = getattr(self, f'stage{s}')(x_list)
        return tuple(y_list[1:])  # hrnet.py:583 (fork)
