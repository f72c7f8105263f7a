"""``mmdet.ChannelMapper`` (a14), restated from
third_party/mmdetection/mmdet/models/necks/channel_mapper.py:50-100."""
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

    def forward(self, inputs):
        assert len(inputs) == len(self.convs)
        outs = [self.convs[i](inputs[i]) for i in range(len(inputs))]
        if self.extra_convs:
            for i in range(len(self.extra_convs)):
                outs.append(self.extra_convs[i](inputs[-1] if i == 0 else outs[-1]))
        return tuple(outs)
