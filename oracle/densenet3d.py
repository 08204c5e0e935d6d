"""DenseNet121, spatial_dims=3 -- CPU restatement of the third-party encoder the reference instantiates.

PARITY UNPINNED: MONAI (requirements.txt: monai>=1.3.0, unpinned) is not in /root/reference and not installed
here.  This file restates monai/networks/nets/densenet.py (class DenseNet / _DenseBlock / _DenseLayer /
_Transition, v1.3) as called at reference sites final_multimodal.py:66-71,
partial_modality_training.py:171-176, simple_fusion.py:182-187:
    DenseNet121(spatial_dims=3, in_channels=1, out_channels=128, pretrained=False)
 -> init_features=64, growth_rate=32, block_config=(6,12,24,16), bn_size=4, act=relu, norm=batch,
    dropout_prob=0.  Sub-module names reproduce MONAI's state_dict keys
    (features.denseblock1.denselayer1.layers.conv1.weight, features.transition1.conv.weight,
    features.norm5.weight, class_layers.out.weight, ...).  Init: conv kaiming_normal_, BN gamma=1 beta=0,
    Linear bias 0.
"""
from collections import OrderedDict

import torch
import torch.nn as nn


class _DenseLayer(nn.Module):
    def __init__(self, in_channels, growth_rate, bn_size):
        super().__init__()
        mid = bn_size * growth_rate
        self.layers = nn.Sequential()
        self.layers.add_module("norm1", nn.BatchNorm3d(in_channels))
        self.layers.add_module("relu1", nn.ReLU(inplace=True))
        self.layers.add_module("conv1", nn.Conv3d(in_channels, mid, kernel_size=1, bias=False))
        self.layers.add_module("norm2", nn.BatchNorm3d(mid))
        self.layers.add_module("relu2", nn.ReLU(inplace=True))
        self.layers.add_module("conv2", nn.Conv3d(mid, growth_rate, kernel_size=3, padding=1, bias=False))

    def forward(self, x):
        return torch.cat([x, self.layers(x)], 1)


class _DenseBlock(nn.Sequential):
    def __init__(self, layers, in_channels, bn_size, growth_rate):
        super().__init__()
        for i in range(layers):
            self.add_module("denselayer%d" % (i + 1), _DenseLayer(in_channels, growth_rate, bn_size))
            in_channels += growth_rate


class _Transition(nn.Sequential):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.add_module("norm", nn.BatchNorm3d(in_channels))
        self.add_module("relu", nn.ReLU(inplace=True))
        self.add_module("conv", nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=False))
        self.add_module("pool", nn.AvgPool3d(kernel_size=2, stride=2))


class DenseNet121(nn.Module):
    def __init__(self, spatial_dims=3, in_channels=1, out_channels=128, pretrained=False,
                 init_features=64, growth_rate=32, block_config=(6, 12, 24, 16), bn_size=4):
        super().__init__()
        assert spatial_dims == 3 and not pretrained
        self.features = nn.Sequential(OrderedDict([
            ("conv0", nn.Conv3d(in_channels, init_features, kernel_size=7, stride=2, padding=3, bias=False)),
            ("norm0", nn.BatchNorm3d(init_features)),
            ("relu0", nn.ReLU(inplace=True)),
            ("pool0", nn.MaxPool3d(kernel_size=3, stride=2, padding=1)),
        ]))
        c = init_features
        for i, n in enumerate(block_config):
            self.features.add_module(f"denseblock{i + 1}", _DenseBlock(n, c, bn_size, growth_rate))
            c += n * growth_rate
            if i == len(block_config) - 1:
                self.features.add_module("norm5", nn.BatchNorm3d(c))
            else:
                self.features.add_module(f"transition{i + 1}", _Transition(c, c // 2))
                c = c // 2
        self.class_layers = nn.Sequential(OrderedDict([
            ("relu", nn.ReLU(inplace=True)),
            ("pool", nn.AdaptiveAvgPool3d(1)),
            ("flatten", nn.Flatten(1)),
            ("out", nn.Linear(c, out_channels)),
        ]))
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return self.class_layers(self.features(x))
