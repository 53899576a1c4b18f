"""Backbone wrapper with the reference's name and state_dict layout.

`ReMobileNetV2` mirrors reference model_feature.py:49-69: `.features` is
torchvision's `mobilenet_v2().features` (stem ConvBNReLU, 17 InvertedResidual
blocks, the unused 320->1280 ConvBNReLU at index 18) and the forward taps
features[0:2], [2:4], [4:7], [7:14], [14:18].  torchvision is not a dependency:
the layer table is restated here; weights arrive through `load_state_dict`
(the reference's `pretrained=True` download is not reproduced).  These modules
only hold parameters -- the arithmetic runs in the HIP engine (engine.py).
"""
import torch.nn as nn

# (expand ratio, out channels, repeats, first stride) of MobileNetV2
INVERTED_RESIDUAL_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
                             (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
TAP_BOUNDS = (2, 4, 7, 14, 18)   # model_feature.py:63-67


def _no_eager(name):
    raise RuntimeError(
        "%s.forward: sub-modules only hold parameters here; the arithmetic of the UAVSal hot path "
        "runs in the HIP engine. Call UAVSal.forward / UAVSal.forward_clips." % name)


class ConvBNReLU(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3, stride=1, groups=1):
        super().__init__(
            nn.Conv2d(cin, cout, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=False),
            nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))
        self.stride = stride

    def forward(self, x):
        _no_eager("ConvBNReLU")


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, expand_ratio):
        super().__init__()
        hidden = int(round(cin * expand_ratio))
        self.stride, self.cin, self.cout, self.hidden = stride, cin, cout, hidden
        self.expand_ratio = expand_ratio
        self.dilation = 1
        self.use_res_connect = stride == 1 and cin == cout
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU(cin, hidden, 1))
        layers += [ConvBNReLU(hidden, hidden, 3, stride, groups=hidden),
                   nn.Conv2d(hidden, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        _no_eager("InvertedResidual")


def mobilenet_v2_features() -> nn.Sequential:
    layers = [ConvBNReLU(3, 32, 3, stride=2)]
    cin = 32
    for t, c, n, s in INVERTED_RESIDUAL_SETTING:
        for i in range(n):
            layers.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
            cin = c
    layers.append(ConvBNReLU(cin, 1280, 1))          # held, never executed (model_feature.py:68)
    return nn.Sequential(*layers)


class ReMobileNetV2(nn.Module):
    def __init__(self, name="mobilenet_v2"):
        super().__init__()
        if name != "mobilenet_v2":
            raise ValueError(name)
        self.features = mobilenet_v2_features()

    def forward(self, x):
        _no_eager("ReMobileNetV2")
