"""DLA-34 backbone: parameter tree + HIP forward.

API and state_dict keys of reference model/backbone/dla.py (BasicBlock :12-51, Root :107-132,
Tree :135-205, DLA :208-301); the arithmetic (7x7 stem, fused conv+BN+ReLU(+residual) on the
fp32 MFMA pipe, 2x2 max-pool, virtual-concat 1x1 roots) runs in libmonocon_hip.so
(csrc/conv_mfma.h, csrc/kernels_misc.hip) through ``mc_backbone_forward`` /
``mc_forward_infer``.
"""
import math
import os
from typing import List, Tuple

import torch
import torch.nn as nn

from hipmonocon.params import BNParams, ConvParams, Holders, HipRuntime, module_state, _Holder


class BasicBlock(_Holder):
    def __init__(self, inplanes: int, planes: int, stride: int = 1, dilation: int = 1):
        super().__init__()
        if dilation != 1:
            raise NotImplementedError("dilation != 1 is not on the MonoCon path")
        self.conv1 = ConvParams(inplanes, planes, 3)
        self.bn1 = BNParams(planes)
        self.conv2 = ConvParams(planes, planes, 3)
        self.bn2 = BNParams(planes)
        self.stride = stride


class Root(_Holder):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 1, residual: bool = False):
        super().__init__()
        if kernel_size != 1 or residual:
            raise NotImplementedError("DLA-34 roots are 1x1 without residual")
        self.conv = ConvParams(in_channels, out_channels, 1)
        self.bn = BNParams(out_channels)
        self.residual = residual


class Tree(_Holder):
    """Same registration order as the reference (tree1, tree2, root, project)."""

    def __init__(self, levels: int, block, in_channels: int, out_channels: int, stride: int = 1,
                 level_root: bool = False, root_dim: int = 0, root_kernel_size: int = 1, dilation: int = 1,
                 root_residual: bool = False):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * out_channels
        if level_root:
            root_dim += in_channels
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride)
            self.tree2 = block(out_channels, out_channels, 1)
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        else:
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels, root_dim=root_dim + out_channels)
        self.level_root, self.root_dim, self.levels = level_root, root_dim, levels
        self.downsample = None                     # 2x2 max-pool has no state; done in the HIP plan
        self.project = None
        if in_channels != out_channels:
            self.project = Holders(ConvParams(in_channels, out_channels, 1), BNParams(out_channels))


class DLA(nn.Module):
    arch_settings = {34: (BasicBlock, (1, 1, 1, 2, 2, 1), (16, 32, 64, 128, 256, 512), False)}

    def __init__(self, num_layers: int = 34, in_channels: int = 3, pretrained: bool = True):
        super().__init__()
        if num_layers not in self.arch_settings:
            raise NotImplementedError("only DLA-34 (the configuration MonoCon uses, config/monocon_configs.py:36) "
                                      "is built for MI355X; got %r" % num_layers)
        if in_channels != 3:
            raise NotImplementedError("the stem kernel is specialised for 3 input channels")
        block, levels, channels, residual_root = self.arch_settings[num_layers]
        self.num_layers, self.in_channels, self.channels = num_layers, in_channels, channels
        self.base_layer = Holders(ConvParams(in_channels, channels[0], 7), BNParams(channels[0]))
        self.level0 = Holders(ConvParams(channels[0], channels[0], 3), BNParams(channels[0]))
        self.level1 = Holders(ConvParams(channels[0], channels[1], 3), BNParams(channels[1]))
        self.level2 = Tree(levels[2], block, channels[1], channels[2], 2, level_root=False)
        self.level3 = Tree(levels[3], block, channels[2], channels[3], 2, level_root=True)
        self.level4 = Tree(levels[4], block, channels[3], channels[4], 2, level_root=True)
        self.level5 = Tree(levels[5], block, channels[4], channels[5], 2, level_root=True)
        self._rt = HipRuntime()
        self.init_weights()
        if pretrained:
            self.load_imagenet_weights(num_layers)

    def load_imagenet_weights(self, num_layers: int):
        """The reference downloads dla34-ba72cf86.pth from dl.yf.io (dla.py:248-262).  There is no
        network here: point MONOCON_DLA34_IMAGENET at a local copy of that file instead."""
        path = os.environ.get("MONOCON_DLA34_IMAGENET", "")
        if not path or not os.path.isfile(path):
            raise FileNotFoundError(
                "pretrained=True needs the ImageNet DLA-34 checkpoint (dla34-ba72cf86.pth); set "
                "MONOCON_DLA34_IMAGENET=/path/to/file or construct with pretrained=False")
        self.load_state_dict(torch.load(path, map_location="cpu"), strict=False)

    def init_weights(self):
        """reference dla.py:264-271: conv ~ N(0, sqrt(2 / (k*k*C_out))), BN gamma=1 beta=0."""
        for m in self.modules():
            if isinstance(m, ConvParams):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, BNParams) and m.affine:
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor]:
        if self.training:
            raise NotImplementedError("stand-alone DLA.forward is eval-only; training runs through MonoConDetector")
        eng = self._rt.get(module_state(self, "backbone."))
        return eng.backbone_forward(x.contiguous())

    def get_out_channels(self, start_level: int) -> List[int]:
        return list(self.channels[start_level:])
