"""DLAUp / IDAUp neck: parameter tree + HIP forward.

API and state_dict keys of reference model/backbone/dla_neck.py (Conv2dBlock :11-38, IDAUp
:41-106, DLAUp :109-143).  proj 3x3 -> depthwise 4x4/2 transposed conv -> node 3x3 over the
(virtual) concat run in libmonocon_hip.so through ``mc_neck_forward`` / ``mc_forward_infer``.
"""
import math
from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn

from hipmonocon.params import BNParams, ConvParams, DeconvParams, HipRuntime, module_state, _Holder


class Conv2dBlock(_Holder):
    def __init__(self, in_planes: int, out_planes: int, kernel_size: int = 3, stride: int = 1, bias: bool = True):
        super().__init__()
        self.conv = ConvParams(in_planes, out_planes, kernel_size, bias=bias)
        self.add_module('bn1', BNParams(out_planes))

    @property
    def norm1(self):
        return getattr(self, 'bn1')


class IDAUp(_Holder):
    def __init__(self, in_channels_list: Tuple[int], up_factors_list: Tuple[int], out_channels: int):
        super().__init__()
        self.in_channels_list, self.out_channels = in_channels_list, out_channels
        for i in range(1, len(in_channels_list)):
            if int(up_factors_list[i]) != 2:
                raise NotImplementedError("only x2 up-sampling steps occur on the MonoCon path")
            up = DeconvParams(out_channels, 4)
            self.fill_upconv_weights(up)
            setattr(self, 'proj_' + str(i), Conv2dBlock(in_channels_list[i], out_channels, 3, 1, bias=False))
            setattr(self, 'up_' + str(i), up)
            setattr(self, 'node_' + str(i), Conv2dBlock(out_channels * 2, out_channels, 3, 1, bias=False))
        self.init_weights()

    def init_weights(self):
        """reference dla_neck.py:74-81 (runs over Conv2d modules only; the deconvs keep the bilinear fill)."""
        for m in self.modules():
            if isinstance(m, ConvParams):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, BNParams):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def fill_upconv_weights(self, upconv) -> None:
        """bilinear kernel copied to every channel (reference dla_neck.py:83-92)."""
        w = upconv.weight.data
        f = math.ceil(w.size(2) / 2)
        c = (2 * f - 1 - f % 2) / (2.0 * f)
        k = torch.tensor([1 - math.fabs(i / f - c) for i in range(w.size(2))], dtype=w.dtype)
        w.copy_((k[:, None] * k[None, :]).expand_as(w))


class DLAUp(nn.Module):
    def __init__(self, in_channels_list: List[int] = (64, 128, 256, 512), scales_list: Tuple[int] = (1, 2, 4, 8),
                 start_level: int = 2):
        super().__init__()
        in_channels_list = list(in_channels_list)
        if in_channels_list != [64, 128, 256, 512] or tuple(scales_list) != (1, 2, 4, 8) or start_level != 2:
            raise NotImplementedError("the HIP plan is built for the DLA-34 neck (64,128,256,512 @ x1,2,4,8)")
        scales = np.array(scales_list, dtype=int)
        self.in_channels_list, self.start_level = in_channels_list, start_level
        for i in range(len(in_channels_list) - 1):
            j = -i - 2
            setattr(self, 'ida_{}'.format(i), IDAUp(in_channels_list[j:], scales[j:] // scales[j], in_channels_list[j]))
            scales[j + 1:] = scales[j]
            in_channels_list[j + 1:] = [in_channels_list[j] for _ in in_channels_list[j + 1:]]
        self._rt = HipRuntime()

    def forward(self, layers: Tuple[torch.Tensor]) -> List[torch.Tensor]:
        if self.training:
            raise NotImplementedError("stand-alone DLAUp.forward is eval-only; training runs through MonoConDetector")
        eng = self._rt.get(module_state(self, "neck."))
        return [eng.neck_forward([None if l is None else l.contiguous() for l in layers])]
