"""Attentive normalisation: parameter holder.

state_dict keys of reference model/norm/attentive_norm.py:118-164 (``weight_``, ``bias_``,
BatchNorm buffers, ``attn_weights.attention.{0,1}``).  The computation -- instance statistics
from the fused 3x3 head conv, attention 1x1 + BN(10) + hard-sigmoid, per-sample affine folded
with the BN running statistics -- is csrc/kernels_misc.hip: head_attn_kernel / head_apply_kernel.
"""
import torch
import torch.nn as nn

from hipmonocon.params import BNParams, ConvParams, Holders, _Holder


class AttnWeights(_Holder):
    def __init__(self, num_features: int, num_affine_trans: int, eps: float = 1e-3):
        super().__init__()
        self.num_affine_trans, self.eps = num_affine_trans, eps
        self.attention = Holders(ConvParams(num_features, num_affine_trans, 1, bias=False),
                                 BNParams(num_affine_trans))
        nn.init.kaiming_normal_(self.attention[0].weight, a=0.0, mode='fan_out', nonlinearity='relu')


class AttnBatchNorm2d(BNParams):
    def __init__(self, num_features: int, num_affine_trans: int, attn_mode: int = 0, eps: float = 1e-5,
                 momentum: float = 0.1, track_running_stats: bool = True, use_rsd: bool = True,
                 use_maxpool: bool = False, use_bn: bool = True, eps_var: float = 1e-3):
        if attn_mode != 0 or not use_rsd or use_maxpool or not use_bn or not track_running_stats:
            raise NotImplementedError("only the MonoCon configuration (attn_mode=0, rsd, BN attention) is built")
        super().__init__(num_features, eps=eps, momentum=momentum, affine=False)
        self.num_affine_trans, self.eps_var = num_affine_trans, eps_var
        self.weight_ = nn.Parameter(torch.Tensor(num_affine_trans, num_features))
        self.bias_ = nn.Parameter(torch.Tensor(num_affine_trans, num_features))
        self.attn_weights = AttnWeights(num_features, num_affine_trans, eps=eps_var)
        self.init_weights()

    def init_weights(self):
        nn.init.normal_(self.weight_, 1.0, 0.1)
        nn.init.normal_(self.bias_, 0.0, 0.1)
