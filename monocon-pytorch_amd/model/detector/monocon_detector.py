"""MonoConDetector: DLA-34 -> DLAUp -> MonoCon dense heads on MI355X.

Same constructor, attributes (``backbone``, ``neck``, ``head``), ``forward`` / ``batch_eval`` /
``load_checkpoint`` and state_dict keys as reference model/detector/monocon_detector.py:28-87.
In eval mode one ``mc_forward_infer`` call runs the whole NHWC plan (no NCHW round trips
between stages, no torch.cat); the ten prediction maps come back NCHW as the reference's.
"""
from typing import Any, Dict, Tuple

import torch
import torch.nn as nn

from hipmonocon.params import HipRuntime
from model.backbone import DLA, DLAUp
from model.dense_heads import MonoConDenseHeads

default_head_config = {'num_classes': 3, 'num_kpts': 9, 'num_alpha_bins': 12, 'max_objs': 30}
default_test_config = {'topk': 30, 'local_maximum_kernel': 3, 'max_per_img': 30, 'test_thres': 0.4}


class MonoConDetector(nn.Module):
    def __init__(self, num_dla_layers: int = 34, pretrained_backbone: bool = True,
                 head_config: Dict[str, Any] = None, test_config: Dict[str, Any] = None):
        super().__init__()
        self.backbone = DLA(num_dla_layers, pretrained=pretrained_backbone)
        self.neck = DLAUp(self.backbone.get_out_channels(start_level=2), start_level=2)
        if head_config is None:
            head_config = default_head_config
        if test_config is None:
            test_config = default_test_config
        self.head = MonoConDenseHeads(in_ch=64, test_config=test_config, **head_config)
        self._rt = HipRuntime()

    def set_precision(self, mode: str = "fp32"):
        """Arithmetic of the convolutions.  'fp32' (default): the fp32 matrix pipe.  'bf16x3' / 'f16x2': fp32 values
        emulated on the bf16 / fp16 matrix pipe by a 3-way / 2-way operand split (same parity tolerances as 'fp32').
        'bf16': bf16 MFMA operands with fp32 accumulation, activations, master weights, BN statistics and losses
        (no reference counterpart, not within the parity tolerance)."""
        self._rt.set_precision(mode)
        return self

    def _engine(self):
        return self._rt.get(self.state_dict(keep_vars=True))

    def finish_batch(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        """a collated batch of transforms.DeferredImage samples -- raw uint8 frames (B, Hp, Wp, 3) + ``img_aug`` (B, 24) -- gets
        its float32 (B, 3, Hp, Wp) frames here, in one launch (mc_preprocess_augmented: the train augmentations' image work +
        Normalize + Pad + ToTensor, bit-identical to the host transforms).  Any other batch passes through."""
        if 'img_aug' in data_dict:
            eng = self._rt.engine if self._rt.engine is not None else self._engine()
            from dataset.monocon_dataset import IMG_MEAN, IMG_STD
            data_dict['img'] = eng.preprocess_augmented(data_dict['img'].contiguous(), data_dict.pop('img_aug').contiguous(),
                                                        IMG_MEAN, IMG_STD)
        return data_dict

    def forward(self, data_dict: Dict[str, Any], return_loss: bool = True) -> Tuple[Dict[str, torch.Tensor]]:
        img = self.finish_batch(data_dict)['img']
        if self.training:
            from hipmonocon.train import forward_train
            pred_dict, loss_dict = forward_train(self, data_dict)
            return (pred_dict, loss_dict) if return_loss else pred_dict
        return self._engine().forward_infer(img.contiguous())

    def batch_eval(self, data_dict: Dict[str, Any], get_vis_format: bool = False) -> Dict[str, Any]:
        if self.training:
            raise Exception("Model is in training mode. Please use '.eval()' first.")
        pred_dict = self.forward(data_dict, return_loss=False)
        return self.head._get_eval_formats(data_dict, pred_dict, get_vis_format=get_vis_format,
                                           engine=self._rt.engine)

    def load_checkpoint(self, ckpt_file: str):
        # the reference pickles whole engine objects; torch >= 2.6 needs weights_only=False for those
        from utils.engine_utils import load_checkpoint_file
        model_dict = load_checkpoint_file(ckpt_file)['state_dict']['model']
        self.load_state_dict(model_dict)

    def _extract_feat_from_data_dict(self, data_dict: Dict[str, Any]) -> torch.Tensor:
        _, feat = self._engine().forward_infer(self.finish_batch(data_dict)['img'].contiguous(), want_feat=True)
        return feat
