"""MonoCon dense heads: parameter tree, HIP prediction pass and HIP decode.

API surface and state_dict keys of reference model/dense_heads/monocon_heads.py
(MonoConDenseHeads :38-586).  Device work is libmonocon_hip.so:
  * _get_predictions  -> fused 3x3 64->9x64 conv (+bias, instance statistics) on the fp32 MFMA
                         pipe, AttnBN attention, normalise+ReLU+1x1+sigmoid/clamp/depth epilogue
                         (csrc/conv_mfma.h, csrc/kernels_misc.hip);
  * decode_heatmap    -> 3x3 local-maximum, exact per-image top-K, gathers, 2D/3D box assembly
                         (csrc/kernels_decode.hip), with a precomputed (B,3,4)/(B,4,4) calibration
                         tensor instead of the reference's per-image CPU loop (:501,541-555).
"""
from typing import Any, Dict, List, Tuple

import numpy as np
import torch
import torch.nn as nn

from hipmonocon.params import ConvParams, Holders, HipRuntime, module_state
from hipmonocon.engine import p2_inverse
from model.norm import AttnBatchNorm2d

EPS = 1e-12
PI = np.pi

DEFAULT_TEST_CFG = {'topk': 30, 'local_maximum_kernel': 3, 'max_per_img': 30, 'test_thres': 0.4}


class _HipLosses(torch.autograd.Function):
    """(10,) loss vector of ten prediction maps; backward returns d(sum_i g_i * loss_i) / d pred_k."""

    @staticmethod
    def forward(ctx, eng, T, max_objs, *preds):
        from hipmonocon import netspec
        keys = [k for k, _ in netspec.PRED_KEYS]
        pd = {k: p.detach().contiguous() for k, p in zip(keys, preds)}
        ctx.eng, ctx.T, ctx.pd, ctx.max_objs = eng, T, pd, max_objs
        return eng.losses(pd, T, max_objs=max_objs)

    @staticmethod
    def backward(ctx, grad_losses):
        d = ctx.eng.losses_backward(ctx.pd, ctx.T, grad_losses.contiguous().float(), max_objs=ctx.max_objs, wrt_pred=True)
        return (None, None, None, *[d[k] for k in ctx.pd])


class MonoConDenseHeads(nn.Module):
    def __init__(self, in_ch: int = 64, feat_ch: int = 64, num_kpts: int = 9, num_alpha_bins: int = 12,
                 num_classes: int = 3, max_objs: int = 30, test_config: Dict[str, Any] = None):
        super().__init__()
        if (in_ch, feat_ch, num_kpts, num_alpha_bins, num_classes) != (64, 64, 9, 12, 3):
            raise NotImplementedError("the HIP head kernels are built for the MonoCon/DLA-34 configuration "
                                      "(64 ch, 9 keypoints, 12 bins, 3 classes)")
        self.max_objs, self.num_kpts = max_objs, num_kpts
        self.num_alpha_bins, self.num_classes = num_alpha_bins, num_classes
        if test_config is None:
            test_config = DEFAULT_TEST_CFG
        self.test_config = test_config
        for k, v in test_config.items():
            setattr(self, k, v)

        self.heatmap_head = self._build_head(in_ch, feat_ch, num_classes)
        self.wh_head = self._build_head(in_ch, feat_ch, 2)
        self.offset_head = self._build_head(in_ch, feat_ch, 2)
        self.center2kpt_offset_head = self._build_head(in_ch, feat_ch, num_kpts * 2)
        self.kpt_heatmap_head = self._build_head(in_ch, feat_ch, num_kpts)
        self.kpt_heatmap_offset_head = self._build_head(in_ch, feat_ch, 2)
        self.dim_head = self._build_head(in_ch, feat_ch, 3)
        self.depth_head = self._build_head(in_ch, feat_ch, 2)
        self.dir_feat, self.dir_cls, self.dir_reg = self._build_dir_head(in_ch, feat_ch)
        self._rt = HipRuntime()
        self.init_weights()

    def _build_head(self, in_ch: int, feat_ch: int, out_channel: int) -> nn.Module:
        # indices 0,1,3 carry parameters; index 2 is the (state-less) ReLU of the reference Sequential
        return Holders(ConvParams(in_ch, feat_ch, 3, bias=True),
                       AttnBatchNorm2d(feat_ch, 10, momentum=0.03, eps=0.001), None,
                       ConvParams(feat_ch, out_channel, 1, bias=True))

    def _build_dir_head(self, in_ch: int, feat_ch: int) -> Tuple[nn.Module]:
        dir_feat = Holders(ConvParams(in_ch, feat_ch, 3, bias=True),
                           AttnBatchNorm2d(feat_ch, 10, momentum=0.03, eps=0.001))
        dir_cls = Holders(ConvParams(feat_ch, self.num_alpha_bins, 1, bias=True))
        dir_reg = Holders(ConvParams(feat_ch, self.num_alpha_bins, 1, bias=True))
        return dir_feat, dir_cls, dir_reg

    def init_weights(self, prior_prob: float = 0.1) -> None:
        """reference monocon_heads.py:134-146."""
        bias_init = float(-np.log((1 - prior_prob) / prior_prob))
        self.heatmap_head[-1].bias.data.fill_(bias_init)
        self.kpt_heatmap_head[-1].bias.data.fill_(bias_init)
        for head in [self.wh_head, self.offset_head, self.center2kpt_offset_head, self.kpt_heatmap_offset_head,
                     self.depth_head, self.dim_head, self.dir_feat, self.dir_cls, self.dir_reg]:
            for m in head.modules():
                if isinstance(m, ConvParams):
                    nn.init.normal_(m.weight, 0.0, 0.001)
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0.0)

    # ------------------------------------------------------------------ forward
    def forward_train(self, feat: torch.Tensor, data_dict: Dict[str, Any]):
        """reference monocon_heads.py:150-157: ``(pred_dict, loss_dict)`` from a neck output -- target generation,
        train-mode predictions (AttnBatchNorm2d on batch statistics, running buffers updated) and the ten losses in
        one HIP plan (``mc_head_forward_train``); ``sum(loss_dict.values()).backward()`` fills the head parameters'
        ``.grad`` and the gradient of ``feat`` (``mc_head_backward``)."""
        from hipmonocon.train import head_forward_train
        return head_forward_train(self, feat, data_dict)

    def forward_test(self, feat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self._get_predictions(feat)

    def _engine(self, prefix="head."):
        return self._rt.get(module_state(self, prefix))

    def _get_predictions(self, feat: torch.Tensor) -> Dict[str, torch.Tensor]:
        if self.training:
            raise NotImplementedError("stand-alone head prediction is eval-only; training runs through MonoConDetector")
        return self._engine().head_forward(feat.contiguous())

    # ------------------------------------------------------------------ losses
    def _get_losses(self, pred_dict: Dict[str, torch.Tensor], target_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """reference monocon_heads.py:203-310: the ten losses of a prediction dict against a target dict (as produced
        by the reference's TargetGenerator or ``hipmonocon.Engine.make_targets``), autograd-connected to the
        prediction tensors: forward = ``mc_losses`` (two focal reductions + one gather kernel), backward =
        ``mc_losses_backward_pred`` (gradient wrt the maps in ``pred_dict``)."""
        from hipmonocon import netspec
        eng = self._engine()
        keys = [k for k, _ in netspec.PRED_KEYS]
        T = {}
        for k, v in target_dict.items():
            if k in ("indices", "indices_kpt"):
                T[k] = v.to(torch.int64).contiguous()
            elif k == "mask_target":
                T[k] = v.to(torch.bool).contiguous()
            else:
                T[k] = v.to(torch.float32).contiguous()
        losses = _HipLosses.apply(eng, T, self.max_objs, *[pred_dict[k] for k in keys])
        return {k: losses[i] for i, k in enumerate(netspec.LOSS_KEYS)}

    # ------------------------------------------------------------------ decode
    @staticmethod
    def _calib_tensors(batched_calib, device):
        P2 = np.stack([np.asarray(c.P2, dtype=np.float32).reshape(3, 4) for c in batched_calib])
        return torch.from_numpy(P2).to(device), torch.from_numpy(p2_inverse(P2)).to(device)

    def _decode_dense(self, data_dict: Dict[str, Any], pred_dict: Dict[str, torch.Tensor], engine=None):
        img_h, img_w = data_dict['img_metas']['pad_shape'][0]
        heat = pred_dict['center_heatmap_pred']
        k = int(self.local_maximum_kernel)
        if k < 1 or k % 2 == 0:
            # the reference fails here too: max_pool2d(heat, k, 1, (k - 1) // 2) of an even k is one pixel smaller than
            # the heat map and `hmax == heat` does not broadcast (utils/tensor_ops.py:17-21)
            raise RuntimeError("local_maximum_kernel=%r: the peak filter needs an odd window" % self.local_maximum_kernel)
        calib = data_dict['calib']
        if not isinstance(calib, (list, tuple)):
            calib = [calib] * heat.shape[0]
        P2, P2inv = self._calib_tensors(calib, heat.device)
        eng = engine if engine is not None else self._engine()
        return eng.decode({k: v.contiguous() for k, v in pred_dict.items()}, P2, P2inv, (img_h, img_w),
                          self.topk, self.test_thres, local_maximum_kernel=k)

    def decode_heatmap(self, data_dict: Dict[str, Any], pred_dict: Dict[str, torch.Tensor],
                       engine=None) -> Tuple[List[torch.Tensor]]:
        """reference monocon_heads.py:399-482: ragged per-image lists of (n,5), (n,7), (n,)."""
        R = self._decode_dense(data_dict, pred_dict, engine)
        box3d = R['box3d'].clone()
        box3d[..., 1] -= 0.5 * box3d[..., 4]            # undo the origin shift: decode_heatmap returns centres
        masks = R['box_mask']
        return ([b[m] for b, m in zip(R['box2d'], masks)],
                [b[m] for b, m in zip(box3d, masks)],
                [c[m] for c, m in zip(R['cls'], masks)])

    def _get_bboxes(self, data_dict: Dict[str, Any], pred_dict: Dict[str, torch.Tensor], engine=None):
        """reference monocon_heads.py:313-329 (origin (0.5,0.5,0.5) -> (0.5,1.0,0.5) is fused in the kernel)."""
        R = self._decode_dense(data_dict, pred_dict, engine)
        masks = R['box_mask']
        return ([b[m] for b, m in zip(R['box2d'], masks)],
                [b[m] for b, m in zip(R['box3d'], masks)],
                [c[m] for c, m in zip(R['cls'], masks)])

    def _get_eval_formats(self, data_dict: Dict[str, Any], pred_dict: Dict[str, torch.Tensor],
                          get_vis_format: bool = False, engine=None) -> Dict[str, Any]:
        """reference monocon_heads.py:333-376."""
        bboxes_2d, bboxes_3d, labels = self._get_bboxes(data_dict, pred_dict, engine)
        result_list = []
        for bbox_2d, bbox_3d, label in zip(bboxes_2d, bboxes_3d, labels):
            result_list.append({
                'img_bbox': self.bbox_3d_to_result(bbox_3d, bbox_2d[:, -1], label),
                'img_bbox2d': self.bbox_2d_to_result(bbox_2d, label, self.num_classes)})
        if get_vis_format:
            return result_list
        from utils.kitti_convert_utils import convert_to_kitti_2d, convert_to_kitti_3d
        kitti_2d = convert_to_kitti_2d([r['img_bbox2d'] for r in result_list], data_dict['img_metas'])
        kitti_3d = convert_to_kitti_3d([r['img_bbox'] for r in result_list], data_dict['img_metas'],
                                       data_dict['calib'])
        return {'img_bbox': kitti_3d, 'img_bbox2d': kitti_2d}

    def bbox_2d_to_result(self, bboxes_2d: torch.Tensor, labels: torch.Tensor, num_classes: int) -> List[np.ndarray]:
        if bboxes_2d.shape[0] == 0:
            return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
        b = bboxes_2d.detach().cpu().numpy()
        l = labels.detach().cpu().numpy()
        return [b[l == c_i, :] for c_i in range(num_classes)]

    def bbox_3d_to_result(self, bboxes_3d: torch.Tensor, scores: torch.Tensor, labels: torch.Tensor):
        return dict(boxes_3d=bboxes_3d.cpu(), scores_3d=scores.cpu(), labels_3d=labels.cpu())
