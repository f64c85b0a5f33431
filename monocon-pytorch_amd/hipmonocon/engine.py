"""Host-side wrapper of one libmonocon_hip handle: binds a torch ``state_dict``, keeps the
packed device copies coherent with the master parameters, and exposes the forward /
decode / op-level calls on torch tensors (raw ``data_ptr()`` + current HIP stream).

PyTorch is plumbing here (device memory, streams); every device computation is a
kernel of libmonocon_hip.so.  No CPU path exists: a non-CUDA tensor raises.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as _lib
from . import netspec

PRED_KEYS = tuple(k for k, _ in netspec.PRED_KEYS)
PRED_CH = tuple(c for _, c in netspec.PRED_KEYS)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_cuda(t, what):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise _lib.MonoconHipError("%s must be a CUDA(HIP) tensor; libmonocon_hip has no CPU path" % what)
    if not t.is_contiguous():
        raise _lib.MonoconHipError("%s must be contiguous" % what)
    return t


class Engine:
    """One handle per process per GPU (SURVEY §8b threading rules)."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.MonoconHipError("no HIP device visible; libmonocon_hip has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        h = C.c_void_p()
        rc = self.lib.mc_create(self.device.index, C.byref(h))
        _lib.check(None, rc, "mc_create")
        self.h = h
        self._sig = None
        self._keep = None
        self.comm_world = 0        # > 0: the handle owns an RCCL communicator (comm_init)
        self._lm_kernel = 3        # the handle's local-maximum window (mc_set_local_maximum_kernel)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def bind_state(self, state):
        """Bind (and pack) a name -> CUDA tensor mapping with the reference's 449 keys.
        Re-binds only when a pointer changed, re-packs only when a tensor's ``_version``
        changed (optimizer step, load_state_dict, ...)."""
        ptr_sig = tuple(v.data_ptr() for v in state.values())
        ver_sig = tuple(v._version for v in state.values())
        if self._sig is not None and self._sig == (ptr_sig, ver_sig):
            return
        if self._sig is None or self._sig[0] != ptr_sig:
            descs = (_lib.TensorDesc * len(state))()
            names = []
            for i, (k, v) in enumerate(state.items()):
                _need_cuda(v, k)
                if v.dtype == torch.float32:
                    dt = _lib.MC_F32
                elif v.dtype == torch.int64:
                    dt = _lib.MC_I64
                else:
                    raise _lib.MonoconHipError("%s: unsupported dtype %s" % (k, v.dtype))
                names.append(k.encode())
                descs[i] = _lib.TensorDesc(names[-1], v.data_ptr(), v.numel(), dt)
            _lib.check(self.h, self.lib.mc_bind_params(self.h, descs, len(state)), "mc_bind_params")
            self._keep = (state, names)
        with torch.cuda.device(self.device):
            _lib.check(self.h, self.lib.mc_pack_params(self.h, 0, _stream()), "mc_pack_params")
        self._sig = (ptr_sig, ver_sig)

    # ------------------------------------------------------------------ forward
    def forward_infer(self, img, want_feat=False):
        """(B,3,H,W) fp32 NCHW -> OrderedDict of the ten NCHW prediction maps."""
        _need_cuda(img, "img")
        if img.dtype != torch.float32 or img.dim() != 4 or img.shape[1] != 3:
            raise _lib.MonoconHipError("img must be (B,3,H,W) float32")
        B, _, H, W = img.shape
        fh, fw = H // 4, W // 4
        preds = [torch.empty((B, c, fh, fw), dtype=torch.float32, device=img.device) for c in PRED_CH]
        feat = torch.empty((B, 64, fh, fw), dtype=torch.float32, device=img.device) if want_feat else None
        arr = (C.c_void_p * _lib.NUM_PREDS)(*[p.data_ptr() for p in preds])
        with torch.cuda.device(img.device):
            rc = self.lib.mc_forward_infer(self.h, _ptr(img), B, H, W, arr, _ptr(feat), _stream())
        _lib.check(self.h, rc, "mc_forward_infer")
        out = dict(zip(PRED_KEYS, preds))
        return (out, feat) if want_feat else out

    # ------------------------------------------------------------------ stage-level forwards
    LEVEL_CH = (16, 32, 64, 128, 256, 512)

    def backbone_forward(self, img):
        """DLA.forward: (B,3,H,W) -> tuple of the six NCHW level outputs."""
        _need_cuda(img, "img")
        B, _, H, W = img.shape
        lv = [torch.empty((B, c, H >> i, W >> i), dtype=torch.float32, device=img.device)
              for i, c in enumerate(self.LEVEL_CH)]
        arr = (C.c_void_p * 6)(*[t.data_ptr() for t in lv])
        with torch.cuda.device(img.device):
            rc = self.lib.mc_backbone_forward(self.h, _ptr(img), B, H, W, arr, _stream())
        _lib.check(self.h, rc, "mc_backbone_forward")
        return tuple(lv)

    def neck_forward(self, levels):
        """DLAUp.forward: six NCHW levels (only 2..5 are read) -> (B,64,H/4,W/4)."""
        l2 = _need_cuda(levels[2], "levels[2]")
        B, _, fh, fw = l2.shape
        H, W = fh * 4, fw * 4
        arr = (C.c_void_p * 6)()
        for i in range(2, 6):
            t = _need_cuda(levels[i], "levels[%d]" % i)
            if tuple(t.shape) != (B, self.LEVEL_CH[i], H >> i, W >> i):
                raise _lib.MonoconHipError("levels[%d] has shape %s" % (i, tuple(t.shape)))
            arr[i] = t.data_ptr()
        feat = torch.empty((B, 64, fh, fw), dtype=torch.float32, device=l2.device)
        with torch.cuda.device(l2.device):
            rc = self.lib.mc_neck_forward(self.h, arr, B, H, W, _ptr(feat), _stream())
        _lib.check(self.h, rc, "mc_neck_forward")
        return feat

    def head_forward(self, feat):
        """MonoConDenseHeads._get_predictions: (B,64,h,w) NCHW -> dict of ten NCHW maps."""
        _need_cuda(feat, "feat")
        B, c, fh, fw = feat.shape
        if c != 64 or feat.dtype != torch.float32:
            raise _lib.MonoconHipError("feat must be (B,64,h,w) float32")
        preds = [torch.empty((B, pc, fh, fw), dtype=torch.float32, device=feat.device) for pc in PRED_CH]
        arr = (C.c_void_p * _lib.NUM_PREDS)(*[p.data_ptr() for p in preds])
        with torch.cuda.device(feat.device):
            rc = self.lib.mc_head_forward(self.h, _ptr(feat), B, fh * 4, fw * 4, arr, _stream())
        _lib.check(self.h, rc, "mc_head_forward")
        return dict(zip(PRED_KEYS, preds))

    def forward_cost(self, B, H, W):
        """-> dict(conv_flops, other_flops, conv_bytes, other_bytes) of one forward."""
        fl, by = (C.c_double * 2)(), (C.c_double * 2)()
        _lib.check(self.h, self.lib.mc_forward_cost(self.h, B, H, W, fl, by), "mc_forward_cost")
        return {"conv_flops": fl[0], "other_flops": fl[1], "conv_bytes": by[0], "other_bytes": by[1]}

    def profile_forward(self, iters=3):
        ms = (C.c_float * 3)()
        n = (C.c_int * 3)()
        with torch.cuda.device(self.device):
            _lib.check(self.h, self.lib.mc_profile_forward(self.h, iters, ms, n, _stream()), "mc_profile_forward")
        return {"conv_ms": ms[0], "other_ms": ms[1], "total_ms": ms[2],
                "n_conv": n[0], "n_other": n[1], "n_ops": n[2]}

    def profile_train(self, iters=2):
        """per kernel family durations of the last train step (after forward_train + backward)."""
        ms = (C.c_double * 3)()
        fl = (C.c_double * 3)()
        by = (C.c_double * 3)()
        n = (C.c_int * 3)()
        with torch.cuda.device(self.device):
            _lib.check(self.h, self.lib.mc_profile_train(self.h, iters, ms, fl, by, n, _stream()), "mc_profile_train")
        names = ("other", "conv", "wgrad")
        return {k: {"ms": ms[i], "flops": fl[i], "bytes": by[i], "launches": n[i]} for i, k in enumerate(names)}

    def query_workspace(self, B, H, W, mode="train"):
        """device bytes the plan for this shape needs, without building it (mode: 'infer', 'train', 'head_train')"""
        out = C.c_size_t(0)
        m = {"infer": 0, "train": 1, "head_train": 2}[mode]
        with torch.cuda.device(self.device):
            _lib.check(self.h, self.lib.mc_query_workspace(self.h, int(B), int(H), int(W), m, C.byref(out)), "mc_query_workspace")
        return int(out.value)

    def workspace_bytes(self):
        return int(self.lib.mc_workspace_bytes(self.h))

    # ------------------------------------------------------------------ KITTI evaluation (device part)
    def rotate_iou(self, boxes, query_boxes, criterion=-1):
        """(N,5), (K,5) float32 CUDA tensors [cx, cy, dx, dy, angle] -> (N,K) float32 CUDA tensor (mc_rotate_iou_eval)."""
        if torch.is_tensor(boxes) and torch.is_tensor(query_boxes):
            boxes, query_boxes = boxes.contiguous(), query_boxes.contiguous()
        _need_cuda(boxes, "boxes"); _need_cuda(query_boxes, "query_boxes")
        if boxes.dtype != torch.float32 or query_boxes.dtype != torch.float32:
            raise _lib.MonoconHipError("rotate_iou: boxes must be float32")
        if boxes.dim() != 2 or boxes.shape[1] != 5 or query_boxes.dim() != 2 or query_boxes.shape[1] != 5:
            raise _lib.MonoconHipError("rotate_iou: boxes must be (N,5) and (K,5)")
        N, K = boxes.shape[0], query_boxes.shape[0]
        out = torch.zeros((N, K), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.mc_rotate_iou_eval(self.h, _ptr(boxes), _ptr(query_boxes), N, K, int(criterion), _ptr(out), _stream())
        _lib.check(self.h, rc, "mc_rotate_iou_eval")
        return out

    def box3d_overlap(self, boxes, query_boxes, criterion=-1):
        """(N,7), (K,7) float64 CUDA tensors, camera-frame [x, y, z, l, h, w, ry] -> (N,K) float64 (mc_box3d_overlap)."""
        if torch.is_tensor(boxes) and torch.is_tensor(query_boxes):
            boxes, query_boxes = boxes.contiguous(), query_boxes.contiguous()
        _need_cuda(boxes, "boxes"); _need_cuda(query_boxes, "query_boxes")
        if boxes.dtype != torch.float64 or query_boxes.dtype != torch.float64:
            raise _lib.MonoconHipError("box3d_overlap: boxes must be float64")
        if boxes.dim() != 2 or boxes.shape[1] != 7 or query_boxes.dim() != 2 or query_boxes.shape[1] != 7:
            raise _lib.MonoconHipError("box3d_overlap: boxes must be (N,7) and (K,7)")
        N, K = boxes.shape[0], query_boxes.shape[0]
        out = torch.zeros((N, K), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.mc_box3d_overlap(self.h, _ptr(boxes), _ptr(query_boxes), N, K, int(criterion), _ptr(out), _stream())
        _lib.check(self.h, rc, "mc_box3d_overlap")
        return out

    # ------------------------------------------------------------------ decode
    def decode(self, pred, P2, P2inv, pad_hw, topk, thres, want_keep=False, local_maximum_kernel=3):
        """Dense decode.  pred: dict of NCHW CUDA maps; P2 (B,3,4), P2inv (B,4,4) CUDA fp32.
        local_maximum_kernel: window of the peak filter (odd; test_config['local_maximum_kernel'])."""
        heat = _need_cuda(pred["center_heatmap_pred"], "center_heatmap_pred")
        if int(local_maximum_kernel) != self._lm_kernel:
            _lib.check(self.h, self.lib.mc_set_local_maximum_kernel(self.h, int(local_maximum_kernel)), "mc_set_local_maximum_kernel")
            self._lm_kernel = int(local_maximum_kernel)
        B, Cc, H, W = heat.shape
        dev = heat.device
        arr = (C.c_void_p * _lib.NUM_PREDS)()
        for i, k in enumerate(PRED_KEYS):
            t = pred.get(k)
            arr[i] = _need_cuda(t, k).data_ptr() if t is not None else None
        _need_cuda(P2, "P2"); _need_cuda(P2inv, "P2inv")
        K = int(topk)
        scores = torch.empty((B, K), dtype=torch.float32, device=dev)
        flat = torch.empty((B, K), dtype=torch.int64, device=dev)
        cls = torch.empty((B, K), dtype=torch.int64, device=dev)
        box2d = torch.empty((B, K, 5), dtype=torch.float32, device=dev)
        box3d = torch.empty((B, K, 7), dtype=torch.float32, device=dev)
        keep = torch.empty((B, Cc, H, W), dtype=torch.uint8, device=dev) if want_keep else None
        kthr = torch.empty((B, K), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = self.lib.mc_decode(self.h, arr, _ptr(P2), _ptr(P2inv), B, Cc, H, W, K, float(thres),
                                    float(pad_hw[0]), float(pad_hw[1]), _ptr(scores), _ptr(flat), _ptr(cls),
                                    _ptr(box2d), _ptr(box3d), _ptr(keep), _ptr(kthr), _stream())
        _lib.check(self.h, rc, "mc_decode")
        return dict(scores=scores, flat_index=flat, cls=cls, box2d=box2d, box3d=box3d, keep=keep,
                    box_mask=kthr.bool())

    # ------------------------------------------------------------------ targets / losses
    TARGET_SHAPES = (("center_heatmap_target", "map3", torch.float32), ("wh_target", 2, torch.float32),
                     ("offset_target", 2, torch.float32), ("dim_target", 3, torch.float32),
                     ("alpha_cls_target", 1, torch.float32), ("alpha_offset_target", 1, torch.float32),
                     ("depth_target", 1, torch.float32), ("center2kpt_offset_target", 18, torch.float32),
                     ("kpt_heatmap_target", "map9", torch.float32), ("kpt_heatmap_offset_target", 18, torch.float32),
                     ("indices", 0, torch.int64), ("indices_kpt", 9, torch.int64), ("mask_target", 0, torch.bool),
                     ("mask_center2kpt_offset", 18, torch.float32), ("mask_kpt_heatmap_offset", 18, torch.float32))

    def _targets_struct(self, T):
        st = _lib.Targets()
        for name in _lib.TARGET_FIELDS:
            setattr(st, name, _need_cuda(T[name], name).data_ptr())
        return st

    def make_targets(self, label, pad_hw, feat_hw, max_objs=30):
        """TargetGenerator.__call__: dict of fp32 CUDA label tensors -> dict of the 15 target tensors."""
        mask = _need_cuda(label["mask"], "label.mask")
        B, dev = mask.shape[0], mask.device
        fh, fw = feat_hw
        lab = _lib.Labels()
        keep = []
        for f in ("gt_bboxes", "gt_labels", "gt_bboxes_3d", "depths", "gt_kpts_2d", "gt_kpts_valid_mask", "mask"):
            t = _need_cuda(label[f], "label." + f)
            if t.dtype != torch.float32:
                raise _lib.MonoconHipError("label.%s must be float32 (collate_fn contract)" % f)
            keep.append(t)
            setattr(lab, f, t.data_ptr())
        T = {}
        for name, kind, dt in self.TARGET_SHAPES:
            if kind == "map3":
                shape = (B, 3, fh, fw)
            elif kind == "map9":
                shape = (B, 9, fh, fw)
            elif kind == 0:
                shape = (B, max_objs)
            elif name == "indices_kpt":
                shape = (B, max_objs * 9)
            else:
                shape = (B, max_objs, kind)
            T[name] = torch.empty(shape, dtype=dt, device=dev)
        st = self._targets_struct(T)
        with torch.cuda.device(dev):
            rc = self.lib.mc_make_targets(self.h, C.byref(lab), B, max_objs, int(pad_hw[0]), int(pad_hw[1]), fh, fw,
                                          C.byref(st), _stream())
        _lib.check(self.h, rc, "mc_make_targets")
        return T

    def losses(self, pred, T, max_objs=30):
        """_get_losses: -> (10,) CUDA tensor in the reference's loss_dict order."""
        heat = _need_cuda(pred["center_heatmap_pred"], "center_heatmap_pred")
        B, _, fh, fw = heat.shape
        arr = (C.c_void_p * _lib.NUM_PREDS)(*[_need_cuda(pred[k], k).data_ptr() for k in PRED_KEYS])
        out = torch.zeros(10, dtype=torch.float32, device=heat.device)
        st = self._targets_struct(T)
        with torch.cuda.device(heat.device):
            rc = self.lib.mc_losses(self.h, arr, C.byref(st), B, max_objs, fh, fw, _ptr(out), _stream())
        _lib.check(self.h, rc, "mc_losses")
        return out

    def losses_backward(self, pred, T, grad_losses, max_objs=30, wrt_pred=False):
        """gradient of sum(grad_losses * losses) wrt the raw 1x1 outputs behind each prediction map, or
        (``wrt_pred``) wrt the prediction maps themselves."""
        heat = _need_cuda(pred["center_heatmap_pred"], "center_heatmap_pred")
        B, _, fh, fw = heat.shape
        arr = (C.c_void_p * _lib.NUM_PREDS)(*[_need_cuda(pred[k], k).data_ptr() for k in PRED_KEYS])
        d = [torch.empty_like(pred[k]) for k in PRED_KEYS]
        darr = (C.c_void_p * _lib.NUM_PREDS)(*[t.data_ptr() for t in d])
        st = self._targets_struct(T)
        g = _need_cuda(grad_losses, "grad_losses")
        with torch.cuda.device(heat.device):
            fn = self.lib.mc_losses_backward_pred if wrt_pred else self.lib.mc_losses_backward
            rc = fn(self.h, arr, C.byref(st), B, max_objs, fh, fw, _ptr(g), darr, _stream())
        _lib.check(self.h, rc, "mc_losses_backward")
        return dict(zip(PRED_KEYS, d))

    # ------------------------------------------------------------------ op level (tests)
    # ------------------------------------------------------------------ input pipeline
    def preprocess(self, images, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), size_divisor=32):
        """Normalize + Pad + ToTensor + collate on the GPU: a list of HWC uint8 / float32 CUDA tensors (sizes may
        differ) -> ((B,3,Hp,Wp) float32 batch, [(Hp, Wp)] * B); Hp, Wp = the largest image rounded up to
        ``size_divisor`` (the reference pads every image of a KITTI batch to the same 384x1248/1280)."""
        if not images:
            raise _lib.MonoconHipError("preprocess: empty image list")
        for t in images:
            _need_cuda(t, "image")
            if t.dim() != 3 or t.shape[2] != 3 or t.dtype not in (torch.uint8, torch.float32) or not t.is_contiguous():
                raise _lib.MonoconHipError("preprocess: images must be contiguous (H,W,3) uint8 or float32 tensors")
        d = int(size_divisor)
        Hp = max((int(t.shape[0]) + d - 1) // d * d for t in images)
        Wp = max((int(t.shape[1]) + d - 1) // d * d for t in images)
        out = torch.empty((len(images), 3, Hp, Wp), dtype=torch.float32, device=images[0].device)
        m = (C.c_double * 3)(*[float(v) for v in mean])
        s = (C.c_double * 3)(*[float(v) for v in std])
        with torch.cuda.device(out.device):
            for b, t in enumerate(images):
                rc = self.lib.mc_preprocess(self.h, _ptr(t), 2 if t.dtype == torch.uint8 else 0, int(t.shape[0]),
                                            int(t.shape[1]), m, s, Hp, Wp, _ptr(out[b]), _stream())
                _lib.check(self.h, rc, "mc_preprocess")
        return out, [(Hp, Wp)] * len(images)

    def preprocess_augmented(self, frames, params, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375)):
        """the image work of the random train augmentations + Normalize + Pad + ToTensor for a batch in one launch
        (mc_preprocess_augmented): ``frames`` (B, Hp, Wp, 3) uint8 -- the decoded frames, zero-padded to the padded size --
        and ``params`` (B, 24) float32 (transforms.DeferredImage writes them) -> (B, 3, Hp, Wp) float32, bit-identical to the
        host transforms."""
        _need_cuda(frames, "frames"); _need_cuda(params, "params")
        if frames.dim() != 4 or frames.shape[3] != 3 or frames.dtype != torch.uint8 or not frames.is_contiguous():
            raise _lib.MonoconHipError("preprocess_augmented: frames must be a contiguous (B,H,W,3) uint8 tensor")
        B, Hp, Wp = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        if tuple(params.shape) != (B, 24) or params.dtype != torch.float32 or not params.is_contiguous():
            raise _lib.MonoconHipError("preprocess_augmented: params must be a contiguous (B,24) float32 tensor")
        out = torch.empty((B, 3, Hp, Wp), dtype=torch.float32, device=frames.device)
        m = (C.c_double * 3)(*[float(v) for v in mean])
        s = (C.c_double * 3)(*[float(v) for v in std])
        with torch.cuda.device(out.device):
            rc = self.lib.mc_preprocess_augmented(self.h, _ptr(frames), _ptr(params), B, Hp, Wp, m, s, Hp, Wp, _ptr(out), _stream())
            _lib.check(self.h, rc, "mc_preprocess_augmented")
        return out

    def set_precision(self, mode):
        """0 = fp32 MFMA, 1 = bf16 MFMA operands (config 3), 2 = fp32 emulated by a 3-way bf16 split, 3 = by a 2-way fp16 split"""
        _lib.check(self.h, self.lib.mc_set_precision(self.h, int(mode)), "mc_set_precision")
        self._sig = None          # panels must be re-packed
        self.precision = int(mode)

    # ---- data parallelism through the C-ABI (csrc/mc_comm.hip)
    def comm_unique_id(self):
        """rank 0: the 128-byte RCCL id the other ranks need for comm_init (ship it by any host-side channel)"""
        buf = C.create_string_buffer(128)
        _lib.check(self.h, self.lib.mc_comm_unique_id(self.h, buf), "mc_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        """collective over all ranks: the handle gets its own RCCL communicator; from then on mc_backward exchanges
        (averages) the gradients itself, overlapped with the backbone's backward"""
        assert len(unique_id) == 128
        with torch.cuda.device(self.device):
            _lib.check(self.h, self.lib.mc_comm_init(self.h, int(rank), int(world), C.c_char_p(unique_id)), "mc_comm_init")
        self.comm_world = int(world)

    def comm_destroy(self):
        _lib.check(self.h, self.lib.mc_comm_destroy(self.h), "mc_comm_destroy")
        self.comm_world = 0

    def comm_set_overlap(self, on):
        _lib.check(self.h, self.lib.mc_comm_set_overlap(self.h, int(bool(on))), "mc_comm_set_overlap")

    def comm_info(self):
        r, w, o, n = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        la = C.c_ulonglong(0)
        path = C.create_string_buffer(256)
        _lib.check(self.h, self.lib.mc_comm_info(self.h, C.byref(r), C.byref(w), C.byref(o), C.byref(n), C.byref(la), path, 256),
                   "mc_comm_info")
        return {"rank": r.value, "world": w.value, "overlap": bool(o.value), "collectives_per_exchange": n.value,
                "launches": la.value, "library": path.value.decode()}

    def comm_exposed_ms(self):
        """how long the last overlapped exchange kept the stream waiting at the end of mc_backward (synchronises)"""
        ms = C.c_float(-1.0)
        _lib.check(self.h, self.lib.mc_comm_exposed_ms(self.h, C.byref(ms)), "mc_comm_exposed_ms")
        return float(ms.value)

    def allreduce_grads(self):
        """every bound '<key>#grad' tensor <- its average over the ranks (for a backward that ran with overlap off)"""
        with torch.cuda.device(self.device):
            _lib.check(self.h, self.lib.mc_allreduce_grads(self.h, _stream()), "mc_allreduce_grads")

    # ---- data-parallel start-up: identical convolution tilings on every rank
    def build_train_plan(self, B, H, W):
        """build (and autotune) the train plan of a shape without running it -- no kernel of the step, no collective"""
        with torch.cuda.device(self.device):
            _lib.check(self.h, self.lib.mc_build_train_plan(self.h, int(B), int(H), int(W)), "mc_build_train_plan")

    def tune_export(self):
        """the autotuned workgroup shapes as a list of ints ([key length, key..., shape id] per entry)"""
        n = C.c_int(0)
        _lib.check(self.h, self.lib.mc_tune_export(self.h, None, 0, C.byref(n)), "mc_tune_export")
        buf = (C.c_int * max(n.value, 1))()
        _lib.check(self.h, self.lib.mc_tune_export(self.h, buf, n.value, C.byref(n)), "mc_tune_export")
        return list(buf[:n.value])

    def tune_import(self, table):
        """adopt another rank's table; returns the number of entries"""
        buf = (C.c_int * max(len(table), 1))(*[int(v) for v in table])
        rc = self.lib.mc_tune_import(self.h, buf, len(table))
        if rc < 0:
            _lib.check(self.h, rc, "mc_tune_import")
        return rc

    def set_conv_cfg(self, cfg):
        """force a workgroup shape of the fused conv (tuning / tests); 0 = automatic."""
        _lib.check(self.h, self.lib.mc_set_conv_cfg(self.h, int(cfg)), "mc_set_conv_cfg")

    def op_conv(self, srcs, weight, stride=1, scale=None, bias=None, residual=None, relu=False):
        """srcs: list of NHWC CUDA tensors (virtual concat); weight OIHW; returns NHWC."""
        for s in srcs:
            _need_cuda(s, "src")
        B, H, W, _ = srcs[0].shape
        Cout, _, k, _ = weight.shape
        Ho = (H + 2 * (k // 2) - k) // stride + 1
        Wo = (W + 2 * (k // 2) - k) // stride + 1
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=srcs[0].device)
        sp = (C.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
        sc = (C.c_int * len(srcs))(*[s.shape[3] for s in srcs])
        with torch.cuda.device(out.device):
            rc = self.lib.mc_op_conv(self.h, sp, sc, len(srcs), B, H, W, _ptr(_need_cuda(weight, "weight")), Cout, k,
                                     stride, _ptr(scale), _ptr(bias), _ptr(residual), int(relu), _ptr(out), _stream())
        _lib.check(self.h, rc, "mc_op_conv")
        return out

    def op_conv_dgrad(self, dy, weight, in_hw, c_off=0, cs=None, stride=1, accumulate_into=None):
        """data gradient wrt input channels [c_off, c_off + cs) of conv2d(x, weight, stride, pad k//2):
        dy NHWC (B,Ho,Wo,Cout) -> NHWC (B,Hin,Win,cs)."""
        _need_cuda(dy, "dy"); _need_cuda(weight, "weight")
        B, Ho, Wo, Cout = dy.shape
        _, cin_total, k, _ = weight.shape
        cs = cin_total - c_off if cs is None else cs
        Hin, Win = in_hw
        out = accumulate_into if accumulate_into is not None else torch.empty((B, Hin, Win, cs), dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device):
            rc = self.lib.mc_op_conv_dgrad(self.h, _ptr(dy), _ptr(weight), B, Hin, Win, cin_total, c_off, cs, Cout, k, stride,
                                           int(accumulate_into is not None), _ptr(out), _stream())
        _lib.check(self.h, rc, "mc_op_conv_dgrad")
        return out

    def op_conv_wgrad(self, srcs, dy, ksize, stride=1):
        """weight gradient of the fused conv: srcs NHWC list, dy NHWC -> (Cout, sum C, k, k)."""
        B, H, W, _ = srcs[0].shape
        Cout = dy.shape[3]
        cin = sum(s.shape[3] for s in srcs)
        dw = torch.empty((Cout, cin, ksize, ksize), dtype=torch.float32, device=dy.device)
        sp = (C.c_void_p * len(srcs))(*[_need_cuda(s, "src").data_ptr() for s in srcs])
        sc = (C.c_int * len(srcs))(*[s.shape[3] for s in srcs])
        with torch.cuda.device(dy.device):
            rc = self.lib.mc_op_conv_wgrad(self.h, sp, sc, len(srcs), B, H, W, _ptr(_need_cuda(dy, "dy")), Cout, ksize,
                                           stride, _ptr(dw), _stream())
        _lib.check(self.h, rc, "mc_op_conv_wgrad")
        return dw

    def op_stem(self, img, weight, scale, bias):
        B, _, H, W = img.shape
        out = torch.empty((B, H, W, 16), dtype=torch.float32, device=img.device)
        with torch.cuda.device(out.device):
            rc = self.lib.mc_op_stem(self.h, _ptr(_need_cuda(img, "img")), B, H, W, _ptr(weight), _ptr(scale),
                                     _ptr(bias), _ptr(out), _stream())
        _lib.check(self.h, rc, "mc_op_stem")
        return out

    def op_maxpool2(self, x):
        B, H, W, Cc = x.shape
        out = torch.empty((B, H // 2, W // 2, Cc), dtype=torch.float32, device=x.device)
        with torch.cuda.device(out.device):
            rc = self.lib.mc_op_maxpool2(self.h, _ptr(_need_cuda(x, "x")), B, H, W, Cc, _ptr(out), _stream())
        _lib.check(self.h, rc, "mc_op_maxpool2")
        return out

    def op_deconv4x4(self, x, weight):
        B, H, W, Cc = x.shape
        out = torch.empty((B, 2 * H, 2 * W, Cc), dtype=torch.float32, device=x.device)
        with torch.cuda.device(out.device):
            rc = self.lib.mc_op_deconv4x4(self.h, _ptr(_need_cuda(x, "x")), B, H, W, Cc, _ptr(weight), _ptr(out),
                                          _stream())
        _lib.check(self.h, rc, "mc_op_deconv4x4")
        return out

    def to_nhwc(self, x):
        B, Cc, H, W = x.shape
        out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=x.device)
        with torch.cuda.device(out.device):
            rc = self.lib.mc_op_nchw_to_nhwc(self.h, _ptr(_need_cuda(x, "x")), B, Cc, H, W, _ptr(out), _stream())
        _lib.check(self.h, rc, "mc_op_nchw_to_nhwc")
        return out

    def to_nchw(self, x):
        B, H, W, Cc = x.shape
        out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
        with torch.cuda.device(out.device):
            rc = self.lib.mc_op_nhwc_to_nchw(self.h, _ptr(_need_cuda(x, "x")), B, Cc, H, W, _ptr(out), _stream())
        _lib.check(self.h, rc, "mc_op_nhwc_to_nchw")
        return out


def p2_inverse(P2):
    """(B,3,4) numpy/torch -> (B,4,4) fp32 inverse of the view-padded projection
    (reference monocon_heads.py:544-546 builds eye(4) with P2 in the top rows and inverts it).
    Host-side 4x4 inverses in float64, rounded to fp32."""
    P2 = np.asarray(P2, dtype=np.float64).reshape(-1, 3, 4)
    out = np.zeros((P2.shape[0], 4, 4), dtype=np.float64)
    for i, p in enumerate(P2):
        v = np.eye(4)
        v[:3, :4] = p
        out[i] = np.linalg.inv(v)
    return out.astype(np.float32)
