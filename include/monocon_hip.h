/*
 * monocon_hip.h -- C-ABI of libmonocon_hip.so, the MI355X (gfx950) implementation of the
 * MonoCon hot path (DLA-34 -> DLAUp -> attentive-norm dense heads -> decode).
 *
 * The reference (2gunsu/monocon-pytorch) is pure Python over torch.nn and defines no FFI;
 * these entry points are what a ctypes binding for its hot path binds instead of the
 * torch.nn modules.  Each entry cites the reference code it replaces (paths relative to
 * the reference repository root).  Conventions:
 *   - plain C types only: raw device pointers, sizes, a hipStream_t passed as void*;
 *   - every function returns 0 on success, <0 on error (text via mc_last_error);
 *   - all work is enqueued asynchronously on the given stream; no hidden device sync;
 *   - the caller owns every tensor it passes and keeps it alive until the stream reaches
 *     the op; the handle owns its workspace, packed weights and plan cache;
 *   - external tensors use the reference's layouts (NCHW fp32 images / prediction maps,
 *     OIHW fp32 weights); NHWC is internal.
 * One handle per process per GPU; not thread-safe.
 */
#ifndef MONOCON_HIP_H
#define MONOCON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mc_handle mc_handle;

enum { MC_F32 = 0, MC_I64 = 1 };

/* One named tensor of the model's state_dict (names = the reference's state_dict keys,
 * e.g. "backbone.level2.tree1.conv1.weight"; "<key>#grad" binds the gradient buffer). */
typedef struct mc_tensor_desc {
    const char *name;
    void *ptr;          /* device pointer */
    int64_t numel;
    int32_t dtype;      /* MC_F32 | MC_I64 */
} mc_tensor_desc;

/* Order of the ten prediction maps == reference pred_dict order
 * (model/dense_heads/monocon_heads.py:190-200); channels 3,9,2,2,2,18,3,2,12,12. */
enum {
    MC_PRED_CENTER_HEATMAP = 0, MC_PRED_KPT_HEATMAP, MC_PRED_WH, MC_PRED_OFFSET,
    MC_PRED_KPT_HEATMAP_OFFSET, MC_PRED_CENTER2KPT_OFFSET, MC_PRED_DIM, MC_PRED_DEPTH,
    MC_PRED_ALPHA_CLS, MC_PRED_ALPHA_OFFSET, MC_NUM_PREDS
};

/* ---- lifetime ---------------------------------------------------------------------- */
int mc_create(int device, mc_handle **out);
int mc_destroy(mc_handle *h);
const char *mc_last_error(mc_handle *h);      /* h may be NULL: last create() error */
int mc_version(void);

/* ---- parameters -------------------------------------------------------------------- */
/* Replaces nn.Module parameter ownership: MonoConDetector.state_dict()/load_state_dict()
 * (model/detector/monocon_detector.py:80-82, engine/base_engine.py:178,208).  Binds the
 * caller-owned master tensors.  A stage can be used once all keys of its group (backbone. /
 * neck. / head.) are bound; mc_forward_infer needs all 449. */
int mc_bind_params(mc_handle *h, const mc_tensor_desc *descs, int n);
/* Re-derive the device-side packed weights (OIHW -> K-major MFMA panels) and, for eval
 * mode, the folded BatchNorm scale/shift.  Call after binding and after any parameter
 * update.  train_mode=0: BN folded from running statistics (model.eval()). */
int mc_pack_params(mc_handle *h, int train_mode, void *stream);

/* ---- inference forward -------------------------------------------------------------
 * Replaces MonoConDetector.forward in eval mode: neck(backbone(img))[0] then
 * MonoConDenseHeads.forward_test (model/detector/monocon_detector.py:53-66,85-87;
 * model/backbone/dla.py:273-278; model/backbone/dla_neck.py:136-143;
 * model/dense_heads/monocon_heads.py:165-200; model/norm/attentive_norm.py:154-164).
 * img: (B,3,H,W) fp32 NCHW, H and W multiples of 32.  preds[i]: (B,C_i,H/4,W/4) fp32 NCHW.
 * feat_nchw (optional, may be NULL): (B,64,H/4,W/4) copy of the neck output. */
int mc_forward_infer(mc_handle *h, const float *img, int B, int H, int W,
                     float *const preds[MC_NUM_PREDS], float *feat_nchw, void *stream);

/* ---- stage-level forwards (eval mode) ---------------------------------------------------
 * Sub-module parity for callers that use the reference's modules on their own; the detector
 * itself uses mc_forward_infer (no NCHW round trips between stages).  (H, W) = image size.
 * mc_backbone_forward replaces DLA.forward (model/backbone/dla.py:273-278): levels[i] is
 *   (B, {16,32,64,128,256,512}[i], H/{1,2,4,8,16,32}[i], W/...) NCHW or NULL to skip it.
 * mc_neck_forward replaces DLAUp.forward (model/backbone/dla_neck.py:136-143): reads
 *   levels[2..5], writes feat (B,64,H/4,W/4).
 * mc_head_forward replaces MonoConDenseHeads._get_predictions
 *   (model/dense_heads/monocon_heads.py:165-200). */
int mc_backbone_forward(mc_handle *h, const float *img, int B, int H, int W,
                        float *const levels[6], void *stream);
int mc_neck_forward(mc_handle *h, const float *const levels[6], int B, int H, int W,
                    float *feat, void *stream);
int mc_head_forward(mc_handle *h, const float *feat, int B, int H, int W,
                    float *const preds[MC_NUM_PREDS], void *stream);

/* ---- decode -------------------------------------------------------------------------
 * Replaces MonoConDenseHeads.decode_heatmap + _get_bboxes origin shift
 * (model/dense_heads/monocon_heads.py:313-329,379-558) and utils/tensor_ops.py:17-31
 * (get_local_maximum, get_topk_from_heatmap).  k x k local-maximum filter (k = 3 unless
 * mc_set_local_maximum_kernel changed it), per-image top-K
 * over the flattened C*H*W map (ties: score desc, flat index asc), gathers, box assembly.
 * preds: the ten NCHW maps (only those the reference reads are touched); P2: (B,3,4);
 * P2inv: (B,4,4) inverse of the view-padded projection (monocon_heads.py:544-546).
 * Outputs (all dense, caller-allocated):
 *   scores (B,K) f32 raw heat-map peaks; flat_index (B,K) i64 into C*H*W; cls (B,K) i64;
 *   box2d (B,K,5) = x1,y1,x2,y2,score*sigma; box3d (B,K,7) = x,y+h/2,z,dim(3),rot_y;
 *   keep_localmax (B,C,H,W) u8 or NULL; keep_thr (B,K) u8 = box2d[...,4] > thr. */
/* test_config['local_maximum_kernel'] (model/detector/monocon_detector.py:18, utils/tensor_ops.py:17 `kernel`): the window
 * of the filter, odd, 1 .. 31; a setting of the handle (default 3), read by every later mc_decode. */
int mc_set_local_maximum_kernel(mc_handle *h, int kernel);
int mc_decode(mc_handle *h, const float *const preds[MC_NUM_PREDS], const float *P2,
              const float *P2inv, int B, int C, int H, int W, int K, float thr,
              float pad_h, float pad_w, float *scores, int64_t *flat_index, int64_t *cls,
              float *box2d, float *box3d, uint8_t *keep_localmax, uint8_t *keep_thr,
              void *stream);

/* ---- training: targets and losses ---------------------------------------------------------
 * Label tensors exactly as the reference's collate_fn delivers them (all float32, device):
 * gt_bboxes (B,M,4), gt_labels (B,M), gt_bboxes_3d (B,M,7), depths (B,M), gt_kpts_2d (B,M,18),
 * gt_kpts_valid_mask (B,M,9), mask (B,M)   (dataset/monocon_dataset.py:160-171). */
typedef struct mc_labels {
    const float *gt_bboxes, *gt_labels, *gt_bboxes_3d, *depths, *gt_kpts_2d, *gt_kpts_valid_mask, *mask;
} mc_labels;
/* The 15 target tensors of reference utils/target_generator.py:152-177 (caller-allocated, device):
 * heat-maps (B,3,h,w)/(B,9,h,w) f32; regression targets (B,M,{2,2,3,1,1,1,18,18}) f32; indices (B,M)
 * and indices_kpt (B,9M) i64; mask_target (B,M) u8/bool; the two keypoint masks (B,M,18) f32. */
typedef struct mc_targets {
    float *center_heatmap_target, *wh_target, *offset_target, *dim_target, *alpha_cls_target,
        *alpha_offset_target, *depth_target, *center2kpt_offset_target, *kpt_heatmap_target,
        *kpt_heatmap_offset_target;
    int64_t *indices, *indices_kpt;
    uint8_t *mask_target;
    float *mask_center2kpt_offset, *mask_kpt_heatmap_offset;
} mc_targets;
/* Replaces TargetGenerator.__call__ (utils/target_generator.py:30-138) and the gaussian helpers
 * (utils/tensor_ops.py:62-125): one launch instead of a Python loop over batch x objects x 9. */
int mc_make_targets(mc_handle *h, const mc_labels *labels, int B, int max_objs, int pad_h,
                    int pad_w, int feat_h, int feat_w, const mc_targets *targets, void *stream);
/* Replaces MonoConDenseHeads._get_losses + losses/{l1,dim,depth,focal,cross_entropy}_loss.py (model/dense_heads/monocon_heads.py:203-310):
 * losses[10] (device) in the reference's loss_dict order: center_heatmap, wh, offset, dim,
 * center2kpt_offset, kpt_heatmap, kpt_heatmap_offset, alpha_cls, alpha_reg, depth. */
int mc_losses(mc_handle *h, const float *const preds[MC_NUM_PREDS], const mc_targets *targets,
              int B, int max_objs, int feat_h, int feat_w, float *losses, void *stream);
/* Gradient of sum_i grad_losses[i] * loss_i (grad_losses: device [10]) with respect to the RAW
 * 1x1-conv outputs behind each prediction map (i.e. through sigmoid+clamp / the depth transform). */
int mc_losses_backward(mc_handle *h, const float *const preds[MC_NUM_PREDS],
                       const mc_targets *targets, int B, int max_objs, int feat_h, int feat_w,
                       const float *grad_losses, float *const dpreds[MC_NUM_PREDS], void *stream);
/* The same gradient with respect to the prediction maps THEMSELVES (the tensors _get_losses receives:
 * post sigmoid+clamp heat-maps, transformed depth) -- what autograd hands back from
 * MonoConDenseHeads._get_losses(pred_dict, target_dict) (model/dense_heads/monocon_heads.py:203-310). */
int mc_losses_backward_pred(mc_handle *h, const float *const preds[MC_NUM_PREDS],
                            const mc_targets *targets, int B, int max_objs, int feat_h, int feat_w,
                            const float *grad_losses, float *const dpreds[MC_NUM_PREDS], void *stream);

/* ---- training step ---------------------------------------------------------------------
 * mc_forward_train replaces `pred_dict, loss_dict = model(data_dict)` in train mode
 * (model/detector/monocon_detector.py:53-61 -> MonoConDenseHeads.forward_train,
 * model/dense_heads/monocon_heads.py:150-157): BatchNorm / AttnBN use batch statistics and update
 * their running buffers in place (the bound tensors), targets are generated from the labels,
 * preds[10] (NCHW) and losses[10] (device floats, loss_dict order) are written.
 * mc_backward replaces total_loss.backward() (engine/monocon_engine.py:86): grad_losses is the
 * device [10] upstream gradient of each loss (ones for `sum(loss_dict.values())`); gradients of
 * every live parameter are WRITTEN to the tensors bound as "<key>#grad" (the six parameters the
 * reference never back-propagates into -- SURVEY 8a quirk (i) -- are not touched).
 * Data parallelism: every rank runs this on its own shard; averaging the "#grad" tensors across
 * ranks precedes the optimizer -- mc_backward does it itself once mc_comm_init gave the handle a communicator (see
 * "data parallelism" below), otherwise the host does (torch.distributed, or mc_allreduce_grads).  The packed weight panels are refreshed first when mc_bind_params or mc_clip_adamw_step made them
 * stale; after any other in-place parameter update call mc_pack_params yourself. */
int mc_forward_train(mc_handle *h, const float *img, const mc_labels *labels, int B, int H, int W,
                     int max_objs, float *const preds[MC_NUM_PREDS], float *losses, void *stream);
int mc_backward(mc_handle *h, const float *grad_losses, void *stream);
/* The train plan keeps ONE set of saved activations: mc_backward always differentiates the latest
 * mc_forward_train of the handle.  *generation = id of that forward (0 before the first one; every
 * forward gets a new id).  A caller that may interleave forwards and backwards (gradient accumulation
 * over micro-batches, backward on an older loss -- autograd allows both, engine/monocon_engine.py:84-86
 * never does) records the id after its forward and compares before mc_backward; the Python binding
 * raises on a mismatch instead of back-propagating through the wrong activations. */
int mc_train_generation(mc_handle *h, unsigned long long *generation);
/* The dense heads on their own: replaces MonoConDenseHeads.forward_train(feat, data_dict)
 * (model/dense_heads/monocon_heads.py:150-157: target generation, train-mode predictions of the nine heads --
 * Conv3x3 + AttnBatchNorm2d on batch statistics + ReLU + Conv1x1 -- and the ten losses) for a caller that runs
 * its own backbone / neck.  feat: (B,64,pad_h/4,pad_w/4) fp32 NCHW, the neck output; only the "head." keys need
 * to be bound (and their "#grad" buffers).  mc_head_backward writes the head parameters' gradients and, when
 * grad_feat is not NULL, d(sum_i grad_losses[i] * loss_i)/d feat as (B,64,pad_h/4,pad_w/4) NCHW. */
int mc_head_forward_train(mc_handle *h, const float *feat, const mc_labels *labels, int B, int pad_h, int pad_w,
                          int max_objs, float *const preds[MC_NUM_PREDS], float *losses, void *stream);
int mc_head_backward(mc_handle *h, const float *grad_losses, float *grad_feat, void *stream);
/* Debugging aid: activation (which=0) or gradient (which=1) of node `node` of the train plan as NCHW. */
int mc_train_debug_node(mc_handle *h, int node, int which, float *out_nchw, int dims[4], void *stream);

/* ---- data parallelism -----------------------------------------------------------
 * The reference is single-GPU (README.MD:11,15: "multi-GPU training is not supported"); these entry points are the
 * MI355X-side addition BASELINE.json's north_star asks for: one process per GPU, every rank a full replica, the
 * gradients averaged over the ranks once per step on RCCL over xGMI, then identical clip + AdamW on every rank.  They
 * stand where torch's DistributedDataParallel would wrap `self.model` in engine/monocon_engine.py:28-32.
 * mc_comm_unique_id: rank 0 creates the 128-byte RCCL id and hands it to the other ranks by any host-side channel.
 * mc_comm_init: collective over all ranks (ncclCommInitRank); the handle owns the communicator and its stream.
 * mc_allreduce_grads: every tensor bound as "<key>#grad" <- its average over the ranks, in place, on `stream` (one
 *   ncclAllReduce per gradient bucket when the tensors form dense ranges, see csrc/mc_comm.hip).
 * With a communicator in place (and overlap on, the default) mc_backward performs this exchange itself, bucket by
 * bucket on the communicator's stream while the rest of the backward runs, and returns with `stream` waiting for it:
 * do NOT call mc_allreduce_grads again after such a backward.  mc_comm_set_overlap(h, 0) turns that off.
 * mc_comm_info: rank / world / overlap flag, the number of collectives one exchange issues for the current binding,
 * the all-reduce calls issued so far, and the RCCL library in use.  mc_comm_exposed_ms: how long the last overlapped
 * exchange kept `stream` waiting at the end of mc_backward (synchronises). */
int mc_comm_unique_id(mc_handle *h, void *id128);
int mc_comm_init(mc_handle *h, int rank, int world, const void *id128);
int mc_comm_destroy(mc_handle *h);
int mc_comm_set_overlap(mc_handle *h, int on);
int mc_comm_info(mc_handle *h, int *rank, int *world, int *overlap, int *n_collectives, unsigned long long *launches,
                 char *lib_path, int lib_path_len);
int mc_comm_exposed_ms(mc_handle *h, float *ms);
int mc_allreduce_grads(mc_handle *h, void *stream);
/* Start-up of a data-parallel job (no reference counterpart: README.MD:11,15).  The workgroup shape of every
 * convolution is chosen by timing when a plan is built; ranks that tune concurrently can pick different shapes (results
 * stay bit-identical, step times do not).  So rank 0 builds its plan first -- mc_build_train_plan builds and tunes the
 * train plan of a shape without running a kernel of the step or a collective -- exports the table (mc_tune_export: ints,
 * per entry [key length, key..., shape id]; buf == NULL returns the size in *n_ints), and every other rank imports it
 * (mc_tune_import, returns the number of entries or -1) before it builds its own plan.
 * mc_comm_init runs ncclCommInitRank under a watchdog (MONOCON_HIP_COMM_TIMEOUT_S, default 60): if not every rank
 * arrives, the call fails with a message naming this rank instead of hanging. */
int mc_build_train_plan(mc_handle *h, int B, int H, int W);
int mc_tune_export(mc_handle *h, int *buf, int cap, int *n_ints);
int mc_tune_import(mc_handle *h, const int *buf, int n_ints);

/* ---- optimizer ---------------------------------------------------------------------------
 * Replaces clip_grad_norm_(max_norm, L2) + torch.optim.AdamW.step (engine/monocon_engine.py:94-102;
 * AdamW(lr 2.25e-4, wd 1e-5, betas (0.95,0.99)) :39-43).  mc_optim_bind registers n parameter
 * tensors with their gradient and moment buffers (caller-owned fp32 device tensors);
 * mc_clip_adamw_step computes the global gradient norm over them, the clip coefficient
 * min(1, max_norm/(norm+1e-6)) (max_norm <= 0: no clipping) and applies one AdamW update in
 * place.  step is 1-based.  out_norm (device float, optional) receives the pre-clip norm. */
int mc_optim_bind(mc_handle *h, int n, float *const params[], float *const grads[],
                  float *const exp_avg[], float *const exp_avg_sq[], const int64_t numel[]);
int mc_clip_adamw_step(mc_handle *h, double lr, double beta1, double beta2, double eps,
                       double weight_decay, double max_norm, int step, float *out_norm, void *stream);

/* ---- op-level entry points (unit parity tests; same kernels the forward uses) -------
 * Fused convolution, NHWC fp32: out = act(conv(cat(src...)) * scale + bias + residual).
 * Replaces nn.Conv2d + nn.BatchNorm2d(eval) + ReLU (+ torch.cat, + residual add) of
 * BasicBlock / Root / Conv2dBlock (model/backbone/dla.py:34-51,124-132,
 * model/backbone/dla_neck.py:34-38).  weight: OIHW fp32 (O, sum C_i, k, k), k in {1,3},
 * stride in {1,2}, pad = k/2; scale/bias/residual may be NULL. */
int mc_op_conv(mc_handle *h, const float *const src[], const int src_channels[], int nsrc,
               int B, int Hin, int Win, const float *weight_oihw, int Cout, int ksize,
               int stride, const float *scale, const float *bias, const float *residual,
               int relu, float *out, void *stream);
/* Weight gradient of the same convolution: dW (O, sum C_i, k, k) = sum over pixels of dY x X
 * (autograd's conv2d weight backward); dy: NHWC (B,Hout,Wout,Cout). */
int mc_op_conv_wgrad(mc_handle *h, const float *const src[], const int src_channels[], int nsrc,
                     int B, int Hin, int Win, const float *dy, int Cout, int ksize, int stride,
                     float *dw_oihw, void *stream);
/* 7x7 stem: NCHW (B,3,H,W) in -> NHWC (B,H,W,16) out (model/backbone/dla.py:231-234). */
int mc_op_stem(mc_handle *h, const float *img_nchw, int B, int H, int W,
               const float *weight_oihw, const float *scale, const float *bias, float *out,
               void *stream);
/* 2x2/2 max-pool, NHWC (model/backbone/dla.py:178-179). */
int mc_op_maxpool2(mc_handle *h, const float *in, int B, int H, int W, int C, float *out,
                   void *stream);
/* depthwise ConvTranspose2d k=4,s=2,p=1, NHWC; weight (C,1,4,4)
 * (model/backbone/dla_neck.py:58-65). */
int mc_op_deconv4x4(mc_handle *h, const float *in, int B, int H, int W, int C,
                    const float *weight, float *out, void *stream);
/* Data gradient of the same convolution wrt ONE source of its virtual concat (autograd's conv2d input
 * backward): dx (B,Hin,Win,Cs) NHWC = conv_transpose(dy (B,Hout,Wout,Cout), W[:, c_off:c_off+Cs]); accumulate != 0
 * adds to dx.  Runs exactly what the train plan launches: the fused conv kernel on a transposed / flipped weight
 * panel for stride 1, four output-parity window convs for stride 2. */
int mc_op_conv_dgrad(mc_handle *h, const float *dy, const float *weight_oihw, int B, int Hin, int Win, int CinTotal,
                     int c_off, int Cs, int Cout, int ksize, int stride, int accumulate, float *dx, void *stream);
/* layout helpers */
int mc_op_nchw_to_nhwc(mc_handle *h, const float *in, int B, int C, int H, int W, float *out,
                       void *stream);
int mc_op_nhwc_to_nchw(mc_handle *h, const float *in, int B, int C, int H, int W, float *out,
                       void *stream);

/* ---- input pipeline -----------------------------------------------------
 * Replaces Normalize + Pad + ToTensor of the reference's test / train transform lists
 * (transforms/default_transforms.py:375-452, dataset/monocon_dataset.py:32-33,39-40) for one image that is
 * already on the device: HWC (dtype 0 = float32, 2 = uint8; 3 channels) -> CHW float32 of shape
 * (3, pad_h, pad_w), value (x - mean[c]) / std[c] evaluated in float64 and rounded once to float32 exactly as
 * numpy + torch.Tensor() do, zeros outside (H, W).  Write B images into one (B,3,pad_h,pad_w) batch by calling
 * it B times with out_chw advanced by 3*pad_h*pad_w. */
int mc_preprocess(mc_handle *h, const void *img_hwc, int dtype, int H, int W, const double mean[3],
                  const double std[3], int pad_h, int pad_w, float *out_chw, void *stream);

/* The IMAGE work of the reference's random train augmentations -- PhotometricDistortion, RandomShift, RandomHorizontalFlip,
 * RandomCrop3D (transforms/default_transforms.py:27-372; dataset/monocon_dataset.py:39-46 lists them for the 'train' split) --
 * followed by Normalize + Pad + ToTensor, for B raw frames in one launch.  The host side (the loader's workers) still draws
 * the random numbers and moves labels and calibration; it ships each decoded uint8 frame, zero-padded to (src_h, src_w, 3),
 * with 24 float32 parameters: 0 H, 1 W (the frame's own size), 2 flags, 3 brightness delta, 4 contrast before the HSV stage,
 * 5 saturation, 6 hue delta, 7 contrast after it, 8-10 channel permutation (on BGR), 11 shift x, 12 shift y, 13-16 crop window
 * x0 y0 x1 y1.  flags: 1 colour stage present (it is a float32 BGR -> HSV -> BGR round trip even when nothing is drawn),
 * 2 brightness, 4 contrast before, 8 saturation, 16 hue, 32 contrast after, 64 permutation, 128 shift, 256 flip, 512 window.
 * frames_hwc: (B, src_h, src_w, 3) uint8; params: (B, 24) float32; out_bchw: (B, 3, pad_h, pad_w) float32, all on the device.
 * Every value is bit-identical to the host pipeline's (float32 operation by operation, then the float64 normalisation of
 * mc_preprocess); with flags 0 it is mc_preprocess of a uint8 frame. */
int mc_preprocess_augmented(mc_handle *h, const unsigned char *frames_hwc, const float *params, int B, int src_h, int src_w,
                            const double mean[3], const double std[3], int pad_h, int pad_w, float *out_bchw, void *stream);

/* ---- KITTI AP evaluation (last row) -------------------------------------------------------------
 * Device part: pairwise overlaps of rotated boxes.  All pointers are device pointers; results are row-major (N, K).
 *
 * mc_rotate_iou_eval replaces rotate_iou_gpu_eval / rotate_iou_kernel_eval, the reference's only GPU kernel
 * (engine/kitti_eval/rotate_iou.py:280-378): boxes (N,5) and query_boxes (K,5) float32 [cx, cy, dx, dy, angle], angle
 * clockwise-positive (camera-frame bird's-eye view); criterion -1: IoU, 0: intersection / area(query box),
 * 1: intersection / area(box), any other value: the bare intersection area (rotate_iou.py:252-277).  The host wrapper's
 * H2D / D2H copies are the caller's business here (engine/kitti_eval/rotate_iou.py of this package does them).
 *
 * mc_box3d_overlap replaces d3_box_overlap (engine/kitti_eval/eval.py:128-164: the same GPU kernel asked for bare
 * areas, then a numba CPU pass for the height overlap) in ONE launch: boxes (N,7), query_boxes (K,7) float64 camera-frame
 * [x, y, z, l, h, w, ry] (y = bottom face, pointing down); criterion -1: 3D IoU, 0: / volume(box), 1: / volume(query). */
int mc_rotate_iou_eval(mc_handle *h, const float *boxes, const float *query_boxes, long long N, long long K,
                       int criterion, float *iou, void *stream);
int mc_box3d_overlap(mc_handle *h, const double *boxes, const double *query_boxes, long long N, long long K,
                     int criterion, double *overlap, void *stream);

/* Host part (no handle, no device: runs on any box): the loops the reference JIT-compiles with numba.
 * mc_kitti_image_overlap = image_box_overlap (eval.py:90-119): axis-aligned (x1,y1,x2,y2) boxes, criterion -1 IoU,
 * 0: / area(box), 1: / area(query), else the bare intersection.
 * mc_kitti_statistics_part runs compute_statistics_jit (eval.py:167-285) over the frames of one "part" (frames whose
 * boxes were concatenated, eval.py:347-422): overlaps is the part's (sum dt_nums, sum gt_nums) matrix, gt_datas
 * (sum gt, 5) = bbox + alpha, dt_datas (sum dt, 6) = bbox + alpha + score, dontcares (sum dc, 4), ignore flags as
 * produced by clean_data (eval.py:35-87).  mode 0: scores of the true positives without false-positive accounting,
 * appended to scores_out (capacity sum gt_nums) -- the first pass of eval_class (eval.py:490-505) -- and, when pr is
 * not NULL, pr[0..3] += that pass's own (tp, 0, fn, 0); mode 1:
 * fused_compute_statistics (eval.py:297-344), pr[n_thresholds][4] += (tp, fp, fn, similarity).  Returns 0, or -1 on a
 * bad argument. */
int mc_kitti_image_overlap(const double *boxes, long long N, const double *query_boxes, long long K, int criterion,
                           double *overlap);
int mc_kitti_statistics_part(int mode, const double *overlaps, long long n_frames, const long long *gt_nums,
                             const long long *dt_nums, const long long *dc_nums, const double *gt_datas,
                             const double *dt_datas, const double *dontcares, const long long *ignored_gts,
                             const long long *ignored_dets, int metric, double min_overlap, const double *thresholds,
                             long long n_thresholds, int compute_aos, double *pr, double *scores_out,
                             long long *n_scores);

/* Device memory a plan of the handle needs for one input shape, WITHOUT building it (lets the caller size
 * the batch against the 288 GB of HBM before the first step; the plan builder runs dry -- nothing is allocated or
 * launched).  mode 0: inference forward (mc_forward_infer), 1: train step (mc_forward_train + mc_backward), 2: heads-only
 * train step (mc_head_forward_train).  The parameters the mode needs must be bound and packed.  Replaces nothing in
 * the reference (torch's caching allocator grows on demand, engine/base_engine.py has no such query). */
int mc_query_workspace(mc_handle *h, int B, int H, int W, int mode, size_t *bytes);

/* Measurement aid for bench.py: re-runs the launches of the last mc_forward_train + mc_backward
 * `iters` times with a HIP event pair around every launch group on `stream` and returns, per
 * kernel family k (0 = everything else, 1 = conv_mfma_kernel: forward convs + data gradients,
 * 2 = wgrad_mfma_kernel: weight gradients), the summed duration ms[k], the algorithmic FLOPs
 * flops[k] and bytes[k] (every operand once) and the number of launch groups, averaged per step.  (Mutates BN running
 * statistics like any train-mode forward.) */
int mc_profile_train(mc_handle *h, int iters, double ms[3], double flops[3], double bytes[3], int launches[3],
                     void *stream);
/* Arithmetic of the convolutions (forward + data gradients).  0 (default): fp32 MFMA -- the parity
 * path.  1: both MFMA operands rounded to bf16 on their way into the matrix pipe, fp32 accumulation,
 * fp32 activations / master weights / BN statistics / losses (BASELINE config 3; not within the 1e-4
 * parity tolerance).  2: fp32 EMULATED on the bf16 pipe -- both operands split into three bf16 pieces, six
 * partial products per multiply accumulated in fp32; as close to the fp64 reference as the fp32 MFMA path (same
 * parity tolerances), ~2.7x its matrix rate.  3: fp32 EMULATED on the fp16 pipe -- every operand tensor scaled by the
 * power of two its max |x| dictates, split into two fp16 pieces, three partial products per multiply; same parity
 * tolerances, half the matrix work of mode 2.  (Round 4's experimental mode 4 -- mode 3 on activations stored pre-split
 * and DMA-staged -- was retired in round 5: it lost in the step and its weight gradients were not bit-reproducible beside the
 * weight-gradient stream; mode 4 is an error.)
 * Re-pack (mc_pack_params) before the next forward. */
int mc_set_precision(mc_handle *h, int mode);
/* Tuning / test aid: force one workgroup shape of the fused convolution (ids in
 * csrc/conv_mfma.h: 1..8 = pixel x channel tile, +16 = wave-specialised kernel, +64 = the weight-resident persistent kernel
 * (csrc/conv_wres.hip) wherever a launch is eligible, 32 = the LDS-free kernel for 16/32-channel 3x3 layers; 0 = automatic)
 * for mc_op_conv and for every layer of plans built afterwards that has no fixed shape. */
int mc_set_conv_cfg(mc_handle *h, int cfg);
/* Tuning aid: average duration (ms) of `iters` launches of one fused-conv shape on random data.
 * cfg: workgroup shape id (0 = the library's own choice; see csrc/conv_mfma.h ConvCfgId). */
int mc_bench_conv(mc_handle *h, int B, int Hin, int Win, int nsrc, const int src_channels[],
                  int Cout, int ksize, int stride, int cfg, int iters, float *ms_avg);

/* Tuning aid: sustained TFLOP/s of back-to-back v_mfma_f32_32x32x2_f32 with no memory traffic. */
int mc_bench_mfma_peak(mc_handle *h, int waves_per_simd, int iters, float *tflops);

/* ---- introspection ------------------------------------------------------------------ */
/* Bytes of handle-owned device memory (workspace + packed weights). */
size_t mc_workspace_bytes(mc_handle *h);
/* Algorithmic FLOPs / HBM bytes of one inference forward at (B,H,W), computed from the
 * layer table (SURVEY §8d fusion model): [0] = the fused conv-MFMA launches, [1] = all other
 * launches (stem, pools, deconvs, head passes). */
int mc_forward_cost(mc_handle *h, int B, int H, int W, double flops[2], double bytes[2]);
/* Time the launches of the last forward plan by kind with HIP events on `stream`
 * (bench.py roofline leg): runs the cached plan `iters` times.  out_ms: [0] all conv-MFMA
 * kernels, [1] everything else, [2] whole forward; out_n: launches per forward by kind. */
int mc_profile_forward(mc_handle *h, int iters, float out_ms[3], int out_n[3], void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MONOCON_HIP_H */
