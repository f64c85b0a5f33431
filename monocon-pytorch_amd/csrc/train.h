// Training-side launch argument blocks (targets, losses, BatchNorm train mode, backward, optimizer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "conv_mfma.h"

namespace mc {

// ---- target generator (reference utils/target_generator.py:30-177)
struct TargetArgs {
    // labels, fp32 (B, max_objs, .) as collate_fn delivers them (dataset/monocon_dataset.py:160-171)
    const float *gt_bboxes, *gt_labels, *gt_bboxes_3d, *depths, *gt_kpts_2d, *gt_kpts_valid, *mask;
    int B, max_objs, num_kpt, num_classes, fh, fw;
    float h_ratio, w_ratio;
    // outputs (zero-filled by the caller before the launch)
    float *center_heatmap, *kpt_heatmap;                 // (B,3,fh,fw), (B,9,fh,fw)
    float *wh, *offset, *dim, *alpha_cls, *alpha_offset, *depth, *c2k, *kho;
    long long *indices, *indices_kpt;                    // (B,30), (B,270)
    uint8_t *mask_target;                                // (B,30)
    float *mask_c2k, *mask_kho;                          // (B,30,18)
};
hipError_t launch_make_targets(const TargetArgs &a, hipStream_t st);

// ---- losses
hipError_t launch_focal(const float *p, const float *t, size_t n, float *partial, float *loss_out, float *aux,
                        hipStream_t st);
hipError_t launch_focal_grad(const float *p, const float *t, size_t n, const float *aux, const float *gscale, int gidx,
                             float *dlogit, hipStream_t st, int wrt_pred = 0);
int focal_partial_floats();

struct GatherLossArgs {
    const float *pred[10];       // NCHW prediction maps (pred order)
    float *dpred[10];            // NCHW gradient wrt the raw 1x1 outputs (mode 1), zero-filled
    const long long *indices, *indices_kpt;
    const uint8_t *mask_target;
    const float *wh, *offset, *dim, *alpha_cls, *alpha_offset, *depth, *c2k, *kho, *mask_c2k, *mask_kho;
    int B, max_objs, HW;
    float *losses;               // [10] LOSS order
    float *aux;                  // [1] number of valid objects
    const float *gscale;         // [10] upstream gradient of each loss (mode 1)
    int wrt_pred;                // mode 1: 0 = gradient wrt the raw 1x1 outputs, 1 = wrt the prediction maps themselves
    double *partial;             // workspace of gathered_loss_ws_doubles() doubles (per-workgroup sums, then the totals)
    int nwg;                     // (set by the launcher)
};
size_t gathered_loss_ws_doubles();
hipError_t launch_gathered_losses(const GatherLossArgs &a, int mode, hipStream_t st);

// ---- batched weight packing / BatchNorm folding (pack_batch.hip)
struct PackJobDesc {
    const float *w;        // master weight, OIHW fp32
    float *dst32;          // fp32 panel or null
    void *dst16;           // bf16 piece planes or null
    int kind;              // 0: forward panel, 1: data-gradient panel
    int Cout, Cin, k;      // forward: shape of w.  data gradient: Cin = channels of the source (Cs)
    int CinTotal;          // forward: channels of the panel (virtual concat).  data gradient: Cin of the forward weight
    int CoutP;             // forward: padded columns.  data gradient: CoutPad (K of the data gradient)
    int n_off, c_off;      // forward: column / channel offset in the panel.  data gradient: c_off of the source
    int CsP, cls, nsplit;  // data gradient: padded source channels, parity class (-1: stride 1); pieces: 1 / 3 bf16, 2 fp16
    int block_begin, nblocks;
    unsigned *amax;        // nsplit == 2: slot holding max |w| of the master weight(s) behind the panel (conv_mfma.h)
};
struct PackBatch {
    std::vector<PackJobDesc> jobs;
    PackJobDesc *dev = nullptr;
    int total_blocks = 0;
    bool uploaded = false;
    void clear();
    void add(PackJobDesc j);
    // with_amax: first fold max |w| of every forward job into its slot (slots zeroed by the caller)
    hipError_t launch(hipStream_t st, bool with_amax = false);
    PackBatch() = default;
    PackBatch(const PackBatch &) = delete;
    PackBatch &operator=(const PackBatch &) = delete;
    ~PackBatch();
};
struct FoldJobDesc {
    const float *g, *b, *rm, *rv;
    float eps;
    int C;
    float *scale, *shift;
};
struct FoldBatch {
    std::vector<FoldJobDesc> jobs;
    FoldJobDesc *dev = nullptr;
    bool uploaded = false;
    int maxC = 0;
    void clear();
    void add(const FoldJobDesc &j);
    hipError_t launch(hipStream_t st);
    FoldBatch() = default;
    FoldBatch(const FoldBatch &) = delete;
    FoldBatch &operator=(const FoldBatch &) = delete;
    ~FoldBatch();
};

// ---- fused clip + AdamW (reference engine/monocon_engine.py:94-102)
struct OptTensor { float *p, *g, *m, *v; };
struct OptChunk { int tensor, begin, count; };
struct AdamHyper { float decay, beta1, one_minus_beta1, beta2, one_minus_beta2, sqrt_bc2, eps, step_size; };
hipError_t launch_clip_adamw(const OptTensor *tab, const OptChunk *chunks, int nchunks, float *partial, float *normcoef,
                             float max_norm, const AdamHyper &hp, hipStream_t st);
int opt_partial_floats();

// ---- train-mode element-wise / reduction kernels (kernels_train.hip)
int chan_reduce_blocks(int B, int rows_per_img);
hipError_t launch_chan_reduce(const float *y, const float *dz, const float *z, const float *shift, int B, int rows_per_img,
                              int C, int mode, int relu, float *partial, int Cstride, hipStream_t st, const float *fa = nullptr,
                              const float *fb = nullptr);
hipError_t launch_bn_finalize(const float *partial, int nb, int Cstride, double n, int C, const float *shift,
                              const float *gamma, const float *beta, float eps, float momentum, float *rm, float *rv,
                              long long *nbt, float *a, float *b, float *mean, float *rstd, hipStream_t st,
                              double *fold = nullptr, const unsigned *ymax = nullptr, unsigned *zmax = nullptr, int zrelu = 1);
// (ymax / zmax: a lazy activation's bound of max |z| from max |y| -- see bn_finalize_kernel)
size_t partial_fold_doubles(int nb, int C);   // scratch for `fold` (0: the partial list is short, no pre-pass)
hipError_t launch_affine_act(const float *y, const float *a, const float *b, const float *res, int B, size_t rows_per_img,
                             int C, int per_sample, int relu, float *z, hipStream_t st, unsigned *amax = nullptr,
                             const float *res_a = nullptr, const float *res_b = nullptr, int res_relu = 0,    // lazy residual
                             unsigned *zbits = nullptr);      // also the ReLU mask bit-packed, [row][C / 32] words (ConvArgs::bm_zbits)
hipError_t launch_bn_bwd_finalize(const float *partial, int nb, int Cstride, double n, int C, const float *gamma,
                                  const float *mean, const float *rstd, float *dgamma, float *dbeta, float *coef,
                                  hipStream_t st, double *fold = nullptr);
hipError_t launch_affine_bwd(const float *dz, const float *z, const float *y, const float *coef, int B, size_t rows_per_img,
                             int C, int per_sample, int relu, float *dy, float *gres, int gres_mode, hipStream_t st,
                             const float *fa = nullptr, const float *fb = nullptr, float *csum = nullptr, float *csum_out = nullptr,
                             unsigned *amax = nullptr);   // amax: max |z| / max |dy| folded into the slot (see ConvArgs::amax_in)
// csum: scratch of affine_bwd_blocks(B, rows, C) * C * 2 floats -> csum_out[C] = column sums of dy (the conv bias gradient)
int affine_bwd_blocks(int B, size_t rows_per_img, int C);
hipError_t launch_add(float *a, const float *b, size_t n, hipStream_t st);
size_t colsum_partial_floats(size_t rows, int ld);
hipError_t launch_colsum(const float *x, size_t rows, int C, int ld, float *partial, float *out, hipStream_t st);
hipError_t launch_colsum_final(const float *partial, int nb, int C, float *out, hipStream_t st);   // partial [nb][C][2] -> out[C]
hipError_t launch_maxpool2_bwd(const float *x, const float *dout, int B, int H, int W, int C, float *dx, int accumulate,
                               hipStream_t st, const float *la = nullptr, const float *lb = nullptr,    // la / lb: lazy x (ConvSrc::la)
                               float *stats_partial = nullptr);   // [maxpool2_bwd_blocks()][C][2]: also mask + BatchNorm-backward partials
int maxpool2_bwd_blocks(int B, int H, int W, int C);
hipError_t launch_deconv4_bwd_data(const float *dout, int B, int H, int W, int C, const float *wpk, float *din,
                                   hipStream_t st);
size_t deconv4_bwd_w_partial_floats(int B, int H, int C);
hipError_t launch_deconv4_bwd_w(const float *in, const float *dout, int B, int H, int W, int C, float *partial, float *dw,
                                hipStream_t st, const float *la = nullptr, const float *lb = nullptr,    // la / lb: lazy in
                                const float *wpk = nullptr, float *din = nullptr,   // both given: the same pass also writes din (= launch_deconv4_bwd_data)
                                float *stats = nullptr);   // [B * H][C][2] (lazy `in` only): din masked by its ReLU + its BatchNorm-backward partials

// ---- head / stem train kernels (kernels_head_train.hip)
hipError_t launch_pack_conv_w_dgrad(const float *w, int Cout, int CinTotal, int k, int c_off, int Cs, int CsP, int CoutPad,
                                    int cls, float *dst, hipStream_t st);
hipError_t launch_dpred_pack(const float *const dpred[10], int ld, int B, int HW, float *out, hipStream_t st);
struct AttnTrainArgs {
    const float *stats;            // [B][chunks][stat_ld][2] partial (sum, sumsq) of (x - running_mean)
    int chunks, stat_ld, B, HW;
    double *stats64;               // [B][stat_ld][2] scratch: the partials of every image summed in fp64 (fixed order)
    float *rm[9], *rv[9];          // AttnBN running statistics (updated in place, momentum 0.03)
    long long *nbt[9];
    const float *att_w[9], *att_g[9], *att_b[9];
    float *att_rm[9], *att_rv[9];  // attention BN(10) running statistics (momentum 0.1)
    long long *att_nbt[9];
    const float *weight_[9], *bias_[9];
    // saved for backward
    float *sv_inst;                // [B][576][3]  s, instance mean, instance (unbiased) variance
    float *mu_r;                   // [576][2]     batch mean, rstd
    float *bn10;                   // [9][10][2]   attention batch mean, rstd
    float *that, *yatt;            // [B][9][10]   normalised attention logits, hard-sigmoid outputs
    float *gamma_p;                // [B][576]     per-sample gamma'
    float *scale, *shift;          // [B][576]     folded per-sample affine
};
struct AttnGradPtrs { float *d_weight_[9], *d_bias_[9], *d_att_w[9], *d_att_g[9], *d_att_b[9]; };
hipError_t launch_attn_train_fwd(const AttnTrainArgs &a, hipStream_t st);
hipError_t launch_attn_train_bwd(const AttnTrainArgs &a, const float *partial, int rb_per_img, const AttnGradPtrs &gp,
                                 float *coef, hipStream_t st);
hipError_t launch_head_bwd(const float *draw, int ld, const float *z, const float *x, const float *w1, int B, int HW,
                           int blocks, float *d, float *dw_partial, float *red_partial, hipStream_t st,
                           const float *scale = nullptr, const float *shift = nullptr);
hipError_t launch_head_dx(const float *draw, int ld, const float *x, const float *w1, const float *coef, int B, int HW, int blocks,
                          float *dx, float *csum, float *csum_out, unsigned *amax, hipStream_t st, const float *scale,
                          const float *shift);
hipError_t launch_splitk_reduce(const float *partial, int ksplit, int T, int Cout, int Cin, float *dw, hipStream_t st);
int stem_wgrad_blocks(int B, int H, int W);
// img_amax / dy_amax (mode 3): max-|x| slots of the image and of dY -> the fp16-pipe kernel (stem_f16.hip)
// (fused form, mode 3 only: y / coef / y_amax non-null -- see launch_stem_wgrad_f16)
hipError_t launch_stem_wgrad(const float *img, const float *dy, int B, int H, int W, float *partial, float *dw,
                             hipStream_t st, const unsigned *img_amax = nullptr, const unsigned *dy_amax = nullptr,
                             const float *y = nullptr, const float *coef = nullptr, const unsigned *y_amax = nullptr);

// ---- conv weight gradient (wgrad_mfma.hip)
struct WgradArgs {
    ConvSrc src[4];
    int nsrc;
    int B, Hin, Win, Hout, Wout, Cin, Cout;
    const float *dy;          // NHWC (B,Hout,Wout,dy_ld); channels >= Cout must be zero / are ignored
    int dy_ld;
    float *partial;           // [ksplit][k*k][Cout][Cin]
    int ksplit, n_tiles, c_tiles, ppr, ppi, groups_per_img;
    int small;                // 1: 16-input-channel layer on the LDS-free 16x16x4 kernel (ksplit = workgroups)
    int prec;                 // 1: bf16 MFMA operands, 2: 3-way bf16 split, 3: 2-way fp16 split -- where wgrad_bf16_ok() (wgrad_bf16.hip)
    int pb;                   // 32-pixel patches per staged group (2, or 1 for the split kernels)
    int pipe;                 // != 0: the software-pipelined fp16-split kernel (wgrad_pipe.hip) with this tile; ksplit = slices
    const unsigned *amax_x[4], *amax_dy;   // prec 3: max |x| (bit patterns) of every source and of dY (see ConvArgs::amax_in)
};
void wgrad_plan(WgradArgs &a, int ks, int stride);            // fills the tiling fields
size_t wgrad_partial_floats(const WgradArgs &a, int ks);
hipError_t launch_wgrad(const WgradArgs &a, int ks, int stride, float *dw_oihw, hipStream_t st);
// lazy X sources (ConvSrc::la in conv_mfma.h): does the kernel wgrad_plan() chose for `a` form them on load?  (mode 3:
// wgrad_pipe_kernel's default tile, wgrad_bf16_kernel<SPL = 2>, wgrad_thin16_kernel; anything else fails such a launch)
bool wgrad_lazy_capable(const WgradArgs &a, int ks, int stride);
bool wgrad_bf16_ok(const WgradArgs &a, int ks, int stride);
hipError_t launch_wgrad_bf16(const WgradArgs &a, int ks, int stride, int WN, int WC, hipStream_t st);
int wgrad_bf16_patches(int prec);
int wgrad_pipe_tile(const WgradArgs &a, int ks, int stride);      // wgrad_pipe.hip: 0 = not eligible
void wgrad_pipe_plan(WgradArgs &a, int tile);
hipError_t launch_wgrad_pipe(const WgradArgs &a, int ks, hipStream_t st);
bool wgrad_thin_ok(const WgradArgs &a, int ks, int stride);        // conv_thin.hip: the 16 -> 16 layer on the fp16 pipe (mode 3)
hipError_t launch_wgrad_thin(const WgradArgs &a, hipStream_t st);

}  // namespace mc
