// Launchers of the non-GEMM kernels (stem, pooling, up-sampling, packing, heads, decode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

// ---- parameter preparation ---------------------------------------------------------------
// OIHW (Cout, Cin, k, k) -> [k*k][CinTotal/4][CoutP][4] at column offset n_off / row offset c_off
hipError_t launch_pack_conv_w(const float *w_oihw, int Cout, int Cin, int ks, float *dst, int CinTotal,
                              int CoutP, int n_off, int c_off, hipStream_t st);
hipError_t launch_zero(float *p, size_t n, hipStream_t st);
hipError_t launch_noise_fill(float *p, size_t n, unsigned seed, hipStream_t st);   // autotuning aid (kernels_misc.hip)
// eval-mode BN fold: scale = g*rsqrt(rv+eps), shift = b - rm*scale (g/b may be null => 1/0)
hipError_t launch_fold_bn(const float *g, const float *b, const float *rm, const float *rv, float eps,
                          int C, float *scale, float *shift, hipStream_t st);
// stem weights OIHW (16,3,7,7) -> [c][r][s][16]
hipError_t launch_pack_stem_w(const float *w, float *dst, hipStream_t st);
// deconv weights (C,1,4,4) -> [ky][kx][C]
hipError_t launch_pack_deconv_w(const float *w, int C, float *dst, hipStream_t st);
hipError_t launch_copy(const float *src, float *dst, size_t n, hipStream_t st);
// up to COPY_BATCH_MAX small device-to-device copies in ONE launch (the per-step shuffling of head weights / gradients
// between the parameter tensors and the fused-head layouts was ~70 separate 4-us copies)
constexpr int COPY_BATCH_MAX = 32;
struct CopyBatch {
    const float *src[COPY_BATCH_MAX];
    float *dst[COPY_BATCH_MAX];
    int n[COPY_BATCH_MAX];
    int count = 0;
    bool add(const float *s, float *d, size_t nn) {
        if (count >= COPY_BATCH_MAX || nn > 0x7fffffff) return false;
        src[count] = s; dst[count] = d; n[count] = (int)nn; ++count;
        return true;
    }
};
hipError_t launch_copy_batch(const CopyBatch &cb, hipStream_t st);

// ---- backbone / neck element kernels -----------------------------------------------------
// prec 3: the fp16-pipe kernel (stem_f16.hip), every other mode the VALU kernel; amax: max |out| folded into the slot
hipError_t launch_stem(const float *img_nchw, int B, int H, int W, const float *wpk, const float *scale,
                       const float *shift, float *out_nhwc, hipStream_t st, int relu = 1, int prec = 0,
                       unsigned *amax = nullptr);
// the fp16-pipe kernel; stats (optional): [B][H][16][2] partial sums of (out - stat_shift), (out - stat_shift)^2 per output row
hipError_t launch_stem_f16(const float *img, int B, int H, int W, const float *wpk, const float *scale, const float *shift,
                           float *out, hipStream_t st, int relu, unsigned *amax_out, float *stats = nullptr,
                           const float *stat_shift = nullptr, unsigned *img_amax = nullptr);
// the stem's weight gradient on the fp16 pipe (partials in the layout of stem_wgrad_lds_kernel; the caller reduces them)
// (y != null: `dy` is the MASKED gradient d of the stem's activation map and dY = P d + Q y + R is formed on the fly from the
//  BatchNorm-backward coefficients coef[16][4]; dy_amax = max |d|, y_amax = max |y|)
hipError_t launch_stem_wgrad_f16(const float *img, const float *dy, int B, int H, int W, float *partial, int nblocks,
                                 const unsigned *img_amax, const unsigned *dy_amax, hipStream_t st, const float *y = nullptr,
                                 const float *coef = nullptr, const unsigned *y_amax = nullptr);
bool stem_f16_enabled();       // false when compiled out (-DMC_NO_STEM_F16)
// la / lb (both kernels): the input is a lazy tensor -- raw conv output + BatchNorm coefficients, ConvSrc::la in conv_mfma.h
hipError_t launch_maxpool2(const float *in, int B, int H, int W, int C, float *out, hipStream_t st, const float *la = nullptr,
                           const float *lb = nullptr);
hipError_t launch_deconv4(const float *in, int B, int H, int W, int C, const float *wpk, float *out,
                          hipStream_t st, unsigned *amax = nullptr,   // amax: max |out| folded into the slot (conv_mfma.h)
                          const float *la = nullptr, const float *lb = nullptr);
hipError_t launch_nchw_to_nhwc(const float *in, int B, int C, int H, int W, float *out, hipStream_t st);
hipError_t launch_nhwc_to_nchw(const float *in, int B, int C, int H, int W, float *out, hipStream_t st);

// ---- dense heads -------------------------------------------------------------------------
constexpr int NUM_HEADS = 9;        // 8 regression/heat-map branches + dir_feat
constexpr int HEAD_CH = 64;
constexpr int NUM_AFFINE = 10;
constexpr int NUM_OUT_ROWS = 65;    // 3+2+2+18+9+2+3+2+12+12

// per-head parameter pointers for the AttnBN attention path (eval mode)
struct HeadAttnParams {
    const float *att_w[NUM_HEADS];      // (10,64) attention 1x1 conv
    const float *att_scale[NUM_HEADS];  // folded BN(10) scale / shift
    const float *att_shift[NUM_HEADS];
    const float *weight_[NUM_HEADS];    // (10,64)
    const float *bias_[NUM_HEADS];      // (10,64)
    const float *rm[NUM_HEADS];         // AttnBN running mean / var (64)
    const float *rv[NUM_HEADS];
};
// stats: [B][chunks][576][2] partial (sum, sumsq) of (x - rm);  out: scale/shift [B][9][64]
hipError_t launch_head_attn(const float *stats, int B, int chunks, int HW, const HeadAttnParams &p,
                            float *scale, float *shift, hipStream_t st);

// one row of the 65 output channels
struct HeadRow {
    int head;      // 0..8 producing hidden block
    int pred;      // destination prediction tensor 0..9
    int ch;        // channel inside that tensor
    int epi;       // 0 identity, 1 sigmoid+clamp[1e-4,1-1e-4], 2 depth 1/(sigmoid+1e-12)-1
};
struct HeadApplyArgs {
    const float *hidden;     // [B][HW][576]
    const float *scale;      // [B][9][64]
    const float *shift;
    const float *__restrict__ w;   // [64][65] transposed 1x1 weights (columns in HeadRow order)
    const float *__restrict__ b;   // [65]
    float *pred[10];         // NCHW outputs
    int pred_c[10];
    int B, HW;
    float *z_out;            // optional [B][HW][576]: the normalised + ReLU'd hidden maps (saved for the train-mode backward)
};
hipError_t launch_head_apply(const HeadApplyArgs &a, hipStream_t st);
const HeadRow *head_rows();          // host table, NUM_OUT_ROWS entries, grouped by head
const int *head_row_begin();         // [NUM_HEADS+1]

// ---- decode ------------------------------------------------------------------------------
struct DecodeArgs {
    const float *pred[10];
    const float *P2, *P2inv;
    int B, C, H, W, K;
    int lm_kernel;           // window of the local-maximum filter (odd; utils/tensor_ops.py:17 `kernel`)
    float thr, pad_h, pad_w;
    float *scores;
    int64_t *flat_index, *cls;
    float *box2d, *box3d;
    uint8_t *keep_localmax, *keep_thr;
    float *filt;             // workspace [B][C*H*W]
    unsigned *cand_key;      // workspace [B][C*H*W]: order-preserving keys of the positive local maxima, in chunks
    int *cand_idx;           // workspace [B][C*H*W]: their flat indices
    unsigned *cand_count;    // workspace [B][chunks]: live entries at the head of each chunk (written by every call)
};
hipError_t launch_decode(const DecodeArgs &a, hipStream_t st);
int decode_chunks(int n_per_image);    // candidate-list chunks per image (sizes DecodeArgs::cand_count)

hipError_t launch_preprocess(const void *img_hwc, int is_u8, int H, int W, const double mean[3], const double std[3], int Hp,
                             int Wp, float *out_chw, hipStream_t st);
hipError_t launch_preprocess_aug(const unsigned char *frames, const float *prm, int B, int Hs, int Ws, const double mean[3],
                                 const double std[3], int Hp, int Wp, float *out, hipStream_t st);

}  // namespace mc
