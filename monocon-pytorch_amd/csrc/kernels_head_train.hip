// Train-mode kernels specific to the dense heads, plus the dgrad weight packer and the 7x7 stem
// weight gradient.
//
// Head train pipeline (reference model/dense_heads/monocon_heads.py:114-131,165-200 and
// model/norm/attentive_norm.py:79-91,154-164 under autograd):
//   x   = conv3x3(feat) + bias                      fused MFMA conv, 9 heads side by side (576 ch)
//   AttnBN statistics / attention / per-sample affine            attn_train_fwd_kernel (one WG per head)
//   h   = relu(scale_bc * x + shift_bc), nine 1x1 convs + b, sigmoid/clamp | depth transform | identity,
//         NCHW predictions                          head_apply_kernel (kernels_misc.hip; also stores h)
// backwards: head_bwd_kernel (1x1 weight gradients, ReLU-masked data gradient, AttnBN reductions in one
// pass); the AttnBN backward then reduces to one per-(image, channel) affine map dx = P*d + Q*x + R
// (attn_train_bwd_kernel computes P, Q, R and all the small parameter gradients).
#include "kernels.h"
#include "train.h"

namespace mc {

// ------------------------------------------------------------------ dgrad weight pack
// forward W (Cout, CinTotal, k, k) -> panel of the transposed / flipped convolution that maps
// dY (Cout channels) to dX of ONE source (channels [c_off, c_off + Cs)):
//   dst[tap'][n/4][c_local (padded to CsP)][n%4] = W[n][c_off + c_local][k-1-r'][k-1-s']
// cls < 0: the stride-1 case above.  cls = 2*py + px: output-parity class of a stride-2 3x3 data gradient,
//   dX[2i+py][2j+px] = sum_{dr<KH, ds<KW} dY[i+dr][j+ds] . W[.][.][r(dr)][s(ds)],   KH = 1 + py, KW = 1 + px,
//   r(dr) = 1 (py = 0) or 2 - 2*dr (py = 1), same for s: the panel keeps only those KH*KW taps, window order.
__global__ void pack_conv_w_dgrad_kernel(const float *__restrict__ w, int Cout, int CinTotal, int k, int c_off, int Cs,
                                         int CsP, int CoutPad, int cls, float *__restrict__ dst) {
    const int kk = k * k;
    const size_t total = (size_t)Cout * Cs * kk;
    const int py = cls >> 1, px = cls & 1;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int tap = e % kk;
        const int cl = (e / kk) % Cs;
        const int n = e / ((size_t)kk * Cs);
        const int r = tap / k, s = tap % k;
        int tapd;
        if (cls < 0) {
            tapd = (k - 1 - r) * k + (k - 1 - s);
        } else {
            if ((py == 0) != (r == 1) || (px == 0) != (s == 1)) continue;   // tap of the other parity
            const int dr = py ? (2 - r) / 2 : 0, ds = px ? (2 - s) / 2 : 0;
            tapd = dr * (1 + px) + ds;
        }
        dst[(((size_t)tapd * (CoutPad >> 2) + (n >> 2)) * CsP + cl) * 4 + (n & 3)] =
            w[(((size_t)n * CinTotal + c_off + cl) * k + r) * k + s];
    }
}
hipError_t launch_pack_conv_w_dgrad(const float *w, int Cout, int CinTotal, int k, int c_off, int Cs, int CsP, int CoutPad,
                                    int cls, float *dst, hipStream_t st) {
    const size_t total = (size_t)Cout * Cs * k * k;
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_conv_w_dgrad_kernel, dim3((unsigned)g), dim3(256), 0, st, w, Cout, CinTotal, k, c_off, Cs, CsP,
                       CoutPad, cls, dst);
    return hipGetLastError();
}

// ------------------------------------------------------------------ backward of the nine 1x1 convs, fused
// One pass over the hidden maps does everything between the raw-output gradient and the AttnBN backward:
//   dW1[r][c]  = sum_px draw[px][r] * z[px][head(r)*64 + c]                (partials per workgroup)
//   d[px][hc]  = [z > 0] * sum_{r in head} draw[px][r] * W1[r][c]          (ReLU-masked data gradient, written)
//   (sum d, sum d*x) per (image, row block, channel)                        (the chan_reduce mode-1 partials)
// A wave owns one head (lane = channel, so every z / x / d access is one 256-byte row), the rows of
// draw are wave-uniform and come down the scalar path.  Replaces a dense 576->65 wgrad, a dense 65->576
// dgrad (8/9 of both multiplying structural zeros) and a three-tensor reduction pass.
// (reference model/dense_heads/monocon_heads.py:119-120 under autograd)
struct HeadBwdArgs {
    const float *draw; int ld;
    const float *z, *x, *w1;          // z may be null: recomputed as relu(scale_bc * x + shift_bc), bit-identical to head_apply
    const float *scale, *shift;       // [B][CP] AttnBN coefficients of the forward (used when z is null)
    float *d, *dw_partial, *red_partial;
    int HW, rows_per_block, blocks_per_img;
    // MODE 2 (head_dx_kernel): [B][CP] float4 (P, Q, R, -) of the AttnBN backward; d receives dx = P*d + Q*x + R;
    // csum [blocks][CP][2]: column sums of dx per workgroup; amax: max |dx|
    const float *coef;
    float *csum;
    unsigned *amax;
};
// The rows of draw are the same for all lanes of a wave.  Round 2 took them down the scalar path (s_load per pixel):
// measured in round 3 (rocprofv3: 2.07 ms, 2.4 TB/s, waves parked 85 % of the time) that path has no prefetch -- every
// pixel of the 24-row head waited a full memory round trip for its 96 bytes.  Now a workgroup stages the rows of HB_PX
// pixels in LDS with coalesced 16-byte loads (fetched one block ahead into registers) and every wave reads its rows as
// LDS broadcasts.
constexpr int HB_PX = 64, HB_LD = 80;
// Work per pixel is proportional to a head's row count (2 ... 24): with one wave per head in a nine-wave workgroup the two
// widest heads set the pace of all nine (round 5, rows cut to two per head: 1.17 -> 0.50 ms).  The head is a GRID dimension
// now: a workgroup = (pixel block, head), four waves that take the pixels i % 4 == wave of every staged block and fold their
// sums through LDS in wave order (fixed order: deterministic); the hardware's workgroup scheduler does the balancing.
constexpr int HB_WAVES = 4, HB_NT = HB_WAVES * 64, HB_MAXR = 24;

// MODE 0: the pass as described above; 1: the same without storing d (the plan's default since round 5: 2.3 GB less written
// here and 2.3 GB less read by the pass that follows); 2: head_dx_kernel -- d is formed again by the same fma chain
// (bit-identical) and leaves as dx = P*d + Q*x + R, s1 = column sum of dx, s2 = max |dx|
template <int RB, int NR, int MODE>
struct HeadPart {
    static constexpr int CP = NUM_HEADS * HEAD_CH;
    float w[NR], acc[NR];
    float s1, s2, zsc, zsh, cp, cq, cr;
    const float *xp, *zp;
    float *dp;
    bool rez;
    int h, sub, nsub;

    __device__ __forceinline__ void init(const HeadBwdArgs &a, int h_, int lane, size_t p0, int b, int sub_, int nsub_) {
        h = h_; sub = sub_; nsub = nsub_;
#pragma unroll
        for (int r = 0; r < NR; ++r) { w[r] = a.w1[(RB + r) * HEAD_CH + lane]; acc[r] = 0.f; }
        s1 = 0.f; s2 = 0.f;
        rez = a.z == nullptr;
        zsc = rez ? a.scale[(size_t)b * CP + h * HEAD_CH + lane] : 0.f;
        zsh = rez ? a.shift[(size_t)b * CP + h * HEAD_CH + lane] : 0.f;
        xp = a.x + p0 * CP + h * HEAD_CH + lane;
        zp = rez ? xp : a.z + p0 * CP + h * HEAD_CH + lane;
        dp = MODE == 1 ? nullptr : a.d + p0 * CP + h * HEAD_CH + lane;
        cp = cq = cr = 0.f;
        if (MODE == 2) {
            const float4 cf = reinterpret_cast<const float4 *>(a.coef)[(size_t)b * CP + h * HEAD_CH + lane];
            cp = cf.x; cq = cf.y; cr = cf.z;
        }
    }
    __device__ __forceinline__ void one(const float *gsh, int i, size_t gp, float zv, float xv) {
        const float *g = gsh + i * NR;          // the head's NR raw-gradient rows of pixel i (LDS broadcast)
        float dh = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float gv = g[r];
            if (MODE != 2) acc[r] = fmaf(gv, zv, acc[r]);
            dh = fmaf(gv, w[r], dh);
        }
        const float dv = zv > 0.f ? dh : 0.f;
        if (MODE == 2) {
            const float o = fmaf(cp, dv, fmaf(cq, xv, cr));      // (affine_bwd_kernel's expression)
            s1 += o;
            s2 = fmaxf(s2, fabsf(o));
            dp[gp * CP] = o;
        } else {
            s1 += dv;
            s2 = fmaf(dv, xv, s2);
            if (MODE == 0) dp[gp * CP] = dv;
        }
    }
    // this part's pixels of the staged block [px0, px0 + n)
    __device__ __forceinline__ void block(const float *gsh, int px0, int n) {
        constexpr int U = NR > 8 ? 4 : 8;                 // x loads in flight (a wave sees 16 pixels of a staged block)
        const int cnt = n > sub ? (n - sub + nsub - 1) / nsub : 0;
        int k = 0;
        for (; k + U <= cnt; k += U) {
            float zv[U], xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t gp = (size_t)(px0 + sub + (k + u) * nsub);
                xv[u] = xp[gp * CP];
                if (!rez) zv[u] = zp[gp * CP];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = sub + (k + u) * nsub;
                one(gsh, i, (size_t)(px0 + i), rez ? fmaxf(fmaf(xv[u], zsc, zsh), 0.f) : zv[u], xv[u]);
                if (NR > 8) __builtin_amdgcn_sched_barrier(0);      // keep the wide heads from hoisting 8 pixels x NR row values into registers
            }
        }
        for (; k < cnt; ++k) {
            const int i = sub + k * nsub;
            const size_t gp = (size_t)(px0 + i);
            const float xv1 = xp[gp * CP];
            one(gsh, i, gp, rez ? fmaxf(fmaf(xv1, zsc, zsh), 0.f) : zp[gp * CP], xv1);
        }
    }
    // parts 1 .. n-1 of a shared head park their sums in LDS ...
    __device__ __forceinline__ void spill(float *red, int lane) const {
        if (nsub == 1 || sub == 0) return;
        float *q = red + (size_t)(sub - 1) * (NR + 2) * 64 + lane;
        if (MODE != 2) {
#pragma unroll
            for (int r = 0; r < NR; ++r) q[r * 64] = acc[r];
        }
        q[NR * 64] = s1;
        q[(NR + 1) * 64] = s2;
    }
    // ... part 0 adds them in part order and writes the workgroup's partials
    __device__ __forceinline__ void finish(const HeadBwdArgs &a, const float *red, int blk, int lane) {
        if (MODE == 2 && a.amax) amax_update_wave(a.amax, s2);      // (a maximum: every part commits its own)
        if (sub != 0) return;
        for (int k = 1; k < nsub; ++k) {
            const float *q = red + (size_t)(k - 1) * (NR + 2) * 64 + lane;
            if (MODE != 2) {
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[r] += q[r * 64];
                s2 += q[(NR + 1) * 64];
            }
            s1 += q[NR * 64];
        }
        if (MODE == 2) {
            a.csum[((size_t)blk * CP + h * HEAD_CH + lane) * 2] = s1;      // colsum_final_kernel's partial format
            return;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) a.dw_partial[((size_t)blk * NUM_OUT_ROWS + RB + r) * HEAD_CH + lane] = acc[r];
        float *rp = a.red_partial + ((size_t)blk * CP + h * HEAD_CH + lane) * 2;
        rp[0] = s1;
        rp[1] = s2;
    }
};

// One head of one pixel block: the head's NR columns of the raw-gradient rows are staged HB_PX pixels at a time (fetched one
// block ahead into registers), every wave reads them as LDS broadcasts.
template <int RB, int NR, int MODE>
__device__ __forceinline__ void head_bwd_one(const HeadBwdArgs &a, float *gsh, float *red, int h, int wv, int lane, size_t p0, int np,
                                             int blk, int b) {
    HeadPart<RB, NR, MODE> P;
    P.init(a, h, lane, p0, b, wv, HB_WAVES);
    constexpr int NE = HB_PX * NR, NL = (NE + HB_NT - 1) / HB_NT;      // staged floats per block, loads per thread
    const int tid = threadIdx.x;
    const float *gsrc = a.draw + p0 * a.ld + RB;
    float pre[NL];
    auto fetch = [&](int px0) {
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int e = tid + k * HB_NT, px = e / NR, r = e % NR;
            pre[k] = (e < NE && px0 + px < np) ? gsrc[(size_t)(px0 + px) * a.ld + r] : 0.f;
        }
    };
    fetch(0);
    for (int px0 = 0; px0 < np; px0 += HB_PX) {
        __syncthreads();                              // the previous block's rows are no longer read
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int e = tid + k * HB_NT;
            if (e < NE) gsh[e] = pre[k];
        }
        __syncthreads();
        if (px0 + HB_PX < np) fetch(px0 + HB_PX);     // next block's rows travel while this one is processed
        P.block(gsh, px0, min(HB_PX, np - px0));
    }
    P.spill(red, lane);
    __syncthreads();
    P.finish(a, red, blk, lane);
}
// WIDE = false: the six heads of two or three rows (every one of them a pure stream over x: few registers, many waves in
// flight); true: the heads of 9, 18 and 24 rows.  Two launches, so that the narrow heads do not inherit the wide heads' registers.
template <int MODE, bool WIDE>
__global__ __launch_bounds__(HB_NT) void head_bwd_kernel(const HeadBwdArgs a) {
    __shared__ float gsh[HB_PX * (WIDE ? HB_MAXR : 3)];
    __shared__ float red[(HB_WAVES - 1) * ((WIDE ? HB_MAXR : 3) + 2) * 64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x;
    const int b = blk / a.blocks_per_img, rb = blk % a.blocks_per_img;
    const int r0 = rb * a.rows_per_block;
    const int np = min(a.HW, r0 + a.rows_per_block) - r0;
    const size_t p0 = (size_t)b * a.HW + r0;
    // (first row, row count) of each head in HeadRow order: 0:(0,3) 1:(3,2) 2:(5,2) 3:(7,18) 4:(25,9) 5:(34,2) 6:(36,3) 7:(39,2) 8:(41,24)
    if (WIDE) {
        switch (blockIdx.y) {
            case 0: head_bwd_one<41, 24, MODE>(a, gsh, red, 8, wv, lane, p0, np, blk, b); break;
            case 1: head_bwd_one<7, 18, MODE>(a, gsh, red, 3, wv, lane, p0, np, blk, b); break;
            default: head_bwd_one<25, 9, MODE>(a, gsh, red, 4, wv, lane, p0, np, blk, b); break;
        }
    } else {
        switch (blockIdx.y) {
            case 0: head_bwd_one<0, 3, MODE>(a, gsh, red, 0, wv, lane, p0, np, blk, b); break;
            case 1: head_bwd_one<3, 2, MODE>(a, gsh, red, 1, wv, lane, p0, np, blk, b); break;
            case 2: head_bwd_one<5, 2, MODE>(a, gsh, red, 2, wv, lane, p0, np, blk, b); break;
            case 3: head_bwd_one<34, 2, MODE>(a, gsh, red, 5, wv, lane, p0, np, blk, b); break;
            case 4: head_bwd_one<36, 3, MODE>(a, gsh, red, 6, wv, lane, p0, np, blk, b); break;
            default: head_bwd_one<39, 2, MODE>(a, gsh, red, 7, wv, lane, p0, np, blk, b); break;
        }
    }
}
template <int MODE>
static void head_bwd_launch(const HeadBwdArgs &a, int blocks, hipStream_t st) {
    hipLaunchKernelGGL((head_bwd_kernel<MODE, true>), dim3(blocks, 3), dim3(HB_NT), 0, st, a);
    hipLaunchKernelGGL((head_bwd_kernel<MODE, false>), dim3(blocks, 6), dim3(HB_NT), 0, st, a);
}
// blocks = chan_reduce_blocks(B, HW); rows_per_block = that partition's row count (kernels_train.hip)
static hipError_t head_bwd_args(HeadBwdArgs &a, const float *draw, int ld, const float *z, const float *x, const float *w1, int B,
                                int HW, int blocks, const float *scale, const float *shift) {
    const int *rbeg = head_row_begin();
    static const int RB[NUM_HEADS + 1] = {0, 3, 5, 7, 25, 34, 36, 39, 41, 65};
    for (int i = 0; i <= NUM_HEADS; ++i)
        if (rbeg[i] != RB[i]) return hipErrorInvalidValue;      // the switch above hard-codes the HeadRow table
    if (blocks % B || ld != HB_LD) return hipErrorInvalidValue;
    if (!z && (!scale || !shift)) return hipErrorInvalidValue;
    a = HeadBwdArgs{};
    a.draw = draw; a.ld = ld; a.z = z; a.x = x; a.w1 = w1;
    a.scale = scale; a.shift = shift;
    a.HW = HW; a.blocks_per_img = blocks / B; a.rows_per_block = (HW + a.blocks_per_img - 1) / a.blocks_per_img;
    return hipSuccess;
}
// d == nullptr: the masked gradient is not stored (launch_head_dx forms it again)
hipError_t launch_head_bwd(const float *draw, int ld, const float *z, const float *x, const float *w1, int B, int HW,
                           int blocks, float *d, float *dw_partial, float *red_partial, hipStream_t st, const float *scale,
                           const float *shift) {
    HeadBwdArgs a;
    hipError_t e = head_bwd_args(a, draw, ld, z, x, w1, B, HW, blocks, scale, shift);
    if (e != hipSuccess) return e;
    a.d = d; a.dw_partial = dw_partial; a.red_partial = red_partial;
    if (d) head_bwd_launch<0>(a, blocks, st);
    else head_bwd_launch<1>(a, blocks, st);
    return hipGetLastError();
}
// The AttnBN backward dx = P*d + Q*x + R (coef: [B][CP] float4, attn_train_bwd_kernel) from (draw, x) instead of a stored d:
// same partition and per-pixel arithmetic as head_bwd_kernel, so d is the value that kernel reduced.  csum: [blocks][CP][2]
// workspace; csum_out [CP]: column sums of dx (bias gradients of the fused 3x3 convs); amax: slot of max |dx| (or null).
hipError_t launch_head_dx(const float *draw, int ld, const float *x, const float *w1, const float *coef, int B, int HW, int blocks,
                          float *dx, float *csum, float *csum_out, unsigned *amax, hipStream_t st, const float *scale,
                          const float *shift) {
    if (!coef || !dx || !csum || !csum_out) return hipErrorInvalidValue;
    HeadBwdArgs a;
    hipError_t e = head_bwd_args(a, draw, ld, nullptr, x, w1, B, HW, blocks, scale, shift);
    if (e != hipSuccess) return e;
    a.d = dx; a.coef = coef; a.csum = csum; a.amax = amax;
    head_bwd_launch<2>(a, blocks, st);
    return launch_colsum_final(csum, blocks, NUM_HEADS * HEAD_CH, csum_out, st);
}


// ten NCHW gradient maps -> (B,HW,ld) rows in HeadRow order, zero-padded to ld columns
struct DpredPackArgs {
    const float *dpred[10];
    int ld, B, HW;
    float *out;
    int row_pred[NUM_OUT_ROWS], row_ch[NUM_OUT_ROWS], pred_c[10];
};
__global__ __launch_bounds__(256) void dpred_pack_kernel(const DpredPackArgs a) {
    __shared__ float t[64][NUM_OUT_ROWS + 2];
    __shared__ const float *rowp[NUM_OUT_ROWS];      // where row r of this (image, tile) starts: the kernel-argument tables
                                                     // are indexed dynamically ONCE per row here, not once per element
                                                     // (they live in scratch memory then: 267 us for 570 MB in round 5)
    const int tiles = (a.HW + 63) / 64;
    const int b = blockIdx.x / tiles, hw0 = (blockIdx.x % tiles) * 64;
    if (threadIdx.x < NUM_OUT_ROWS) {
        const int r = threadIdx.x, p = a.row_pred[r];
        rowp[r] = a.dpred[p] + ((size_t)b * a.pred_c[p] + a.row_ch[r]) * a.HW + hw0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * NUM_OUT_ROWS; e += 256) {
        const int r = e / 64, px = e % 64;
        t[px][r] = (hw0 + px < a.HW) ? rowp[r][px] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * a.ld; e += 256) {
        const int px = e / a.ld, r = e % a.ld;
        if (hw0 + px < a.HW) a.out[((size_t)b * a.HW + hw0 + px) * a.ld + r] = r < NUM_OUT_ROWS ? t[px][r] : 0.f;
    }
}
hipError_t launch_dpred_pack(const float *const dpred[10], int ld, int B, int HW, float *out, hipStream_t st) {
    DpredPackArgs a;
    a.ld = ld; a.B = B; a.HW = HW; a.out = out;
    const HeadRow *rows = head_rows();
    static const int PC[10] = {3, 9, 2, 2, 2, 18, 3, 2, 12, 12};
    for (int i = 0; i < 10; ++i) { a.dpred[i] = dpred[i]; a.pred_c[i] = PC[i]; }
    for (int r = 0; r < NUM_OUT_ROWS; ++r) { a.row_pred[r] = rows[r].pred; a.row_ch[r] = rows[r].ch; }
    hipLaunchKernelGGL(dpred_pack_kernel, dim3(B * ((HW + 63) / 64)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ AttnBN train forward (one WG per head, lane = channel)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// reduce a per-thread (s1, s2) pair across the 16 row-groups of a 1024-thread workgroup (lane = channel)
__device__ __forceinline__ void group_reduce2(double &s1, double &s2, double (*sh)[64][2], int grp, int c) {
    __syncthreads();
    sh[grp][c][0] = s1; sh[grp][c][1] = s2;
    __syncthreads();
    s1 = 0; s2 = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) { s1 += sh[g][c][0]; s2 += sh[g][c][1]; }
}

// per-(image, channel) sums of per-patch (or per-row-block) partials, fp64: sixteen interleaved parts, each with four loads
// in flight, combined in a fixed order.  (Round 5: four parts and one load in flight took 140 us for the head conv's 135 MB --
// 1 TB/s, every load a full round trip; it also serves the backward now, where nine workgroups used to walk the partials
// of all images one after the other.)
__global__ __launch_bounds__(1024) void inst_stats_kernel(const float *__restrict__ stats, int chunks, int stat_ld,
                                                          double *__restrict__ out) {
    const int b = blockIdx.x, c = threadIdx.x & 63, ch = blockIdx.y * 64 + c, part = threadIdx.x >> 6;
    const float2 *q = reinterpret_cast<const float2 *>(stats) + (size_t)b * chunks * stat_ld + ch;
    double s1 = 0.0, s2 = 0.0;
    int k = part;
    for (; k + 48 < chunks; k += 64) {
        const float2 v0 = q[(size_t)k * stat_ld], v1 = q[(size_t)(k + 16) * stat_ld], v2 = q[(size_t)(k + 32) * stat_ld],
                     v3 = q[(size_t)(k + 48) * stat_ld];
        s1 += (double)v0.x; s2 += (double)v0.y;
        s1 += (double)v1.x; s2 += (double)v1.y;
        s1 += (double)v2.x; s2 += (double)v2.y;
        s1 += (double)v3.x; s2 += (double)v3.y;
    }
    for (; k < chunks; k += 16) {
        const float2 v = q[(size_t)k * stat_ld];
        s1 += (double)v.x; s2 += (double)v.y;
    }
    __shared__ double red[2][16][64];
    red[0][part][c] = s1;
    red[1][part][c] = s2;
    __syncthreads();
    if (part < 2) {        // wave 0 folds the sums, wave 1 the sums of squares: parts in order
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[part][g][c];
        out[((size_t)b * stat_ld + ch) * 2 + part] = t;
    }
}

__global__ __launch_bounds__(1024) void attn_train_fwd_kernel(const AttnTrainArgs a) {
    __shared__ double shred[16][64][2];
    const int h = blockIdx.x, c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int CP = NUM_HEADS * HEAD_CH, ch = h * HEAD_CH + c;
    const double HW = (double)a.HW, n = HW * a.B;
    __shared__ float att[64][NUM_AFFINE];          // a[b][k] then y[b][k]   (B <= 64)
    // the B x 10 attention inputs are 64-term dot products over the channels: one per thread from these two tables.  (As
    // 320 wave reductions in a row on one wave -- six dependent cross-lane steps each -- they were most of this kernel's time.)
    __shared__ float svl[64][HEAD_CH], awl[NUM_AFFINE][HEAD_CH];
    const float shift0 = a.rm[h][c];                // statistics were accumulated around the old running mean
    double S1 = 0, S2 = 0;
    double pre1[8], pre2[8];                    // (stats64: eight images' sums requested together, not one round trip per image)
    // every small operand used further down is requested here, together (fetched where it is used, each is a full memory
    // round trip in a kernel of nine workgroups with nothing to hide it behind)
    float aw[NUM_AFFINE], wk_[NUM_AFFINE], bk_[NUM_AFFINE];
#pragma unroll
    for (int k = 0; k < NUM_AFFINE; ++k) {
        aw[k] = a.att_w[h][k * HEAD_CH + c]; wk_[k] = a.weight_[h][k * HEAD_CH + c]; bk_[k] = a.bias_[h][k * HEAD_CH + c];
    }
    const int ck = c < NUM_AFFINE ? c : 0;
    const float att_g_c = a.att_g[h][ck], att_b_c = a.att_b[h][ck], att_rm_c = a.att_rm[h][ck], att_rv_c = a.att_rv[h][ck];
    const float rv_c = a.rv[h][c];
    for (int b = 0; b < a.B; ++b) {
        double s1 = 0, s2 = 0;
        if (a.stats64) {                        // pre-reduced by inst_stats_kernel (B x 9 workgroups instead of 9)
            if ((b & 7) == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int bb = b + u < a.B ? b + u : a.B - 1;
                    const double2 v = *reinterpret_cast<const double2 *>(a.stats64 + ((size_t)bb * a.stat_ld + ch) * 2);
                    pre1[u] = v.x; pre2[u] = v.y;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if ((b & 7) == u) { s1 = pre1[u]; s2 = pre2[u]; }
        } else {
            for (int k = grp; k < a.chunks; k += 16) {
                const float *q = a.stats + (((size_t)b * a.chunks + k) * a.stat_ld + ch) * 2;
                s1 += q[0]; s2 += q[1];
            }
            group_reduce2(s1, s2, shred, grp, c);   // every row-group now holds the full per-image sums
        }
        S1 += s1; S2 += s2;
        if (grp != 0) continue;                 // wave 0 (lane = channel) does the per-image attention input
        const double m0 = s1 / HW;
        const double mean = shift0 + m0, var = (s2 - s1 * m0) / (HW - 1.0);
        const float sv = (float)(mean / sqrt(var + 1e-3));
        a.sv_inst[((size_t)b * CP + ch) * 3 + 0] = sv;
        a.sv_inst[((size_t)b * CP + ch) * 3 + 1] = (float)mean;
        a.sv_inst[((size_t)b * CP + ch) * 3 + 2] = (float)var;
        svl[b][c] = sv;
    }
    if (grp == 0) {
#pragma unroll
        for (int k = 0; k < NUM_AFFINE; ++k) awl[k][c] = aw[k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < a.B * NUM_AFFINE; e += 1024) {
        const int b = e / NUM_AFFINE, k = e % NUM_AFFINE;
        float v = 0.f;
#pragma unroll 8
        for (int cc = 0; cc < HEAD_CH; ++cc) v = fmaf(svl[b][cc], awl[k][cc], v);
        att[b][k] = v;
    }
    const double m0 = S1 / n;
    const double mu = shift0 + m0;
    double var = S2 / n - m0 * m0;
    if (var < 0) var = 0;
    const float r = (float)(1.0 / sqrt(var + 1e-3));
    if (grp == 0) {
        a.mu_r[ch * 2 + 0] = (float)mu;
        a.mu_r[ch * 2 + 1] = r;
        a.rm[h][c] = (1.f - 0.03f) * shift0 + 0.03f * (float)mu;          // (shift0 = the old running mean)
        a.rv[h][c] = (1.f - 0.03f) * rv_c + 0.03f * (float)(var * n / (n - 1.0));
        if (c == 0) *a.nbt[h] += 1;
    }
    __syncthreads();
    // BN(10) over the batch dimension (train mode): lane k < 10 owns attention channel k
    if (grp == 0 && c < NUM_AFFINE) {
        double s = 0, q = 0;
        for (int b = 0; b < a.B; ++b) { s += att[b][c]; }
        const double am = s / a.B;
        for (int b = 0; b < a.B; ++b) { const double d = att[b][c] - am; q += d * d; }
        const double av = q / a.B;
        const float rk = (float)(1.0 / sqrt(av + 1e-5));
        a.bn10[(h * NUM_AFFINE + c) * 2 + 0] = (float)am;
        a.bn10[(h * NUM_AFFINE + c) * 2 + 1] = rk;
        a.att_rm[h][c] = 0.9f * att_rm_c + 0.1f * (float)am;
        a.att_rv[h][c] = 0.9f * att_rv_c + 0.1f * (float)(av * a.B / (a.B - 1.0));
        if (c == 0) *a.att_nbt[h] += 1;
        for (int b = 0; b < a.B; ++b) {
            const float that = (att[b][c] - (float)am) * rk;
            a.that[((size_t)b * NUM_HEADS + h) * NUM_AFFINE + c] = that;
            const float t = that * att_g_c + att_b_c;
            att[b][c] = fminf(fmaxf(t + 3.f, 0.f), 6.f) / 6.f;
        }
    }
    __syncthreads();
    for (int b = grp; b < a.B; b += 16) {
        float gam = 0.f, bet = 0.f;
#pragma unroll
        for (int k = 0; k < NUM_AFFINE; ++k) {
            gam = fmaf(att[b][k], wk_[k], gam);
            bet = fmaf(att[b][k], bk_[k], bet);
        }
        const size_t o = (size_t)b * CP + ch;
        a.gamma_p[o] = gam;
        const float sc = gam * r;
        a.scale[o] = sc;
        a.shift[o] = bet - (float)mu * sc;
        if (c < NUM_AFFINE) a.yatt[((size_t)b * NUM_HEADS + h) * NUM_AFFINE + c] = att[b][c];
    }
}
hipError_t launch_attn_train_fwd(const AttnTrainArgs &a, hipStream_t st) {
    if (a.B > 64 || a.B < 2) return hipErrorInvalidValue;
    if (a.stats64) {
        if (a.stat_ld % 64) return hipErrorInvalidValue;
        hipLaunchKernelGGL(inst_stats_kernel, dim3(a.B, a.stat_ld / 64), dim3(1024), 0, st, a.stats, a.chunks, a.stat_ld, a.stats64);
    }
    hipLaunchKernelGGL(attn_train_fwd_kernel, dim3(NUM_HEADS), dim3(1024), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ AttnBN train backward (one WG per head)
// partial: [B*rb][576][2] = per row-block (sum dout, sum dout*x), dout = dh*[h>0]
__global__ __launch_bounds__(1024) void attn_train_bwd_kernel(const AttnTrainArgs a, const float *__restrict__ partial,
                                                              int rb_per_img, AttnGradPtrs gp, float *__restrict__ coef,
                                                              const double *__restrict__ d64 /*[B][CP][2] pre-reduced, or null*/) {
    __shared__ double shred[16][64][2];
    const int h = blockIdx.x, c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int CP = NUM_HEADS * HEAD_CH, ch = h * HEAD_CH + c;
    const double HW = (double)a.HW, n = HW * a.B;
    const float mu = a.mu_r[ch * 2], r = a.mu_r[ch * 2 + 1];
    __shared__ float dgam[64][HEAD_CH], dbet[64][HEAD_CH];      // [b][c]
    __shared__ float dyk[64][NUM_AFFINE], dak[64][NUM_AFFINE];
    __shared__ float yat[64][NUM_AFFINE], tht[64][NUM_AFFINE];       // this head's y[b][k] and t_hat[b][k] of the forward
    __shared__ float wl[NUM_AFFINE][HEAD_CH], bl[NUM_AFFINE][HEAD_CH];   // weight_ / bias_ of the head (for the dot products below)
    for (int e = threadIdx.x; e < a.B * NUM_AFFINE; e += 1024) {
        const int b = e / NUM_AFFINE, k = e % NUM_AFFINE;
        yat[b][k] = a.yatt[((size_t)b * NUM_HEADS + h) * NUM_AFFINE + k];
        tht[b][k] = a.that[((size_t)b * NUM_HEADS + h) * NUM_AFFINE + k];
    }
    // every small operand of the single-wave sections below is requested here, together: fetched where it is used, each was
    // a full memory round trip on the critical path of the backward (nine workgroups, nothing else to hide it behind)
    float wk_[NUM_AFFINE], bk_[NUM_AFFINE], aw[NUM_AFFINE];
#pragma unroll
    for (int k = 0; k < NUM_AFFINE; ++k) {
        wk_[k] = a.weight_[h][k * HEAD_CH + c]; bk_[k] = a.bias_[h][k * HEAD_CH + c]; aw[k] = a.att_w[h][k * HEAD_CH + c];
    }
    const int ck = c < NUM_AFFINE ? c : 0;
    const float att_g_c = a.att_g[h][ck], att_b_c = a.att_b[h][ck], bn10_rk = a.bn10[(h * NUM_AFFINE + ck) * 2 + 1];
    double M1 = 0, M2 = 0;
    double2 pd[8];          // (eight images' operands requested together, not one round trip per image)
    float pg[8];
    for (int b = 0; b < a.B; ++b) {
        double d1 = 0, d2 = 0;
        if ((b & 7) == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bb = b + u < a.B ? b + u : a.B - 1;
                if (d64) pd[u] = *reinterpret_cast<const double2 *>(d64 + ((size_t)bb * CP + ch) * 2);
                pg[u] = a.gamma_p[(size_t)bb * CP + ch];
            }
        }
        float gp_ = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if ((b & 7) == u) { gp_ = pg[u]; if (d64) { d1 = pd[u].x; d2 = pd[u].y; } }
        if (d64) {
        } else {
            for (int k = grp; k < rb_per_img; k += 16) {
                const float *q = partial + (((size_t)b * rb_per_img + k) * CP + ch) * 2;
                d1 += q[0]; d2 += q[1];
            }
            group_reduce2(d1, d2, shred, grp, c);
        }
        const float db = (float)d1, dg = (float)(r * (d2 - mu * d1));
        dbet[b][c] = db; dgam[b][c] = dg;
        M1 += (double)gp_ * db; M2 += (double)gp_ * dg;
    }
    M1 /= n; M2 /= n;
    if (grp == 0) {
#pragma unroll
        for (int k = 0; k < NUM_AFFINE; ++k) { wl[k][c] = wk_[k]; bl[k][c] = bk_[k]; }
    }
    __syncthreads();
    // dy[b][k] = sum_c dgamma[b][c] * weight_[k][c] + dbeta[b][c] * bias_[k][c]: one 64-term dot product per thread (as 320 wave
    // reductions in a row on one wave they were most of this kernel's time)
    for (int e = threadIdx.x; e < a.B * NUM_AFFINE; e += 1024) {
        const int b = e / NUM_AFFINE, k = e % NUM_AFFINE;
        float v = 0.f;
#pragma unroll 8
        for (int cc = 0; cc < HEAD_CH; ++cc) v = fmaf(dgam[b][cc], wl[k][cc], fmaf(dbet[b][cc], bl[k][cc], v));
        dyk[b][k] = v;
    }
    __syncthreads();
    if (grp != 0) return;      // the rest is small: one wave (lane = channel); no block-wide barrier below is skipped
    // gradients of weight_ / bias_ and dy[b][k]
#pragma unroll
    for (int k = 0; k < NUM_AFFINE; ++k) {
        float gw = 0.f, gb = 0.f;
        for (int b = 0; b < a.B; ++b) {
            const float yv = yat[b][k];
            gw = fmaf(yv, dgam[b][c], gw);
            gb = fmaf(yv, dbet[b][c], gb);
        }
        gp.d_weight_[h][k * HEAD_CH + c] = gw;
        gp.d_bias_[h][k * HEAD_CH + c] = gb;
    }
    __syncthreads();
    // hard-sigmoid + BN(10) backward over the batch: lane k owns attention channel k
    if (c < NUM_AFFINE) {
        const float g = att_g_c, be = att_b_c, rk = bn10_rk;
        double s_dt = 0, s_dtt = 0;
        for (int b = 0; b < a.B; ++b) {
            const float that = tht[b][c];
            const float t = that * g + be;
            const float dt = (t + 3.f > 0.f && t + 3.f < 6.f) ? dyk[b][c] / 6.f : 0.f;
            dyk[b][c] = dt;
            s_dt += dt; s_dtt += (double)dt * that;
        }
        gp.d_att_g[h][c] = (float)s_dtt;
        gp.d_att_b[h][c] = (float)s_dt;
        for (int b = 0; b < a.B; ++b) {
            const float that = tht[b][c];
            dak[b][c] = g * rk * (dyk[b][c] - (float)(s_dt / a.B) - that * (float)(s_dtt / a.B));
        }
    }
    __syncthreads();
    float gwa[NUM_AFFINE];
#pragma unroll
    for (int k = 0; k < NUM_AFFINE; ++k) gwa[k] = 0.f;
    float psv[8][3];
    for (int b = 0; b < a.B; ++b) {
        const size_t o = (size_t)b * CP + ch;
        if ((b & 7) == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t oo = (size_t)(b + u < a.B ? b + u : a.B - 1) * CP + ch;
                psv[u][0] = a.sv_inst[oo * 3]; psv[u][1] = a.sv_inst[oo * 3 + 1]; psv[u][2] = a.sv_inst[oo * 3 + 2];
                pg[u] = a.gamma_p[oo];
            }
        }
        float sv = 0.f, m = 0.f, v = 0.f, gp_ = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if ((b & 7) == u) { sv = psv[u][0]; m = psv[u][1]; v = psv[u][2]; gp_ = pg[u]; }
        float ds = 0.f;
#pragma unroll
        for (int k = 0; k < NUM_AFFINE; ++k) {
            ds = fmaf(dak[b][k], aw[k], ds);
            gwa[k] = fmaf(dak[b][k], sv, gwa[k]);
        }
        const double ve = (double)v + 1e-3;
        const double dm = ds / sqrt(ve), dv = (double)ds * m * (-0.5) / (ve * sqrt(ve));
        float *cf = coef + o * 4;
        cf[0] = r * gp_;
        cf[1] = (float)(-(double)r * r * M2 + 2.0 * dv / (HW - 1.0));
        cf[2] = (float)(-(double)r * M1 + (double)r * r * mu * M2 + dm / HW - 2.0 * dv * m / (HW - 1.0));
        cf[3] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NUM_AFFINE; ++k) gp.d_att_w[h][k * HEAD_CH + c] = gwa[k];
}
hipError_t launch_attn_train_bwd(const AttnTrainArgs &a, const float *partial, int rb_per_img, const AttnGradPtrs &gp,
                                 float *coef, hipStream_t st) {
    if (a.B > 64) return hipErrorInvalidValue;
    // the forward's fp64 scratch ([B][stat_ld][2], stat_ld >= 576) is free again: the row-block partials of every image are
    // summed there by B x 9 workgroups first
    constexpr int CP = NUM_HEADS * HEAD_CH;
    const double *d64 = nullptr;
    if (a.stats64 && a.stat_ld >= CP) {
        hipLaunchKernelGGL(inst_stats_kernel, dim3(a.B, CP / 64), dim3(1024), 0, st, partial, rb_per_img, CP, a.stats64);
        d64 = a.stats64;
    }
    hipLaunchKernelGGL(attn_train_bwd_kernel, dim3(NUM_HEADS), dim3(1024), 0, st, a, partial, rb_per_img, gp, coef, d64);
    return hipGetLastError();
}

// ------------------------------------------------------------------ 7x7 stem weight gradient
// dW[o][c][r][s] = sum_{b,y,x} dY[b,y,x,o] * img[b,c,y-3+r,x-3+s].  VALU kernel: 16x64 pixel tile per WG,
// thread t < 147 owns tap (c,r,s) and accumulates the 16 output channels; partial [tiles][147][16].
constexpr int STEM_WG_TILES = 8;   // image tiles accumulated per workgroup (8x fewer split-K partials)
__global__ __launch_bounds__(192) void stem_wgrad_kernel(const float *__restrict__ img, const float *__restrict__ dy, int B,
                                                         int H, int W, float *__restrict__ partial) {
    constexpr int TH = 8, TW = 32;
    __shared__ float it[3][TH + 6][TW + 6];
    __shared__ __attribute__((aligned(16))) float dt[TH * TW][16];
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = B * tiles_x * tiles_y;
    const int t = threadIdx.x;
    const int c = (t < 147 ? t : 0) / 49, r = ((t < 147 ? t : 0) % 49) / 7, s = (t < 147 ? t : 0) % 7;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.f;
    for (int tile = blockIdx.x * STEM_WG_TILES; tile < ntiles && tile < (blockIdx.x + 1) * STEM_WG_TILES; ++tile) {
        const int b = tile / (tiles_x * tiles_y);
        const int ty0 = ((tile / tiles_x) % tiles_y) * TH, tx0 = (tile % tiles_x) * TW;
        __syncthreads();
        for (int e = threadIdx.x; e < 3 * (TH + 6) * (TW + 6); e += 192) {
            const int lx = e % (TW + 6), ly = (e / (TW + 6)) % (TH + 6), cc = e / ((TW + 6) * (TH + 6));
            const int y = ty0 - 3 + ly, x = tx0 - 3 + lx;
            it[cc][ly][lx] = (y >= 0 && y < H && x >= 0 && x < W) ? img[(((size_t)b * 3 + cc) * H + y) * W + x] : 0.f;
        }
        for (int e = threadIdx.x; e < TH * TW * 4; e += 192) {
            const int q = e % 4, px = e / 4;
            const int y = ty0 + px / TW, x = tx0 + px % TW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y < H && x < W) v = *reinterpret_cast<const f32x4 *>(dy + (((size_t)b * H + y) * W + x) * 16 + q * 4);
            *reinterpret_cast<f32x4 *>(&dt[px][q * 4]) = v;
        }
        __syncthreads();
        if (t < 147) {
            for (int py = 0; py < TH; ++py)
                for (int px = 0; px < TW; ++px) {
                    const float v = it[c][py + r][px + s];
                    const f32x4 *d = reinterpret_cast<const f32x4 *>(&dt[py * TW + px][0]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 dv = d[q];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[q * 4 + j] = fmaf(v, dv[j], acc[q * 4 + j]);
                    }
                }
        }
    }
    if (t < 147) {
#pragma unroll
        for (int o = 0; o < 16; ++o) partial[((size_t)blockIdx.x * 147 + t) * 16 + o] = acc[o];
    }
}
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float *__restrict__ partial, int nblocks,
                                                                float *__restrict__ dw /*(16,3,7,7)*/) {
    const int t = blockIdx.x / 16, o = blockIdx.x % 16;
    double s = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[((size_t)i * 147 + t) * 16 + o];
    __shared__ double sh[4];
    for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dw[o * 147 + t] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}
// ---- the same gradient on v_mfma_f32_16x16x4_f32 (see wgrad_small_kernel in wgrad_mfma.hip for the mapping):
// A[n][k] = dY[pixel k][n] (one coalesced dword per lane), B[k][j] = img[c(t)][y + r(t) - 3][x + k + s(t) - 3] for the
// 16 taps t = 16*jt + j of tap tile jt (ten tiles cover the 147 taps).  Fetching B straight from global memory (ten
// dword loads per group of four pixels, the 64 lanes spread over the (channel, tap-row) rows of the window) is bound by
// the texture-address path (measured: MFMA busy 0.23, 2.2 ms); so a workgroup stages the 3 x (4+6) x (128+6) image
// window of a 4-row x 128-pixel output tile once (coalesced), and the B operands are ds_read_b32 at immediate offsets
// (row pitch 136: consecutive tap rows land 8 banks apart, the two K lanes of a tap share an address): 0.88 ms.
typedef float f32x4w __attribute__((ext_vector_type(4)));
constexpr int SWL_XT = 128, SWL_P = 136, SWL_ROWS = 10;
__global__ __launch_bounds__(256) void stem_wgrad_lds_kernel(const float *__restrict__ img, const float *__restrict__ dy,
                                                             int B, int H, int W, float *__restrict__ partial) {
    constexpr int NJT = 10;
    __shared__ float tile[3 * SWL_ROWS * SWL_P];
    __shared__ float red[NJT * 4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = lane >> 4, j = lane & 15;
    int boff[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        const int t = jt * 16 + j;
        const int c = t / 49, r = (t % 49) / 7, s = t % 7;
        boff[jt] = t < 147 ? (c * SWL_ROWS + wave + r) * SWL_P + s + k : 0;     // dead taps: any finite address
    }
    f32x4w acc[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) acc[jt] = f32x4w{0.f, 0.f, 0.f, 0.f};
    const int va = (k * 16 + j) * 4;
    const int tiles_x = (W + SWL_XT - 1) / SWL_XT, tiles_y = (H + 3) / 4;
    const int ntiles = B * tiles_y * tiles_x;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int b = tl / (tiles_y * tiles_x), rem = tl - b * tiles_y * tiles_x;
        const int y0 = (rem / tiles_x) * 4, x0 = (rem % tiles_x) * SWL_XT;
        __syncthreads();                         // the previous tile's reads are done
        for (int e = tid; e < 3 * SWL_ROWS * (SWL_XT + 6); e += 256) {
            const int row = e / (SWL_XT + 6), col = e - row * (SWL_XT + 6);
            const int c = row / SWL_ROWS, rr = row - c * SWL_ROWS;
            const int gy = y0 - 3 + rr, gx = x0 - 3 + col;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = img[(((size_t)b * 3 + c) * H + gy) * W + gx];
            tile[row * SWL_P + col] = v;
        }
        __syncthreads();
        const int y = y0 + wave;
        const __amdgpu_buffer_rsrc_t r_dy =
            make_rsrc(dy + ((size_t)b * H + (y < H ? y : 0)) * W * 16, y < H ? (unsigned)(W * 16) * 4u : 0u);
#pragma unroll 4
        for (int g = 0; g < SWL_XT / 4; ++g) {
            const float av = buf_load1(r_dy, va, (x0 + g * 4) * 64);      // beyond the row: zero
            float bv[NJT];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) bv[jt] = tile[boff[jt] + g * 4];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[jt], acc[jt], 0, 0, 0);
        }
    }
    // workgroup reduction, wave after wave (fixed order); D: row (out channel n) = 4*(lane>>4) + q, column = tap j
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float *dst = &red[jt * 4 + q][lane];
                    *dst = w == 0 ? acc[jt][q] : *dst + acc[jt][q];
                }
        }
        __syncthreads();
    }
    for (int e = tid; e < NJT * 4 * 64; e += 256) {
        const int l = e & 63, idx = e >> 6;
        const int q = idx & 3, jt = idx >> 2;
        const int t = jt * 16 + (l & 15), n = 4 * (l >> 4) + q;
        if (t < 147) partial[((size_t)blockIdx.x * 147 + t) * 16 + n] = red[idx][l];
    }
}
static bool stem_wgrad_use_mfma(int W) { return W % 4 == 0 && W >= 16; }
int stem_wgrad_blocks(int B, int H, int W) {
    if (stem_wgrad_use_mfma(W)) {
        const int nb = B * ((H + 3) / 4) * ((W + SWL_XT - 1) / SWL_XT);
        return nb > 1024 ? 1024 : nb;
    }
    return (B * ((W + 31) / 32) * ((H + 7) / 8) + STEM_WG_TILES - 1) / STEM_WG_TILES;
}
hipError_t launch_stem_wgrad(const float *img, const float *dy, int B, int H, int W, float *partial, float *dw,
                             hipStream_t st, const unsigned *img_amax, const unsigned *dy_amax, const float *y, const float *coef,
                             const unsigned *y_amax) {
    const int nb = stem_wgrad_blocks(B, H, W);
    if (y && !(img_amax && dy_amax && stem_wgrad_use_mfma(W))) return hipErrorInvalidValue;   // the fused form: fp16-pipe kernel only
#ifdef MC_NO_STEM_WG_F16
    if (y) return hipErrorInvalidValue;
#endif
#ifndef MC_NO_STEM_WG_F16
    if (img_amax && dy_amax && stem_wgrad_use_mfma(W)) {
        hipError_t e = launch_stem_wgrad_f16(img, dy, B, H, W, partial, nb, img_amax, dy_amax, st, y, coef, y_amax);
        if (e != hipSuccess) return e;
    } else
#endif
    if (stem_wgrad_use_mfma(W))
        hipLaunchKernelGGL(stem_wgrad_lds_kernel, dim3(nb), dim3(256), 0, st, img, dy, B, H, W, partial);
    else
        hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nb), dim3(192), 0, st, img, dy, B, H, W, partial);
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(147 * 16), dim3(256), 0, st, partial, nb, dw);
    return hipGetLastError();
}

}  // namespace mc
