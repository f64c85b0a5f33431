// Internal definitions shared by the C-ABI translation units (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/monocon_hip.h"
#include "conv_mfma.h"
#include "kernels.h"
#include "train.h"

using namespace mc;

struct TrainState;   // mc_train_plan.hip
struct CommState;    // mc_comm.hip

// data parallelism: the bound "<key>#grad" tensors in groups, in the order the backward pass completes them
// (0: heads + neck, 1: backbone.level5, 2: backbone.level4, 3: the rest of the backbone)
constexpr int MC_NUM_GRAD_BUCKETS = 4;
struct GradBucket {
    float *p = nullptr;                                  // dense address range [p, p + n) ...
    size_t n = 0;
    std::vector<std::pair<float *, size_t>> parts;       // ... or, when the tensors are scattered, one entry per tensor
    int tensors = 0;
};
int mc_grad_bucket_of(const std::string &param_name);
bool mc_comm_overlap_active(mc_handle *h);               // a communicator exists and mc_backward overlaps the exchange
int mc_comm_prepare(mc_handle *h);
int mc_comm_fire_bucket(mc_handle *h, int bucket, hipStream_t main, hipStream_t side);
int mc_comm_join(mc_handle *h, hipStream_t main);

struct Bound {
    void *ptr;
    int64_t numel;
    int dtype;
};

struct Tensor {   // NHWC activation
    float *p = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    unsigned *amax = nullptr;   // slot receiving max |x| of the tensor (precision mode 3: operand scale of its consumers)
    size_t numel() const { return (size_t)B * H * W * C; }
};

struct ConvLayer {
    std::string conv, bn;   // state_dict prefixes ("" bn => bias-only / raw)
    int ks = 1, stride = 1, cin = 0, cout = 0, coutp = 0, cfg = 0;
    float bn_eps = 1e-5f;
    float *wpk = nullptr, *scale = nullptr, *shift = nullptr;
    void *wpk16 = nullptr;   // bf16 / fp16 piece panels (precision modes 1..3)
    unsigned *w_amax = nullptr;   // max |w| of the master weight(s) (mode 3: the panel's power-of-two scale)
};

struct DeconvLayer {
    std::string name;
    int C = 0;
    float *wpk = nullptr;
};

enum OpKind { OP_STEM, OP_CONV, OP_POOL, OP_DECONV, OP_HEAD_ATTN, OP_HEAD_APPLY, OP_TO_NCHW };

struct Op {
    OpKind kind;
    // conv
    ConvArgs ca{};
    int ks = 0, stride = 0;
    // generic
    const float *in = nullptr;
    float *out = nullptr;
    const float *w = nullptr, *scale = nullptr, *shift = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    int chunks = 0;
    HeadApplyArgs ha{};
    double flops = 0, bytes = 0;
    unsigned *amax = nullptr;   // OP_DECONV in mode 3: slot of the output tensor
};

struct Plan {
    int B = 0, H = 0, W = 0;
    std::vector<Op> ops;
    std::vector<void *> bufs;
    size_t bytes = 0;
    Tensor feat, lv[6];
    unsigned *amax_arena = nullptr;   // mode 3: one slot per activation tensor, zeroed at the start of every forward
    int amax_used = 0;
    int stem_op = -1, head_apply_op = -1;
    int n_backbone_ops = 0, n_neck_ops = 0;
    double flops = 0, hbm_bytes = 0;
};

struct mc_handle {
    int device = 0;
    std::string err;
    std::unordered_map<std::string, Bound> bound;
    std::map<std::string, ConvLayer> convs;
    std::map<std::string, DeconvLayer> deconvs;
    // stem
    float *stem_w = nullptr, *stem_scale = nullptr, *stem_shift = nullptr;
    // fused head 3x3 (64 -> 9*64) and second pass
    ConvLayer head3;
    float *head_bias = nullptr, *head_rm = nullptr;
    float *att_scale = nullptr, *att_shift = nullptr;   // [9][10]
    float *head_w1 = nullptr, *head_w1t = nullptr, *head_b1 = nullptr;   // [65][64], transposed [64][65], [65]
    HeadAttnParams hap{};
    bool layers_built = false, packed = false;
    int packed_groups = 0;   // bit0 backbone, bit1 neck, bit2 head
    size_t param_bytes = 0;
    std::vector<void *> param_bufs;
    std::map<std::tuple<int, int, int>, std::unique_ptr<Plan>> plans;
    Plan *last_plan = nullptr;
    float *decode_filt = nullptr;
    size_t decode_filt_n = 0;
    size_t decode_count_n = 0;
    int force_cfg = 0;   // tuning aid (mc_bench_conv)
    int lm_kernel = 3;   // mc_set_local_maximum_kernel: window of the decode's local-maximum filter (test_config['local_maximum_kernel'])
    int prec = 0;        // mc_set_precision: 0 fp32 MFMA, 1 bf16 operands, 2 three-way bf16 split, 3 two-way fp16 split
    unsigned *w_amax_arena = nullptr;                    // mode 3: max |w| per conv layer (+ the fused head panel)
    int w_amax_n = 0;
    std::map<const float *, unsigned *> w_amax_of;       // master weight -> its slot (train plan: data-gradient panels)
    int autotune = 1;    // time the workgroup shapes of every distinct conv once (MONOCON_HIP_AUTOTUNE=0: heuristic)
    std::map<std::vector<int>, int> tuned;   // conv signature -> shape id
    float *loss_ws = nullptr;   // focal partials + small reduction scratch
    // fused optimizer tables (device)
    mc::OptTensor *opt_tab = nullptr;
    mc::OptChunk *opt_chunks = nullptr;
    int opt_nchunks = 0, opt_ntensors = 0;
    float *opt_ws = nullptr;    // partials + [norm, coef]
    // train-step plan (mc_train_plan.hip)
    TrainState *train = nullptr;
    size_t train_bytes = 0;   // device memory owned by the train plan
    // mc_query_workspace: plan builders run "dry" -- buffers are counted, not allocated (fake addresses that are never
    // dereferenced), nothing is launched or cached
    bool dry_alloc = false;
    size_t dry_next = 0;
    unsigned long long train_generation = 0;   // id of the forward whose activations the train plan holds (0: none)
    void (*train_free)(TrainState *) = nullptr;
    unsigned long long bind_gen = 0;
    bool pack_clean = false;   // packed panels match the bound parameters (cleared by bind / optimizer step)
    // job tables of mc_pack_params (forward panels of all layers, BatchNorm folds): rebuilt when the binding or the
    // precision mode changes, launched as one grid each
    mc::PackBatch fwd_pack;
    mc::FoldBatch folds;
    unsigned long long pack_tab_gen = ~0ull;
    int pack_tab_prec = -1;
    // RCCL communicator owned by the handle (mc_comm_init)
    CommState *comm = nullptr;
    void (*comm_free)(CommState *) = nullptr;
    // train plan: all target tensors / all regression-gradient maps live in one arena each, so the
    // per-step zero fill is one memset instead of 17 + 8 (mc_make_targets / mc_losses_backward)
    void *tgt_arena = nullptr, *dp_arena = nullptr;
    size_t tgt_arena_bytes = 0, dp_arena_bytes = 0;
};

extern std::string g_create_err;

static inline int fail(mc_handle *h, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return -1;
}

// device scratch of the op-level (test) entry points: released on every return path
struct ScratchBuf {
    void *p = nullptr;
    ScratchBuf() = default;
    ScratchBuf(const ScratchBuf &) = delete;
    ScratchBuf &operator=(const ScratchBuf &) = delete;
    ~ScratchBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

#define HIPCHK(h, expr)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(h, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Workgroup shape for a fused conv launch: h->force_cfg if set, else (autotune on) the fastest of the
// candidate shapes timed on the device with these very arguments, cached per conv signature; else
// the static heuristic.  The accumulation order of an output element does not depend on the shape,
// so the choice never changes results (per-image statistics partials are regrouped, not reordered
// within a partial).  `a.stats` must be null while tuning (callers attach it afterwards).
int mc_choose_conv_cfg(mc_handle *h, const mc::ConvArgs &a, int ks, int stride);

