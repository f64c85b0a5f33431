// Train-step plan: train-mode forward (BatchNorm batch statistics, running-stat updates,
// targets, losses) and the full backward pass, as two lists of kernel launches built once per
// input shape and replayed on the caller's stream.
//
// Replaces what autograd does for the reference's `_, loss_dict = self.model(data_dict);
// total_loss.backward()` (engine/monocon_engine.py:84-86): the graph here is static, gradients
// of tensors with several consumers (tree children, residuals, neck skips) accumulate in place
// through the fused conv's residual input, and every conv uses the MFMA dgrad (= the forward
// kernel on transposed/flipped panels) and the MFMA split-K wgrad.
#include <atomic>
#include <deque>
#include <functional>

#include "mc_internal.h"

using namespace mc;

namespace {

using Fn = std::function<int(mc_handle *, hipStream_t)>;

struct TNode {
    Tensor t;
    float *g = nullptr;
    bool ginit = false, needs_grad = true;
    // the launch that wrote g last, when that is a generic stride-1 data-gradient conv (else null): its epilogue
    // sees the COMPLETE gradient of this map and can take over the reductions of the BatchNorm backward
    ConvArgs *last_conv = nullptr;
    // ... or, when it was a max-pool backward that accumulated into g (round 6): that launch can mask the total and leave the
    // BatchNorm-backward partials (launch_maxpool2_bwd's stats_partial)
    struct PoolBwdArgs *last_pool = nullptr;
    float **last_deconv_stats = nullptr;      // ... or the fused depthwise-deconv backward (this map's only consumer): where to attach `stats`
    // LAZY activation (round 6, precision mode 3): the post-BatchNorm map z = act(la[c] * y + lb[c]) is never written --
    // t.p is the producing layer's raw conv output y, and every consumer forms z while it loads its operand (ConvSrc::la
    // in conv_mfma.h; act = ReLU when lrelu).  null: t.p holds the values.  TB::materialise() turns a lazy node into a
    // stored one (an affine_act pass appended to the forward) for a consumer that cannot form it.
    const float *la = nullptr, *lb = nullptr;
    bool lrelu = true;
};

struct PoolBwdArgs {       // a max-pool backward launch (stable address: bn_backward may still attach `stats`)
    const float *x, *dout, *la, *lb;
    float *dx, *stats;
    int B, H, W, C, acc;
};

struct PackJob {           // dgrad panel refreshed from the master weights before every forward
    const float *w;
    int Cout, CinTotal, k, c_off, Cs, CsP, CoutPad;
    int cls;           // -1: stride-1 panel; 0..3: output-parity class of a stride-2 data gradient
    float *dst;
    void *dst16;       // bf16 / fp16 piece planes of the panel (precision modes 1..3) or null
    unsigned *amax;    // mode 3: max-|w| slot of the master weight
};

enum RecKind { REC_STEM, REC_CONV, REC_POOL, REC_DECONV, REC_HEAD };
struct Rec {
    RecKind kind;
    ConvLayer *L = nullptr;
    DeconvLayer *D = nullptr;
    std::vector<int> srcs;
    int res = -1, z = -1, in = -1;
    bool relu = true, dead = false;
    Tensor y;
    float *mean = nullptr, *rstd = nullptr;
    float *ca = nullptr, *cb = nullptr;   // forward BN coefficients z = act(ca*y + cb (+ res))
    unsigned *zbits = nullptr;            // residual layers: the ReLU mask of z, bit-packed by the forward (ConvArgs::bm_zbits)
    std::string bn;
};

}  // namespace

struct TrainState {
    int B = 0, H = 0, W = 0;
    unsigned long long bind_gen = 0;
    std::vector<void *> bufs;
    size_t bytes = 0;
    std::vector<TNode> nodes;
    std::vector<Rec> recs;
    std::vector<Fn> fwd, bwd;
    std::deque<ConvArgs> dgrads;     // data-gradient launches (stable addresses: bn_backward may still patch them)
    std::deque<PoolBwdArgs> pool_bwds;
    std::deque<float *> deconv_stats;      // per fused deconv backward: its statistics buffer (null until bn_backward attaches one)
    // backward closures that only produce weight gradients (nothing downstream in the step reads them) run
    // on a second stream: MFMA-bound wgrad overlaps the HBM-bound BN passes and the tails of the dgrad chain
    std::vector<char> bwd_side;
    // data parallelism: gradient bucket b (mc_internal.h) is complete once backward closure bucket_after[b] has been
    // enqueued (-1: no closure writes into it)
    int bucket_after[MC_NUM_GRAD_BUCKETS] = {-1, -1, -1, -1};
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> side_ev;
    hipEvent_t side_done = nullptr;
    // gradient-map pool (see TB::g_acquire): before main-stream closure `first` writes into a recycled buffer it waits for
    // side-stream closure number `second` (the weight gradient that read the buffer's previous content)
    unsigned *img_amax = nullptr;      // mode 3: max |image| slot (written by the forward stem, read by its weight gradient)
    std::map<int, int> wait_side;
    std::vector<hipEvent_t> side_fin;
    bool dual = true;
    std::vector<PackJob> packs;
    mc::PackBatch pack_batch;        // the data-gradient panels of `packs`, one grid per forward
    std::vector<Fn> pack_fns;
    // per-call external pointers
    const float *img = nullptr;
    mc_labels labels{};
    float *preds[10] = {nullptr};
    float *losses = nullptr;
    const float *grad_losses = nullptr;
    int pad_h = 0, pad_w = 0, max_objs = 30;
    // head-only plan (mc_head_forward_train): the neck output comes in as an external NCHW tensor, its gradient
    // goes out the same way
    bool head_only = false;
    const float *feat_ext = nullptr;
    float *gfeat_ext = nullptr;
    bool skip_feat_dgrad = false;    // head-only plan, mc_head_backward(grad_feat = NULL)
    int feat_dgrad_closure = -1;     // index in `bwd` of the data gradient into the external feat node
    // precision mode 3: one max-|x| slot per activation / gradient tensor, zeroed at the start of every forward
    unsigned *amax_arena = nullptr;
    int amax_used = 0;
    // plan-owned
    mc_targets targets{};
    float *dpred[10] = {nullptr};
    bool ok = true;
};

// generation of the activations the plan currently holds: bumped by every mc_forward_train on the handle (the plan
// keeps ONE set of saved activations, so mc_backward always differentiates the LATEST forward)
static std::atomic<unsigned long long> g_train_generation{0};

static void train_free(TrainState *t) {
    if (!t) return;
    for (hipEvent_t e : t->side_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : t->side_fin)
        if (e) (void)hipEventDestroy(e);
    if (t->side_done) (void)hipEventDestroy(t->side_done);
    if (t->side) (void)hipStreamDestroy(t->side);
    for (void *q : t->bufs) (void)hipFree(q);
    delete t;
}

namespace {

struct TB {   // train plan builder
    mc_handle *h;
    TrainState *ts;
    std::map<int, int> pooled;      // node -> its 2x2 max-pooled node
    void *last_panel16 = nullptr;   // bf16 twin of the panel the last pack_job() made

    static constexpr int AMAX_SLOTS = 1024;
    // bn_backward(): the caller consumes (d, y, coef) itself -- no element-wise dY pass (the stem, whose dY is read by its
    // weight gradient only); honoured on the backward-statistics-epilogue path, reported back in did_skip_affine
    bool want_skip_affine = false, did_skip_affine = false;
    float *skip_coef = nullptr;
    unsigned *slot() {        // mode 3: a fresh max-|x| slot (null in the other modes)
        if (h->prec != 3) return nullptr;
        if (!ts->amax_arena) ts->amax_arena = reinterpret_cast<unsigned *>(alloc((size_t)AMAX_SLOTS * AMAX_WORDS));
        if (ts->amax_used >= AMAX_SLOTS) { ts->ok = false; h->err = "train plan: amax slot table overflow"; return nullptr; }
        return ts->amax_arena ? ts->amax_arena + (size_t)(ts->amax_used++) * AMAX_WORDS : nullptr;
    }
    unsigned *w_slot(const float *w_master) {
        if (h->prec != 3) return nullptr;
        auto it = h->w_amax_of.find(w_master);
        if (it == h->w_amax_of.end()) { ts->ok = false; h->err = "train plan: no max-|w| slot for a master weight"; return nullptr; }
        return it->second;
    }
    float *alloc(size_t n) {
        float *p = nullptr;
        void *q = nullptr;
        const size_t bytes = (n ? n : 1) * sizeof(float);
        if (h->dry_alloc) {                   // mc_query_workspace: count only (never dereferenced)
            h->dry_next += (bytes + 255) / 256 * 256;
            ts->bytes += bytes;
            return reinterpret_cast<float *>((uintptr_t)0x100000 + h->dry_next);
        }
        if (hipMalloc(&q, bytes) != hipSuccess || hipMemset(q, 0, bytes) != hipSuccess) {
            ts->ok = false;
            h->err = "train plan: out of device memory";
            return nullptr;
        }
        ts->bufs.push_back(q);
        ts->bytes += bytes;
        p = static_cast<float *>(q);
        return p;
    }
    // activation / gradient maps the autotuner times kernels on: optionally ReLU-shaped noise instead of zeros (see
    // launch_noise_fill).  MEASURED (round 4, one session, B=32): 56.52 / 56.56 ms per step tuned on zeros, 56.51 / 56.62
    // tuned on noise -- the ranking of the shapes does not depend on it; off by default (MONOCON_HIP_TUNE_NOISE=1).
    bool tune_noise = [] { const char *e = std::getenv("MONOCON_HIP_TUNE_NOISE"); return e && std::atoi(e) != 0; }();
    int wres_bwd = [] { const char *e = std::getenv("MONOCON_HIP_WRES_BWD"); return e ? std::atoi(e) : 0; }();
    float *alloc_map(size_t n) {
        float *p = alloc(n);
        if (p && !h->dry_alloc && h->autotune && tune_noise && n >= 4096)
            (void)launch_noise_fill(p, n, (unsigned)ts->bufs.size() * 7919u, nullptr);
        return p;
    }
    // ---- gradient maps.  The backward closures are BUILT in the order they run, so the life of a map's gradient is known
    // while building: it starts at its first writer (a data-gradient conv, a pooling / deconv backward, the residual
    // share of an affine pass) and ends with the layer that produced the map (whose affine pass turns dZ into dY in
    // place; dY is then read by that layer's data- and weight-gradient launches).  Buffers are handed out at the first
    // write and returned after the producing layer, oldest first; a buffer whose last reader ran on the weight-gradient
    // stream carries that closure's number, and its next first writer waits for it (TrainState::wait_side).
    // MONOCON_HIP_GRAD_POOL=0: one private buffer per map.  Measured at B=32 (one session, scratch/ab/pool_ab.sh):
    // 37.7 GB / 59.77 ms without the pool, 32.9 GB / 60.09 ms recycling immediately, 33.8 GB / 59.89 ms with two buffers
    // of head start (the default).
    struct PoolBuf { float *p; int side_k; };
    std::map<size_t, std::deque<PoolBuf>> gpool;
    bool pool_on = [] { const char *e = std::getenv("MONOCON_HIP_GRAD_POOL"); return !e || std::atoi(e) != 0; }();
    // a returned buffer is handed out again only once `pool_cool` younger ones of its size wait behind it: the weight
    // gradient that still reads it has then had that many layers of head start, and the wait is a formality
    int pool_cool = [] { const char *e = std::getenv("MONOCON_HIP_GRAD_POOL_COOL"); return e ? std::atoi(e) : 2; }();
    float *g_acquire(int node) {
        TNode &n = ts->nodes[node];
        if (n.g) return n.g;
        const size_t ne = n.t.numel();
        auto &q = gpool[ne];
        if (pool_on && (int)q.size() > pool_cool) {
            const PoolBuf b = q.front();
            q.pop_front();
            if (b.side_k >= 0) {
                auto it = ts->wait_side.find((int)ts->bwd.size());
                if (it == ts->wait_side.end()) ts->wait_side[(int)ts->bwd.size()] = b.side_k;
                else it->second = std::max(it->second, b.side_k);
            }
            n.g = b.p;
        } else {
            n.g = alloc_map(ne);
        }
        return n.g;
    }
    void g_release(int node, int side_k) {
        TNode &n = ts->nodes[node];
        if (!pool_on || !n.g) return;
        gpool[n.t.numel()].push_back({n.g, side_k});
        n.g = nullptr;
    }
    int last_side_closure() const {       // number (among the side-stream closures) of the last one pushed, -1 if none
        int k = 0;
        for (char c : ts->bwd_side) k += c != 0;
        return k - 1;
    }
    // MONOCON_HIP_LAZY_Z (bit mask, default 15): which post-BatchNorm activations are never stored (TNode::la).  1: the
    // BatchNorm + ReLU outputs without residual (stem, level0 / level1, BasicBlock conv1, Root, neck proj / node); 2: a
    // Tree's `project` branch (BatchNorm without ReLU, consumed as the residual of the block beside it); 0: every
    // activation is stored (rounds 1-5).  Outputs of a residual add are always stored.
    int lazy_mask = [] { const char *e = std::getenv("MONOCON_HIP_LAZY_Z"); return e ? std::atoi(e) : 3; }();
    // A lazy map costs its MFMA-kernel consumers one fma + one v_med3 per staged element (measured, B = 32: a 3x3 weight
    // gradient +0.03 ms, a conv +0.01 ms per launch) and saves one element-wise pass over the map (2 x its bytes at ~5.3
    // TB/s): worth it for the large maps only.  MONOCON_HIP_LAZY_MIN: elements per image from which a ReLU'd map whose
    // consumers are convolutions is lazy; maps with element-wise consumers only (neck proj -> deconv, project -> residual)
    // always are.  The default sits between the 256-channel 24x80 maps (491 520 elements: stored) and the 128-channel maps
    // of the quarter-resolution level (983 040 at the benchmark's width 1280, 958 464 at KITTI's 1248: lazy): with 983 040
    // itself a KITTI batch lost those maps and 0.3 ms per step (48.66 -> 48.35 ms at 384x1248, two runs each).
    long long lazy_min = [] { const char *e = std::getenv("MONOCON_HIP_LAZY_MIN"); return e ? std::atoll(e) : 900000ll; }();
    int n_lazy = 0, n_materialised = 0;
    // a consumer that cannot form a lazy activation on load: store it after all (one element-wise pass appended to the
    // forward at this point of the build -- i.e. before the consumer's own launch -- and the node is an ordinary one from
    // here on; earlier consumers keep reading y).  Its max-|z| slot already holds bn_finalize's bound.
    void materialise(int node_i) {
        TNode &n = ts->nodes[node_i];
        if (!n.la) return;
        float *z = alloc_map(n.t.numel());
        const float *y = n.t.p, *la = n.la, *lb = n.lb;
        const int B = n.t.B, C = n.t.C, rl = n.lrelu ? 1 : 0;
        const size_t rows = (size_t)n.t.H * n.t.W;
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
            HIPCHK(hh, launch_affine_act(y, la, lb, nullptr, B, rows, C, 0, rl, z, st, nullptr));
            return 0;
        });
        if (std::getenv("MONOCON_HIP_PLAN_DEBUG"))
            fprintf(stderr, "[plan] lazy node %d (%d ch %dx%d) materialised for a consumer that cannot form it on load\n", node_i, C, n.t.H, n.t.W);
        n.t.p = z; n.la = n.lb = nullptr;
        ++n_materialised;
    }
    void fill_src(ConvSrc &s, int node_i) {
        const TNode &n = ts->nodes[node_i];
        s.p = n.t.p; s.C = n.t.C; s.la = n.la; s.lb = n.lb;
    }
    int node(int B, int H, int W, int C, bool needs_grad = true, bool storage = true) {
        TNode n;
        n.t.B = B; n.t.H = H; n.t.W = W; n.t.C = C;
        n.t.p = storage ? alloc_map(n.t.numel()) : nullptr;
        n.t.amax = slot();
        n.needs_grad = needs_grad;
        if (needs_grad && !pool_on) n.g = alloc(n.t.numel());
        ts->nodes.push_back(n);
        return (int)ts->nodes.size() - 1;
    }
    float *P(const std::string &name) {
        auto it = h->bound.find(name);
        if (it == h->bound.end()) { ts->ok = false; h->err = "parameter not bound: " + name; return nullptr; }
        return static_cast<float *>(it->second.ptr);
    }
    float *G(const std::string &name) { return P(name + "#grad"); }
    long long *NBT(const std::string &name) { return reinterpret_cast<long long *>(P(name)); }
    ConvLayer &L(const std::string &n) {
        auto it = h->convs.find(n);
        if (it == h->convs.end()) { ts->ok = false; h->err = "no layer " + n; static ConvLayer d; return d; }
        return it->second;
    }

    double *fold_scratch(int nb, int C) {
        const size_t nd = partial_fold_doubles(nb, C);
        return nd ? reinterpret_cast<double *>(alloc(nd * 2)) : nullptr;
    }

    // ---------------------------------------------------------------- forward pieces
    void bn_train_ops(const Tensor &y, const float *stats, int nb, int cstride, const std::string &bn, float eps,
                      float mom, float *a, float *b, float *mean, float *rstd, const unsigned *ymax = nullptr,
                      unsigned *zmax = nullptr, int zrelu = 1) {
        float *g = P(bn + ".weight"), *be = P(bn + ".bias"), *rm = P(bn + ".running_mean"), *rv = P(bn + ".running_var");
        long long *nbt = NBT(bn + ".num_batches_tracked");
        const double n = (double)y.B * y.H * y.W;
        const int C = y.C;
        double *fold = fold_scratch(nb, C);
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
            HIPCHK(hh, launch_bn_finalize(stats, nb, cstride, n, C, rm, g, be, eps, mom, rm, rv, nbt, a, b, mean, rstd, st, fold,
                                          ymax, zmax, zrelu));
            return 0;
        });
    }

    int conv_bn(ConvLayer &Lr, const std::vector<int> &srcs, int res, bool relu, bool dead = false, bool elementwise_consumers = false,
                bool never_lazy = false) {
        const Tensor s0 = ts->nodes[srcs[0]].t;   // by value: node() below may reallocate ts->nodes
        const int B = s0.B;
        const int Ho = (s0.H + 2 * (Lr.ks / 2) - Lr.ks) / Lr.stride + 1, Wo = (s0.W + 2 * (Lr.ks / 2) - Lr.ks) / Lr.stride + 1;
        Rec r;
        r.kind = REC_CONV; r.L = &Lr; r.srcs = srcs; r.res = res; r.relu = relu; r.dead = dead; r.bn = Lr.bn;
        r.y.B = B; r.y.H = Ho; r.y.W = Wo; r.y.C = Lr.cout;
        r.y.p = alloc(r.y.numel());
        // lazy output: BatchNorm (+ ReLU) without residual in mode 3 (the convs of every kernel family leave max |y|)
        const bool lazy = !dead && !never_lazy && res < 0 && h->prec == 3 && (lazy_mask & (relu ? 1 : 2)) != 0 &&
                          (!relu || elementwise_consumers || (long long)Ho * Wo * Lr.cout >= lazy_min);
        r.z = dead ? -1 : node(B, Ho, Wo, Lr.cout, true, !lazy);
        ConvArgs a{};
        a.nsrc = (int)srcs.size();
        int cin = 0;
        for (int i = 0; i < a.nsrc; ++i) {
            if (ts->nodes[srcs[i]].la && !ts->nodes[srcs[i]].lrelu) materialise(srcs[i]);     // (a conv forms ReLU'd maps only)
            fill_src(a.src[i], srcs[i]);
            cin += a.src[i].C;
        }
        if (cin != Lr.cin) { ts->ok = false; h->err = "train plan: channel mismatch at " + Lr.conv; }
        a.B = B; a.Hin = s0.H; a.Win = s0.W; a.Hout = Ho; a.Wout = Wo; a.Cin = cin; a.Cout = Lr.cout; a.CoutP = Lr.coutp;
        a.wpk = Lr.wpk; a.out = r.y.p; a.out_ld = Lr.cout;
        a.wpk16 = Lr.wpk16; a.prec = h->prec;
        if (h->prec == 3) {
            for (int i = 0; i < a.nsrc; ++i) a.amax_in[i] = ts->nodes[srcs[i]].t.amax;
            a.amax_w = Lr.w_amax;
        }
        {
            bool any_lazy = false;
            for (int i = 0; i < a.nsrc; ++i) any_lazy |= a.src[i].la != nullptr;
            if (any_lazy && !conv_lazy_capable(a, Lr.ks, Lr.stride)) {
                for (int i = 0; i < a.nsrc; ++i) { materialise(srcs[i]); fill_src(a.src[i], srcs[i]); }
            }
        }
        unsigned *yslot = lazy ? slot() : nullptr;      // max |y|, left by the conv's epilogue: bn_finalize bounds max |z| with it
        a.amax_out = yslot;
        a.cfg = ts->ok ? mc_choose_conv_cfg(h, a, Lr.ks, Lr.stride) : CFG_128x32;
        const int chunks = conv_chunks_per_image(a.cfg, Ho, Wo);
        float *stats = alloc((size_t)B * chunks * Lr.coutp * 2);
        a.stats = stats;
        a.stat_shift = P(Lr.bn + ".running_mean");
        const int ks = Lr.ks, stride = Lr.stride;
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) { HIPCHK(hh, launch_conv(a, ks, stride, st)); return 0; });
        float *ca = alloc(Lr.cout), *cb = alloc(Lr.cout);
        r.ca = ca; r.cb = cb;
        r.mean = alloc(Lr.cout); r.rstd = alloc(Lr.cout);
        bn_train_ops(r.y, stats, B * chunks, Lr.coutp, Lr.bn, 1e-5f, 0.1f, ca, cb, r.mean, r.rstd, yslot,
                     lazy ? ts->nodes[r.z].t.amax : nullptr, relu ? 1 : 0);
        if (lazy) {
            TNode &zn = ts->nodes[r.z];
            zn.t.p = r.y.p; zn.la = ca; zn.lb = cb; zn.lrelu = relu;
            ++n_lazy;
        } else if (!dead) {
            const TNode &rn = ts->nodes[res >= 0 ? res : 0];
            const float *yp = r.y.p, *rp = res >= 0 ? rn.t.p : nullptr;
            const float *ra = res >= 0 ? rn.la : nullptr, *rb = res >= 0 ? rn.lb : nullptr;     // the residual may be lazy
            const int rrelu = (res >= 0 && rn.lrelu) ? 1 : 0;
            float *zp = ts->nodes[r.z].t.p;
            unsigned *zmax = ts->nodes[r.z].t.amax;
            const size_t rows = (size_t)Ho * Wo;
            const int C = Lr.cout, rl = relu;
            // a residual layer's ReLU mask cannot be recomputed from y alone: the forward leaves it bit-packed for the
            // backward-statistics epilogue of the data gradient that completes this map's gradient (1/32 of the bytes of z;
            // MONOCON_HIP_ZBITS=0: that epilogue reads z as in rounds 1-5)
            const bool zbits_on = [] { const char *e = std::getenv("MONOCON_HIP_ZBITS"); return !e || std::atoi(e) != 0; }();      // (per plan build)
            unsigned *zb = (zbits_on && res >= 0 && relu && C % 32 == 0) ? reinterpret_cast<unsigned *>(alloc((size_t)B * rows * (C / 32))) : nullptr;
            r.zbits = zb;
            ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
                HIPCHK(hh, launch_affine_act(yp, ca, cb, rp, B, rows, C, 0, rl, zp, st, zmax, ra, rb, rrelu, zb));
                return 0;
            });
        }
        ts->recs.push_back(r);
        return r.z;
    }

    int pool(int x) {
        const Tensor t = ts->nodes[x].t;           // by value (see conv_bn)
        auto it = pooled.find(x);
        if (it != pooled.end()) return it->second;
        if (ts->nodes[x].la && !ts->nodes[x].lrelu) materialise(x);
        const Tensor tx = ts->nodes[x].t;
        const int o = node(t.B, t.H / 2, t.W / 2, t.C, true);
        ts->nodes[o].t.amax = t.amax;          // max |pool(x)| <= max |x|: the input's slot serves
        const float *ip = tx.p, *la = ts->nodes[x].la, *lb = ts->nodes[x].lb;      // (lazy x: pooled over relu(la * y + lb))
        float *op = ts->nodes[o].t.p;
        const int B = t.B, H = t.H, W = t.W, C = t.C;
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
            HIPCHK(hh, launch_maxpool2(ip, B, H, W, C, op, st, la, lb));
            return 0;
        });
        Rec r;
        r.kind = REC_POOL; r.in = x; r.z = o;
        ts->recs.push_back(r);
        pooled[x] = o;
        return o;
    }

    int deconv(DeconvLayer &D, int x) {
        if (ts->nodes[x].la && !ts->nodes[x].lrelu) materialise(x);
        const Tensor t = ts->nodes[x].t;           // by value (see conv_bn)
        const int o = node(t.B, t.H * 2, t.W * 2, t.C, true);
        const float *ip = t.p, *w = D.wpk, *la = ts->nodes[x].la, *lb = ts->nodes[x].lb;
        float *op = ts->nodes[o].t.p;
        unsigned *omax = ts->nodes[o].t.amax;
        const int B = t.B, H = t.H, W = t.W, C = t.C;
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
            HIPCHK(hh, launch_deconv4(ip, B, H, W, C, w, op, st, omax, la, lb));
            return 0;
        });
        Rec r;
        r.kind = REC_DECONV; r.in = x; r.z = o; r.D = &D;
        ts->recs.push_back(r);
        return o;
    }

    int block(const std::string &n, int x, int residual) {
        const int y = conv_bn(L(n + ".conv1"), {x}, -1, true);
        return conv_bn(L(n + ".conv2"), {y}, residual >= 0 ? residual : x, true);
    }

    int tree(const std::string &n, int levels, int cin, int cout, int stride, bool level_root, int x, std::vector<int> children,
             bool outer_of_nested = false) {
        // reference model/backbone/dla.py:187-205; the outer `project` of a two-level tree only ticks
        // its BN running statistics (its output is recomputed inside the nested tree and never used)
        (void)outer_of_nested;
        const int bottom = stride > 1 ? pool(x) : x;
        if (level_root) children.push_back(bottom);
        if (levels == 1) {
            int residual = bottom;
            if (cin != cout) residual = conv_bn(L(n + ".project.0"), {bottom}, -1, false);
            const int x1 = block(n + ".tree1", x, residual);
            const int x2 = block(n + ".tree2", x1, -1);
            std::vector<int> cat = {x2, x1};
            for (int c : children) cat.push_back(c);
            return conv_bn(L(n + ".root.conv"), cat, -1, true);
        }
        if (cin != cout) conv_bn(L(n + ".project.0"), {bottom}, -1, false, /*dead=*/true);
        const int x1 = tree(n + ".tree1", levels - 1, cin, cout, stride, false, x, {});
        children.push_back(x1);
        return tree(n + ".tree2", levels - 1, cout, cout, 1, false, x1, children);
    }

    // ---------------------------------------------------------------- backward pieces
    void pack_job(const float *w, int Cout, int CinTotal, int k, int c_off, int Cs, int CoutPad, float **dst_out, int *csp_out,
                  int cls = -1) {
        PackJob j;
        j.w = w; j.Cout = Cout; j.CinTotal = CinTotal; j.k = k; j.c_off = c_off; j.Cs = Cs;
        j.CsP = conv_coutp(Cs); j.CoutPad = CoutPad; j.cls = cls;
        const int taps = cls < 0 ? k * k : (1 + (cls >> 1)) * (1 + (cls & 1));
        j.dst = alloc((size_t)taps * CoutPad * j.CsP);
        j.dst16 = (h->prec >= 1 && CoutPad % 8 == 0) ? alloc((3 * (size_t)taps * CoutPad * j.CsP + 1) / 2) : nullptr;
        j.amax = w_slot(w);
        last_panel16 = j.dst16;
        ts->packs.push_back(j);
        *dst_out = j.dst;
        *csp_out = j.CsP;
    }

    // dgrad: g_src (+)= conv_s1(dy (optionally dilated), flipped panel)
    void emit_dgrad(const float *w_master, const Tensor &dy, int Cout_fwd, int CinTotal, int ks, int stride, int c_off, int srcnode,
                    int CoutPad) {
        TNode &sn = ts->nodes[srcnode];
        if (!sn.needs_grad) return;
        g_acquire(srcnode);
        if (stride == 2 && ks == 3) {
            // four output-parity classes, each a small stride-1 window conv over dY that scatters to every
            // second pixel of g_src (no zero-dilated copy of dY, 9 instead of 36 tap-MACs per output quad)
            if (2 * dy.H != sn.t.H || 2 * dy.W != sn.t.W) { ts->ok = false; h->err = "train plan: stride-2 dgrad shape mismatch"; }
            // the 32 -> 16 layer at full resolution (level1): all four classes in one pass of its own kernel (conv_thin.hip;
            // MONOCON_HIP_DGRAD_S2_THIN=0: the four launches below)
            const bool s2_thin = [] { const char *e = std::getenv("MONOCON_HIP_DGRAD_S2_THIN"); return !e || std::atoi(e) != 0; }();
            if (s2_thin && ts->ok && dgrad_s2_thin_ok(h->prec, ks, stride, dy.C, sn.t.C, CinTotal, c_off, dy.amax, w_slot(w_master), dy.H, dy.W)) {
                const float *dyp = dy.p;
                float *gp = sn.g;
                const int B = dy.B, Hd = dy.H, Wd = dy.W, Cd = dy.C, acc = sn.ginit ? 1 : 0;
                const unsigned *dmax = dy.amax, *wmax = w_slot(w_master);
                ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                    HIPCHK(hh, launch_dgrad_s2_thin(dyp, B, Hd, Wd, Cd, w_master, CinTotal, c_off, gp, acc, dmax, wmax, st));
                    return 0;
                });
                sn.ginit = true;
                sn.last_conv = nullptr; sn.last_pool = nullptr;
                return;
            }
            for (int cls = 0; cls < 4; ++cls) {
                const int py = cls >> 1, px = cls & 1;
                float *panel;
                int csp;
                pack_job(w_master, Cout_fwd, CinTotal, ks, c_off, sn.t.C, CoutPad, &panel, &csp, cls);
                ConvArgs d{};
                d.nsrc = 1;
                d.src[0].p = dy.p; d.src[0].C = dy.C;
                d.B = dy.B; d.Hin = dy.H; d.Win = dy.W; d.Hout = dy.H; d.Wout = dy.W;
                d.Cin = dy.C; d.Cout = sn.t.C; d.CoutP = csp; d.wpk = panel;
                d.wpk16 = last_panel16; d.prec = last_panel16 ? h->prec : 0;
                d.amax_in[0] = dy.amax; d.amax_w = w_slot(w_master);
                const int ld = sn.t.C;
                d.out = sn.g + ((size_t)py * sn.t.W + px) * ld; d.out_ld = ld;
                d.o_px = 2 * ld; d.o_row = 2 * sn.t.W * ld; d.o_img = sn.t.H * sn.t.W * ld;
                if (sn.ginit) { d.res = d.out; d.res_ld = ld; d.r_px = d.o_px; d.r_row = d.o_row; d.r_img = d.o_img; }
                const int kcode = (1 + py) * 10 + (1 + px);
                const int kk = kcode == 11 ? 1 : kcode;
                d.cfg = ts->ok ? mc_choose_conv_cfg(h, d, kk, 1) : CFG_128x32;
                ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) { HIPCHK(hh, launch_conv(d, kk, 1, st)); return 0; });
            }
            sn.ginit = true;
            sn.last_conv = nullptr; sn.last_pool = nullptr;
            return;
        }
        float *panel;
        int csp;
        pack_job(w_master, Cout_fwd, CinTotal, ks, c_off, sn.t.C, CoutPad, &panel, &csp);
        const float *dyp = dy.p;
        int Hd = dy.H, Wd = dy.W;
        ConvArgs d{};
        d.nsrc = 1;
        d.src[0].p = dyp; d.src[0].C = dy.C;
        d.B = dy.B; d.Hin = Hd; d.Win = Wd; d.Hout = Hd; d.Wout = Wd;
        d.Cin = dy.C; d.Cout = sn.t.C; d.CoutP = csp; d.wpk = panel;
        d.wpk16 = last_panel16; d.prec = last_panel16 ? h->prec : 0;
        d.amax_in[0] = dy.amax; d.amax_w = w_slot(w_master);
        d.out = sn.g; d.out_ld = sn.t.C;
        if (sn.ginit) { d.res = sn.g; d.res_ld = sn.t.C; }
        if (Hd != sn.t.H || Wd != sn.t.W) { ts->ok = false; h->err = "train plan: dgrad shape mismatch"; }
        d.cfg = ts->ok ? mc_choose_conv_cfg(h, d, ks, 1) : CFG_128x32;
        // The weight-resident kernel (conv_wres.hip) owns its CU -- four waves with the whole register file -- so beside the
        // weight-gradient stream it cannot share one the way the tiled kernels do and the two streams take
        // turns: measured in the step (rocprofv3, round 5) a plain 64 -> 64 data gradient takes ~595 us on it against 389 us
        // on conv_bf16_kernel, although it is the faster kernel alone (244 vs 284 us); its backward-statistics twins take
        // 546 us in the step (303 alone).  One-session A/B of the whole step: 52.61 ms without it in the backward, 53.24 ms
        // with the twins on it.  Backward launches therefore keep the tiled kernels.
        // MONOCON_HIP_WRES_BWD: 0 (default) = never in the backward, 1 = the twins (bn_backward sets the flag), 2 = wherever
        // the autotuner chose it
        if (wres_bwd < 2) d.cfg &= ~CFG_WRES;
        ts->dgrads.push_back(d);
        ConvArgs *dp = &ts->dgrads.back();
        ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) { HIPCHK(hh, launch_conv(*dp, ks, 1, st)); return 0; });
        sn.ginit = true;
        // (the fp32 row kernel has no backward-statistics epilogue; its fp16-pipe replacement for 16 -> 16 layers has)
        sn.last_conv = (d.cfg == CFG_SMALL && !conv_thin_ok(d, ks, 1)) ? nullptr : dp;
        sn.last_pool = nullptr;
    }

    void emit_wgrad(const std::vector<int> &srcs, const Tensor &dy, int dy_ld, int Cout, int ks, int stride, float *dw) {
        WgradArgs a{};
        a.nsrc = (int)srcs.size();
        auto fill = [&] {
            int cin = 0;
            for (int i = 0; i < a.nsrc; ++i) {
                fill_src(a.src[i], srcs[i]);
                a.amax_x[i] = ts->nodes[srcs[i]].t.amax;
                cin += a.src[i].C;
            }
            return cin;
        };
        for (int s_ : srcs)
            if (ts->nodes[s_].la && !ts->nodes[s_].lrelu) materialise(s_);
        int cin = fill();
        a.amax_dy = dy.amax;
        const Tensor &s0 = ts->nodes[srcs[0]].t;
        a.B = s0.B; a.Hin = s0.H; a.Win = s0.W; a.Hout = dy.H; a.Wout = dy.W; a.Cin = cin; a.Cout = Cout;
        a.dy = dy.p; a.dy_ld = dy_ld;
        a.prec = h->prec;
        wgrad_plan(a, ks, stride);
        bool any_lazy = false;
        for (int i = 0; i < a.nsrc; ++i) any_lazy |= a.src[i].la != nullptr;
        if (any_lazy && !wgrad_lazy_capable(a, ks, stride)) {      // X is read by a kernel that cannot form it: store it after all
            for (int s_ : srcs) materialise(s_);
            fill();
        }
        a.partial = alloc(wgrad_partial_floats(a, ks));
        ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) { HIPCHK(hh, launch_wgrad(a, ks, stride, dw, st)); return 0; });
    }

    // BN(+ReLU)(+residual) backward: returns dy (gradient wrt the raw conv output)
    Tensor bn_backward(const Rec &r, const std::string &bn) {
        TNode &zn = ts->nodes[r.z];
        Tensor dy = r.y;
        // dY (the gradient wrt the raw conv output) is written IN PLACE over dZ: the affine pass is elementwise and the
        // gradient of z has no reader after it (the residual branch receives its share in the same pass)
        dy.p = zn.g;
        dy.amax = slot();
        unsigned *dymax = dy.amax;
        const int B = r.y.B, C = r.y.C, rows = r.y.H * r.y.W;
        const float *yp = r.y.p, *gz = zn.g, *zp = zn.t.p, *gamma = P(bn + ".weight"), *mean = r.mean, *rstd = r.rstd;
        float *dg = G(bn + ".weight"), *db = G(bn + ".bias"), *dyp = dy.p;
        float *coef = alloc((size_t)C * 4);
        // ReLU without residual: the mask is recomputed from y (bit-identical to z > 0), z is not read
        const int relu = r.relu ? ((r.res < 0 && r.ca && r.cb) ? 2 : 1) : 0;
        const float *fa = r.ca, *fb = r.cb;
        float *gres = nullptr;
        int gmode = 0;
        if (r.res >= 0 && ts->nodes[r.res].needs_grad) {
            gres = g_acquire(r.res);
            gmode = ts->nodes[r.res].ginit ? 2 : 1;
            ts->nodes[r.res].ginit = true;
            ts->nodes[r.res].last_conv = nullptr; ts->nodes[r.res].last_pool = nullptr;
        }
        const double n = (double)B * rows;
        ConvArgs *lc = zn.last_conv;
        // MONOCON_HIP_BM_EPILOGUE: 0 = never take over the reductions in the data gradient's epilogue (always the reduction
        // pass), N > 1 = only for maps of at most N pixels per image
        {
            static const int bm_mode = [] { const char *e = std::getenv("MONOCON_HIP_BM_EPILOGUE"); return e ? std::atoi(e) : 1; }();
            if (bm_mode == 0 || (bm_mode > 1 && r.y.H * r.y.W > bm_mode)) lc = nullptr;
        }
        if (std::getenv("MONOCON_HIP_PLAN_DEBUG"))
            fprintf(stderr, "[plan] bn_backward %-40s %4d ch %4dx%-4d relu %d res %d last-writer-conv %d\n", bn.c_str(), C, r.y.H,
                    r.y.W, relu, r.res >= 0, lc != nullptr);
        if (lc && lc->out == zn.g && lc->Cout == C && lc->out_ld == C && lc->Hout == r.y.H && lc->Wout == r.y.W && !lc->stats) {
            // the gradient of this map was completed by a data-gradient conv: its epilogue masks it and emits the
            // (sum d, sum d*y) partials per 4x8 patch -- no reduction pass, and the affine pass needs no mask
            const int ppi = conv_chunks_per_image(lc->cfg, r.y.H, r.y.W), nbp = B * ppi, cstride = lc->CoutP;   // per 4x8 patch / per row
            float *partial = alloc((size_t)nbp * cstride * 2);
            lc->stats = partial;
            lc->bm_y = yp; lc->bm_z = zp; lc->bm_a = fa; lc->bm_b = fb; lc->bm_relu = relu;
            lc->bm_zbits = relu == 1 ? r.zbits : nullptr;
            if (wres_bwd >= 1 && !(lc->cfg & (CFG_SMALL | CFG_WS)) && conv_wres_ok(*lc, 3, 1)) lc->cfg |= CFG_WRES;      // (see emit_dgrad)
            if (std::getenv("MONOCON_HIP_PLAN_DEBUG"))
                fprintf(stderr, "[plan]   twin of %-36s cfg %3d  K %4d  Cout %3d  %dx%d  res %d  nsrc %d srcC %d wres %d\n", bn.c_str(), lc->cfg,
                        lc->Cin, lc->Cout, lc->Hout, lc->Wout, lc->res != nullptr, lc->nsrc, lc->src[0].C, (lc->cfg & CFG_WRES) != 0);
            double *fold = fold_scratch(nbp, C);
            if (want_skip_affine && !gres) {
                // the epilogue also leaves max |d| (for the consumer's operand scale); only the coefficients are computed here
                lc->amax_out = dymax;
                ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                    HIPCHK(hh, launch_bn_bwd_finalize(partial, nbp, cstride, n, C, gamma, mean, rstd, dg, db, coef, st, fold));
                    return 0;
                });
                did_skip_affine = true;
                skip_coef = coef;
                return dy;        // .p = the masked gradient d, .amax = max |d|
            }
            ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                HIPCHK(hh, launch_bn_bwd_finalize(partial, nbp, cstride, n, C, gamma, mean, rstd, dg, db, coef, st, fold));
                HIPCHK(hh, launch_affine_bwd(gz, zp, yp, coef, B, (size_t)rows, C, 0, 0, dyp, gres, gmode, st, nullptr, nullptr, nullptr,
                                             nullptr, dymax));
                return 0;
            });
            return dy;
        }
        // the gradient of this map was completed by a max-pool backward over a LAZY map (it holds y, forms z for its window
        // comparison anyway): that launch masks the total and leaves the partials -- no reduction pass (MONOCON_HIP_POOL_STATS=0: off)
        {
            const bool pool_stats = [] { const char *e = std::getenv("MONOCON_HIP_POOL_STATS"); return !e || std::atoi(e) != 0; }();
            PoolBwdArgs *pl = zn.last_pool;
            if (pool_stats && pl && relu == 2 && pl->la == fa && pl->lb == fb && pl->x == yp && pl->acc && pl->dx == zn.g && pl->C == C &&
                pl->H * pl->W == rows && !pl->stats && !want_skip_affine && C % 4 == 0 && 256 % (C / 4) == 0) {
                const int nbp = maxpool2_bwd_blocks(B, pl->H, pl->W, C);
                float *partial = alloc((size_t)nbp * C * 2);
                pl->stats = partial;
                double *fold = fold_scratch(nbp, C);
                if (std::getenv("MONOCON_HIP_PLAN_DEBUG"))
                    fprintf(stderr, "[plan]   statistics of %-36s left by the max-pool backward (%d partial rows)\n", bn.c_str(), nbp);
                ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                    HIPCHK(hh, launch_bn_bwd_finalize(partial, nbp, C, n, C, gamma, mean, rstd, dg, db, coef, st, fold));
                    HIPCHK(hh, launch_affine_bwd(gz, zp, yp, coef, B, (size_t)rows, C, 0, 0, dyp, gres, gmode, st, nullptr, nullptr, nullptr,
                                                 nullptr, dymax));
                    return 0;
                });
                return dy;
            }
        }
        // ... or by the fused backward of the depthwise deconv that is its only consumer (neck proj -> up): same contract
        if (float **ds = zn.last_deconv_stats) {
            const bool dc_stats = [] { const char *e = std::getenv("MONOCON_HIP_DECONV_STATS"); return !e || std::atoi(e) != 0; }();
            if (dc_stats && relu == 2 && zn.la == fa && zn.lb == fb && !*ds && !want_skip_affine && !gres) {
                const int nbp = B * r.y.H;            // one workgroup per (image, row) of the deconv's input
                float *partial = alloc((size_t)nbp * C * 2);
                *ds = partial;
                double *fold = fold_scratch(nbp, C);
                if (std::getenv("MONOCON_HIP_PLAN_DEBUG"))
                    fprintf(stderr, "[plan]   statistics of %-36s left by the deconv backward (%d partial rows)\n", bn.c_str(), nbp);
                ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                    HIPCHK(hh, launch_bn_bwd_finalize(partial, nbp, C, n, C, gamma, mean, rstd, dg, db, coef, st, fold));
                    HIPCHK(hh, launch_affine_bwd(gz, zp, yp, coef, B, (size_t)rows, C, 0, 0, dyp, nullptr, 0, st, nullptr, nullptr, nullptr,
                                                 nullptr, dymax));
                    return 0;
                });
                return dy;
            }
        }
        const int nb = chan_reduce_blocks(B, rows);
        float *partial = alloc((size_t)nb * C * 2);
        double *fold2 = fold_scratch(nb, C);      // (61 440 partial rows at full resolution: 16 workgroups walking them took 87 us)
        ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
            HIPCHK(hh, launch_chan_reduce(yp, gz, zp, nullptr, B, rows, C, 1, relu, partial, C, st, fa, fb));
            HIPCHK(hh, launch_bn_bwd_finalize(partial, nb, C, n, C, gamma, mean, rstd, dg, db, coef, st, fold2));
            HIPCHK(hh, launch_affine_bwd(gz, zp, yp, coef, B, (size_t)rows, C, 0, relu, dyp, gres, gmode, st, fa, fb, nullptr, nullptr,
                                             dymax));
            return 0;
        });
        return dy;
    }
};

}  // namespace

// ------------------------------------------------------------------------------------ build
static TrainState *build_train(mc_handle *h, int B, int H, int W, bool head_only = false) {
    std::unique_ptr<TrainState, void (*)(TrainState *)> tsp(new TrainState(), train_free);
    TrainState *ts = tsp.get();
    ts->B = B; ts->H = H; ts->W = W; ts->bind_gen = h->bind_gen; ts->head_only = head_only;
    TB b{h, ts};
    const int fh = H / 4, fw = W / 4, HW = fh * fw;
    int feat = -1;
    if (h->prec == 3) {      // the max-|x| slots start every forward at zero: their producers only raise them
        (void)b.slot();
        --ts->amax_used;
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
            HIPCHK(hh, hipMemsetAsync(ts->amax_arena, 0, (size_t)ts->amax_used * AMAX_WORDS * sizeof(unsigned), st));
            return 0;
        });
    }
    if (head_only) {
        // the heads on their own (MonoConDenseHeads.forward_train, monocon_heads.py:150-157): the neck output is an
        // external NCHW tensor, copied into the plan's NHWC node; its gradient is copied out after the backward
        feat = b.node(B, fh, fw, 64, true);
        float *fp = ts->nodes[feat].t.p;
        unsigned *fmax = ts->nodes[feat].t.amax;
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
            HIPCHK(hh, launch_nchw_to_nhwc(ts->feat_ext, B, 64, fh, fw, fp, st));
            if (fmax) HIPCHK(hh, launch_absmax(fp, (size_t)B * fh * fw * 64, fmax, st));
            return 0;
        });
    } else {

    // ---- stem (raw conv -> batch stats -> normalise + ReLU)
    Rec stem;
    stem.kind = REC_STEM; stem.bn = "backbone.base_layer.1"; stem.relu = true;
    stem.y.B = B; stem.y.H = H; stem.y.W = W; stem.y.C = 16;
    stem.y.p = b.alloc(stem.y.numel());
    // (mode 3 with the fp16-pipe stem, which leaves max |y|: the stem's activation is lazy like conv_bn's, TNode::la)
    const bool stem_lazy = h->prec == 3 && stem_f16_enabled() && (b.lazy_mask & 1);
    stem.z = b.node(B, H, W, 16, true, !stem_lazy);
    float *ones16 = b.alloc(16), *zeros16 = b.alloc(16);
    {
        std::vector<float> one(16, 1.f);
        if (!h->dry_alloc && hipMemcpy(ones16, one.data(), 64, hipMemcpyHostToDevice) != hipSuccess) ts->ok = false;
        // mode 3: the fp16-pipe stem leaves the (sum, sum of squares) partials per output row itself; the other modes reduce
        // the raw map in a second pass
        const bool fused_stats = h->prec == 3 && stem_f16_enabled();
        const int nb = fused_stats ? B * H : chan_reduce_blocks(B, H * W);
        float *partial = b.alloc((size_t)nb * 16 * 2), *ca = b.alloc(16), *cb = b.alloc(16);
        stem.mean = b.alloc(16); stem.rstd = b.alloc(16);
        float *yp = stem.y.p, *zp = ts->nodes[stem.z].t.p, *rm = b.P(stem.bn + ".running_mean");
        unsigned *zmax = ts->nodes[stem.z].t.amax;
        unsigned *imax = fused_stats ? b.slot() : nullptr;     // max |image|, left by the forward stem for its weight gradient
        ts->img_amax = imax;
        unsigned *ymax = fused_stats ? b.slot() : nullptr;     // max |raw stem output|: operand-scale bound of the fused weight gradient
        stem.y.amax = ymax;
        const float *sw = h->stem_w;
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
            if (fused_stats) {
                HIPCHK(hh, launch_stem_f16(ts->img, B, H, W, sw, ones16, zeros16, yp, st, 0, ymax, partial, rm, imax));
            } else {
                HIPCHK(hh, launch_stem(ts->img, B, H, W, sw, ones16, zeros16, yp, st, 0, hh->prec));
                HIPCHK(hh, launch_chan_reduce(yp, nullptr, nullptr, rm, B, H * W, 16, 0, 0, partial, 16, st));
            }
            return 0;
        });
        b.bn_train_ops(stem.y, partial, nb, 16, stem.bn, 1e-5f, 0.1f, ca, cb, stem.mean, stem.rstd, stem_lazy ? ymax : nullptr,
                       stem_lazy ? zmax : nullptr, 1);
        stem.ca = ca; stem.cb = cb;
        if (stem_lazy) {
            TNode &zn = ts->nodes[stem.z];
            zn.t.p = yp; zn.la = ca; zn.lb = cb; zn.lrelu = true;
            ++b.n_lazy;
        } else {
            ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
                HIPCHK(hh, launch_affine_act(yp, ca, cb, nullptr, B, (size_t)H * W, 16, 0, 1, zp, st, zmax));
                return 0;
            });
        }
        ts->recs.push_back(stem);
    }
    const int x0 = stem.z;
    const int l0 = b.conv_bn(b.L("backbone.level0.0"), {x0}, -1, true);
    const int l1 = b.conv_bn(b.L("backbone.level1.0"), {l0}, -1, true);
    const int l2 = b.tree("backbone.level2", 1, 32, 64, 2, false, l1, {});
    const int l3 = b.tree("backbone.level3", 2, 64, 128, 2, true, l2, {});
    const int l4 = b.tree("backbone.level4", 2, 128, 256, 2, true, l3, {});
    const int l5 = b.tree("backbone.level5", 1, 256, 512, 2, true, l4, {});
    std::vector<int> layers = {l2, l3, l4, l5};
    for (int i = 0; i < 3; ++i) {
        const int j = 4 - i - 2;
        for (int t = 1; t < 4 - j; ++t) {
            const std::string pre = "neck.ida_" + std::to_string(i) + ".", tsn = std::to_string(t);
            const int p = b.conv_bn(b.L(pre + "proj_" + tsn + ".conv"), {layers[j + t]}, -1, true, false, /*elementwise_consumers=*/true);
            const int u = b.deconv(h->deconvs[pre + "up_" + tsn], p);
            // the neck's LAST node is `feat`: its consumers are the fused 64 -> 576 head conv and that conv's weight gradient, the
            // two longest launches of the step -- formed on load it cost them 0.15 + 0.22 ms (alone) to save a 0.095 ms pass:
            // stored (MONOCON_HIP_LAZY_FEAT=1: lazy like the other nodes)
            const char *lf_env = std::getenv("MONOCON_HIP_LAZY_FEAT");                  // read per plan build
            const bool lazy_feat = lf_env && std::atoi(lf_env) != 0;
            const bool is_feat = i == 2 && t == 3;
            layers[j + t] = b.conv_bn(b.L(pre + "node_" + tsn + ".conv"), {layers[j + t - 1], u}, -1, true, false, false, is_feat && !lazy_feat);
        }
    }
    feat = layers[3];
    }   // !head_only

    // ---- heads
    const int CP = NUM_HEADS * HEAD_CH, LD = 80;
    Tensor xh; xh.B = B; xh.H = fh; xh.W = fw; xh.C = CP; xh.p = b.alloc(xh.numel());
    // (round 2: the normalised hidden maps are no longer stored -- head_bwd_kernel recomputes relu(scale*x + shift) from the
    //  conv output it reads anyway: one 2.3 GB write and one 2.3 GB read per step less)

    Tensor raw; raw.B = B; raw.H = fh; raw.W = fw; raw.C = LD; raw.p = b.alloc(raw.numel());
    AttnTrainArgs at{};
    static const char *HN[NUM_HEADS] = {"heatmap_head", "wh_head", "offset_head", "center2kpt_offset_head", "kpt_heatmap_head",
                                        "kpt_heatmap_offset_head", "dim_head", "depth_head", "dir_feat"};
    AttnGradPtrs gp{};
    for (int hd = 0; hd < NUM_HEADS; ++hd) {
        const std::string an = std::string("head.") + HN[hd] + ".1";
        at.rm[hd] = b.P(an + ".running_mean"); at.rv[hd] = b.P(an + ".running_var");
        at.nbt[hd] = b.NBT(an + ".num_batches_tracked");
        at.att_w[hd] = b.P(an + ".attn_weights.attention.0.weight");
        at.att_g[hd] = b.P(an + ".attn_weights.attention.1.weight");
        at.att_b[hd] = b.P(an + ".attn_weights.attention.1.bias");
        at.att_rm[hd] = b.P(an + ".attn_weights.attention.1.running_mean");
        at.att_rv[hd] = b.P(an + ".attn_weights.attention.1.running_var");
        at.att_nbt[hd] = b.NBT(an + ".attn_weights.attention.1.num_batches_tracked");
        at.weight_[hd] = b.P(an + ".weight_"); at.bias_[hd] = b.P(an + ".bias_");
        gp.d_weight_[hd] = b.G(an + ".weight_"); gp.d_bias_[hd] = b.G(an + ".bias_");
        gp.d_att_w[hd] = b.G(an + ".attn_weights.attention.0.weight");
        gp.d_att_g[hd] = b.G(an + ".attn_weights.attention.1.weight");
        gp.d_att_b[hd] = b.G(an + ".attn_weights.attention.1.bias");
    }
    ConvArgs c3{};
    {
        c3.nsrc = 1;
        b.fill_src(c3.src[0], feat);
        const TNode &fn = ts->nodes[feat];
        c3.B = B; c3.Hin = fh; c3.Win = fw; c3.Hout = fh; c3.Wout = fw; c3.Cin = 64; c3.Cout = CP; c3.CoutP = h->head3.coutp;
        c3.wpk = h->head3.wpk; c3.bias = h->head_bias; c3.out = xh.p; c3.out_ld = CP; c3.cfg = h->head3.cfg;
        c3.wpk16 = h->head3.wpk16; c3.prec = h->prec;
        if (h->prec == 3) { c3.amax_in[0] = fn.t.amax; c3.amax_w = h->head3.w_amax; }
        if (c3.src[0].la && (!fn.lrelu || !conv_lazy_capable(c3, 3, 1))) {
            b.materialise(feat);
            b.fill_src(c3.src[0], feat);
        }
        at.chunks = conv_chunks_per_image(c3.cfg, fh, fw);
        at.stat_ld = h->head3.coutp;
        float *stats = b.alloc((size_t)B * at.chunks * at.stat_ld * 2);
        c3.stats = stats; c3.stat_shift = h->head_rm;
        at.stats = stats; at.B = B; at.HW = HW;
        at.stats64 = reinterpret_cast<double *>(b.alloc((size_t)B * at.stat_ld * 4));   // [B][stat_ld][2] doubles
        at.sv_inst = b.alloc((size_t)B * CP * 3); at.mu_r = b.alloc((size_t)CP * 2); at.bn10 = b.alloc(NUM_HEADS * NUM_AFFINE * 2);
        at.that = b.alloc((size_t)B * NUM_HEADS * NUM_AFFINE); at.yatt = b.alloc((size_t)B * NUM_HEADS * NUM_AFFINE);
        at.gamma_p = b.alloc((size_t)B * CP); at.scale = b.alloc((size_t)B * CP); at.shift = b.alloc((size_t)B * CP);
    }
    float *w3dense = b.alloc((size_t)CP * 64 * 9);
    // AttnBN apply + ReLU + the nine 1x1 convs + prediction epilogues in ONE pass over the hidden maps
    // (head_apply_kernel, the inference kernel), which also stores the normalised maps for the backward
    HeadApplyArgs ha{};
    {
        ha.hidden = xh.p; ha.scale = at.scale; ha.shift = at.shift; ha.w = h->head_w1t; ha.b = h->head_b1;
        ha.B = B; ha.HW = HW; ha.z_out = nullptr;
        static const int PCH[10] = {3, 9, 2, 2, 2, 18, 3, 2, 12, 12};
        for (int i = 0; i < 10; ++i) ha.pred_c[i] = PCH[i];     // the prediction pointers are per call (ts->preds)
        // one closure per kernel family so that mc_profile_train attributes the durations correctly
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) { HIPCHK(hh, launch_conv(c3, 3, 1, st)); return 0; });
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
            HIPCHK(hh, launch_attn_train_fwd(at, st));
            HeadApplyArgs a2 = ha;
            for (int i = 0; i < 10; ++i) a2.pred[i] = ts->preds[i];
            HIPCHK(hh, launch_head_apply(a2, st));
            return 0;
        });
    }
    // targets + losses
    {
        mc_targets &T = ts->targets;
        const size_t R = (size_t)B * ts->max_objs;
        // one arena for the targets, one for the regression-gradient maps (single zero fill each)
        auto up = [](size_t n) { return (n + 63) / 64 * 64; };
        const size_t tn[15] = {(size_t)B * 3 * HW, (size_t)B * 9 * HW, R * 2, R * 2, R * 3, R, R, R, R * 18, R * 18,
                               R * 2, R * 18, R / 4 + 1, R * 18, R * 18};
        size_t ttot = 0;
        for (size_t n : tn) ttot += up(n);
        float *ta = b.alloc(ttot);
        float *tp[15];
        { size_t o = 0; for (int i = 0; i < 15; ++i) { tp[i] = ta ? ta + o : nullptr; o += up(tn[i]); } }
        T.center_heatmap_target = tp[0]; T.kpt_heatmap_target = tp[1];
        T.wh_target = tp[2]; T.offset_target = tp[3]; T.dim_target = tp[4];
        T.alpha_cls_target = tp[5]; T.alpha_offset_target = tp[6]; T.depth_target = tp[7];
        T.center2kpt_offset_target = tp[8]; T.kpt_heatmap_offset_target = tp[9];
        T.indices = reinterpret_cast<int64_t *>(tp[10]); T.indices_kpt = reinterpret_cast<int64_t *>(tp[11]);
        T.mask_target = reinterpret_cast<uint8_t *>(tp[12]);
        T.mask_center2kpt_offset = tp[13]; T.mask_kpt_heatmap_offset = tp[14];
        h->tgt_arena = ta; h->tgt_arena_bytes = ttot * sizeof(float);
        static const int PC[10] = {3, 9, 2, 2, 2, 18, 3, 2, 12, 12};
        ts->dpred[0] = b.alloc((size_t)B * PC[0] * HW);
        ts->dpred[1] = b.alloc((size_t)B * PC[1] * HW);
        size_t dtot = 0;
        for (int i = 2; i < 10; ++i) dtot += up((size_t)B * PC[i] * HW);
        float *da = b.alloc(dtot);
        { size_t o = 0; for (int i = 2; i < 10; ++i) { ts->dpred[i] = da ? da + o : nullptr; o += up((size_t)B * PC[i] * HW); } }
        h->dp_arena = da; h->dp_arena_bytes = dtot * sizeof(float);
        ts->fwd.push_back([=](mc_handle *hh, hipStream_t st) {
            if (mc_make_targets(hh, &ts->labels, B, ts->max_objs, ts->pad_h, ts->pad_w, fh, fw, &ts->targets, st)) return -1;
            if (mc_losses(hh, ts->preds, &ts->targets, B, ts->max_objs, fh, fw, ts->losses, st)) return -1;
            return 0;
        });
    }

    // ===================================================================== backward
    // ---- losses -> raw gradients -> heads
    {
        float *draw = b.alloc(raw.numel());
        float *cs1 = b.alloc(colsum_partial_floats((size_t)B * HW, LD));
        float *db1 = b.alloc(NUM_OUT_ROWS), *dw1 = b.alloc((size_t)NUM_OUT_ROWS * HEAD_CH);
        ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
            if (mc_losses_backward(hh, ts->preds, &ts->targets, B, ts->max_objs, fh, fw, ts->grad_losses, ts->dpred, st)) return -1;
            HIPCHK(hh, launch_dpred_pack(ts->dpred, LD, B, HW, draw, st));
            HIPCHK(hh, launch_colsum(draw, (size_t)B * HW, NUM_OUT_ROWS, LD, cs1, db1, st));
            return 0;
        });
        // the nine 1x1 convs backwards in one pass (head_bwd_kernel): weight-gradient partials, the ReLU-masked
        // data gradient d, and the (sum d, sum d*x) partials of the AttnBN backward
        const int nbr = chan_reduce_blocks(B, HW), rb_per_img = nbr / B;
        // MONOCON_HIP_HEAD_DX_FUSE=0: round 4's passes (head_bwd_kernel stores the masked gradient d, affine_bwd_kernel reads it
        // back).  Default: d is never stored -- the AttnBN backward forms it again from the 65 raw-gradient rows
        // (launch_head_dx; 7 fma per element on average against 4.3 GB less traffic per step at B = 32)
        const bool dx_fuse = [] { const char *e = std::getenv("MONOCON_HIP_HEAD_DX_FUSE"); return !e || std::atoi(e) != 0; }();
        float *dh = b.alloc(xh.numel());
        float *dw1p = b.alloc((size_t)nbr * NUM_OUT_ROWS * HEAD_CH);
        float *partial = b.alloc((size_t)nbr * CP * 2), *coef = b.alloc((size_t)B * CP * 4);
        float *dx = dh;        // the AttnBN backward (an elementwise affine pass) runs in place on the masked gradient
        // scatter dw1 / db1 rows to the parameter gradient tensors (rows are in concatenation order)
        const int *rb = head_row_begin();
        struct Seg { float *dst_w, *dst_b; int r0, nr; };
        std::vector<Seg> segs;
        for (int hd = 0; hd < 8; ++hd)
            segs.push_back(Seg{b.G(std::string("head.") + HN[hd] + ".3.weight"), b.G(std::string("head.") + HN[hd] + ".3.bias"),
                               rb[hd], rb[hd + 1] - rb[hd]});
        segs.push_back(Seg{b.G("head.dir_cls.0.weight"), b.G("head.dir_cls.0.bias"), rb[8], 12});
        segs.push_back(Seg{b.G("head.dir_reg.0.weight"), b.G("head.dir_reg.0.bias"), rb[8] + 12, 12});
        CopyBatch segcb;
        for (const Seg &s : segs) {
            if (!segcb.add(dw1 + (size_t)s.r0 * HEAD_CH, s.dst_w, (size_t)s.nr * HEAD_CH) || !segcb.add(db1 + s.r0, s.dst_b, s.nr))
                ts->ok = false;
        }
        {
            const float *xp = xh.p, *w1 = h->head_w1, *zsc = at.scale, *zsh = at.shift;
            ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                HIPCHK(hh, launch_head_bwd(draw, LD, nullptr, xp, w1, B, HW, nbr, dx_fuse ? nullptr : dh, dw1p, partial, st, zsc, zsh));
                HIPCHK(hh, launch_splitk_reduce(dw1p, nbr, 1, NUM_OUT_ROWS, HEAD_CH, dw1, st));
                HIPCHK(hh, launch_copy_batch(segcb, st));
                return 0;
            });
        }
        float *db3 = b.alloc(CP), *dw3 = b.alloc((size_t)CP * 64 * 9);
        float *cs3 = b.alloc((size_t)std::max(affine_bwd_blocks(B, (size_t)HW, CP), nbr) * CP * 2);
        Tensor dxT = xh; dxT.p = dx; dxT.amax = b.slot();
        unsigned *dxmax = dxT.amax;
        {
            const float *xp = xh.p, *w1h = h->head_w1, *zsc2 = at.scale, *zsh2 = at.shift;
            ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                // d is already masked: the AttnBN backward is the plain per-(image, channel) affine map; the same pass
                // leaves the column sums of dx (the 3x3 convs' bias gradients) instead of a second read of dx
                HIPCHK(hh, launch_attn_train_bwd(at, partial, rb_per_img, gp, coef, st));
                if (dx_fuse)
                    HIPCHK(hh, launch_head_dx(draw, LD, xp, w1h, coef, B, HW, nbr, dx, cs3, db3, dxmax, st, zsc2, zsh2));
                else
                    HIPCHK(hh, launch_affine_bwd(dh, nullptr, xp, coef, B, (size_t)HW, CP, 1, 0, dx, nullptr, 0, st, nullptr, nullptr, cs3, db3,
                                                 dxmax));
                return 0;
            });
        }
        // the fused 64 -> 576 weight gradient (2 ms at B = 32, the longest launch of the backward) and the scatter of its
        // result go to the weight-gradient stream like every other layer's: dx is a private buffer, nothing writes it again
        // in this step (round 4: it used to run on the caller's stream, in front of the whole neck / backbone backward)
        ts->bwd_side.resize(ts->bwd.size(), 0);
        b.emit_wgrad({feat}, dxT, CP, CP, 3, 1, dw3);
        const size_t head_wgrad_first = ts->bwd_side.size();
        std::vector<float *> g3w, g3b;
        for (int hd = 0; hd < NUM_HEADS; ++hd) {
            g3w.push_back(b.G(std::string("head.") + HN[hd] + ".0.weight"));
            g3b.push_back(b.G(std::string("head.") + HN[hd] + ".0.bias"));
        }
        CopyBatch g3cb;
        for (int hd = 0; hd < NUM_HEADS; ++hd) {
            if (!g3cb.add(dw3 + (size_t)hd * 64 * 64 * 9, g3w[hd], (size_t)64 * 64 * 9) || !g3cb.add(db3 + hd * 64, g3b[hd], 64))
                ts->ok = false;
        }
        ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) { HIPCHK(hh, launch_copy_batch(g3cb, st)); return 0; });
        (void)head_wgrad_first;
        ts->bwd_side.resize(ts->bwd.size(), 1);
        // dense OIHW (576,64,3,3) copy of the nine head convs for the dgrad panel
        std::vector<const float *> w3;
        for (int hd = 0; hd < NUM_HEADS; ++hd) w3.push_back(b.P(std::string("head.") + HN[hd] + ".0.weight"));
        CopyBatch w3cb;
        for (int hd = 0; hd < NUM_HEADS; ++hd)
            if (!w3cb.add(w3[hd], w3dense + (size_t)hd * 64 * 64 * 9, (size_t)64 * 64 * 9)) ts->ok = false;
        ts->pack_fns.push_back([=](mc_handle *hh, hipStream_t st) { HIPCHK(hh, launch_copy_batch(w3cb, st)); return 0; });
        if (h->prec == 3) h->w_amax_of[w3dense] = h->head3.w_amax;     // the dense copy shares the fused head panel's maximum
        b.emit_dgrad(w3dense, dxT, CP, 64, 3, 1, 0, feat, CP);
        if (head_only) ts->feat_dgrad_closure = (int)ts->bwd.size() - 1;
    }
    // everything enqueued so far writes head gradients (bucket 0)
    ts->bucket_after[0] = (int)ts->bwd.size() - 1;
    // ---- neck + backbone in reverse forward order
    for (int ri = (int)ts->recs.size() - 1; ri >= 0; --ri) {
        const Rec r = ts->recs[ri];
        struct Mark {      // closures pushed while this record is processed complete gradients of its layer group
            TrainState *t; int b; size_t n0;
            ~Mark() { if (t->bwd.size() > n0) t->bucket_after[b] = std::max(t->bucket_after[b], (int)t->bwd.size() - 1); }
        } mark{ts, mc_grad_bucket_of(r.kind == REC_DECONV ? r.D->name : (r.kind == REC_POOL ? std::string("backbone.") : r.bn)),
               ts->bwd.size()};
        if (r.kind == REC_POOL) {
            TNode &in = ts->nodes[r.in];
            const TNode &o = ts->nodes[r.z];
            if (!o.ginit || !in.needs_grad) continue;
            float *gi = b.g_acquire(r.in);
            PoolBwdArgs pa{in.t.p, o.g, in.la, in.lb, gi, nullptr, in.t.B, in.t.H, in.t.W, in.t.C, in.ginit ? 1 : 0};     // (lazy x: its ReLU'd values are compared)
            ts->pool_bwds.push_back(pa);
            PoolBwdArgs *pp = &ts->pool_bwds.back();
            ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                HIPCHK(hh, launch_maxpool2_bwd(pp->x, pp->dout, pp->B, pp->H, pp->W, pp->C, pp->dx, pp->acc, st, pp->la, pp->lb, pp->stats));
                return 0;
            });
            in.last_pool = pp;
            in.ginit = true;
            in.last_conv = nullptr;
            b.g_release(r.z, -1);
        } else if (r.kind == REC_DECONV) {
            TNode &in = ts->nodes[r.in];
            const TNode &o = ts->nodes[r.z];
            if (!o.ginit) continue;
            if (in.ginit) { ts->ok = false; h->err = "train plan: deconv input has more than one consumer"; }
            const float *xp = in.t.p, *go = o.g, *wp = r.D->wpk, *xla = in.la, *xlb = in.lb;
            float *gi = b.g_acquire(r.in), *dw = b.G(r.D->name + ".weight");
            const int Bq = in.t.B, Hq = in.t.H, Wq = in.t.W, Cq = in.t.C;
            float *part = b.alloc(deconv4_bwd_w_partial_floats(Bq, Hq, Cq));
            // MONOCON_HIP_DECONV_FUSE=0: the data gradient and the weight gradient of the depthwise deconv as two passes (rounds 1-5)
            const bool dc_fuse = [] { const char *e = std::getenv("MONOCON_HIP_DECONV_FUSE"); return !e || std::atoi(e) != 0; }();
            ts->deconv_stats.push_back(nullptr);
            float **dstats = &ts->deconv_stats.back();
            ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                if (dc_fuse) {
                    HIPCHK(hh, launch_deconv4_bwd_w(xp, go, Bq, Hq, Wq, Cq, part, dw, st, xla, xlb, wp, gi, *dstats));
                } else {
                    HIPCHK(hh, launch_deconv4_bwd_data(go, Bq, Hq, Wq, Cq, wp, gi, st));
                    HIPCHK(hh, launch_deconv4_bwd_w(xp, go, Bq, Hq, Wq, Cq, part, dw, st, xla, xlb));
                }
                return 0;
            });
            in.ginit = true;
            in.last_conv = nullptr; in.last_pool = nullptr;
            in.last_deconv_stats = (dc_fuse && xla && xlb) ? dstats : nullptr;
            b.g_release(r.z, -1);
        } else if (r.kind == REC_CONV) {
            if (r.dead || !ts->nodes[r.z].ginit) continue;
            Tensor dy = b.bn_backward(r, r.bn);
            ts->bwd_side.resize(ts->bwd.size(), 0);
            b.emit_wgrad(r.srcs, dy, r.L->cout, r.L->cout, r.L->ks, r.L->stride, b.G(r.L->conv + ".weight"));
            ts->bwd_side.resize(ts->bwd.size(), 1);   // the closure(s) emit_wgrad just added
            const float *wm = b.P(r.L->conv + ".weight");
            int c_off = 0;
            for (int s : r.srcs) {
                b.emit_dgrad(wm, dy, r.L->cout, r.L->cin, r.L->ks, r.L->stride, c_off, s, r.L->cout);
                c_off += ts->nodes[s].t.C;
            }
            b.g_release(r.z, b.last_side_closure());     // dZ / dY of this layer: last read by its weight gradient
        } else if (r.kind == REC_STEM) {
            if (!ts->nodes[r.z].ginit) continue;
            // mode 3: the stem's dY has ONE reader, its weight gradient -- which forms it on the fly from (d, y, coefficients)
            // instead of reading what an element-wise pass wrote (MONOCON_HIP_STEM_FUSE=0: the separate pass)
            static const bool stem_fuse = [] { const char *e = std::getenv("MONOCON_HIP_STEM_FUSE"); return !e || std::atoi(e) != 0; }();
            b.want_skip_affine = stem_fuse && h->prec == 3 && ts->img_amax && r.y.amax && W % 4 == 0 && W >= 16;
            b.did_skip_affine = false;
            Tensor dy = b.bn_backward(r, r.bn);
            b.want_skip_affine = false;
            const bool fused = b.did_skip_affine;
            float *part = b.alloc((size_t)stem_wgrad_blocks(B, H, W) * 147 * 16), *dw = b.G("backbone.base_layer.0.weight");
            const float *dyp = dy.p, *yfp = fused ? r.y.p : nullptr, *cfp = fused ? b.skip_coef : nullptr;
            const unsigned *imax = ts->img_amax, *dymax = dy.amax, *yfmax = fused ? r.y.amax : nullptr;
            ts->bwd_side.resize(ts->bwd.size(), 0);
            ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
                HIPCHK(hh, launch_stem_wgrad(ts->img, dyp, B, H, W, part, dw, st, imax, imax ? dymax : nullptr, yfp, cfp, yfmax));
                return 0;
            });
            // MONOCON_HIP_STEM_WGRAD_MAIN=1 (measured, round 6; default off): the stem's weight gradient on the caller's stream.
            // When it can start (after level0's data gradient and the BatchNorm-backward coefficients) the caller's stream has
            // nothing left to do while the weight-gradient stream is busy with level0's weight gradient for another 0.8 ms
            // (rocprofv3: main idle from 46.27 ms of the step, side busy until 47.87) -- yet run side by side the two
            // front-end weight gradients finish no earlier: 48.77 / 48.82 ms per step (three runs each, one session).  Also
            // measured neutral in the same round: the data-gradient panels packed on the weight-gradient stream at the head of
            // the backward instead of on the caller's stream in front of the forward (49.64 / 49.69 ms).
            static const bool stem_main = [] { const char *e = std::getenv("MONOCON_HIP_STEM_WGRAD_MAIN"); return e && std::atoi(e) != 0; }();
            ts->bwd_side.resize(ts->bwd.size(), stem_main ? 0 : 1);
            b.g_release(r.z, stem_main ? -1 : b.last_side_closure());
        }
    }
    if (head_only) {
        const float *gp = ts->nodes[feat].g;
        ts->bwd.push_back([=](mc_handle *hh, hipStream_t st) {
            if (ts->gfeat_ext) HIPCHK(hh, launch_nhwc_to_nchw(gp, B, 64, fh, fw, ts->gfeat_ext, st));
            ts->gfeat_ext = nullptr;      // written once, for the call that asked for it: never a stale pointer later
            return 0;
        });
    }
    if (!ts->ok) return nullptr;
    if (std::getenv("MONOCON_HIP_PLAN_DEBUG"))
        fprintf(stderr, "[plan] lazy activations: %d never stored, %d stored after all (lazy mask %d), %.2f GB\n", b.n_lazy - b.n_materialised,
                b.n_materialised, b.lazy_mask, ts->bytes * 1e-9);
    if (h->dry_alloc) return tsp.release();   // mc_query_workspace: sizes only
    ts->bwd_side.resize(ts->bwd.size(), 0);
    if (const char *e = std::getenv("MONOCON_HIP_DUAL_STREAM")) ts->dual = std::atoi(e) != 0;
    if (ts->dual) {
        size_t nside = 0;
        for (char c : ts->bwd_side) nside += c != 0;
        // MONOCON_HIP_SIDE_PRIORITY = low / high: the weight-gradient stream below / above the caller's stream in the
        // hardware queues' priority order (default: the same)
        int prio = 0, lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = numerically greatest = least urgent
        if (const char *pe = std::getenv("MONOCON_HIP_SIDE_PRIORITY")) prio = pe[0] == 'l' ? lo : (pe[0] == 'h' ? hi : 0);
        if (hipStreamCreateWithPriority(&ts->side, hipStreamNonBlocking, prio) != hipSuccess ||
            hipEventCreateWithFlags(&ts->side_done, hipEventDisableTiming) != hipSuccess) ts->dual = false;
        ts->side_ev.resize(ts->dual ? nside : 0);
        for (auto &ev : ts->side_ev)
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ts->dual = false; break; }
        // one "finished" event per side closure that a recycled gradient buffer waits for
        ts->side_fin.assign(ts->dual ? nside : 0, nullptr);
        for (const auto &w : ts->wait_side)
            if (ts->dual && w.second >= 0 && w.second < (int)nside && !ts->side_fin[w.second] &&
                hipEventCreateWithFlags(&ts->side_fin[w.second], hipEventDisableTiming) != hipSuccess) { ts->dual = false; break; }
    }
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    return tsp.release();
}

// ====================================================================================== C ABI
extern "C" {

// the train plan of this shape (built -- and its convolution shapes autotuned -- on first use)
static TrainState *ensure_train_plan(mc_handle *h, int B, int H, int W, bool head_only) {
    TrainState *ts = h->train;
    if (!ts || ts->B != B || ts->H != H || ts->W != W || ts->bind_gen != h->bind_gen || ts->head_only != head_only) {
        if (ts && h->train_free) h->train_free(ts);
        h->train = nullptr;
        h->tgt_arena = h->dp_arena = nullptr;
        h->train_bytes = 0;
        h->tgt_arena_bytes = h->dp_arena_bytes = 0;
        ts = build_train(h, B, H, W, head_only);
        if (!ts) return nullptr;
        h->train = ts;
        h->train_bytes = ts->bytes;
        h->train_free = train_free;
    }
    return ts;
}

// Build (and autotune) the train plan of a shape WITHOUT running it: no kernel of the step, no collective.  Data-parallel
// start-up: rank 0 calls this, exports its tune table (mc_tune_export), the other ranks import it and build theirs.
int mc_build_train_plan(mc_handle *h, int B, int H, int W) {
    if (!h) return -1;
    if (B < 2 || B > 64) return fail(h, "mc_build_train_plan: batch %d (2..64 per GPU)", B);
    if (H < 32 || W < 32 || (H % 32) || (W % 32)) return fail(h, "mc_build_train_plan: H, W must be multiples of 32");
    if (h->packed_groups != 7) return fail(h, "mc_build_train_plan: bind all parameters and call mc_pack_params first");
    HIPCHK(h, hipSetDevice(h->device));
    return ensure_train_plan(h, B, H, W, false) ? 0 : -1;
}

static int forward_train_impl(mc_handle *h, const float *img, const mc_labels *labels, int B, int H, int W, int max_objs,
                              float *const preds[MC_NUM_PREDS], float *losses, void *stream, bool head_only) {
    if (!h) return -1;
    const char *fn = head_only ? "mc_head_forward_train" : "mc_forward_train";
    if (!img || !labels || !preds || !losses) return fail(h, "%s: null argument", fn);
    if (B < 2 || B > 64) return fail(h, "%s: batch %d (2..64 per GPU; BatchNorm over the attention vector needs >= 2)", fn, B);
    if (H < 32 || W < 32 || (H % 32) || (W % 32)) return fail(h, "%s: H, W must be multiples of 32", fn);
    if (max_objs != 30) return fail(h, "%s: max_objs=%d (the MonoCon configuration uses 30)", fn, max_objs);
    if (head_only ? !(h->packed_groups & 4) : h->packed_groups != 7)
        return fail(h, "%s: bind all %sparameters and call mc_pack_params first", fn, head_only ? "head. " : "");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    TrainState *ts = ensure_train_plan(h, B, H, W, head_only);
    if (!ts) return -1;
    ts->img = img; ts->feat_ext = img; ts->labels = *labels; ts->losses = losses; ts->pad_h = H; ts->pad_w = W; ts->max_objs = max_objs;
    for (int i = 0; i < MC_NUM_PREDS; ++i) {
        if (!preds[i]) return fail(h, "%s: preds[%d] is NULL", fn, i);
        ts->preds[i] = preds[i];
    }
    // refresh derived weights: forward panels (no BN folding in train mode), dgrad panels
    if (!h->pack_clean && mc_pack_params(h, 1, stream)) return -1;   // mc_pack_params / the optimizer step track staleness
    for (auto &f : ts->pack_fns)       // dense head weight copies first: some dgrad panels are cut from them
        if (f(h, st)) return -1;
    if (ts->pack_batch.jobs.size() != ts->packs.size()) {      // first forward of the plan: table of the data-gradient panels
        ts->pack_batch.clear();
        for (const PackJob &j : ts->packs) {
            mc::PackJobDesc d{};
            d.w = j.w; d.dst32 = j.dst; d.dst16 = j.dst16; d.kind = 1;
            d.Cout = j.Cout; d.Cin = j.Cs; d.k = j.k; d.CinTotal = j.CinTotal; d.CoutP = j.CoutPad; d.n_off = 0; d.c_off = j.c_off;
            d.CsP = j.CsP; d.cls = j.cls; d.nsplit = h->prec == 2 ? 3 : (h->prec == 3 ? 2 : 1);
            d.amax = j.amax;
            ts->pack_batch.add(d);
        }
    }
    HIPCHK(h, ts->pack_batch.launch(st));
    h->train_generation = ++g_train_generation;
    ts->gfeat_ext = nullptr;      // (a previous mc_head_backward's output tensor may be gone by now)
    for (auto &f : ts->fwd)
        if (f(h, st)) return -1;
    return 0;
}

int mc_forward_train(mc_handle *h, const float *img, const mc_labels *labels, int B, int H, int W, int max_objs,
                     float *const preds[MC_NUM_PREDS], float *losses, void *stream) {
    return forward_train_impl(h, img, labels, B, H, W, max_objs, preds, losses, stream, false);
}

int mc_head_forward_train(mc_handle *h, const float *feat, const mc_labels *labels, int B, int pad_h, int pad_w, int max_objs,
                          float *const preds[MC_NUM_PREDS], float *losses, void *stream) {
    return forward_train_impl(h, feat, labels, B, pad_h, pad_w, max_objs, preds, losses, stream, true);
}

// sizes of the train plan (mc_query_workspace modes 1 / 2): the builder runs dry, nothing is allocated or launched
int mc_train_query_workspace(mc_handle *h, int B, int H, int W, int head_only, size_t *bytes) {
    if (!h || !bytes) return -1;
    if (B < 2 || B > 64) return fail(h, "mc_query_workspace: train batch %d (2..64 per GPU)", B);
    if (head_only ? !(h->packed_groups & 4) : h->packed_groups != 7)
        return fail(h, "mc_query_workspace: bind the parameters (and their \"#grad\" buffers) and call mc_pack_params first");
    TrainState *cur = h->train;
    if (cur && cur->B == B && cur->H == H && cur->W == W && cur->head_only == (head_only != 0) && cur->bind_gen == h->bind_gen) {
        *bytes = cur->bytes;
        return 0;
    }
    void *ta = h->tgt_arena, *da = h->dp_arena;
    const size_t tb = h->tgt_arena_bytes, db = h->dp_arena_bytes;
    h->dry_alloc = true; h->dry_next = 0;
    TrainState *ts = build_train(h, B, H, W, head_only != 0);
    h->dry_alloc = false;
    h->tgt_arena = ta; h->dp_arena = da; h->tgt_arena_bytes = tb; h->dp_arena_bytes = db;   // (the builder points these at its arenas)
    if (!ts) return -1;
    *bytes = ts->bytes;
    ts->bufs.clear();
    train_free(ts);
    return 0;
}

int mc_train_generation(mc_handle *h, unsigned long long *out) {
    if (!h || !out) return -1;
    *out = h->train_generation;
    return 0;
}

// debugging aid: copy activation node `node` (train plan order: 0 = stem output, 1 = level0, ...) as NCHW;
// which = 0 activation, 1 its gradient.  dims[4] receives (B, C, H, W).
int mc_train_debug_node(mc_handle *h, int node, int which, float *out_nchw, int dims[4], void *stream) {
    if (!h || !h->train) return fail(h, "mc_train_debug_node: no train plan");
    TrainState *ts = h->train;
    if (node < 0 || node >= (int)ts->nodes.size()) return fail(h, "mc_train_debug_node: node %d of %d", node, (int)ts->nodes.size());
    const TNode &n = ts->nodes[node];
    if (dims) { dims[0] = n.t.B; dims[1] = n.t.C; dims[2] = n.t.H; dims[3] = n.t.W; }
    if (!out_nchw) return 0;
    const float *src = which ? n.g : n.t.p;
    if (!src) return fail(h, "mc_train_debug_node: node has no such buffer");
    hipStream_t dst_st = static_cast<hipStream_t>(stream);
    if (!which && n.la) {       // a lazy activation exists nowhere in memory: form it for the caller
        ScratchBuf tmp;
        HIPCHK(h, tmp.alloc(n.t.numel() * sizeof(float)));
        HIPCHK(h, launch_affine_act(n.t.p, n.la, n.lb, nullptr, n.t.B, (size_t)n.t.H * n.t.W, n.t.C, 0, n.lrelu ? 1 : 0, tmp.as<float>(),
                                    dst_st, nullptr));
        HIPCHK(h, launch_nhwc_to_nchw(tmp.as<float>(), n.t.B, n.t.C, n.t.H, n.t.W, out_nchw, dst_st));
        HIPCHK(h, hipStreamSynchronize(dst_st));
        return 0;
    }
    HIPCHK(h, launch_nhwc_to_nchw(src, n.t.B, n.t.C, n.t.H, n.t.W, out_nchw, dst_st));
    return 0;
}

static int backward_impl(mc_handle *h, TrainState *ts, const float *grad_losses, void *stream);

int mc_backward(mc_handle *h, const float *grad_losses, void *stream) {
    if (!h) return -1;
    if (!grad_losses) return fail(h, "mc_backward: grad_losses is NULL");
    if (h->train && h->train->head_only)
        return fail(h, "mc_backward: the handle holds a heads-only plan (mc_head_forward_train): use mc_head_backward");
    TrainState *ts = h->train;
    if (!ts || !ts->img) return fail(h, "mc_backward: call mc_forward_train first");
    return backward_impl(h, ts, grad_losses, stream);
}

static int backward_impl(mc_handle *h, TrainState *ts, const float *grad_losses, void *stream) {
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    ts->grad_losses = grad_losses;
    size_t k = 0;
    bool used_side = false;
    // data parallelism with a communicator owned by the handle: every gradient bucket is exchanged (averaged over the
    // ranks) on the communicator's stream as soon as its last writer has been enqueued
    const bool dp = !ts->head_only && mc_comm_overlap_active(h);
    if (dp && mc_comm_prepare(h)) return -1;
    for (size_t i = 0; i < ts->bwd.size(); ++i) {
        if (ts->skip_feat_dgrad && (int)i == ts->feat_dgrad_closure) continue;
        if (ts->dual && ts->bwd_side[i] && k < ts->side_ev.size()) {
            // everything this closure reads (dY of its layer, forward activations) is ready at this point of
            // the main stream; its outputs (the weight gradient) are first needed after mc_backward
            HIPCHK(h, hipEventRecord(ts->side_ev[k], st));
            HIPCHK(h, hipStreamWaitEvent(ts->side, ts->side_ev[k], 0));
            if (ts->bwd[i](h, ts->side)) return -1;
            if (k < ts->side_fin.size() && ts->side_fin[k]) HIPCHK(h, hipEventRecord(ts->side_fin[k], ts->side));
            {   // debugging aid: MONOCON_HIP_SIDE_SYNC=lo:hi makes the main stream wait for side closures lo <= k < hi
                static const std::pair<int, int> rng = [] {
                    const char *e = std::getenv("MONOCON_HIP_SIDE_SYNC");
                    int lo = 0, hi = 0;
                    if (e && std::sscanf(e, "%d:%d", &lo, &hi) != 2) lo = hi = 0;
                    return std::make_pair(lo, hi);
                }();
                if ((int)k >= rng.first && (int)k < rng.second) {
                    HIPCHK(h, hipEventRecord(ts->side_done, ts->side));
                    HIPCHK(h, hipStreamWaitEvent(st, ts->side_done, 0));
                }
            }
            ++k;
            used_side = true;
        } else {
            if (used_side) {      // a recycled gradient buffer: its previous content was read on the side stream
                auto w = ts->wait_side.find((int)i);
                if (w != ts->wait_side.end() && w->second < (int)ts->side_fin.size() && ts->side_fin[w->second])
                    HIPCHK(h, hipStreamWaitEvent(st, ts->side_fin[w->second], 0));
            }
            if (ts->bwd[i](h, st)) return -1;
        }
        if (dp)
            for (int b = 0; b < MC_NUM_GRAD_BUCKETS; ++b)
                if (ts->bucket_after[b] == (int)i && mc_comm_fire_bucket(h, b, st, used_side ? ts->side : nullptr)) return -1;
    }
    if (used_side) {
        HIPCHK(h, hipEventRecord(ts->side_done, ts->side));
        HIPCHK(h, hipStreamWaitEvent(st, ts->side_done, 0));
    }
    if (dp) {
        for (int b = 0; b < MC_NUM_GRAD_BUCKETS; ++b)       // a bucket no closure of this plan writes into still has to be exchanged
            if (ts->bucket_after[b] < 0 && mc_comm_fire_bucket(h, b, st, nullptr)) return -1;
        if (mc_comm_join(h, st)) return -1;
    }
    return 0;
}

int mc_head_backward(mc_handle *h, const float *grad_losses, float *grad_feat, void *stream) {
    if (!h) return -1;
    TrainState *ts = h->train;
    if (!ts || !ts->head_only || !ts->feat_ext) return fail(h, "mc_head_backward: call mc_head_forward_train first");
    if (!grad_losses) return fail(h, "mc_head_backward: grad_losses is NULL");
    ts->gfeat_ext = grad_feat;
    ts->skip_feat_dgrad = grad_feat == nullptr;      // feat does not require grad: its 3x3 data gradient is not computed
    return backward_impl(h, ts, grad_losses, stream);
}

int mc_profile_train(mc_handle *h, int iters, double ms[3], double flops[3], double bytes[3], int launches[3],
                     void *stream) {
    if (!h || !ms || !flops || !bytes || !launches) return -1;
    TrainState *ts = h->train;
    if (!ts || !ts->img || !ts->grad_losses) return fail(h, "mc_profile_train: run mc_forward_train + mc_backward first");
    if (iters < 1) iters = 1;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = ts->fwd.size() + ts->bwd.size();
    std::vector<hipEvent_t> ev(2 * n);
    for (auto &e : ev) HIPCHK(h, hipEventCreate(&e));
    std::vector<ProfLast> tag(n);
    for (int k = 0; k < 3; ++k) { ms[k] = 0; flops[k] = 0; bytes[k] = 0; launches[k] = 0; }
    int rc = 0;
    for (int it = 0; it < iters && !rc; ++it) {
        size_t i = 0;
        for (auto *list : {&ts->fwd, &ts->bwd})
            for (auto &f : *list) {
                prof_last = {0, 0.0, 0.0};
                (void)hipEventRecord(ev[2 * i], st);
                if (f(h, st)) { rc = -1; break; }
                (void)hipEventRecord(ev[2 * i + 1], st);
                tag[i++] = prof_last;
            }
        if (hipStreamSynchronize(st) != hipSuccess) rc = fail(h, "mc_profile_train: stream error");
        for (size_t j = 0; j < i && !rc; ++j) {
            float t = 0.f;
            (void)hipEventElapsedTime(&t, ev[2 * j], ev[2 * j + 1]);
            const int k = tag[j].kind;
            ms[k] += t; flops[k] += tag[j].flops; bytes[k] += tag[j].bytes; launches[k] += 1;
            if (it == 0 && std::getenv("MONOCON_HIP_PROFILE_DUMP"))
                std::fprintf(stderr, "prof %zu %s kind %d ms %.4f gflop %.3f mb %.2f\n", j, j < ts->fwd.size() ? "fwd" : "bwd", k, t,
                             tag[j].flops * 1e-9, tag[j].bytes * 1e-6);
        }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    for (int k = 0; k < 3; ++k) { ms[k] /= iters; flops[k] /= iters; bytes[k] /= iters; launches[k] /= iters; }
    return rc;
}

}  // extern "C"
