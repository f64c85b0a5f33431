// Data-parallel gradient exchange behind the C-ABI (mc_comm_init / mc_allreduce_grads).
//
// The reference is single-GPU (README.MD:11,15); this is the MI355X-side design: one process per GPU, every rank
// a full replica, ONE exchange per step -- the average of the live gradients over the ranks -- on RCCL over xGMI.
// The library owns the communicator so that a host WITHOUT PyTorch can train data-parallel through the boundary; the
// Python binding uses the same entry points (hipmonocon/dist.py keeps the torch.distributed path, which the tests compare it with).
//
//   * RCCL is reached with dlopen, not at link time: a process that never calls mc_comm_* carries no dependency, and a
//     PyTorch process re-uses the librccl torch already mapped (RTLD_NOLOAD first) instead of a second instance.
//   * the gradients the caller bound as "<key>#grad" are grouped into BUCKETS by layer group, in the order the backward
//     pass completes them (heads + neck, level5, level4, the rest of the backbone).  When the tensors of a bucket sit
//     in one dense address range (the Python binding allocates all gradients in one flat buffer, in parameter order)
//     the bucket is ONE ncclAllReduce(ncclAvg); otherwise a group of per-tensor calls.
//   * overlap: mc_backward launches a bucket on the communicator's own stream as soon as the last launch that writes
//     into it has been enqueued (event from the compute streams), and joins that stream at its end -- the exchange of
//     the early buckets hides behind the remaining backbone backward; only the last bucket's is exposed.
//   * xGMI is point-to-point (7 links x ~153 GB/s per GPU): 78 MB in 4 buckets of 8..32 MB keeps every ring step well
//     above the latency-bound regime while leaving 3 of 4 buckets overlappable.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include "mc_internal.h"

namespace {

// the slice of the RCCL ABI this file uses (rccl.h of ROCm 7.2: ncclUniqueId is 128 opaque bytes, ncclFloat32 = 7,
// ncclSum = 0, ncclAvg = 4, ncclSuccess = 0)
struct UniqueId { char internal[128]; };
typedef void *Comm;
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(Comm *, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, Comm, hipStream_t);
typedef int (*GroupFn)();
typedef const char *(*ErrStrFn)(int);
constexpr int NCCL_FLOAT32 = 7, NCCL_AVG = 4;

struct Rccl {
    void *lib = nullptr;
    std::string path;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    GroupFn group_start = nullptr, group_end = nullptr;
    ErrStrFn err_str = nullptr;
};

Rccl g_rccl;

int load_rccl(mc_handle *h) {
    if (g_rccl.lib) return 0;
    std::vector<std::pair<std::string, int>> tries;
    if (const char *e = std::getenv("MONOCON_HIP_RCCL_LIB")) tries.push_back({e, RTLD_NOW | RTLD_GLOBAL});
    for (const char *n : {"librccl.so", "librccl.so.1"}) tries.push_back({n, RTLD_NOW | RTLD_NOLOAD});   // already mapped (torch)
    for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) tries.push_back({n, RTLD_NOW | RTLD_GLOBAL});
    for (auto &t : tries) {
        void *lib = dlopen(t.first.c_str(), t.second);
        if (!lib) continue;
        Rccl r;
        r.lib = lib; r.path = t.first + ((t.second & RTLD_NOLOAD) ? " (already mapped)" : "");
        r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(lib, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(lib, "ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(lib, "ncclCommDestroy"));
        r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(lib, "ncclAllReduce"));
        r.group_start = reinterpret_cast<GroupFn>(dlsym(lib, "ncclGroupStart"));
        r.group_end = reinterpret_cast<GroupFn>(dlsym(lib, "ncclGroupEnd"));
        r.err_str = reinterpret_cast<ErrStrFn>(dlsym(lib, "ncclGetErrorString"));
        if (r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce && r.group_start && r.group_end) {
            g_rccl = r;
            return 0;
        }
    }
    return fail(h, "mc_comm: librccl.so not found / incomplete (set MONOCON_HIP_RCCL_LIB); dlerror: %s", dlerror());
}

const char *rccl_err(int rc) { return g_rccl.err_str ? g_rccl.err_str(rc) : "?"; }

}  // namespace

struct CommState {
    Comm comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;          // the exchange runs here, beside the compute streams
    hipEvent_t ready = nullptr, ready2 = nullptr, done = nullptr;
    hipEvent_t t_compute = nullptr, t_joined = nullptr;   // timing: compute of the backward enqueued / exchange joined
    bool timed = false;
    bool overlap = true;
    // buckets of the bound "#grad" tensors (rebuilt when the binding changes)
    unsigned long long bind_gen = ~0ull;
    std::vector<GradBucket> buckets;
    size_t total_floats = 0;
    unsigned long long launches = 0;       // all-reduce calls issued so far (introspection / tests)
};

static void comm_free(CommState *c) {
    if (!c) return;
    if (c->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->ready2) (void)hipEventDestroy(c->ready2);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->t_compute) (void)hipEventDestroy(c->t_compute);
    if (c->t_joined) (void)hipEventDestroy(c->t_joined);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// layer group of a parameter name, in the order the backward pass completes them
int mc_grad_bucket_of(const std::string &name) {
    if (name.compare(0, 5, "head.") == 0 || name.compare(0, 5, "neck.") == 0) return 0;
    if (name.compare(0, 15, "backbone.level5") == 0) return 1;
    if (name.compare(0, 15, "backbone.level4") == 0) return 2;
    return 3;
}

static int build_buckets(mc_handle *h, CommState *c) {
    if (c->bind_gen == h->bind_gen && !c->buckets.empty()) return 0;
    struct T { float *p; size_t n; int b; };
    std::vector<T> ts;
    for (auto &kv : h->bound) {
        const std::string &k = kv.first;
        if (k.size() < 6 || k.compare(k.size() - 5, 5, "#grad") != 0 || kv.second.dtype != MC_F32) continue;
        ts.push_back({static_cast<float *>(kv.second.ptr), (size_t)kv.second.numel, mc_grad_bucket_of(k.substr(0, k.size() - 5))});
    }
    if (ts.empty()) return fail(h, "mc_allreduce_grads: no \"<key>#grad\" tensors are bound");
    c->buckets.assign(MC_NUM_GRAD_BUCKETS, GradBucket{});
    c->total_floats = 0;
    for (int b = 0; b < MC_NUM_GRAD_BUCKETS; ++b) {
        GradBucket &g = c->buckets[b];
        std::vector<T> sel;
        for (auto &t : ts) if (t.b == b) sel.push_back(t);
        if (sel.empty()) continue;
        std::sort(sel.begin(), sel.end(), [](const T &x, const T &y) { return x.p < y.p; });
        size_t sum = 0;
        bool dense = true;
        for (size_t i = 0; i < sel.size(); ++i) {
            sum += sel[i].n;
            // dense: the next tensor starts where this one ends, up to the 16-byte alignment padding of a flat buffer
            if (i + 1 < sel.size() && (sel[i + 1].p < sel[i].p + sel[i].n || sel[i + 1].p > sel[i].p + sel[i].n + 3)) dense = false;
        }
        c->total_floats += sum;
        if (dense) {
            g.p = sel.front().p;
            g.n = (size_t)(sel.back().p + sel.back().n - sel.front().p);
        } else {
            for (auto &t : sel) g.parts.push_back({t.p, t.n});
        }
        g.tensors = (int)sel.size();
    }
    // a dense range must not swallow another bucket's tensors (alignment gaps are zero-filled padding, harmless)
    for (int a = 0; a < MC_NUM_GRAD_BUCKETS; ++a)
        for (int b = 0; b < MC_NUM_GRAD_BUCKETS; ++b) {
            if (a == b || !c->buckets[a].p || !c->buckets[b].p) continue;
            const float *a0 = c->buckets[a].p, *a1 = a0 + c->buckets[a].n, *b0 = c->buckets[b].p, *b1 = b0 + c->buckets[b].n;
            if (a0 < b1 && b0 < a1) return fail(h, "mc_allreduce_grads: gradient buckets %d and %d overlap in memory", a, b);
        }
    c->bind_gen = h->bind_gen;
    return 0;
}

static int launch_bucket(mc_handle *h, CommState *c, const GradBucket &g, hipStream_t st) {
    if (g.p) {
        const int rc = g_rccl.all_reduce(g.p, g.p, g.n, NCCL_FLOAT32, NCCL_AVG, c->comm, st);
        if (rc) return fail(h, "ncclAllReduce(%zu floats): %s", g.n, rccl_err(rc));
        ++c->launches;
    } else if (!g.parts.empty()) {
        int rc = g_rccl.group_start();
        for (auto &p : g.parts) {
            if (rc) break;
            rc = g_rccl.all_reduce(p.first, p.first, p.second, NCCL_FLOAT32, NCCL_AVG, c->comm, st);
            ++c->launches;
        }
        const int rc2 = g_rccl.group_end();
        if (rc || rc2) return fail(h, "ncclAllReduce group: %s", rccl_err(rc ? rc : rc2));
    }
    return 0;
}

// ---- used by mc_backward (mc_train_plan.hip) -------------------------------------------------------------------------
bool mc_comm_overlap_active(mc_handle *h) { return h->comm && h->comm->comm && h->comm->overlap; }

int mc_comm_prepare(mc_handle *h) { return build_buckets(h, h->comm); }

// bucket b is complete once everything enqueued so far on `main` (and `side`, if non-null) has run: exchange it on the
// communicator's stream
int mc_comm_fire_bucket(mc_handle *h, int b, hipStream_t main, hipStream_t side) {
    CommState *c = h->comm;
    HIPCHK(h, hipEventRecord(c->ready, main));
    HIPCHK(h, hipStreamWaitEvent(c->stream, c->ready, 0));
    if (side) {
        HIPCHK(h, hipEventRecord(c->ready2, side));
        HIPCHK(h, hipStreamWaitEvent(c->stream, c->ready2, 0));
    }
    return launch_bucket(h, c, c->buckets[b], c->stream);
}

int mc_comm_join(mc_handle *h, hipStream_t main) {
    CommState *c = h->comm;
    HIPCHK(h, hipEventRecord(c->t_compute, main));       // everything the backward itself enqueued ends here ...
    HIPCHK(h, hipEventRecord(c->done, c->stream));
    HIPCHK(h, hipStreamWaitEvent(main, c->done, 0));
    HIPCHK(h, hipEventRecord(c->t_joined, main));        // ... and here the exchange has landed: the gap is what was exposed
    c->timed = true;
    return 0;
}

extern "C" {

int mc_comm_unique_id(mc_handle *h, void *id128) {
    if (!h || !id128) return fail(h, "mc_comm_unique_id: null argument");
    if (load_rccl(h)) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    UniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc) return fail(h, "ncclGetUniqueId: %s", rccl_err(rc));
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

int mc_comm_init(mc_handle *h, int rank, int world, const void *id128) {
    if (!h || !id128) return fail(h, "mc_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(h, "mc_comm_init: rank %d of %d", rank, world);
    if (load_rccl(h)) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->comm) { h->comm_free(h->comm); h->comm = nullptr; }
    std::unique_ptr<CommState, void (*)(CommState *)> c(new CommState(), comm_free);
    c->rank = rank; c->world = world;
    UniqueId id;
    std::memcpy(&id, id128, sizeof id);
    // ncclCommInitRank blocks until ALL ranks have joined: a rank that died on its way here (or never got the id) would
    // leave the others waiting forever, with nothing on the screen.  The call therefore runs on a helper thread under a
    // watchdog (MONOCON_HIP_COMM_TIMEOUT_S; default 60 s + 5 s per rank: a cold multi-process start-up takes longer the
    // more ranks there are): on expiry this rank reports who it is and fails -- the binding then raises / falls back on every
    // rank together.  The helper stays blocked inside RCCL and holds no reference to the handle; should the late rank still
    // arrive, the helper finds the call ABANDONED and destroys the communicator it was handed (it used to leak
    // on this rank while the peers held a live one).
    struct Pending { std::mutex m; std::condition_variable cv; bool done = false, abandoned = false; int rc = 0; Comm comm = nullptr; };
    auto pend = std::make_shared<Pending>();
    const int device = h->device;
    const CommInitRankFn init_fn = g_rccl.comm_init_rank;
    const auto destroy_fn = g_rccl.comm_destroy;
    std::thread([pend, device, init_fn, destroy_fn, world, id, rank] {
        (void)hipSetDevice(device);
        Comm cm = nullptr;
        const int r = init_fn(&cm, world, id, rank);
        bool orphan = false;
        {
            std::lock_guard<std::mutex> lk(pend->m);
            pend->rc = r; pend->comm = cm; pend->done = true;
            orphan = pend->abandoned;
            pend->cv.notify_all();
        }
        if (orphan && r == 0 && cm && destroy_fn) (void)destroy_fn(cm);
    }).detach();
    double timeout_s = 60.0 + 5.0 * world;
    if (const char *e = std::getenv("MONOCON_HIP_COMM_TIMEOUT_S")) timeout_s = std::atof(e) > 0 ? std::atof(e) : timeout_s;
    {
        std::unique_lock<std::mutex> lk(pend->m);
        if (!pend->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return pend->done; })) {
            pend->abandoned = true;
            return fail(h, "mc_comm_init: rank %d of %d still waits in ncclCommInitRank after %.0f s -- not every rank reached it "
                           "(device %d; MONOCON_HIP_COMM_TIMEOUT_S changes the limit)", rank, world, timeout_s, device);
        }
        c->comm = pend->comm;
        if (pend->rc) return fail(h, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_err(pend->rc));
    }
    HIPCHK(h, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&c->ready2, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    HIPCHK(h, hipEventCreate(&c->t_compute));
    HIPCHK(h, hipEventCreate(&c->t_joined));
    if (const char *e = std::getenv("MONOCON_HIP_COMM_OVERLAP")) c->overlap = std::atoi(e) != 0;
    h->comm = c.release();
    h->comm_free = comm_free;
    return 0;
}

int mc_comm_destroy(mc_handle *h) {
    if (!h) return -1;
    if (h->comm) {
        HIPCHK(h, hipSetDevice(h->device));
        (void)hipDeviceSynchronize();
        h->comm_free(h->comm);
        h->comm = nullptr;
    }
    return 0;
}

int mc_comm_set_overlap(mc_handle *h, int on) {
    if (!h || !h->comm) return fail(h, "mc_comm_set_overlap: no communicator (mc_comm_init first)");
    h->comm->overlap = on != 0;
    return 0;
}

int mc_comm_info(mc_handle *h, int *rank, int *world, int *overlap, int *n_collectives, unsigned long long *launches,
                 char *lib_path, int lib_path_len) {
    if (!h) return -1;
    CommState *c = h->comm;
    if (rank) *rank = c ? c->rank : 0;
    if (world) *world = c ? c->world : 0;
    if (overlap) *overlap = c ? (int)c->overlap : 0;
    if (launches) *launches = c ? c->launches : 0;
    if (n_collectives) {
        *n_collectives = 0;
        if (c && !build_buckets(h, c))
            for (auto &g : c->buckets) *n_collectives += g.p ? 1 : (int)g.parts.size();
    }
    if (lib_path && lib_path_len > 0) {
        std::strncpy(lib_path, g_rccl.path.c_str(), (size_t)lib_path_len - 1);
        lib_path[lib_path_len - 1] = 0;
    }
    return 0;
}

// exposed part of the last overlapped exchange in ms (time the main stream waited for the communicator's stream at the
// end of mc_backward); blocks until that point of the stream has been reached.  -1: nothing measured yet.
int mc_comm_exposed_ms(mc_handle *h, float *ms) {
    if (!h || !ms) return -1;
    CommState *c = h->comm;
    *ms = -1.f;
    if (!c || !c->timed) return 0;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(c->t_joined));
    HIPCHK(h, hipEventElapsedTime(ms, c->t_compute, c->t_joined));
    return 0;
}

// all bound "<key>#grad" tensors <- their average over the ranks, in place, enqueued on `stream`
int mc_allreduce_grads(mc_handle *h, void *stream) {
    if (!h) return -1;
    CommState *c = h->comm;
    if (!c || !c->comm) return fail(h, "mc_allreduce_grads: no communicator (mc_comm_init first)");
    HIPCHK(h, hipSetDevice(h->device));
    if (build_buckets(h, c)) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (auto &g : c->buckets)
        if (launch_bucket(h, c, g, st)) return -1;
    return 0;
}

}  // extern "C"
