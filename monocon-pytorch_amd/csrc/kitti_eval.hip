// KITTI AP evaluation, the native parts (last row):
//
//   * device: pairwise overlap of rotated boxes -- bird's-eye-view IoU and 3D IoU of camera-frame boxes.  Replaces the
//     reference's only GPU kernel, the numba.cuda rotate_iou_kernel_eval (engine/kitti_eval/rotate_iou.py:280-334, host
//     wrapper :337-378) and, for 3D, the numba CPU pass that follows it (engine/kitti_eval/eval.py:128-164).
//   * host: the matching / counting loops the reference JIT-compiles with numba (eval.py:90-119 image_box_overlap,
//     :167-285 compute_statistics_jit, :297-344 fused_compute_statistics).  Plain C++ behind the C-ABI: no device, no
//     handle, callable on a CPU-only box.
//
// Device mapping (not the reference's): a workgroup owns a 64 x 64 tile of the (boxes x queries) matrix.  The corners of
// its 128 boxes are computed ONCE into LDS (the reference recomputes sin/cos and the corners for every pair), then each
// of the 256 lanes walks 16 pairs of one box row.  The intersection polygon of a pair (up to 16 candidate vertices) lives in an
// LDS stripe private to the lane, laid out slot-major (slot * 256 + lane) so that the data-dependent indexing of the
// vertex sort is a conflict-free ds_read / ds_write instead of scratch memory.
//
// Arithmetic follows the reference kernel's float32 formulation step by step (corner rotation, the >= tests of
// point-in-quadrilateral, the determinant form of the segment intersection, centroid-angle insertion sort, triangle
// fan area accumulated in double); the fused-multiply-add contraction of either compiler is not reproducible, so parity
// is to float32 round-off (tests/test_kitti_eval.py), not bit-exact.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "mc_internal.h"

namespace mc {

constexpr int RT = 64;          // tile edge (boxes and queries per workgroup)
constexpr int RTHREADS = 256;
constexpr int MAXV = 16;        // vertex slots: at most 8 corners inside the other box + 8 edge crossings.  (The reference's
                                // buffer holds 8 points; nearly coincident boxes produce more and overrun it there --
                                // undefined in the reference, computed properly here.)

struct RBox { float cx, cy, dx, dy, ang; };

__device__ __forceinline__ void rbox_corners(const RBox b, float *c /* [8] in LDS or registers */, int stride) {
    const float a_cos = cosf(b.ang), a_sin = sinf(b.ang);
    const float hx = b.dx / 2, hy = b.dy / 2;
    const float px[4] = {-hx, -hx, hx, hx};
    const float py[4] = {-hy, hy, hy, -hy};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[(2 * i) * stride] = a_cos * px[i] + a_sin * py[i] + b.cx;
        c[(2 * i + 1) * stride] = -a_sin * px[i] + a_cos * py[i] + b.cy;
    }
}

__device__ __forceinline__ bool point_in_quad(float x, float y, const float (&q)[8]) {
    const float ab0 = q[2] - q[0], ab1 = q[3] - q[1];
    const float ad0 = q[6] - q[0], ad1 = q[7] - q[1];
    const float ap0 = x - q[0], ap1 = y - q[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0.f && adad >= adap && adap >= 0.f;
}

// edge i of quadrilateral p against edge j of quadrilateral q
__device__ __forceinline__ bool edge_cross(const float (&p)[8], const float (&q)[8], int i, int j, float &ox, float &oy) {
    const float A0 = p[2 * i], A1 = p[2 * i + 1], B0 = p[2 * ((i + 1) & 3)], B1 = p[2 * ((i + 1) & 3) + 1];
    const float C0 = q[2 * j], C1 = q[2 * j + 1], D0 = q[2 * ((j + 1) & 3)], D1 = q[2 * ((j + 1) & 3) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const bool acd = DA1 * CA0 > CA1 * DA0;
    const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd == bcd) return false;
    const bool abc = CA1 * BA0 > BA1 * CA0;
    const bool abd = DA1 * BA0 > BA1 * DA0;
    if (abc == abd) return false;
    const float DC0 = D0 - C0, DC1 = D1 - C1;
    const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
    const float DH = BA1 * DC0 - BA0 * DC1;
    ox = (ABBA * DC0 - BA0 * CDDC) / DH;
    oy = (ABBA * DC1 - BA1 * CDDC) / DH;
    return true;
}

// area of the intersection of quadrilaterals p (the query) and q (the box); vx / vy / vs: this lane's LDS stripes
__device__ double quad_intersection_area(const float (&p)[8], const float (&q)[8], float *vx, float *vy, float *vs) {
    int n = 0;
    auto push = [&](float x, float y) {
        if (n < MAXV) {
            vx[n * RTHREADS] = x;
            vy[n * RTHREADS] = y;
            ++n;
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (point_in_quad(p[2 * i], p[2 * i + 1], q)) push(p[2 * i], p[2 * i + 1]);
        if (point_in_quad(q[2 * i], q[2 * i + 1], p)) push(q[2 * i], q[2 * i + 1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x, y;
            if (edge_cross(p, q, i, j, x, y)) push(x, y);
        }
    if (n == 0) return 0.0;
    // order the vertices by the angle around their centroid (key: cosine, mirrored for the lower half plane)
    float c0 = 0.f, c1 = 0.f;
    for (int i = 0; i < n; ++i) { c0 += vx[i * RTHREADS]; c1 += vy[i * RTHREADS]; }
    c0 = (float)((double)c0 / (double)n);
    c1 = (float)((double)c1 / (double)n);
    for (int i = 0; i < n; ++i) {
        float v0 = vx[i * RTHREADS] - c0, v1 = vy[i * RTHREADS] - c1;
        const float d = sqrtf(v0 * v0 + v1 * v1);
        v0 = v0 / d;
        v1 = v1 / d;
        if (v1 < 0.f) v0 = -2.f - v0;
        vs[i * RTHREADS] = v0;
    }
    for (int i = 1; i < n; ++i) {
        if (vs[(i - 1) * RTHREADS] > vs[i * RTHREADS]) {
            const float key = vs[i * RTHREADS], tx = vx[i * RTHREADS], ty = vy[i * RTHREADS];
            int j = i;
            while (j > 0 && vs[(j - 1) * RTHREADS] > key) {
                vs[j * RTHREADS] = vs[(j - 1) * RTHREADS];
                vx[j * RTHREADS] = vx[(j - 1) * RTHREADS];
                vy[j * RTHREADS] = vy[(j - 1) * RTHREADS];
                --j;
            }
            vs[j * RTHREADS] = key;
            vx[j * RTHREADS] = tx;
            vy[j * RTHREADS] = ty;
        }
    }
    // triangle fan from vertex 0; each triangle's area in float32, the sum in double (the reference's typing)
    double area = 0.0;
    const float a0 = vx[0], a1 = vy[0];
    for (int i = 0; i < n - 2; ++i) {
        const float b0 = vx[(i + 1) * RTHREADS], b1 = vy[(i + 1) * RTHREADS];
        const float e0 = vx[(i + 2) * RTHREADS], e1 = vy[(i + 2) * RTHREADS];
        const float cr = (a0 - e0) * (b1 - e1) - (a1 - e1) * (b0 - e0);
        area += fabs((double)cr / 2.0);
    }
    return area;
}

// MODE 0: boxes (N,5) float32 [cx, cy, dx, dy, angle] -> out float32 (N,K), criterion -1 / 0 / 1 / other as in
//         rotate_iou.py:252-277 (rbox1 = the QUERY box).
// MODE 1: boxes (N,7) float64 camera-frame [x, y, z, l, h, w, ry] -> out float64 (N,K) 3D overlap, criterion as in
//         eval.py:128-157 (area1 = volume of the box, area2 = of the query box).
template <int MODE>
__global__ __launch_bounds__(RTHREADS) void rotate_overlap_kernel(const void *boxes_, const void *qboxes_, long long N,
                                                                  long long K, int criterion, void *out_) {
    __shared__ float corners[2][8][RT];          // [boxes | queries][coordinate][index in tile]
    __shared__ float vbuf[3][MAXV][RTHREADS];    // vertex x, vertex y, sort key -- one stripe per lane
    const int tid = threadIdx.x;
    const long long n0 = (long long)blockIdx.x * RT, k0 = (long long)blockIdx.y * RT;
    if (tid < 2 * RT) {
        const int which = tid / RT, i = tid % RT;
        const long long idx = (which ? k0 : n0) + i;
        if (idx < (which ? K : N)) {
            RBox b;
            if (MODE == 0) {
                const float *s = static_cast<const float *>(which ? qboxes_ : boxes_) + idx * 5;
                b = {s[0], s[1], s[2], s[3], s[4]};
            } else {        // bird's-eye view of a camera-frame box: (x, z, l, w, ry), cast to float32 as the reference does
                const double *s = static_cast<const double *>(which ? qboxes_ : boxes_) + idx * 7;
                b = {(float)s[0], (float)s[2], (float)s[3], (float)s[5], (float)s[6]};
            }
            rbox_corners(b, &corners[which][0][i], RT);
        }
    }
    __syncthreads();
    const int row = tid & (RT - 1), cseg = tid >> 6;            // box row of the tile, 16-query segment
    const long long n = n0 + row;
    if (n >= N) return;
    float q[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) q[c] = corners[0][c][row];
    float *vx = &vbuf[0][0][tid], *vy = &vbuf[1][0][tid], *vs = &vbuf[2][0][tid];
    for (int kk = 0; kk < RT / 4; ++kk) {
        const int col = cseg * (RT / 4) + kk;
        const long long k = k0 + col;
        if (k >= K) break;
        float p[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) p[c] = corners[1][c][col];
        const double ai = quad_intersection_area(p, q, vx, vy, vs);
        if (MODE == 0) {
            const float *bb = static_cast<const float *>(boxes_) + n * 5, *qq = static_cast<const float *>(qboxes_) + k * 5;
            const float area1 = qq[2] * qq[3], area2 = bb[2] * bb[3];
            // (the intersection area is a double in the reference's expression, the box areas float32)
            double r;
            if (criterion == -1) r = ai / ((double)(area1 + area2) - ai);
            else if (criterion == 0) r = ai / (double)area1;
            else if (criterion == 1) r = ai / (double)area2;
            else r = ai;
            static_cast<float *>(out_)[n * K + k] = (float)r;
        } else {
            const double *bb = static_cast<const double *>(boxes_) + n * 7, *qq = static_cast<const double *>(qboxes_) + k * 7;
            const float inter = (float)ai;       // the reference passes the bare area through a float32 array
            double r = 0.0;
            if (inter > 0.f) {
                // camera frame: y points down, a box spans [y - h, y]
                const double iw = fmin(bb[1], qq[1]) - fmax(bb[1] - bb[4], qq[1] - qq[4]);
                if (iw > 0.0) {
                    const double area1 = bb[3] * bb[4] * bb[5], area2 = qq[3] * qq[4] * qq[5];
                    const double inc = iw * (double)inter;
                    double ua;
                    if (criterion == -1) ua = area1 + area2 - inc;
                    else if (criterion == 0) ua = area1;
                    else if (criterion == 1) ua = area2;
                    else ua = inc;
                    r = inc / ua;
                }
            }
            static_cast<double *>(out_)[n * K + k] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host statistics
static void image_overlap(const double *boxes, long long N, const double *qboxes, long long K, int criterion, double *out) {
    for (long long i = 0; i < N * K; ++i) out[i] = 0.0;
    for (long long k = 0; k < K; ++k) {
        const double *qb = qboxes + k * 4;
        const double qarea = (qb[2] - qb[0]) * (qb[3] - qb[1]);
        for (long long n = 0; n < N; ++n) {
            const double *b = boxes + n * 4;
            const double iw = std::min(b[2], qb[2]) - std::max(b[0], qb[0]);
            if (iw <= 0) continue;
            const double ih = std::min(b[3], qb[3]) - std::max(b[1], qb[1]);
            if (ih <= 0) continue;
            double ua;
            if (criterion == -1) ua = (b[2] - b[0]) * (b[3] - b[1]) + qarea - iw * ih;
            else if (criterion == 0) ua = (b[2] - b[0]) * (b[3] - b[1]);
            else if (criterion == 1) ua = qarea;
            else ua = 1.0;
            out[n * K + k] = iw * ih / ua;
        }
    }
}

struct StatOut { long long tp, fp, fn; double similarity; };

// one image, one class / difficulty (already encoded in the ignore flags), one overlap threshold.
// overlaps: (det, gt) block with leading dimension ld.  gt_datas (gt,5) = bbox + alpha, dt_datas (det,6) = bbox + alpha + score.
static StatOut image_statistics(const double *overlaps, long long ld, const double *gt_datas, long long gt_size,
                                const double *dt_datas, long long det_size, const long long *ignored_gt,
                                const long long *ignored_det, const double *dc_boxes, long long n_dc, int metric,
                                double min_overlap, double thresh, bool compute_fp, bool compute_aos,
                                std::vector<double> *tp_scores) {
    constexpr double NO_DETECTION = -10000000.0;
    std::vector<char> assigned(det_size, 0), below(det_size, 0);
    if (compute_fp)
        for (long long j = 0; j < det_size; ++j) below[j] = dt_datas[j * 6 + 5] < thresh;
    StatOut o{0, 0, 0, 0.0};
    std::vector<double> delta;
    for (long long i = 0; i < gt_size; ++i) {
        if (ignored_gt[i] == -1) continue;
        long long det_idx = -1;
        double valid_detection = NO_DETECTION, max_overlap = 0.0;
        bool assigned_ignored_det = false;
        for (long long j = 0; j < det_size; ++j) {
            if (ignored_det[j] == -1 || assigned[j] || below[j]) continue;
            const double overlap = overlaps[j * ld + i], score = dt_datas[j * 6 + 5];
            if (!compute_fp && overlap > min_overlap && score > valid_detection) {
                det_idx = j;
                valid_detection = score;
            } else if (compute_fp && overlap > min_overlap && (overlap > max_overlap || assigned_ignored_det) &&
                       ignored_det[j] == 0) {
                max_overlap = overlap;
                det_idx = j;
                valid_detection = 1;
                assigned_ignored_det = false;
            } else if (compute_fp && overlap > min_overlap && valid_detection == NO_DETECTION && ignored_det[j] == 1) {
                det_idx = j;
                valid_detection = 1;
                assigned_ignored_det = true;
            }
        }
        if (valid_detection == NO_DETECTION && ignored_gt[i] == 0) {
            ++o.fn;
        } else if (valid_detection != NO_DETECTION && (ignored_gt[i] == 1 || ignored_det[det_idx] == 1)) {
            assigned[det_idx] = 1;
        } else if (valid_detection != NO_DETECTION) {
            ++o.tp;
            if (tp_scores) tp_scores->push_back(dt_datas[det_idx * 6 + 5]);
            if (compute_aos) delta.push_back(gt_datas[i * 5 + 4] - dt_datas[det_idx * 6 + 4]);
            assigned[det_idx] = 1;
        }
    }
    if (compute_fp) {
        for (long long j = 0; j < det_size; ++j)
            if (!(assigned[j] || ignored_det[j] == -1 || ignored_det[j] == 1 || below[j])) ++o.fp;
        long long nstuff = 0;
        if (metric == 0 && n_dc > 0 && det_size > 0) {
            std::vector<double> dtb(det_size * 4), ov(det_size * n_dc);
            for (long long j = 0; j < det_size; ++j)
                for (int c = 0; c < 4; ++c) dtb[j * 4 + c] = dt_datas[j * 6 + c];
            image_overlap(dtb.data(), det_size, dc_boxes, n_dc, 0, ov.data());
            for (long long i = 0; i < n_dc; ++i)
                for (long long j = 0; j < det_size; ++j) {
                    if (assigned[j] || ignored_det[j] == -1 || ignored_det[j] == 1 || below[j]) continue;
                    if (ov[j * n_dc + i] > min_overlap) {
                        assigned[j] = 1;
                        ++nstuff;
                    }
                }
        }
        o.fp -= nstuff;
        if (compute_aos) {
            if (o.tp > 0 || o.fp > 0) {
                double s = 0.0;
                for (double d : delta) s += (1.0 + std::cos(d)) / 2.0;
                o.similarity = s;
            } else {
                o.similarity = -1.0;
            }
        }
    }
    return o;
}

}  // namespace mc

using namespace mc;

extern "C" {

int mc_rotate_iou_eval(mc_handle *h, const float *boxes, const float *query_boxes, long long N, long long K, int criterion,
                       float *iou, void *stream) {
    if (!h) return -1;
    if (N < 0 || K < 0) return fail(h, "mc_rotate_iou_eval: negative box count");
    if (N == 0 || K == 0) return 0;
    if (!boxes || !query_boxes || !iou) return fail(h, "mc_rotate_iou_eval: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    const dim3 grid((unsigned)((N + RT - 1) / RT), (unsigned)((K + RT - 1) / RT));
    hipLaunchKernelGGL((rotate_overlap_kernel<0>), grid, dim3(RTHREADS), 0, static_cast<hipStream_t>(stream), boxes, query_boxes,
                       N, K, criterion, iou);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int mc_box3d_overlap(mc_handle *h, const double *boxes, const double *query_boxes, long long N, long long K, int criterion,
                     double *overlap, void *stream) {
    if (!h) return -1;
    if (N < 0 || K < 0) return fail(h, "mc_box3d_overlap: negative box count");
    if (N == 0 || K == 0) return 0;
    if (!boxes || !query_boxes || !overlap) return fail(h, "mc_box3d_overlap: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    const dim3 grid((unsigned)((N + RT - 1) / RT), (unsigned)((K + RT - 1) / RT));
    hipLaunchKernelGGL((rotate_overlap_kernel<1>), grid, dim3(RTHREADS), 0, static_cast<hipStream_t>(stream), boxes, query_boxes,
                       N, K, criterion, overlap);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int mc_kitti_image_overlap(const double *boxes, long long N, const double *query_boxes, long long K, int criterion,
                           double *overlap) {
    if (N < 0 || K < 0 || (N > 0 && !boxes) || (K > 0 && !query_boxes) || (N * K > 0 && !overlap)) return -1;
    image_overlap(boxes, N, query_boxes, K, criterion, overlap);
    return 0;
}

// One "part" (a run of consecutive frames whose boxes were concatenated): for every frame the block
// overlaps[dt0 : dt0 + dt_nums[f], gt0 : gt0 + gt_nums[f]] of the part's (sum dt, sum gt) matrix.
//   mode 0: scores of the true positives at threshold 0 without false-positive accounting (the first pass of eval_class,
//           eval.py:490-505) appended to scores_out (capacity = sum gt) -> *n_scores; pr (optional, [4]) += that pass's own
//           (tp, 0, fn, 0)
//   mode 1: tp / fp / fn / similarity accumulated into pr[n_thresholds][4] for every score threshold (eval.py:297-344)
int mc_kitti_statistics_part(int mode, const double *overlaps, long long n_frames, const long long *gt_nums,
                             const long long *dt_nums, const long long *dc_nums, const double *gt_datas,
                             const double *dt_datas, const double *dontcares, const long long *ignored_gts,
                             const long long *ignored_dets, int metric, double min_overlap, const double *thresholds,
                             long long n_thresholds, int compute_aos, double *pr, double *scores_out, long long *n_scores) {
    if (n_frames < 0 || (mode != 0 && mode != 1)) return -1;
    if (n_frames > 0 && (!gt_nums || !dt_nums || !dc_nums)) return -1;
    if (mode == 1 && n_thresholds > 0 && (!thresholds || !pr)) return -1;
    if (mode == 0 && !n_scores) return -1;
    long long tot_gt = 0;
    for (long long f = 0; f < n_frames; ++f) tot_gt += gt_nums[f];
    std::vector<double> scores;
    long long g0 = 0, d0 = 0, c0 = 0;
    for (long long f = 0; f < n_frames; ++f) {
        const long long ng = gt_nums[f], nd = dt_nums[f], nc = dc_nums[f];
        const double *ov = overlaps ? overlaps + d0 * tot_gt + g0 : nullptr;
        const double *gd = gt_datas ? gt_datas + g0 * 5 : nullptr, *dd = dt_datas ? dt_datas + d0 * 6 : nullptr;
        const double *dc = dontcares ? dontcares + c0 * 4 : nullptr;
        const long long *ig = ignored_gts ? ignored_gts + g0 : nullptr, *id = ignored_dets ? ignored_dets + d0 : nullptr;
        if (mode == 0) {
            const StatOut o = image_statistics(ov, tot_gt, gd, ng, dd, nd, ig, id, dc, nc, metric, min_overlap, 0.0, false, false, &scores);
            if (pr) {      // optional: the pass's own (tp, fp = 0, fn) -- a valid GT matched to an ignored detection is NEITHER
                pr[0] += (double)o.tp; pr[1] += (double)o.fp; pr[2] += (double)o.fn;
                if (o.similarity != -1.0) pr[3] += o.similarity;
            }
        } else {
            for (long long t = 0; t < n_thresholds; ++t) {
                const StatOut o = image_statistics(ov, tot_gt, gd, ng, dd, nd, ig, id, dc, nc, metric, min_overlap,
                                                   thresholds[t], true, compute_aos != 0, nullptr);
                pr[t * 4 + 0] += (double)o.tp;
                pr[t * 4 + 1] += (double)o.fp;
                pr[t * 4 + 2] += (double)o.fn;
                if (o.similarity != -1.0) pr[t * 4 + 3] += o.similarity;
            }
        }
        g0 += ng; d0 += nd; c0 += nc;
    }
    if (mode == 0) {
        if ((long long)scores.size() > 0 && !scores_out) return -1;
        for (size_t i = 0; i < scores.size(); ++i) scores_out[i] = scores[i];
        *n_scores = (long long)scores.size();
    }
    return 0;
}

}  // extern "C"
