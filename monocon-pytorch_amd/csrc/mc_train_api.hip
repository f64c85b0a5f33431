// Training-side C-ABI: target generation, losses (+ gradients wrt the prediction maps).
#include "mc_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

static int check_targets(mc_handle *h, const mc_targets *t, const char *who) {
    if (!t) return fail(h, "%s: targets is NULL", who);
    const void *p[] = {t->center_heatmap_target, t->wh_target, t->offset_target, t->dim_target, t->alpha_cls_target,
                       t->alpha_offset_target, t->depth_target, t->center2kpt_offset_target, t->kpt_heatmap_target,
                       t->kpt_heatmap_offset_target, t->indices, t->indices_kpt, t->mask_target,
                       t->mask_center2kpt_offset, t->mask_kpt_heatmap_offset};
    for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); ++i)
        if (!p[i]) return fail(h, "%s: targets field %d is NULL", who, (int)i);
    return 0;
}

extern "C" {

int mc_make_targets(mc_handle *h, const mc_labels *lab, int B, int max_objs, int pad_h, int pad_w, int fh, int fw,
                    const mc_targets *t, void *stream) {
    if (!h) return -1;
    if (!lab || !lab->gt_bboxes || !lab->gt_labels || !lab->gt_bboxes_3d || !lab->depths || !lab->gt_kpts_2d ||
        !lab->gt_kpts_valid_mask || !lab->mask)
        return fail(h, "mc_make_targets: a label pointer is NULL");
    if (check_targets(h, t, "mc_make_targets")) return -1;
    if (B < 1 || max_objs < 1 || fh < 1 || fw < 1 || pad_h < 1 || pad_w < 1)
        return fail(h, "mc_make_targets: bad shape");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t HW = (size_t)fh * fw, R = (size_t)B * max_objs;
    if (h->tgt_arena && t->center_heatmap_target == h->tgt_arena) {
        HIPCHK(h, hipMemsetAsync(h->tgt_arena, 0, h->tgt_arena_bytes, st));
    } else {
        HIPCHK(h, hipMemsetAsync(t->center_heatmap_target, 0, B * 3 * HW * 4, st));
        HIPCHK(h, hipMemsetAsync(t->kpt_heatmap_target, 0, B * 9 * HW * 4, st));
        HIPCHK(h, hipMemsetAsync(t->wh_target, 0, R * 2 * 4, st));
        HIPCHK(h, hipMemsetAsync(t->offset_target, 0, R * 2 * 4, st));
        HIPCHK(h, hipMemsetAsync(t->dim_target, 0, R * 3 * 4, st));
        HIPCHK(h, hipMemsetAsync(t->alpha_cls_target, 0, R * 4, st));
        HIPCHK(h, hipMemsetAsync(t->alpha_offset_target, 0, R * 4, st));
        HIPCHK(h, hipMemsetAsync(t->depth_target, 0, R * 4, st));
        HIPCHK(h, hipMemsetAsync(t->center2kpt_offset_target, 0, R * 18 * 4, st));
        HIPCHK(h, hipMemsetAsync(t->kpt_heatmap_offset_target, 0, R * 18 * 4, st));
        HIPCHK(h, hipMemsetAsync(t->indices, 0, R * 8, st));
        HIPCHK(h, hipMemsetAsync(t->indices_kpt, 0, R * 9 * 8, st));
        HIPCHK(h, hipMemsetAsync(t->mask_target, 0, R, st));
        HIPCHK(h, hipMemsetAsync(t->mask_center2kpt_offset, 0, R * 18 * 4, st));
        HIPCHK(h, hipMemsetAsync(t->mask_kpt_heatmap_offset, 0, R * 18 * 4, st));
    }
    mc::TargetArgs a{};
    a.gt_bboxes = lab->gt_bboxes; a.gt_labels = lab->gt_labels; a.gt_bboxes_3d = lab->gt_bboxes_3d;
    a.depths = lab->depths; a.gt_kpts_2d = lab->gt_kpts_2d; a.gt_kpts_valid = lab->gt_kpts_valid_mask; a.mask = lab->mask;
    a.B = B; a.max_objs = max_objs; a.num_kpt = 9; a.num_classes = 3; a.fh = fh; a.fw = fw;
    a.h_ratio = (float)((double)fh / (double)pad_h);
    a.w_ratio = (float)((double)fw / (double)pad_w);
    a.center_heatmap = t->center_heatmap_target; a.kpt_heatmap = t->kpt_heatmap_target;
    a.wh = t->wh_target; a.offset = t->offset_target; a.dim = t->dim_target; a.alpha_cls = t->alpha_cls_target;
    a.alpha_offset = t->alpha_offset_target; a.depth = t->depth_target; a.c2k = t->center2kpt_offset_target;
    a.kho = t->kpt_heatmap_offset_target;
    a.indices = reinterpret_cast<long long *>(t->indices);
    a.indices_kpt = reinterpret_cast<long long *>(t->indices_kpt);
    a.mask_target = t->mask_target; a.mask_c2k = t->mask_center2kpt_offset; a.mask_kho = t->mask_kpt_heatmap_offset;
    HIPCHK(h, mc::launch_make_targets(a, st));
    return 0;
}

static int ensure_loss_ws(mc_handle *h) {
    if (h->loss_ws) return 0;
    void *q = nullptr;
    // focal partials (2 x npf), 64 floats of small results (aux, scratch losses), then the gathered losses' fp64 workspace
    const size_t n = (size_t)2 * mc::focal_partial_floats() + 64 + 2 * mc::gathered_loss_ws_doubles() + 2;
    HIPCHK(h, hipMalloc(&q, n * sizeof(float)));
    h->loss_ws = static_cast<float *>(q);
    return 0;
}

static void fill_gather_args(mc::GatherLossArgs &g, const float *const preds[10], float *const dpreds[10],
                             const mc_targets *t, int B, int max_objs, int HW, float *losses, float *aux,
                             const float *gscale) {
    for (int i = 0; i < 10; ++i) { g.pred[i] = preds[i]; g.dpred[i] = dpreds ? dpreds[i] : nullptr; }
    g.indices = reinterpret_cast<const long long *>(t->indices);
    g.indices_kpt = reinterpret_cast<const long long *>(t->indices_kpt);
    g.mask_target = t->mask_target;
    g.wh = t->wh_target; g.offset = t->offset_target; g.dim = t->dim_target; g.alpha_cls = t->alpha_cls_target;
    g.alpha_offset = t->alpha_offset_target; g.depth = t->depth_target; g.c2k = t->center2kpt_offset_target;
    g.kho = t->kpt_heatmap_offset_target; g.mask_c2k = t->mask_center2kpt_offset; g.mask_kho = t->mask_kpt_heatmap_offset;
    g.B = B; g.max_objs = max_objs; g.HW = HW; g.losses = losses; g.aux = aux; g.gscale = gscale;
}
static double *gather_ws(mc_handle *h) {       // 8-byte aligned, behind the float part of the loss workspace
    uintptr_t p = reinterpret_cast<uintptr_t>(h->loss_ws + 2 * mc::focal_partial_floats() + 64);
    return reinterpret_cast<double *>((p + 7) & ~(uintptr_t)7);
}

int mc_losses(mc_handle *h, const float *const preds[MC_NUM_PREDS], const mc_targets *t, int B, int max_objs, int fh,
              int fw, float *losses, void *stream) {
    if (!h) return -1;
    if (!preds || !losses) return fail(h, "mc_losses: null argument");
    for (int i = 0; i < MC_NUM_PREDS; ++i)
        if (!preds[i]) return fail(h, "mc_losses: preds[%d] is NULL", i);
    if (check_targets(h, t, "mc_losses")) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    if (ensure_loss_ws(h)) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t HW = (size_t)fh * fw;
    const int npf = mc::focal_partial_floats();
    float *aux = h->loss_ws + 2 * npf;   // [0] npos center, [1] npos kpt, [2] n objects
    HIPCHK(h, mc::launch_focal(preds[0], t->center_heatmap_target, (size_t)B * 3 * HW, h->loss_ws, losses + 0, aux + 0, st));
    HIPCHK(h, mc::launch_focal(preds[1], t->kpt_heatmap_target, (size_t)B * 9 * HW, h->loss_ws + npf, losses + 5, aux + 1, st));
    mc::GatherLossArgs g{};
    fill_gather_args(g, preds, nullptr, t, B, max_objs, (int)HW, losses, aux + 2, nullptr);
    g.partial = gather_ws(h);
    HIPCHK(h, mc::launch_gathered_losses(g, 0, st));
    return 0;
}

static int losses_backward_impl(mc_handle *h, const float *const preds[MC_NUM_PREDS], const mc_targets *t, int B, int max_objs,
                                int fh, int fw, const float *grad_losses, float *const dpreds[MC_NUM_PREDS], void *stream,
                                int wrt_pred) {
    if (!h) return -1;
    if (!preds || !dpreds || !grad_losses) return fail(h, "mc_losses_backward: null argument");
    for (int i = 0; i < MC_NUM_PREDS; ++i)
        if (!preds[i] || !dpreds[i]) return fail(h, "mc_losses_backward: preds/dpreds[%d] is NULL", i);
    if (check_targets(h, t, "mc_losses_backward")) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    if (ensure_loss_ws(h)) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t HW = (size_t)fh * fw;
    const int npf = mc::focal_partial_floats();
    float *aux = h->loss_ws + 2 * npf;
    float *scratch_losses = aux + 8;
    static const int PC[10] = {3, 9, 2, 2, 2, 18, 3, 2, 12, 12};
    // recompute the reductions the gradients need (npos, object count, dim compensation weight)
    HIPCHK(h, mc::launch_focal(preds[0], t->center_heatmap_target, (size_t)B * 3 * HW, h->loss_ws, scratch_losses + 0, aux + 0, st));
    HIPCHK(h, mc::launch_focal(preds[1], t->kpt_heatmap_target, (size_t)B * 9 * HW, h->loss_ws + npf, scratch_losses + 5, aux + 1, st));
    if (h->dp_arena && dpreds[2] == h->dp_arena) {
        HIPCHK(h, hipMemsetAsync(h->dp_arena, 0, h->dp_arena_bytes, st));
    } else {
        for (int i = 2; i < 10; ++i) HIPCHK(h, hipMemsetAsync(dpreds[i], 0, (size_t)B * PC[i] * HW * 4, st));
    }
    HIPCHK(h, mc::launch_focal_grad(preds[0], t->center_heatmap_target, (size_t)B * 3 * HW, aux + 0, grad_losses, 0, dpreds[0], st, wrt_pred));
    HIPCHK(h, mc::launch_focal_grad(preds[1], t->kpt_heatmap_target, (size_t)B * 9 * HW, aux + 1, grad_losses, 5, dpreds[1], st, wrt_pred));
    mc::GatherLossArgs g{};
    fill_gather_args(g, preds, dpreds, t, B, max_objs, (int)HW, scratch_losses, aux + 2, grad_losses);
    g.wrt_pred = wrt_pred;
    g.partial = gather_ws(h);
    HIPCHK(h, mc::launch_gathered_losses(g, 1, st));
    return 0;
}

int mc_losses_backward(mc_handle *h, const float *const preds[MC_NUM_PREDS], const mc_targets *t, int B, int max_objs,
                       int fh, int fw, const float *grad_losses, float *const dpreds[MC_NUM_PREDS], void *stream) {
    return losses_backward_impl(h, preds, t, B, max_objs, fh, fw, grad_losses, dpreds, stream, 0);
}

int mc_losses_backward_pred(mc_handle *h, const float *const preds[MC_NUM_PREDS], const mc_targets *t, int B, int max_objs,
                            int fh, int fw, const float *grad_losses, float *const dpreds[MC_NUM_PREDS], void *stream) {
    return losses_backward_impl(h, preds, t, B, max_objs, fh, fw, grad_losses, dpreds, stream, 1);
}

int mc_op_conv_wgrad(mc_handle *h, const float *const src[], const int src_channels[], int nsrc, int B, int Hin, int Win,
                     const float *dy, int Cout, int ksize, int stride, float *dw_oihw, void *stream) {
    if (!h) return -1;
    if (!src || !src_channels || !dy || !dw_oihw || nsrc < 1 || nsrc > 4) return fail(h, "mc_op_conv_wgrad: bad argument");
    if (!((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 1 && stride == 1)))
        return fail(h, "mc_op_conv_wgrad: unsupported k=%d stride=%d", ksize, stride);
    if (Cout % 4) return fail(h, "mc_op_conv_wgrad: Cout must be a multiple of 4");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    mc::WgradArgs a{};
    int cin = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (!src[i] || src_channels[i] % 16) return fail(h, "mc_op_conv_wgrad: source %d needs C %% 16 == 0", i);
        a.src[i].p = src[i]; a.src[i].C = src_channels[i];
        cin += src_channels[i];
    }
    a.nsrc = nsrc; a.B = B; a.Hin = Hin; a.Win = Win;
    a.Hout = (Hin + 2 * (ksize / 2) - ksize) / stride + 1;
    a.Wout = (Win + 2 * (ksize / 2) - ksize) / stride + 1;
    a.Cin = cin; a.Cout = Cout; a.dy = dy; a.dy_ld = Cout;
    a.prec = h->prec;
    ScratchBuf slots;       // mode 3: max |x| of every source and of dY
    if (h->prec == 3) {
        HIPCHK(h, slots.alloc(5 * mc::AMAX_WORDS * sizeof(unsigned)));
        unsigned *sl = slots.as<unsigned>();
        HIPCHK(h, hipMemsetAsync(sl, 0, 5 * mc::AMAX_WORDS * sizeof(unsigned), st));
        for (int i = 0; i < nsrc; ++i) {
            HIPCHK(h, mc::launch_absmax(src[i], (size_t)B * Hin * Win * src_channels[i], sl + i * mc::AMAX_WORDS, st));
            a.amax_x[i] = sl + i * mc::AMAX_WORDS;
        }
        HIPCHK(h, mc::launch_absmax(dy, (size_t)B * a.Hout * a.Wout * Cout, sl + 4 * mc::AMAX_WORDS, st));
        a.amax_dy = sl + 4 * mc::AMAX_WORDS;
    }
    mc::wgrad_plan(a, ksize, stride);
    void *part = nullptr;
    HIPCHK(h, hipMalloc(&part, mc::wgrad_partial_floats(a, ksize) * sizeof(float)));
    a.partial = static_cast<float *>(part);
    hipError_t e = mc::launch_wgrad(a, ksize, stride, dw_oihw, st);
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(part);
    HIPCHK(h, e);
    HIPCHK(h, e2);
    return 0;
}

int mc_op_conv_dgrad(mc_handle *h, const float *dy, const float *weight_oihw, int B, int Hin, int Win, int CinTotal,
                     int c_off, int Cs, int Cout, int ksize, int stride, int accumulate, float *dx, void *stream) {
    if (!h) return -1;
    if (!dy || !weight_oihw || !dx) return fail(h, "mc_op_conv_dgrad: null argument");
    if (!((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 1 && stride == 1)))
        return fail(h, "mc_op_conv_dgrad: unsupported k=%d stride=%d", ksize, stride);
    if (Cout % 16 || Cs % 4 || c_off < 0 || c_off + Cs > CinTotal) return fail(h, "mc_op_conv_dgrad: bad channel arguments");
    if (stride == 2 && ((Hin | Win) & 1)) return fail(h, "mc_op_conv_dgrad: stride 2 needs even Hin, Win");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int Ho = (Hin + 2 * (ksize / 2) - ksize) / stride + 1, Wo = (Win + 2 * (ksize / 2) - ksize) / stride + 1;
    const int CsP = mc::conv_coutp(Cs), pieces = h->prec == 2 ? 3 : (h->prec == 3 ? 2 : 1);
    const int nclass = stride == 2 ? 4 : 1;
    std::vector<void *> tmp;
    hipError_t e = hipSuccess;
    ScratchBuf slots;       // mode 3: max |dy| and max |w|
    unsigned *sl = nullptr;
    if (h->prec == 3) {
        HIPCHK(h, slots.alloc((mc::AMAX_WORDS + 1) * sizeof(unsigned)));
        sl = slots.as<unsigned>();
        HIPCHK(h, hipMemsetAsync(sl, 0, (mc::AMAX_WORDS + 1) * sizeof(unsigned), st));
        HIPCHK(h, mc::launch_absmax(dy, (size_t)B * Ho * Wo * Cout, sl, st));
        HIPCHK(h, mc::launch_absmax(weight_oihw, (size_t)Cout * CinTotal * ksize * ksize, sl + mc::AMAX_WORDS, st, true));
    }
    for (int cls = 0; cls < nclass && e == hipSuccess; ++cls) {
        const int cid = stride == 2 ? cls : -1;
        const int py = cls >> 1, px = cls & 1;
        const int taps = stride == 2 ? (1 + py) * (1 + px) : ksize * ksize;
        const size_t pn = (size_t)taps * Cout * CsP;
        void *panel = nullptr, *panel16 = nullptr;
        if (hipMalloc(&panel, pn * 4) != hipSuccess) { e = hipErrorOutOfMemory; break; }
        tmp.push_back(panel);
        (void)hipMemsetAsync(panel, 0, pn * 4, st);
        e = mc::launch_pack_conv_w_dgrad(weight_oihw, Cout, CinTotal, ksize, c_off, Cs, CsP, Cout, cid, static_cast<float *>(panel), st);
        if (e != hipSuccess) break;
        if (h->prec >= 1 && Cout % 32 == 0) {
            if (hipMalloc(&panel16, pn * 2 * pieces) != hipSuccess) { e = hipErrorOutOfMemory; break; }
            tmp.push_back(panel16);
            (void)hipMemsetAsync(panel16, 0, pn * 2 * pieces, st);
            e = mc::launch_pack_conv_w_dgrad_bf16(weight_oihw, Cout, CinTotal, ksize, c_off, Cs, CsP, Cout, cid, pieces, panel16, st,
                                                  sl ? sl + mc::AMAX_WORDS : nullptr);
            if (e != hipSuccess) break;
        }
        mc::ConvArgs d{};
        d.nsrc = 1; d.src[0].p = dy; d.src[0].C = Cout;
        d.B = B; d.Hin = Ho; d.Win = Wo; d.Hout = Ho; d.Wout = Wo;
        d.Cin = Cout; d.Cout = Cs; d.CoutP = CsP; d.wpk = static_cast<float *>(panel);
        d.wpk16 = panel16; d.prec = panel16 ? h->prec : 0;
        if (sl) { d.amax_in[0] = sl; d.amax_w = sl + mc::AMAX_WORDS; }
        d.out = dx; d.out_ld = Cs;
        int kk = ksize;
        if (stride == 2) {
            d.out = dx + ((size_t)py * Win + px) * Cs;
            d.o_px = 2 * Cs; d.o_row = 2 * Win * Cs; d.o_img = Hin * Win * Cs;
            kk = (1 + py) * 10 + (1 + px);
            if (kk == 11) kk = 1;
        }
        if (accumulate) { d.res = d.out; d.res_ld = Cs; d.r_px = d.o_px; d.r_row = d.o_row; d.r_img = d.o_img; }
        d.cfg = h->force_cfg;
        e = mc::launch_conv(d, kk, 1, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);   // test entry point: the panels are temporaries
    for (void *q : tmp) (void)hipFree(q);
    HIPCHK(h, e);
    HIPCHK(h, e2);
    return 0;
}

int mc_optim_bind(mc_handle *h, int n, float *const params[], float *const grads[], float *const exp_avg[],
                  float *const exp_avg_sq[], const int64_t numel[]) {
    if (!h) return -1;
    if (n < 1 || !params || !grads || !exp_avg || !exp_avg_sq || !numel) return fail(h, "mc_optim_bind: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<mc::OptTensor> tab(n);
    std::vector<mc::OptChunk> chunks;
    const int CH = 1 << 16;
    for (int i = 0; i < n; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 1)
            return fail(h, "mc_optim_bind: tensor %d has a null pointer / empty size", i);
        tab[i] = mc::OptTensor{params[i], grads[i], exp_avg[i], exp_avg_sq[i]};
        for (int64_t b = 0; b < numel[i]; b += CH)
            chunks.push_back(mc::OptChunk{i, (int)b, (int)std::min<int64_t>(CH, numel[i] - b)});
    }
    if (h->opt_tab) (void)hipFree(h->opt_tab);
    if (h->opt_chunks) (void)hipFree(h->opt_chunks);
    h->opt_tab = nullptr; h->opt_chunks = nullptr;
    void *q = nullptr;
    HIPCHK(h, hipMalloc(&q, tab.size() * sizeof(mc::OptTensor)));
    h->opt_tab = static_cast<mc::OptTensor *>(q);
    HIPCHK(h, hipMalloc(&q, chunks.size() * sizeof(mc::OptChunk)));
    h->opt_chunks = static_cast<mc::OptChunk *>(q);
    HIPCHK(h, hipMemcpy(h->opt_tab, tab.data(), tab.size() * sizeof(mc::OptTensor), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->opt_chunks, chunks.data(), chunks.size() * sizeof(mc::OptChunk), hipMemcpyHostToDevice));
    h->opt_nchunks = (int)chunks.size();
    h->opt_ntensors = n;
    if (!h->opt_ws) {
        HIPCHK(h, hipMalloc(&q, (mc::opt_partial_floats() + 8) * sizeof(float)));
        h->opt_ws = static_cast<float *>(q);
    }
    return 0;
}

int mc_clip_adamw_step(mc_handle *h, double lr, double beta1, double beta2, double eps, double weight_decay,
                       double max_norm, int step, float *out_norm, void *stream) {
    if (!h) return -1;
    if (!h->opt_tab) return fail(h, "mc_clip_adamw_step: call mc_optim_bind first");
    if (step < 1) return fail(h, "mc_clip_adamw_step: step must be >= 1 (1-based, as torch.optim counts)");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    mc::AdamHyper hp;
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    hp.decay = (float)(1.0 - lr * weight_decay);
    hp.beta1 = (float)beta1; hp.one_minus_beta1 = (float)(1.0 - beta1);
    hp.beta2 = (float)beta2; hp.one_minus_beta2 = (float)(1.0 - beta2);
    hp.sqrt_bc2 = (float)std::sqrt(bc2);
    hp.eps = (float)eps;
    hp.step_size = (float)(lr / bc1);
    float *normcoef = h->opt_ws + mc::opt_partial_floats();
    HIPCHK(h, mc::launch_clip_adamw(h->opt_tab, h->opt_chunks, h->opt_nchunks, h->opt_ws, normcoef, (float)max_norm, hp, st));
    if (out_norm) HIPCHK(h, hipMemcpyAsync(out_norm, normcoef, sizeof(float), hipMemcpyDeviceToDevice, st));
    h->pack_clean = false;   // parameters changed: the packed panels are stale
    return 0;
}

}  // extern "C"
