// Training-side launch argument blocks (targets, losses, BatchNorm train mode, backward, optimizer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
                             float *dlogit, hipStream_t st);
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
};
hipError_t launch_gathered_losses(const GatherLossArgs &a, int mode, hipStream_t st);

// ---- fused clip + AdamW (reference engine/monocon_engine.py:94-102)
struct OptTensor { float *p, *g, *m, *v; };
struct OptChunk { int tensor, begin, count; };
struct AdamHyper { float decay, beta1, one_minus_beta1, beta2, one_minus_beta2, sqrt_bc2, eps, step_size; };
hipError_t launch_clip_adamw(const OptTensor *tab, const OptChunk *chunks, int nchunks, float *partial, float *normcoef,
                             float max_norm, const AdamHyper &hp, hipStream_t st);
int opt_partial_floats();

}  // namespace mc
