#pragma once
#include "common.h"

namespace dgdm {

// One ConditionalResidualBlock1D (generator/diffusion_utils.py:75-120), device pointers.
// Conv weights are MFMA images (csrc/unet.hip conv_mfma) except the single-channel ones; Linear weights [in][out].
struct UnetRes {
    int cin, cout;
    const float *c0_w, *c0_b, *g0_w, *g0_b;      // blocks.0: Conv1d k5 (image; [tap][cout] when cin == 1), GroupNorm
    const float *c1_w, *c1_b, *g1_w, *g1_b;      // blocks.1
    const float *cond_wt, *cond_b;               // cond_encoder.1: Linear(cond_dim, 2*cout)
    const float *res_w, *res_b;                  // residual_conv 1x1 (image; [cout] when cin == 1) or null
    int c0_e, c1_e, res_e;                       // f16x3 images: the convolution's weights are stored times 2^e (unet.hip conv_mfma_f16x3)
};

struct UnetParams {
    int d0, d1, dsed, groups, cmax;
    int bf16;                                    // what the MFMA convolution images are: 0 float32 (conv_mfma), 1 bf16 (conv_mfma_bf16), 2 two f16 pieces (conv_mfma_f16x3)
    int down_e, up_e_even, up_e_odd, fin_e;      // f16x3: scale exponents of the images below
    const float *freqs;                          // SinusoidalPosEmb frequencies [dsed/2]
    const float *se1_wt, *se1_b, *se3_wt, *se3_b;
    UnetRes res[8];                              // down0.0 down0.1 down1.0 down1.1 mid0 mid1 up0.0 up0.1
    const float *down_w, *down_b;                // Downsample1d conv k3 s2 (image)
    const float *up_w_even, *up_w_odd, *up_b;    // Upsample1d ConvTranspose1d k4 s2: images of taps (1,3) and (2,0)
    const float *fin_w, *fin_b, *fin_gw, *fin_gb;// final_conv.0
    const float *out_w, *out_b;                  // final_conv.1 (d0 -> 1)
};

// p: host copy (sizes); p_dev: the same struct in device memory (what the kernel reads)
// f16x3: p_dev holds the two-piece f16 images (the kernel then needs LDS for the split slabs: unet_f16x3_fits)
int unet_launch(const UnetParams &p, const UnetParams *p_dev, bool f16x3, const float *sample, const int *timestep, float *eps, int B, int L, hipStream_t s);
bool unet_f16x3_fits(const UnetParams &p, int L);
// Batched form for large batches (f16x3 only; unet.hip "batched form"): layer-by-layer launches, several samples per workgroup, activations in a
// zero-initialised global workspace.  unet_batched_samples: samples per workgroup, 0 when the shape is not supported.
int unet_batched_samples(const UnetParams &p, int L);
size_t unet_batched_ws_floats(const UnetParams &p, int B, int L);
int unet_launch_batched(const UnetParams &q, const UnetParams *q_dev, float *ws, int Bcap, const float *sample, const int *timestep, float *eps, int B, int L, hipStream_t s);

}  // namespace dgdm
