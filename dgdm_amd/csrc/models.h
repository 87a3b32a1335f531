#pragma once
#include "common.h"
#include "blob.h"
#include "unet.h"
#include "trunk.h"
#include "pointnet.h"
#include "smallnet.h"
#include <memory>

struct DgdmUnet1d {
    dgdm::Blob blob;
    dgdm::UnetParams p;
    dgdm::DevBuf p_dev;     // copy of `p` in device memory (kernel argument)
    dgdm::DevBuf w16;       // bf16 images of the MFMA convolutions
    dgdm::DevBuf p16_dev;   // `p` with those images (bf16 = 1)
    dgdm::DevBuf wf16;      // two-piece f16 images of the MFMA convolutions (unet.hip conv_mfma_f16x3)
    dgdm::DevBuf pf16_dev;  // `p` with those images (bf16 = 2) and their scale exponents
    int mode = 2;           // dgdm_unet1d_set_contraction_dtype: 0 float32 MFMA chain, 1 bf16, 2 f16x3 (the default float32 form)
    dgdm::UnetParams pf16;  // host copy of what pf16_dev holds (the batched form's launches take their image pointers from it)
    dgdm::DevBuf bws;       // batched form: activation workspace (halo rows zero), sized for (bws_B, bws_L)
    int bws_B = 0, bws_L = 0;
    // batches of at least this many samples run the batched form (DGDM_UNET_BATCHED_MIN; 0 = never).  Four samples per workgroup: below
    // ~3/4 of the chip's 256 CUs' worth of workgroups the per-sample kernel (one workgroup per sample) is faster - measured at 256
    // samples, L = 14 (BASELINE configs[1]): 1.5 ms per step per-sample, 3.1 ms batched (64 workgroups x 20 launches)
    int batched_min = 768;
};

namespace dgdm {
struct DynOff {   // offsets (floats) into the blob
    size_t g0_wt, g0_b, g0_w, g2_wt, g2_b, g2_w;
    size_t w1c_wt, w1c_w, w1p_wt, w1t_wt, b1, w1o_wt;
    size_t b2;
    size_t bf[8];
    size_t wfwd, wbwd;      // continuous forward / backward weight streams (trunk.h)
    size_t fwd_floats, bwd_floats;
    size_t wout, bout;
    size_t te0_wt, te0_b, te2_wt, te2_b, oe0_wt, oe0_b, oe2_wt, oe2_b, tfreq;
    size_t sa1_w0t, sa1_b0, sa1_w1, sa1_b1, sa2_wf_t, sa2_b0, sa2_vx, sa2_w1_img, sa2_b1, sa3_w_img, sa3_wx, sa3_b;
};
struct DynOff64 {   // offsets (doubles) into blob64: the unrounded folds of the stages that run in float64 (smallnet.h, pointnet64)
    size_t g0_wt, g0_b, g0_w, g2_wt, g2_b, g2_w;
    size_t w1c_wt, w1c_w, w1p_wt, w1t_wt, b1, w1o_wt;
    size_t te0_wt, te0_b, te2_wt, te2_b, oe0_wt, oe0_b, oe2_wt, oe2_b;
    size_t sa1_w0t, sa1_b0, sa1_w1, sa1_b1, sa2_wf_t, sa2_b0, sa2_vx, sa2_w1_img, sa2_b1, sa3_w_img, sa3_wx, sa3_b;
};
}  // namespace dgdm

struct DgdmDynamics {
    int kind = 0, L = 0, object_ch = 0, W1 = 256, n_mid = 7, thalf = 64;
    dgdm::Blob blob;
    dgdm::DynOff off{};
    dgdm::Blob64 blob64;
    dgdm::DynOff64 off64{};
    dgdm::DevBuf ws;        // grow-only workspace of the plain forward entry points
    dgdm::DevBuf ws2;
    dgdm::DevBuf w16;       // bf16 weight streams of the trunk (trunk_bf16.hip): forward then backward
    size_t fwd16_bytes = 0, bwd16_bytes = 0, sa3_16_offset = 0;     // sa3 bf16 image (z16_kernel) follows the two trunk streams
    dgdm::DevBuf wf16;      // two-way f16 split streams of the trunk (trunk_f16l.hip): forward then backward
    size_t fwdh_bytes = 0, bwdh_bytes = 0;
    dgdm::TrunkF16Scales f16_scales{};
    dgdm::DevBuf z1_unit;   // [W1] floats 2^e: the library carries unit j of the first trunk layer as (true value) x 2^e_j (models_api.hip TrunkEquil)

    void fill_trunk(dgdm::TrunkParams *p) const;
    void fill_trunk_f16(dgdm::TrunkParams *p, dgdm::TrunkF16Scales *sc) const;
    void fill_trunk_bf16(dgdm::TrunkParams *p) const;
    dgdm::PnWeights pn() const;
    int gripper_forward(const float *x, int ldx, float *V, float *genc, int rows, hipStream_t s) const;
    int time_part(const float *t_dev, float t_scalar, float *tmp, float *out, int rows, hipStream_t s) const;
    int object_part_2d(const float *obj, float *tmp, float *out, int n, bool accumulate, hipStream_t s) const;
    // the guided path's versions in float64 (smallnet.h linear64): hidden layer V64 and encoding genc64 [rows][256] doubles
    int gripper_forward64(const float *x, int ldx, double *V64, double *genc64, int rows, hipStream_t s) const;
    // out64[1][W1] = W1'[:, time part] * time_feature(t) + b1'   (tmp: 768 floats, tmp64: 512 doubles)
    int time_part64(float t_scalar, float *tmp, double *tmp64, double *out64, hipStream_t s) const;
    // out64[n][W1] = W1'[:, object part] * object_encoder(obj)   (tmp64: n * 512 doubles)
    int object_part_2d64(const float *obj, double *tmp64, double *out64, int n, hipStream_t s) const;
    dgdm::PnWeights64 pn64() const;
};
