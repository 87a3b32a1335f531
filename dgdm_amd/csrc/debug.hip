// Unit-test hook for the MFMA chain: Y[256 x 32] = W[256 x 256] * X[256 x 32] for one tile.
// X, Y are row-major [32 rows][256 features] (as every feature table in this library).
#include "common.h"
#include "mfma_chain.h"
#include "models.h"
#include "smallnet.h"

namespace dgdm {
__global__ __launch_bounds__(64, 1) void debug_chain_kernel(const float4 *__restrict__ Wimg, const float *__restrict__ bias,
                                                           const float *__restrict__ X, float *__restrict__ Y) {
    const int lane = threadIdx.x & 63, n = lane & 31, h4 = (lane >> 5) * 4;
    f32x16 in[8], out[8];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = feat4(X + n * 256, o, q, h4);
            in[o][4 * q + 0] = v.x; in[o][4 * q + 1] = v.y; in[o][4 * q + 2] = v.z; in[o][4 * q + 3] = v.w;
        }
    chain_layer<8, 8, CHAIN_BIAS>(Wimg, bias, in, out, lane);
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v;
            v.x = out[o][4 * q + 0]; v.y = out[o][4 * q + 1]; v.z = out[o][4 * q + 2]; v.w = out[o][4 * q + 3];
            *reinterpret_cast<float4 *>(Y + n * 256 + 32 * o + 8 * q + h4) = v;
        }
}
}  // namespace dgdm

// W_host [256][256] row-major, bias_host [256], X_dev/Y_dev [32][256]
extern "C" int dgdm_debug_chain_layer(const float *W_host, const float *bias_host, const float *X_dev, float *Y_dev, void *stream) {
    using namespace dgdm;
    std::vector<float> img = pack_chain(W_host, 256, 256);
    DevBuf dw, db;
    int rc;
    if ((rc = dw.upload(img.data(), img.size() * 4)) || (rc = db.upload(bias_host, 256 * 4))) return rc;
    hipLaunchKernelGGL(debug_chain_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dw.as<float4>(), db.as<float>(), X_dev, Y_dev);
    DGDM_HIP_CHECK(hipGetLastError());
    DGDM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return DGDM_OK;
}

// Index decisions of the PointNet++ pipeline for one cloud (see include/dgdm_hip.h).
extern "C" int dgdm_debug_pointnet_indices(DgdmDynamics *m, const float *xyz_dev, int N, const int32_t *perm_dev, int perm_len,
                                           int32_t *fps512_dev, int32_t *fps128_dev, int32_t *fps128_flags_dev, int32_t *ball1_dev,
                                           int32_t *ball2_dev, int32_t *ball2_count_dev, int32_t *crowded_dev, void *stream) {
    using namespace dgdm;
    DGDM_REQUIRE(m && xyz_dev && perm_dev && fps512_dev && fps128_dev && fps128_flags_dev && ball1_dev && ball2_dev && ball2_count_dev && crowded_dev,
                 DGDM_EINVAL, "dgdm_debug_pointnet_indices: null argument");
    if (m->kind != 3) { set_error("model type not supported: PointNet++ belongs to the 3-D model"); return DGDM_EMODE; }
    DGDM_REQUIRE(N >= 128 && N <= 1024 && perm_len > 0 && perm_len <= N, DGDM_EINVAL, "dgdm_debug_pointnet_indices: N %d / perm_len %d unsupported", N, perm_len);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = pn_fps_table(xyz_dev, N, N, 512, fps512_dev, nullptr, s))) return rc;
    if ((rc = pn_fps_table(xyz_dev, N, N, 128, fps128_dev, fps128_flags_dev, s))) return rc;
    return pn_debug_indices(xyz_dev, N, m->pn(), perm_dev, perm_len, ball1_dev, ball2_dev, ball2_count_dev, crowded_dev, s);
}

// ------------------------------------------------------------------------------------------------ the reference's index functions (include/dgdm_hip.h)
extern "C" int dgdm_farthest_point_sample(const float *xyz_dev, const int64_t *start_host, int B, int N, int npoint, int32_t *out_dev, void *stream) {
    using namespace dgdm;
    DGDM_REQUIRE(xyz_dev && start_host && out_dev && B >= 0 && npoint > 0, DGDM_EINVAL, "dgdm_farthest_point_sample: bad argument");
    DGDM_REQUIRE(N > 0 && N <= 1024, DGDM_EINVAL, "clouds of %d points unsupported (1..1024)", N);
    if (B == 0) return DGDM_OK;
    std::vector<int> st(B);
    for (int i = 0; i < B; ++i) {
        DGDM_REQUIRE(start_host[i] >= 0 && start_host[i] < N, DGDM_EINVAL, "FPS start %lld outside [0, %d)", (long long)start_host[i], N);
        st[i] = (int)start_host[i];
    }
    DevBuf d;
    int rc;
    if ((rc = d.upload(st.data(), sizeof(int) * B))) return rc;
    if ((rc = pn_fps_rows(xyz_dev, d.as<int>(), B, N, npoint, out_dev, (hipStream_t)stream))) return rc;
    DGDM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));           // `d` dies here
    return DGDM_OK;
}

extern "C" int dgdm_query_ball_point(float radius_squared, int nsample, const float *xyz_dev, const float *new_xyz_dev, int B, int N, int S, int32_t *out_dev,
                                     void *stream) {
    using namespace dgdm;
    DGDM_REQUIRE(xyz_dev && new_xyz_dev && out_dev && B >= 0 && N > 0 && S >= 0 && nsample > 0, DGDM_EINVAL, "dgdm_query_ball_point: bad argument");
    if (B == 0 || S == 0) return DGDM_OK;
    return pn_ball_rows(xyz_dev, new_xyz_dev, B, N, S, radius_squared, nsample, out_dev, (hipStream_t)stream);
}

extern "C" int dgdm_square_distance(const float *src_dev, const float *dst_dev, int B, int S, int N, float *out_dev, void *stream) {
    using namespace dgdm;
    DGDM_REQUIRE(src_dev && dst_dev && out_dev && B >= 0 && S >= 0 && N >= 0, DGDM_EINVAL, "dgdm_square_distance: bad argument");
    if (B == 0 || S == 0 || N == 0) return DGDM_OK;          // empty result
    return pn_sqdist_rows(src_dev, dst_dev, B, S, N, out_dev, (hipStream_t)stream);
}

extern "C" int dgdm_index_points(const float *points_dev, const int32_t *idx_dev, int B, int N, int M, int C, float *out_dev, void *stream) {
    using namespace dgdm;
    DGDM_REQUIRE(points_dev && idx_dev && out_dev && B >= 0 && N > 0 && M >= 0 && C > 0, DGDM_EINVAL, "dgdm_index_points: bad argument");
    if (B == 0 || M == 0) return DGDM_OK;                      // empty result
    return pn_index_rows(points_dev, idx_dev, B, N, M, C, out_dev, (hipStream_t)stream);
}


// ------------------------------------------------------------------------------------------------ PointNetSetAbstraction.forward on its own
// (dynamics/models/pointnet2_utils.py:184-210; include/dgdm_hip.h): the shared MLP of a set-abstraction level on grouped rows, one layer at
// a time, and the max over a group's samples.  The guided path evaluates the three levels fused as per-object tables (pointnet.hip);
// these two entry points exist so that the reference's layer class is callable for arbitrary inputs.
namespace dgdm {
__global__ void group_max_kernel(const float *__restrict__ x, int64_t G, int ns, int C, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= G * C) return;
    const int64_t g = e / C;
    const int c = (int)(e - g * C);
    const float *p = x + g * ns * C + c;
    float best = p[0];
    for (int n = 1; n < ns; ++n) best = fmaxf(best, p[(int64_t)n * C]);
    out[e] = best;
}
}  // namespace dgdm

extern "C" int dgdm_linear_act(const float *x_dev, int ldx, const float *wt_dev, const float *bias_dev, float *y_dev, int ldy, int rows, int K, int N, int act,
                               void *stream) {
    using namespace dgdm;
    DGDM_REQUIRE(x_dev && wt_dev && y_dev && rows >= 0 && K > 0 && N > 0 && ldx >= K && ldy >= N, DGDM_EINVAL, "dgdm_linear_act: bad argument");
    DGDM_REQUIRE(act == ACT_NONE || act == ACT_RELU || act == ACT_SILU, DGDM_EINVAL, "dgdm_linear_act: activation %d", act);
    return linear(x_dev, ldx, wt_dev, bias_dev, nullptr, 1, y_dev, ldy, rows, K, N, act, false, (hipStream_t)stream);
}

extern "C" int dgdm_group_max(const float *x_dev, int64_t groups, int nsample, int C, float *out_dev, void *stream) {
    using namespace dgdm;
    DGDM_REQUIRE(x_dev && out_dev && groups >= 0 && nsample > 0 && C > 0, DGDM_EINVAL, "dgdm_group_max: bad argument");
    if (groups == 0) return DGDM_OK;
    hipLaunchKernelGGL(group_max_kernel, dim3((unsigned)((groups * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, groups, nsample, C, out_dev);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
