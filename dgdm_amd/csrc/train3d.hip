// Trainer.step / Trainer.inference of the 3-D dynamics model (dynamics/trainer.py:53-146 with ProfileForward3DModel,
// dynamics/profile_forward_3d.py:13-86; PointNet2 dynamics/models/pointnet2.py:11-32; PointNetSetAbstraction
// dynamics/models/pointnet2_utils.py:169-210) on gfx950: PointNet++ in TRAINING mode (BatchNorm2d batch statistics over every grouped
// point of the batch, running statistics updated), trunk with BatchNorm1d batch statistics, nn.MSELoss, backward with weight
// gradients through max-pool, grouping and the FPS-indexed gathers, torch.optim.Adam (trainer.py:46).  SURVEY.md 8(f) rank 4.
//
// Unlike the guided path (per-object tables, eval-mode statistics: pointnet.hip), training evaluates the network AS WRITTEN: batch
// statistics couple every grouped point of the batch, so each set-abstraction level is materialised as rows x channels in HBM -
//   sa1  rows = B * 512 centres * 32 neighbours,  3 -> 64 -> 128     (B = 2048: 33.5 M rows)
//   sa2  rows = B * 128 centres * 64 neighbours,  131 -> 128 -> 256  (16.8 M rows)
//   sa3  rows = B * 128,                          259 -> 256
// (68 MB of activations and gradients per cloud: a sub-batch of dynamics/train_dynamics_3d.sh, --sub_bs=2048, is 139 GB - what the
// 288 GB of one MI355X are for) - and every 1x1 convolution / Linear is a rows-by-channels GEMM on the float32 MFMA (train_gemm.h:
// rowgemm for forward and input gradients, colgemm for the weight gradients with an ordered sum over row splits).  BatchNorm is two
// ordered column-sum passes (mean, then centred squares, as torch computes it) + an elementwise pass; the max over a group's samples
// fuses BatchNorm + ReLU on the way in and keeps the arg-max, the backward pass scatters through it, BatchNorm's backward is the
// affine form dY = sc (dZ - mean dZ - xhat mean(dZ xhat)), index_points' backward (l1 features gathered by sa2's ball query) is a
// per-(cloud, point) ordered sum over the ball lists.  FPS / ball query / gathers are the device functions the guided path's tables are
// built from (dgdm_farthest_point_sample ...: bit-exact index lists).  No atomics: a step is reproducible bit for bit.
#include "train_gemm.h"
#include "pointnet.h"
#include <cmath>
#include <cstring>
#include <memory>
#include <algorithm>

namespace dgdm {
namespace {

// rows of a set-abstraction level: (cloud r, centre c, sample n) -> [xyz[k] - xyz[centre] | pts[k]] padded to Cp floats, k = idx[r][c][n]
// (pointnet2_utils.py:118-146; idx == null: group_all, k = n, no centre is subtracted, :149-166)
__global__ void group_kernel(const float *__restrict__ xyz, const float *__restrict__ pts, const int *__restrict__ cidx, const int *__restrict__ idx, int N,
                             int S, int ns, int D, int Cp, int64_t rows, float *__restrict__ out) {
    const int q4 = Cp / 4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * q4) return;
    const int64_t m = e / q4;
    const int q = (int)(e - m * q4);
    const int64_t r = m / ((int64_t)S * ns);
    const int c = (int)((m / ns) % S);
    const int k = idx ? idx[m] : (int)(m % ((int64_t)S * ns));
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = 4 * q + j;
        if (ch < 3) {
            float a = xyz[(r * N + k) * 3 + ch];
            if (cidx) a -= xyz[(r * N + cidx[r * S + c]) * 3 + ch];
            v[j] = a;
        } else v[j] = ch - 3 < D ? pts[(r * N + k) * D + (ch - 3)] : 0.f;
    }
    *reinterpret_cast<float4 *>(out + m * Cp + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
}

// per-block column sums of y (mean == null) or of (y - mean)^2: the two passes of BatchNorm's batch statistics
__global__ __launch_bounds__(256) void colstat_kernel(const float *__restrict__ D, int64_t rs, int64_t M, int N, int64_t rows_per_block,
                                                      const float *__restrict__ mean, float *__restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float a = 0.f;
    if (c < N) {
        if (mean) {
            const float mu = mean[c];
            for (int64_t r = r0 + rl; r < r1; r += 4) { const float d = D[r * rs + c] - mu; a = fmaf(d, d, a); }
        } else {
            for (int64_t r = r0 + rl; r < r1; r += 4) a += D[r * rs + c];
        }
    }
    red[rl][threadIdx.x & 63] = a;
    __syncthreads();
    if (rl == 0 && c < N) part[(int64_t)blockIdx.x * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// coef rows of CW floats: 0 sc = gamma rstd, 1 sh = beta - mean sc, 2 mean, 3 rstd, 4 mean(dZ), 5 mean(dZ xhat)
constexpr int CW = 512;
__global__ void bn_mean_kernel(const float *__restrict__ sum, double M, int C, float *__restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) coef[2 * CW + c] = (float)((double)sum[c] / M);
}
// torch.nn.BatchNorm{1,2}d in training mode: biased variance for the normalisation, unbiased for the running estimate, momentum 0.1
__global__ void bn_var_kernel(const float *__restrict__ sumsq, double M, int C, const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float mom,
                              float *__restrict__ rmean, float *__restrict__ rvar, float *__restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double var = (double)sumsq[c] / M;
    const float mu = coef[2 * CW + c];
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), sc = gamma[c] * rstd;
    coef[c] = sc; coef[CW + c] = beta[c] - mu * sc; coef[3 * CW + c] = rstd;
    rmean[c] = (1.f - mom) * rmean[c] + mom * mu;
    rvar[c] = (1.f - mom) * rvar[c] + mom * (float)(var * M / (M - 1.0));
}
__global__ void bn_eval_kernel(int C, const float *__restrict__ gamma, const float *__restrict__ beta, float eps, const float *__restrict__ rmean,
                               const float *__restrict__ rvar, float *__restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.f / sqrtf(rvar[c] + eps), sc = gamma[c] * rstd;
    coef[c] = sc; coef[CW + c] = beta[c] - rmean[c] * sc; coef[2 * CW + c] = rmean[c]; coef[3 * CW + c] = rstd;
}
// a = relu(sc y + sh)   (coef == null: plain ReLU)
__global__ void bn_relu_kernel(const float *__restrict__ y, const float *__restrict__ coef, float *__restrict__ a, int64_t M, int C, int64_t ld_a) {
    const int c4 = C / 4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * c4) return;
    const int64_t m = e / c4;
    const int c = (int)(e - m * c4) * 4;
    const float4 v = *reinterpret_cast<const float4 *>(y + m * C + c);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (coef) { sc = *reinterpret_cast<const float4 *>(coef + c); sh = *reinterpret_cast<const float4 *>(coef + CW + c); }
    *reinterpret_cast<float4 *>(a + m * ld_a + c) = make_float4(fmaxf(fmaf(sc.x, v.x, sh.x), 0.f), fmaxf(fmaf(sc.y, v.y, sh.y), 0.f),
                                                                fmaxf(fmaf(sc.z, v.z, sh.z), 0.f), fmaxf(fmaf(sc.w, v.w, sh.w), 0.f));
}
// torch.max(new_points, 2)[0] over a group's samples of relu(bn(y)) (pointnet2_utils.py:204-206): value and first arg-max
__global__ void maxpool_kernel(const float *__restrict__ y, const float *__restrict__ coef, int ns, int C, int64_t G, float *__restrict__ out, int64_t ld_out,
                               int *__restrict__ arg) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= G * C) return;
    const int64_t g = e / C;
    const int c = (int)(e - g * C);
    const float sc = coef[c], sh = coef[CW + c];
    const float *p = y + g * ns * C + c;
    float best = -1.f;
    int bi = 0;
    for (int n = 0; n < ns; ++n) {
        const float v = fmaxf(fmaf(sc, p[(int64_t)n * C], sh), 0.f);
        if (v > best) { best = v; bi = n; }
    }
    out[g * ld_out + c] = best;
    arg[e] = bi;
}
// backward of max -> ReLU -> BatchNorm for the pooled layer: dZ is non-zero at the arg-max sample only.  Pass 1: dzv = dpool where the
// pooled value is positive, per-block column sums of dZ and dZ xhat.
__global__ __launch_bounds__(256) void pool_bwd_stat_kernel(const float *__restrict__ dpool, int64_t ld_dp, const float *__restrict__ pooled, int64_t ld_p,
                                                            const int *__restrict__ arg, const float *__restrict__ y, const float *__restrict__ coef, int ns, int C,
                                                            int64_t G, int64_t groups_per_block, float *__restrict__ dzv, float *__restrict__ part) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t g0 = (int64_t)blockIdx.x * groups_per_block, g1 = min(G, g0 + groups_per_block);
    const float mu = coef[2 * CW + c], rstd = coef[3 * CW + c];
    float s1 = 0.f, s2 = 0.f;
    for (int64_t g = g0; g < g1; ++g) {
        const float d = pooled[g * ld_p + c] > 0.f ? dpool[g * ld_dp + c] : 0.f;
        dzv[g * C + c] = d;
        const float xh = (y[(g * ns + arg[g * C + c]) * C + c] - mu) * rstd;
        s1 += d; s2 = fmaf(d, xh, s2);
    }
    part[((int64_t)blockIdx.x * 2 + 0) * C + c] = s1;
    part[((int64_t)blockIdx.x * 2 + 1) * C + c] = s2;
}
// sums of the partial rows (float64, fixed order) -> BatchNorm's parameter gradients and the two means of its backward
__global__ void bn_bwd_finalize_kernel(const float *__restrict__ part, int64_t T, int C, double M, float *__restrict__ coef, float *__restrict__ dgamma,
                                       float *__restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int64_t t = 0; t < T; ++t) { a += (double)part[(t * 2 + 0) * C + c]; b += (double)part[(t * 2 + 1) * C + c]; }
    coef[4 * CW + c] = (float)(a / M); coef[5 * CW + c] = (float)(b / M);
    dgamma[c] = (float)b; dbeta[c] = (float)a;
}
// Pass 2: dY[m][c] = sc (dZ - mean dZ - xhat mean(dZ xhat)), dense over all rows
__global__ void pool_bwd_apply_kernel(const float *__restrict__ dzv, const int *__restrict__ arg, const float *__restrict__ y, const float *__restrict__ coef, int ns,
                                      int C, int64_t M, float *__restrict__ dy) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * C) return;
    const int64_t m = e / C;
    const int c = (int)(e - m * C);
    const int64_t g = m / ns;
    const int n = (int)(m - g * ns);
    const float dz = arg[g * C + c] == n ? dzv[g * C + c] : 0.f;
    const float xh = (y[e] - coef[2 * CW + c]) * coef[3 * CW + c];
    dy[e] = coef[c] * (dz - coef[4 * CW + c] - xh * coef[5 * CW + c]);
}
// backward of ReLU -> BatchNorm for a layer whose output gradient dA is dense.  Pass 1: per-block sums of dZ = dA [z > 0] and dZ xhat
__global__ __launch_bounds__(256) void relu_bwd_stat_kernel(const float *__restrict__ da, const float *__restrict__ y, const float *__restrict__ coef, int64_t M, int C,
                                                            int64_t rows_per_block, float *__restrict__ part) {
    __shared__ float red[2][4][64];
    const int c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const float sc = coef[c], sh = coef[CW + c], mu = coef[2 * CW + c], rstd = coef[3 * CW + c];
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const float yy = y[r * C + c];
            const float d = fmaf(sc, yy, sh) > 0.f ? da[r * C + c] : 0.f;
            s1 += d; s2 = fmaf(d, (yy - mu) * rstd, s2);
        }
    }
    red[0][rl][threadIdx.x & 63] = s1; red[1][rl][threadIdx.x & 63] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int t = threadIdx.x;
        part[((int64_t)blockIdx.x * 2 + 0) * C + c] = (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
        part[((int64_t)blockIdx.x * 2 + 1) * C + c] = (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
    }
}
// Pass 2, in place: dA -> dY.  coef == null: plain ReLU (the gripper encoder has no BatchNorm): dY = dA [y > 0]
__global__ void relu_bwd_apply_kernel(float *__restrict__ da, const float *__restrict__ y, const float *__restrict__ coef, int64_t M, int C) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * C) return;
    const int c = (int)(e % C);
    const float yy = y[e];
    if (!coef) { da[e] = yy > 0.f ? da[e] : 0.f; return; }
    const float sc = coef[c];
    const float dz = fmaf(sc, yy, coef[CW + c]) > 0.f ? da[e] : 0.f;
    da[e] = sc * (dz - coef[4 * CW + c] - (yy - coef[2 * CW + c]) * coef[3 * CW + c] * coef[5 * CW + c]);
}
// index_points' backward (pointnet2_utils.py:51-68 under autograd): dpts[r][k][:] = sum over the entries e of cloud r's ball lists with
// idx[r][e] == k, in ascending e, of dfeat[r][e][3 : 3 + D].  One wave per (cloud, point): ballot over 64 list entries at a time.
__global__ __launch_bounds__(256) void index_points_bwd_kernel(const float *__restrict__ dfeat, int Cp, const int *__restrict__ idx, int E, int N, int D, int64_t R,
                                                               float *__restrict__ dpts) {
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= R * N) return;
    const int64_t r = wv / N;
    const int k = (int)(wv - r * N);
    const int *il = idx + r * E;
    const float *df = dfeat + r * E * Cp + 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;        // lane covers channels lane, lane + 64, lane + 128, lane + 192 (D <= 256)
    for (int e0 = 0; e0 < E; e0 += 64) {
        const bool hit = e0 + lane < E && il[e0 + lane] == k;
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float *row = df + (int64_t)(e0 + j) * Cp;
            if (lane < D) a0 += row[lane];
            if (lane + 64 < D) a1 += row[lane + 64];
            if (lane + 128 < D) a2 += row[lane + 128];
            if (lane + 192 < D) a3 += row[lane + 192];
        }
    }
    float *o = dpts + (r * N + k) * D;
    if (lane < D) o[lane] = a0;
    if (lane + 64 < D) o[lane + 64] = a1;
    if (lane + 128 < D) o[lane + 128] = a2;
    if (lane + 192 < D) o[lane + 192] = a3;
}
// noisy control values of channel 1 (trainer.py:68-79: noise = cat(zeros, randn, zeros) -> only channel 1 is noised, and only channel 1
// reaches the model, profile_forward_3d.py:77), pose embedding and raw timestep embedding into the trunk's input
// X0 [R][800] = [object (256) | gripper (256) | pose (27) | time (256) | 0 x 5]   (profile_forward_3d.py:78-84)
__global__ void prep3d_kernel(const float *__restrict__ ctrl1, const float *__restrict__ noise, const float *__restrict__ sa, const float *__restrict__ sb,
                              const float *__restrict__ ori, const float *__restrict__ pos, const float *__restrict__ t, const float *__restrict__ freqs, int64_t R, int L,
                              int Lp, float *__restrict__ bufC, float *__restrict__ X0) {
    const int S = Lp + 27 + 256;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * S) return;
    const int64_t r = e / S;
    int s = (int)(e - r * S);
    if (s < Lp) {
        bufC[r * Lp + s] = s < L ? (noise ? add_rn(mul_rn(sa[r], ctrl1[r * L + s]), mul_rn(sb[r], noise[r * L + s])) : ctrl1[r * L + s]) : 0.f;
        return;
    }
    s -= Lp;
    if (s < 27) {       // get_embedder(d, 4): [x, sin(2^k x), cos(2^k x)]_k; cat(embed(ori) [9], embed(pos) [18])
        float v = 0.f;
        if (s == 0) v = ori[r];
        else if (s < 9) { const int k = (s - 1) >> 1; const float a = ori[r] * (float)(1 << k); v = (s - 1) & 1 ? cosf(a) : sinf(a); }
        else if (s < 11) v = pos[2 * r + (s - 9)];
        else { const int q = s - 11, k = q >> 2, which = q & 3; const float a = pos[2 * r + (which & 1)] * (float)(1 << k); v = which & 2 ? cosf(a) : sinf(a); }
        X0[r * 800 + 512 + s] = v;
        return;
    }
    s -= 27;            // timestep_embedding(t, 256) = [cos(t f) | sin(t f)] (profile_forward_2d.py:58-76)
    const float a = t[r] * freqs[s & 127];
    X0[r * 800 + 539 + s] = s < 128 ? cosf(a) : sinf(a);
}
// nn.MSELoss()(pred, score) (trainer.py:88,99): d loss / d pred = 2 (pred - score) / (3 R); per-row squared errors
__global__ void mse_kernel(const float *__restrict__ pred, const float *__restrict__ score, int64_t R, float inv, float *__restrict__ dpred, float *__restrict__ part) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) { const float d = pred[r * 4 + j] - score[r * 3 + j]; a = fmaf(d, d, a); dpred[r * 4 + j] = d * inv; }
    dpred[r * 4 + 3] = 0.f;
    part[r] = a;
}
__global__ void mse_finish_kernel(const float *__restrict__ part, int64_t R, int64_t RT, float *__restrict__ loss) {
    if (threadIdx.x || blockIdx.x) return;
    double a = 0.0;
    for (int64_t r = 0; r < R; ++r) a += (double)part[r];
    *loss = (float)(a / (3.0 * (double)RT));          // RT rows in the batch the mean runs over (R of them here)
}
__global__ void copy_cols_kernel(const float *__restrict__ src, int64_t ld_s, float *__restrict__ dst, int64_t ld_d, int64_t M, int C) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * C) return;
    const int64_t m = e / C;
    const int c = (int)(e - m * C);
    dst[m * ld_d + c] = src[m * ld_s + c];
}

std::vector<float> tfreqs128() {      // timestep_embedding (profile_forward_2d.py:68-71), float32 ops
    std::vector<float> f(128);
    const float l = -(float)std::log(10000.0);
    for (int i = 0; i < 128; ++i) f[i] = expf(l * (float)i / 128.f);
    return f;
}

}  // namespace
}  // namespace dgdm

using namespace dgdm;

struct DgdmTrainer3d {
    struct Named { std::string name; size_t off; int64_t numel; int kind; };      // kind 0 parameter, 1 running_mean, 2 running_var
    struct Lin { int in = 0, out = 0; size_t w = 0, b = 0; int F = -1, B = -1; };
    struct Bn { int C = 0; size_t g = 0; int slot = 0; };
    std::vector<Named> named;
    size_t n_params = 0, n_trainable = 0;
    DevBuf P, G, M1, V, IMG, descs_dev, ws, wpart, run /* [slots][2][CW] */, coef /* [slots][6][CW] */, freqs, loss_dev;
    std::vector<ImgDesc> descs;
    size_t n_img = 0;
    int max_img_elems = 0, L = 42, Lp = 48, N = 512, n_bn = 0;
    float beta1 = 0.9f, beta2 = 0.95f, eps = 1e-8f, wd = 0.f;
    int64_t adam_steps = 0, bn_batches = 0, R_ws = 0, wpart_floats = 0;
    // the two FPS start vectors of a forward: pinned staging (two slots, alternated per call) -> device, no allocation and no pipeline drain
    // per step (the index entry point dgdm_farthest_point_sample allocates, uploads and synchronises on every call)
    int *start_pin = nullptr; DevBuf start_dev; int64_t start_cap = 0; int start_slot = 0;
    ~DgdmTrainer3d() { if (start_pin) (void)hipHostFree(start_pin); }
    int upload_starts(const int64_t *s1, const int64_t *s2, int64_t R, hipStream_t s, const int **d1, const int **d2);
    Lin g0, g2, sa[5], tr[8], outl, te0, te2;
    Bn sabn[5], trbn[8];
    // workspace
    float *xyz = nullptr, *nx1 = nullptr, *nx2 = nullptr, *feat1 = nullptr, *y11 = nullptr, *a11 = nullptr, *y12 = nullptr, *l1p = nullptr, *feat2 = nullptr,
          *y21 = nullptr, *a21 = nullptr, *y22 = nullptr, *l2p = nullptr, *feat3 = nullptr, *y3 = nullptr, *dy3 = nullptr, *dfeat3 = nullptr, *dy22 = nullptr,
          *da21 = nullptr, *dfeat2 = nullptr, *dl1p = nullptr, *dy12 = nullptr, *da11 = nullptr, *dzv = nullptr, *bufC = nullptr, *gh = nullptr, *ga = nullptr,
          *X0 = nullptr, *dX0 = nullptr, *ty[8] = {}, *ta[8] = {}, *td[2] = {}, *pred = nullptr, *dpred = nullptr, *lpart = nullptr, *cpart = nullptr, *sums = nullptr;
    int *fps1 = nullptr, *idx1 = nullptr, *arg1 = nullptr, *fps2 = nullptr, *idx2 = nullptr, *arg2 = nullptr, *arg3 = nullptr;

    float *p(size_t o) const { return P.as<float>() + o; }
    float *gr(size_t o) const { return G.as<float>() + o; }
    float *cf(int slot) const { return coef.as<float>() + (size_t)slot * 6 * CW; }
    float *rn(int slot, int which) const { return run.as<float>() + ((size_t)slot * 2 + which) * CW; }
    size_t add_param(const std::string &name, int64_t numel, int kind = 0) { named.push_back({name, n_params, numel, kind}); const size_t o = n_params; n_params += (size_t)numel; return o; }
    int add_img(size_t w_off, int Kblk, int N_, int s_kc, int s_n);
    void make_lin(Lin &l, const std::string &name, int in, int out);
    void make_bn(Bn &b, const std::string &name, int C);
    int reserve(int64_t R);
    int rowgemm(const float *A, int64_t a_rs, int img, float *C, int64_t c_rs, const float *bias, int64_t M, hipStream_t s) const;
    int colgemm(const float *A, int64_t a_rs, int img, const float *D, int64_t d_rs, int64_t M, hipStream_t s);
    int bias_grad(const float *D, int64_t rs, int64_t M, int C, size_t b_off, hipStream_t s);
    int lin_fwd(const Lin &l, const float *A, int64_t a_rs, float *C, int64_t c_rs, int64_t M, hipStream_t s) const { return rowgemm(A, a_rs, l.F, C, c_rs, p(l.b), M, s); }
    int lin_bwd(const Lin &l, const float *A, int64_t a_rs, const float *dY, int64_t dy_rs, float *dX, int64_t dx_rs, int64_t M, hipStream_t s);
    int bn_fwd(const Bn &b, const float *y, int64_t M, bool train, hipStream_t s);
    int bn_bwd_finalize(const Bn &b, int64_t T, int64_t M, hipStream_t s);
    int relu_bwd(const Bn *b, float *da, const float *y, int64_t M, int C, hipStream_t s);
    int run_step(const float *ctrl1, const float *noise, const float *sa_, const float *sb_, const float *t, const float *ori, const float *pos, const float *xyz_in,
                 const int64_t *start1, const int64_t *start2, const float *score, int64_t R, float lr, int train, float *pred_out, float *loss_host, hipStream_t s,
                 int64_t total_rows = 0 /* 0: R - the loss is the mean over this many rows (a data-parallel chunk: the whole batch's) */, bool apply = true);
    int adam(float lr, hipStream_t s);
    int repack(hipStream_t s);
    int copy_state(int which, DgdmTensor *t, int n, bool to_device);
};

int DgdmTrainer3d::add_img(size_t w_off, int Kblk, int N_, int s_kc, int s_n) {
    ImgDesc d{};
    d.src = (int64_t)w_off; d.dst = (int64_t)n_img; d.Kblk = Kblk; d.ntaps = 1; d.taps[0] = 0;
    d.K = Kblk; d.Kp = round_up(Kblk, KC); d.N = N_; d.Np = round_up(N_, TN); d.s_kc = s_kc; d.s_n = s_n;
    n_img += (size_t)d.Kp * d.Np;
    max_img_elems = std::max(max_img_elems, d.Kp * d.Np);
    descs.push_back(d);
    return (int)descs.size() - 1;
}
// Linear / Conv2d(1x1) weight [out][in]: forward image [in][out], input-gradient image [out][in]
void DgdmTrainer3d::make_lin(Lin &l, const std::string &name, int in, int out) {
    l.in = in; l.out = out;
    l.w = add_param(name + ".weight", (int64_t)in * out);
    l.b = add_param(name + ".bias", out);
    l.F = add_img(l.w, in, out, 1, in);
    l.B = add_img(l.w, out, in, in, 1);
}
void DgdmTrainer3d::make_bn(Bn &b, const std::string &name, int C) {
    b.C = C; b.slot = n_bn++;
    b.g = add_param(name + ".weight", C);
    add_param(name + ".bias", C);
}

int DgdmTrainer3d::rowgemm(const float *A, int64_t a_rs, int img, float *C, int64_t c_rs, const float *bias, int64_t M, hipStream_t s) const {
    const ImgDesc &d = descs[img];
    RowGemm g{};
    g.A = A; g.a_rs = a_rs; g.B = IMG.as<float>() + d.dst; g.Kp = d.Kp; g.Np = d.Np; g.C = C; g.c_rs = c_rs; g.N = d.N; g.bias = bias; g.M = M;
    g.mk = RowMask{0, 0, 0}; g.scalar_a = (a_rs & 3) != 0 || (reinterpret_cast<uintptr_t>(A) & 15) != 0;
    hipLaunchKernelGGL(rowgemm_kernel, dim3((unsigned)((M + TM - 1) / TM), (unsigned)(d.Np / TN)), dim3(256), 0, s, g);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
int DgdmTrainer3d::colgemm(const float *A, int64_t a_rs, int img, const float *D, int64_t d_rs, int64_t M, hipStream_t s) {
    const ImgDesc &d = descs[img];
    const int kt = (d.Kp + TM - 1) / TM, nt = d.Np / TN;
    int64_t splits = std::min<int64_t>(std::max<int64_t>(1, M / 256), std::max(1, 1024 / (kt * nt)));
    int64_t per = ((M + splits - 1) / splits + KC - 1) / KC * KC;
    splits = (M + per - 1) / per;
    ColGemm g{};
    g.A = A; g.a_rs = a_rs; g.D = D; g.d_rs = d_rs; g.part = wpart.as<float>(); g.ldp = nt * TN; g.split_stride = (int64_t)kt * TM * g.ldp; g.M = M; g.m_per_split = per;
    g.scalar_a = (a_rs & 3) != 0 || (reinterpret_cast<uintptr_t>(A) & 15) != 0;
    g.scalar_d = (d_rs & 3) != 0 || (reinterpret_cast<uintptr_t>(D) & 15) != 0;
    DGDM_REQUIRE(splits * g.split_stride <= wpart_floats, DGDM_EINVAL, "trainer3d: weight-gradient partials do not fit");
    hipLaunchKernelGGL(colgemm_kernel, dim3(kt, nt, (unsigned)splits), dim3(256), 0, s, g);
    DGDM_HIP_CHECK(hipGetLastError());
    const int64_t n = (int64_t)d.K * d.N;
    hipLaunchKernelGGL(wgrad_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wpart.as<float>(), (int)splits, g.split_stride, g.ldp,
                       descs_dev.as<ImgDesc>(), img, G.as<float>());
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
static int64_t stat_blocks(int64_t M, int64_t &rows_per_block) {
    int64_t blocks = std::min<int64_t>(std::max<int64_t>(1, (M + 255) / 256), 2048);
    rows_per_block = (M + blocks - 1) / blocks;
    return (M + rows_per_block - 1) / rows_per_block;
}
int DgdmTrainer3d::bias_grad(const float *D, int64_t rs, int64_t M, int C, size_t b_off, hipStream_t s) {
    int64_t rpb;
    const int64_t blocks = stat_blocks(M, rpb);
    hipLaunchKernelGGL(colstat_kernel, dim3((unsigned)blocks, (C + 63) / 64), dim3(256), 0, s, D, rs, M, C, rpb, (const float *)nullptr, cpart);
    DGDM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(rows_sum_kernel, rows_sum_grid(C), dim3(256), 0, s, cpart, blocks, C, gr(b_off));
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
// input gradient (dX == null: the layer reads data), weight and bias gradients of y = A W^T + b
int DgdmTrainer3d::lin_bwd(const Lin &l, const float *A, int64_t a_rs, const float *dY, int64_t dy_rs, float *dX, int64_t dx_rs, int64_t M, hipStream_t s) {
    int rc;
    if (dX && (rc = rowgemm(dY, dy_rs, l.B, dX, dx_rs, nullptr, M, s))) return rc;
    if ((rc = colgemm(A, a_rs, l.F, dY, dy_rs, M, s))) return rc;
    return bias_grad(dY, dy_rs, M, l.out, l.b, s);
}
// batch statistics (train) or the running ones (eval) -> coef
int DgdmTrainer3d::bn_fwd(const Bn &b, const float *y, int64_t M, bool train, hipStream_t s) {
    const int C = b.C;
    if (!train) {
        hipLaunchKernelGGL(bn_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, s, C, p(b.g), p(b.g) + C, 1e-5f, rn(b.slot, 0), rn(b.slot, 1), cf(b.slot));
        DGDM_HIP_CHECK(hipGetLastError());
        return DGDM_OK;
    }
    int64_t rpb;
    const int64_t blocks = stat_blocks(M, rpb);
    hipLaunchKernelGGL(colstat_kernel, dim3((unsigned)blocks, (C + 63) / 64), dim3(256), 0, s, y, (int64_t)C, M, C, rpb, (const float *)nullptr, cpart);
    hipLaunchKernelGGL(rows_sum_kernel, rows_sum_grid(C), dim3(256), 0, s, cpart, blocks, C, sums);
    hipLaunchKernelGGL(bn_mean_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, (double)M, C, cf(b.slot));
    hipLaunchKernelGGL(colstat_kernel, dim3((unsigned)blocks, (C + 63) / 64), dim3(256), 0, s, y, (int64_t)C, M, C, rpb, (const float *)(cf(b.slot) + 2 * CW), cpart);
    hipLaunchKernelGGL(rows_sum_kernel, rows_sum_grid(C), dim3(256), 0, s, cpart, blocks, C, sums);
    hipLaunchKernelGGL(bn_var_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, (double)M, C, p(b.g), p(b.g) + C, 1e-5f, 0.1f, rn(b.slot, 0), rn(b.slot, 1), cf(b.slot));
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
int DgdmTrainer3d::bn_bwd_finalize(const Bn &b, int64_t T, int64_t M, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((b.C + 255) / 256), dim3(256), 0, s, cpart, T, b.C, (double)M, cf(b.slot), gr(b.g), gr(b.g) + b.C);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
// dA -> dY in place through ReLU (-> BatchNorm when b != null; its parameter gradients are written)
int DgdmTrainer3d::relu_bwd(const Bn *b, float *da, const float *y, int64_t M, int C, hipStream_t s) {
    if (b) {
        int64_t rpb;
        const int64_t blocks = stat_blocks(M, rpb);
        hipLaunchKernelGGL(relu_bwd_stat_kernel, dim3((unsigned)blocks, (C + 63) / 64), dim3(256), 0, s, da, y, cf(b->slot), M, C, rpb, cpart);
        DGDM_HIP_CHECK(hipGetLastError());
        int rc = bn_bwd_finalize(*b, blocks, M, s);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(relu_bwd_apply_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, s, da, y, b ? cf(b->slot) : (const float *)nullptr, M, C);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int DgdmTrainer3d::reserve(int64_t R) {
    if (R <= R_ws) return DGDM_OK;
    const int64_t Mr1 = R * 512, M1r = Mr1 * 32, Mr2 = R * 128, M2r = Mr2 * 64;
    std::vector<std::pair<void **, int64_t>> want;
    auto F = [&](float *&q, int64_t n) { want.push_back({(void **)&q, n + 2 * GUARD}); };
    auto I = [&](int *&q, int64_t n) { want.push_back({(void **)&q, n + 2 * GUARD}); };
    F(xyz, R * N * 3); F(nx1, Mr1 * 3); F(nx2, Mr2 * 3);
    I(fps1, Mr1); I(idx1, M1r); I(arg1, Mr1 * 128); I(fps2, Mr2); I(idx2, M2r); I(arg2, Mr2 * 256); I(arg3, R * 256);
    F(feat1, M1r * 4); F(y11, M1r * 64); F(a11, M1r * 64); F(y12, M1r * 128); F(l1p, Mr1 * 128);
    F(feat2, M2r * 132); F(y21, M2r * 128); F(a21, M2r * 128); F(y22, M2r * 256); F(l2p, Mr2 * 256);
    F(feat3, Mr2 * 260); F(y3, Mr2 * 256);
    F(dy3, Mr2 * 256); F(dfeat3, Mr2 * 260); F(dy22, M2r * 256); F(da21, M2r * 128); F(dfeat2, M2r * 132); F(dl1p, Mr1 * 128); F(dy12, M1r * 128); F(da11, M1r * 64);
    F(dzv, std::max(Mr1 * 128, Mr2 * 256));
    F(bufC, R * Lp); F(gh, R * 256); F(ga, R * 256); F(X0, R * 800); F(dX0, R * 800);
    for (int k = 0; k < 8; ++k) { F(ty[k], R * 512); F(ta[k], R * 512); }
    F(td[0], R * 512); F(td[1], R * 512); F(pred, R * 4); F(dpred, R * 4); F(lpart, R);
    F(cpart, (int64_t)2048 * 2 * CW); F(sums, 2 * CW);
    int64_t total = 0;
    for (auto &w : want) total += (w.second + 63) / 64 * 64;
    int rc = ws.alloc((size_t)total * sizeof(float));
    if (rc) { set_error("trainer3d: the workspace for %lld clouds per call is %.1f GB (68 MB per cloud): %s", (long long)R, (double)total * 4e-9, dgdm_last_error()); return rc; }
    DGDM_HIP_CHECK(hipMemset(ws.p, 0, (size_t)total * sizeof(float)));
    float *q = ws.as<float>();
    for (auto &w : want) { *w.first = q + GUARD; q += (w.second + 63) / 64 * 64; }
    wpart_floats = (int64_t)1024 * TM * TN + (int64_t)64 * TM * TN;
    if ((rc = wpart.alloc((size_t)wpart_floats * sizeof(float)))) return rc;
    R_ws = R;
    return DGDM_OK;
}

int DgdmTrainer3d::repack(hipStream_t s) {
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)((max_img_elems + 255) / 256), (unsigned)descs.size()), dim3(256), 0, s, P.as<float>(), IMG.as<float>(),
                       descs_dev.as<ImgDesc>());
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
// torch.optim.Adam over the parameters that received a gradient (time_encoder is never called by ProfileForward3DModel.forward: its
// .grad stays None and torch skips it - those tensors sit behind n_trainable)
int DgdmTrainer3d::adam(float lr, hipStream_t s) {
    ++adam_steps;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)adam_steps), bc2 = 1.0 - std::pow((double)beta2, (double)adam_steps);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n_trainable + 255) / 256)), dim3(256), 0, s, P.as<float>(), G.as<float>(), M1.as<float>(), V.as<float>(),
                       (int64_t)n_trainable, beta1, beta2, eps, wd, (float)((double)lr / bc1), (float)std::sqrt(bc2));
    DGDM_HIP_CHECK(hipGetLastError());
    return repack(s);
}

int DgdmTrainer3d::upload_starts(const int64_t *s1, const int64_t *s2, int64_t R, hipStream_t s, const int **d1, const int **d2) {
    if (R > start_cap) {
        if (start_pin) { DGDM_HIP_CHECK(hipStreamSynchronize(s)); DGDM_HIP_CHECK(hipHostFree(start_pin)); start_pin = nullptr; }
        DGDM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&start_pin), (size_t)4 * R * sizeof(int), hipHostMallocDefault));
        int rc = start_dev.alloc((size_t)4 * R * sizeof(int));
        if (rc) return rc;
        start_cap = R;
    }
    start_slot ^= 1;
    int *pin = start_pin + (size_t)start_slot * 2 * start_cap;
    for (int64_t i = 0; i < R; ++i) {
        DGDM_REQUIRE(s1[i] >= 0 && s1[i] < N, DGDM_EINVAL, "FPS start %lld outside [0, %d)", (long long)s1[i], N);
        DGDM_REQUIRE(s2[i] >= 0 && s2[i] < 512, DGDM_EINVAL, "FPS start %lld outside [0, 512)", (long long)s2[i]);
        pin[i] = (int)s1[i]; pin[start_cap + i] = (int)s2[i];
    }
    int *dev = start_dev.as<int>() + (size_t)start_slot * 2 * start_cap;
    DGDM_HIP_CHECK(hipMemcpyAsync(dev, pin, (size_t)2 * start_cap * sizeof(int), hipMemcpyHostToDevice, s));
    *d1 = dev; *d2 = dev + start_cap;
    return DGDM_OK;
}

int DgdmTrainer3d::run_step(const float *ctrl1, const float *noise, const float *sa_, const float *sb_, const float *t, const float *ori, const float *pos,
                            const float *xyz_in, const int64_t *start1, const int64_t *start2, const float *score, int64_t R, float lr, int train, float *pred_out,
                            float *loss_host, hipStream_t s, int64_t total_rows, bool apply) {
    int rc = reserve(R);
    if (rc) return rc;
    const int64_t RT = total_rows > 0 ? total_rows : R;
    const bool tr_ = train != 0;
    const int64_t Mr1 = R * 512, M1r = Mr1 * 32, Mr2 = R * 128, M2r = Mr2 * 64;
    auto grid = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    // ---- inputs
    DGDM_HIP_CHECK(hipMemcpyAsync(xyz, xyz_in, (size_t)R * N * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(prep3d_kernel, grid(R * (Lp + 283)), dim3(256), 0, s, ctrl1, noise, sa_, sb_, ori, pos, t, freqs.as<float>(), R, L, Lp, bufC, X0);
    DGDM_HIP_CHECK(hipGetLastError());
    // ---- gripper encoder: Linear -> ReLU -> Linear into X0[:, 256:512]   (profile_forward_3d.py:33-37,77)
    if ((rc = lin_fwd(g0, bufC, Lp, gh, 256, R, s))) return rc;
    hipLaunchKernelGGL(bn_relu_kernel, grid(R * 64), dim3(256), 0, s, gh, (const float *)nullptr, ga, R, 256, (int64_t)256);
    if ((rc = lin_fwd(g2, ga, 256, X0 + 256, 800, R, s))) return rc;
    // ---- PointNet++ (pointnet2.py:21-32).  sa1: FPS(512) from the drawn start, ball query r = 0.2 / 32 in the original order
    const int *st1 = nullptr, *st2 = nullptr;
    if ((rc = upload_starts(start1, start2, R, s, &st1, &st2))) return rc;
    if ((rc = pn_fps_rows(xyz, st1, (int)R, N, 512, fps1, s))) return rc;
    if ((rc = dgdm_index_points(xyz, fps1, (int)R, N, 512, 3, nx1, s))) return rc;
    if ((rc = dgdm_query_ball_point((float)(0.2 * 0.2), 32, xyz, nx1, (int)R, N, 512, idx1, s))) return rc;
    hipLaunchKernelGGL(group_kernel, grid(M1r), dim3(256), 0, s, xyz, (const float *)nullptr, fps1, idx1, N, 512, 32, 0, 4, M1r, feat1);
    DGDM_HIP_CHECK(hipGetLastError());
    if ((rc = lin_fwd(sa[0], feat1, 4, y11, 64, M1r, s))) return rc;
    if ((rc = bn_fwd(sabn[0], y11, M1r, tr_, s))) return rc;
    hipLaunchKernelGGL(bn_relu_kernel, grid(M1r * 16), dim3(256), 0, s, y11, (const float *)cf(sabn[0].slot), a11, M1r, 64, (int64_t)64);
    if ((rc = lin_fwd(sa[1], a11, 64, y12, 128, M1r, s))) return rc;
    if ((rc = bn_fwd(sabn[1], y12, M1r, tr_, s))) return rc;
    hipLaunchKernelGGL(maxpool_kernel, grid(Mr1 * 128), dim3(256), 0, s, y12, (const float *)cf(sabn[1].slot), 32, 128, Mr1, l1p, (int64_t)128, arg1);
    DGDM_HIP_CHECK(hipGetLastError());
    // sa2 on the 512 sampled points (order fps1) with their features: FPS(128), ball query r = 0.4 / 64
    if ((rc = pn_fps_rows(nx1, st2, (int)R, 512, 128, fps2, s))) return rc;
    if ((rc = dgdm_index_points(nx1, fps2, (int)R, 512, 128, 3, nx2, s))) return rc;
    if ((rc = dgdm_query_ball_point((float)(0.4 * 0.4), 64, nx1, nx2, (int)R, 512, 128, idx2, s))) return rc;
    hipLaunchKernelGGL(group_kernel, grid(M2r * 33), dim3(256), 0, s, nx1, (const float *)l1p, fps2, idx2, 512, 128, 64, 128, 132, M2r, feat2);
    DGDM_HIP_CHECK(hipGetLastError());
    if ((rc = lin_fwd(sa[2], feat2, 132, y21, 128, M2r, s))) return rc;
    if ((rc = bn_fwd(sabn[2], y21, M2r, tr_, s))) return rc;
    hipLaunchKernelGGL(bn_relu_kernel, grid(M2r * 32), dim3(256), 0, s, y21, (const float *)cf(sabn[2].slot), a21, M2r, 128, (int64_t)128);
    if ((rc = lin_fwd(sa[3], a21, 128, y22, 256, M2r, s))) return rc;
    if ((rc = bn_fwd(sabn[3], y22, M2r, tr_, s))) return rc;
    hipLaunchKernelGGL(maxpool_kernel, grid(Mr2 * 256), dim3(256), 0, s, y22, (const float *)cf(sabn[3].slot), 64, 256, Mr2, l2p, (int64_t)256, arg2);
    DGDM_HIP_CHECK(hipGetLastError());
    // sa3: group_all over the 128 centres with their ABSOLUTE coordinates (sample_and_group_all)
    hipLaunchKernelGGL(group_kernel, grid(Mr2 * 65), dim3(256), 0, s, nx2, (const float *)l2p, (const int *)nullptr, (const int *)nullptr, 128, 1, 128, 256, 260, Mr2, feat3);
    DGDM_HIP_CHECK(hipGetLastError());
    if ((rc = lin_fwd(sa[4], feat3, 260, y3, 256, Mr2, s))) return rc;
    if ((rc = bn_fwd(sabn[4], y3, Mr2, tr_, s))) return rc;
    hipLaunchKernelGGL(maxpool_kernel, grid(R * 256), dim3(256), 0, s, y3, (const float *)cf(sabn[4].slot), 128, 256, R, X0, (int64_t)800, arg3);
    DGDM_HIP_CHECK(hipGetLastError());
    // ---- trunk: Linear -> BatchNorm1d -> ReLU x 8, output layer (profile_forward_3d.py:39-65,84-85)
    for (int k = 0; k < 8; ++k) {
        const float *in = k == 0 ? X0 : ta[k - 1];
        if ((rc = lin_fwd(tr[k], in, k == 0 ? 800 : tr[k].in, ty[k], tr[k].out, R, s))) return rc;
        if ((rc = bn_fwd(trbn[k], ty[k], R, tr_, s))) return rc;
        hipLaunchKernelGGL(bn_relu_kernel, grid(R * tr[k].out / 4), dim3(256), 0, s, ty[k], (const float *)cf(trbn[k].slot), ta[k], R, tr[k].out, (int64_t)tr[k].out);
        DGDM_HIP_CHECK(hipGetLastError());
    }
    if ((rc = rowgemm(ta[7], 256, outl.F, pred, 4, p(outl.b), R, s))) return rc;
    hipLaunchKernelGGL(mse_kernel, grid(R), dim3(256), 0, s, pred, score, R, (float)(2.0 / (3.0 * (double)RT)), dpred, lpart);
    hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(1), 0, s, lpart, R, RT, loss_dev.as<float>());
    DGDM_HIP_CHECK(hipGetLastError());
    if (pred_out) {
        hipLaunchKernelGGL(copy_cols_kernel, grid(R * 3), dim3(256), 0, s, pred, (int64_t)4, pred_out, (int64_t)3, R, 3);
        DGDM_HIP_CHECK(hipGetLastError());
    }
    if (tr_) {
        // ---- backward: output layer, trunk
        float *d = td[0], *dn = td[1];
        if ((rc = rowgemm(dpred, 4, outl.B, d, 256, nullptr, R, s))) return rc;
        if ((rc = colgemm(ta[7], 256, outl.F, dpred, 4, R, s))) return rc;
        if ((rc = bias_grad(dpred, 4, R, 3, outl.b, s))) return rc;
        for (int k = 7; k >= 0; --k) {
            if ((rc = relu_bwd(&trbn[k], d, ty[k], R, tr[k].out, s))) return rc;
            const float *in = k == 0 ? X0 : ta[k - 1];
            float *dx = k == 0 ? dX0 : dn;
            const int64_t ld = k == 0 ? 800 : tr[k].in;
            if ((rc = lin_bwd(tr[k], in, ld, d, tr[k].out, dx, ld, R, s))) return rc;
            std::swap(d, dn);
        }
        // gripper encoder: its output is columns 256..511 of X0
        if ((rc = lin_bwd(g2, ga, 256, dX0 + 256, 800, d, 256, R, s))) return rc;
        if ((rc = relu_bwd(nullptr, d, gh, R, 256, s))) return rc;
        if ((rc = lin_bwd(g0, bufC, Lp, d, 256, nullptr, 0, R, s))) return rc;
        // ---- PointNet++ backward; the object embedding is columns 0..255 of X0
        auto pool_bwd = [&](const Bn &b, const float *dpool, int64_t ld_dp, const float *pooled, int64_t ld_p, const int *arg, const float *y, int ns, int64_t Gr,
                            float *dy) -> int {
            const int C = b.C;
            int64_t blocks = std::min<int64_t>(std::max<int64_t>(1, (Gr + 63) / 64), 2048);
            const int64_t gpb = (Gr + blocks - 1) / blocks;
            blocks = (Gr + gpb - 1) / gpb;
            hipLaunchKernelGGL(pool_bwd_stat_kernel, dim3((unsigned)blocks, (C + 255) / 256), dim3(256), 0, s, dpool, ld_dp, pooled, ld_p, arg, y, (const float *)cf(b.slot), ns,
                               C, Gr, gpb, dzv, cpart);
            DGDM_HIP_CHECK(hipGetLastError());
            int rc2 = bn_bwd_finalize(b, blocks, Gr * ns, s);
            if (rc2) return rc2;
            hipLaunchKernelGGL(pool_bwd_apply_kernel, grid(Gr * ns * C), dim3(256), 0, s, (const float *)dzv, arg, y, (const float *)cf(b.slot), ns, C, Gr * ns, dy);
            DGDM_HIP_CHECK(hipGetLastError());
            return DGDM_OK;
        };
        if ((rc = pool_bwd(sabn[4], dX0, 800, X0, 800, arg3, y3, 128, R, dy3))) return rc;
        if ((rc = lin_bwd(sa[4], feat3, 260, dy3, 256, dfeat3, 260, Mr2, s))) return rc;
        if ((rc = pool_bwd(sabn[3], dfeat3 + 3, 260, l2p, 256, arg2, y22, 64, Mr2, dy22))) return rc;
        if ((rc = lin_bwd(sa[3], a21, 128, dy22, 256, da21, 128, M2r, s))) return rc;
        if ((rc = relu_bwd(&sabn[2], da21, y21, M2r, 128, s))) return rc;
        if ((rc = lin_bwd(sa[2], feat2, 132, da21, 128, dfeat2, 132, M2r, s))) return rc;
        hipLaunchKernelGGL(index_points_bwd_kernel, dim3((unsigned)((Mr1 + 3) / 4)), dim3(256), 0, s, (const float *)dfeat2, 132, (const int *)idx2, 128 * 64, 512, 128, R, dl1p);
        DGDM_HIP_CHECK(hipGetLastError());
        if ((rc = pool_bwd(sabn[1], dl1p, 128, l1p, 128, arg1, y12, 32, Mr1, dy12))) return rc;
        if ((rc = lin_bwd(sa[1], a11, 64, dy12, 128, da11, 64, M1r, s))) return rc;
        if ((rc = relu_bwd(&sabn[0], da11, y11, M1r, 64, s))) return rc;
        if ((rc = lin_bwd(sa[0], feat1, 4, da11, 64, nullptr, 0, M1r, s))) return rc;
        if (apply && (rc = adam(lr, s))) return rc;
        ++bn_batches;
    }
    if (loss_host) {
        DGDM_HIP_CHECK(hipMemcpyAsync(loss_host, loss_dev.p, sizeof(float), hipMemcpyDeviceToHost, s));
        DGDM_HIP_CHECK(hipStreamSynchronize(s));
    }
    return DGDM_OK;
}

// which: 0 parameters + BatchNorm running statistics, 1 gradients, 2 / 3 Adam's exp_avg / exp_avg_sq
int DgdmTrainer3d::copy_state(int which, DgdmTensor *t, int n, bool to_device) {
    DevBuf *src = which == 0 ? &P : which == 1 ? &G : which == 2 ? &M1 : &V;
    std::vector<float> host(n_params), rs((size_t)n_bn * 2 * CW);
    DGDM_HIP_CHECK(hipDeviceSynchronize());
    DGDM_HIP_CHECK(hipMemcpy(host.data(), src->p, n_params * sizeof(float), hipMemcpyDeviceToHost));
    DGDM_HIP_CHECK(hipMemcpy(rs.data(), run.p, rs.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::map<std::string, DgdmTensor *> by;
    for (int i = 0; i < n; ++i) by[t[i].name] = &t[i];
    for (const Named &nm : named) {
        if (nm.kind != 0 && which != 0) continue;
        auto it = by.find(nm.name);
        if (it == by.end()) { set_error("state_dict key '%s' missing", nm.name.c_str()); return DGDM_EKEY; }
        if (it->second->dtype != 0 || it->second->numel != nm.numel) {
            set_error("state_dict key '%s': expected %lld float32 values, got %lld", nm.name.c_str(), (long long)nm.numel, (long long)it->second->numel);
            return DGDM_EKEY;
        }
        float *user = const_cast<float *>(static_cast<const float *>(it->second->data));
        float *mine = nm.kind == 0 ? &host[nm.off] : &rs[nm.off];
        if (to_device) memcpy(mine, user, (size_t)nm.numel * sizeof(float));
        else memcpy(user, mine, (size_t)nm.numel * sizeof(float));
    }
    if (to_device) {
        DGDM_HIP_CHECK(hipMemcpy(src->p, host.data(), n_params * sizeof(float), hipMemcpyHostToDevice));
        if (which == 0) {
            DGDM_HIP_CHECK(hipMemcpy(run.p, rs.data(), rs.size() * sizeof(float), hipMemcpyHostToDevice));
            int rc = repack(0);
            if (rc) return rc;
            DGDM_HIP_CHECK(hipDeviceSynchronize());
        }
    }
    return DGDM_OK;
}

extern "C" int dgdm_trainer3d_create(DgdmTrainer3d **out, const DgdmTensor *state_dict, int n_tensors, int params_ch, int num_object_points, float beta1,
                                     float beta2, float eps, float weight_decay) {
    DGDM_REQUIRE(out && state_dict && params_ch > 0 && params_ch <= 256, DGDM_EINVAL, "dgdm_trainer3d_create: bad argument");
    DGDM_REQUIRE(num_object_points == 512, DGDM_EINVAL, "dgdm_trainer3d_create: %d object points; the device FPS / ball-query functions and sa1 (npoint = 512) are "
                 "built for the 512 points of dynamics/train_dynamics_3d.sh", num_object_points);
    std::unique_ptr<DgdmTrainer3d> m(new DgdmTrainer3d());
    m->L = params_ch; m->Lp = round_up(params_ch, KC); m->N = num_object_points;
    m->beta1 = beta1; m->beta2 = beta2; m->eps = eps; m->wd = weight_decay;
    auto bn = [&](DgdmTrainer3d::Bn &b, const std::string &name, int C) {
        m->make_bn(b, name, C);
        m->named.push_back({name + ".running_mean", ((size_t)b.slot * 2 + 0) * CW, C, 1});
        m->named.push_back({name + ".running_var", ((size_t)b.slot * 2 + 1) * CW, C, 2});
    };
    const int sadim[5][2] = {{3, 64}, {64, 128}, {131, 128}, {128, 256}, {259, 256}};
    const char *saname[5] = {"object_encoder.sa1", "object_encoder.sa1", "object_encoder.sa2", "object_encoder.sa2", "object_encoder.sa3"};
    const int saidx[5] = {0, 1, 0, 1, 0};
    for (int i = 0; i < 5; ++i) {
        m->make_lin(m->sa[i], std::string(saname[i]) + ".mlp_convs." + std::to_string(saidx[i]), sadim[i][0], sadim[i][1]);
        bn(m->sabn[i], std::string(saname[i]) + ".mlp_bns." + std::to_string(saidx[i]), sadim[i][1]);
    }
    m->make_lin(m->g0, "gripper_encoder.0", params_ch, 256);
    m->make_lin(m->g2, "gripper_encoder.2", 256, 256);
    const int trin[8] = {795, 512, 256, 256, 256, 256, 256, 256}, trout[8] = {512, 256, 256, 256, 256, 256, 256, 256};
    for (int k = 0; k < 8; ++k) {
        m->make_lin(m->tr[k], "linears." + std::to_string(3 * k), trin[k], trout[k]);
        bn(m->trbn[k], "linears." + std::to_string(3 * k + 1), trout[k]);
    }
    m->make_lin(m->outl, "output", 256, 3);
    m->n_trainable = m->n_params;
    m->make_lin(m->te0, "time_encoder.0", 128, 256);       // constructed by the reference (profile_forward_3d.py:27-31), never called: no gradient, no update
    m->make_lin(m->te2, "time_encoder.2", 256, 256);
    int rc;
    for (DevBuf *b : {&m->P, &m->G, &m->M1, &m->V}) {
        if ((rc = b->alloc(m->n_params * sizeof(float)))) return rc;
        DGDM_HIP_CHECK(hipMemset(b->p, 0, m->n_params * sizeof(float)));
    }
    if ((rc = m->IMG.alloc(m->n_img * sizeof(float)))) return rc;
    if ((rc = m->descs_dev.upload(m->descs.data(), m->descs.size() * sizeof(ImgDesc)))) return rc;
    if ((rc = m->run.alloc((size_t)m->n_bn * 2 * CW * sizeof(float)))) return rc;
    DGDM_HIP_CHECK(hipMemset(m->run.p, 0, (size_t)m->n_bn * 2 * CW * sizeof(float)));
    if ((rc = m->coef.alloc((size_t)m->n_bn * 6 * CW * sizeof(float)))) return rc;
    DGDM_HIP_CHECK(hipMemset(m->coef.p, 0, (size_t)m->n_bn * 6 * CW * sizeof(float)));
    if ((rc = m->loss_dev.alloc(64))) return rc;
    const std::vector<float> f = tfreqs128();
    if ((rc = m->freqs.upload(f.data(), f.size() * sizeof(float)))) return rc;
    if ((rc = m->copy_state(0, const_cast<DgdmTensor *>(state_dict), n_tensors, true))) return rc;
    *out = m.release();
    return DGDM_OK;
}

extern "C" void dgdm_trainer3d_destroy(DgdmTrainer3d *m) { delete m; }

extern "C" int dgdm_trainer3d_step(DgdmTrainer3d *m, const float *ctrl1_dev, const float *noise_dev, const float *sqrt_abar_dev, const float *sqrt_1m_abar_dev,
                                   const float *t_dev, const float *ori_dev, const float *pos_dev, const float *xyz_dev, const int64_t *start_sa1_host,
                                   const int64_t *start_sa2_host, const float *score_dev, int64_t rows, float lr, int train, float *pred_dev, float *loss_host,
                                   void *stream) {
    DGDM_REQUIRE(m && ctrl1_dev && t_dev && ori_dev && pos_dev && xyz_dev && start_sa1_host && start_sa2_host && score_dev, DGDM_EINVAL,
                 "dgdm_trainer3d_step: null argument");
    DGDM_REQUIRE(!noise_dev || (sqrt_abar_dev && sqrt_1m_abar_dev), DGDM_EINVAL, "dgdm_trainer3d_step: noise without its two scale vectors");
    DGDM_REQUIRE(rows >= (train ? 2 : 1) && rows <= 4096, DGDM_EINVAL, "dgdm_trainer3d_step: %lld rows (BatchNorm1d in training mode needs at least 2; one call "
                 "holds at most 4096 clouds - 68 MB of activations each - use the sub-batches of --use_sub_batch)", (long long)rows);
    return m->run_step(ctrl1_dev, noise_dev, sqrt_abar_dev, sqrt_1m_abar_dev, t_dev, ori_dev, pos_dev, xyz_dev, start_sa1_host, start_sa2_host, score_dev, rows, lr, train,
                       pred_dev, loss_host, (hipStream_t)stream);
}

// ---- data-parallel pieces (dgdm_amd/dynamics/trainer.py: nn.DataParallel's semantics, dynamics/trainer.py:41-43, one process per GPU)
extern "C" int dgdm_trainer3d_forward_backward(DgdmTrainer3d *m, const float *ctrl1_dev, const float *noise_dev, const float *sqrt_abar_dev,
                                               const float *sqrt_1m_abar_dev, const float *t_dev, const float *ori_dev, const float *pos_dev, const float *xyz_dev,
                                               const int64_t *start_sa1_host, const int64_t *start_sa2_host, const float *score_dev, int64_t rows, int64_t total_rows,
                                               float *pred_dev, float *loss_host, void *stream) {
    DGDM_REQUIRE(m && ctrl1_dev && t_dev && ori_dev && pos_dev && xyz_dev && start_sa1_host && start_sa2_host && score_dev, DGDM_EINVAL,
                 "dgdm_trainer3d_forward_backward: null argument");
    DGDM_REQUIRE(!noise_dev || (sqrt_abar_dev && sqrt_1m_abar_dev), DGDM_EINVAL, "dgdm_trainer3d_forward_backward: noise without its two scale vectors");
    DGDM_REQUIRE(rows >= 2 && rows <= 4096 && total_rows >= rows, DGDM_EINVAL, "dgdm_trainer3d_forward_backward: %lld of %lld rows", (long long)rows, (long long)total_rows);
    return m->run_step(ctrl1_dev, noise_dev, sqrt_abar_dev, sqrt_1m_abar_dev, t_dev, ori_dev, pos_dev, xyz_dev, start_sa1_host, start_sa2_host, score_dev, rows, 0.f, 1,
                       pred_dev, loss_host, (hipStream_t)stream, total_rows, false);
}
extern "C" int64_t dgdm_trainer3d_gradient_count(const DgdmTrainer3d *m) { return m ? (int64_t)m->n_trainable : -1; }
extern "C" int dgdm_trainer3d_gradients(DgdmTrainer3d *m, float *flat_dev, int64_t numel, int to_trainer, void *stream) {
    DGDM_REQUIRE(m && flat_dev && numel == (int64_t)m->n_trainable, DGDM_EINVAL, "dgdm_trainer3d_gradients: expected %lld floats", m ? (long long)m->n_trainable : -1LL);
    DGDM_HIP_CHECK(hipMemcpyAsync(to_trainer ? m->G.p : (void *)flat_dev, to_trainer ? (const void *)flat_dev : m->G.p, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return DGDM_OK;
}
extern "C" int dgdm_trainer3d_apply(DgdmTrainer3d *m, float lr, void *stream) {
    DGDM_REQUIRE(m, DGDM_EINVAL, "dgdm_trainer3d_apply: null handle");
    return m->adam(lr, (hipStream_t)stream);
}
extern "C" int64_t dgdm_trainer3d_running_stats_count(const DgdmTrainer3d *m) { return m ? (int64_t)m->n_bn * 2 * CW : -1; }
extern "C" int dgdm_trainer3d_running_stats(DgdmTrainer3d *m, float *flat_dev, int64_t numel, int to_trainer, void *stream) {
    DGDM_REQUIRE(m && flat_dev && numel == (int64_t)m->n_bn * 2 * CW, DGDM_EINVAL, "dgdm_trainer3d_running_stats: expected %lld floats", m ? (long long)m->n_bn * 2 * CW : -1LL);
    DGDM_HIP_CHECK(hipMemcpyAsync(to_trainer ? m->run.p : (void *)flat_dev, to_trainer ? (const void *)flat_dev : m->run.p, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return DGDM_OK;
}

extern "C" int dgdm_trainer3d_export(DgdmTrainer3d *m, int which, DgdmTensor *tensors, int n_tensors) {
    DGDM_REQUIRE(m && tensors && which >= 0 && which <= 3, DGDM_EINVAL, "dgdm_trainer3d_export: bad argument");
    return m->copy_state(which, tensors, n_tensors, false);
}

extern "C" int64_t dgdm_trainer3d_steps(const DgdmTrainer3d *m) { return m ? m->bn_batches : -1; }

// test hook: intermediate tensors of the last call (device -> out_dev): 0 X0 [R][800], 1 l1p [R*512][128], 2 l2p [R*128][256], 3 feat1 [..][4],
// 4 y11 [..][64], 5 nx1 [R][512][3] (float), 6 fps1 (int32 bits), 7 idx1 (int32 bits), 8 dX0 [R][800], 9 pred [R][4]
extern "C" int dgdm_trainer3d_debug_read(DgdmTrainer3d *m, int which, void *out_dev, int64_t count, void *stream) {
    DGDM_REQUIRE(m && out_dev && count > 0, DGDM_EINVAL, "dgdm_trainer3d_debug_read: bad argument");
    const void *src[] = {m->X0, m->l1p, m->l2p, m->feat1, m->y11, m->nx1, m->fps1, m->idx1, m->dX0, m->pred};
    DGDM_REQUIRE(which >= 0 && which < 10, DGDM_EINVAL, "dgdm_trainer3d_debug_read: which");
    DGDM_HIP_CHECK(hipMemcpyAsync(out_dev, src[which], (size_t)count * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DGDM_OK;
}
