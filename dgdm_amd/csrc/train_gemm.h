// Shared pieces of the training paths (unet_train.hip, train3d.hip): the two float32-MFMA GEMM kernels over row WINDOWS, weight images
// and their inverse (weight-gradient scatter), ordered column sums, Adam.  Every kernel is internal to the translation unit that
// includes this header (anonymous namespace).  See unet_train.hip for the layout the windows come from.
#pragma once
#include "common.h"
#include "mfma_chain.h"

namespace dgdm {
namespace {

constexpr int TM = 128, TN = 128, KC = 16;
constexpr int64_t GUARD = 8192;        // floats in front of / behind every activation buffer: windows of the first / last rows stay inside

struct RowMask { int rp, pad, lv; };   // rows per sample, leading padding rows, valid rows;  rp == 0: every row is valid

__device__ __forceinline__ bool row_valid(const RowMask mk, int64_t m, int64_t M) {
    if (m >= M) return false;
    if (mk.rp == 0) return true;
    const int p = (int)(m % mk.rp);
    return p >= mk.pad && p < mk.pad + mk.lv;
}

// C[m][n] = sum_kk A(m)[kk] * B[kk][n] (+ bias[n]) (+ add[m][n]) on the valid rows, 0 on the others (not touched when accumulating)
struct RowGemm {
    const float *A; int64_t a_rs;        // row m's window starts at A + m * a_rs
    const float *B; int Kp, Np;          // weight image [Kp][Np], Kp a multiple of 16, Np of 128, zero beyond (K, N)
    float *C; int64_t c_rs; int N;
    const float *add; int64_t add_rs;
    const float *bias;
    int64_t M;
    RowMask mk;
    int scalar_a;                        // windows not 16-byte aligned (single-channel inputs): dword loads
};

__global__ __launch_bounds__(256, 2) void rowgemm_kernel(const RowGemm g) {
    __shared__ __attribute__((aligned(16))) float sA[KC][TM];
    __shared__ __attribute__((aligned(16))) float sB[KC][TN];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wi = w & 1, wj = w >> 1, n = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    const int arow = tid & 127, akq = tid >> 7, bx4 = tid & 31, bkr = tid >> 5;
    const int64_t am = min(m0 + arow, g.M - 1);
    const float *ap = g.A + am * g.a_rs + 4 * akq;
    const float *bp = g.B + (int64_t)bkr * g.Np + n0 + 4 * bx4;
    float4 ra[2], rb[2];
    auto issue = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float *p = ap + k0 + 8 * u;
            if (g.scalar_a) ra[u] = make_float4(p[0], p[1], p[2], p[3]);
            else ra[u] = *reinterpret_cast<const float4 *>(p);
            rb[u] = *reinterpret_cast<const float4 *>(bp + (int64_t)(k0 + 8 * u) * g.Np);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[s][u][q] = 0.f;
    issue(0);
    for (int k0 = 0; k0 < g.Kp; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kb = 4 * (akq + 2 * u);
            sA[kb + 0][arow] = ra[u].x; sA[kb + 1][arow] = ra[u].y; sA[kb + 2][arow] = ra[u].z; sA[kb + 3][arow] = ra[u].w;
            *reinterpret_cast<float4 *>(&sB[bkr + 8 * u][4 * bx4]) = rb[u];
        }
        __syncthreads();
        if (k0 + KC < g.Kp) issue(k0 + KC);
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) a[s] = sA[2 * kk + h][64 * wi + 32 * s + n];
#pragma unroll
            for (int u = 0; u < 2; ++u) b[u] = sB[2 * kk + h][64 * wj + 32 * u + n];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[s][u] = mfma32(a[s], b[u], acc[s][u]);
        }
    }
    // acc[s][u][q] of lane (n, h) = C[m0 + 64 wi + 32 s + rho(q, h)][n0 + 64 wj + 32 u + n]
    float bj[2];
    bool nv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = n0 + 64 * wj + 32 * u + n;
        nv[u] = j < g.N;
        bj[u] = g.bias && nv[u] ? g.bias[j] : 0.f;
    }
    const bool accumulate = g.add == g.C;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t m = m0 + 64 * wi + 32 * s + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (m >= g.M) continue;
            const bool valid = row_valid(g.mk, m, g.M);
            if (!valid && accumulate) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!nv[u]) continue;
                const int j = n0 + 64 * wj + 32 * u + n;
                float v = 0.f;
                if (valid) {
                    v = acc[s][u][q] + bj[u];
                    if (g.add) v += g.add[m * g.add_rs + j];
                }
                g.C[m * g.c_rs + j] = v;
            }
        }
}

// part[z][i][j] = sum over the rows m of split z of A(m)[i] * D(m)[j]      (i < 128 gridDim.x, j < 128 gridDim.y)
struct ColGemm {
    const float *A; int64_t a_rs;
    const float *D; int64_t d_rs;
    float *part; int64_t split_stride; int ldp;
    int64_t M, m_per_split;
    int scalar_a, scalar_d;
};

__global__ __launch_bounds__(256, 2) void colgemm_kernel(const ColGemm g) {
    __shared__ __attribute__((aligned(16))) float sP[KC][TM];
    __shared__ __attribute__((aligned(16))) float sQ[KC][TN];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wi = w & 1, wj = w >> 1, n = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * TM, j0 = blockIdx.y * TN;
    const int64_t mbeg = (int64_t)blockIdx.z * g.m_per_split, mend = min(g.M, mbeg + g.m_per_split);
    const int x4 = tid & 31, rr = tid >> 5;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rp[2], rq[2];
    auto issue = [&](int64_t m) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t r = m + rr + 8 * u;
            if (r < mend) {
                const float *pa = g.A + r * g.a_rs + i0 + 4 * x4, *pd = g.D + r * g.d_rs + j0 + 4 * x4;
                rp[u] = g.scalar_a ? make_float4(pa[0], pa[1], pa[2], pa[3]) : *reinterpret_cast<const float4 *>(pa);
                rq[u] = g.scalar_d ? make_float4(pd[0], pd[1], pd[2], pd[3]) : *reinterpret_cast<const float4 *>(pd);
            } else { rp[u] = zero4; rq[u] = zero4; }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[s][u][q] = 0.f;
    if (mbeg < mend) issue(mbeg);
    for (int64_t m = mbeg; m < mend; m += KC) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<float4 *>(&sP[rr + 8 * u][4 * x4]) = rp[u];
            *reinterpret_cast<float4 *>(&sQ[rr + 8 * u][4 * x4]) = rq[u];
        }
        __syncthreads();
        if (m + KC < mend) issue(m + KC);
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) a[s] = sP[2 * kk + h][64 * wi + 32 * s + n];
#pragma unroll
            for (int u = 0; u < 2; ++u) b[u] = sQ[2 * kk + h][64 * wj + 32 * u + n];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[s][u] = mfma32(a[s], b[u], acc[s][u]);
        }
    }
    float *dst = g.part + (int64_t)blockIdx.z * g.split_stride;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = i0 + 64 * wi + 32 * s + (q & 3) + 8 * (q >> 2) + 4 * h;
#pragma unroll
            for (int u = 0; u < 2; ++u) dst[(int64_t)i * g.ldp + j0 + 64 * wj + 32 * u + n] = acc[s][u][q];
        }
}

// One GEMM image of a weight tensor: image element (kk, n), kk = j * Kblk + r, is the tensor's element  src + r * s_kc + n * s_n + taps[j]
struct ImgDesc {
    int64_t src, dst;
    int Kblk, ntaps, taps[5];
    int K, Kp, N, Np, s_kc, s_n;
};
__device__ __forceinline__ int64_t img_src(const ImgDesc &d, int kk, int n) {
    const int j = kk / d.Kblk, r = kk - j * d.Kblk;
    return d.src + (int64_t)r * d.s_kc + (int64_t)n * d.s_n + d.taps[j];
}
__global__ void repack_kernel(const float *__restrict__ P, float *__restrict__ IMG, const ImgDesc *__restrict__ descs) {
    const ImgDesc d = descs[blockIdx.y];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)d.Kp * d.Np) return;
    const int kk = (int)(e / d.Np), n = (int)(e - (int64_t)kk * d.Np);
    IMG[d.dst + e] = kk < d.K && n < d.N ? P[img_src(d, kk, n)] : 0.f;
}
// weight-gradient partial tiles -> the tensor's own layout: G[src(kk, n)] = sum over the splits (fixed order)
__global__ void wgrad_scatter_kernel(const float *__restrict__ part, int splits, int64_t split_stride, int ldp, const ImgDesc *__restrict__ descs, int img,
                                     float *__restrict__ G) {
    const ImgDesc d = descs[img];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)d.K * d.N) return;
    const int kk = (int)(e / d.N), n = (int)(e - (int64_t)kk * d.N);
    const float *p = part + (int64_t)kk * ldp + n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 3 < splits; s += 4) {
        a0 += p[(int64_t)s * split_stride]; a1 += p[(int64_t)(s + 1) * split_stride];
        a2 += p[(int64_t)(s + 2) * split_stride]; a3 += p[(int64_t)(s + 3) * split_stride];
    }
    for (; s < splits; ++s) a0 += p[(int64_t)s * split_stride];
    G[img_src(d, kk, n)] = (a0 + a1) + (a2 + a3);
}

// column sums of a [M][N] matrix (row stride rs) over blocks of rows_per_block rows; then colsum_finish adds the blocks in float64
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ D, int64_t rs, int64_t M, int N, int64_t rows_per_block, float *__restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float a = 0.f;
    if (c < N)
        for (int64_t r = r0 + rl; r < r1; r += 4) a += D[r * rs + c];
    red[rl][threadIdx.x & 63] = a;
    __syncthreads();
    if (rl == 0 && c < N) part[(int64_t)blockIdx.x * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// out[c] = sum_t in[t][c]  (float64, fixed order): four interleaved partial sums per column - t = 0, 4, 8, ..., t = 1, 5, ..., ... over the
// first 4 floor(T / 4) rows, the remaining rows onto the first - folded as (a0 + a1) + (a2 + a3).  One thread per (column, partial sum):
// a workgroup covers 64 columns (the four sums of a column were one thread's four accumulators until round 5 - the same additions in
// the same order, bit for bit, on four times the threads and a grid of W / 64 instead of W / 256 workgroups: the serial walk over up
// to 2048 partial rows made this reduction 16 % of an eps-net training step).
__global__ __launch_bounds__(256) void rows_sum_kernel(const float *__restrict__ in, int64_t T, int W, float *__restrict__ out) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    double a = 0.0;
    if (c < W) {
        const int64_t T4 = T & ~(int64_t)3;
        int64_t t = q;
        for (; t + 12 < T4; t += 16) {                       // four loads in flight, added in order
            const float x0 = in[t * W + c], x1 = in[(t + 4) * W + c], x2 = in[(t + 8) * W + c], x3 = in[(t + 12) * W + c];
            a += (double)x0; a += (double)x1; a += (double)x2; a += (double)x3;
        }
        for (; t < T4; t += 4) a += (double)in[t * W + c];
        if (q == 0) for (t = T4; t < T; ++t) a += (double)in[t * W + c];
    }
    red[q][threadIdx.x & 63] = a;
    __syncthreads();
    if (q == 0 && c < W) out[c] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}
inline dim3 rows_sum_grid(int W) { return dim3((unsigned)((W + 63) / 64)); }


// torch.optim.Adam (single-tensor form: lerp first moment, bias corrections on the host), as train2d.hip
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, int64_t n, float b1, float b2,
                            float eps, float wd, float step_size, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    if (wd != 0.f) gi = fmaf(wd, p[i], gi);
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}
// dst (+)= src
__global__ void add_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t n, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = accumulate ? dst[i] + src[i] : src[i];
}
__global__ void scale_kernel(float *__restrict__ g, int64_t n, float f) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] *= f;
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }


}  // namespace
}  // namespace dgdm
