// Trainer.step of the 2-D dynamics model (dynamics/trainer.py:53-103 with ProfileForward2DModel, dynamics/profile_forward_2d.py:78-156)
// on gfx950: forward with BatchNorm1d in TRAINING mode (batch statistics, running statistics updated), MSE loss, backward with
// weight gradients, torch.optim.Adam (trainer.py:46).  SURVEY.md §8(f) rank 4.
//
// All fourteen Linear layers run through one LDS-tiled float32 MFMA GEMM (v_mfma_f32_32x32x2_f32, 128 x 256 tile per 256-thread
// workgroup, 16-deep contraction chunks):
//     C[i][j] = sum_r P[i][r] * Q[j][r]
//   forward      Y  [n][m] = sum_k a[n][k]  Wt[k][m]     i = n (rows),  j = m,  r = k
//   input grad   G  [n][k] = sum_m dY[n][m] W [m][k]     i = n,         j = k,  r = m
//   weight grad  dW [k][m] = sum_n a[n][k]  dY[n][m]     i = k,         j = m,  r = n   (split over n, deterministic two-stage sum)
// The row count of a training batch is large (batch x orientations x positions), the layers are 256 wide: with the float32 MFMA rate
// the GEMMs are within 3x of HBM-bound, so nothing elementwise gets its own pass over the activations.  What the reference's autograd
// graph does in separate kernels is folded into the operand loaders and the epilogues:
//   * BatchNorm + ReLU of the previous layer is applied while its pre-normalisation output Y is loaded as an operand
//     (a = max(0, sc*Y + sh); Y is the only activation tensor kept per layer),
//   * the forward epilogue adds the bias and produces per-column partial sums (sum y, sum y^2) for the batch statistics,
//   * the input-gradient epilogue applies the activation mask of the layer below and produces the two column sums BatchNorm's
//     backward needs (sum dZ, sum dZ*Y),
//   * BatchNorm's backward is an affine map per column, dY = alpha*dZ + beta'*Y + gamma', applied while dZ and Y are loaded as the
//     operand of the next two GEMMs; Linear-bias gradients are column sums taken by the weight-gradient loader.
// Column statistics are summed per workgroup in float32 and across workgroups in float64, in a fixed order: a step is reproducible
// bit for bit.
#include "common.h"
#include "mfma_chain.h"
#include "smallnet.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <memory>
#include <map>
#include <string>
#include <type_traits>

namespace dgdm {
namespace {

#ifndef DGDM_TRAIN_RC
#define DGDM_TRAIN_RC 16
#endif
constexpr int TI = 128, TJ = 256, RC = DGDM_TRAIN_RC, PU = RC / 8, QU = RC / 4, W = 256;      // RC = contraction depth of one LDS chunk
enum { EPI_FWD = 0, EPI_BWD = 1, EPI_WGRAD = 2 };
enum { MASK_NONE = 0, MASK_RELU = 1, MASK_SILU = 2, MASK_RELU_BN = 3 };

// A GEMM operand: T0 (and T1) row-major with leading dimension ld, turned into the operand value on load (X_* below) with the
// per-column coefficient vectors c0, c1, c2.
struct Operand {
    const float *t0, *t1;
    int64_t ld;
    const float *c0, *c1, *c2;
    int xf;
};

struct GemmArgs {
    Operand P, Q;
    int64_t I, R, r_per_split;
    int J;
    float *C;
    int64_t ldc, split_stride;
    const float *bias;      // EPI_FWD
    float *stats;           // [i tiles][2][J] partial column sums, or null
    const float *Yp;        // EPI_BWD: pre-activation output of the layer below, [I][ldy]
    int64_t ldy;
    const float *m0, *m2;   // MASK_RELU_BN: z = m0*Yp + m2
    int mask;
    int64_t p_bytes, q_bytes, c_bytes, y_bytes;     // extents of the tensors behind P, Q, C, Yp (set by the launcher)
};

__device__ __forceinline__ float silu(float z) { return z / (1.f + expf(-z)); }
__device__ __forceinline__ float silu_grad(float z) {
    const float s = 1.f / (1.f + expf(-z));
    return s * (1.f + z * (1.f - s));
}
__device__ __forceinline__ float bn_pre(float c0, float y, float c2) { return fmaf(c0, y, c2); }   // ONE expression for forward and mask

// Operand transforms are compile-time (straight-line loaders: every global load of a chunk is issued back to back, invalid rows
// and columns read a clamped address and are zeroed by a select when the value is written to LDS).
//   PX / QX:  X_PLAIN  T0 | X_RELU  max(0, c0*T0 + c2) | X_SILU  silu(T0) | X_AFF2  c0*T0 + c1*T1 + c2
enum { X_PLAIN = 0, X_RELU = 1, X_SILU = 2, X_AFF2 = 3 };

template <int X>
__device__ __forceinline__ float xf1(float a, float b, float c0, float c1, float c2) {
    if (X == X_PLAIN) return a;
    if (X == X_RELU) return fmaxf(bn_pre(c0, a, c2), 0.f);
    if (X == X_SILU) return silu(a);
    return fmaf(c0, a, fmaf(c1, b, c2));
}
template <int X>
__device__ __forceinline__ float4 xf4(float4 a, float4 b, float4 c0, float4 c1, float4 c2, bool valid) {
    float4 v = make_float4(xf1<X>(a.x, b.x, c0.x, c1.x, c2.x), xf1<X>(a.y, b.y, c0.y, c1.y, c2.y), xf1<X>(a.z, b.z, c0.z, c1.z, c2.z),
                           xf1<X>(a.w, b.w, c0.w, c1.w, c2.w));
    if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Memory access of tgemm_kernel goes through buffer descriptors: the float32 MFMA runs at the vector-ALU rate and does not overlap
// with other VALU work (scripts/micro/mfma_dep.hip: every filler instruction beside v_mfma_f32_32x32x2_f32 adds its full 4-16
// cycles), so every 64-bit address computation, bounds compare and select in the loop is time taken from the matrix pipe.  With a
// descriptor the tile / chunk offset moves the BASE (scalar ALU, a window of the tensor from there to its end), the thread's part
// is one constant 32-bit VGPR offset, and the hardware's range check returns 0 for (drops stores to) anything past the tensor's end.
__device__ void llvm_amdgcn_raw_buffer_store_f32(float data, wrsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ float llvm_amdgcn_raw_buffer_load_f32(wrsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
constexpr int OOB = (int)0x80000000;       // a voffset no window reaches

// window of the tensor at `base` ([.., bytes)) starting `off` bytes in; base, off, bytes wave-uniform
__device__ __forceinline__ wrsrc_t window(const float *base, int64_t off, int64_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base) + (uint64_t)off;
    int64_t rem = bytes - off;
    rem = rem < 0 ? 0 : (rem > 0x7fffffffLL ? 0x7fffffffLL : rem);
    wrsrc_t rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
    rs.z = __builtin_amdgcn_readfirstlane((int)rem);
    rs.w = 0x00020000;
    return rs;
}
__device__ __forceinline__ float4 bload4(wrsrc_t rs, int voff) {
    const v4f32 v = llvm_amdgcn_raw_buffer_load_v4f32(rs, voff, 0, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}

template <bool PTRANS, int EPI, int PX, int QX, int MASK>
__global__ __launch_bounds__(256, 2) void tgemm_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float sP[RC][TI];
    __shared__ __attribute__((aligned(16))) float sQ[RC][TJ];
    constexpr bool PC = PX == X_RELU || PX == X_AFF2, QC = QX == X_RELU || QX == X_AFF2;      // operand has per-column coefficients
    __shared__ __attribute__((aligned(16))) float sC[PTRANS && PC ? 3 : 1][PTRANS && PC ? W : 4];   // P's coefficients (transposed source: R <= 256 columns)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wi = w & 1, wj = w >> 1, n = lane & 31, h = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * TI;
    const int j0 = blockIdx.y * TJ;
    const int64_t rbeg = (int64_t)blockIdx.z * g.r_per_split, rend = min(g.R, rbeg + g.r_per_split);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PTRANS && PC) {
        const bool in = tid < g.R;
        sC[0][tid] = in ? g.P.c0[tid] : 0.f;
        sC[2][tid] = in ? g.P.c2[tid] : 0.f;
        sC[1][tid] = in && PX == X_AFF2 ? g.P.c1[tid] : 0.f;
    }

    // ---- loaders.  P transposed source: thread = (tile row px, float4 slots prq, prq + 2 along r); P / Q direct source:
    // thread = (float4 column, rows prr + 8u / qrr + 4u of the chunk).  Per-thread byte offsets inside the chunk's window:
    const int px = tid & 127, prq = tid >> 7, px4 = tid & 31, prr = tid >> 5, qx4 = tid & 63, qrr = tid >> 6;
    const int pld4 = (int)g.P.ld * 4, qld4 = (int)g.Q.ld * 4;
    const bool pcolv = PTRANS || i0 + 4 * px4 < g.I, qcolv = j0 + 4 * qx4 < g.J;
    int pvo[PU], qvo[QU];
#pragma unroll
    for (int u = 0; u < PU; ++u) pvo[u] = !pcolv ? OOB : (PTRANS ? px * pld4 + 16 * (prq + 2 * u) : (prr + 8 * u) * pld4 + 16 * px4);
#pragma unroll
    for (int u = 0; u < QU; ++u) qvo[u] = !qcolv ? OOB : (qrr + 4 * u) * qld4 + 16 * qx4;
    float4 pc0 = zero4, pc1 = zero4, pc2 = zero4, qc0 = zero4, qc1 = zero4, qc2 = zero4;
    if (!PTRANS && PC) { const int64_t c = pcolv ? i0 + 4 * px4 : 0; pc0 = ld4(g.P.c0 + c); pc2 = ld4(g.P.c2 + c); if (PX == X_AFF2) pc1 = ld4(g.P.c1 + c); }
    if (QC) { const int c = qcolv ? j0 + 4 * qx4 : 0; qc0 = ld4(g.Q.c0 + c); qc2 = ld4(g.Q.c2 + c); if (QX == X_AFF2) qc1 = ld4(g.Q.c1 + c); }
    float4 pa[PU], pb[PU], qa[QU], qb[QU];
    float4 qsum = zero4;

    auto issue = [&](int64_t r) {
        const int64_t poff = (PTRANS ? i0 * g.P.ld + r : r * g.P.ld + i0) * 4, qoff = (r * g.Q.ld + j0) * 4;
        const wrsrc_t rp0 = window(g.P.t0, poff, g.p_bytes), rq0 = window(g.Q.t0, qoff, g.q_bytes);
#pragma unroll
        for (int u = 0; u < PU; ++u) pa[u] = bload4(rp0, pvo[u]);
        if (PX == X_AFF2) {
            const wrsrc_t rp1 = window(g.P.t1, poff, g.p_bytes);
#pragma unroll
            for (int u = 0; u < PU; ++u) pb[u] = bload4(rp1, pvo[u]);
        }
#pragma unroll
        for (int u = 0; u < QU; ++u) qa[u] = bload4(rq0, qvo[u]);
        if (QX == X_AFF2) {
            const wrsrc_t rq1 = window(g.Q.t1, qoff, g.q_bytes);
#pragma unroll
            for (int u = 0; u < QU; ++u) qb[u] = bload4(rq1, qvo[u]);
        }
    };
    // Rows past the end of a contraction over rows (weight gradient: the last chunk of the last split) read as 0 but do not TRANSFORM
    // to 0; `tail` = that chunk: there the Q rows are zeroed by hand.  Elsewhere (forward / input gradient) rows past the tensor's end
    // only feed output rows that are never stored.
    const bool bias_sums = EPI == EPI_WGRAD && blockIdx.x == 0;      // only the first i tile of a split takes the column sums of Q
    auto commit = [&](int64_t r, auto tail_c) {
        constexpr bool tail = decltype(tail_c)::value;
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            if (PTRANS) {
                const int r4 = prq + 2 * u;
                if (PC) { const int c = (int)r + 4 * r4; pc0 = ld4(&sC[0][c]); pc2 = ld4(&sC[2][c]); if (PX == X_AFF2) pc1 = ld4(&sC[1][c]); }
                const float4 v = xf4<PX>(pa[u], pb[u], pc0, pc1, pc2, true);
                sP[4 * r4 + 0][px] = v.x; sP[4 * r4 + 1][px] = v.y; sP[4 * r4 + 2][px] = v.z; sP[4 * r4 + 3][px] = v.w;
            } else {
                *reinterpret_cast<float4 *>(&sP[prr + 8 * u][4 * px4]) = xf4<PX>(pa[u], pb[u], pc0, pc1, pc2, true);
            }
        }
#pragma unroll
        for (int u = 0; u < QU; ++u) {
            float4 v = xf4<QX>(qa[u], qb[u], qc0, qc1, qc2, true);
            if (EPI == EPI_WGRAD && tail && (QX != X_PLAIN || PX != X_PLAIN) && r + qrr + 4 * u >= rend) v = zero4;
            *reinterpret_cast<float4 *>(&sQ[qrr + 4 * u][4 * qx4]) = v;
            if (bias_sums) { qsum.x += v.x; qsum.y += v.y; qsum.z += v.z; qsum.w += v.w; }
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[s][u][q] = 0.f;

    if (rbeg < rend) issue(rbeg);
    for (int64_t r = rbeg; r < rend; r += RC) {
        __syncthreads();
        if (r + RC > rend) commit(r, std::true_type{}); else commit(r, std::false_type{});
        __syncthreads();
        if (r + RC < rend) issue(r + RC);
#pragma unroll
        for (int kk = 0; kk < RC / 2; ++kk) {
            float a[2], b[4];
#pragma unroll
            for (int s = 0; s < 2; ++s) a[s] = sP[2 * kk + h][64 * wi + 32 * s + n];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = sQ[2 * kk + h][128 * wj + 32 * u + n];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[s][u] = mfma32(a[s], b[u], acc[s][u]);
        }
    }

    // ---- epilogue.  acc[s][u][q] of lane (n, h) = C[i0 + 64 wi + 32 s + rho(q, h)][j0 + 128 wj + 32 u + n]
    if (EPI == EPI_WGRAD) {
        float *dst = g.C + (int64_t)blockIdx.z * g.split_stride;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t i = i0 + 64 * wi + 32 * s + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (i < g.I)
#pragma unroll
                    for (int u = 0; u < 4; ++u) dst[i * g.ldc + j0 + 128 * wj + 32 * u + n] = acc[s][u][q];
            }
        if (blockIdx.x == 0) {      // column sums of Q = the Linear-bias gradient, stored as row I of the partial
            __syncthreads();
            *reinterpret_cast<float4 *>(&sQ[qrr][4 * qx4]) = qsum;
            __syncthreads();
            if (tid < 64) {
                float4 t = zero4;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float4 v = *reinterpret_cast<const float4 *>(&sQ[k][4 * tid]); t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
                if (j0 + 4 * tid < g.J) *reinterpret_cast<float4 *>(dst + g.I * g.ldc + j0 + 4 * tid) = t;
            }
        }
        return;
    }
    // forward / input gradient: the wave's 64 x 128 block, one output row (two half-rows of 32 lanes) per descriptor window
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float bj[4], m0[4], m2[4];
    const bool colv = j0 + 128 * wj + n + 96 < g.J || g.J % TJ == 0;       // J is a multiple of 256 for every caller; kept general
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + 128 * wj + 32 * u + n, g.J - 1);
        bj[u] = EPI == EPI_FWD ? g.bias[j] : 0.f;
        m0[u] = MASK == MASK_RELU_BN ? g.m0[j] : 1.f;
        m2[u] = MASK == MASK_RELU_BN ? g.m2[j] : 0.f;
    }
    const int ldc4 = (int)g.ldc * 4, ldy4 = (int)g.ldy * 4;
    const int cvo = colv ? 4 * h * ldc4 + (128 * wj + n) * 4 : OOB, yvo = colv ? 4 * h * ldy4 + (128 * wj + n) * 4 : OOB;
    // FULL: every row of the tile exists - no per-row predicate on the column statistics (all but the last row tile)
    auto body = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) {
            // half of a 32-row group at a time: its 32 loads of the layer below are issued together, then consumed
            const int s = sq >> 1, qb = (sq & 1) * 8;
            float y[8][4];
            if (EPI == EPI_BWD && MASK != MASK_NONE) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int64_t row = i0 + 64 * wi + 32 * s + ((qb + q) & 3) + 8 * ((qb + q) >> 2);       // + 4h in the lane's offset
                    const wrsrc_t ry = window(g.Yp, (row * g.ldy + j0) * 4, g.y_bytes);
#pragma unroll
                    for (int u = 0; u < 4; ++u) y[q][u] = llvm_amdgcn_raw_buffer_load_f32(ry, yvo + 128 * u, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int64_t row = i0 + 64 * wi + 32 * s + ((qb + q) & 3) + 8 * ((qb + q) >> 2);
                const wrsrc_t rc = window(g.C, (row * g.ldc + j0) * 4, g.c_bytes);
                const bool rowv = FULL || row + 4 * h < g.I;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a = s == 0 ? acc[0][u][qb + q] : acc[1][u][qb + q];
                    float v;
                    if (EPI == EPI_FWD) {
                        v = a + bj[u];
                        if (rowv) { s1[u] += v; s2[u] = fmaf(v, v, s2[u]); }
                    } else {
                        v = a;
                        if (MASK == MASK_RELU) v = y[q][u] > 0.f ? v : 0.f;
                        else if (MASK == MASK_SILU) v *= silu_grad(y[q][u]);
                        else if (MASK == MASK_RELU_BN) {
                            v = bn_pre(m0[u], y[q][u], m2[u]) > 0.f ? v : 0.f;
                            if (rowv) { s1[u] += v; s2[u] = fmaf(v, y[q][u], s2[u]); }
                        }
                    }
                    llvm_amdgcn_raw_buffer_store_f32(v, rc, cvo + 128 * u, 0, 0);
                }
            }
            if (EPI == EPI_BWD && MASK != MASK_NONE) __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (i0 + TI <= g.I) body(std::true_type{}); else body(std::false_type{});
    if (g.stats) {       // per-workgroup column sums: both half-waves, then the two waves stacked along i
        __syncthreads();
        float *red = &sP[0][0];     // [wi][2][256]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float a = s1[u] + __shfl_xor(s1[u], 32), b = s2[u] + __shfl_xor(s2[u], 32);
            if (h == 0) { red[(wi * 2 + 0) * 256 + 128 * wj + 32 * u + n] = a; red[(wi * 2 + 1) * 256 + 128 * wj + 32 * u + n] = b; }
        }
        __syncthreads();
        if (j0 + tid < g.J) {
            g.stats[((int64_t)blockIdx.x * 2 + 0) * g.J + j0 + tid] = red[0 * 256 + tid] + red[2 * 256 + tid];
            g.stats[((int64_t)blockIdx.x * 2 + 1) * g.J + j0 + tid] = red[1 * 256 + tid] + red[3 * 256 + tid];
        }
    }
}

// column sums of per-workgroup partials in float64: out[s][c] = sum over t = s, s + S, ... of in[t][c]   (fixed order: reproducible)
constexpr int RED_S = 64;
__global__ void reduce_partials_kernel(const float *__restrict__ in, int T, int ncol, double *__restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (c >= ncol) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int t = s;
    for (; t + 3 * RED_S < T; t += 4 * RED_S) {
        a0 += (double)in[(int64_t)t * ncol + c]; a1 += (double)in[(int64_t)(t + RED_S) * ncol + c];
        a2 += (double)in[(int64_t)(t + 2 * RED_S) * ncol + c]; a3 += (double)in[(int64_t)(t + 3 * RED_S) * ncol + c];
    }
    for (; t < T; t += RED_S) a0 += (double)in[(int64_t)t * ncol + c];
    out[(int64_t)s * ncol + c] = (a0 + a1) + (a2 + a3);
}

// Batch statistics of one BatchNorm1d layer from the forward partials (float64 across workgroups), the folded coefficients for the
// next layer's loader, and the running statistics (momentum 0.1, unbiased variance: torch.nn.BatchNorm1d in training mode).
// one block of 256 threads (thread = column) over the RED_S float64 partial rows reduce_partials_kernel left
// coef rows: 0 sc = gamma*rstd, 1 sh = beta - mean*sc, 2 mean, 3 rstd, 4 alpha, 5 beta', 6 gamma' (backward, bn_bwd_finalize)
__global__ void bn_fwd_finalize_kernel(const double *__restrict__ part /*[RED_S / 2][2][256]*/, double N, const float *__restrict__ gamma,
                                       const float *__restrict__ beta, float eps, float mom, float *__restrict__ rmean, float *__restrict__ rvar,
                                       float *__restrict__ coef) {
    const int j = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < RED_S / 2; ++k) { a += part[(k * 2 + 0) * W + j]; b += part[(k * 2 + 1) * W + j]; }
    const double mu = a / N, var = fmax(b / N - mu * mu, 0.0);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), sc = gamma[j] * rstd;
    coef[0 * W + j] = sc;
    coef[1 * W + j] = beta[j] - (float)mu * sc;
    coef[2 * W + j] = (float)mu;
    coef[3 * W + j] = rstd;
    rmean[j] = (1.f - mom) * rmean[j] + mom * (float)mu;
    rvar[j] = (1.f - mom) * rvar[j] + mom * (float)(var * N / (N - 1.0));
}

// eval mode (Trainer.inference, trainer.py:108): the same two loader coefficients from the running statistics
__global__ void bn_eval_coef_kernel(const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ rmean,
                                    const float *__restrict__ rvar, float eps, float *__restrict__ coef) {
    const int j = threadIdx.x;
    const float rstd = 1.f / sqrtf(rvar[j] + eps), sc = gamma[j] * rstd;
    coef[0 * W + j] = sc;
    coef[1 * W + j] = beta[j] - rmean[j] * sc;
    coef[2 * W + j] = rmean[j];
    coef[3 * W + j] = rstd;
}

// BatchNorm backward per column from the partial sums (sum dZ, sum dZ*Y):  with xhat = (Y - mean)*rstd,
//   dgamma = sum dZ*xhat, dbeta = sum dZ,  dY = sc*(dZ - mean(dZ) - xhat*mean(dZ*xhat)) = alpha*dZ + beta'*Y + gamma'
__global__ void bn_bwd_finalize_kernel(const double *__restrict__ part /*[RED_S / 2][2][256]*/, double N, float *__restrict__ coef,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int j = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < RED_S / 2; ++k) { a += part[(k * 2 + 0) * W + j]; b += part[(k * 2 + 1) * W + j]; }
    const double sc = coef[0 * W + j], mu = coef[2 * W + j], rstd = coef[3 * W + j];
    const double sxh = rstd * (b - mu * a);                 // sum dZ*xhat
    const double c1 = a / N, c2 = sxh / N;
    const double bp = -sc * c2 * rstd;
    coef[4 * W + j] = (float)sc;
    coef[5 * W + j] = (float)bp;
    coef[6 * W + j] = (float)(-sc * c1 - bp * mu);
    dgamma[j] = (float)sxh;
    dbeta[j] = (float)a;
}

// Output layer + loss + its backward in one pass over Y8 (profile_forward_2d.py:155, trainer.py:96-100):
//   a = relu(sc*Y8 + sh); pred = a Wout^T + bout; loss = mean (pred - score)^2; dpred = 2 (pred - score) / (3N);
//   G = dpred Wout, dZ8 = G where a > 0 (+ column partial sums for BatchNorm backward), dWout = dpred^T a, dbout = sum dpred.
// One wave per row at a time (lane = 4 columns), 64 rows per workgroup.
constexpr int HEAD_ROWS = 64;
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ Y, const float *__restrict__ coef, const float *__restrict__ Wout,
                                                   const float *__restrict__ bout, const float *__restrict__ score, int64_t N, float *__restrict__ pred,
                                                   float *__restrict__ dZ, float *__restrict__ stats /*[T][2][256]*/,
                                                   float *__restrict__ hpart /*[T][3*256 + 4]*/, int train, double Ntot) {
    __shared__ __attribute__((aligned(16))) float red[4][5 * 256 + 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = 4 * lane;
    const float4 sc = ld4(coef + c), sh = ld4(coef + W + c), w0 = ld4(Wout + c), w1 = ld4(Wout + W + c), w2 = ld4(Wout + 2 * W + c);
    const float b0 = bout[0], b1 = bout[1], b2 = bout[2];
    const float inv = (float)(2.0 / (3.0 * Ntot));      // d mean((pred - score)^2) / d pred over ALL rows of the batch (every rank's)
    float4 sdz = make_float4(0, 0, 0, 0), sdzy = sdz, g0 = sdz, g1 = sdz, g2 = sdz;
    float db0 = 0.f, db1 = 0.f, db2 = 0.f, lsum = 0.f;
    const int64_t rb = (int64_t)blockIdx.x * HEAD_ROWS;
    for (int k = w; k < HEAD_ROWS; k += 4) {
        const int64_t r = rb + k;
        if (r >= N) break;
        const float4 y = ld4(Y + r * W + c);
        const float4 a = make_float4(fmaxf(bn_pre(sc.x, y.x, sh.x), 0.f), fmaxf(bn_pre(sc.y, y.y, sh.y), 0.f), fmaxf(bn_pre(sc.z, y.z, sh.z), 0.f),
                                     fmaxf(bn_pre(sc.w, y.w, sh.w), 0.f));
        float p0 = a.x * w0.x + a.y * w0.y + a.z * w0.z + a.w * w0.w, p1 = a.x * w1.x + a.y * w1.y + a.z * w1.z + a.w * w1.w,
              p2 = a.x * w2.x + a.y * w2.y + a.z * w2.z + a.w * w2.w;
#pragma unroll
        for (int m = 32; m; m >>= 1) { p0 += __shfl_xor(p0, m); p1 += __shfl_xor(p1, m); p2 += __shfl_xor(p2, m); }
        p0 += b0; p1 += b1; p2 += b2;
        const float e0 = p0 - score[r * 3 + 0], e1 = p1 - score[r * 3 + 1], e2 = p2 - score[r * 3 + 2];
        if (lane == 0) { pred[r * 3 + 0] = p0; pred[r * 3 + 1] = p1; pred[r * 3 + 2] = p2; lsum += e0 * e0 + e1 * e1 + e2 * e2; }
        if (!train) continue;
        const float d0 = e0 * inv, d1 = e1 * inv, d2 = e2 * inv;
        float4 gz = make_float4(d0 * w0.x + d1 * w1.x + d2 * w2.x, d0 * w0.y + d1 * w1.y + d2 * w2.y, d0 * w0.z + d1 * w1.z + d2 * w2.z,
                                d0 * w0.w + d1 * w1.w + d2 * w2.w);
        gz.x = a.x > 0.f ? gz.x : 0.f; gz.y = a.y > 0.f ? gz.y : 0.f; gz.z = a.z > 0.f ? gz.z : 0.f; gz.w = a.w > 0.f ? gz.w : 0.f;
        *reinterpret_cast<float4 *>(dZ + r * W + c) = gz;
        sdz.x += gz.x; sdz.y += gz.y; sdz.z += gz.z; sdz.w += gz.w;
        sdzy.x = fmaf(gz.x, y.x, sdzy.x); sdzy.y = fmaf(gz.y, y.y, sdzy.y); sdzy.z = fmaf(gz.z, y.z, sdzy.z); sdzy.w = fmaf(gz.w, y.w, sdzy.w);
        g0.x = fmaf(d0, a.x, g0.x); g0.y = fmaf(d0, a.y, g0.y); g0.z = fmaf(d0, a.z, g0.z); g0.w = fmaf(d0, a.w, g0.w);
        g1.x = fmaf(d1, a.x, g1.x); g1.y = fmaf(d1, a.y, g1.y); g1.z = fmaf(d1, a.z, g1.z); g1.w = fmaf(d1, a.w, g1.w);
        g2.x = fmaf(d2, a.x, g2.x); g2.y = fmaf(d2, a.y, g2.y); g2.z = fmaf(d2, a.z, g2.z); g2.w = fmaf(d2, a.w, g2.w);
        db0 += d0; db1 += d1; db2 += d2;
    }
    float *mine = red[w];
    *reinterpret_cast<float4 *>(mine + c) = sdz;
    *reinterpret_cast<float4 *>(mine + 256 + c) = sdzy;
    *reinterpret_cast<float4 *>(mine + 512 + c) = g0;
    *reinterpret_cast<float4 *>(mine + 768 + c) = g1;
    *reinterpret_cast<float4 *>(mine + 1024 + c) = g2;
    if (lane == 0) { mine[1280] = db0; mine[1281] = db1; mine[1282] = db2; mine[1283] = lsum; }
    __syncthreads();
    const int t = threadIdx.x;
    auto col = [&](int o) { return red[0][o] + red[1][o] + red[2][o] + red[3][o]; };
    stats[((int64_t)blockIdx.x * 2 + 0) * W + t] = col(t);
    stats[((int64_t)blockIdx.x * 2 + 1) * W + t] = col(256 + t);
    float *hp = hpart + (int64_t)blockIdx.x * (3 * W + 4);
    hp[t] = col(512 + t); hp[W + t] = col(768 + t); hp[2 * W + t] = col(1024 + t);
    if (t < 4) hp[3 * W + t] = col(1280 + t);
}

// sums the head partials: gWout [3][256], gbout [3], loss (mean squared error)
__global__ void head_finalize_kernel(const double *__restrict__ part /*[RED_S][3*256 + 4]*/, double N, float *__restrict__ gW, float *__restrict__ gb,
                                     float *__restrict__ loss) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * W + 4) return;
    double a = 0.0;
    for (int k = 0; k < RED_S; ++k) a += part[k * (3 * W + 4) + e];
    if (e < 3 * W) { if (gW) gW[e] = (float)a; }
    else if (e < 3 * W + 3) { if (gb) gb[e - 3 * W] = (float)a; }
    else *loss = (float)(a / (3.0 * N));
}

// weight-gradient partials [split][I + 1][256] (row k, column m; row I = bias) -> gW [256][I] (the Linear's own layout), gb [256]
__global__ void wgrad_reduce_kernel(const float *__restrict__ part, int splits, int64_t stride, int I, float *__restrict__ gW, float *__restrict__ gb) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (I + 1) * W) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 3 < splits; s += 4) {
        a0 += part[(int64_t)s * stride + e]; a1 += part[(int64_t)(s + 1) * stride + e];
        a2 += part[(int64_t)(s + 2) * stride + e]; a3 += part[(int64_t)(s + 3) * stride + e];
    }
    for (; s < splits; ++s) a0 += part[(int64_t)s * stride + e];
    const float a = (a0 + a1) + (a2 + a3);
    const int k = e / W, m = e - k * W;
    if (k < I) gW[(int64_t)m * I + k] = a;
    else gb[m] = a;
}

// torch.optim.Adam, single-tensor form (lerp for the first moment, bias corrections on the host)
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

struct TrDesc { int64_t w, wt; int K; };      // offsets of W [256][K] in the parameter buffer and of Wt [K][256] in the transposed one
__global__ void transpose_all_kernel(const float *__restrict__ P, float *__restrict__ WT, const TrDesc *__restrict__ d) {
    const TrDesc t = d[blockIdx.y];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;        // index into Wt: k * 256 + m
    if (e >= t.K * W) return;
    const int k = e / W, m = e - k * W;
    WT[t.wt + e] = P[t.w + (int64_t)m * t.K + k];
}

// ---- exact de-duplication of the row-invariant encoders (the same idea as the guided path's tables, DESIGN_HISTORY.md 4.1): the time
// encoder sees num_train_timesteps distinct inputs and the object encoder one input per sample, whatever the number of rows.
// dst[n][0..256) (row stride ldd) = src[group(n)][0..256),  group(n) = idx[n] or n / run
__global__ void expand_groups_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, int run, float *__restrict__ dst, int64_t ldd, int64_t N) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one float4
    if (e >= N * 64) return;
    const int64_t n = e >> 6;
    const int c = (int)(e & 63) * 4;
    const int64_t gsrc = idx ? idx[n] : n / run;
    *reinterpret_cast<float4 *>(dst + n * ldd + c) = ld4(src + gsrc * W + c);
}
// gradient of a grouped encoder output: part[block][group][256] = sum over the block's 256 rows with that group index (rows in order,
// one thread per column: a fixed summation order); groups <= 32 (LDS)
constexpr int SEG_ROWS = 256, SEG_GROUPS = 32;
__global__ __launch_bounds__(256) void segsum_index_kernel(const float *__restrict__ G, int64_t ldg, const int32_t *__restrict__ idx, int ngroups, int64_t N,
                                                           float *__restrict__ part) {
    __shared__ float acc[SEG_GROUPS][W];
    const int c = threadIdx.x;
    for (int k = 0; k < ngroups; ++k) acc[k][c] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * SEG_ROWS, r1 = min(N, r0 + SEG_ROWS);
    for (int64_t r = r0; r < r1; ++r) acc[idx[r]][c] += G[r * ldg + c];
    for (int k = 0; k < ngroups; ++k) part[((int64_t)blockIdx.x * ngroups + k) * W + c] = acc[k][c];
}
// the same for groups that are runs of `run` consecutive rows: grid (groups, chunks of 256 rows of a run)
__global__ __launch_bounds__(256) void segsum_run_kernel(const float *__restrict__ G, int64_t ldg, int run, float *__restrict__ part /*[groups][chunks][256]*/) {
    const int c = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * run + (int64_t)blockIdx.y * SEG_ROWS, r1 = min((int64_t)(blockIdx.x + 1) * run, r0 + SEG_ROWS);
    float a = 0.f;
    for (int64_t r = r0; r < r1; ++r) a += G[r * ldg + c];
    part[((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * W + c] = a;
}
// out[g][c] = sum over k < nk of in[(g * nk + k) * stride_k ...]: double accumulation, fixed order
__global__ void sum_chunks_kernel(const float *__restrict__ in, int nk, float *__restrict__ out, int64_t n /* groups * 256 */) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int64_t g = e / W, c = e % W;
    double a = 0.0;
    for (int k = 0; k < nk; ++k) a += (double)in[(g * nk + k) * W + c];
    out[e] = (float)a;
}
__global__ void sum_red_kernel(const double *__restrict__ red, int ncol, float *__restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    double a = 0.0;
    for (int k = 0; k < RED_S; ++k) a += red[(int64_t)k * ncol + c];
    out[c] = (float)a;
}

// noisy control points (DDIMScheduler.add_noise per row), zero-padded copies of the inputs, pose embedding into columns 768.. of X0
__global__ void prep2d_kernel(const float *__restrict__ ctrl, const float *__restrict__ noise, const float *__restrict__ sa, const float *__restrict__ sb,
                              const float *__restrict__ ori, const float *__restrict__ pos, const float *__restrict__ obj, int64_t N, int L, int Lp,
                              int OC, int OCp, float *__restrict__ bufC, float *__restrict__ bufO, float *__restrict__ X0, int obj_run, int64_t obj_rows) {
    const int S = Lp + OCp + 32;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * S) return;
    const int64_t r = e / S;
    int s = (int)(e - r * S);
    if (s < Lp) {
        bufC[r * Lp + s] = s < L ? (noise ? add_rn(mul_rn(sa[r], ctrl[r * L + s]), mul_rn(sb[r], noise[r * L + s])) : ctrl[r * L + s]) : 0.f;
        return;
    }
    s -= Lp;
    if (s < OCp) { if (r < obj_rows) bufO[r * OCp + s] = s < OC ? obj[r * obj_run * OC + s] : 0.f; return; }      // grouped: row r = the first row of run r
    s -= OCp;
    // get_embedder(d, 4): [x, sin(2^k x), cos(2^k x)]_k; pose = cat(embed(ori) [9], embed(pos) [18])  (profile_forward_2d.py:10-56, 149-151)
    float v = 0.f;
    if (s == 0) v = ori[r];
    else if (s < 9) { const int k = (s - 1) >> 1; const float a = ori[r] * (float)(1 << k); v = (s - 1) & 1 ? cosf(a) : sinf(a); }
    else if (s < 11) v = pos[2 * r + (s - 9)];
    else if (s < 27) { const int q = s - 11, k = q >> 2, which = q & 3; const float a = pos[2 * r + (which & 1)] * (float)(1 << k); v = which & 2 ? cosf(a) : sinf(a); }
    X0[r * 800 + 768 + s] = v;
}

std::vector<float> train_tfreqs(int half) {   // timestep_embedding (profile_forward_2d.py:68-71), float32 ops
    std::vector<float> f(half);
    const float l = -(float)std::log(10000.0);
    for (int i = 0; i < half; ++i) f[i] = expf(l * (float)i / (float)half);
    return f;
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace
}  // namespace dgdm

using namespace dgdm;

// Linear layers: 0 gripper_encoder.0, 1 gripper_encoder.2, 2 object_encoder.0, 3 object_encoder.2, 4 time_encoder.0, 5 time_encoder.2,
// 6..13 linears.{0,3,...,21}; then the eight BatchNorm1d layers linears.{1,4,...,22} and the output layer.
struct DgdmTrainer2d {
    struct Lin { std::string name; int K = 0, Kp = 0; size_t w = 0, b = 0, wt = 0; };
    int L = 0, OC = 0, Lp = 0, OCp = 0;
    float beta1 = 0.9f, beta2 = 0.95f, eps = 1e-8f, wd = 0.f;
    int64_t adam_steps = 0, bn_batches = 0;
    Lin lin[14];
    size_t bn_g[8], bn_b[8], out_w = 0, out_b = 0, n_params = 0, n_wt = 0;
    DevBuf P, G, M, V, WT, bn_run /* [8][2][256] running mean, var */, coef /* [8][7][256] */, tfreq, trdesc, loss_dev;
    DevBuf red /* [RED_S][3*256 + 4] float64 */, unit /* [256] ones, [256] zeros: coefficients of a ReLU without BatchNorm */;
    DevBuf ws, seg /* grouped encoders: outputs, their gradients, partial sums (grow-only) */;
    int64_t ws_rows = 0;
    DgdmTrainGroups groups{};      // one-shot hints of dgdm_trainer2d_set_groups
    // workspace pointers (set by reserve)
    float *bufC = nullptr, *bufO = nullptr, *bufT = nullptr, *H[3] = {nullptr, nullptr, nullptr}, *X0 = nullptr, *Y[8] = {}, *D[2] = {nullptr, nullptr},
          *G0 = nullptr, *stats = nullptr, *hpart = nullptr, *wpart = nullptr;
    int64_t wpart_floats = 0;

    float *p(size_t o) const { return P.as<float>() + o; }
    float *gr(size_t o) const { return G.as<float>() + o; }
    float *cf(int l, int row) const { return coef.as<float>() + ((size_t)l * 7 + row) * 256; }
    int col_of(int l, int c) const;      // internal column c of layer l -> column of the reference's weight, or -1 (padding)
    int reserve(int64_t N);
    int gemm(bool ptrans, int epi, GemmArgs &g, hipStream_t s) const;
    int reduce(const float *part, int T, int ncol, hipStream_t s) const;
    int wgrad(int l, const Operand &a, const Operand &dy, int64_t N, hipStream_t s);
    int run(const float *ctrl, const float *noise, const float *sa, const float *sb, const float *t, const float *ori, const float *pos,
            const float *obj, const float *score, int64_t N, int64_t Ntot, float lr, int train, bool update, float *pred, float *loss_host, hipStream_t s);
    int adam(float lr, hipStream_t s);
    int copy_state(int which, DgdmTensor *t, int n, bool to_device);
};

int DgdmTrainer2d::col_of(int l, int c) const {
    const Lin &x = lin[l];
    if (l != 6) return c < x.K ? c : -1;
    // linears.0 reads cat([object, gripper, pose, time]) (profile_forward_2d.py:154); internally [object | gripper | time | pose | pad]
    if (c < 512) return c;
    if (c < 768) return c - 512 + 539;
    if (c < 795) return c - 768 + 512;
    return -1;
}

int DgdmTrainer2d::reserve(int64_t N) {
    if (N <= ws_rows) return DGDM_OK;
    const int64_t T = (N + TI - 1) / TI, TH = (N + HEAD_ROWS - 1) / HEAD_ROWS;
    const int64_t splits_max = 256;
    wpart_floats = splits_max * (800 + 1) * 256;
    const int64_t per_row = Lp + OCp + 128 + 3 * 256 + 800 + 8 * 256 + 2 * 256 + 768;
    const int64_t total = N * per_row + std::max(T, TH) * 2 * 256 + TH * (3 * 256 + 4) + wpart_floats + 1024;
    int rc = ws.alloc((size_t)total * sizeof(float));
    if (rc) return rc;
    float *q = ws.as<float>();
    auto take = [&](int64_t n) { float *r = q; q += (n + 63) / 64 * 64; return r; };
    bufC = take(N * Lp); bufO = take(N * OCp); bufT = take(N * 128);
    for (int k = 0; k < 3; ++k) H[k] = take(N * 256);
    X0 = take(N * 800);
    for (int k = 0; k < 8; ++k) Y[k] = take(N * 256);
    D[0] = take(N * 256); D[1] = take(N * 256);
    G0 = take(N * 768);
    stats = take(std::max(T, TH) * 2 * 256);
    hpart = take(TH * (3 * 256 + 4));
    wpart = take(wpart_floats);
    ws_rows = N;
    DGDM_HIP_CHECK(hipMemset(X0, 0, (size_t)N * 800 * sizeof(float)));       // the padding columns 795..799 stay zero
    return DGDM_OK;
}

int DgdmTrainer2d::gemm(bool ptrans, int epi, GemmArgs &g, hipStream_t s) const {
    const int64_t it = (g.I + TI - 1) / TI;
    const int jt = (g.J + TJ - 1) / TJ;
    const int64_t splits = (g.R + g.r_per_split - 1) / g.r_per_split;
    const dim3 grid((unsigned)it, (unsigned)jt, (unsigned)splits), block(256);
    // bytes from each tensor's first element to the end of its last row's used columns: what the kernel's descriptor windows end at
    g.p_bytes = (ptrans ? (g.I - 1) * g.P.ld + g.R : (g.R - 1) * g.P.ld + g.I) * 4;
    g.q_bytes = ((g.R - 1) * g.Q.ld + g.J) * 4;
    g.c_bytes = epi == EPI_WGRAD ? 0 : ((g.I - 1) * g.ldc + g.J) * 4;
    g.y_bytes = g.Yp ? ((g.I - 1) * g.ldy + g.J) * 4 : 0;
    DGDM_REQUIRE(g.P.ld * 4 * 136 < 0x7fffffffLL && g.Q.ld * 4 * 36 < 0x7fffffffLL && g.ldc * 4 * 136 < 0x7fffffffLL, DGDM_EINVAL, "train2d gemm: row pitch too large");
    const int key = epi * 1000 + g.P.xf * 100 + g.Q.xf * 10 + g.mask;
    DGDM_REQUIRE(!ptrans || g.R <= 256 || g.P.xf == X_PLAIN, DGDM_EINVAL, "train2d gemm: transposed operand with coefficients wider than 256");
#define DGDM_TG(PT, E, PXV, QXV, MK) hipLaunchKernelGGL((tgemm_kernel<PT, E, PXV, QXV, MK>), grid, block, 0, s, g)
    switch (key) {
    case EPI_FWD * 1000 + X_PLAIN * 100: DGDM_TG(true, EPI_FWD, X_PLAIN, X_PLAIN, MASK_NONE); break;
    case EPI_FWD * 1000 + X_RELU * 100: DGDM_TG(true, EPI_FWD, X_RELU, X_PLAIN, MASK_NONE); break;
    case EPI_FWD * 1000 + X_SILU * 100: DGDM_TG(true, EPI_FWD, X_SILU, X_PLAIN, MASK_NONE); break;
    case EPI_BWD * 1000 + X_AFF2 * 100 + MASK_RELU_BN: DGDM_TG(true, EPI_BWD, X_AFF2, X_PLAIN, MASK_RELU_BN); break;
    case EPI_BWD * 1000 + X_AFF2 * 100 + MASK_NONE: DGDM_TG(true, EPI_BWD, X_AFF2, X_PLAIN, MASK_NONE); break;
    case EPI_BWD * 1000 + X_PLAIN * 100 + MASK_RELU: DGDM_TG(true, EPI_BWD, X_PLAIN, X_PLAIN, MASK_RELU); break;
    case EPI_BWD * 1000 + X_PLAIN * 100 + MASK_SILU: DGDM_TG(true, EPI_BWD, X_PLAIN, X_PLAIN, MASK_SILU); break;
    case EPI_WGRAD * 1000 + X_RELU * 100 + X_AFF2 * 10: DGDM_TG(false, EPI_WGRAD, X_RELU, X_AFF2, MASK_NONE); break;
    case EPI_WGRAD * 1000 + X_PLAIN * 100 + X_AFF2 * 10: DGDM_TG(false, EPI_WGRAD, X_PLAIN, X_AFF2, MASK_NONE); break;
    case EPI_WGRAD * 1000 + X_RELU * 100 + X_PLAIN * 10: DGDM_TG(false, EPI_WGRAD, X_RELU, X_PLAIN, MASK_NONE); break;
    case EPI_WGRAD * 1000 + X_SILU * 100 + X_PLAIN * 10: DGDM_TG(false, EPI_WGRAD, X_SILU, X_PLAIN, MASK_NONE); break;
    case EPI_WGRAD * 1000 + X_PLAIN * 100 + X_PLAIN * 10: DGDM_TG(false, EPI_WGRAD, X_PLAIN, X_PLAIN, MASK_NONE); break;
    default: DGDM_REQUIRE(false, DGDM_EINVAL, "train2d gemm: no kernel for epilogue %d, operands %d/%d, mask %d", epi, g.P.xf, g.Q.xf, g.mask);
    }
#undef DGDM_TG
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// column sums of `T` per-workgroup partial rows of `ncol` floats into the RED_S float64 rows the finalize kernels read
int DgdmTrainer2d::reduce(const float *part, int T, int ncol, hipStream_t s) const {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((ncol + 255) / 256, RED_S), dim3(256), 0, s, part, T, ncol, red.as<double>());
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// dW_l [k][m] = sum_n a[n][k] dY[n][m] and db_l = sum_n dY[n]: split over the rows, then one ordered sum
int DgdmTrainer2d::wgrad(int l, const Operand &a, const Operand &dy, int64_t N, hipStream_t s) {
    const Lin &x = lin[l];
    const int it = (x.Kp + TI - 1) / TI;
    int64_t splits = std::min<int64_t>(std::max<int64_t>(1, (N + 255) / 256), std::max(1, 512 / it));
    int64_t per = ((N + splits - 1) / splits + RC - 1) / RC * RC;
    splits = (N + per - 1) / per;
    GemmArgs g{};
    g.P = a; g.Q = dy; g.I = x.Kp; g.J = 256; g.R = N; g.r_per_split = per;
    g.C = wpart; g.ldc = 256; g.split_stride = (int64_t)(x.Kp + 1) * 256;
    int rc = gemm(false, EPI_WGRAD, g, s);
    if (rc) return rc;
    const int n = (x.Kp + 1) * 256;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, wpart, (int)splits, g.split_stride, x.Kp, gr(x.w), gr(x.b));
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// torch.optim.Adam(lr, betas, weight_decay) over every parameter (trainer.py:46), then the transposed weight copies
int DgdmTrainer2d::adam(float lr, hipStream_t s) {
    ++adam_steps;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)adam_steps), bc2 = 1.0 - std::pow((double)beta2, (double)adam_steps);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, s, P.as<float>(), G.as<float>(), M.as<float>(), V.as<float>(),
                       (int64_t)n_params, beta1, beta2, eps, wd, (float)((double)lr / bc1), (float)std::sqrt(bc2));
    DGDM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(transpose_all_kernel, dim3((800 * 256 + 255) / 256, 14), dim3(256), 0, s, P.as<float>(), WT.as<float>(), trdesc.as<TrDesc>());
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// N = this call's rows (BatchNorm statistics are taken over them); Ntot = the rows of the whole batch the loss is a mean over
// (= N for one process; the sum over the ranks for data-parallel training, dgdm_trainer2d_forward_backward)
int DgdmTrainer2d::run(const float *ctrl, const float *noise, const float *sa, const float *sb, const float *t, const float *ori, const float *pos,
                       const float *obj, const float *score, int64_t N, int64_t Ntot, float lr, int train, bool update, float *pred, float *loss_host,
                       hipStream_t s) {
    int rc = reserve(N);
    if (rc) return rc;
    const int T = (int)((N + TI - 1) / TI), TH = (int)((N + HEAD_ROWS - 1) / HEAD_ROWS);
    // one-shot grouping hints (dgdm_trainer2d_set_groups): which encoders run on their distinct inputs only
    const DgdmTrainGroups gh = groups;
    groups = DgdmTrainGroups{};
    const bool tgrp = gh.t_index_dev && gh.t_values_dev && gh.n_t > 0 && gh.n_t <= SEG_GROUPS;
    const bool ogrp = gh.rows_per_object > 1 && N % gh.rows_per_object == 0;
    const int orun = ogrp ? gh.rows_per_object : 1;
    const int64_t nobj = N / orun, erows[3] = {N, nobj, tgrp ? gh.n_t : N};
    const int ochunks = (orun + SEG_ROWS - 1) / SEG_ROWS;
    const int64_t tblocks = (N + SEG_ROWS - 1) / SEG_ROWS;
    float *encO = nullptr, *dencO = nullptr, *encT = nullptr, *dencT = nullptr, *segpart = nullptr;
    if (tgrp || ogrp) {
        const int64_t part = std::max<int64_t>(tgrp ? tblocks * gh.n_t * 256 : 0, ogrp ? nobj * ochunks * 256 : 0);
        if ((rc = seg.alloc((size_t)(2 * nobj * 256 + 2 * SEG_GROUPS * 256 + part + 256) * sizeof(float)))) return rc;
        encO = seg.as<float>(); dencO = encO + nobj * 256; encT = dencO + nobj * 256; dencT = encT + SEG_GROUPS * 256; segpart = dencT + SEG_GROUPS * 256;
    }
    {
        const int64_t n = N * (Lp + OCp + 32);
        hipLaunchKernelGGL(prep2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ctrl, noise, sa, sb, ori, pos, obj, N, L, Lp, OC, OCp, bufC, bufO, X0,
                           orun, nobj);
        DGDM_HIP_CHECK(hipGetLastError());
        if ((rc = time_embed(tgrp ? gh.t_values_dev : t, 0.f, tfreq.as<float>(), bufT, (int)erows[2], 64, s))) return rc;
    }
    const float *ones = unit.as<float>(), *zeros = unit.as<float>() + 256;
    auto plain = [](const float *p, int64_t ld) { Operand o{}; o.t0 = p; o.ld = ld; o.xf = X_PLAIN; return o; };
    auto act = [&](const float *p, int64_t ld, int a, const float *c0, const float *c2) {
        Operand o{};
        o.t0 = p; o.ld = ld; o.xf = a == ACT_SILU ? X_SILU : X_RELU; o.c0 = c0 ? c0 : ones; o.c2 = c2 ? c2 : zeros;
        return o;
    };
    auto weight_t = [&](int l) { return plain(WT.as<float>() + lin[l].wt, 256); };
    auto forward = [&](int l, const Operand &in, float *out, int64_t ldo, float *st, int64_t rows) {
        GemmArgs g{};
        g.P = in; g.Q = weight_t(l); g.I = rows; g.J = 256; g.R = lin[l].Kp; g.r_per_split = lin[l].Kp;
        g.C = out; g.ldc = ldo; g.bias = p(lin[l].b); g.stats = st;
        return gemm(true, EPI_FWD, g, s);
    };
    // encoders (profile_forward_2d.py:147-153); their second layers write straight into the concatenated trunk input X0 - or, grouped,
    // into a [groups][256] table that is then expanded over the rows
    const float *enc_in[3] = {bufC, bufO, bufT};
    const int enc_ld[3] = {Lp, OCp, 128}, enc_act[3] = {ACT_RELU, ACT_RELU, ACT_SILU}, enc_col[3] = {256, 0, 512};
    float *enc_tab[3] = {nullptr, ogrp ? encO : nullptr, tgrp ? encT : nullptr}, *enc_dtab[3] = {nullptr, dencO, dencT};
    for (int e = 0; e < 3; ++e) {
        if ((rc = forward(2 * e, plain(enc_in[e], enc_ld[e]), H[e], 256, nullptr, erows[e]))) return rc;
        if (!enc_tab[e]) {
            if ((rc = forward(2 * e + 1, act(H[e], 256, enc_act[e], nullptr, nullptr), X0 + enc_col[e], 800, nullptr, N))) return rc;
            continue;
        }
        if ((rc = forward(2 * e + 1, act(H[e], 256, enc_act[e], nullptr, nullptr), enc_tab[e], 256, nullptr, erows[e]))) return rc;
        hipLaunchKernelGGL(expand_groups_kernel, dim3((unsigned)((N * 64 + 255) / 256)), dim3(256), 0, s, enc_tab[e], e == 2 ? gh.t_index_dev : nullptr, orun,
                           X0 + enc_col[e], (int64_t)800, N);
        DGDM_HIP_CHECK(hipGetLastError());
    }
    // trunk: Linear -> BatchNorm1d (batch statistics) -> ReLU, eight times (profile_forward_2d.py:108-133)
    float *rm = bn_run.as<float>();
    if (!train)
        for (int k = 0; k < 8; ++k) {
            hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(1), dim3(256), 0, s, p(bn_g[k]), p(bn_b[k]), rm + (size_t)k * 512, rm + (size_t)k * 512 + 256, 1e-5f, cf(k, 0));
            DGDM_HIP_CHECK(hipGetLastError());
        }
    for (int k = 0; k < 8; ++k) {
        const Operand in = k == 0 ? plain(X0, 800) : act(Y[k - 1], 256, ACT_RELU, cf(k - 1, 0), cf(k - 1, 1));
        if ((rc = forward(6 + k, in, Y[k], 256, train ? stats : nullptr, N))) return rc;
        if (train) {
            if ((rc = reduce(stats, 2 * T, 256, s))) return rc;      // partial tile t = rows 2t (sum y), 2t + 1 (sum y^2): even / odd rows stay apart (RED_S is even)
            hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(1), dim3(256), 0, s, red.as<double>(), (double)N, p(bn_g[k]), p(bn_b[k]), 1e-5f, 0.1f,
                               rm + (size_t)k * 512, rm + (size_t)k * 512 + 256, cf(k, 0));
            DGDM_HIP_CHECK(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(head_kernel, dim3(TH), dim3(256), 0, s, Y[7], cf(7, 0), p(out_w), p(out_b), score, N, pred, D[0], stats, hpart, train, (double)Ntot);
    DGDM_HIP_CHECK(hipGetLastError());
    if ((rc = reduce(hpart, TH, 3 * 256 + 4, s))) return rc;
    hipLaunchKernelGGL(head_finalize_kernel, dim3(4), dim3(256), 0, s, red.as<double>(), (double)Ntot, train ? gr(out_w) : nullptr, train ? gr(out_b) : nullptr,
                       loss_dev.as<float>());      // eval mode leaves the gradient buffer alone
    DGDM_HIP_CHECK(hipGetLastError());
    if (train) {
        int nstat = TH;      // number of partial tiles `stats` holds for the layer about to be finalised
        for (int k = 7; k >= 0; --k) {
            if ((rc = reduce(stats, 2 * nstat, 256, s))) return rc;
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(256), 0, s, red.as<double>(), (double)N, cf(k, 0), gr(bn_g[k]), gr(bn_b[k]));
            DGDM_HIP_CHECK(hipGetLastError());
            float *dz = D[(7 - k) & 1], *dnext = D[(8 - k) & 1];
            Operand dy{};
            dy.t0 = dz; dy.t1 = Y[k]; dy.ld = 256; dy.xf = X_AFF2; dy.c0 = cf(k, 4); dy.c1 = cf(k, 5); dy.c2 = cf(k, 6);
            const Operand a = k == 0 ? plain(X0, 800) : act(Y[k - 1], 256, ACT_RELU, cf(k - 1, 0), cf(k - 1, 1));
            if ((rc = wgrad(6 + k, a, dy, N, s))) return rc;
            GemmArgs g{};
            g.P = dy; g.Q = plain(p(lin[6 + k].w), lin[6 + k].Kp); g.I = N; g.R = 256; g.r_per_split = 256;
            if (k > 0) {
                g.J = 256; g.C = dnext; g.ldc = 256; g.stats = stats; g.Yp = Y[k - 1]; g.ldy = 256; g.m0 = cf(k - 1, 0); g.m2 = cf(k - 1, 1); g.mask = MASK_RELU_BN;
            } else {
                g.J = 768; g.C = G0; g.ldc = 768; g.mask = MASK_NONE;       // the pose columns need no gradient
            }
            if ((rc = gemm(true, EPI_BWD, g, s))) return rc;
            nstat = T;
        }
        for (int e = 0; e < 3; ++e) {
            Operand gsec = plain(G0 + enc_col[e], 768);
            if (enc_tab[e]) {       // gradient of the table: the rows' gradients summed per group, in a fixed order
                if (e == 2) {
                    hipLaunchKernelGGL(segsum_index_kernel, dim3((unsigned)tblocks), dim3(256), 0, s, G0 + enc_col[e], (int64_t)768, gh.t_index_dev, gh.n_t, N, segpart);
                    DGDM_HIP_CHECK(hipGetLastError());
                    if ((rc = reduce(segpart, (int)tblocks, gh.n_t * 256, s))) return rc;
                    hipLaunchKernelGGL(sum_red_kernel, dim3((gh.n_t * 256 + 255) / 256), dim3(256), 0, s, red.as<double>(), gh.n_t * 256, enc_dtab[e]);
                } else {
                    hipLaunchKernelGGL(segsum_run_kernel, dim3((unsigned)nobj, (unsigned)ochunks), dim3(256), 0, s, G0 + enc_col[e], (int64_t)768, orun, segpart);
                    DGDM_HIP_CHECK(hipGetLastError());
                    hipLaunchKernelGGL(sum_chunks_kernel, dim3((unsigned)((nobj * 256 + 255) / 256)), dim3(256), 0, s, segpart, ochunks, enc_dtab[e], nobj * 256);
                }
                DGDM_HIP_CHECK(hipGetLastError());
                gsec = plain(enc_dtab[e], 256);
            }
            if ((rc = wgrad(2 * e + 1, act(H[e], 256, enc_act[e], nullptr, nullptr), gsec, erows[e], s))) return rc;
            GemmArgs g{};
            g.P = gsec; g.Q = plain(p(lin[2 * e + 1].w), 256); g.I = erows[e]; g.J = 256; g.R = 256; g.r_per_split = 256;
            g.C = D[0]; g.ldc = 256; g.Yp = H[e]; g.ldy = 256; g.mask = enc_act[e] == ACT_SILU ? MASK_SILU : MASK_RELU;
            if ((rc = gemm(true, EPI_BWD, g, s))) return rc;
            if ((rc = wgrad(2 * e, plain(enc_in[e], enc_ld[e]), plain(D[0], 256), erows[e], s))) return rc;
        }
        if (update && (rc = adam(lr, s))) return rc;
        ++bn_batches;
    }
    if (loss_host) {
        DGDM_HIP_CHECK(hipMemcpyAsync(loss_host, loss_dev.p, sizeof(float), hipMemcpyDeviceToHost, s));
        DGDM_HIP_CHECK(hipStreamSynchronize(s));
    }
    return DGDM_OK;
}

// which: 0 parameters and BatchNorm running statistics, 1 gradients, 2 / 3 Adam's first / second moments
int DgdmTrainer2d::copy_state(int which, DgdmTensor *t, int n, bool to_device) {
    DevBuf *src = which == 0 ? &P : which == 1 ? &G : which == 2 ? &M : &V;
    std::vector<float> host(n_params), run(8 * 512);
    DGDM_HIP_CHECK(hipDeviceSynchronize());
    DGDM_HIP_CHECK(hipMemcpy(host.data(), src->p, n_params * sizeof(float), hipMemcpyDeviceToHost));
    DGDM_HIP_CHECK(hipMemcpy(run.data(), bn_run.p, run.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::map<std::string, DgdmTensor *> by;
    for (int i = 0; i < n; ++i) by[t[i].name] = &t[i];
    auto slot = [&](const std::string &k, int64_t numel, bool required) -> float * {
        auto it = by.find(k);
        if (it == by.end()) { if (required) set_error("state_dict key '%s' missing", k.c_str()); return nullptr; }
        if (it->second->dtype != 0 || it->second->numel != numel) { set_error("state_dict key '%s': expected %lld float32 values", k.c_str(), (long long)numel); return nullptr; }
        return const_cast<float *>(static_cast<const float *>(it->second->data));
    };
    auto xfer = [&](float *user, float *mine, size_t cnt) { if (to_device) memcpy(mine, user, cnt * sizeof(float)); else memcpy(user, mine, cnt * sizeof(float)); };
    for (int l = 0; l < 14; ++l) {
        const Lin &x = lin[l];
        float *wu = slot(x.name + ".weight", (int64_t)256 * x.K, true), *bu = slot(x.name + ".bias", 256, true);
        if (!wu || !bu) return DGDM_EKEY;
        for (int m = 0; m < 256; ++m)
            for (int c = 0; c < x.Kp; ++c) {
                const int rc = col_of(l, c);
                if (rc < 0) { if (to_device) host[x.w + (size_t)m * x.Kp + c] = 0.f; continue; }
                xfer(wu + (size_t)m * x.K + rc, &host[x.w + (size_t)m * x.Kp + c], 1);
            }
        xfer(bu, &host[x.b], 256);
    }
    for (int k = 0; k < 8; ++k) {
        const std::string bn = "linears." + std::to_string(3 * k + 1);
        float *gu = slot(bn + ".weight", 256, true), *bu = slot(bn + ".bias", 256, true);
        if (!gu || !bu) return DGDM_EKEY;
        xfer(gu, &host[bn_g[k]], 256); xfer(bu, &host[bn_b[k]], 256);
        if (which == 0) {
            float *mu = slot(bn + ".running_mean", 256, true), *vu = slot(bn + ".running_var", 256, true);
            if (!mu || !vu) return DGDM_EKEY;
            xfer(mu, &run[(size_t)k * 512], 256); xfer(vu, &run[(size_t)k * 512 + 256], 256);
        }
    }
    float *wu = slot("output.weight", 3 * 256, true), *bu = slot("output.bias", 3, true);
    if (!wu || !bu) return DGDM_EKEY;
    xfer(wu, &host[out_w], 768); xfer(bu, &host[out_b], 3);
    if (to_device) {
        DGDM_HIP_CHECK(hipMemcpy(src->p, host.data(), n_params * sizeof(float), hipMemcpyHostToDevice));
        if (which == 0) {
            DGDM_HIP_CHECK(hipMemcpy(bn_run.p, run.data(), run.size() * sizeof(float), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(transpose_all_kernel, dim3((800 * 256 + 255) / 256, 14), dim3(256), 0, 0, P.as<float>(), WT.as<float>(), trdesc.as<TrDesc>());
            DGDM_HIP_CHECK(hipGetLastError());
            DGDM_HIP_CHECK(hipDeviceSynchronize());
        }
    }
    return DGDM_OK;
}

extern "C" int dgdm_trainer2d_create(DgdmTrainer2d **out, const DgdmTensor *state_dict, int n_tensors, int params_ch, int object_ch, float beta1,
                                     float beta2, float eps, float weight_decay) {
    DGDM_REQUIRE(out && state_dict && params_ch > 0 && object_ch > 0, DGDM_EINVAL, "dgdm_trainer2d_create: bad argument");
    std::unique_ptr<DgdmTrainer2d> m(new DgdmTrainer2d());
    m->L = params_ch; m->OC = object_ch; m->Lp = round_up(params_ch, RC); m->OCp = round_up(object_ch, RC);
    m->beta1 = beta1; m->beta2 = beta2; m->eps = eps; m->wd = weight_decay;
    const char *names[14] = {"gripper_encoder.0", "gripper_encoder.2", "object_encoder.0", "object_encoder.2", "time_encoder.0", "time_encoder.2",
                             "linears.0", "linears.3", "linears.6", "linears.9", "linears.12", "linears.15", "linears.18", "linears.21"};
    const int K[14] = {params_ch, 256, object_ch, 256, 128, 256, 795, 256, 256, 256, 256, 256, 256, 256};
    size_t o = 0, ot = 0;
    std::vector<TrDesc> td(14);
    for (int l = 0; l < 14; ++l) {
        DgdmTrainer2d::Lin &x = m->lin[l];
        x.name = names[l]; x.K = K[l]; x.Kp = round_up(K[l], RC);
        x.w = o; o += (size_t)256 * x.Kp; x.b = o; o += 256;
        x.wt = ot; ot += (size_t)256 * x.Kp;
        td[l] = TrDesc{(int64_t)x.w, (int64_t)x.wt, x.Kp};
    }
    for (int k = 0; k < 8; ++k) { m->bn_g[k] = o; o += 256; m->bn_b[k] = o; o += 256; }
    m->out_w = o; o += 768; m->out_b = o; o += 64;
    m->n_params = o; m->n_wt = ot;
    int rc;
    for (DevBuf *b : {&m->P, &m->G, &m->M, &m->V}) {
        if ((rc = b->alloc(o * sizeof(float)))) return rc;
        DGDM_HIP_CHECK(hipMemset(b->p, 0, o * sizeof(float)));
    }
    if ((rc = m->WT.alloc(ot * sizeof(float)))) return rc;
    if ((rc = m->bn_run.alloc(8 * 512 * sizeof(float)))) return rc;
    if ((rc = m->coef.alloc(8 * 7 * 256 * sizeof(float)))) return rc;
    DGDM_HIP_CHECK(hipMemset(m->coef.p, 0, 8 * 7 * 256 * sizeof(float)));
    if ((rc = m->loss_dev.alloc(64))) return rc;
    if ((rc = m->red.alloc((size_t)RED_S * SEG_GROUPS * 256 * sizeof(double)))) return rc;       // also >= RED_S * (3*256 + 4)
    {
        std::vector<float> u(512, 0.f);
        std::fill(u.begin(), u.begin() + 256, 1.f);
        if ((rc = m->unit.upload(u.data(), u.size() * sizeof(float)))) return rc;
    }
    const std::vector<float> f = train_tfreqs(64);
    if ((rc = m->tfreq.upload(f.data(), f.size() * sizeof(float)))) return rc;
    if ((rc = m->trdesc.upload(td.data(), td.size() * sizeof(TrDesc)))) return rc;
    if ((rc = m->copy_state(0, const_cast<DgdmTensor *>(state_dict), n_tensors, true))) return rc;
    *out = m.release();
    return DGDM_OK;
}

extern "C" void dgdm_trainer2d_destroy(DgdmTrainer2d *m) { delete m; }

extern "C" int dgdm_trainer2d_step(DgdmTrainer2d *m, const float *ctrl_dev, const float *noise_dev, const float *sqrt_abar_dev,
                                   const float *sqrt_1m_abar_dev, const float *t_dev, const float *ori_dev, const float *pos_dev, const float *object_dev,
                                   const float *score_dev, int64_t rows, float lr, int train, float *pred_dev, float *loss_host, void *stream) {
    DGDM_REQUIRE(m && ctrl_dev && t_dev && ori_dev && pos_dev && object_dev && score_dev && pred_dev, DGDM_EINVAL, "dgdm_trainer2d_step: null argument");
    DGDM_REQUIRE(!noise_dev || (sqrt_abar_dev && sqrt_1m_abar_dev), DGDM_EINVAL, "dgdm_trainer2d_step: noise without its two scale vectors");
    DGDM_REQUIRE(rows >= (train ? 2 : 1) && rows < ((int64_t)1 << 31) / 800, DGDM_EINVAL,
                 "dgdm_trainer2d_step: %lld rows (BatchNorm1d in training mode needs at least 2; the workspace index math stops at 2^31/800)", (long long)rows);
    return m->run(ctrl_dev, noise_dev, sqrt_abar_dev, sqrt_1m_abar_dev, t_dev, ori_dev, pos_dev, object_dev, score_dev, rows, rows, lr, train, true,
                  pred_dev, loss_host, (hipStream_t)stream);
}

extern "C" int dgdm_trainer2d_forward_backward(DgdmTrainer2d *m, const float *ctrl_dev, const float *noise_dev, const float *sqrt_abar_dev,
                                               const float *sqrt_1m_abar_dev, const float *t_dev, const float *ori_dev, const float *pos_dev,
                                               const float *object_dev, const float *score_dev, int64_t rows, int64_t total_rows, float *pred_dev,
                                               float *loss_host, void *stream) {
    DGDM_REQUIRE(m && ctrl_dev && t_dev && ori_dev && pos_dev && object_dev && score_dev && pred_dev, DGDM_EINVAL, "dgdm_trainer2d_forward_backward: null argument");
    DGDM_REQUIRE(!noise_dev || (sqrt_abar_dev && sqrt_1m_abar_dev), DGDM_EINVAL, "dgdm_trainer2d_forward_backward: noise without its two scale vectors");
    DGDM_REQUIRE(rows >= 2 && rows <= total_rows && rows < ((int64_t)1 << 31) / 800, DGDM_EINVAL, "dgdm_trainer2d_forward_backward: %lld of %lld rows",
                 (long long)rows, (long long)total_rows);
    return m->run(ctrl_dev, noise_dev, sqrt_abar_dev, sqrt_1m_abar_dev, t_dev, ori_dev, pos_dev, object_dev, score_dev, rows, total_rows, 0.f, 1, false,
                  pred_dev, loss_host, (hipStream_t)stream);
}

extern "C" int dgdm_trainer2d_set_groups(DgdmTrainer2d *m, const DgdmTrainGroups *g) {
    DGDM_REQUIRE(m, DGDM_EINVAL, "dgdm_trainer2d_set_groups: null handle");
    m->groups = g ? *g : DgdmTrainGroups{};
    return DGDM_OK;
}

extern "C" int64_t dgdm_trainer2d_gradient_count(const DgdmTrainer2d *m) { return m ? (int64_t)m->n_params : -1; }

extern "C" int dgdm_trainer2d_gradients(DgdmTrainer2d *m, float *flat_dev, int64_t numel, int to_trainer, void *stream) {
    DGDM_REQUIRE(m && flat_dev && numel == (int64_t)m->n_params, DGDM_EINVAL, "dgdm_trainer2d_gradients: expected %lld values", m ? (long long)m->n_params : 0LL);
    DGDM_HIP_CHECK(hipMemcpyAsync(to_trainer ? m->G.p : (void *)flat_dev, to_trainer ? (const void *)flat_dev : m->G.p, (size_t)numel * sizeof(float),
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DGDM_OK;
}

// BatchNorm running statistics [8 layers][mean | var][256] device <-> flat_dev: under data parallelism every rank updates them from ITS
// chunk, and nn.DataParallel keeps replica 0's (dynamics/trainer.py:41-43) - rank 0's buffers are broadcast after every step
extern "C" int dgdm_trainer2d_running_stats(DgdmTrainer2d *m, float *flat_dev, int64_t numel, int to_trainer, void *stream) {
    DGDM_REQUIRE(m && flat_dev && numel == 8 * 512, DGDM_EINVAL, "dgdm_trainer2d_running_stats: expected %d values", 8 * 512);
    DGDM_HIP_CHECK(hipMemcpyAsync(to_trainer ? m->bn_run.p : (void *)flat_dev, to_trainer ? (const void *)flat_dev : m->bn_run.p, (size_t)numel * sizeof(float),
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DGDM_OK;
}

extern "C" int dgdm_trainer2d_apply(DgdmTrainer2d *m, float lr, void *stream) {
    DGDM_REQUIRE(m, DGDM_EINVAL, "dgdm_trainer2d_apply: null handle");
    return m->adam(lr, (hipStream_t)stream);
}

extern "C" int dgdm_trainer2d_export(DgdmTrainer2d *m, int which, DgdmTensor *tensors, int n_tensors) {
    DGDM_REQUIRE(m && tensors && which >= 0 && which <= 3, DGDM_EINVAL, "dgdm_trainer2d_export: bad argument");
    return m->copy_state(which, tensors, n_tensors, false);
}

extern "C" int64_t dgdm_trainer2d_steps(const DgdmTrainer2d *m) { return m ? m->bn_batches : -1; }
