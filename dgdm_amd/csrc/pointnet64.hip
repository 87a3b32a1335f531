// The per-object stages of the PointNet++ table pipeline (pointnet.hip: T2 sa1 features, T4 sa2's pair MLP, T6 sa3's layer) with
// float64 accumulation.  These stages run once per OBJECT, not once per replicated row, so double precision costs a few per cent of a
// denoise step - and it takes the object embedding from 1.7e-6 (k-ordered float32 chains, BatchNorm folded into rounded weights) to
// the float32 rounding of an exact result (~1e-7; torch's own float32 embedding is 3e-7 from exact), which is what decides how many
// ReLU pre-activations of the trunk land on the wrong side of zero (DESIGN_HISTORY.md 4.9, scripts/exp_ties.py).
//
// Same dataflow, same index decisions (float32 distances in the reference's operation order), same float32 tables out: every
// table entry is ONE rounding of a float64 accumulation over float32 inputs, with the BatchNorm fold kept in float64 (Folded64).
// The two contractions that matter (sa2 128 -> 256 on every in-radius pair, sa3 256 -> 256 on every (variant, crowded centre)) run
// on v_mfma_f64_16x16x4_f64: one wave = 16 rows x all 256 output features, weights as the A operand from a pre-arranged image
// (pack_mfma64) through the same buffer-load ring as the float32 chain kernels, activations as the B operand from registers.
//   A: lane (m = l & 15, kq = l >> 4) supplies W[f(mt, m)][kq * K/4 + ks]     B: lane (n = l & 15, kq) supplies in[row n][kq * K/4 + ks]
//   C/D: lane (n, rb = l >> 4), register i = output feature f(mt, rb + 4 i) = 16 mt + 4 rb + i of row n  (four consecutive features)
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include "mfma_chain.h"
#include "pointnet.h"
#include "pointnet_dev.h"

#pragma clang fp contract(off)      // as in pointnet.hip (the float32 index decisions); the float64 arithmetic below uses explicit fma()

namespace dgdm {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// Weight ring of depth D (D divides TOTAL) carried across calls, as mfma_chain.h stream_cont with a chosen depth: two waves per SIMD
// hide the L2 latency for each other, so 8 entries in flight per wave suffice and the kernels fit 256 registers.
template <int D>
__device__ __forceinline__ void ring64_fill(wrsrc_t rs, int voff, int base_off, float4 (&ring)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) ring[i] = wload(rs, voff, base_off + i * 1024);
}

template <int TOTAL, int D, class Body>
__device__ __forceinline__ void ring64_stream(wrsrc_t rs, int voff, int base_off, float4 (&ring)[D], Body &&body) {
    static_assert(TOTAL % D == 0, "pass length must be a multiple of the ring depth");
#pragma unroll
    for (int i = 0; i < TOTAL; ++i) {
        const float4 a = ring[i % D];
        ring[i % D] = wload(rs, voff, base_off + (i + D) * 1024);
        body(i, a);
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr int RING64 = 8;
constexpr int PN64_GRID = 1024;       // workgroups of the launches whose item count is device data (two per CU resident, two rounds)

// ------------------------------------------------------------------------------------------------ T2
// sa1 feature of every point as a centre (pointnet.hip sa1_kernel) -> F1 [N][128] doubles
__global__ __launch_bounds__(128) void sa1_64_kernel(const float *xyz, int N, float r2, const double *__restrict__ w0t /*[3][64]*/,
                                                     const double *__restrict__ b0, const double *__restrict__ w1 /*[128][64]*/,
                                                     const double *__restrict__ b1, double *F1 /*[N][128]*/) {
    __shared__ int nbr[32];
    __shared__ double h1[32][64];
    const int t = threadIdx.x, lane = t & 63;
    xyz += (size_t)blockIdx.y * 3 * N; F1 += (size_t)blockIdx.y * N * 128;      // object of a batched launch
#ifdef DGDM_SA1_CLOCKS
    long long tk[6]; int nk = 0;
#define SA1_STAMP() do { if (nk < 6) tk[nk++] = __builtin_readcyclecounter(); } while (0)
#else
#define SA1_STAMP() do {} while (0)
#endif
    SA1_STAMP();
    double wrow[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wrow[k] = w1[t * 64 + k];
    const double bias1 = b1[t];
    for (int p = blockIdx.x; p < N; p += gridDim.x) {
        const float cx = xyz[3 * p], cy = xyz[3 * p + 1], cz = xyz[3 * p + 2];
        SA1_STAMP();
        if (t < 64) ball_first32(xyz, N, p, cx, cy, cz, sq3(cx, cy, cz), r2, nbr, lane);      // wave 0; float32 distances (index decision)
        __syncthreads();
        SA1_STAMP();
        for (int i = t; i < 32 * 64; i += 128) {          // layer 0 on the relative coordinates (exact differences of float32 values)
            const int s = i >> 6, c = i & 63, k = nbr[s];
            const double dx = (double)xyz[3 * k] - (double)cx, dy = (double)xyz[3 * k + 1] - (double)cy, dz = (double)xyz[3 * k + 2] - (double)cz;
            h1[s][c] = fmax(fma(w0t[128 + c], dz, fma(w0t[64 + c], dy, fma(w0t[c], dx, b0[c]))), 0.0);
        }
        __syncthreads();
        SA1_STAMP();
        double best = 0.0;                                 // ReLU outputs are >= 0 and the group is never empty
        for (int s = 0; s < 32; ++s) {
            double acc = bias1;
#pragma unroll
            for (int k = 0; k < 64; ++k) acc = fma(wrow[k], h1[s][k], acc);
            best = fmax(best, acc);
        }
        F1[(size_t)p * 128 + t] = best;
        __syncthreads();
        SA1_STAMP();
#ifdef DGDM_SA1_CLOCKS
        if (t == 0 && blockIdx.x == 100 && blockIdx.y == 3) printf("sa1: weights %lld ball %lld layer0 %lld layer1 %lld\n", tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3]);
#endif
    }
}

// ------------------------------------------------------------------------------------------------ T4
// Y[pair][256] = float32( ReLU(W2b' ReLU(U[k] + Vx (xyz_k - xyz_c)) + b2b') ) for every in-radius ordered pair (pointnet.hip pair_kernel)
__global__ __launch_bounds__(256, 2) void pair64_kernel(const float *__restrict__ xyz, int N, const double *__restrict__ U /*[N][128]*/,
                                                     const double *__restrict__ vx /*[3][128]*/, const double *__restrict__ img,
                                                     const double *__restrict__ bias, const int *__restrict__ pairs,
                                                     const int *__restrict__ off /*[N+1]*/, float *__restrict__ Y) {
    const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total = off[N];
    struct D2 { double lo, hi; };
    const wrsrc_t rs = weight_rsrc(reinterpret_cast<const float4 *>(img), 256 * 1024);
    const double *v = vx + kq * 32;
    // the pair count is device data: a bounded grid whose waves stride over the 16-pair tiles
#pragma nounroll
    for (int tile = blockIdx.x * 4 + wave; tile * 16 < total; tile += gridDim.x * 4) {
        asm volatile("" ::: "memory");                     // keeps the loop-invariant bias loads inside (hoisted they spill)
        const int p = min(tile * 16 + n, total - 1);
        const int ck = pairs[p], c = ck >> 16, k = ck & 0xffff;
        const double dx = (double)xyz[3 * k] - (double)xyz[3 * c], dy = (double)xyz[3 * k + 1] - (double)xyz[3 * c + 1],
                     dz = (double)xyz[3 * k + 2] - (double)xyz[3 * c + 2];
        const double *urow = U + (size_t)k * 128 + kq * 32;
        f64x4 acc[16];
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[mt][i] = bias[16 * mt + 4 * kq + i];
        }
        // K = 128 in two chunks of 16 K-steps (a rolled loop): the lane's 16 inputs of a chunk are made just before it runs, the weight
        // ring is carried across - keeps the kernel inside 256 registers, i.e. two waves per SIMD
        float4 ring[RING64];
        ring64_fill<RING64>(rs, lane * 16, 0, ring);
#pragma nounroll
        for (int ch = 0; ch < 2; ++ch) {
            double in[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int f = 16 * ch + j;
                in[j] = fmax(fma(v[256 + f], dz, fma(v[128 + f], dy, fma(v[f], dx, urow[f]))), 0.0);
            }
            ring64_stream<128, RING64>(rs, lane * 16, ch * 128 * 1024, ring, [&](int e, const float4 a) {
                const int ks = e / 8, mp = e % 8;
                const D2 w = __builtin_bit_cast(D2, a);
                const double b = in[ks];
                acc[2 * mp] = __builtin_amdgcn_mfma_f64_16x16x4f64(w.lo, b, acc[2 * mp], 0, 0, 0);
                acc[2 * mp + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(w.hi, b, acc[2 * mp + 1], 0, 0, 0);
            });
        }
        if (tile * 16 + n < total) {
            float *dst = Y + (size_t)p * 256 + 4 * kq;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt)
                *reinterpret_cast<float4 *>(dst + 16 * mt) = make_float4(fmaxf((float)acc[mt][0], 0.f), fmaxf((float)acc[mt][1], 0.f),
                                                                          fmaxf((float)acc[mt][2], 0.f), fmaxf((float)acc[mt][3], 0.f));
        }
    }
}

// ------------------------------------------------------------------------------------------------ T6
// Z[row][256] = float32( ReLU(W3'[:,3:] L2[row] + W3'[:,0:3] xyz_c + b3') ),  rows as pointnet.hip z_kernel's two modes in ONE launch:
// items 0 .. N-1 are slot 0's rows (every centre), item N + j is (variant 1 + j / ncr, crowded centre clist[j % ncr]).  The item count
// is device data: a bounded grid whose waves stride over the 16-row tiles (a worst-case grid is 4096 workgroups of which ~330 find work).
// (Round 5, tried: two tiles per wave against one pass over the weight image - half the L2 traffic per FLOP, one wave per SIMD with
//  256 accumulator registers: 232 us per object instead of 143, a single wave does not hide the latency of the stream.)
__global__ __launch_bounds__(256, 2) void z64_kernel(const float *__restrict__ xyz, int N, int nv, const float *__restrict__ L2,
                                                  const double *__restrict__ img, const double *__restrict__ w3x /*[3][256]*/,
                                                  const double *__restrict__ bias, float *__restrict__ Z,
                                                  const int *__restrict__ clist, const int *__restrict__ ncr) {
    const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ncrv = nv > 1 ? *ncr : 0;
    const int items = N + (nv - 1) * ncrv;                 // <= nv * N
    struct D2 { double lo, hi; };
    const wrsrc_t rs = weight_rsrc(reinterpret_cast<const float4 *>(img), 512 * 1024);
#pragma nounroll
    for (int tile = blockIdx.x * 4 + wave; tile * 16 < items; tile += gridDim.x * 4) {
        asm volatile("" ::: "memory");                     // keeps the loop-invariant bias / coordinate-weight loads inside (hoisted they spill)
        const int item = min(tile * 16 + n, items - 1);
        const int j = item - N;
        const int c = j < 0 ? item : clist[j % ncrv];
        const int row = j < 0 ? c : (1 + j / ncrv) * N + c;
        const double x = xyz[3 * c], y = xyz[3 * c + 1], z = xyz[3 * c + 2];
        const float4 *src = reinterpret_cast<const float4 *>(L2 + (size_t)row * 256 + kq * 64);
        f64x4 acc[16];
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 16 * mt + 4 * kq + i;
                acc[mt][i] = fma(w3x[512 + f], z, fma(w3x[256 + f], y, fma(w3x[f], x, bias[f])));
            }
        }
        // K = 256 in four chunks of 16 K-steps (a rolled loop: 1024 MFMAs in one basic block is more than hipcc unrolls); the lane's 16
        // inputs of the next chunk are loaded while the current one runs, the weight ring is carried across the chunks
        float4 ring[RING64];
        ring64_fill<RING64>(rs, lane * 16, 0, ring);
        float4 cur[4], nxt[4];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) cur[j4] = src[j4];
#pragma nounroll
        for (int ch = 0; ch < 4; ++ch) {
            const int cn = min(ch + 1, 3);
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) nxt[j4] = src[4 * cn + j4];
            float in[16];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) { in[4 * j4] = cur[j4].x; in[4 * j4 + 1] = cur[j4].y; in[4 * j4 + 2] = cur[j4].z; in[4 * j4 + 3] = cur[j4].w; }
            ring64_stream<128, RING64>(rs, lane * 16, ch * 128 * 1024, ring, [&](int e, const float4 a) {
                const int ks = e / 8, mp = e % 8;
                const D2 w = __builtin_bit_cast(D2, a);
                const double b = (double)in[ks];
                acc[2 * mp] = __builtin_amdgcn_mfma_f64_16x16x4f64(w.lo, b, acc[2 * mp], 0, 0, 0);
                acc[2 * mp + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(w.hi, b, acc[2 * mp + 1], 0, 0, 0);
            });
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) cur[j4] = nxt[j4];
        }
        if (tile * 16 + n < items) {
            float *dst = Z + (size_t)row * 256 + 4 * kq;       // rows of one launch are distinct (item -> row is injective)
#pragma unroll
            for (int mt = 0; mt < 16; ++mt)
                *reinterpret_cast<float4 *>(dst + 16 * mt) = make_float4(fmaxf((float)acc[mt][0], 0.f), fmaxf((float)acc[mt][1], 0.f),
                                                                          fmaxf((float)acc[mt][2], 0.f), fmaxf((float)acc[mt][3], 0.f));
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
int pn_sa1_64(const float *xyz, int N, float r1sq, const PnWeights64 &w, double *F1_64, hipStream_t s, int nobj) {
    hipLaunchKernelGGL(sa1_64_kernel, dim3(std::min(N, 1024), nobj), dim3(128), 0, s, xyz, N, r1sq, w.sa1_w0t, w.sa1_b0, w.sa1_w1, w.sa1_b1, F1_64);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_pairs64(const float *xyz, int N, const double *U64, const PnWeights64 &w, const int *pairs, const int *off, float *Y, hipStream_t s) {
    const int tiles = N * ((N + 15) / 16);     // worst case (every point inside every ball); surplus workgroups leave at once
    hipLaunchKernelGGL(pair64_kernel, dim3(std::min((tiles + 3) / 4, PN64_GRID)), dim3(256), 0, s, xyz, N, U64, w.sa2_vx, w.sa2_w1_img, w.sa2_b1, pairs, off, Y);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_z64(const float *xyz, int N, int nv, const PnWeights64 &w, const float *L2, float *Z, const int *clist, const int *ncr, hipStream_t s) {
    const int64_t tiles = ((int64_t)nv * N + 15) / 16;     // worst case (every centre crowded)
    hipLaunchKernelGGL(z64_kernel, dim3((unsigned)std::min<int64_t>((tiles + 3) / 4, PN64_GRID)), dim3(256), 0, s, xyz, N, nv, L2, w.sa3_w_img, w.sa3_wx,
                       w.sa3_b, Z, clist, ncr);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

}  // namespace dgdm
