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

// max(v, v of the lane a DPP pattern pairs this lane with): quad_perm [1,0,3,2] 0xB1, [2,3,0,1] 0x4E, row_half_mirror 0x141, row_mirror
// 0x140 - four steps make the maximum over a row of 16 lanes on the VALU alone (a shuffle through LDS costs a round trip per step)
template <int CTRL>
__device__ __forceinline__ double dpp_max64(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xf, 0xf, false);
    return fmax(v, __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo));
}

// ------------------------------------------------------------------------------------------------ T2
// sa1 feature of every point as a centre (pointnet.hip sa1_kernel) -> F1 [N][128] doubles.
// One wave takes TWO centres: lane = (centre, one of its 32 neighbours) holds that row's 64 layer-0 values in registers, and the
// 64 -> 128 layer runs feature by feature with the weight row as a wave-uniform operand: acc = b[f]; acc = fma(w[f][k], h[k], acc) for
// k ascending - the same chain as ever, so F1 keeps its bits - then the group's maximum over the 32 lanes of the centre.  The weights
// (64 KiB of doubles) are staged in LDS once per workgroup (persistent workgroups, two per CU, stride over all (object, centre pair)
// tasks) and read as broadcasts, 32 ds_read_b128 per 64 FMAs: 0.5 LDS cycles per FMA cycle and CU.  (As scalar loads - s_load into
// SGPRs, one per v_fma_f64 - the same loop ran 50x slower than its FMA count: every wave of the chip pulls the same 64 KiB through
// 16 KiB scalar caches.  Round 5's form had the roles the other way round - lane = output feature, weights in registers, every
// ACTIVATION a broadcast read, 2 KiB of them per centre and lane - at 35 cycles per v_fma_f64: 2.1 ms for the 32 objects of a bench
// step, on the critical path of the table build.)
constexpr int SA1_LDS_BYTES = (128 * 64 + 128) * 8;
__global__ __launch_bounds__(256) void sa1_64_kernel(const float *xyz_all, int N, int nobj, float r2, const double *__restrict__ w0t /*[3][64]*/,
                                                     const double *__restrict__ b0, const double *__restrict__ w1 /*[128][64]*/,
                                                     const double *__restrict__ b1, double *F1_all /*[nobj][N][128]*/) {
    extern __shared__ __attribute__((aligned(16))) double sa1_w[];      // [128][64] weights, [128] biases
    __shared__ int nbr_all[4][2][32];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 128 * 64; i += 256) sa1_w[i] = w1[i];
    if (threadIdx.x < 128) sa1_w[128 * 64 + threadIdx.x] = b1[threadIdx.x];
    __syncthreads();
    int (*nbr)[32] = nbr_all[wave];
    const int half = lane >> 5, sl = lane & 31;
    const int ppo = (N + 1) / 2;                            // centre pairs per object
    for (int task = blockIdx.x * 4 + wave; task < nobj * ppo; task += gridDim.x * 4) {
        const int ob = task / ppo, pp = task - ob * ppo;
        const float *xyz = xyz_all + (size_t)ob * 3 * N;
        double *F1 = F1_all + (size_t)ob * N * 128;
        const int p0 = 2 * pp, p1 = min(2 * pp + 1, N - 1);
#ifdef DGDM_SA1_CLOCKS
        const long long tk0 = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int c = 0; c < 2; ++c) {                        // float32 distances (index decision), one centre after the other
            const int pc = c ? p1 : p0;
            const float cx = xyz[3 * pc], cy = xyz[3 * pc + 1], cz = xyz[3 * pc + 2];
            ball_first32(xyz, N, pc, cx, cy, cz, sq3(cx, cy, cz), r2, nbr[c], lane);
        }
        __builtin_amdgcn_wave_barrier();
        const int p = half ? p1 : p0, k = nbr[half][sl];
        // layer 0 on the relative coordinates (exact differences of float32 values)
        const double dx = (double)xyz[3 * k] - (double)xyz[3 * p], dy = (double)xyz[3 * k + 1] - (double)xyz[3 * p + 1],
                     dz = (double)xyz[3 * k + 2] - (double)xyz[3 * p + 2];
        double h[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) h[c] = fmax(fma(w0t[128 + c], dz, fma(w0t[64 + c], dy, fma(w0t[c], dx, b0[c]))), 0.0);
        double res[4] = {0.0, 0.0, 0.0, 0.0};
#ifdef DGDM_SA1_CLOCKS
        const long long tk1 = __builtin_readcyclecounter();
#endif
#pragma nounroll
        for (int f = 0; f < 128; ++f) {
            const double *wf = sa1_w + f * 64;               // wave-uniform: broadcast reads
            double acc = sa1_w[128 * 64 + f];
#pragma unroll
            for (int kk = 0; kk < 64; ++kk) acc = fma(wf[kk], h[kk], acc);
            // the group's maximum (32 lanes of one centre): four DPP steps inside the rows of 16, one exchange between the two rows
            acc = dpp_max64<0xB1>(acc); acc = dpp_max64<0x4E>(acc); acc = dpp_max64<0x141>(acc); acc = dpp_max64<0x140>(acc);
            acc = fmax(acc, __shfl_xor(acc, 16));
            acc = fmax(acc, 0.0);                            // ReLU outputs are >= 0 and the group is never empty
            if (sl == (f & 31)) res[f >> 5] = acc;
        }
#ifdef DGDM_SA1_CLOCKS
        const long long tk2 = __builtin_readcyclecounter();
        if (threadIdx.x == 0 && blockIdx.x == 100 && task < 4 * gridDim.x * 2) printf("sa1 task %d: ball + layer 0 %lld cycles, layer 1 %lld\n", task, tk1 - tk0, tk2 - tk1);
#endif
        if (half == 0 || 2 * pp + 1 < N) {
#pragma unroll
            for (int j = 0; j < 4; ++j) F1[(size_t)p * 128 + 32 * j + sl] = res[j];
        }
        __builtin_amdgcn_wave_barrier();                     // nbr is rewritten by the next task
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
    static bool attr_set = false;
    if (!attr_set) {
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)sa1_64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SA1_LDS_BYTES));
        attr_set = true;
    }
    const int tasks = nobj * ((N + 1) / 2);
    hipLaunchKernelGGL(sa1_64_kernel, dim3(std::min((tasks + 3) / 4, 512)), dim3(256), SA1_LDS_BYTES, s, xyz, N, nobj, r1sq, w.sa1_w0t, w.sa1_b0, w.sa1_w1,
                       w.sa1_b1, F1_64);
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
