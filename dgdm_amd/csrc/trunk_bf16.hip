// bf16-contraction variant of the fused dynamics trunk (BASELINE configs[4]: "bf16 contractions, f32 accumulate/statistics").
//
// Same contract as trunk_kernel (trunk.hip): same first-layer tables, same TrunkParams, same per-tile partial d/dz1 output,
// so everything around it (tables, dyn_post_kernel, DDIM step) is shared and stays float32.  What changes is the arithmetic
// of the contractions: weights (BatchNorm folded in float64, then rounded once to bf16, round-to-nearest-even) and the
// activations / gradients entering a contraction (rounded to bf16 by v_cvt_pk_bf16_f32) are multiplied on
// v_mfma_f32_32x32x16_bf16 and accumulated in float32; biases, the first-layer tables, the objective and every sum over rows
// are float32.  oracle/dgdm_oracle.py::trunk_grad_bf16 states the same rounding points on the CPU.
//
// Layout.  One wave owns TWO 32-row tiles (64 rows), so every weight operand read from L2 feeds two MFMAs.  An activation
// block (32 features x 32 rows) is 8 VGPRs of packed bf16 pairs:
//     dword d of block o, lane (n = l & 31, h = l >> 5)  =  { H[32 o + rho(2d, h)][n] , H[32 o + rho(2d+1, h)][n] },
//     rho(r, h) = (r & 3) + 8 (r >> 2) + 4 h                      (the 32x32 MFMA C/D row of accumulator register r)
// i.e. the accumulators of a layer, converted pairwise, ARE the next layer's B operand: K-step s of input block o is dwords
// 4s..4s+3.  The matching A-operand image (host: pack_chain_bf16) holds, for lane (i, h) and slot j of K-step s,
// W[32 o' + i][32 o + rho(8 s + j, h)].
//
// ReLU and its derivative work on the packed pairs: relu = v_pk_max_i16(x, 0) (a negative bf16 is a negative int16), the
// sign bits of the pre-activations are collected into one dword per (layer, block) per lane (bits 15-k and 31-k for pair
// k = d + 8 tile) and kept in LDS, and the backward pass turns them back into 16-bit masks (shift, v_pk_ashrrev_i16 15,
// and-not).  A pre-activation that rounds to -0.0/+0.0 follows its sign bit; torch's relu'(0) = 0 differs only there.
//
// Pipeline.  A "stage" is one output block: NK weight entries (1 KiB each, prefetched 16 ahead through the buffer-load ring
// of mfma_chain.h) x 2 tiles.  The convert/ReLU/mask epilogue of stage k runs in the first steps of stage k+1 (accumulators
// are double buffered), also across layer boundaries, so the matrix pipe does not wait for the VALU between blocks.
#include <type_traits>
#include "common.h"
#include "mfma_chain.h"
#include "trunk.h"

namespace dgdm {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf16(const float4 a, const u32x4_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Two 32-row tiles x 8 feature blocks of packed activations: v[tile][block][kstep] (4 dwords each)
struct Act16 {
    u32x4_t v[2][8][2];
};

// forward epilogue of one pair with shift SH = pair + 8 tile: convert, record the sign bits (mk starts at 0 per block), ReLU
template <int SH>
__device__ __forceinline__ uint32_t fwd_pair(float lo, float hi, uint32_t &mk) {
    const uint32_t p = pack_bf16(lo, hi);
    mk |= (p >> SH) & (0x80008000u >> SH);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, p), s16x2_t{0, 0}));
}

// backward epilogue: convert, zero the entries whose forward pre-activation had its sign bit set
template <int SH>
__device__ __forceinline__ uint32_t bwd_pair(float lo, float hi, uint32_t mk) {
    const uint32_t p = pack_bf16(lo, hi);
    const s16x2_t neg = __builtin_bit_cast(s16x2_t, mk << SH) >> 15;        // 0xffff where the sign bit was set
    return p & ~__builtin_bit_cast(uint32_t, neg);
}

// End of a pipeline step.  sched_barrier(0) stops the machine schedulers, but it is a no-memory intrinsic: instruction selection
// still moved the weight loads of a step below the MFMAs of the following steps (seen in front3d: eight MFMAs back to back, then
// seventeen loads).  The empty asm with a memory clobber orders the loads as well.
#define STEP_FENCE()                          \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        asm volatile("" ::: "memory");        \
    } while (0)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Epilogue work item Q (0..15) of a finished stage -> Y.v[tile][OB][pair / 4][pair % 4]
template <bool FWD, int OB, int Q>
__device__ __forceinline__ void epi_item(const f32x16 (&acc)[2], uint32_t &mk, Act16 &Y) {
    constexpr int T = (Q / 4) & 1, D = (Q & 3) + 4 * (Q / 8);      // order: K-step 0 dwords of both tiles first, then K-step 1
    constexpr int SH = D + 8 * T;                                    // bit position of the pair in the mask word
    if (FWD) Y.v[T][OB][D / 4][D % 4] = fwd_pair<SH>(acc[T][2 * D], acc[T][2 * D + 1], mk);
    else Y.v[T][OB][D / 4][D % 4] = bwd_pair<SH>(acc[T][2 * D], acc[T][2 * D + 1], mk);
}

// The 16 items of a stage are spread over the first min(NK, 16) steps of the following stage: with NK = 16 one item per step,
// i.e. ~7 VALU instructions (the accumulators sit in AGPRs, so each item starts with two v_accvgpr_read) beside the step's two
// MFMAs - measured on MI355X, a v_mfma_f32_32x32x16_bf16 hides at most 5 other instructions, and two items per step (11 VALU
// per MFMA) left the matrix pipe idle half of every stage.  Items (tile 0, pairs 0..3), (1, 0..3), (0, 4..7), (1, 4..7): the
// dwords of K-step 0 of a pending block are complete after step 7, those of K-step 1 at step 15, where they are first read.
template <bool FWD, int OB, int NK, int I>
__device__ __forceinline__ void epi_step(const f32x16 (&acc)[2], uint32_t &mk, Act16 &Y) {
    constexpr int STEPS = NK < 16 ? NK : 16, PER = 16 / STEPS;
    if constexpr (I < STEPS) {
        static_for<0, PER>([&](auto jc) { epi_item<FWD, OB, I * PER + decltype(jc)::value>(acc, mk, Y); });
    }
}

__device__ __forceinline__ void load_f32x16(const float *block, int h4, f32x16 &b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(block + 8 * q + h4);
        b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
    }
}

// State that crosses stage and layer boundaries.
struct Pipe {
    f32x16 acc[2][2];      // [parity][tile]; after a layer, acc[1] holds its block 7 with the epilogue still to do
    f32x16 bias0;          // forward: bias block the next stage starts its accumulators from (prefetched one stage ahead)
    uint32_t mk;           // backward: mask word of the pending block
};

// One layer Y = epi(W X (+ bias)) on both tiles: 8 output blocks, NK K-steps each (NK = 16: K = 256; NK = 2: K = 32).
// When PEND is set, the epilogue of the PREVIOUS layer's block 7 (pipe.acc[1] -> X block 7) is finished during stage 0 -
// X block 7 is first read in the last two steps of a stage.  This layer's own block 7 is left pending the same way.
//   FWD : C starts from the bias block; epilogue = convert + sign bits (stored to smask[slot0 + block]) + ReLU;
//         the pending block's sign bits go to smask[prev_slot].
//   !FWD: C starts from zero; epilogue = convert + mask smask[slot0 + block] (pending block: pipe.mk).
// `woff` is the byte offset of this layer's 8 NK entries in the weight stream.
template <bool FWD, bool PEND, int NK>
__device__ __forceinline__ void layer16(const wrsrc_t rs, const int voff, const int woff, float4 (&ring)[CONT_DEPTH], const float *bias,
                                        const float *bias_next, Act16 &X, Act16 &Y, Pipe &pipe, uint32_t (*smask)[256], const int prev_slot,
                                        const int slot0, const int tid, const int h4) {
    constexpr int EA = (NK < 16 ? NK : 16) - 1;      // last step that carries epilogue items
    constexpr int BS = NK > 9 ? 9 : NK - 1;          // step that prefetches the next stage's bias block
    constexpr int MS = NK > 8 ? 8 : NK - 1;          // backward: step that fetches the mask word of this stage's own epilogue
    uint32_t mk = FWD ? 0u : pipe.mk, mkn = 0u;
    static_for<0, 8>([&](auto opc) {
        constexpr int OP = decltype(opc)::value;
        constexpr int PAR = OP & 1;
        static_for<0, NK>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int E = OP * NK + I;
            constexpr int IB = I / 2, S = I % 2;
            const float4 a = ring[E % CONT_DEPTH];
            ring[E % CONT_DEPTH] = wload(rs, voff, woff + (E + CONT_DEPTH) * 1024);
            if constexpr (I == 0) {
                if (FWD) mk = 0u;
            }
            if constexpr (OP == 0) {
                if constexpr (PEND) epi_step<FWD, 7, NK, I>(pipe.acc[1], mk, X);
            } else {
                epi_step<FWD, OP - 1, NK, I>(pipe.acc[PAR ^ 1], mk, Y);
            }
            if constexpr (I == EA) {
                if (FWD) {                            // the previous stage's sign-bit word is complete
                    if constexpr (OP == 0) { if (PEND) smask[prev_slot][tid] = mk; }
                    else smask[slot0 + OP - 1][tid] = mk;
                }
            }
            if constexpr (!FWD && I == MS) mkn = smask[slot0 + OP][tid];     // for this stage's own epilogue, which runs in the next stage
            if constexpr (!FWD && I == NK - 1) mk = mkn;
            if constexpr (I == 0) {
                if (FWD) {
                    pipe.acc[PAR][0] = mfma_bf16(a, X.v[0][IB][S], pipe.bias0);
                    pipe.acc[PAR][1] = mfma_bf16(a, X.v[1][IB][S], pipe.bias0);
                } else {
                    f32x16 zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                    pipe.acc[PAR][0] = mfma_bf16(a, X.v[0][IB][S], zero);
                    pipe.acc[PAR][1] = mfma_bf16(a, X.v[1][IB][S], zero);
                }
            } else {
                pipe.acc[PAR][0] = mfma_bf16(a, X.v[0][IB][S], pipe.acc[PAR][0]);
                pipe.acc[PAR][1] = mfma_bf16(a, X.v[1][IB][S], pipe.acc[PAR][1]);
            }
            if constexpr (FWD && I == BS) load_f32x16(OP < 7 ? bias + 32 * (OP + 1) : bias_next, h4, pipe.bias0);
            STEP_FENCE();
        });
    });
    pipe.mk = mk;
}

// Finish a pending block 7 outside a layer (used before code that is not a layer16).
template <bool FWD>
__device__ __forceinline__ void finish_pending(Act16 &X, Pipe &pipe, uint32_t (*smask)[256], int prev_slot, int tid) {
    uint32_t mk = FWD ? 0u : pipe.mk;
    static_for<0, 16>([&](auto qc) { epi_item<FWD, 7, decltype(qc)::value>(pipe.acc[1], mk, X); });
    if (FWD) smask[prev_slot][tid] = mk;
}

// Per-tile row bookkeeping (same mapping as trunk_kernel: a tile is one finger of one chain against 32 pose cells)
struct TileRows {
    int chain, b;
    bool valid;
    int64_t r;
    const float *arow;
    const float4 *ptile;      // this tile's 32 cells in the tiled pose table, + lane
};

__device__ __forceinline__ TileRows tile_rows(const TrunkParams &p, int tile, int n, int lane, int W1) {
    TileRows t;
    const int per_chain = p.B * p.tiles_per_b;
    t.chain = tile / per_chain;
    const int rem = tile - t.chain * per_chain;
    t.b = rem / p.tiles_per_b;
    const int c = (rem - t.b * p.tiles_per_b) * 32 + n;
    t.valid = c < p.C;
    const int cc = t.valid ? c : p.C - 1;
    t.r = (int64_t)cc * p.B + t.b;
    t.arow = p.Atab + (size_t)(t.chain * p.B + t.b) * W1;
    t.ptile = reinterpret_cast<const float4 *>(p.PtabT) + (size_t)(rem - t.b * p.tiles_per_b) * (W1 / 32) * 4 * 64 + lane;
    return t;
}

// z1 block `blk` of a tile from the first-layer tables (float32), in two halves so that the loads of the next block can be put
// in flight before the arithmetic of the current one (left to itself hipcc emits load, load, s_waitcnt vmcnt(0), 64 times over:
// 36 k cycles of exposed L2 latency per wave)
struct TabRaw {
    float4 a[4], w[4];
};

__device__ __forceinline__ void table_load(const TileRows &t, int blk, int h4, TabRaw &raw) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        raw.a[q] = *reinterpret_cast<const float4 *>(t.arow + 32 * blk + 8 * q + h4);
        raw.w[q] = t.ptile[(blk * 4 + q) * 64];                                // coalesced: 1 KiB per (block, quarter)
    }
}

__device__ __forceinline__ void table_sum(const TabRaw &raw, f32x16 &z) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        z[4 * q + 0] = raw.a[q].x + raw.w[q].x; z[4 * q + 1] = raw.a[q].y + raw.w[q].y;
        z[4 * q + 2] = raw.a[q].z + raw.w[q].z; z[4 * q + 3] = raw.a[q].w + raw.w[q].w;
    }
}

// ---- the front's weight stream through LDS, shared by the workgroup's four waves.
// In layers 1-2 of the 3-D model a weight entry feeds ONE MFMA (the phase runs per tile), i.e. 1 KiB per 32 cycles and wave = 128 B/clk/CU
// against the 64 B/clk a CU's L1 delivers: the phase ran at 1 765 cycles per 16-entry block against 512 of MFMA issue.  All four waves
// read the SAME 512-entry stream, twice (once per tile): each wave loads a quarter of every 16-entry chunk (16 KiB) from L2 and puts it
// into one of three LDS chunk buffers, and every wave takes its A operands from there (ds_read_b128: 256 B/clk/CU).  Chunk cc is written
// two chunks ahead of its use (iteration cc - 2, before that iteration's MFMAs), one workgroup barrier per chunk makes it visible, and the
// 16-entry register ring is refilled from chunk cc + 1 while chunk cc is consumed - the same ring discipline as with buffer loads.
// The two tile passes are one sequence of 64 chunks; the last refill comes from global memory again (the head of the stack's stream).
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4v_t lds_f32x4_t;      // keeps ds_read / ds_write (a generic pointer would turn into flat accesses)
constexpr int FRONT_CHUNKS = 64;                                      // 2 passes x 32 chunks of 16 entries

#ifndef DGDM_FRONT_DEPTH
#define DGDM_FRONT_DEPTH 1
#endif
constexpr int FSD = DGDM_FRONT_DEPTH;      // chunks a quarter-chunk load is given to arrive beyond the first (measured: 2-4 are no faster, and spill)

struct FrontStage {
    float4 q[FSD][4];              // this wave's quarter of chunks cc + 2 .. cc + 1 + FSD, in flight from L2; chunk k sits in slot (k - 2) % FSD
};

__device__ __forceinline__ void front_stage_load(const wrsrc_t rs, const int voff, const int wave, const int cc, float4 (&q)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = wload(rs, voff, (((cc % 32) * 16) + 4 * wave + j) * 1024);
}

__device__ __forceinline__ void front_stage_store(lds_f32x4_t *wb, const int wave, const int cc, const float4 (&q)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) wb[(((cc % 3) * 16) + 4 * wave + j) * 64] = __builtin_bit_cast(f32x4v_t, q[j]);
}

// start of chunk CC's MFMAs: chunk CC + 2 goes to LDS, chunk CC + 2 + FSD is requested into the slot that frees
template <int CC>
__device__ __forceinline__ void front_chunk_begin(const wrsrc_t rs, const int voff, const int wave, lds_f32x4_t *wb, FrontStage &st) {
    if constexpr (CC + 2 < FRONT_CHUNKS) front_stage_store(wb, wave, CC + 2, st.q[CC % FSD]);
    if constexpr (CC + 2 + FSD < FRONT_CHUNKS) front_stage_load(rs, voff, wave, CC + 2 + FSD, st.q[CC % FSD]);
}

// end of chunk CC: everyone has written its part of chunk CC + 2 and finished reading chunk CC + 1's predecessor.  A bare s_barrier
// behind an LDS-only wait: __syncthreads() would also wait for the global loads just requested (vmcnt(0)) and so undo the prefetch.
__device__ __forceinline__ void front_chunk_end() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The A operands of the front come out of LDS through an 8-entry register ring in CONSUMPTION order (LDS latency is ~100 cycles: eight
// steps of cover are plenty, and the other half of the usual 16-entry ring pays for the second accumulator below).  Consumption order
// within a chunk: a layer-1 chunk (one output block, 16 K-steps) as stored; a layer-2 chunk (8 output blocks x 2 K-steps, stored block-
// major) K-step-major, so that consecutive MFMAs never hit the same accumulator - a dependent v_mfma_f32_32x32x16_bf16 issues only every
// ~68 cycles, which is what this phase was really bound by (1 560 cycles per 16-MFMA chunk with the weights already in LDS).
// Chunk sequence of a pass: z(0), z(1), l2(0), z(2), l2(1), ..., z(15), l2(14), l2(15).
#ifdef DGDM_FRONT_STAMPS
__device__ long long g_front_stamps[16];
#define FSTAMP(i) do { if (T == 0 && KB == 5 && blockIdx.x == gridDim.x / 2 && tid == 0) g_front_stamps[i] = clock64(); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

constexpr bool front_is_l2(int cc) { return (cc % 32) == 31 || ((cc % 32) >= 2 && (cc % 32) % 2 == 0); }
constexpr int front_entry(int cc, int j) { return front_is_l2(cc) ? 2 * (j % 8) + j / 8 : j; }

struct FrontRing {
    float4 r[8];
};

__device__ __forceinline__ void front_ring_fill(FrontRing &fr, lds_f32x4_t *wb) {           // chunk 0 (a layer-1 chunk), steps 0..7
#pragma unroll
    for (int j = 0; j < 8; ++j) fr.r[j] = __builtin_bit_cast(float4, wb[j * 64]);
}

// step J of chunk CC: hands out its entry and requests the one eight steps on
template <int CC, int J>
__device__ __forceinline__ float4 front_take(FrontRing &fr, lds_f32x4_t *wb) {
    constexpr int N = 16 * CC + J, M = N + 8, CM = M / 16, JM = M % 16;
    const float4 a = fr.r[N % 8];
    if constexpr (CM < FRONT_CHUNKS) fr.r[N % 8] = __builtin_bit_cast(float4, wb[(((CM % 3) * 16) + front_entry(CM, JM)) * 64]);
    return a;
}

// 3-D layers 1 and 2 of ONE tile T (the 512-wide layer 1 does not fit in registers for two tiles at once):
//   for each 32-feature block kb of layer 1:  z = W1o'[kb] bf(xobj) + (Atab + Ptab)[kb] ; a1 = relu ; acc2 += W2'[:, kb] bf(a1)
// software-pipelined by one block: the stream order is z(0), z(1), l2(0), z(2), l2(1), ..., z(15), l2(14), l2(15) (16 entries each).
// Leaves layer 2's blocks 0..6 packed in X.v[T] and block 7 pending in pipe.acc[1][T].
template <int T>
__device__ __forceinline__ void front3d(const wrsrc_t rs, const int voff, float4 (&ring)[CONT_DEPTH], const TrunkParams &p, const TileRows &tr,
                                        Act16 &X, Pipe &pipe, uint32_t (*smask)[256], const int tid, const int h4, const int wave, lds_f32x4_t *wb,
                                        FrontStage &st, FrontRing &fr) {
    // first z1 block of the tables and the xobj row (all 32 loads in flight together), then xobj -> packed B operand
    TabRaw raw[2];
    table_load(tr, 0, h4, raw[0]);
    u32x4_t xin[8][2];
    const float *xrow = p.xobj + ((size_t)tr.chain * p.xstride + tr.r) * 256;
    f32x16 acc2[8];
    if (p.xobj16 || p.xtab16) {
        // the embedding was produced in bf16 operand order (pointnet.hip xobj kernels / embedding table): the row IS the B operand
        const uint32_t *r16 = p.xtab16 ? p.xtab16[tr.chain] + (size_t)p.xidx[(size_t)tr.chain * p.xstride + tr.r] * 128
                                       : p.xobj16 + ((size_t)tr.chain * p.xstride + tr.r) * 128;
        const u32x4_t *row16 = reinterpret_cast<const u32x4_t *>(r16) + (h4 >> 1);
#pragma unroll
        for (int o = 0; o < 8; ++o) { xin[o][0] = row16[4 * o]; xin[o][1] = row16[4 * o + 1]; }
#pragma unroll
        for (int o = 0; o < 8; ++o) load_f32x16(p.b2 + 32 * o, h4, acc2[o]);
    } else {
        float4 xv[8][4];
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[o][q] = *reinterpret_cast<const float4 *>(xrow + 32 * o + 8 * q + h4);
#pragma unroll
        for (int o = 0; o < 8; ++o) load_f32x16(p.b2 + 32 * o, h4, acc2[o]);
        STEP_FENCE();
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xin[o][q / 2][(2 * q) % 4] = pack_bf16(xv[o][q].x, xv[o][q].y);
                xin[o][q / 2][(2 * q + 1) % 4] = pack_bf16(xv[o][q].z, xv[o][q].w);
            }
        }
    }
    // a layer-1 block accumulates on TWO accumulators (even / odd K-steps): the MFMAs of one accumulator are dependent
    f32x16 zA[2], zB[2];
    u32x4_t zin[2];
    uint32_t mk = 0;
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    static_for<0, 17>([&](auto kc) {
        constexpr int KB = decltype(kc)::value;          // z(KB) for KB < 16, then l2(KB - 1) for KB >= 1
        if constexpr (KB < 16) {
            f32x16 init;
            table_sum(raw[KB & 1], init);
            constexpr int CZ = 32 * T + (KB == 0 ? 0 : 2 * KB - 1);          // chunk number of z(KB) in the two-pass sequence
            FSTAMP(0);
            front_chunk_begin<CZ>(rs, voff, wave, wb, st);
            FSTAMP(1);
            static_for<0, 16>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                const float4 a = front_take<CZ, I>(fr, wb);
                if constexpr (KB > 0 && I == 0) mk = 0u;
                if constexpr (KB > 0 && I < 8) {          // epilogue of z(KB-1): one pair per step
                    constexpr int Q = (KB - 1) & 1;
#ifdef DGDM_FRONT_TWO_ACC
                    zin[I / 4][I % 4] = fwd_pair<I + 8 * T>(zA[Q][2 * I] + zB[Q][2 * I], zA[Q][2 * I + 1] + zB[Q][2 * I + 1], mk);
#else
                    zin[I / 4][I % 4] = fwd_pair<I + 8 * T>(zA[Q][2 * I], zA[Q][2 * I + 1], mk);
#endif
                }
#ifdef DGDM_FRONT_TWO_ACC
                if constexpr (I == 0) zA[KB & 1] = mfma_bf16(a, xin[0][0], init);
                else if constexpr (I == 1) zB[KB & 1] = mfma_bf16(a, xin[0][1], zero);
                else if constexpr (I % 2 == 0) zA[KB & 1] = mfma_bf16(a, xin[I / 2][0], zA[KB & 1]);
                else zB[KB & 1] = mfma_bf16(a, xin[I / 2][1], zB[KB & 1]);
#else
                if constexpr (I == 0) zA[KB & 1] = mfma_bf16(a, xin[0][0], init);
                else zA[KB & 1] = mfma_bf16(a, xin[I / 2][I % 2], zA[KB & 1]);
#endif
                if constexpr (I == 8 && KB < 15) table_load(tr, KB + 1, h4, raw[(KB + 1) & 1]);     // next block's table part
                STEP_FENCE();
                if constexpr (I == 7) FSTAMP(2);
            });
            FSTAMP(3);
            front_chunk_end();
            FSTAMP(4);
        } else {
            mk = 0u;
            static_for<0, 8>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
#ifdef DGDM_FRONT_TWO_ACC
                zin[I / 4][I % 4] = fwd_pair<I + 8 * T>(zA[1][2 * I] + zB[1][2 * I], zA[1][2 * I + 1] + zB[1][2 * I + 1], mk);
#else
                zin[I / 4][I % 4] = fwd_pair<I + 8 * T>(zA[1][2 * I], zA[1][2 * I + 1], mk);
#endif
            });
        }
        if constexpr (KB >= 1) {
            constexpr int KP = KB - 1;                   // l2(KP): uses zin = a1 block KP
            // sign bits of a1 block KP: this tile's 8 pairs; the two tile passes share the word
            if (T == 0) smask[KP][tid] = mk;
            else (void)__hip_atomic_fetch_or(&smask[KP][tid], mk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // ds_or_b32, no read-back
            constexpr int CL = 32 * T + (KP == 15 ? 31 : 2 * KP + 2);        // chunk number of l2(KP)
            front_chunk_begin<CL>(rs, voff, wave, wb, st);
            FSTAMP(5);
            static_for<0, 16>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                constexpr int OP = J % 8, S = J / 8;       // K-step-major: eight different accumulators in a row
                const float4 a = front_take<CL, J>(fr, wb);
                acc2[OP] = mfma_bf16(a, zin[S], acc2[OP]);
                STEP_FENCE();
            });
            FSTAMP(6);
            front_chunk_end();
            FSTAMP(7);
        }
    });
    if (T == 1) ring_fill(rs, voff, 512 * 1024, ring);       // the stack's first 16 entries, in flight during the epilogue below
    // layer 2 epilogue: blocks 0..6 here, block 7 left pending
    static_for<0, 7>([&](auto oc) {
        constexpr int O = decltype(oc)::value;
        uint32_t m2 = 0;
        static_for<0, 8>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            X.v[T][O][D / 4][D % 4] = fwd_pair<D + 8 * T>(acc2[O][2 * D], acc2[O][2 * D + 1], m2);
        });
        if (T == 0) smask[16 + O][tid] = m2;
        else (void)__hip_atomic_fetch_or(&smask[16 + O][tid], m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    });
    pipe.acc[1][T] = acc2[7];
}

template <int KIND>
__global__ __launch_bounds__(256, 1) void trunk_bf16_kernel(const TrunkParams p) {
    constexpr int W1B = (KIND == 3) ? 16 : 8;
    constexpr int W1 = W1B * 32;
    constexpr int S_MID = (KIND == 3) ? 24 : 8;           // mask slot of mid layer 0's output
    constexpr int NSLOT = S_MID + 8 * ((KIND == 3) ? 6 : 7);
    __shared__ uint32_t smask[NSLOT][256];
    __shared__ __attribute__((aligned(16))) float red[4][32][36];       // per wave: one 32-feature block x 32 rows (+pad) for the final fold
    __shared__ f32x4v_t wbuf[KIND == 3 ? 3 : 1][KIND == 3 ? 16 : 1][64];  // 3-D front: three 16-entry chunks of the weight stream (above)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile0 = (blockIdx.x * 4 + wave) * 2;
    // 2-D: a wave without a tile leaves (no barrier anywhere).  3-D: the front's weight stream is staged by all four waves together,
    // so a wave without a tile goes through the front on the last tile's rows (nothing it computes is stored) and leaves after it.
    const bool active = tile0 < p.ntiles;
    if (KIND != 3 && !active) return;
    if (!active) tile0 = (p.ntiles - 1) & ~1;
    const bool has1 = tile0 + 1 < p.ntiles;               // an odd tile count leaves the last wave one real tile
    const int n = lane & 31;
    const int h4 = (lane >> 5) * 4;
    const int voff = lane * 16;

    const TileRows tr0 = tile_rows(p, tile0, n, lane, W1);
    const TileRows tr1 = tile_rows(p, has1 ? tile0 + 1 : tile0, n, lane, W1);

#ifdef DGDM_TRUNK_CLOCKS
    long long tstamp[8];
    int nts = 0;
#define STAMP() tstamp[nts++] = __builtin_readcyclecounter()
#else
#define STAMP()
#endif
    STAMP();
    Act16 X, Y;
    Pipe pipe;
    pipe.mk = 0;
    load_f32x16(p.bf[0], h4, pipe.bias0);
    const wrsrc_t rsF = weight_rsrc(p.Wfwd, p.fwd_bytes);
    float4 ring[CONT_DEPTH];
    if (KIND == 2) ring_fill(rsF, voff, 0, ring);          // 3-D: the front takes its entries out of LDS and fills the ring when it ends
    int woff = 0;

    if (KIND == 2) {
        // ---- layer 1 from the tables; blocks 0..6 finished here, block 7 left pending like any other layer's
        TabRaw raw[2][2];                                  // [parity][tile]: block O+1 is in flight while block O is converted
        table_load(tr0, 0, h4, raw[0][0]);
        table_load(tr1, 0, h4, raw[0][1]);
        static_for<0, 8>([&](auto oc) {
            constexpr int O = decltype(oc)::value;
            if constexpr (O < 7) {
                table_load(tr0, O + 1, h4, raw[(O + 1) & 1][0]);
                table_load(tr1, O + 1, h4, raw[(O + 1) & 1][1]);
            }
            STEP_FENCE();
            f32x16 z0, z1;
            table_sum(raw[O & 1][0], z0);
            table_sum(raw[O & 1][1], z1);
            if constexpr (O < 7) {
                uint32_t mk = 0;
                static_for<0, 8>([&](auto dc) {
                    constexpr int D = decltype(dc)::value;
                    X.v[0][O][D / 4][D % 4] = fwd_pair<D>(z0[2 * D], z0[2 * D + 1], mk);
                });
                static_for<0, 8>([&](auto dc) {
                    constexpr int D = decltype(dc)::value;
                    X.v[1][O][D / 4][D % 4] = fwd_pair<D + 8>(z1[2 * D], z1[2 * D + 1], mk);
                });
                smask[O][tid] = mk;
            } else {
                pipe.acc[1][0] = z0;
                pipe.acc[1][1] = z1;
            }
            STEP_FENCE();
        });
    } else {
        // chunks 0 and 1 of the stream into LDS, chunk 2 requested
        lds_f32x4_t *wb = (lds_f32x4_t *)&wbuf[0][0][lane];
        FrontStage st;
        FrontRing fr;
        front_stage_load(rsF, voff, wave, 0, st.q[0]);
        front_stage_store(wb, wave, 0, st.q[0]);
        front_stage_load(rsF, voff, wave, 1, st.q[0]);
        front_stage_store(wb, wave, 1, st.q[0]);
#pragma unroll
        for (int i = 0; i < FSD; ++i) front_stage_load(rsF, voff, wave, 2 + i, st.q[i]);
        front_chunk_end();
        front_ring_fill(fr, wb);
        front3d<0>(rsF, voff, ring, p, tr0, X, pipe, smask, tid, h4, wave, wb, st, fr);
        front3d<1>(rsF, voff, ring, p, tr1, X, pipe, smask, tid, h4, wave, wb, st, fr);
        woff = 512 * 1024;
        if (!active) return;
    }
    int pend_slot = (KIND == 3) ? 16 + 7 : 7;
    STAMP();

    // ---- 256 -> 256 layers, two per iteration (X -> Y -> X)
    int l = 0;
    for (; l + 1 < p.n_mid; l += 2) {
        layer16<true, true, 16>(rsF, voff, woff, ring, p.bf[l], p.bf[l + 1], X, Y, pipe, smask, pend_slot, S_MID + 8 * l, tid, h4);
        woff += 128 * 1024;
        layer16<true, true, 16>(rsF, voff, woff, ring, p.bf[l + 1], p.bf[l + 2 < p.n_mid ? l + 2 : l + 1], Y, X, pipe, smask, S_MID + 8 * l + 7, S_MID + 8 * (l + 1), tid, h4);
        woff += 128 * 1024;
        pend_slot = S_MID + 8 * (l + 1) + 7;
    }
    if (KIND == 2) {                                       // n_mid = 7: one more, result in Y
        layer16<true, true, 16>(rsF, voff, woff, ring, p.bf[l], p.bf[l], X, Y, pipe, smask, pend_slot, S_MID + 8 * l, tid, h4);
        woff += 128 * 1024;
        pend_slot = S_MID + 8 * l + 7;
    }
    Act16 &H = (KIND == 2) ? Y : X;                        // a_8, block 7 pending
    STAMP();

    // ---- output layer 256 -> 3 (padded to one 32-row block): 16 entries; finishes the pending block on the way
    f32x16 lo[2];
    {
        uint32_t mk = 0;
        static_for<0, 16>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            const float4 a = ring[I];
            ring[I] = wload(rsF, voff, woff + (I + CONT_DEPTH) * 1024);   // past the end of the stream: clipped to zero, unused
            epi_step<true, 7, 16, I>(pipe.acc[1], mk, H);
            if constexpr (I == 15) smask[pend_slot][tid] = mk;
            if constexpr (I == 0) {
                f32x16 zero;
#pragma unroll
                for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                lo[0] = mfma_bf16(a, H.v[0][0][0], zero);
                lo[1] = mfma_bf16(a, H.v[1][0][0], zero);
            } else {
                lo[0] = mfma_bf16(a, H.v[0][I / 2][I % 2], lo[0]);
                lo[1] = mfma_bf16(a, H.v[1][I / 2][I % 2], lo[1]);
            }
            STEP_FENCE();
        });
    }

    // ---- objective gradient per row (float32), packed as the B operand of the transposed output layer
    const wrsrc_t rsB = weight_rsrc(p.Wbwd, p.bwd_bytes);
    ring_fill(rsB, voff, 0, ring);
    woff = 0;
    Act16 &GA = (KIND == 2) ? X : Y;                       // whichever of X/Y is not H
    Act16 &GB = H;
    {
        const float b0 = p.bout[0], b1 = p.bout[1], b2 = p.bout[2];
        const TileRows *trs[2] = {&tr0, &tr1};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const TileRows &tr = *trs[t];
            const TrunkObjective ob = p.obj[tr.chain];
            const float d0 = lo[t][0] + b0, d1 = lo[t][1] + b1, d2 = lo[t][2] + b2;     // rows 0..2 of the block live in lanes h = 0
            float g0 = ob.lin[0] + 2.f * ob.quad[0] * d0;
            float g1 = ob.lin[1] + 2.f * ob.quad[1] * d1;
            float g2 = ob.lin[2] + 2.f * ob.quad[2] * d2;
            if (ob.use_rowcoef) g0 = p.rowcoef[(size_t)tr.chain * p.R + tr.r];
            const bool live = tr.valid && lane < 32 && (t == 0 || has1);
            if (!live) { g0 = 0.f; g1 = 0.f; g2 = 0.f; }
            GB.v[t][0][0][0] = pack_bf16(g0, g1);
            GB.v[t][0][0][1] = pack_bf16(g2, 0.f);
            GB.v[t][0][0][2] = 0u; GB.v[t][0][0][3] = 0u;
            GB.v[t][0][1] = u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    STAMP();
    // ---- transposed output layer: 8 blocks x 2 K-steps; masks of a_8
    layer16<false, false, 2>(rsB, voff, woff, ring, nullptr, nullptr, GB, GA, pipe, smask, 0, S_MID + 8 * (p.n_mid - 1), tid, h4);
    woff += 16 * 1024;

    STAMP();
    // ---- backward through the mid layers, two per iteration (GA -> GB -> GA); layer j masks with the output of layer j-1
    for (int j = p.n_mid - 1; j >= ((KIND == 2) ? 2 : 1); j -= 2) {
        layer16<false, true, 16>(rsB, voff, woff, ring, nullptr, nullptr, GA, GB, pipe, smask, 0, S_MID + 8 * (j - 1), tid, h4);
        woff += 128 * 1024;
        layer16<false, true, 16>(rsB, voff, woff, ring, nullptr, nullptr, GB, GA, pipe, smask, 0, S_MID + 8 * (j - 2), tid, h4);
        woff += 128 * 1024;
    }

    STAMP();
    // ---- last layer back (2-D: W2'^T, 8 blocks; 3-D: W2'^T onto the 512-wide layer 1, 16 blocks): float32 epilogue,
    //      mask of a_1, fold of the tile's 32 cells, one partial vector per tile
    const bool same_b = has1 && tr0.chain == tr1.chain && tr0.b == tr1.b;
    float *dst0 = p.partial + (size_t)tile0 * W1;
    float *dst1 = dst0 + W1;
    static_for<0, W1B>([&](auto oc) {
        constexpr int OB = decltype(oc)::value;
        f32x16 g[2];
        const uint32_t mk1 = smask[OB][tid];
        static_for<0, 16>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int E = OB * 16 + I;
            const float4 a = ring[E % CONT_DEPTH];
            ring[E % CONT_DEPTH] = wload(rsB, voff, woff + (E + CONT_DEPTH) * 1024);
            if constexpr (OB == 0) epi_step<false, 7, 16, I>(pipe.acc[1], pipe.mk, GA);
            if constexpr (I == 0) {
                f32x16 zero;
#pragma unroll
                for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                g[0] = mfma_bf16(a, GA.v[0][0][0], zero);
                g[1] = mfma_bf16(a, GA.v[1][0][0], zero);
            } else {
                g[0] = mfma_bf16(a, GA.v[0][I / 2][I % 2], g[0]);
                g[1] = mfma_bf16(a, GA.v[1][I / 2][I % 2], g[1]);
            }
            STEP_FENCE();
        });
        // mask: pair d of tile t has its sign bits at 15 - (d + 8t) (low half, register 2d) and 31 - (d + 8t) (register 2d+1)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = r >> 1, hi = r & 1;
            if ((mk1 >> (15 - d + 16 * hi)) & 1u) g[0][r] = 0.f;
            if ((mk1 >> (15 - (d + 8) + 16 * hi)) & 1u) g[1][r] = 0.f;
        }
        // fold of the 32 cells of a tile: through LDS, transposed - lane (feature f, half) adds 16 rows of its feature, one
        // cross-half add finishes it (36 instructions per block instead of 5 shuffle+add rounds on each of 16-32 registers),
        // and the result leaves as one coalesced 128-byte store per tile and block.
        float (*rw)[36] = red[wave];
        const int fl = lane & 31, hf = lane >> 5;
        auto fold = [&](const f32x16 &v, float *dst) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rw[(r & 3) + 8 * (r >> 2) + h4][n] = v[r];
            __builtin_amdgcn_wave_barrier();
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 q4 = *reinterpret_cast<const float4 *>(&rw[fl][16 * hf + 4 * j]);
                s += (q4.x + q4.y) + (q4.z + q4.w);
            }
            s += __shfl_xor(s, 32);
            __builtin_amdgcn_wave_barrier();
            if (lane < 32) dst[32 * OB + fl] = s;
        };
        if (same_b) {
            // both tiles belong to the same finger: dyn_post_kernel adds their partials anyway, so add first, fold once
            f32x16 both;
#pragma unroll
            for (int r = 0; r < 16; ++r) both[r] = g[0][r] + g[1][r];
            fold(both, dst0);
            if (lane < 32) dst1[32 * OB + fl] = 0.f;
        } else {
            fold(g[0], dst0);
            if (has1) fold(g[1], dst1);
        }
    });
    STAMP();
#ifdef DGDM_TRUNK_CLOCKS
    if (p.clk && lane == 0)
        for (int i = 0; i < 7; ++i) p.clk[(size_t)(blockIdx.x * 4 + wave) * 8 + i] = tstamp[i];
#endif
}

int trunk_bf16_launch(int kind, const TrunkParams &p, hipStream_t s) {
    if (p.n_mid != (kind == 3 ? 6 : 7)) return DGDM_EINVAL;
    const int waves = (p.ntiles + 1) / 2, grid = (waves + 3) / 4;
    if (grid == 0) return DGDM_OK;
    const double rows = (double)(p.ntiles / (p.tiles_per_b > 0 ? p.tiles_per_b : 1)) * p.C;      // real rows, not the 32-row tile padding
    const double mid = 2.0 * 256 * 256 * p.n_mid;
    const double per_row = (kind == 3) ? (2.0 * 256 * 512 * 2 + mid) + (2.0 * 256 * 512 + mid) : 2.0 * mid;
    prof_begin(s, DGDM_STAGE_TRUNK);
    if (kind == 2) hipLaunchKernelGGL((trunk_bf16_kernel<2>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((trunk_bf16_kernel<3>), dim3(grid), dim3(256), 0, s, p);
    DGDM_HIP_CHECK(hipGetLastError());
#ifdef DGDM_FRONT_STAMPS
    if (kind == 3) {
        long long st[16];
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_front_stamps), sizeof(st));
        fprintf(stderr, "front stamps (z(5) chunk: begin-staging %lld, steps 0-7 %lld, steps 8-15 %lld, barrier %lld | l2(4) chunk: staging %lld, 16 steps %lld, barrier %lld)\n",
                st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6]);
    }
#endif
    prof_end(s, DGDM_STAGE_TRUNK, rows * per_row);
    return DGDM_OK;
}

}  // namespace dgdm
