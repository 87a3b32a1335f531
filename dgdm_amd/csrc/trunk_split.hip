// Fused dynamics trunk, forward + input-gradient backward (same contract as trunk_kernel, trunk.hip), with every float32 contraction
// carried by the 16-bit matrix pipe:  x = x_h + x_m + x_l  EXACTLY, three bf16 pieces (8 + 8 + 8 significant bits), the same for the
// weights, and of the nine piece products the six that matter (hh, hm, mh, mm, hl, lh; the rest are below 2^-24 of the product) go
// through v_mfma_f32_32x32x16_bf16 with float32 accumulation.  A bf16 x bf16 product is exact in float32, so what is lost per product
// is 2^-24 - the size of one float32 rounding - while the accumulation chain of an output shrinks from 128 dependent float32 fmas
// (v_mfma_f32_32x32x2_f32, K = 256) to 16 K-steps.  Measured against float64 (scripts/micro/split_mfma.hip, K = 256, He-init
// weights, post-ReLU inputs): rms error 1.6e-7 of the rms output for this form, 2.0e-7 for the float32 MFMA chain, 2.5e-8 for a
// single rounding of the exact result; bias of the matrix pipe's internal sum: -8e-9.  It is float32-grade arithmetic at six 32-cycle
// instructions per 16 features instead of eight 64-cycle ones: 2.7x fewer matrix-pipe cycles.
//
// One wave = one tile of 32 rows through all layers, forward and backward, in registers:
//   Y [8] f32x16   a layer's output in the MFMA C/D layout (register r, lane (n, h)  <->  feature 32 o + rho(r, h) of row n)
//   X [3][8][2]    the layer's input as three sets of packed bf16 B operands: K-step s of block o = registers 8s..8s+7 of Y[o],
//                  converted pairwise (trunk_bf16.hip's operand order; the weight images are pack_chain_bf16 images of the pieces)
// A 256 -> 256 layer is input-streaming (stream_layer): the K loop runs over the input blocks while all eight output blocks accumulate,
// twelve MFMAs per (K-step, pair of output blocks), consecutive MFMAs never on the same accumulator; the previous layer's epilogue -
// ReLU + sign bits (exact float32 semantics, x > 0; bits kept in LDS for the backward pass), then the split of an accumulator pair
// into the three pieces (x - bf16(x) is exact in float32, so the pieces reproduce x bit for bit) - is done just in time, one register
// pair per group of twelve MFMAs, in the shadow of the matrix pipe.  The 512-wide first layers of the 3-D model and the last layer
// back are produced block by block (split_block_out).
// Weight streams (host: Split3 / split_layer_stream, models_api.hip): per (K-step, block pair) six 1 KiB entries [A.h A.m A.l B.h B.m
// B.l] in consumption order, read through a 12-entry buffer-load ring.
#include "common.h"
#include <algorithm>
#include "mfma_chain.h"
#include "trunk.h"

namespace dgdm {

typedef __bf16 sbf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t su32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) su32x4_t lds_su32x4_t;      // an LDS pointer must keep its address space: a generic one turns
                                                                       // the reads into flat loads, which also wait on the weight ring

__device__ __forceinline__ f32x16 smfma(const float4 a, const su32x4_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sbf16x8_t, a), __builtin_bit_cast(sbf16x8_t, b), c, 0, 0, 0);
}

struct Act3 {
    su32x4_t v[3][8][2];      // [piece][32-feature block][K-step]: the B operands of one 32-row tile
};

// one pair of accumulator registers -> the three packed pieces (h | m | l), exactly: lo + hi == sum of the pieces' values
__device__ __forceinline__ void split_pair(float lo, float hi, uint32_t &ph, uint32_t &pm, uint32_t &pl) {
    ph = pack_bf16(lo, hi);
    const float r0 = lo - __uint_as_float(ph << 16), r1 = hi - __uint_as_float(ph & 0xffff0000u);
    pm = pack_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(pm << 16), s1 = r1 - __uint_as_float(pm & 0xffff0000u);
    pl = pack_bf16(s0, s1);
}

__device__ __forceinline__ void split_block(const f32x16 &y, su32x4_t (&h)[2], su32x4_t (&m)[2], su32x4_t (&l)[2]) {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        uint32_t a, b, c;
        split_pair(y[2 * d], y[2 * d + 1], a, b, c);
        h[d / 4][d % 4] = a; m[d / 4][d % 4] = b; l[d / 4][d % 4] = c;
    }
}

__device__ __forceinline__ void split_all(const f32x16 (&Y)[8], Act3 &X) {
#pragma unroll
    for (int o = 0; o < 8; ++o) split_block(Y[o], X.v[0][o], X.v[1][o], X.v[2][o]);
}

#ifndef DGDM_SPLIT_RING
#define DGDM_SPLIT_RING 12
#endif
constexpr int SRD = DGDM_SPLIT_RING;      // ring depth in entries: two (pair, K-step) groups ahead

struct SplitRing {
    float4 e[SRD];
};

__device__ __forceinline__ void sring_fill(wrsrc_t rs, int voff, int base, SplitRing &r) {
#pragma unroll
    for (int i = 0; i < SRD; ++i) r.e[i] = wload(rs, voff, base + i * 1024);
}

// the six terms of one K-step for two accumulators (A: entries 0..2 = h m l, B: entries 3..5); small terms first
#define SPLIT_STEP(accA, accB, w, xh, xm, xl)      \
    do {                                           \
        accA = smfma(w[2], xh, accA);              \
        accB = smfma(w[5], xh, accB);              \
        accA = smfma(w[0], xl, accA);              \
        accB = smfma(w[3], xl, accB);              \
        accA = smfma(w[1], xm, accA);              \
        accB = smfma(w[4], xm, accB);              \
        accA = smfma(w[1], xh, accA);              \
        accB = smfma(w[4], xh, accB);              \
        accA = smfma(w[0], xm, accA);              \
        accB = smfma(w[3], xm, accB);              \
        accA = smfma(w[0], xh, accA);              \
        accB = smfma(w[3], xh, accB);              \
    } while (0)

// One 256 -> 256 layer, input-streaming form.  The layer's INPUT arrives as the previous layer's float32 accumulators Yp (pre-activation
// in the forward pass, unmasked gradient in the backward pass); its epilogue - ReLU + sign bits (forward) or the ReLU mask (backward),
// then the exact three-way bf16 split - is done block by block JUST IN TIME: the K loop runs over the input blocks, all eight output
// blocks accumulate at once (Y, 128 registers), and while block b's two K-steps (96 MFMAs) run, the eight register pairs of block b+1
// are converted, one per group of twelve MFMAs, so the VALU work hides in the shadow of the matrix pipe.  Only block 0's conversion
// is exposed.  Stream order: (K-step, block pair) x [A.h A.m A.l B.h B.m B.l]; 6 entries per twelve MFMAs, ring carried.
//   FWD : sign bits of Yp's blocks go to smask[slot_in + pair] (the dword layout of relu_mask);  !FWD: they are read from there.
template <bool FWD, bool BIAS>
__device__ __forceinline__ void stream_layer(const wrsrc_t rs, const int voff, const int woff, SplitRing &ring, const float *__restrict__ bias,
                                             const f32x16 (&Yp)[8], f32x16 (&Y)[8], uint32_t (*smask)[256], const int slot_in, const int tid,
                                             const int h4) {
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        if (BIAS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = feat4(bias, o, q, h4);
                Y[o][4 * q + 0] = a.x; Y[o][4 * q + 1] = a.y; Y[o][4 * q + 2] = a.z; Y[o][4 * q + 3] = a.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[o][r] = 0.f;
        }
    }
    su32x4_t P[2][3][2];                           // [block parity][piece][K-step]: the B operands of the current and the next input block
    uint32_t mk = FWD ? 0u : smask[slot_in][tid];
    // one register pair (2d, 2d+1) of input block blk: epilogue + split into P[blk & 1]
    auto item = [&](const f32x16 &y, const int blk, const int d) {
        float lo = y[2 * d], hi = y[2 * d + 1];
        const int sh = 2 * d + 16 * (blk & 1);
        if (FWD) {
            mk |= (lo > 0.f ? 1u : 0u) << sh;
            mk |= (hi > 0.f ? 1u : 0u) << (sh + 1);
            // one v_max each (fmaxf would first canonicalise its operand: a second v_max per element)
            asm("v_max_f32 %0, 0, %1" : "=v"(lo) : "v"(lo));
            asm("v_max_f32 %0, 0, %1" : "=v"(hi) : "v"(hi));
        } else {
            lo = ((mk >> sh) & 1u) ? lo : 0.f;
            hi = ((mk >> (sh + 1)) & 1u) ? hi : 0.f;
        }
        uint32_t a, b, c;
        split_pair(lo, hi, a, b, c);
        P[blk & 1][0][d / 4][d % 4] = a; P[blk & 1][1][d / 4][d % 4] = b; P[blk & 1][2][d / 4][d % 4] = c;
    };
#pragma unroll
    for (int d = 0; d < 8; ++d) item(Yp[0], 0, d);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int E = (((b * 2 + sx) * 4) + pp) * 6;
                float4 w[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    w[j] = ring.e[(E + j) % SRD];
#ifdef DGDM_SPLIT_SAMEW      // experiment: every load hits the same 12 KiB (wrong results; shows what the weight stream's latency costs)
                    ring.e[(E + j) % SRD] = wload(rs, voff, ((E + j) % SRD) * 1024);
#else
                    ring.e[(E + j) % SRD] = wload(rs, voff, woff + (E + j + SRD) * 1024);
#endif
                }
                if (b < 7) {                       // the next input block's pair q, in the shadow of this group's MFMAs
                    const int q = sx * 4 + pp, nb = b + 1;
                    if (q == 0 && (nb & 1) == 0) mk = FWD ? 0u : smask[slot_in + nb / 2][tid];
                    item(Yp[nb], nb, q);
                    if (FWD && q == 7 && (nb & 1) == 1) smask[slot_in + nb / 2][tid] = mk;
                }
                SPLIT_STEP(Y[2 * pp], Y[2 * pp + 1], w, P[b & 1][0][sx], P[b & 1][1][sx], P[b & 1][2][sx]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// one 32-feature output block from a 256-feature input on two alternating accumulators (16 K-steps x 3 entries [h m l]): z = za + zb.
// XL_LDS: the l pieces of the input are read from xl_lds ([K-step][lane], this wave's copy) instead of X.v[2] -
// the 3-D forward front parks them in LDS, which takes 64 registers out of its live set; the term that needs them is the LAST of its
// K-step, so the read has five MFMAs to land.  side(ks): VALU work of the caller's (the previous block's epilogue), one slice per K-step,
// placed in the shadow of that step's MFMAs.
template <bool XL_LDS, class Side>
__device__ __forceinline__ f32x16 split_block_out(const wrsrc_t rs, const int voff, const int woff, SplitRing &ring, const Act3 &X, f32x16 za,
                                                  f32x16 zb, const lds_su32x4_t *xl_lds, Side &&side) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int E = ks * 3;
        float4 w[3];
        su32x4_t xl;
        if (XL_LDS) xl = xl_lds[ks * 64];
        else xl = X.v[2][ks / 2][ks % 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            w[j] = ring.e[(E + j) % SRD];
            ring.e[(E + j) % SRD] = wload(rs, voff, woff + (E + j + SRD) * 1024);
        }
        const su32x4_t xh = X.v[0][ks / 2][ks % 2], xm = X.v[1][ks / 2][ks % 2];
        side(ks);
        za = smfma(w[2], xh, za);
        zb = smfma(w[1], xm, zb);
        za = smfma(w[1], xh, za);
        zb = smfma(w[0], xm, zb);
        za = smfma(w[0], xh, za);
        zb = smfma(w[0], xl, zb);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) za[r] += zb[r];
    return za;
}

#ifdef DGDM_SPLIT_STAMPS
// experiment hook: cycle stamps of one wave at the phase boundaries (printed by trunk_split_launch)
__device__ long long g_split_stamps[32];
#define STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && tid == 0) g_split_stamps[i] = clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <int KIND>
__global__ __launch_bounds__(256, 1) void trunk_split_kernel(const TrunkParams p) {
    constexpr int W1B = (KIND == 3) ? 16 : 8;
    constexpr int W1 = W1B * 32;
    constexpr int NSLOT = (KIND == 3) ? 8 + 7 * 4 : 8 * 4;
    __shared__ uint32_t smask[NSLOT][256];
    __shared__ su32x4_t xl_park[KIND == 3 ? 4 : 1][KIND == 3 ? 16 : 1][64];      // 3-D front: the l pieces of the xobj row, per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= p.ntiles) return;                             // wave-uniform; the kernel has no barrier
    const int n = lane & 31;
    const int h4 = (lane >> 5) * 4;
    const int voff = lane * 16;

    // ---- which rows (trunk.hip, table mode): a tile is one finger b of one chain against 32 consecutive pose cells
    const int per_chain = p.B * p.tiles_per_b;
    const int chain = tile / per_chain;
    const int rem = tile - chain * per_chain;
    const int b = rem / p.tiles_per_b;
    const int c = (rem - b * p.tiles_per_b) * 32 + n;
    const bool valid = c < p.C;
    const int64_t r = (int64_t)(valid ? c : p.C - 1) * p.B + b;
    const float *arow = p.Atab + (size_t)(chain * p.B + b) * W1;
    const float4 *ptile = reinterpret_cast<const float4 *>(p.PtabT) + (size_t)(rem - b * p.tiles_per_b) * W1B * 4 * 64 + lane;

    Act3 X;
    f32x16 Y[8];
    uint32_t m[4];
    int slot = 0;
    STAMP(0);
    const wrsrc_t rsF = weight_rsrc(p.Wfwd, p.fwd_bytes);
    SplitRing ring;
    int woff = 0;
    sring_fill(rsF, voff, 0, ring);

    if (KIND == 2) {
        // ---- layer 1: table lookups
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = feat4(arow, o, q, h4);
                const float4 w = ptile[(o * 4 + q) * 64];
                Y[o][4 * q + 0] = v.x + w.x; Y[o][4 * q + 1] = v.y + w.y; Y[o][4 * q + 2] = v.z + w.z; Y[o][4 * q + 3] = v.w + w.w;
            }
        }
    } else {
        // ---- 3-D layers 1 and 2, streamed over the 16 blocks of the 512-wide layer 1 (the xobj row is the B operand of layer 1)
        const float *xrow = p.xtab ? p.xtab[chain] + (size_t)p.xidx[(size_t)chain * p.xstride + r] * 256 : p.xobj + ((size_t)chain * p.xstride + r) * 256;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = feat4(xrow, o, q, h4);
                Y[o][4 * q + 0] = v.x; Y[o][4 * q + 1] = v.y; Y[o][4 * q + 2] = v.z; Y[o][4 * q + 3] = v.w;
            }
        }
        split_all(Y, X);
        const lds_su32x4_t *xl = (const lds_su32x4_t *)&xl_park[wave][0][lane];
#pragma unroll
        for (int o = 0; o < 8; ++o) { xl_park[wave][2 * o][lane] = X.v[2][o][0]; xl_park[wave][2 * o + 1][lane] = X.v[2][o][1]; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = feat4(p.b2, o, q, h4);
                Y[o][4 * q + 0] = b4.x; Y[o][4 * q + 1] = b4.y; Y[o][4 * q + 2] = b4.z; Y[o][4 * q + 3] = b4.w;
            }
        }
        // Layers 1-2 block by block: 96 layer-1 MFMAs make one 32-feature block (accumulated from ZERO: the two table terms are added to
        // the finished sum, one rounding - accumulating on top of them would round every MFMA's contribution at the tables' magnitude
        // instead of the running sum's: 48 roundings at that size where now there is one), epilogue (ReLU, sign bits, three-way split), 96 layer-2
        // MFMAs.  The table terms of block kb + 1 (finger part + pose-cell part) are loaded and added to each other while block kb's
        // layer-2 MFMAs run, so that neither a load nor its latency sits between the two MFMA phases.
        f32x16 zero, tt;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        auto table_terms = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(arow + 32 * kb + 8 * q + h4);
                const float4 w = ptile[(kb * 4 + q) * 64];
                tt[4 * q + 0] = v.x + w.x; tt[4 * q + 1] = v.y + w.y; tt[4 * q + 2] = v.z + w.z; tt[4 * q + 3] = v.w + w.w;
            }
        };
        table_terms(0);
        for (int blk = 0; blk < 16; blk += 2) {
            uint32_t bits2 = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kb = blk + e;
                f32x16 z = split_block_out<true>(rsF, voff, woff, ring, X, zero, zero, xl, [](int) __attribute__((always_inline)) {});
                woff += 48 * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] += tt[r];
                su32x4_t ah[2], am[2], al[2];
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    float lo = z[2 * d], hi = z[2 * d + 1];
                    const int sh = 2 * d + 16 * e;
                    bits2 |= (lo > 0.f ? 1u : 0u) << sh;
                    bits2 |= (hi > 0.f ? 1u : 0u) << (sh + 1);
                    asm("v_max_f32 %0, 0, %1" : "=v"(lo) : "v"(lo));
                    asm("v_max_f32 %0, 0, %1" : "=v"(hi) : "v"(hi));
                    uint32_t a, b, c;
                    split_pair(lo, hi, a, b, c);
                    ah[d / 4][d % 4] = a; am[d / 4][d % 4] = b; al[d / 4][d % 4] = c;
                }
                table_terms((kb + 1) & 15);
                // layer 2: Y[op] += W2'[op][kb] a1[kb], two output blocks at a time, both K-steps of the block
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
                    for (int sx = 0; sx < 2; ++sx) {
                        const int E = (pp * 2 + sx) * 6;
                        float4 w[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            w[j] = ring.e[(E + j) % SRD];
                            ring.e[(E + j) % SRD] = wload(rsF, voff, woff + (E + j + SRD) * 1024);
                        }
                        SPLIT_STEP(Y[2 * pp], Y[2 * pp + 1], w, ah[sx], am[sx], al[sx]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                woff += 48 * 1024;
            }
            smask[blk / 2][tid] = bits2;
        }
        slot = 8;
    }
    // Y = pre-activations of the layer in front of the 256 -> 256 stack (2-D: layer 1; 3-D: layer 2); its ReLU, sign bits and split
    // happen inside the first stack layer, and so on down the stack (stream_layer); the last layer's output gets the plain epilogue
    constexpr int BASE = (KIND == 3) ? 8 : 0;          // mask slots: BASE.. = the layer in front, BASE + 4 + 4 l.. = stack layer l
    f32x16 Z[8];
    STAMP(1);
    for (int l = 0; l < p.n_mid; ++l) {
        stream_layer<true, true>(rsF, voff, woff, ring, p.bf[l], Y, Z, smask, BASE + 4 * l, tid, h4);
        woff += 384 * 1024;
#pragma unroll
        for (int o = 0; o < 8; ++o) Y[o] = Z[o];
        STAMP(2 + l);
    }
    slot = BASE + 4 * p.n_mid;
    relu_mask<8>(Y, m);
#pragma unroll
    for (int i = 0; i < 4; ++i) smask[slot + i][tid] = m[i];
    slot += 4;

    // ---- output layer (256 -> 3) on the VALU, objective gradient (as trunk_kernel)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w0 = feat4(p.Wout, o, q, h4);
            const float4 w1 = feat4(p.Wout + 256, o, q, h4);
            const float4 w2 = feat4(p.Wout + 512, o, q, h4);
            const float x0 = Y[o][4 * q + 0], x1 = Y[o][4 * q + 1], x2 = Y[o][4 * q + 2], x3 = Y[o][4 * q + 3];
            s0 = fmaf(w0.w, x3, fmaf(w0.z, x2, fmaf(w0.y, x1, fmaf(w0.x, x0, s0))));
            s1 = fmaf(w1.w, x3, fmaf(w1.z, x2, fmaf(w1.y, x1, fmaf(w1.x, x0, s1))));
            s2 = fmaf(w2.w, x3, fmaf(w2.z, x2, fmaf(w2.y, x1, fmaf(w2.x, x0, s2))));
        }
    }
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    const float d0 = s0 + p.bout[0], d1 = s1 + p.bout[1], d2 = s2 + p.bout[2];

    const wrsrc_t rsB = weight_rsrc(p.Wbwd, p.bwd_bytes);
    woff = 0;
    sring_fill(rsB, voff, 0, ring);                           // in flight while the objective runs on the VALU
    const TrunkObjective ob = p.obj[chain];
    float g0 = ob.lin[0] + 2.f * ob.quad[0] * d0;
    float g1 = ob.lin[1] + 2.f * ob.quad[1] * d1;
    float g2 = ob.lin[2] + 2.f * ob.quad[2] * d2;
    if (ob.use_rowcoef) g0 = p.rowcoef[(size_t)chain * p.R + r];
    if (!valid) { g0 = 0.f; g1 = 0.f; g2 = 0.f; }

#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w0 = feat4(p.Wout, o, q, h4);
            const float4 w1 = feat4(p.Wout + 256, o, q, h4);
            const float4 w2 = feat4(p.Wout + 512, o, q, h4);
            Y[o][4 * q + 0] = fmaf(g2, w2.x, fmaf(g1, w1.x, g0 * w0.x));
            Y[o][4 * q + 1] = fmaf(g2, w2.y, fmaf(g1, w1.y, g0 * w0.y));
            Y[o][4 * q + 2] = fmaf(g2, w2.z, fmaf(g1, w1.z, g0 * w0.z));
            Y[o][4 * q + 3] = fmaf(g2, w2.w, fmaf(g1, w1.w, g0 * w0.w));
        }
    }

    // ---- backward through the 256 -> 256 layers: layer l masks its incoming gradient with the sign bits of stack layer l's output
    STAMP(10);
    for (int l = p.n_mid - 1; l >= 0; --l) {
        stream_layer<false, false>(rsB, voff, woff, ring, nullptr, Y, Z, smask, BASE + 4 + 4 * l, tid, h4);
        woff += 384 * 1024;
#pragma unroll
        for (int o = 0; o < 8; ++o) Y[o] = Z[o];
        STAMP(11 + (p.n_mid - 1 - l));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = smask[BASE + i][tid];
    apply_mask<8>(Y, m);

    float *dst = p.partial + (size_t)tile * W1;
    if (KIND == 2) {
        // Y = d/dz1 of every row of the tile; fold the 32 cells
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = rows_sum(Y[o][4 * q + 0]); v.y = rows_sum(Y[o][4 * q + 1]);
                v.z = rows_sum(Y[o][4 * q + 2]); v.w = rows_sum(Y[o][4 * q + 3]);
                if (n == ROWS_SUM_LANE) *reinterpret_cast<float4 *>(dst + 32 * o + 8 * q + h4) = v;
            }
        }
    } else {
        // 3-D: one more layer back (256 -> 512), block by block, straight into the fold - block kb - 1's mask and fold (DPP adds) run in
        // the shadow of block kb's MFMAs, one feature per K-step
        split_all(Y, X);
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        f32x16 g = split_block_out<false>(rsB, voff, woff, ring, X, zero, zero, nullptr, [](int) __attribute__((always_inline)) {});
        woff += 48 * 1024;
        float4 acc;
        auto fold_one = [&](const int r, const int kb, const uint32_t bits) __attribute__((always_inline)) {
            const float v = rows_sum(apply_bit(g[r], bits, r));
            if (r % 4 == 0) acc.x = v;
            else if (r % 4 == 1) acc.y = v;
            else if (r % 4 == 2) acc.z = v;
            else {
                acc.w = v;
                if (n == ROWS_SUM_LANE) *reinterpret_cast<float4 *>(dst + 32 * kb + 8 * (r / 4) + h4) = acc;
            }
        };
        for (int kb = 1; kb < 16; ++kb) {
            const uint32_t bits = smask[(kb - 1) / 2][tid] >> (16 * ((kb - 1) & 1));
            const f32x16 gn = split_block_out<false>(rsB, voff, woff, ring, X, zero, zero, nullptr, [&](const int ks) __attribute__((always_inline)) { fold_one(ks, kb - 1, bits); });
            woff += 48 * 1024;
            g = gn;
        }
        const uint32_t bits = smask[7][tid] >> 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) fold_one(r, 15, bits);
    }
    STAMP(20);
}

int trunk_split_launch(int kind, const TrunkParams &p, hipStream_t s) {
    if (p.n_mid != (kind == 3 ? 6 : 7)) return DGDM_EINVAL;
    const int grid = (p.ntiles + 3) / 4;
    if (grid == 0) return DGDM_OK;
    // algorithmic FLOPs (float32 contraction FLOPs of the MFMA layers on the real rows, DESIGN_HISTORY.md 5) - the matrix pipe issues six bf16
    // products per float32 product, i.e. 6x this number of bf16 FLOPs
    const double rows = (double)(p.ntiles / std::max(1, p.tiles_per_b)) * p.C;
    const double mid = 2.0 * 256 * 256 * p.n_mid;
    const double per_row = (kind == 3) ? (2.0 * 256 * 512 * 2 + mid) + (2.0 * 256 * 512 + mid) : 2.0 * mid;
    prof_begin(s, DGDM_STAGE_TRUNK);
    if (kind == 2) hipLaunchKernelGGL((trunk_split_kernel<2>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((trunk_split_kernel<3>), dim3(grid), dim3(256), 0, s, p);
    DGDM_HIP_CHECK(hipGetLastError());
    prof_end(s, DGDM_STAGE_TRUNK, rows * per_row);
#ifdef DGDM_SPLIT_STAMPS
    {
        long long st[32];
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_split_stamps), sizeof(st));
        fprintf(stderr, "split stamps kind %d:", kind);
        long long prev = st[0];
        for (int i = 1; i <= 20; ++i) if (st[i]) { fprintf(stderr, " [%d]%lld", i, st[i] - prev); prev = st[i]; }
        fprintf(stderr, " total %lld\n", st[20] - st[0]);
    }
#endif
    return DGDM_OK;
}

}  // namespace dgdm
