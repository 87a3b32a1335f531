// Register-resident MLP chains on v_mfma_f32_32x32x2_f32 (gfx950).
//
// One wave owns 32 batch rows.  An activation matrix H [F features x 32 rows] lives in
// F/32 blocks of 16 VGPRs in the MFMA C/D layout:
//
//   block o, register r, lane l (n = l & 31, h = l >> 5)   <->   H[32 o + rho(r,h)][row n],
//   rho(r,h) = (r & 3) + 8 (r >> 2) + 4 h
//
// A layer is computed transposed, D[out x rows] += W[out x in] * H[in x rows], with the weights as
// the MFMA A operand and the activations as the B operand.  B of 32x32x2 wants lane (n,h) to supply
// H[k_h][n] for the two k of the step - and register (o,r) of that lane already holds
// H[32 o + rho(r,0)][n] in the lower half-wave and H[32 o + rho(r,1)][n] in the upper one.  So the
// K-step (o,r) contracts over exactly that feature pair: the previous layer's accumulators ARE the
// next layer's B operands.  Activations never leave the VGPRs, there is no LDS traffic and no
// barrier between layers; only the weights stream (L2-resident, one coalesced 1 KiB load per 4 MFMAs).
//
// Weight image (host side: dgdm::pack_chain): for A, lane (i = l & 31, h) needs W[32 o' + i][32 o + rho(r,h)].
// Registers r = 4q..4q+3 are four consecutive input features, so one float4 per lane serves 4 MFMAs:
//   img[((o' * KB + o) * 4 + q) * 64 + l] = W[32 o' + i][32 o + 8 q + 4 h + {0,1,2,3}]
//
// The same image format with W^T gives the input-gradient pass G_in = W^T G_out.
// Numerics: every output is a k-ordered float32 fma chain (exact f32, no reduced precision).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dgdm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// bf16 helpers shared by the bf16-contraction kernels.  pack_bf16 is v_cvt_pk_bf16_f32 (round to nearest even).  Rows of
// NON-NEGATIVE bf16 values (post-ReLU features) compare like unsigned 16-bit integers, so their elementwise max is v_pk_max_u16;
// and rounding is monotonic, so max(bf(a), bf(b)) == bf(max(a, b)): a feature table whose only consumer rounds to bf16 can be
// stored in bf16 without changing a bit of the result.
typedef __bf16 dgdm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float dgdm_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short dgdm_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    const dgdm_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dgdm_bf16x2));
}
__device__ __forceinline__ uint32_t pkmax_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(dgdm_u16x2, a), __builtin_bit_cast(dgdm_u16x2, b)));
}
// A 256-feature row in bf16 "operand order" (512 bytes): dword 16 o + 8 h + d = features 32 o + rho(2d, h), +1 - i.e. exactly the
// eight dwords lane (n, h) of the bf16 trunk needs as the B operand of block o, contiguous (trunk_bf16.hip).

// float4 holding features 32 o + 8 q + 4 h + {0..3} of a row-major feature vector
__device__ __forceinline__ float4 feat4(const float *__restrict__ row, int o, int q, int h4) {
    return *reinterpret_cast<const float4 *>(row + 32 * o + 8 * q + h4);
}

// Depth of the software prefetch ring on the weight stream: one float4 per lane feeds 4 MFMAs
// (256 cycles), so DEPTH entries cover DEPTH*256 cycles of L2 latency with one wave per SIMD.
#ifndef DGDM_CHAIN_DEPTH
#define DGDM_CHAIN_DEPTH 12      // A/B on MI355X: 8 -> 12 is +6 % on the 3-D trunk, 0 on the 2-D one; 16 = 12
#endif

// Weight images are read through a buffer descriptor (wave-uniform base in SGPRs, per-lane byte
// offset lane*16 in ONE VGPR, entry offset as scalar/immediate): no 64-bit per-load address math in
// VGPRs, which is what lets the compiler keep a deep prefetch ring beside 128 operand registers.
// (hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b128 lowers to a ONE-dword load splatted over the vector -
// checked in the .s - so the LLVM intrinsic is bound directly, the way composable_kernel does.)
typedef int wrsrc_t __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
__device__ v4f32 llvm_amdgcn_raw_buffer_load_v4f32(wrsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

__device__ __forceinline__ wrsrc_t weight_rsrc(const float4 *base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    wrsrc_t rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));     // stride 0, no swizzle
    rs.z = (int)bytes;
    rs.w = 0x00020000;
    return rs;
}

__device__ __forceinline__ float4 wload(wrsrc_t rs, int voff, int soff) {
    const v4f32 v = llvm_amdgcn_raw_buffer_load_v4f32(rs, voff, soff, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}

// Streams `TOTAL` consecutive float4-per-lane weight entries (1 KiB apart) starting at byte offset
// `base_off` of the image and feeds them to `body(i, a)` in order, keeping DGDM_CHAIN_DEPTH loads in flight.
template <int TOTAL, class Body>
__device__ __forceinline__ void stream_weights(wrsrc_t rs, int voff, int base_off, Body &&body) {
    constexpr int D = DGDM_CHAIN_DEPTH < TOTAL ? DGDM_CHAIN_DEPTH : TOTAL;
    float4 ring[D];
#pragma unroll
    for (int i = 0; i < D; ++i) ring[i] = wload(rs, voff, base_off + i * 1024);
#pragma unroll
    for (int i = 0; i < TOTAL; ++i) {
        const float4 a = ring[i % D];
        if (i + D < TOTAL) ring[i % D] = wload(rs, voff, base_off + (i + D) * 1024);
        body(i, a);
        // pin the step: [wait entry i | 4 MFMA | issue entry i+D].  Left alone, hipcc's scheduler
        // sinks the loads next to their uses over long stretches (vmcnt(1) instead of vmcnt(D-1)).
        __builtin_amdgcn_sched_barrier(0);
    }
}

enum { CHAIN_ZERO = 0, CHAIN_BIAS = 1, CHAIN_KEEP = 2 };

// out[MB] (+)= W * in[KB].  Wp: chain image of W [32 MB x 32 KB]; INIT selects the accumulator start:
// zero, the bias vector [32 MB], or whatever the caller already put into out[].
template <int KB, int MB, int INIT>
__device__ __forceinline__ void chain_layer(const float4 *__restrict__ Wp, const float *__restrict__ bias,
                                            const f32x16 (&in)[KB], f32x16 (&out)[MB], int lane) {
    const int h4 = (lane >> 5) * 4;
    if (INIT != CHAIN_KEEP) {
#pragma unroll
        for (int op = 0; op < MB; ++op) {
            if (INIT == CHAIN_BIAS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = feat4(bias, op, q, h4);
                    out[op][4 * q + 0] = b4.x; out[op][4 * q + 1] = b4.y; out[op][4 * q + 2] = b4.z; out[op][4 * q + 3] = b4.w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[op][r] = 0.f;
            }
        }
    }
    stream_weights<MB * KB * 4>(weight_rsrc(Wp, MB * KB * 4 * 1024), lane * 16, 0, [&](int i, const float4 a) {
        const int op = i / (KB * 4), o = (i / 4) % KB, q = i % 4;
        out[op] = mfma32(a.x, in[o][4 * q + 0], out[op]);
        out[op] = mfma32(a.y, in[o][4 * q + 1], out[op]);
        out[op] = mfma32(a.z, in[o][4 * q + 2], out[op]);
        out[op] = mfma32(a.w, in[o][4 * q + 3], out[op]);
    });
}

// ------------------------------------------------------------------------------------------------
// Continuous weight stream: the images of consecutive layers (or block passes) are laid out back to back in ONE buffer in
// the order the kernel consumes them, and a 16-entry ring (16 divides every pass length used: 32, 64, 256) is carried
// in registers from pass to pass.  A pass always prefetches 16 entries past its own end - the head of the next pass -
// so there is no ring drain/refill bubble between layers; reads past the end of the buffer are clipped to zero by the
// buffer descriptor and never used.
constexpr int CONT_DEPTH = 16;

__device__ __forceinline__ void ring_fill(wrsrc_t rs, int voff, int base_off, float4 (&ring)[CONT_DEPTH]) {
#pragma unroll
    for (int i = 0; i < CONT_DEPTH; ++i) ring[i] = wload(rs, voff, base_off + i * 1024);
}

template <int TOTAL, class Body>
__device__ __forceinline__ void stream_cont(wrsrc_t rs, int voff, int base_off, float4 (&ring)[CONT_DEPTH], Body &&body) {
    static_assert(TOTAL % CONT_DEPTH == 0, "pass length must be a multiple of the ring depth");
#pragma unroll
    for (int i = 0; i < TOTAL; ++i) {
        const float4 a = ring[i % CONT_DEPTH];
        ring[i % CONT_DEPTH] = wload(rs, voff, base_off + (i + CONT_DEPTH) * 1024);
        body(i, a);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// out[MB] (+)= W * in[KB] with the weights taken from the continuous stream at byte offset base_off (256 -> 256: 256 entries).
template <int KB, int MB, int INIT>
__device__ __forceinline__ void chain_layer_cont(wrsrc_t rs, int base_off, float4 (&ring)[CONT_DEPTH], const float *__restrict__ bias,
                                                 const f32x16 (&in)[KB], f32x16 (&out)[MB], int lane) {
    const int h4 = (lane >> 5) * 4;
    if (INIT != CHAIN_KEEP) {
#pragma unroll
        for (int op = 0; op < MB; ++op) {
            if (INIT == CHAIN_BIAS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = feat4(bias, op, q, h4);
                    out[op][4 * q + 0] = b4.x; out[op][4 * q + 1] = b4.y; out[op][4 * q + 2] = b4.z; out[op][4 * q + 3] = b4.w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[op][r] = 0.f;
            }
        }
    }
    stream_cont<MB * KB * 4>(rs, lane * 16, base_off, ring, [&](int i, const float4 a) {
        const int op = i / (KB * 4), o = (i / 4) % KB, q = i % 4;
        out[op] = mfma32(a.x, in[o][4 * q + 0], out[op]);
        out[op] = mfma32(a.y, in[o][4 * q + 1], out[op]);
        out[op] = mfma32(a.z, in[o][4 * q + 2], out[op]);
        out[op] = mfma32(a.w, in[o][4 * q + 3], out[op]);
    });
}

// one 32-feature output block: acc = W_block * in[KB]   (KB*4 entries)
template <int KB>
__device__ __forceinline__ f32x16 chain_block_cont(wrsrc_t rs, int base_off, float4 (&ring)[CONT_DEPTH], const f32x16 (&in)[KB], int lane) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    stream_cont<KB * 4>(rs, lane * 16, base_off, ring, [&](int i, const float4 a) {
        const int o = i / 4, q = i % 4;
        acc = mfma32(a.x, in[o][4 * q + 0], acc);
        acc = mfma32(a.y, in[o][4 * q + 1], acc);
        acc = mfma32(a.z, in[o][4 * q + 2], acc);
        acc = mfma32(a.w, in[o][4 * q + 3], acc);
    });
    return acc;
}

// acc[MB] += W[:, one 32-feature input block] * x   (MB*4 entries, ordered (op, q))
template <int MB>
__device__ __forceinline__ void chain_accumulate_cont(wrsrc_t rs, int base_off, float4 (&ring)[CONT_DEPTH], const f32x16 &x,
                                                      f32x16 (&acc)[MB], int lane) {
    stream_cont<MB * 4>(rs, lane * 16, base_off, ring, [&](int i, const float4 a) {
        const int op = i / 4, q = i % 4;
        acc[op] = mfma32(a.x, x[4 * q + 0], acc[op]);
        acc[op] = mfma32(a.y, x[4 * q + 1], acc[op]);
        acc[op] = mfma32(a.z, x[4 * q + 2], acc[op]);
        acc[op] = mfma32(a.w, x[4 * q + 3], acc[op]);
    });
}

// ReLU in place; returns the sign bits (bit r set <=> x[r] > 0), torch's relu'(0) = 0.
__device__ __forceinline__ uint32_t relu_bits(f32x16 &x) {
    uint32_t bits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool pos = x[r] > 0.f;
        bits |= pos ? (1u << r) : 0u;
        x[r] = pos ? x[r] : 0.f;
    }
    return bits;
}

// g where bit r of bits is set, else +0 (two VALU: a sign-extended one-bit field as the AND mask)
__device__ __forceinline__ float apply_bit(float g, uint32_t bits, int r) {
    return __builtin_bit_cast(float, __builtin_bit_cast(int, g) & __builtin_amdgcn_sbfe((int)bits, r, 1));
}

__device__ __forceinline__ void apply_bits(f32x16 &g, uint32_t bits) {
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = apply_bit(g[r], bits, r);
}

template <int NB>
__device__ __forceinline__ void relu_mask(f32x16 (&x)[NB], uint32_t (&m)[NB / 2]) {
#pragma unroll
    for (int o = 0; o < NB; o += 2) m[o / 2] = relu_bits(x[o]) | (relu_bits(x[o + 1]) << 16);
}

template <int NB>
__device__ __forceinline__ void apply_mask(f32x16 (&g)[NB], const uint32_t (&m)[NB / 2]) {
#pragma unroll
    for (int o = 0; o < NB; o += 2) {
        apply_bits(g[o], m[o / 2] & 0xffffu);
        apply_bits(g[o + 1], m[o / 2] >> 16);
    }
}

// Sum over the 32 rows of the tile (lanes 0..31 and, separately, 32..63), on the VALU's DPP paths - no LDS round trips.  The adds pair up
// exactly as in an xor butterfly (1, 2, 4, 8, 16).  The total is valid on the UPPER 16 lanes of each half-wave only (n >= 16):
// store from lane n == ROWS_SUM_LANE.
constexpr int ROWS_SUM_LANE = 16;
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float rows_sum(float v) {
    v += dpp_f32<0xb1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_f32<0x4e, 0xf>(v);        // quad_perm [2,3,0,1]
    v += dpp_f32<0x141, 0xf>(v);       // row_half_mirror: the other quad of the 8
    v += dpp_f32<0x140, 0xf>(v);       // row_mirror: the other 8 of the 16
    v += dpp_f32<0x142, 0xa>(v);       // row_bcast15 into rows 1 and 3: the other 16 of the 32
    return v;
}

}  // namespace dgdm
