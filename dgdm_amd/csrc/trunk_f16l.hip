// Fused dynamics trunk, forward + input-gradient backward (same contract as trunk_kernel, trunk.hip), float32-grade on the f16
// matrix pipe: the float32 trunk the library runs by default.
//
// Arithmetic.  Every float32 product as THREE f16 MFMAs instead of six bf16 ones:
//     x = x_h + x_l,  x_h = f16(x),  x_l = f16(x - x_h)   (x - x_h is exact in float32; 11 + 11 significant bits, 2^-23 of x at worst)
//     w x  ~  w_h x_h + w_h x_l + w_l x_h                  (each exact in float32; the dropped w_l x_l is below 2^-22 of the product)
// on v_mfma_f32_32x32x16_f16 with float32 accumulation.  f16 has five exponent bits where bf16 has float32's eight, so both operands
// are brought into its range by EXACT powers of two first:
//   * weights: one scale per matrix, chosen on the host so that max |w 2^ew| lies in [2^12, 2^13) (Split2, models_api.hip);
//   * activations / gradients: one scale per ROW of the 32-row tile, from the row's largest magnitude at the layer's input (128
//     v_max per lane + one exchange with the lane that holds the row's other half), max |x 2^k| in [2^12, 2^13); entries 2^16 below
//     their row's maximum keep their full 22 bits, smaller ones an absolute error of 2^-38 of the maximum;
//   * the one input that is consumed while it is still being produced - 3-D layer 1's output, block by block into layer 2 - takes its
//     row scale from an upper BOUND known before the first block exists: |z_j| <= max |A[finger]| + max |P[cell]| + ||W1o_j||_1 max |x|
//     (the two table terms' largest magnitudes: one reduction per tile, one array built with the pose table; the largest absolute row
//     sum of the weights: host).  A bound 2^b above the true maximum only lifts the absolute floor from 2^-38 to 2^(b - 38) of it.
// The accumulators then hold the true pre-activations times 2^E, E = (input scale) + ew per row; ReLU and its sign bits do not care, the
// bias is added as b 2^E, and E is taken out where true values are needed (the output layer, the folded tile sums).  Measured against
// float64 (scripts/micro/split_mfma.hip, K = 256, He-init weights, post-ReLU inputs): rms error 1.9e-7 of the rms output, between the
// 1.6e-7 of the six-product bf16 form and the 2.0e-7 of the v_mfma_f32 chain.  Half the matrix-pipe instructions and two thirds of the
// weight bytes of the six-bf16-product form it replaced (rounds 3-5, retired in round 6; DESIGN_HISTORY.md 4.10).
//
// Structure.  One wave = one tile of 32 rows through all layers, forward and backward, in registers:
//   Y [8] f32x16   a layer's output in the MFMA C/D layout (register r, lane (n, h)  <->  feature 32 o + rho(r, h) of row n)
//   P / Act2       the layer's input as two sets of packed f16 B operands: K-step s of block o = registers 8s..8s+7 of Y[o], converted
//                  pairwise (the weight images are images of the two pieces in the same operand order)
// A 256 -> 256 layer is input-streaming (stream_layer): the K loop runs over the input blocks while all eight output blocks
// accumulate, six MFMAs per (K-step, pair of output blocks), consecutive MFMAs never on the same accumulator; the previous layer's
// epilogue - scale, ReLU + sign bits (exact float32 semantics, x > 0; bits kept in LDS for the backward pass), then the split of an
// accumulator pair into its two pieces - is done just in time, one register pair per group of six MFMAs, in the shadow of the matrix
// pipe.  The 512-wide first layers of the 3-D model and the last layer back are produced block by block (block_out).
// Stream entries per (K-step, output-block pair): [A.h A.l B.h B.l], per block-out K-step: [h l] (host: Split2 / f16_layer_stream,
// models_api.hip).
//
// The weight stream is SHARED by the workgroup's four waves through LDS.  Pulled through the CU's L1 once per wave (the ring form of
// round 4's first version, 683 bytes per MFMA) it arrived at ~50 of the 85 B/clk/CU the MFMAs ask for: a 256 -> 256 layer took 22 k
// cycles for 12.3 k of MFMA issue.  The four waves consume the SAME stream, so each wave fetches a QUARTER of every 16 KiB chunk,
// straight into LDS (buffer_load_dwordx4 ... lds: no staging registers), and all four read their MFMA A operands from there
// (ds_read_b128, 1 KiB per instruction, linear: conflict-free): a quarter of the L1 traffic.  Pipeline (chunk = 16 entries = one K-step
// of all four output-block pairs): four LDS slots; while chunk c is consumed, c + 1 and c + 2 are in flight; before the LAST group of
// chunk c every wave waits for its own quarter of c + 1 (s_waitcnt vmcnt(4): c + 2's four loads may stay in flight), one s_barrier
// makes the whole chunk visible - and proves every wave done reading chunk c - 1 ... c (a group's operands are read one group ahead) -
// then c + 3 is issued into the slot of c - 1.  One barrier per 768 cycles of MFMA issue.  A wave without a tile runs the last tile
// again and stores nothing (the barriers need all four).
//
// The LDS-DMA loads are issued from inline assembly (LStream::issue says why: hipcc puts s_waitcnt vmcnt(0) in front of every LDS read
// that follows an LDS-DMA load it knows of).  scripts/micro/mfma_dep.hip: the group of six MFMAs + the splitting + four ds_read_b128
// runs at 223 cycles on its own (14.3 k per layer); in the kernel a group takes 270-295: what is left is the stream itself (LDS written
// by the DMA while it is read, barrier skew between the four waves), not the instruction mix.  Measured numbers: DESIGN_HISTORY.md 4.12.
#include "common.h"
#include <algorithm>
#include "mfma_chain.h"
#include "trunk.h"

namespace dgdm {
#ifdef DGDM_F16_STAMPS
// experiment hook: cycle stamps of one wave at the phase boundaries (printed by trunk_f16l_launch)
__device__ long long g_f16l_stamps[48];
#define HSTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_f16l_stamps[i] = clock64(); } while (0)
#else
#define HSTAMP(i) do { } while (0)
#endif
namespace f16l {

typedef _Float16 hf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x2_t __attribute__((ext_vector_type(2)));
typedef float hf32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t hu32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 hmfma(const v4f32 a, const hu32x4_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8_t, a), __builtin_bit_cast(hf16x8_t, b), c, 0, 0, 0);
}

// 2^e as a float (e clamped to the normal range)
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((uint32_t)(min(max(e, -126), 127) + 127) << 23); }

// k with m 2^k in [2^12, 2^13) for the row whose two half-wave lanes hold m; 0 for a row of zeros.  lo..hi: the range k may take
__device__ __forceinline__ int scale_exp(float m, int lo, int hi) {
    m = fmaxf(m, __shfl_xor(m, 32));
    const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
    const int k = (m > 0.f && e < 255) ? 12 + 127 - e : 0;
    return min(max(k, lo), hi);
}

// an already scaled pair -> packed f16 (h | l), h + l == the pair to 2^-23
__device__ __forceinline__ void split2(float lo, float hi, uint32_t &ph, uint32_t &pl) {
    const hf32x2_t v = {lo, hi};
    const hf16x2_t h = __builtin_convertvector(v, hf16x2_t);            // v_cvt_pk_f16_f32 (round to nearest even)
    const hf32x2_t d = v - __builtin_convertvector(h, hf32x2_t);        // exact
    ph = __builtin_bit_cast(uint32_t, h);
    pl = __builtin_bit_cast(uint32_t, __builtin_convertvector(d, hf16x2_t));
}
// The same values in three instructions instead of five (round 5): the residual and its conversion are ONE mixed-precision fma per value
// (v_fma_mixlo/mixhi_f16: (-h) x 1.0 + v evaluated in float32 - exact - and rounded once to f16: bit for bit what convert-back, subtract,
// convert give).  The work between the MFMAs is NOT free (removing it all: 7.12 -> 6.19 ms per 3-D launch, a round-5 timing build), so every
// instruction there counts.  Inline assembly (hipcc does not form the mix instructions from this pattern), and therefore ONLY where the
// pieces are consumed many instructions later (the stack layers' items: at least four MFMA groups): the hazard recogniser does not see an
// asm statement as a VALU write, so an MFMA reading the pieces right behind it gets no wait states - used in the 3-D front that way the
// results were nondeterministic (scripts/det_bits.sh).
__device__ __forceinline__ void split2_mix(float lo, float hi, uint32_t &ph, uint32_t &pl) {
    uint32_t h, l;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(lo), "v"(hi));
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(lo));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(hi));
    ph = h; pl = l;
}
struct Act2 {
    hu32x4_t v[2][8][2];      // [piece][32-feature block][K-step]
};

// The shared weight stream (see the file header).
// Chunk = CH stream entries (1 KiB each): 16 (one K-step of all four output-block pairs) in four slots.  -DDGDM_F16_CHUNK=32 (round 5
// experiment): a whole 32-feature input block (both K-steps) per chunk in THREE slots (96 KiB) - half the barriers and counted waits per
// MFMA (one per 1 536 cycles of issue); c + 3 goes into c's slot once every wave has passed the barrier behind its last reads of c.
// Bit-identical results, and 1.5 % SLOWER on the same box (3-D launch 6.92 -> 7.03 ms, three alternating runs each): the barrier count
// is not what holds the stack at 0.6 of its MFMA rate.
#ifndef DGDM_F16_CHUNK
#define DGDM_F16_CHUNK 16
#endif
#ifndef DGDM_F16_SLOTS
#define DGDM_F16_SLOTS 4
#endif
// LDS slots of the stream and chunks requested ahead of the one being consumed (AHEAD = NBUF - 2 with 16 KiB chunks: the slot a request
// lands in is the one the PREVIOUS chunk left, proven free by the barrier; 32 KiB chunks keep their three slots and two chunks ahead)
constexpr int CH = DGDM_F16_CHUNK, NBUF = CH == 16 ? DGDM_F16_SLOTS : 3, AHEAD = CH == 16 ? NBUF - 2 : 2;
constexpr int CQ = CH / 4 /* entries a wave fetches per chunk */, CG = CH / 4 /* groups of four entries per chunk */;
static_assert(NBUF >= 3 && NBUF <= 6 && CQ * AHEAD <= 48, "stream slots: 3 .. 6 (LDS), at most 48 DMA loads in flight per wave (vmcnt)");
typedef __attribute__((address_space(3))) v4f32 lds_f4_t;      // (a plain vector type: HIP's float4 class has no address-space-qualified copy)
__device__ void llvm_amdgcn_raw_buffer_load_lds(wrsrc_t rsrc, __attribute__((address_space(3))) void *lds, int size, int voffset, int soffset, int offset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.lds");

struct LStream {
    wrsrc_t rs;
    lds_f4_t *buf;           // [NBUF][CH][64 lanes]
    int voff, wave, lane;
    int base;                // byte offset of the stream's chunk 0
    int cur, slot;           // chunk whose entries read() addresses (relative to base), and its slot
#ifdef DGDM_F16_STAMPS
    long long stall_vm = 0, stall_bar = 0;
#endif
    // This wave's quarter of chunk c, straight into LDS.  Inline assembly on purpose: behind an LDS-DMA load hipcc can see, it orders EVERY
    // LDS read after ALL such loads in flight (s_waitcnt vmcnt(0) in front of each ds_read_b128 - it cannot tell the four slots apart),
    // which serialises the stream with its own prefetch; the first version of this kernel was 20 % slower than the per-wave ring for that.
    // Unseen by the compiler the loads only make its own vmcnt waits longer than necessary (memory returns in order), never too short;
    // what makes a slot safe to read is advance()'s counted wait + barrier.
    __device__ __forceinline__ void issue(int c) {
        const int sl = c % NBUF;
#pragma unroll
        for (int j = 0; j < CQ; ++j) {
            const int e = CQ * wave + j;
            const uint32_t la = (uint32_t)(uintptr_t)(buf + (sl * CH + e) * 64);
            const int so = base + (c * CH + e) * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(la), "v"(voff), "s"(rs), "s"(so) : "memory", "m0");
        }
    }
    // a new stream: chunks 0 .. AHEAD on their way, chunk 0 readable on return
    __device__ __forceinline__ void start(const wrsrc_t rs_, int base_) {
        __builtin_amdgcn_s_barrier();                      // nobody reads the previous stream's slots any more
        rs = rs_; base = base_; cur = 0; slot = 0;
#pragma unroll
        for (int c = 0; c <= AHEAD; ++c) issue(c);
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CQ * AHEAD) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    // chunk cur + 1 becomes readable, chunk cur + AHEAD + 1 is requested; cur moves on
    __device__ __forceinline__ void advance() {
        // (lgkmcnt(0): this wave's LDS reads of the chunk it leaves have returned - with three slots the chunk requested below lands in that slot)
#ifdef DGDM_F16_STAMPS
        const long long t0 = clock64();
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(CQ * (AHEAD - 1)) : "memory");
        const long long t1 = clock64();
        __builtin_amdgcn_s_barrier();
        const long long t2 = clock64();
        stall_vm += t1 - t0; stall_bar += t2 - t1;
#else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(CQ * (AHEAD - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
#endif
        issue(cur + AHEAD + 1);
        ++cur;
        slot = slot == NBUF - 1 ? 0 : slot + 1;
    }
    // entry e (0 .. 15) of chunk cur
    __device__ __forceinline__ v4f32 read(const int e) const { return buf[(slot * CH + e) * 64 + lane]; }
};

// the three terms of one K-step for two accumulators (w: [A.h A.l B.h B.l]); small terms first
#define F16_STEP(accA, accB, w, xh, xl)      \
    do {                                     \
        accA = hmfma(w[1], xh, accA);        \
        accB = hmfma(w[3], xh, accB);        \
        accA = hmfma(w[0], xl, accA);        \
        accB = hmfma(w[2], xl, accB);        \
        accA = hmfma(w[0], xh, accA);        \
        accB = hmfma(w[2], xh, accB);        \
    } while (0)
// One 256 -> 256 layer, input-streaming (see the file header).  Yp: the previous layer's accumulators = true values x 2^E per
// row (E: this lane's row); on return Y = this layer's accumulators and E their scale.  ew: the weight matrix' scale exponent.
// wn: the operands of the NEXT group to be consumed (read from LDS one group ahead); precondition of every consumer below: wn holds the
// first group of the pass about to start, postcondition: the first group of the pass that follows in the stream.
template <bool FWD, bool BIAS>
__device__ __forceinline__ void stream_layer(LStream &ls, v4f32 (&wn)[4], const lds_f4_t *bias /* LDS: the layer's 256 values */, const f32x16 (&Yp)[8], f32x16 (&Y)[8],
                                             uint32_t (*smask)[256], const int slot_in, const int tid, const int h4, int &E, const int ew) {
    float mx = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = FWD ? fmaxf(mx, Yp[o][r]) : fmaxf(mx, fabsf(Yp[o][r]));
    const int k = scale_exp(mx, -100 - E - ew, 100 - E - ew);
    const float f = pow2f(k);
    E += k + ew;
    const float fb = pow2f(E);
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        if (BIAS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f32 a = bias[(32 * o + 8 * q + h4) >> 2];
                Y[o][4 * q + 0] = a.x * fb; Y[o][4 * q + 1] = a.y * fb; Y[o][4 * q + 2] = a.z * fb; Y[o][4 * q + 3] = a.w * fb;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[o][r] = 0.f;
        }
    }
    hu32x4_t P[2][2][2];                           // [block parity][piece][K-step]
    HSTAMP(FWD ? 23 : 26);
    uint32_t mk = FWD ? 0u : smask[slot_in][tid];
    auto item = [&](const f32x16 &y, const int blk, const int d, const bool far) __attribute__((always_inline)) {
        // scale first (one packed multiply; exact, f is a power of two), then ReLU / mask: the same values as the other way round
        const int sh = 2 * d + 16 * (blk & 1);
        float lo, hi;
        if (FWD) {
            // forward: ONE packed fma makes -(y f) (+0 for a zero of either sign: y x (-f) + (+0)); its sign bit is (y > 0) - the mask bit,
            // pushed through a shift register, one v_alignbit_b32 per value (mk = mk << 1 | sign) instead of compare + select + or: the
            // 32 values of two blocks arrive in the order of their bit numbers, so the finished word is bitreverse(mk) (where it is
            // stored) - and max(0, -that) is the scaled ReLU.  (Round 5's steps to this form, each bit-identical: DESIGN.md 4.1;
            // what the kernel costs without this item at all, -13 %, and without its LDS operand reads, -1.7 %: measured there too.)
            const hf32x2_t nv = __builtin_elementwise_fma(hf32x2_t{y[2 * d], y[2 * d + 1]}, hf32x2_t{-f, -f}, hf32x2_t{0.f, 0.f});
            mk = __builtin_amdgcn_alignbit(mk, __float_as_uint(nv.x), 31);
            mk = __builtin_amdgcn_alignbit(mk, __float_as_uint(nv.y), 31);
            asm("v_max_f32 %0, 0, -%1" : "=v"(lo) : "v"(nv.x));
            asm("v_max_f32 %0, 0, -%1" : "=v"(hi) : "v"(nv.y));
        } else {
            const hf32x2_t v = hf32x2_t{y[2 * d], y[2 * d + 1]} * hf32x2_t{f, f};
            lo = apply_bit(v.x, mk, sh);
            hi = apply_bit(v.y, mk, sh + 1);
        }
        uint32_t a, b;
        if (far) split2_mix(lo, hi, a, b);         // consumed at least four MFMA groups later (see split2_mix)
        else split2(lo, hi, a, b);
        P[blk & 1][0][d / 4][d % 4] = a; P[blk & 1][1][d / 4][d % 4] = b;
    };
#pragma unroll
    for (int d = 0; d < 8; ++d) item(Yp[0], 0, d, false);        // block 0's pieces feed the first MFMAs
    HSTAMP(FWD ? 24 : 27);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {           // one chunk: the K-step's four output-block pairs
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                v4f32 w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = wn[j];
                const int gi = (sx * 4 + pp) & (CG - 1);         // this group's place in its chunk
                if (gi == CG - 1) ls.advance();    // the next chunk (the next layer's first, after the last one)
#pragma unroll
                for (int j = 0; j < 4; ++j) wn[j] = ls.read(((gi + 1) & (CG - 1)) * 4 + j);
                if (b < 7) {
                    const int q = sx * 4 + pp, nb = b + 1;
                    if (q == 0 && (nb & 1) == 0) mk = FWD ? 0u : smask[slot_in + nb / 2][tid];
                    item(Yp[nb], nb, q, true);
                    if (FWD && q == 7 && (nb & 1) == 1) smask[slot_in + nb / 2][tid] = __builtin_bitreverse32(mk);
                }
                F16_STEP(Y[2 * pp], Y[2 * pp + 1], w, P[b & 1][0][sx], P[b & 1][1][sx]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    HSTAMP(FWD ? 25 : 28);
}

// one 32-feature output block from a 256-feature input on two alternating accumulators (16 K-steps x [h l]): z = za + zb
// (32 entries = two chunks).  Three MFMAs per K-step are 96 cycles - less than an LDS round trip under load - so the operands are read TWO
// K-steps ahead: on entry wn[0..1] / wn[2..3] hold K-steps 0 / 1 of this pass (= its entries 0 .. 3, the same precondition as a stack
// layer's), on return entries 0 .. 3 of the pass that follows.
template <class Side>
__device__ __forceinline__ f32x16 block_out(LStream &ls, v4f32 (&wn)[4], const Act2 &X, f32x16 za, f32x16 zb, Side &&side) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int s = (ks & 1) * 2;
        const v4f32 w0 = wn[s], w1 = wn[s + 1];
        if ((ks & (CH / 2 - 1)) == CH / 2 - 2) ls.advance();
        wn[s] = ls.read(((ks + 2) & (CH / 2 - 1)) * 2); wn[s + 1] = ls.read(((ks + 2) & (CH / 2 - 1)) * 2 + 1);
        const hu32x4_t xh = X.v[0][ks / 2][ks % 2], xl = X.v[1][ks / 2][ks % 2];
        side(ks);
        za = hmfma(w1, xh, za);
        zb = hmfma(w0, xl, zb);
        za = hmfma(w0, xh, za);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) za[r] += zb[r];
    return za;
}

// true values x 2^k of a whole activation matrix -> the two f16 pieces; returns k (row scale)
__device__ __forceinline__ int split_rows(const f32x16 (&Y)[8], Act2 &X) {
    float mx = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(Y[o][r]));
    const int k = scale_exp(mx, -100, 100);
    const float f = pow2f(k);
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            uint32_t a, b;
            split2(Y[o][2 * d] * f, Y[o][2 * d + 1] * f, a, b);
            X.v[0][o][d / 4][d % 4] = a; X.v[1][o][d / 4][d % 4] = b;
        }
    return k;
}

// rows_sum (mfma_chain.h: the sum over the tile's 32 rows, five DPP adds pairing up as an xor butterfly) of TWO values, as v_add_f32_dpp
// - one instruction per step where update_dpp + add compile to a v_mov_b32_dpp, an add and a zero fill - with the two dependent chains
// interleaved (a DPP read needs two wait states behind the VALU write of its source: the other chain's instruction and one s_nop).
// Same operations in the same order: bit-identical.  Valid on the upper 16 lanes of each half-wave (ROWS_SUM_LANE).
__device__ __forceinline__ void rows_sum2(float &a, float &b) {
    asm("s_nop 1\n\t"             // (the compiler does not know that what follows reads its operands through DPP)
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf"
        : "+v"(a), "+v"(b));
}

}  // namespace f16l

using namespace f16l;


template <int KIND>
__global__ __launch_bounds__(256, 1) void trunk_f16l_kernel(const TrunkParams p, const TrunkF16Scales sc) {
    constexpr int W1B = (KIND == 3) ? 16 : 8;
    constexpr int W1 = W1B * 32;
    constexpr int NSLOT = (KIND == 3) ? 8 + 7 * 4 : 8 * 4;
    __shared__ uint32_t smask[NSLOT][256];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ __attribute__((aligned(16))) v4f32 wbuf[NBUF * CH * 64];      // the shared weight stream: 3 slots of 32 KiB (4 of 16 KiB with -DDGDM_F16_CHUNK=16)
    // the stack's biases and the output layer's three rows, copied once: read from global memory where they are used, every one of these
    // 32-load bursts is a full memory round trip behind the stream's prefetches (3.5 k cycles per layer prologue, 15 k for the output phase)
    __shared__ __attribute__((aligned(16))) v4f32 small[(8 * 256 + 768) / 4];
    const bool live = blockIdx.x * 4 + wave < p.ntiles;      // a wave without a tile runs the last tile again (the barriers need all four waves)
    const int tile = min(blockIdx.x * 4 + wave, p.ntiles - 1);
    const int n = lane & 31;
    const int h4 = (lane >> 5) * 4;
    const int voff = lane * 16;

    const int per_chain = p.B * p.tiles_per_b;
    const int chain = tile / per_chain;
    const int rem = tile - chain * per_chain;
    const int b = rem / p.tiles_per_b;
    const int c = (rem - b * p.tiles_per_b) * 32 + n;
    const bool valid = c < p.C;
    const int64_t r = (int64_t)(valid ? c : p.C - 1) * p.B + b;
    const float *arow = p.Atab + (size_t)(chain * p.B + b) * W1;
    const float4 *ptile = reinterpret_cast<const float4 *>(p.PtabT) + (size_t)(rem - b * p.tiles_per_b) * W1B * 4 * 64 + lane;

    f32x16 Y[8];
    uint32_t m[4];
    int slot = 0;
    int E = 0;                                                // Y = true values x 2^E for this lane's row
    HSTAMP(0);
    const wrsrc_t rsF = weight_rsrc(p.Wfwd, p.fwd_bytes);
    LStream ls;
    ls.buf = (lds_f4_t *)wbuf; ls.voff = voff; ls.wave = wave; ls.lane = lane;
    v4f32 wn[4];
    // the tile's own input first (2-D: the two table terms; 3-D: the embedding row - HBM, the kernel's only traffic of size R): its latency
    // runs beside the stream's first chunks
    if (KIND == 2) {
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
        const float *xrow = p.xtab ? p.xtab[chain] + (size_t)p.xidx[(size_t)chain * p.xstride + r] * 256 : p.xobj + ((size_t)chain * p.xstride + r) * 256;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = feat4(xrow, o, q, h4);
                Y[o][4 * q + 0] = v.x; Y[o][4 * q + 1] = v.y; Y[o][4 * q + 2] = v.z; Y[o][4 * q + 3] = v.w;
            }
        }
    }
    for (int i = tid; i < p.n_mid * 64; i += 256) small[i] = *reinterpret_cast<const v4f32 *>(p.bf[i >> 6] + 4 * (i & 63));
    if (tid < 192) small[512 + tid] = *reinterpret_cast<const v4f32 *>(p.Wout + 4 * tid);
    ls.start(rsF, 0);                                         // (its barriers publish `small`)
    const lds_f4_t *sbias = (const lds_f4_t *)small, *swout = (const lds_f4_t *)small + 512;

    if (KIND == 3) {
        // ---- 3-D layers 1 and 2, streamed over the 16 blocks of the 512-wide layer 1.  Layer 1's input is the embedding row (scale from its
        // largest entry); layer 2's input arrives block by block and takes its row scale from a bound (below)
#pragma unroll
        for (int j = 0; j < 4; ++j) wn[j] = ls.read(j);
        HSTAMP(36);
        Act2 X;
        const int kx = split_rows(Y, X);
        HSTAMP(37);
        const float un1 = pow2f(-(kx + sc.ew_l1));            // layer-1 accumulators -> true values
        // The f16 scale of layer 2's input rows, from a bound on layer 1's output that is known now (the file header says why a bound
        // does): |z_j| <= max |A[finger]| + max |P[cell]| + ||W1o_j||_1 max |x|, and max |x| < 2^(13 - kx) by the choice of kx.
        float am = 0.f;
        {
            const float4 a0 = *reinterpret_cast<const float4 *>(arow + 8 * lane), a1 = *reinterpret_cast<const float4 *>(arow + 8 * lane + 4);
            am = fmaxf(fmaxf(fmaxf(fabsf(a0.x), fabsf(a0.y)), fmaxf(fabsf(a0.z), fabsf(a0.w))), fmaxf(fmaxf(fabsf(a1.x), fabsf(a1.y)), fmaxf(fabsf(a1.z), fabsf(a1.w))));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
        }
        const float zb = am + p.Pmax[valid ? c : p.C - 1] + sc.l1_norm1 * pow2f(13 - kx);
        const int eb = (int)((__float_as_uint(zb) >> 23) & 0xffu);
        const int k2 = (zb > 0.f && eb < 255) ? min(max(12 + 127 - eb, -100), 100) : 0;      // zb 2^k2 in [2^12, 2^13)
        const float f2 = pow2f(k2);
        E = k2 + sc.ew_l2;                                    // layer-2 accumulators = true values x 2^E
        const float fb2 = pow2f(E);
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = feat4(p.b2, o, q, h4);
                Y[o][4 * q + 0] = b4.x * fb2; Y[o][4 * q + 1] = b4.y * fb2; Y[o][4 * q + 2] = b4.z * fb2; Y[o][4 * q + 3] = b4.w * fb2;
            }
        }
        f32x16 zero, tt;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) zero[rr] = 0.f;
        // the block's table terms: requested a whole block ahead, added up where they are used - the wait for these loads is a wait for every
        // load issued before them (vmcnt counts in order), stream chunks included, so it must not come right behind the request
        float4 tv[4], tw[4];
        auto table_terms_request = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tv[q] = *reinterpret_cast<const float4 *>(arow + 32 * kb + 8 * q + h4);
                tw[q] = ptile[(kb * 4 + q) * 64];
            }
        };
        auto table_terms = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tt[4 * q + 0] = tv[q].x + tw[q].x; tt[4 * q + 1] = tv[q].y + tw[q].y; tt[4 * q + 2] = tv[q].z + tw[q].z; tt[4 * q + 3] = tv[q].w + tw[q].w;
            }
        };
        table_terms_request(0);
        HSTAMP(38);
        for (int blk = 0; blk < 16; blk += 2) {
            uint32_t bits2 = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kb = blk + e;
                if (kb == 8) HSTAMP(32);
                f32x16 z = block_out(ls, wn, X, zero, zero, [](int) __attribute__((always_inline)) {});      // leaves wn = entries 0 .. 3 of layer 2's pass
                if (kb == 8) HSTAMP(33);
                table_terms();
                table_terms_request((kb + 1) & 15);
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) z[rr] = fmaf(z[rr], un1, tt[rr]);       // the scaled sum back to true units (exact) + the table terms: one rounding
                hu32x4_t ah[2], al[2];
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    float lo = z[2 * d], hi = z[2 * d + 1];
                    const int sh = 2 * d + 16 * e;
                    bits2 |= (lo > 0.f ? 1u : 0u) << sh;
                    bits2 |= (hi > 0.f ? 1u : 0u) << (sh + 1);
                    asm("v_max_f32 %0, 0, %1" : "=v"(lo) : "v"(lo));
                    asm("v_max_f32 %0, 0, %1" : "=v"(hi) : "v"(hi));
                    uint32_t a, bb;
                    split2(lo * f2, hi * f2, a, bb);
                    ah[d / 4][d % 4] = a; al[d / 4][d % 4] = bb;
                }
                if (kb == 8) HSTAMP(34);
                // layer 2: 32 entries = two chunks, eight groups of [A.h A.l B.h B.l]; the group's operands are read one group ahead (wn holds
                // the first group on entry, the next pass' entries 0 .. 3 - layer 1's next block, or the stack's first group - on return)
#pragma unroll
                for (int gq = 0; gq < 8; ++gq) {
                    const int pp = gq / 2, sx = gq % 2;
                    v4f32 w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = wn[j];
                    if ((gq & (CG - 1)) == CG - 1) ls.advance();
#pragma unroll
                    for (int j = 0; j < 4; ++j) wn[j] = ls.read(((gq + 1) & (CG - 1)) * 4 + j);
                    F16_STEP(Y[2 * pp], Y[2 * pp + 1], w, ah[sx], al[sx]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kb == 8) HSTAMP(35);
            }
            smask[blk / 2][tid] = bits2;
        }
        slot = 8;                                             // wn: entries 0 .. 3 of the stack's first chunk already; E: layer 2's row scale
    }
    if (KIND == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wn[j] = ls.read(j);
    }
    constexpr int BASE = (KIND == 3) ? 8 : 0;
    f32x16 Z[8];
    HSTAMP(1);
    for (int l = 0; l < p.n_mid; ++l) {
        stream_layer<true, true>(ls, wn, sbias + 64 * l, Y, Z, smask, BASE + 4 * l, tid, h4, E, sc.ew_mid[l]);
#pragma unroll
        for (int o = 0; o < 8; ++o) Y[o] = Z[o];
        HSTAMP(2 + l);
    }
    slot = BASE + 4 * p.n_mid;
    relu_mask<8>(Y, m);
#pragma unroll
    for (int i = 0; i < 4; ++i) smask[slot + i][tid] = m[i];
    slot += 4;

    // ---- output layer (256 -> 3) on the VALU on the scaled activations, 2^-E taken out of the three sums
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4f32 w0 = swout[(32 * o + 8 * q + h4) >> 2], w1 = swout[64 + ((32 * o + 8 * q + h4) >> 2)], w2 = swout[128 + ((32 * o + 8 * q + h4) >> 2)];
            const float x0 = Y[o][4 * q + 0], x1 = Y[o][4 * q + 1], x2 = Y[o][4 * q + 2], x3 = Y[o][4 * q + 3];
            s0 = fmaf(w0.w, x3, fmaf(w0.z, x2, fmaf(w0.y, x1, fmaf(w0.x, x0, s0))));
            s1 = fmaf(w1.w, x3, fmaf(w1.z, x2, fmaf(w1.y, x1, fmaf(w1.x, x0, s1))));
            s2 = fmaf(w2.w, x3, fmaf(w2.z, x2, fmaf(w2.y, x1, fmaf(w2.x, x0, s2))));
        }
    }
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    const float unE = pow2f(-E);
    const float d0 = fmaf(s0, unE, p.bout[0]), d1 = fmaf(s1, unE, p.bout[1]), d2 = fmaf(s2, unE, p.bout[2]);

    const wrsrc_t rsB = weight_rsrc(p.Wbwd, p.bwd_bytes);
    ls.start(rsB, 0);                                         // in flight while the objective runs on the VALU
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
            const v4f32 w0 = swout[(32 * o + 8 * q + h4) >> 2], w1 = swout[64 + ((32 * o + 8 * q + h4) >> 2)], w2 = swout[128 + ((32 * o + 8 * q + h4) >> 2)];
            Y[o][4 * q + 0] = fmaf(g2, w2.x, fmaf(g1, w1.x, g0 * w0.x));
            Y[o][4 * q + 1] = fmaf(g2, w2.y, fmaf(g1, w1.y, g0 * w0.y));
            Y[o][4 * q + 2] = fmaf(g2, w2.z, fmaf(g1, w1.z, g0 * w0.z));
            Y[o][4 * q + 3] = fmaf(g2, w2.w, fmaf(g1, w1.w, g0 * w0.w));
        }
    }
    E = 0;                                                    // the gradient seed is in true units
#pragma unroll
    for (int j = 0; j < 4; ++j) wn[j] = ls.read(j);

    // ---- backward through the 256 -> 256 layers
    HSTAMP(10);
    for (int l = p.n_mid - 1; l >= 0; --l) {
        stream_layer<false, false>(ls, wn, nullptr, Y, Z, smask, BASE + 4 + 4 * l, tid, h4, E, sc.ew_mid[l]);
#pragma unroll
        for (int o = 0; o < 8; ++o) Y[o] = Z[o];
        HSTAMP(11 + (p.n_mid - 1 - l));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = smask[BASE + i][tid];
    apply_mask<8>(Y, m);

    float *dst = p.partial + (size_t)tile * W1;
    if (KIND == 2) {
        const float un = pow2f(-E);                           // per row: back to true units before the 32 rows are folded
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = rows_sum(Y[o][4 * q + 0] * un); v.y = rows_sum(Y[o][4 * q + 1] * un);
                v.z = rows_sum(Y[o][4 * q + 2] * un); v.w = rows_sum(Y[o][4 * q + 3] * un);
                if (n == ROWS_SUM_LANE && live) *reinterpret_cast<float4 *>(dst + 32 * o + 8 * q + h4) = v;
            }
        }
    } else {
        // 3-D: one more layer back (256 -> 512), block by block, straight into the fold
        HSTAMP(40);
        Act2 X;
        const int kt = split_rows(Y, X);
        const float un = pow2f(-(E + kt + sc.ew_l2));
        f32x16 zero;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) zero[rr] = 0.f;
        f32x16 g = block_out(ls, wn, X, zero, zero, [](int) __attribute__((always_inline)) {});
        float4 acc;
        // two values at a time (odd K-steps): the two sums' dependent DPP chains interleaved in one block of assembly (rows_sum2)
        auto fold_one = [&](const int rr, const int kb, const uint32_t bits) __attribute__((always_inline)) {
            if ((rr & 1) == 0) return;
            float a = apply_bit(g[rr - 1] * un, bits, rr - 1), b = apply_bit(g[rr] * un, bits, rr);
            rows_sum2(a, b);
            if (rr % 4 == 1) { acc.x = a; acc.y = b; }
            else {
                acc.z = a; acc.w = b;
                if (n == ROWS_SUM_LANE && live) *reinterpret_cast<float4 *>(dst + 32 * kb + 8 * (rr / 4) + h4) = acc;
            }
        };
        HSTAMP(41);
        for (int kb = 1; kb < 16; ++kb) {
            if (kb == 8) HSTAMP(42);
            const uint32_t bits = smask[(kb - 1) / 2][tid] >> (16 * ((kb - 1) & 1));
            const f32x16 gn = block_out(ls, wn, X, zero, zero, [&](const int ks) __attribute__((always_inline)) { fold_one(ks, kb - 1, bits); });
            g = gn;
            if (kb == 8) HSTAMP(43);
        }
        const uint32_t bits = smask[7][tid] >> 16;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) fold_one(rr, 15, bits);
    }
    HSTAMP(20);
#ifdef DGDM_F16_STAMPS
    if (blockIdx.x == gridDim.x / 2 && tid == 0) { g_f16l_stamps[21] = ls.stall_vm; g_f16l_stamps[22] = ls.stall_bar; }
#endif
}

int trunk_f16l_launch(int kind, const TrunkParams &p, const TrunkF16Scales &sc, hipStream_t s) {
    if (p.n_mid != (kind == 3 ? 6 : 7)) return DGDM_EINVAL;
    const int grid = (p.ntiles + 3) / 4;
    if (grid == 0) return DGDM_OK;
    const double rows = (double)(p.ntiles / std::max(1, p.tiles_per_b)) * p.C;
    const double mid = 2.0 * 256 * 256 * p.n_mid;
    const double per_row = (kind == 3) ? (2.0 * 256 * 512 * 2 + mid) + (2.0 * 256 * 512 + mid) : 2.0 * mid;
    prof_begin(s, DGDM_STAGE_TRUNK);
    if (kind == 2) hipLaunchKernelGGL((trunk_f16l_kernel<2>), dim3(grid), dim3(256), 0, s, p, sc);
    else hipLaunchKernelGGL((trunk_f16l_kernel<3>), dim3(grid), dim3(256), 0, s, p, sc);
    DGDM_HIP_CHECK(hipGetLastError());
    prof_end(s, DGDM_STAGE_TRUNK, rows * per_row);
#ifdef DGDM_F16_STAMPS
    {
        long long st[48];
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_f16l_stamps), sizeof(st));
        fprintf(stderr, "f16l stamps kind %d:", kind);
        long long prev = st[0];
        for (int i = 1; i <= 20; ++i) if (st[i]) { fprintf(stderr, " [%d]%lld", i, st[i] - prev); prev = st[i]; }
        fprintf(stderr, " total %lld; advance(): counted wait %lld, barrier %lld; last fwd layer: prologue %lld first items %lld loop %lld; bwd: %lld %lld %lld; front block 8: layer-1 block %lld, epilogue %lld, layer-2 pass %lld; before the front loop: stream start + embedding row %lld, its split %lld, bias + first table terms %lld; tail: split + first block %lld, block 8 with the fold of block 7 %lld\n", st[20] - st[0], st[21], st[22],
                st[23] - st[1 + p.n_mid - 1], st[24] - st[23], st[25] - st[24], st[26] - st[10 + p.n_mid - 1], st[27] - st[26], st[28] - st[27], st[33] - st[32], st[34] - st[33], st[35] - st[34], st[36] - st[0], st[37] - st[36], st[38] - st[37], st[41] - st[40], st[43] - st[42]);
    }
#endif
    return DGDM_OK;
}

}  // namespace dgdm
