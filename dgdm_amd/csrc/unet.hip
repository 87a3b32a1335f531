// ConditionalUnet1D forward (generator/diffusion_utils.py:123-285), one workgroup per sample, one launch per forward.
//
// The whole network runs inside the launch: activations live in LDS as [position + 2][channels + 8]
// (channel-contiguous rows, two zero halo rows on each side, <= 130 KiB for the shipped shapes), weights stream from
// L2, GroupNorm statistics are wave-shuffle reductions, Mish / FiLM / residual adds are fused into the passes that
// already touch the data.  One launch per denoise step instead of ~90 eager ops.
//
// Convolutions are implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact f32): D[co][pos] += W[co][ci,tap] * in[ci][pos+tap-pad],
// weights as the A operand from a pre-arranged global image (one coalesced float4 per lane feeds 4 K-steps),
// activations as the B operand straight from LDS: K-step c of a 16-channel group takes channel 4q + c from lane group q, so a
// lane's operands of four K-steps are four consecutive channels = ONE ds_read_b128 (it was four ds_read_b32, each a 2-way bank
// conflict: rows j and j + 8 of a tile met in one bank with the old row stride of C + 4 floats).  Row stride C + 8 floats
// (8 mod 64 dword banks) makes every 16-lane group of that b128 read hit 64 distinct banks.  Each wave
// owns pairs of 16-channel output tiles and all position tiles, so a weight fragment is loaded once and reused for
// every position tile.  The two convolutions that touch a single channel (1 -> d0 input conv, d0 -> 1 output conv) run
// on the VALU.
#include "common.h"
#include "unet.h"
#include <algorithm>

namespace dgdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef UNET_THREADS
#define UNET_THREADS 512
#endif
#ifndef DGDM_UNET_SLAB_GROUPS
#define DGDM_UNET_SLAB_GROUPS 2
#endif
// f16x3 convolutions: channel groups of 32 split per slab pass (conv_mfma_f16x3; every width is a multiple of 64).  Measured per 1024
// samples at L = 42: 1.52 ms with one group per pass, 1.43 with two (half the barriers).
constexpr int UNET_SLAB_GROUPS = DGDM_UNET_SLAB_GROUPS;
constexpr int UNET_ROW_PAD = 8;      // LDS activation rows are C + 8 floats: see the header comment (bank mapping of the B-operand reads)
// Activations live in LDS.  The device functions below are real calls (not inlined into the kernel), so a plain `float *`
// parameter would be a generic pointer: every access a flat_load/flat_store with 64-bit address arithmetic on the VALU
// (measured: 5.9 VALU instructions per MFMA in this kernel).  Address-space-3 pointers give ds_read/ds_write with
// immediate offsets.
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) f32x4 lds_f4;
// ... and the weight images are global: a generic pointer loaded from the parameter struct would make them flat_load, which
// also counts on lgkmcnt, so every wait for an LDS operand would wait for the weight prefetch as well.
typedef const __attribute__((address_space(1))) f32x4 glb_f4;

// Experiment build only (DGDM_EXTRA_FLAGS=-DDGDM_UNET_CLOCKS): thread 0 of every workgroup stamps the shader clock at every phase
// boundary; dgdm_debug_unet_clocks copies the stamps out (scripts/unet_phases.py prints the mean time per phase).
#ifdef DGDM_UNET_CLOCKS
__device__ long long unet_clk[1024 * 64];
#define UCLK()                                                                                              \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 1024 && clk_i < 64) unet_clk[blockIdx.x * 64 + clk_i] = (long long)__builtin_readcyclecounter(); \
        ++clk_i;                                                                                            \
    } while (0)
#define UCLK_ARG , int &clk_i
#define UCLK_PASS , clk_i
#else
#define UCLK() do {} while (0)
#define UCLK_ARG
#define UCLK_PASS
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float mish(float x) {
    // torch.nn.functional.mish = x * tanh(softplus(x)).  With e = exp(x): tanh(log(1 + e)) = n / (n + 2), n = e (e + 2), which
    // needs one exp and one division instead of log1p + tanh.  torch's softplus returns x above its threshold 20, where
    // tanh(x) rounds to 1 in float32 - as does n / (n + 2) (n > 2e17) - so clamping the exponent at 20 reproduces that branch
    // and keeps e*e finite.  The division is v_rcp_f32 + one correction step (q = n r; q += (n - q d) r: the residual is exact in the
    // fma, the quotient within an ulp of the correctly rounded one) instead of the IEEE sequence (v_div_scale x 2, v_rcp, four
    // fma, v_div_fmas, v_div_fixup): d = n + 2 lies in [2, 2.4e17], where none of what that sequence guards against can happen.
    // The GroupNorm + Mish passes of the eps-net are bound by VALU throughput.  Relative error ~2e-7.
    const float e = __expf(fminf(x, 20.f));
    const float n = e * (e + 2.f);
    const float d = n + 2.f;
    const float r = __builtin_amdgcn_rcpf(d);
    float q = n * r;
    q = fmaf(fmaf(-q, d, n), r, q);
    return x * q;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Generic strided multi-tap convolution as implicit GEMM.
//   out[(l*ostride + ooff) + 2][co] (+)= bias[co] + sum_{t < ntaps} sum_ci W_t[co][ci] * in[l*istride + ioff0 + t*iostep + 2][ci],  l < Lout
// img: [Cout/16][ntaps][Cin/16][64 lanes] float4; lane (i = l&15, q = l>>4), component c -> W_t[16 mt + i][16 g + 4 q + c].
// MODE 0: store, 1: add to what is in `out` (residual).
struct ConvArgs {
    const float4 *img;
    const float  *bias;
    int cin, cout, ntaps, istride, ostride, ooff;
    int ioff0, iostep;        // input row offset of tap t = ioff0 + t*iostep (an indexed array here ends up in scratch)
    int bf16;                 // what img is: 0 float32 image (conv_mfma), 1 bf16 image (conv_mfma_bf16), 2 two-piece f16 image (conv_mfma_f16x3)
    int ew;                   // f16x3: the image holds the weights times 2^ew
    lds_f *red;               // f16x3: 16 floats of LDS scratch (input_scale_exp)
    lds_f *slab;              // f16x3: the split activations of slab_groups channel groups (conv_mfma_f16x3)
    int slab_groups;
};

template <int NT, int MT, int MODE>
__device__ void conv_mfma(const ConvArgs a, const lds_f *in, int CPi, lds_f *out, int CPo, int Lout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int groups = __builtin_amdgcn_readfirstlane(a.cin >> 4), mtiles = __builtin_amdgcn_readfirstlane(a.cout >> 4);
    int base[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) base[nt] = (min(nt * 16 + j, Lout - 1) * a.istride + 2) * CPi + 4 * q;
    const int iters = __builtin_amdgcn_readfirstlane(a.ntaps * groups);
    for (int mp = wave; mp * MT < mtiles; mp += nwave) {
        int mt[MT];
        glb_f4 *w[MT];
        f32x4 nxt[MT], nxt2[MT];                       // weight fragments of the next two iterations (two L2 latencies of cover)
        // Chunked accumulation: a k-ordered float32 fma chain over all ntaps * cin terms (up to 2560) carries a rounding error that
        // grows with the chain length; the chain is cut every CHUNK iterations (128 input channels of one tap), each piece starts
        // from zero and is added to the running total - 4x less accumulation error for 12 packed adds per 192 MFMAs.  The eps-net's
        // error is what perturbs the chain's x between denoise steps (DESIGN_HISTORY.md 7.2).
        constexpr int CHUNK = 8;
        f32x4 acc[MT][NT], tot[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            mt[m] = min(mp * MT + m, mtiles - 1);
            w[m] = (glb_f4 *)a.img + (size_t)mt[m] * iters * 64 + lane;
            nxt[m] = w[m][0];
            nxt2[m] = w[m][(size_t)min(1, iters - 1) * 64];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { acc[m][nt] = (f32x4)(0.f); tot[m][nt] = (f32x4)(0.f); }
        }
        for (int it0 = 0; it0 < iters; it0 += CHUNK) {          // a chunk = CHUNK groups of one tap (groups is a multiple of CHUNK: launcher)
            const int t = it0 / groups, g0 = it0 - t * groups;
            const int tapoff = (a.ioff0 + t * a.iostep) * CPi + g0 * 16;
            for (int jc = 0; jc < CHUNK; ++jc) {
                const int it = it0 + jc;
                float av[MT][4];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    av[m][0] = nxt[m][0]; av[m][1] = nxt[m][1]; av[m][2] = nxt[m][2]; av[m][3] = nxt[m][3];
                    nxt[m] = nxt2[m];
                }
                // hipcc otherwise proves nxt == w[it] and turns the two-deep prefetch back into load-then-use; an opaque index keeps it
                int pre = min(it + 2, iters - 1);
                asm volatile("" : "+v"(pre));
#pragma unroll
                for (int m = 0; m < MT; ++m) nxt2[m] = w[m][(size_t)pre * 64];
                const int off = tapoff + jc * 16;
                f32x4 bv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = *(const lds_f4 *)(in + base[nt] + off);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[m][nt] = mfma16(av[m][c], bv[nt][c], acc[m][nt]);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { tot[m][nt] += acc[m][nt]; acc[m][nt] = (f32x4)(0.f); }
        }
        // D layout: column (position) = lane & 15, rows (channels) = 4*(lane >> 4) + r
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (mp * MT + m >= mtiles) break;
            const float4 b4 = *reinterpret_cast<const float4 *>(a.bias + mt[m] * 16 + 4 * q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int l = nt * 16 + j;
                if (l < Lout) {
                    lds_f4 *p = (lds_f4 *)(out + ((l * a.ostride + a.ooff) + 2) * CPo + 4 * q + mt[m] * 16);
                    const f32x4 sum = tot[m][nt];
                    f32x4 v = {sum[0] + b4.x, sum[1] + b4.y, sum[2] + b4.z, sum[3] + b4.w};
                    if (MODE == 1) v += *p;
                    *p = v;
                }
            }
        }
    }
}

// The same convolution with bf16 operands (BASELINE configs[4], "bf16 contractions"): v_mfma_f32_16x16x32_bf16, K-step = 32 input
// channels of one tap.  img16: [Cout/16][ntaps][Cin/32][64 lanes][8 bf16]; lane (i = l & 15, kg = l >> 4), slot j ->
// W_t[16 mt + i][32 g + 4 kg + j] for j < 4, [32 g + 16 + 4 kg + j - 4] for j >= 4 (rounded to bf16 on the host).  The activations stay
// float32 in LDS; a lane reads its two runs of 4 consecutive channels with two ds_read_b128 and rounds them with four v_cvt_pk_bf16_f32.  Accumulation, bias and everything
// around the convolution are float32.
typedef __bf16 bf16x8_u __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_u __attribute__((ext_vector_type(2)));
typedef float f32x2_u __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const f32x2_u v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_u));
}

template <int NT, int MT, int MODE>
__device__ void conv_mfma_bf16(const ConvArgs a, const lds_f *in, int CPi, lds_f *out, int CPo, int Lout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int groups = __builtin_amdgcn_readfirstlane(a.cin >> 5), mtiles = __builtin_amdgcn_readfirstlane(a.cout >> 4);
    int base[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) base[nt] = (min(nt * 16 + j, Lout - 1) * a.istride + 2) * CPi + 4 * q;
    const int iters = __builtin_amdgcn_readfirstlane(a.ntaps * groups);
    for (int mp = wave; mp * MT < mtiles; mp += nwave) {
        int mt[MT];
        glb_f4 *w[MT];
        f32x4 nxt[MT], nxt2[MT];
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            mt[m] = min(mp * MT + m, mtiles - 1);
            w[m] = (glb_f4 *)a.img + (size_t)mt[m] * iters * 64 + lane;
            nxt[m] = w[m][0];
            nxt2[m] = w[m][(size_t)min(1, iters - 1) * 64];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = (f32x4)(0.f);
        }
        constexpr int UNR = 4;                                   // groups (32 channels each) is a multiple of 4: the weight-fragment rotation unrolls away
        for (int it0 = 0; it0 < iters; it0 += UNR) {
            const int t = it0 / groups, g0 = it0 - t * groups;
            const int tapoff = (a.ioff0 + t * a.iostep) * CPi + g0 * 32;
#pragma unroll
            for (int jc = 0; jc < UNR; ++jc) {
                const int it = it0 + jc;
                bf16x8_u av[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    av[m] = __builtin_bit_cast(bf16x8_u, nxt[m]);
                    nxt[m] = nxt2[m];
                }
                int pre = min(it + 2, iters - 1);
                asm volatile("" : "+v"(pre));
#pragma unroll
                for (int m = 0; m < MT; ++m) nxt2[m] = w[m][(size_t)pre * 64];
                const int off = tapoff + jc * 32;
                bf16x8_u bv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 lo = *(const lds_f4 *)(in + base[nt] + off), hi = *(const lds_f4 *)(in + base[nt] + off + 16);
                    const u32x4_u pk = {pack2_bf16(lo[0], lo[1]), pack2_bf16(lo[2], lo[3]), pack2_bf16(hi[0], hi[1]), pack2_bf16(hi[2], hi[3])};
                    bv[nt] = __builtin_bit_cast(bf16x8_u, pk);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m], bv[nt], acc[m][nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (mp * MT + m >= mtiles) break;
            const float4 b4 = *reinterpret_cast<const float4 *>(a.bias + mt[m] * 16 + 4 * q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int l = nt * 16 + j;
                if (l < Lout) {
                    lds_f4 *p = (lds_f4 *)(out + ((l * a.ostride + a.ooff) + 2) * CPo + 4 * q + mt[m] * 16);
                    f32x4 v = {acc[m][nt][0] + b4.x, acc[m][nt][1] + b4.y, acc[m][nt][2] + b4.z, acc[m][nt][3] + b4.w};
                    if (MODE == 1) v += *p;
                    *p = v;
                }
            }
        }
    }
}

// The same convolution, float32-grade, on the f16 matrix pipe (the default): every float32 product as THREE f16 products,
//     w x ~ w_h x_h + w_h x_l + w_l x_h,   x_h = f16(x), x_l = f16(x - x_h) (exact residual; 11 + 11 significant bits),
// on v_mfma_f32_16x16x32_f16 with float32 accumulation - the arithmetic of the dynamics trunk (trunk_f16l.hip) and the same two exact
// power-of-two scales that keep both operands inside f16's five exponent bits: one per convolution's weights, fixed on the host (a.ew,
// models_api.hip conv_image_f16x3), and one per (sample, convolution input), from the largest magnitude among the rows the convolution
// reads (input_scale_exp).
// The activations stay float32 in LDS.  Splitting them where they are read would be done ntaps x (output tiles / MT) times over - as
// much VALU time as the MFMAs take (measured: 1.75 ms per 1024 samples that way, 1.43 this way) - so the K loop runs
// channel-group-major: for every 64 input channels the workgroup splits the rows the
// convolution reads ONCE into two 6 KB slabs ([h | l][4 k-groups][rows] x 8 halves each: a lane's B operand is one ds_read_b128 per piece,
// linear in the position = conflict-free), and every tap and output tile consumes the slab.  Image: [Cout/16][Cin/32][ntaps][h | l]
// [64 lanes][8] (group-major to match).  The three products of a K-step run product-major over the wave's NT x MT accumulators (no MFMA
// waits for the one before it), small terms first; the chain is cut after every channel group.  Against float64 the eps-net's output
// is closer than with the float32 MFMA chain (4.3e-7 against 5.6e-7, scripts/unet_check.py) at 1/5 of its matrix-pipe cycles.
typedef _Float16 f16x8_u __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4_u lds_u4;

__device__ __forceinline__ float pow2_f(int e) { return __uint_as_float((uint32_t)(min(max(e, -126), 127) + 127) << 23); }

// an already scaled pair -> packed f16 (h | l), h + l == the pair to 2^-23
__device__ __forceinline__ void split2_f16(const f32x2_u v, uint32_t &ph, uint32_t &pl) {
    const f16x2_u h = __builtin_convertvector(v, f16x2_u);              // v_cvt_pk_f16_f32 (round to nearest even)
    const f32x2_u d = v - __builtin_convertvector(h, f32x2_u);          // exact
    ph = __builtin_bit_cast(uint32_t, h);
    pl = __builtin_bit_cast(uint32_t, __builtin_convertvector(d, f16x2_u));
}

// first / last input row (halo included) a convolution reads
__device__ __forceinline__ int conv_row_lo(const ConvArgs &a) { return a.ioff0 + min(0, (a.ntaps - 1) * a.iostep) + 2; }
__device__ __forceinline__ int conv_row_hi(const ConvArgs &a, int Lout) { return (Lout - 1) * a.istride + a.ioff0 + max(0, (a.ntaps - 1) * a.iostep) + 2; }

// k with max |x| 2^k in [2^12, 2^13) over the input rows a convolution reads (halo rows: zeros); 0 for an all-zero input.
// Called by the whole workgroup; two barriers.
__device__ int input_scale_exp(const ConvArgs &a, const lds_f *in, int CPi, int Lout) {
    lds_f *red = a.red;
    const int r0 = conv_row_lo(a), r1 = conv_row_hi(a, Lout);
    const int c4 = a.cin >> 2, n = (r1 - r0 + 1) * c4;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / c4, c = i - r * c4;
        const f32x4 v = *(const lds_f4 *)(in + (r0 + r) * CPi + 4 * c);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();                                        // the previous call's readers of red[] are done
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    float mm = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) mm = fmaxf(mm, red[w]);
    const int e = (int)((__float_as_uint(mm) >> 23) & 0xffu);
    // (clamped: beyond 2^100 pow2_f's own clamp would make the input scale and its inverse disagree; inputs below 2^-88 keep fewer bits)
    return __builtin_amdgcn_readfirstlane((mm > 0.f && e < 255) ? min(max(12 + 127 - e, -100), 100) : 0);
}

template <int NT, int MT, int MODE>
__device__ void conv_mfma_f16x3(const ConvArgs a, const lds_f *in, int CPi, lds_f *out, int CPo, int Lout, const int kx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int groups = __builtin_amdgcn_readfirstlane(a.cin >> 5), mtiles = __builtin_amdgcn_readfirstlane(a.cout >> 4);
    const int ntaps = __builtin_amdgcn_readfirstlane(a.ntaps);
    const int rlo = conv_row_lo(a), R = conv_row_hi(a, Lout) - rlo + 1;
    lds_u4 *slab = (lds_u4 *)a.slab;                             // [GS channel groups][h | l][4 k-groups][R rows]
    constexpr int GS = UNET_SLAB_GROUPS;                         // channel groups per slab pass
    int brow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) brow[nt] = q * R + min(nt * 16 + j, Lout - 1) * a.istride + a.ioff0 + 2 - rlo;
    const int iters = __builtin_amdgcn_readfirstlane(ntaps * groups);
    const float f = pow2_f(kx), un = pow2_f(-(kx + a.ew));
    const f32x2_u f2 = {f, f};
    // 2^-e_co per output channel (the image's rows carry their own scale, models_api.hip conv_image_f16x3): right behind the image
    const float *unw = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.img) + (size_t)4 * a.cout * a.cin * a.ntaps);
    for (int mp0 = 0; mp0 * MT < mtiles; mp0 += nwave) {        // the same trip count for every wave (barriers inside)
        const int mp = mp0 + wave;
        int mt[MT];
        glb_f4 *w[MT];
        f32x4 nh[MT], nl[MT], n2h[MT], n2l[MT];                  // [h | l] weight fragments of the next two K-steps
        f32x4 acc[MT][NT], tot[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            mt[m] = min(mp * MT + m, mtiles - 1);
            w[m] = (glb_f4 *)a.img + (size_t)mt[m] * iters * 128 + lane;
            nh[m] = w[m][0]; nl[m] = w[m][64];
            const size_t o1 = (size_t)min(1, iters - 1) * 128;
            n2h[m] = w[m][o1]; n2l[m] = w[m][o1 + 64];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { acc[m][nt] = (f32x4)(0.f); tot[m][nt] = (f32x4)(0.f); }
        }
        int it = 0;
        for (int g0 = 0; g0 < groups; g0 += GS) {
            __syncthreads();                                     // the previous pass' readers of the slab are done
            for (int i = threadIdx.x; i < GS * 4 * R; i += blockDim.x) {
                const int gg = i / (4 * R), i1 = i - gg * 4 * R, qq = i1 / R, row = i1 - qq * R;
                const lds_f *src = in + (rlo + row) * CPi + (g0 + gg) * 32 + 4 * qq;
                const f32x4 lo = *(const lds_f4 *)src, hi = *(const lds_f4 *)(src + 16);
                uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
                split2_f16(f32x2_u{lo[0], lo[1]} * f2, h0, l0);
                split2_f16(f32x2_u{lo[2], lo[3]} * f2, h1, l1);
                split2_f16(f32x2_u{hi[0], hi[1]} * f2, h2, l2);
                split2_f16(f32x2_u{hi[2], hi[3]} * f2, h3, l3);
                slab[gg * 8 * R + i1] = u32x4_u{h0, h1, h2, h3};
                slab[gg * 8 * R + 4 * R + i1] = u32x4_u{l0, l1, l2, l3};
            }
            __syncthreads();
            for (int gt = 0; gt < GS * ntaps; ++gt, ++it) {
                const int gg = (GS > 1 && gt >= ntaps) ? 1 : 0, t = gt - gg * ntaps;
                f16x8_u ah[MT], al[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[m] = __builtin_bit_cast(f16x8_u, nh[m]); al[m] = __builtin_bit_cast(f16x8_u, nl[m]);
                    nh[m] = n2h[m]; nl[m] = n2l[m];
                }
                // hipcc otherwise proves nh == w[it] and turns the two-deep prefetch back into load-then-use; an opaque index keeps it
                int pre = min(it + 2, iters - 1);
                asm volatile("" : "+v"(pre));
#pragma unroll
                for (int m = 0; m < MT; ++m) { n2h[m] = w[m][(size_t)pre * 128]; n2l[m] = w[m][(size_t)pre * 128 + 64]; }
                const int ro = gg * 8 * R + t * a.iostep;
                f16x8_u bh[NT], bl[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt] = __builtin_bit_cast(f16x8_u, (u32x4_u)slab[brow[nt] + ro]);
                    bl[nt] = __builtin_bit_cast(f16x8_u, (u32x4_u)slab[4 * R + brow[nt] + ro]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[nt], acc[m][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[nt], acc[m][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[nt], acc[m][nt], 0, 0, 0);
                if (t == ntaps - 1) {                            // the chain is cut after every channel group
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) { tot[m][nt] += acc[m][nt]; acc[m][nt] = (f32x4)(0.f); }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (mp * MT + m >= mtiles) break;
            const float4 b4 = *reinterpret_cast<const float4 *>(a.bias + mt[m] * 16 + 4 * q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int l = nt * 16 + j;
                if (l < Lout) {
                    lds_f4 *p = (lds_f4 *)(out + ((l * a.ostride + a.ooff) + 2) * CPo + 4 * q + mt[m] * 16);
                    const f32x4 sum = tot[m][nt];
                    const float4 u4 = *reinterpret_cast<const float4 *>(unw + mt[m] * 16 + 4 * q);
                    f32x4 v = {fmaf(sum[0], un * u4.x, b4.x), fmaf(sum[1], un * u4.y, b4.y), fmaf(sum[2], un * u4.z, b4.z), fmaf(sum[3], un * u4.w, b4.w)};      // (x 2^-k exact: one rounding, the bias add's)
                    if (MODE == 1) v += *p;
                    *p = v;
                }
            }
        }
    }
}

template <int NT, int MODE>
__device__ void conv_nt(const ConvArgs &a, const lds_f *in, int CPi, lds_f *out, int CPo, int Lout) {
    // two output tiles per wave share the activation fragments when there are enough tiles to keep every wave busy
    if (a.bf16 == 2) {
        const int kx = input_scale_exp(a, in, CPi, Lout);
        if ((a.cout >> 4) >= 2 * (int)(blockDim.x >> 6)) conv_mfma_f16x3<NT, 2, MODE>(a, in, CPi, out, CPo, Lout, kx);
        else conv_mfma_f16x3<NT, 1, MODE>(a, in, CPi, out, CPo, Lout, kx);
        return;
    }
    if (a.bf16) {
        if ((a.cout >> 4) >= 2 * (int)(blockDim.x >> 6)) conv_mfma_bf16<NT, 2, MODE>(a, in, CPi, out, CPo, Lout);
        else conv_mfma_bf16<NT, 1, MODE>(a, in, CPi, out, CPo, Lout);
        return;
    }
    if ((a.cout >> 4) >= 2 * (int)(blockDim.x >> 6)) conv_mfma<NT, 2, MODE>(a, in, CPi, out, CPo, Lout);
    else conv_mfma<NT, 1, MODE>(a, in, CPi, out, CPo, Lout);
}

template <int MODE>
__device__ void conv(const ConvArgs &a, const lds_f *in, int CPi, lds_f *out, int CPo, int Lout) {
    if (Lout <= 16) conv_nt<1, MODE>(a, in, CPi, out, CPo, Lout);
    else if (Lout <= 32) conv_nt<2, MODE>(a, in, CPi, out, CPo, Lout);
    else if (Lout <= 48) conv_nt<3, MODE>(a, in, CPi, out, CPo, Lout);
    else conv_nt<4, MODE>(a, in, CPi, out, CPo, Lout);
}

__device__ ConvArgs conv_args(const float *img, const float *bias, int cin, int cout, int ntaps, int pad, int istride, int bf16, int ew, lds_f *red,
                              int slab_groups) {
    ConvArgs a;
    a.bf16 = bf16; a.ew = ew; a.red = red; a.slab = red + 16; a.slab_groups = slab_groups;
    a.img = reinterpret_cast<const float4 *>(img); a.bias = bias; a.cin = cin; a.cout = cout; a.ntaps = ntaps;
    a.istride = istride; a.ostride = 1; a.ooff = 0;
    a.ioff0 = -pad; a.iostep = 1;
    return a;
}

__device__ void zero_halo(lds_f *buf, int CP, int C, int L) {
    for (int i = threadIdx.x; i < 4 * C; i += blockDim.x) {
        const int r = i / C, c = i - r * C;
        buf[(r < 2 ? r : L + r) * CP + c] = 0.f;
    }
}

// GroupNorm(groups, C) -> Mish -> optional FiLM (scale*y + shift), in place on a [pos+2][CP] buffer.  (diffusion_utils.py:65-69,113-116)
// One group g, by the calling wave.
__device__ __forceinline__ void gn_group(lds_f *buf, int CP, int C, int L, int groups, int g, const float *__restrict__ gamma, const float *__restrict__ beta,
                                         const lds_f *film /*LDS [2C] or null*/, __attribute__((address_space(3))) uint32_t *amax = nullptr, bool want_amax = false) {
    const int lane = threadIdx.x & 63;
    const int cg = C / groups, cnt = cg * L;
    // lane -> (row offset, channel) without a division per element when the group width divides the wave (16 or 32 here)
    const bool pow2 = cg <= 64 && (64 % cg) == 0;
    const int rstep = pow2 ? 64 / cg : 0, c0 = pow2 ? lane % cg : 0, l0 = pow2 ? lane / cg : 0;
    lds_f *gb = buf + 2 * CP + g * cg;
    if (pow2 && (L + rstep - 1) / rstep <= 16) {
        // the lane's (at most 16) values in registers: one pipelined LDS read and one write instead of three dependent read passes (each
        // iteration of those exposed an LDS round trip: 27 k cycles per GroupNorm of 4 x 8 groups on 8 waves, measured).  Same operations in
        // the same order as the loops below: identical bits.
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int l = l0 + i * rstep; v[i] = l < L ? gb[l * CP + c0] : 0.f; }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (l0 + i * rstep < L) s += v[i];
        const float mean = wave_sum(s) / (float)cnt;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (l0 + i * rstep < L) { const float d = v[i] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.f / sqrtf(wave_sum(q) / (float)cnt + 1e-5f);
        const int ch = g * cg + c0;
        const float ga = gamma[ch] * rstd, be = beta[ch];
        const float fs = film ? film[ch] : 1.f, fb = film ? film[C + ch] : 0.f;
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int l = l0 + i * rstep;
            if (l < L) {
                float y = mish((v[i] - mean) * ga + be);
                if (film) y = fs * y + fb;
                gb[l * CP + c0] = y;
                mx = fmaxf(mx, fabsf(y));
            }
        }
        if (want_amax) {                                   // largest magnitude of what was written (the batched form's fused blocks scale their second convolution by it)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) atomicMax((uint32_t *)amax, __float_as_uint(mx));
        }
    } else if (pow2) {
        float s = 0.f;
        for (int l = l0; l < L; l += rstep) s += gb[l * CP + c0];
        const float mean = wave_sum(s) / (float)cnt;
        float q = 0.f;
        for (int l = l0; l < L; l += rstep) { const float d = gb[l * CP + c0] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.f / sqrtf(wave_sum(q) / (float)cnt + 1e-5f);
        const int ch = g * cg + c0;
        const float ga = gamma[ch] * rstd, be = beta[ch];
        const float fs = film ? film[ch] : 1.f, fb = film ? film[C + ch] : 0.f;
        float mx = 0.f;
        for (int l = l0; l < L; l += rstep) {
            float y = mish((gb[l * CP + c0] - mean) * ga + be);
            if (film) y = fs * y + fb;
            gb[l * CP + c0] = y;
            mx = fmaxf(mx, fabsf(y));
        }
        if (want_amax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) atomicMax((uint32_t *)amax, __float_as_uint(mx));
        }
    } else {
        float s = 0.f;
        for (int i = lane; i < cnt; i += 64) { const int l = i / cg, c = i - l * cg; s += gb[l * CP + c]; }
        const float mean = wave_sum(s) / (float)cnt;
        float q = 0.f;
        for (int i = lane; i < cnt; i += 64) { const int l = i / cg, c = i - l * cg; const float d = gb[l * CP + c] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.f / sqrtf(wave_sum(q) / (float)cnt + 1e-5f);
        float mxg = 0.f;
        for (int i = lane; i < cnt; i += 64) {
            const int l = i / cg, c = i - l * cg, ch = g * cg + c;
            float y = mish((gb[l * CP + c] - mean) * (gamma[ch] * rstd) + beta[ch]);
            if (film) y = film[ch] * y + film[C + ch];
            gb[l * CP + c] = y;
            mxg = fmaxf(mxg, fabsf(y));
        }
        if (want_amax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mxg = fmaxf(mxg, __shfl_xor(mxg, o));
            if (lane == 0) atomicMax((uint32_t *)amax, __float_as_uint(mxg));
        }
    }
}
__device__ void gn_mish_film(lds_f *buf, int CP, int C, int L, int groups, const float *__restrict__ gamma, const float *__restrict__ beta,
                             const lds_f *film /*LDS [2C] or null*/) {
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int g = wave; g < groups; g += nwave) gn_group(buf, CP, C, L, groups, g, gamma, beta, film);
}

// sum_k WT[k][n] x[k] (x in LDS, K a multiple of 8): eight weight loads in flight per step, the same ascending-k fmaf chain
__device__ __forceinline__ float dot_kn(const float *__restrict__ WT, int N, int n, const lds_f *x, int K) {
    float acc = 0.f;
    for (int k = 0; k < K; k += 8) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = WT[(size_t)(k + j) * N + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf(w[j], x[k + j], acc);
    }
    return acc;
}

// y[n] = b[n] + sum_k WT[k][n] x[k],  x in LDS
__device__ void matvec(const float *__restrict__ WT, const float *__restrict__ b, const lds_f *x, lds_f *y, int K, int N) {
    for (int n = threadIdx.x; n < N; n += blockDim.x) y[n] = dot_kn(WT, N, n, x, K) + b[n];
}

struct Bufs { lds_f *A, *B, *C, *D, *film, *cond, *tmp, *xin, *scr; int bf16, slab_groups; };      // film: [8 blocks][2 * cmax] FiLM scale|shift, all computed up front

// ConditionalResidualBlock1D.forward (diffusion_utils.py:101-120): x(in, cin channels) -> out; t1 scratch.
// `out` may be a wider buffer (row stride CPout >= cout + 4): the concat buffer of the up path.
__device__ void res_block(const UnetRes &w, const lds_f *in, lds_f *t1, lds_f *out, int CPout, int L, int cond_dim, int groups, const Bufs s,
                          const lds_f *film UCLK_ARG) {
    const int CPi = w.cin + UNET_ROW_PAD, CPo = w.cout + UNET_ROW_PAD;
    // the block's FiLM vector (cond_encoder: Mish -> Linear(cond_dim, 2 cout)) was computed with the other seven before the first block
    if (w.cin == 1) {   // first block: single input channel held in s.xin[pos + 2]; conv k5 and the 1x1 residual on the VALU
        for (int i = threadIdx.x; i < L * w.cout; i += blockDim.x) {
            const int l = i / w.cout, co = i - l * w.cout;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 5; ++t) acc = fmaf(w.c0_w[t * w.cout + co], s.xin[l + t], acc);
            t1[(l + 2) * CPo + co] = acc + w.c0_b[co];
        }
    } else {
        conv<0>(conv_args(w.c0_w, w.c0_b, w.cin, w.cout, 5, 2, 1, s.bf16, w.c0_e, s.scr, s.slab_groups), in, CPi, t1, CPo, L);
    }
    zero_halo(t1, CPo, w.cout, L);
    __syncthreads();
    UCLK();
    gn_mish_film(t1, CPo, w.cout, L, groups, w.g0_w, w.g0_b, film);
    __syncthreads();
    UCLK();
    conv<0>(conv_args(w.c1_w, w.c1_b, w.cout, w.cout, 5, 2, 1, s.bf16, w.c1_e, s.scr, s.slab_groups), t1, CPo, out, CPout, L);
    zero_halo(out, CPout, w.cout, L);
    __syncthreads();
    UCLK();
    gn_mish_film(out, CPout, w.cout, L, groups, w.g1_w, w.g1_b, nullptr);
    __syncthreads();
    UCLK();
    if (w.cin == 1) {
        for (int i = threadIdx.x; i < L * w.cout; i += blockDim.x) {
            const int l = i / w.cout, co = i - l * w.cout;
            out[(l + 2) * CPout + co] += fmaf(w.res_w[co], s.xin[l + 2], w.res_b[co]);
        }
    } else if (w.res_w) {
        conv<1>(conv_args(w.res_w, w.res_b, w.cin, w.cout, 1, 0, 1, s.bf16, w.res_e, s.scr, s.slab_groups), in, CPi, out, CPout, L);   // residual 1x1 conv, added in place
    } else {
        for (int i = threadIdx.x; i < L * w.cout; i += blockDim.x) {
            const int l = i / w.cout, c = i - l * w.cout;
            out[(l + 2) * CPout + c] += in[(l + 2) * CPi + c];
        }
    }
    __syncthreads();
    UCLK();
}

// `pp` points at the UnetParams in device memory: passing the struct by value and handing references to its members to the
// (non-inlined) block functions made the compiler copy all 1.1 KB of it to scratch in every thread.
__global__ __launch_bounds__(UNET_THREADS) void unet_kernel(const UnetParams *__restrict__ pp, int bufA, int bufS, const float *__restrict__ sample,
                                                   const int *__restrict__ timestep, float *__restrict__ eps, int L, int slab_groups) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    lds_f *lds = (lds_f *)lds_raw;
    const UnetParams &p = *pp;
    const int b = blockIdx.x, t = threadIdx.x;
    const int L2 = (L - 1) / 2 + 1;                 // Conv1d(k3, s2, p1)
    Bufs s;
    s.A = lds;
    s.B = s.A + bufA;
    s.C = s.B + bufS;
    s.D = s.C + bufS;
    s.film = s.D + bufS;
    s.cond = s.film + 8 * 2 * p.cmax;
    s.tmp = s.cond + p.dsed;
    s.xin = s.tmp + 4 * p.dsed;
    s.scr = s.xin + ((L + 4 + 3) & ~3);            // f16x3: 16 floats for the scale reduction, then the activation slab(s)
    s.bf16 = p.bf16;
    s.slab_groups = slab_groups;
    const int G = p.groups;
#ifdef DGDM_UNET_CLOCKS
    int clk_i = 0;
#endif
    UCLK();

    // ---- diffusion_step_encoder: SinusoidalPosEmb -> Linear -> Mish -> Linear   (diffusion_utils.py:25-37,149-154)
    {
        const int half = p.dsed / 2;
        const float ts = (float)timestep[b];
        if (t < half) {
            const float a = ts * p.freqs[t];
            s.cond[t] = sinf(a);
            s.cond[half + t] = cosf(a);
        }
        __syncthreads();
        matvec(p.se1_wt, p.se1_b, s.cond, s.tmp, p.dsed, 4 * p.dsed);
        __syncthreads();
        for (int i = t; i < 4 * p.dsed; i += blockDim.x) s.tmp[i] = mish(s.tmp[i]);
        __syncthreads();
        matvec(p.se3_wt, p.se3_b, s.tmp, s.cond, 4 * p.dsed, p.dsed);
        __syncthreads();
        for (int i = t; i < p.dsed; i += blockDim.x) s.cond[i] = mish(s.cond[i]);   // every cond_encoder starts with Mish (:90-92)
    }
    for (int i = t; i < L + 4; i += blockDim.x) s.xin[i] = (i >= 2 && i < 2 + L) ? sample[(size_t)b * L + i - 2] : 0.f;
    __syncthreads();
    // all eight cond_encoders at once (they only depend on the step embedding): one phase with 6 independent dot products per
    // thread instead of eight phases of 32 dependent L2 loads each in front of the blocks' first convolutions
    {
        lds_f *films = s.film;
        const int stride = 2 * p.cmax;
        for (int i = t; i < 8 * stride; i += blockDim.x) {
            const int blk = i / stride, n = i - blk * stride, N = 2 * p.res[blk].cout;
            if (n < N) {
                films[i] = dot_kn(p.res[blk].cond_wt, N, n, s.cond, p.dsed) + p.res[blk].cond_b[n];
            }
        }
    }
    __syncthreads();
    UCLK();      // step encoder + FiLM vectors done

    const int CP0 = p.d0 + UNET_ROW_PAD, CP1 = p.d1 + UNET_ROW_PAD, CPcat = 2 * p.d1 + UNET_ROW_PAD;
    res_block(p.res[0], nullptr, s.B, s.C, CP0, L, p.dsed, G, s, s.film + 0 * 2 * p.cmax UCLK_PASS);     // down0.0   1 -> d0
    res_block(p.res[1], s.C, s.B, s.D, CP0, L, p.dsed, G, s, s.film + 1 * 2 * p.cmax UCLK_PASS);         // down0.1   d0 -> d0   (its skip is never consumed, :264-278)
    conv<0>(conv_args(p.down_w, p.down_b, p.d0, p.d0, 3, 1, 2, p.bf16, p.down_e, s.scr, s.slab_groups), s.D, CP0, s.B, CP0, L2);       // Downsample1d (:42)
    zero_halo(s.B, CP0, p.d0, L2);
    __syncthreads();
    UCLK();      // downsample conv
    res_block(p.res[2], s.B, s.C, s.D, CP1, L2, p.dsed, G, s, s.film + 2 * 2 * p.cmax UCLK_PASS);        // down1.0   d0 -> d1
    res_block(p.res[3], s.D, s.B, s.C, CP1, L2, p.dsed, G, s, s.film + 3 * 2 * p.cmax UCLK_PASS);        // down1.1   -> skip, stays in C until the concat
    res_block(p.res[4], s.C, s.B, s.D, CP1, L2, p.dsed, G, s, s.film + 4 * 2 * p.cmax UCLK_PASS);        // mid0
    res_block(p.res[5], s.D, s.B, s.A, CPcat, L2, p.dsed, G, s, s.film + 5 * 2 * p.cmax UCLK_PASS);      // mid1 -> channels [0, d1) of the concat buffer
    for (int i = t; i < (L2 + 4) * p.d1; i += blockDim.x) {           // torch.cat((x, h.pop()), dim=1) (:275): skip -> channels [d1, 2 d1)
        const int r = i / p.d1, c = i - r * p.d1;
        s.A[r * CPcat + p.d1 + c] = s.C[r * CP1 + c];
    }
    __syncthreads();
    UCLK();      // concat copy
    res_block(p.res[6], s.A, s.B, s.D, CP0, L2, p.dsed, G, s, s.film + 6 * 2 * p.cmax UCLK_PASS);        // up0.0   2*d1 -> d0
    res_block(p.res[7], s.D, s.B, s.C, CP0, L2, p.dsed, G, s, s.film + 7 * 2 * p.cmax UCLK_PASS);        // up0.1
    {   // Upsample1d: ConvTranspose1d(d0, d0, 4, 2, 1) (:51): out[2 li] = W1 in[li] + W3 in[li-1];  out[2 li + 1] = W2 in[li] + W0 in[li+1]
        ConvArgs e = conv_args(p.up_w_even, p.up_b, p.d0, p.d0, 2, 0, 1, p.bf16, p.up_e_even, s.scr, s.slab_groups);
        e.ioff0 = 0; e.iostep = -1; e.ostride = 2; e.ooff = 0;
        ConvArgs o = conv_args(p.up_w_odd, p.up_b, p.d0, p.d0, 2, 0, 1, p.bf16, p.up_e_odd, s.scr, s.slab_groups);
        o.ioff0 = 0; o.iostep = 1; o.ostride = 2; o.ooff = 1;
        conv<0>(e, s.C, CP0, s.A, CP0, L2);
        conv<0>(o, s.C, CP0, s.A, CP0, L2);
        zero_halo(s.A, CP0, p.d0, L);                                 // 2*L2 == L (checked by the launcher)
        __syncthreads();
        UCLK();  // upsample convs
    }
    conv<0>(conv_args(p.fin_w, p.fin_b, p.d0, p.d0, 5, 2, 1, p.bf16, p.fin_e, s.scr, s.slab_groups), s.A, CP0, s.B, CP0, L);          // final_conv.0
    __syncthreads();
    UCLK();      // final conv
    gn_mish_film(s.B, CP0, p.d0, L, G, p.fin_gw, p.fin_gb, nullptr);
    __syncthreads();
    UCLK();      // final GroupNorm
    for (int l = t; l < L; l += blockDim.x) {                         // final_conv.1: Conv1d(d0, 1, 1)
        float acc = 0.f;
        for (int c = 0; c < p.d0; ++c) acc = fmaf(p.out_w[c], s.B[(l + 2) * CP0 + c], acc);
        eps[(size_t)b * L + l] = acc + p.out_b[0];
    }
    UCLK();          // output conv
}

// ================================================================================================ batched form (f16x3, large batches)
// One workgroup per sample streams all 7.8 MB of two-piece weight images through its CU for 3 (L = 42) or 2 (L = 21) position tiles per
// weight fragment, pays every convolution's fixed costs (first weight fetch, slab split, barriers) on one sample's worth of MFMAs, and
// pads 21 positions to 32.  For large batches the network runs LAYER BY LAYER instead, UB_S samples per workgroup: activations in global
// memory ([sample][position + 2 halo rows each side][channels], L2 / MALL resident: 22-52 MB per tensor at 1024 samples), one launch per
// convolution with everything that follows it fused (GroupNorm + Mish + FiLM, the block's residual - identity, 1x1 convolution or the
// single-channel input -, Upsample's two phases, the final 1x1 convolution).  A weight fragment now serves UB_S samples' position tiles
// (S x L positions are tiled contiguously: 168 -> 11 tiles, 84 -> 6, instead of 4 x 3 and 4 x 2), the fixed costs are paid once per
// UB_S samples.  The arithmetic is the per-sample kernel's, operation for operation (same images, same slab layout and K order, same
// chunked accumulation, the same per-(sample, convolution input) power-of-two scale - its max now comes from the producing launch's
// epilogue -, gn_group as it is): the two forms return the same bits (tests/test_gpu_parity.py::test_unet_batched_equals_per_sample).
constexpr int UB_THREADS = 512;
#ifdef DGDM_UB_CLOCKS
// experiment build: workgroup 0's thread 0 stamps the shader clock at phase boundaries of the launch numbered DGDM_UB_CLOCKS (printed by the launcher)
__device__ long long g_ub_clk[64];
__device__ int g_ub_clk_n;
#define UBCLK(A_) do { if ((A_).clk_on && blockIdx.x == 0 && threadIdx.x == 0) { const int i_ = g_ub_clk_n; if (i_ < 64) { g_ub_clk[i_] = (long long)__builtin_readcyclecounter(); g_ub_clk_n = i_ + 1; } } } while (0)
#else
#define UBCLK(A_) do { } while (0)
#endif
typedef const __attribute__((address_space(1))) float glb_f;
typedef __attribute__((address_space(3))) int lds_i;
typedef __attribute__((address_space(3))) uint32_t lds_u;

struct UbPass {
    const float *in;            // [B][Lin + 4][in_ld] (channel offset applied), halo rows zero
    const float4 *img;          // two-piece f16 image + per-channel factors (models_api.hip conv_image_f16x3)
    const float *bias;
    const float *amax0, *amax1; // [B] largest |input| per sample (amax1: the other half of a concatenated input, or null)
    int in_ld, Lin, cin, ntaps, istride, ioff0, iostep, ostride, ooff, add;
    int Lpos;                   // positions per sample this pass computes (output row = l * ostride + ooff, l < Lpos)
};
struct UbArgs {
    UbPass pass[2];
    int n_pass, B, S, Lout, cout, groups;
    const float *first_x, *first_w, *first_b;   // pass 0 is the single-channel k5 convolution of x [B][Lout] on the VALU (no image)
    const float *gn_w, *gn_b;                   // GroupNorm + Mish after pass 0 (null: none)
    // a whole ConditionalResidualBlock1D in one launch: after pass 0's GroupNorm + Mish + FiLM the block's second convolution reads its
    // input t1 out of the stage (LDS -> slab, as the per-sample kernel does) and overwrites it; then gn1 + Mish, then the residual
    int fuse;
    UbPass c1;                                  // (in / amax unused: the stage and its own per-sample max)
    const float *gn1_w, *gn1_b;
    const float *film; int film_ld;             // [B][film_ld]: scale [cout] | shift [cout] of this block (null: none); row of sample b: film_idx[b]
    const int *film_idx;
    const float *res_id; int res_ld;            // identity residual [B][Lout + 4][res_ld], added last
    const float *res_x, *res_w, *res_b;         // single-channel residual: + res_w[co] x[b][l] + res_b[co]
    float *out; int out_ld;                     // [B][Lout + 4][out_ld] (channel offset applied)
    float *amax_out;                            // [B]
    const float *fin_w, *fin_b; float *eps;     // final 1x1 convolution -> eps [B][Lout] instead of `out`
    int clk_on;                                 // experiment build (DGDM_UB_CLOCKS): this launch stamps its phases
};

__device__ __forceinline__ int scale_exp_of(float mm) {      // input_scale_exp's exponent for a row set whose largest magnitude is mm
    const int e = (int)((__float_as_uint(mm) >> 23) & 0xffu);
    return (mm > 0.f && e < 255) ? min(max(12 + 127 - e, -100), 100) : 0;
}

// One convolution pass of the batched kernel: conv_mfma_f16x3's arithmetic over ns samples.  Waves 0-3 / 4-7 take the lower / upper half
// of the position tiles, wave & 3 picks MT of the cout / 16 output tiles.
#ifdef DGDM_UB_CLOCKS
__device__ int g_ub_clk_live;
#define UBCLK2() do { if (g_ub_clk_live && blockIdx.x == 0 && threadIdx.x == 0) { const int i_ = g_ub_clk_n; if (i_ < 64) { g_ub_clk[i_] = (long long)__builtin_readcyclecounter(); g_ub_clk_n = i_ + 1; } } } while (0)
#else
#define UBCLK2() do { } while (0)
#endif
template <int NT, int MT, bool SPLIT>
__device__ void ub_conv(const UbPass &a, const int S, const int ns, const int b0, const lds_i *kxs, lds_f *stage, const int CPo, const int LoutS /* output positions per sample */,
                        const int cout, lds_u4 *slab, const bool from_lds, const lds_f *lin /* from_lds: the input rows in LDS ([S][Lin + 4][lin_ld], the stage itself); else a.in.  (A flag,
                        not a null test: the stage sits at LDS offset 0, which compares equal to a null LDS pointer) */, const int lin_ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int groups = __builtin_amdgcn_readfirstlane(a.cin >> 5), ntaps = __builtin_amdgcn_readfirstlane(a.ntaps);
    const int Rs = a.Lin + 4, R = S * Rs;
    // Slab planes [channel group gg][piece][q] of Rp rows (16 B each), Rp a multiple of 16 rows (= all 64 banks): a B-operand read's lane
    // groups hold positions j of plane q and j' of plane q + 1 (q even) that are 16-byte neighbours modulo the plane stride - with planes
    // congruent modulo 256 B they cover the 64 banks exactly once (with R = 184 or 100 rows every read was a 2-way conflict: 12.9 % of the
    // launch's cycles, profiles/r05_pmc_kernels.md).  The fill's stores put a lane's eight 16-byte pieces into the eight (gg, q) planes:
    // one extra row of offset per (gg, q >> 1) spreads them over four bank quads (2-way, hidden under the store's own transfer time).
    const int Rp = (R + 15) & ~15, Gp = 8 * Rp + 2;
    // SPLIT: waves 0-3 / 4-7 take the lower / upper NT position tiles and wave & 3 picks MT output tiles (both halves fetch the same weight
    // fragments); otherwise every wave holds all NT position tiles of its own MT output tiles (no fragment is fetched twice)
    const int half = SPLIT ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0, tg = __builtin_amdgcn_readfirstlane(SPLIT ? (wave & 3) : wave);
    const int Lp = a.Lpos, npos = ns * Lp;
    constexpr int GS = 2;
    int brow[NT], srow[NT];
    float un[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = (half * NT + nt) * 16 + j, nn = min(n, npos - 1);
        const int sm = nn / Lp, l = nn - sm * Lp;
        brow[nt] = q * Rp + (q >> 1) + sm * Rs + l * a.istride + a.ioff0 + 2;
        srow[nt] = n < npos ? sm * (LoutS + 4) + 2 + l * a.ostride + a.ooff : -1;      // the stage keeps two halo rows around every sample, like the global buffers
        un[nt] = pow2_f(-kxs[sm]);
    }
    const int iters = __builtin_amdgcn_readfirstlane(ntaps * groups);
    const float *unw = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.img) + (size_t)4 * cout * a.cin * a.ntaps);
    int mt[MT];
    glb_f4 *w[MT];
    f32x4 nh[MT], nl[MT], n2h[MT], n2l[MT];
    f32x4 acc[MT][NT], tot[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        mt[m] = tg * MT + m;
        w[m] = (glb_f4 *)a.img + (size_t)mt[m] * iters * 128 + lane;
        nh[m] = w[m][0]; nl[m] = w[m][64];
        const size_t o1 = (size_t)min(1, iters - 1) * 128;
        n2h[m] = w[m][o1]; n2l[m] = w[m][o1 + 64];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { acc[m][nt] = (f32x4)(0.f); tot[m][nt] = (f32x4)(0.f); }
    }
    int it = 0;
    for (int g0 = 0; g0 < groups; g0 += GS) {
        __syncthreads();                                     // the previous pass' readers of the slab are done
        // split 64 channels of every row of the ns samples once: lane -> (row, 16-byte piece), eight lanes cover 256 contiguous bytes of a row.
        // The loads of FU items are issued together: one global round trip per FU items instead of one per item (the activations of the
        // previous launch come from the MALL, ~1.5 us away - item by item that latency, not the MFMAs, was the launch's time)
        constexpr int FU = 3;
        for (int i0 = threadIdx.x; i0 < R * GS * 4; i0 += blockDim.x * FU) {
            f32x4 lo[FU], hi[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int i = i0 + u * blockDim.x, row = i >> 3, r8 = i & 7, sm = row / Rs;
                if (i < R * GS * 4 && sm < ns) {
                    if (from_lds) {
                        const lds_f *src = lin + row * lin_ld + (g0 + (r8 >> 2)) * 32 + 4 * (r8 & 3);
                        lo[u] = *(const lds_f4 *)src; hi[u] = *(const lds_f4 *)(src + 16);
                    } else {
                        const glb_f *src = (glb_f *)a.in + ((size_t)(b0 + sm) * Rs + (row - sm * Rs)) * a.in_ld + (g0 + (r8 >> 2)) * 32 + 4 * (r8 & 3);
                        lo[u] = *(glb_f4 *)src; hi[u] = *(glb_f4 *)(src + 16);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int i = i0 + u * blockDim.x, row = i >> 3, r8 = i & 7, gg = r8 >> 2, qq = r8 & 3, sm = row / Rs;
                if (i >= R * GS * 4) break;
                u32x4_u H = {0u, 0u, 0u, 0u}, Lw = {0u, 0u, 0u, 0u};
                if (sm < ns) {
                    const float f = pow2_f(kxs[sm]);
                    const f32x2_u f2 = {f, f};
                    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
                    split2_f16(f32x2_u{lo[u][0], lo[u][1]} * f2, h0, l0);
                    split2_f16(f32x2_u{lo[u][2], lo[u][3]} * f2, h1, l1);
                    split2_f16(f32x2_u{hi[u][0], hi[u][1]} * f2, h2, l2);
                    split2_f16(f32x2_u{hi[u][2], hi[u][3]} * f2, h3, l3);
                    H = u32x4_u{h0, h1, h2, h3}; Lw = u32x4_u{l0, l1, l2, l3};
                }
                slab[gg * Gp + qq * Rp + (qq >> 1) + row] = H;
                slab[gg * Gp + 4 * Rp + qq * Rp + (qq >> 1) + row] = Lw;
            }
        }
        __syncthreads();
        UBCLK2();                                            // slab of this pass filled
#ifdef DGDM_UB_EXP_NOMFMA
        if (S > 0) { it += GS * ntaps; continue; }           // timing experiment (wrong results): everything but the MFMA loop
#endif
        for (int gt = 0; gt < GS * ntaps; ++gt, ++it) {
            const int gg = gt >= ntaps ? 1 : 0, t = gt - gg * ntaps;
            f16x8_u ah[MT], al[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[m] = __builtin_bit_cast(f16x8_u, nh[m]); al[m] = __builtin_bit_cast(f16x8_u, nl[m]);
                nh[m] = n2h[m]; nl[m] = n2l[m];
            }
            int pre = min(it + 2, iters - 1);
            asm volatile("" : "+v"(pre));
#pragma unroll
            for (int m = 0; m < MT; ++m) { n2h[m] = w[m][(size_t)pre * 128]; n2l[m] = w[m][(size_t)pre * 128 + 64]; }
            const int ro = gg * Gp + t * a.iostep;
            f16x8_u bh[NT], bl[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bh[nt] = __builtin_bit_cast(f16x8_u, (u32x4_u)slab[brow[nt] + ro]);
                bl[nt] = __builtin_bit_cast(f16x8_u, (u32x4_u)slab[4 * Rp + brow[nt] + ro]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[nt], acc[m][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[nt], acc[m][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[nt], acc[m][nt], 0, 0, 0);
            if (t == ntaps - 1) {                            // the chain is cut after every channel group
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { tot[m][nt] += acc[m][nt]; acc[m][nt] = (f32x4)(0.f); }
            }
        }
        UBCLK2();                                            // this pass' MFMAs issued
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float4 b4 = *reinterpret_cast<const float4 *>(a.bias + mt[m] * 16 + 4 * q);
        const float4 u4 = *reinterpret_cast<const float4 *>(unw + mt[m] * 16 + 4 * q);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (srow[nt] >= 0) {
                lds_f4 *p = (lds_f4 *)(stage + srow[nt] * CPo + 4 * q + mt[m] * 16);
                const f32x4 sum = tot[m][nt];
                const float u = un[nt];
                f32x4 v = {fmaf(sum[0], u * u4.x, b4.x), fmaf(sum[1], u * u4.y, b4.y), fmaf(sum[2], u * u4.z, b4.z), fmaf(sum[3], u * u4.w, b4.w)};
                if (a.add) v += *p;
                *p = v;
            }
        }
    }
}

__device__ void ub_conv_dispatch(const UbPass &a, int S, int ns, int b0, const lds_i *kxs, lds_f *stage, int CPo, int Lout, int cout, lds_u4 *slab,
                                 bool from_lds = false, const lds_f *lin = nullptr, int lin_ld = 0) {
    const int ntile = (S * a.Lpos + 15) >> 4, nth = (ntile + 1) >> 1;        // position tiles per half of the waves (the host keeps nth <= 6)
    // (the host keeps ntile <= 6 for the 256-wide convolutions - they all run at the half-length level - and <= 12 for the 128-wide ones)
    if ((cout >> 4) == 8) {
        if (nth <= 3) ub_conv<3, 2, true>(a, S, ns, b0, kxs, stage, CPo, Lout, cout, slab, from_lds, lin, lin_ld);
        else ub_conv<6, 2, true>(a, S, ns, b0, kxs, stage, CPo, Lout, cout, slab, from_lds, lin, lin_ld);
    } else {
        // 16 output tiles = two per wave with all (<= 6) position tiles: split in halves the 256-wide convolutions pulled 57 B/clk of weight
        // fragments through the CU's L1 (every fragment twice) - its limit is 64 - and ran at a third of their MFMA rate
        ub_conv<6, 2, false>(a, S, ns, b0, kxs, stage, CPo, Lout, cout, slab, from_lds, lin, lin_ld);
    }
}

__global__ __launch_bounds__(UB_THREADS) void ub_layer_kernel(const UbArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    lds_f *lds = (lds_f *)lds_raw;
    const int S = A.S, b0 = blockIdx.x * S, ns = min(S, A.B - b0);
    const int Lout = A.Lout, cout = A.cout, CPo = cout + UNET_ROW_PAD, Rso = Lout + 4;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, nwave = blockDim.x >> 6;
    lds_f *stage = lds;                                           // [S][Lout + 4][CPo]: two halo rows around every sample (zero when a fused conv reads them)
    lds_f *filmL = stage + S * Rso * CPo;                         // [S][2 cout]
    lds_i *kxs = (lds_i *)(filmL + S * 2 * cout);                 // [8] scale exponents, then [8] amax bits
    lds_u *amx = (lds_u *)(kxs + 8);
    lds_u4 *slab = (lds_u4 *)(kxs + 16);
    if (A.film)
        for (int i = t; i < ns * 2 * cout; i += blockDim.x) { const int sm = i / (2 * cout); filmL[i] = A.film[(size_t)A.film_idx[b0 + sm] * A.film_ld + (i - sm * 2 * cout)]; }
    if (t < 8) amx[t] = 0u;
#ifdef DGDM_UB_CLOCKS
    if (blockIdx.x == 0 && t == 0) { g_ub_clk_live = A.clk_on; if (A.clk_on) g_ub_clk_n = 0; }
#endif
    UBCLK(A);                                                      // [0] start
    if (A.fuse)
        for (int i = t; i < S * 4 * CPo; i += blockDim.x) { const int r = i / CPo, sm = r >> 2, h = r & 3; stage[(sm * Rso + (h < 2 ? h : Lout + h)) * CPo + (i - r * CPo)] = 0.f; }
    auto gn = [&](const float *gw, const float *gb, bool film, bool want_amax) __attribute__((always_inline)) {
        for (int idx = wave; idx < ns * A.groups; idx += nwave) {
            const int sm = idx / A.groups, g = idx - sm * A.groups;
            gn_group(stage + sm * Rso * CPo, CPo, cout, Lout, A.groups, g, gw, gb, film ? filmL + sm * 2 * cout : nullptr, amx + sm, want_amax);
        }
    };
    for (int p = 0; p < A.n_pass; ++p) {
        const UbPass &a = A.pass[p];
        if (p == 0 && A.first_x) {
            // block 0's first convolution: one input channel, on the VALU (unet_kernel's cin == 1 branch)
            for (int i = t; i < ns * Lout * cout; i += blockDim.x) {
                const int co = i % cout, r = i / cout, sm = r / Lout, l = r - sm * Lout;
                const float *x = A.first_x + (size_t)(b0 + sm) * Lout;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 5; ++k) { const int li = l + k - 2; acc = fmaf(A.first_w[k * cout + co], (li >= 0 && li < Lout) ? x[li] : 0.f, acc); }
                stage[(sm * Rso + 2 + l) * CPo + co] = acc + A.first_b[co];
            }
        } else {
            if (t < S) {
                float mm = 0.f;
                if (t < ns) { mm = a.amax0[b0 + t]; if (a.amax1) mm = fmaxf(mm, a.amax1[b0 + t]); }
                kxs[t] = scale_exp_of(mm);
            }
            __syncthreads();
            UBCLK(A);                                              // amax -> kxs
            ub_conv_dispatch(a, S, ns, b0, kxs, stage, CPo, Lout, cout, slab);
        }
        __syncthreads();
        UBCLK(A);                                                  // pass p done (epilogue in the stage)
        if (p == 0) {
#ifndef DGDM_UB_EXP_NOGN
            if (A.gn_w) { gn(A.gn_w, A.gn_b, A.film != nullptr, A.fuse != 0); __syncthreads(); }
#endif
            UBCLK(A);                                              // GroupNorm + Mish (+ FiLM)
            if (A.fuse) {
                // t1 is in the stage, its per-sample magnitude (what the unfused form's producer wrote to amax) came out of the GroupNorm pass
                if (t < S) { kxs[t] = scale_exp_of(t < ns ? __uint_as_float(amx[t]) : 0.f); }
                __syncthreads();
                if (t < 8) amx[t] = 0u;
                UBCLK(A);                                          // t1's magnitude
                ub_conv_dispatch(A.c1, S, ns, b0, kxs, stage, CPo, Lout, cout, slab, true, stage, CPo);
                __syncthreads();
                UBCLK(A);                                          // second convolution done
#ifndef DGDM_UB_EXP_NOGN
                gn(A.gn1_w, A.gn1_b, false, false);
                __syncthreads();
#endif
                UBCLK(A);                                          // GroupNorm + Mish of the block's output
            }
        }
    }
    if (A.eps) {                                                  // final_conv.1: Conv1d(d0, 1, 1)
        for (int r = t; r < ns * Lout; r += blockDim.x) {
            const int sm = r / Lout, l = r - sm * Lout;
            float acc = 0.f;
            for (int c = 0; c < cout; ++c) acc = fmaf(A.fin_w[c], stage[(sm * Rso + 2 + l) * CPo + c], acc);
            A.eps[(size_t)b0 * Lout + r] = acc + A.fin_b[0];
        }
        return;
    }
    // residual, store, and the per-sample magnitude the next launch scales its input by.  One flat loop over all samples' elements, the
    // residual loads of SU elements in flight together (a residual read is a round trip to the MALL)
    const int c4n = cout >> 2, per = Lout * c4n, total = ns * per;
    constexpr int SU = 4;
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = t; i0 < total; i0 += blockDim.x * SU) {
        f32x4 rv[SU];
        float xv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < total) {
                const int sm = i / per, r = i - sm * per, l = r / c4n, c = (r - l * c4n) * 4;
                if (A.res_id) rv[u] = *(glb_f4 *)((glb_f *)A.res_id + ((size_t)(b0 + sm) * Rso + l + 2) * A.res_ld + c);
                if (A.res_x) xv[u] = A.res_x[(size_t)(b0 + sm) * Lout + l];
            }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i >= total) break;
            const int sm = i / per, r = i - sm * per, l = r / c4n, c = (r - l * c4n) * 4;
            f32x4 v = *(lds_f4 *)(stage + (sm * Rso + 2 + l) * CPo + c);
            if (A.res_id) v += rv[u];
            if (A.res_x) {
                const float x = xv[u];
                const float4 rw = *reinterpret_cast<const float4 *>(A.res_w + c), rb = *reinterpret_cast<const float4 *>(A.res_b + c);
                v[0] += fmaf(rw.x, x, rb.x); v[1] += fmaf(rw.y, x, rb.y); v[2] += fmaf(rw.z, x, rb.z); v[3] += fmaf(rw.w, x, rb.w);
            }
            *reinterpret_cast<f32x4 *>(A.out + ((size_t)(b0 + sm) * Rso + l + 2) * A.out_ld + c) = v;
            const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            mx[0] = sm == 0 ? fmaxf(mx[0], m) : mx[0]; mx[1] = sm == 1 ? fmaxf(mx[1], m) : mx[1];
            mx[2] = sm == 2 ? fmaxf(mx[2], m) : mx[2]; mx[3] = sm == 3 ? fmaxf(mx[3], m) : mx[3];
        }
    }
#pragma unroll
    for (int sm = 0; sm < 4; ++sm) {
        float m = mx[sm];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0 && sm < ns) atomicMax((uint32_t *)(amx + sm), __float_as_uint(m));      // non-negative floats order like their bit patterns
    }
    __syncthreads();
    if (t < ns && A.amax_out) A.amax_out[b0 + t] = __uint_as_float(amx[t]);
    UBCLK(A);                                                      // residual + store
}

// step encoder + the eight FiLM vectors of every sample (unet_kernel's first two phases): film [B][8][2 cmax].  They depend on the sample's
// timestep only, and a denoise step hands every sample the same one: a sample whose timestep equals sample 0's takes sample 0's row
// (film_idx) instead of recomputing it (1024 workgroups of dependent L2 round trips -> one: 53 -> 6 us).
__global__ __launch_bounds__(256) void ub_cond_kernel(const UnetParams *__restrict__ pp, const int *__restrict__ timestep, float *__restrict__ film, int *__restrict__ film_idx) {
    __shared__ float cond_raw[32 * 5 + 8];
    const UnetParams &p = *pp;
    lds_f *cond = (lds_f *)cond_raw, *tmp = cond + p.dsed;
    const int b = blockIdx.x, t = threadIdx.x;
    const bool same = b > 0 && timestep[b] == timestep[0];
    if (t == 0) film_idx[b] = same ? 0 : b;
    if (same) return;
    const int half = p.dsed / 2;
    const float ts = (float)timestep[b];
    if (t < half) {
        const float a = ts * p.freqs[t];
        cond[t] = sinf(a);
        cond[half + t] = cosf(a);
    }
    __syncthreads();
    matvec(p.se1_wt, p.se1_b, cond, tmp, p.dsed, 4 * p.dsed);
    __syncthreads();
    for (int i = t; i < 4 * p.dsed; i += blockDim.x) tmp[i] = mish(tmp[i]);
    __syncthreads();
    matvec(p.se3_wt, p.se3_b, tmp, cond, 4 * p.dsed, p.dsed);
    __syncthreads();
    for (int i = t; i < p.dsed; i += blockDim.x) cond[i] = mish(cond[i]);
    __syncthreads();
    const int stride = 2 * p.cmax;
    for (int i = t; i < 8 * stride; i += blockDim.x) {
        const int blk = i / stride, n = i - blk * stride, N = 2 * p.res[blk].cout;
        if (n < N) film[(size_t)b * 8 * stride + i] = dot_kn(p.res[blk].cond_wt, N, n, cond, p.dsed) + p.res[blk].cond_b[n];
    }
}

static size_t ub_lds_bytes(int S, int Lin, int Lout, int cout) {
    const size_t Rp = ((size_t)S * (std::max(Lin, Lout) + 4) + 15) & ~(size_t)15;        // rows per slab plane (ub_conv)
    return ((size_t)S * (Lout + 4) * (cout + UNET_ROW_PAD) + (size_t)S * 2 * cout + 16) * 4 + (size_t)2 * (8 * Rp + 2) * 16;
}

// samples per workgroup: the most (<= 4) for which every launch fits the LDS and a half of the waves holds at most 6 position tiles
int unet_batched_samples(const UnetParams &p, int L) {
    const int L2 = (L - 1) / 2 + 1;
    if (p.dsed > 32 || p.d0 != 128 || p.d1 != 256) return 0;       // the tilings above are written for the shipped widths (down_dims [128, 256])
    if (2 * L2 != L) return 0;                                      // odd lengths: the per-sample kernel (selection, launch and effective_form agree)
    for (int S = 4; S >= 2; --S) {
        const bool fits = ub_lds_bytes(S, L, L, p.d0) <= 160 * 1024 && ub_lds_bytes(S, L2, L2, p.d1) <= 160 * 1024 && ub_lds_bytes(S, L2, L, p.d0) <= 160 * 1024 &&
                          ub_lds_bytes(S, L, L2, p.d0) <= 160 * 1024;
        if (fits && (S * L + 15) / 16 <= 12 && (S * L2 + 15) / 16 <= 6) return S;
    }
    return 0;
}

size_t unet_batched_ws_floats(const UnetParams &p, int B, int L) {
    const int L2 = (L - 1) / 2 + 1;
    return (size_t)B * ((size_t)3 * (L + 4) * p.d0 + (size_t)3 * (L2 + 4) * p.d0 + (size_t)2 * (L2 + 4) * p.d1 + (size_t)(L2 + 4) * 2 * p.d1 + (size_t)8 * 2 * p.cmax + 16);
}

// q: the host copy of the f16x3 parameter block (images + per-channel factors), q_dev the same in device memory; ws: zero-initialised
// workspace of unet_batched_ws_floats floats (the halo rows are never written)
int unet_launch_batched(const UnetParams &q, const UnetParams *q_dev, float *ws, int Bcap, const float *sample, const int *timestep, float *eps, int B, int L, hipStream_t s) {
    const int L2 = (L - 1) / 2 + 1, S = unet_batched_samples(q, L);
    DGDM_REQUIRE(S > 0 && 2 * L2 == L && B <= Bcap, DGDM_EINVAL, "unet_launch_batched: unsupported shape (L = %d, B = %d of %d)", L, B, Bcap);
    const int d0 = q.d0, d1 = q.d1;
    // the workspace is laid out for Bcap samples whatever this call's B: a sample's halo rows are the same addresses in every call
    const size_t n1 = (size_t)Bcap * (L + 4) * d0, n2 = (size_t)Bcap * (L2 + 4) * d0, n3 = (size_t)Bcap * (L2 + 4) * d1;
    float *P = ws, *Q = P + n1, *Rb = Q + n1, *D = Rb + n1, *T6 = D + n2, *O6 = T6 + n2, *T2 = O6 + n2, *O2 = T2 + n3, *CAT = O2 + n3;
    float *film = CAT + 2 * n3, *am = film + (size_t)Bcap * 8 * 2 * q.cmax;
    float *aP = am, *aQ = am + Bcap, *aR = am + 2 * Bcap, *aD = am + 3 * Bcap, *aT6 = am + 4 * Bcap, *aO6 = am + 5 * Bcap, *aT2 = am + 6 * Bcap, *aO2 = am + 7 * Bcap,
          *aC0 = am + 8 * Bcap, *aC1 = am + 9 * Bcap;
    static bool attr_set = false;
    if (!attr_set) {
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)ub_layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int *fidx = reinterpret_cast<int *>(am + (size_t)10 * Bcap);
    hipLaunchKernelGGL(ub_cond_kernel, dim3(B), dim3(256), 0, s, q_dev, timestep, film, fidx);
    const int fl = 8 * 2 * q.cmax;
    auto conv5 = [&](const float *in, int in_ld, int Lin, int cin, const float *img, const float *bias, const float *a0, const float *a1) {
        UbPass a{};
        a.Lpos = Lin;
        a.in = in; a.img = reinterpret_cast<const float4 *>(img); a.bias = bias; a.amax0 = a0; a.amax1 = a1; a.in_ld = in_ld; a.Lin = Lin; a.cin = cin; a.ntaps = 5;
        a.istride = 1; a.ioff0 = -2; a.iostep = 1; a.ostride = 1; a.ooff = 0; a.add = 0;
        return a;
    };
    int launch_no = 0;
    auto launch = [&](UbArgs &A, int Lin) -> int {
        A.B = B; A.S = S; A.groups = q.groups;
#ifdef DGDM_UB_CLOCKS
        A.clk_on = (launch_no == DGDM_UB_CLOCKS && B >= 512) ? 1 : 0;
#endif
        ++launch_no;
        const size_t lds = ub_lds_bytes(S, Lin, A.Lout, A.cout);
        DGDM_REQUIRE(lds <= 160 * 1024, DGDM_EINVAL, "unet_launch_batched: %zu B of LDS", lds);
        hipLaunchKernelGGL(ub_layer_kernel, dim3((B + S - 1) / S), dim3(UB_THREADS), lds, s, A);
        DGDM_HIP_CHECK(hipGetLastError());
#ifdef DGDM_UB_CLOCKS
        if (A.clk_on) {
            static int printed = 0;
            long long st[64]; int n = 0;
            hipStreamSynchronize(s);
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ub_clk), sizeof(st));
            hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ub_clk_n), sizeof(n));
            if (printed++ < 3) {
                fprintf(stderr, "ub stamps launch %d (Lout %d cout %d passes %d fuse %d): ", launch_no - 1, A.Lout, A.cout, A.n_pass, A.fuse);
                for (int i = 1; i < n && i < 64; ++i) fprintf(stderr, "%lld ", st[i] - st[i - 1]);
                fprintf(stderr, "| total %lld\n", n > 0 ? st[n - 1] - st[0] : 0LL);
            }
        }
#endif
        return DGDM_OK;
    };
    int rc;
    // one ConditionalResidualBlock1D (unet_kernel's res_block): conv0 + GN + Mish + FiLM -> t1; conv1 + GN + Mish + residual -> out
    auto block = [&](int bi, const float *in, int in_ld, const float *ain0, const float *ain1, int Lb, float *t1, float *at1, float *out, int out_ld, float *aout) -> int {
        (void)t1; (void)at1;                              // (the unfused form's intermediate: t1 now stays in the stage)
        const UnetRes &w = q.res[bi];
        UbArgs A{};
        A.n_pass = 1; A.Lout = Lb; A.cout = w.cout; A.gn_w = w.g0_w; A.gn_b = w.g0_b; A.film = film + (size_t)bi * 2 * q.cmax; A.film_ld = fl; A.film_idx = fidx;
        A.out = out; A.out_ld = out_ld; A.amax_out = aout;
        if (w.cin == 1) { A.first_x = sample; A.first_w = w.c0_w; A.first_b = w.c0_b; }
        else A.pass[0] = conv5(in, in_ld, Lb, w.cin, w.c0_w, w.c0_b, ain0, ain1);
        A.fuse = 1; A.gn1_w = w.g1_w; A.gn1_b = w.g1_b;
        A.c1 = conv5(nullptr, w.cout + UNET_ROW_PAD, Lb, w.cout, w.c1_w, w.c1_b, nullptr, nullptr);
        if (w.cin == 1) { A.res_x = sample; A.res_w = w.res_w; A.res_b = w.res_b; }
        else if (w.res_w) {
            A.n_pass = 2;
            A.pass[1] = conv5(in, in_ld, Lb, w.cin, w.res_w, w.res_b, ain0, ain1);
            A.pass[1].ntaps = 1; A.pass[1].ioff0 = 0; A.pass[1].add = 1;
        } else { A.res_id = in; A.res_ld = in_ld; }
        return launch(A, Lb);
    };
    if ((rc = block(0, nullptr, 0, nullptr, nullptr, L, P, aP, Q, d0, aQ))) return rc;                       // down0.0   1 -> d0
    if ((rc = block(1, Q, d0, aQ, nullptr, L, P, aP, Rb, d0, aR))) return rc;                                // down0.1
    {   // Downsample1d: Conv1d(d0, d0, 3, 2, 1)
        UbArgs A{};
        A.n_pass = 1; A.Lout = L2; A.cout = d0; A.out = D; A.out_ld = d0; A.amax_out = aD;
        A.pass[0] = conv5(Rb, d0, L, d0, q.down_w, q.down_b, aR, nullptr);
        A.pass[0].ntaps = 3; A.pass[0].istride = 2; A.pass[0].ioff0 = -1; A.pass[0].Lpos = L2;
        if ((rc = launch(A, L))) return rc;
    }
    if ((rc = block(2, D, d0, aD, nullptr, L2, T2, aT2, O2, d1, aO2))) return rc;                            // down1.0   d0 -> d1
    if ((rc = block(3, O2, d1, aO2, nullptr, L2, T2, aT2, CAT + d1, 2 * d1, aC1))) return rc;                // down1.1 -> the skip = channels [d1, 2 d1) of the concat buffer
    if ((rc = block(4, CAT + d1, 2 * d1, aC1, nullptr, L2, T2, aT2, O2, d1, aO2))) return rc;                // mid0
    if ((rc = block(5, O2, d1, aO2, nullptr, L2, T2, aT2, CAT, 2 * d1, aC0))) return rc;                     // mid1 -> channels [0, d1)
    if ((rc = block(6, CAT, 2 * d1, aC0, aC1, L2, T6, aT6, O6, d0, aO6))) return rc;                         // up0.0   2 d1 -> d0
    if ((rc = block(7, O6, d0, aO6, nullptr, L2, T6, aT6, D, d0, aD))) return rc;                            // up0.1
    {   // Upsample1d: ConvTranspose1d(d0, d0, 4, 2, 1): out[2 li] = W1 in[li] + W3 in[li - 1]; out[2 li + 1] = W2 in[li] + W0 in[li + 1]
        UbArgs A{};
        A.n_pass = 2; A.Lout = L; A.cout = d0; A.out = P; A.out_ld = d0; A.amax_out = aP;
        A.pass[0] = conv5(D, d0, L2, d0, q.up_w_even, q.up_b, aD, nullptr);
        A.pass[0].ntaps = 2; A.pass[0].ioff0 = 0; A.pass[0].iostep = -1; A.pass[0].ostride = 2; A.pass[0].ooff = 0;
        A.pass[1] = conv5(D, d0, L2, d0, q.up_w_odd, q.up_b, aD, nullptr);
        A.pass[1].ntaps = 2; A.pass[1].ioff0 = 0; A.pass[1].iostep = 1; A.pass[1].ostride = 2; A.pass[1].ooff = 1;
        if ((rc = launch(A, L2))) return rc;
    }
    {   // final_conv: Conv1dBlock(d0, d0, 5) then Conv1d(d0, 1, 1)
        UbArgs A{};
        A.n_pass = 1; A.Lout = L; A.cout = d0; A.gn_w = q.fin_gw; A.gn_b = q.fin_gb; A.fin_w = q.out_w; A.fin_b = q.out_b; A.eps = eps;
        A.pass[0] = conv5(P, d0, L, d0, q.fin_w, q.fin_b, aP, nullptr);
        if ((rc = launch(A, L))) return rc;
    }
    return DGDM_OK;
}

static size_t unet_lds_floats(const UnetParams &p, int L) {
    const int L2 = (L - 1) / 2 + 1;
    const int bufS = std::max((L + 4) * (p.d0 + UNET_ROW_PAD), (L2 + 4) * (p.d1 + UNET_ROW_PAD));
    const int bufA = std::max((L + 4) * (p.d0 + UNET_ROW_PAD), (L2 + 4) * (2 * p.d1 + UNET_ROW_PAD));
    return (size_t)bufA + 3 * (size_t)bufS + 8 * 2 * p.cmax + p.dsed + 4 * p.dsed + (L + 4 + 3) + 16;
}
static size_t unet_slab_floats(int L) { return 16 + (size_t)UNET_SLAB_GROUPS * (L + 4) * 32; }

bool unet_f16x3_fits(const UnetParams &p, int L) { return (unet_lds_floats(p, L) + unet_slab_floats(L)) * 4 <= 160 * 1024; }

int unet_launch(const UnetParams &p, const UnetParams *p_dev, bool f16x3, const float *sample, const int *timestep, float *eps, int B, int L, hipStream_t s) {
    if (B <= 0) return DGDM_OK;
    const int L2 = (L - 1) / 2 + 1;
    DGDM_REQUIRE(2 * L2 == L, DGDM_EINVAL, "U-Net needs an even number of control points (got %d): the skip concat of the reference "
                 "requires ConvTranspose1d(4,2,1) to restore L", L);
    DGDM_REQUIRE(L <= 64, DGDM_EINVAL, "U-Net kernel supports up to 64 control points (got %d)", L);
    const int bufS = std::max((L + 4) * (p.d0 + UNET_ROW_PAD), (L2 + 4) * (p.d1 + UNET_ROW_PAD));
    const int bufA = std::max((L + 4) * (p.d0 + UNET_ROW_PAD), (L2 + 4) * (2 * p.d1 + UNET_ROW_PAD));
    // (+ the f16x3 convolutions' scratch: 16 floats + the split slabs, UNET_SLAB_GROUPS x (L + 4) rows x 128 B; the caller checked that they fit)
    const size_t lds_floats = unet_lds_floats(p, L) + (f16x3 ? unet_slab_floats(L) : 0);
    DGDM_REQUIRE(lds_floats * 4 <= 160 * 1024, DGDM_EINVAL, "U-Net activations (%zu B) exceed the 160 KiB LDS", lds_floats * 4);
    static bool attr_set = false;
    if (!attr_set) {
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)unet_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(unet_kernel, dim3(B), dim3(UNET_THREADS), lds_floats * 4, s, p_dev, bufA, bufS, sample, timestep, eps, L, UNET_SLAB_GROUPS);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

}  // namespace dgdm

#ifdef DGDM_UNET_CLOCKS
extern "C" int dgdm_debug_unet_clocks(long long *out_host, int n) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(dgdm::unet_clk), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -3;
}
#endif
