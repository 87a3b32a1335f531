// ConditionalUnet1D forward (generator/diffusion_utils.py:123-285), one workgroup per sample.
//
// The whole network runs inside one launch: activations live in LDS ([channel][L+4] with a
// zero halo of 2, <= 128 KiB for the shipped shapes), weights stream from L2, GroupNorm
// statistics are wave-shuffle reductions, Mish/FiLM/residual adds are fused into the passes
// that already touch the data.  One launch per denoise step instead of ~90 eager ops.
//
// Round-1 arithmetic: the k=5 convolutions run on the f32 VALU (same peak rate as f32 MFMA on
// gfx950, but lower achieved efficiency); the net is < 2 % of a guided step.
#include "common.h"
#include "unet.h"
#include <algorithm>

namespace dgdm {

__device__ __forceinline__ float mish(float x) {
    // torch.nn.functional.mish = x * tanh(softplus(x)), softplus threshold 20
    const float sp = x > 20.f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// out[co][l] = bias[co] + sum_{ci,k} W[(ci*KW+k)*Cout+co] * in[ci][l*STRIDE + k - PAD],  l < Lout
// in/out: LDS, row strides LPi/LPo, data starts at column 2.  Each thread item = 2 channels x 7 positions.
template <int KW, int STRIDE, int PAD>
__device__ void conv1d(const float *__restrict__ W, const float *__restrict__ bias, const float *in, int LPi, float *out, int LPo,
                       int Cin, int Cout, int Lout) {
    constexpr int NP = 7;
    constexpr int WIN = (NP - 1) * STRIDE + KW;
    const int nchunk = (Lout + NP - 1) / NP;
    const int half = (Cout + 1) / 2;
    for (int item = threadIdx.x; item < half * nchunk; item += blockDim.x) {
        const int chunk = item / half, co0 = item - chunk * half, co1 = co0 + half;
        const bool two = co1 < Cout;
        const int l0 = chunk * NP;
        float a0[NP], a1[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
        const float *ip = in + 2 + l0 * STRIDE - PAD;
        for (int ci = 0; ci < Cin; ++ci) {
            float x[WIN];
#pragma unroll
            for (int j = 0; j < WIN; ++j) x[j] = ip[ci * LPi + j];
            const float *w = W + (size_t)ci * KW * Cout;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const float w0 = w[k * Cout + co0];
                const float w1 = two ? w[k * Cout + co1] : 0.f;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    a0[i] = fmaf(w0, x[i * STRIDE + k], a0[i]);
                    a1[i] = fmaf(w1, x[i * STRIDE + k], a1[i]);
                }
            }
        }
        const float b0 = bias[co0], b1 = two ? bias[co1] : 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (l0 + i < Lout) {
                out[co0 * LPo + 2 + l0 + i] = a0[i] + b0;
                if (two) out[co1 * LPo + 2 + l0 + i] = a1[i] + b1;
            }
        }
    }
}

// ConvTranspose1d(C, C, 4, stride 2, pad 1) (diffusion_utils.py:51): out[co][lo] = b + sum_{ci,k} in[ci][li] W[ci][k][co],
// lo = 2 li - 1 + k.  W stored [(ci*4+k)*Cout+co].
__device__ void conv_transpose4(const float *__restrict__ W, const float *__restrict__ bias, const float *in, int LPi, float *out, int LPo,
                                int Cin, int Cout, int Lin) {
    const int Lout = 2 * Lin;
    for (int item = threadIdx.x; item < Cout * Lin; item += blockDim.x) {
        const int li = item / Cout, co = item - li * Cout;
        // outputs lo = 2 li (k=1 from li, k=3 from li-1) and lo = 2 li + 1 (k=2 from li, k=0 from li+1)
        float e = 0.f, o = 0.f;
        for (int ci = 0; ci < Cin; ++ci) {
            const float *w = W + (size_t)ci * 4 * Cout + co;
            const float xm = in[ci * LPi + 2 + li - 1], x0 = in[ci * LPi + 2 + li], xp = in[ci * LPi + 2 + li + 1];
            e = fmaf(x0, w[1 * Cout], e);
            e = fmaf(xm, w[3 * Cout], e);
            o = fmaf(x0, w[2 * Cout], o);
            o = fmaf(xp, w[0 * Cout], o);
        }
        out[co * LPo + 2 + 2 * li] = e + bias[co];
        out[co * LPo + 2 + 2 * li + 1] = o + bias[co];
    }
    (void)Lout;
}

__device__ void zero_halo(float *buf, int LP, int C, int L) {
    for (int i = threadIdx.x; i < C * 4; i += blockDim.x) {
        const int c = i >> 2, j = i & 3;
        buf[c * LP + (j < 2 ? j : L + j)] = 0.f;
    }
}

// GroupNorm(groups, C) -> Mish -> optional FiLM (scale*y + shift), in place.  (diffusion_utils.py:65-69,113-116)
__device__ void gn_mish_film(float *buf, int LP, int C, int L, int groups, const float *__restrict__ gamma, const float *__restrict__ beta,
                             const float *film /*LDS [2C] or null*/) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
    const int cg = C / groups, cnt = cg * L;
    for (int g = wave; g < groups; g += nwave) {
        float s = 0.f;
        for (int i = lane; i < cnt; i += 64) { const int c = g * cg + i / L, l = i % L; s += buf[c * LP + 2 + l]; }
        const float mean = wave_sum(s) / (float)cnt;
        float q = 0.f;
        for (int i = lane; i < cnt; i += 64) { const int c = g * cg + i / L, l = i % L; const float d = buf[c * LP + 2 + l] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.f / sqrtf(wave_sum(q) / (float)cnt + 1e-5f);
        for (int i = lane; i < cnt; i += 64) {
            const int c = g * cg + i / L, l = i % L;
            float y = (buf[c * LP + 2 + l] - mean) * rstd * gamma[c] + beta[c];
            y = mish(y);
            if (film) y = film[c] * y + film[C + c];
            buf[c * LP + 2 + l] = y;
        }
    }
}

// y[n] = b[n] + sum_k WT[k][n] x[k],  x in LDS
__device__ void matvec(const float *__restrict__ WT, const float *__restrict__ b, const float *x, float *y, int K, int N) {
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(WT[(size_t)k * N + n], x[k], acc);
        y[n] = acc + b[n];
    }
}

struct Bufs { float *A, *B, *C, *D, *film, *cond, *tmp; };

// ConditionalResidualBlock1D.forward (diffusion_utils.py:101-120): x(in) -> out; t1 scratch.
__device__ void res_block(const UnetRes &w, const float *in, float *t1, float *out, int LP, int L, int cond_dim, int groups, const Bufs &s) {
    matvec(w.cond_wt, w.cond_b, s.cond, s.film, cond_dim, 2 * w.cout);     // cond_encoder: Mish already applied to s.cond
    conv1d<5, 1, 2>(w.c0_w, w.c0_b, in, LP, t1, LP, w.cin, w.cout, L);
    zero_halo(t1, LP, w.cout, L);
    __syncthreads();
    gn_mish_film(t1, LP, w.cout, L, groups, w.g0_w, w.g0_b, s.film);
    __syncthreads();
    conv1d<5, 1, 2>(w.c1_w, w.c1_b, t1, LP, out, LP, w.cout, w.cout, L);
    zero_halo(out, LP, w.cout, L);
    __syncthreads();
    gn_mish_film(out, LP, w.cout, L, groups, w.g1_w, w.g1_b, nullptr);
    __syncthreads();
    if (w.res_w) {   // residual 1x1 conv into t1, then add
        conv1d<1, 1, 0>(w.res_w, w.res_b, in, LP, t1, LP, w.cin, w.cout, L);
        __syncthreads();
        for (int i = threadIdx.x; i < w.cout * L; i += blockDim.x) { const int c = i / L, l = i % L; out[c * LP + 2 + l] += t1[c * LP + 2 + l]; }
    } else {
        for (int i = threadIdx.x; i < w.cout * L; i += blockDim.x) { const int c = i / L, l = i % L; out[c * LP + 2 + l] += in[c * LP + 2 + l]; }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void unet_kernel(const UnetParams p, const float *__restrict__ sample, const int *__restrict__ timestep,
                                                   float *__restrict__ eps, int L) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, t = threadIdx.x;
    const int L2 = (L - 1) / 2 + 1;                 // Conv1d(k3, s2, p1)
    const int LP = L + 4, LP2 = L2 + 4;
    Bufs s;
    s.A = lds;
    s.B = s.A + p.bufA;
    s.C = s.B + p.bufS;
    s.D = s.C + p.bufS;
    s.film = s.D + p.bufS;
    s.cond = s.film + 2 * p.cmax;
    s.tmp = s.cond + p.dsed;
    const int G = p.groups;

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
    // ---- input (B,L,1) -> [1][LP]
    for (int i = t; i < LP; i += blockDim.x) s.A[i] = (i >= 2 && i < 2 + L) ? sample[(size_t)b * L + i - 2] : 0.f;
    __syncthreads();

    res_block(p.res[0], s.A, s.B, s.C, LP, L, p.dsed, G, s);        // down0.0   1 -> d0
    res_block(p.res[1], s.C, s.B, s.D, LP, L, p.dsed, G, s);        // down0.1   d0 -> d0   (its skip is never consumed, :264-278)
    conv1d<3, 2, 1>(p.down_w, p.down_b, s.D, LP, s.B, LP2, p.d0, p.d0, L2);   // Downsample1d (:42)
    zero_halo(s.B, LP2, p.d0, L2);
    __syncthreads();
    res_block(p.res[2], s.B, s.C, s.D, LP2, L2, p.dsed, G, s);      // down1.0   d0 -> d1
    res_block(p.res[3], s.D, s.B, s.C, LP2, L2, p.dsed, G, s);      // down1.1   -> skip (kept in C)
    res_block(p.res[4], s.C, s.B, s.D, LP2, L2, p.dsed, G, s);      // mid0
    res_block(p.res[5], s.D, s.B, s.A, LP2, L2, p.dsed, G, s);      // mid1 -> A[0:d1]
    for (int i = t; i < p.d1 * LP2; i += blockDim.x) s.A[p.d1 * LP2 + i] = s.C[i];   // torch.cat((x, h.pop()), dim=1) (:275)
    __syncthreads();
    res_block(p.res[6], s.A, s.B, s.D, LP2, L2, p.dsed, G, s);      // up0.0   2*d1 -> d0
    res_block(p.res[7], s.D, s.B, s.C, LP2, L2, p.dsed, G, s);      // up0.1
    conv_transpose4(p.up_w, p.up_b, s.C, LP2, s.A, LP, p.d0, p.d0, L2);      // Upsample1d (:51); 2*L2 == L for even L
    zero_halo(s.A, LP, p.d0, L);
    __syncthreads();
    conv1d<5, 1, 2>(p.fin_w, p.fin_b, s.A, LP, s.B, LP, p.d0, p.d0, L);      // final_conv.0
    __syncthreads();
    gn_mish_film(s.B, LP, p.d0, L, G, p.fin_gw, p.fin_gb, nullptr);
    __syncthreads();
    for (int l = t; l < L; l += blockDim.x) {                        // final_conv.1: Conv1d(d0, 1, 1)
        float acc = 0.f;
        for (int c = 0; c < p.d0; ++c) acc = fmaf(p.out_w[c], s.B[c * LP + 2 + l], acc);
        eps[(size_t)b * L + l] = acc + p.out_b[0];
    }
}

int unet_launch(const UnetParams &p_in, const float *sample, const int *timestep, float *eps, int B, int L, hipStream_t s) {
    if (B <= 0) return DGDM_OK;
    UnetParams p = p_in;
    const int L2 = (L - 1) / 2 + 1;
    const int slack = 32;                                    // conv windows of a ragged last chunk read past the row end
    p.bufS = std::max(p.d0 * (L + 4), p.d1 * (L2 + 4)) + slack;
    p.bufA = std::max(p.d0 * (L + 4), 2 * p.d1 * (L2 + 4)) + slack;
    DGDM_REQUIRE(2 * L2 == L, DGDM_EINVAL, "U-Net needs an even number of control points (got %d): the skip concat of the reference "
                 "requires ConvTranspose1d(4,2,1) to restore L", L);
    const size_t lds_floats = (size_t)p.bufA + 3 * (size_t)p.bufS + 2 * p.cmax + p.dsed + 4 * p.dsed;
    DGDM_REQUIRE(lds_floats * 4 <= 160 * 1024, DGDM_EINVAL, "U-Net activations (%zu B) exceed the 160 KiB LDS", lds_floats * 4);
    static bool attr_set = false;
    if (!attr_set) {
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)unet_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(unet_kernel, dim3(B), dim3(256), lds_floats * 4, s, p, sample, timestep, eps, L);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

}  // namespace dgdm
