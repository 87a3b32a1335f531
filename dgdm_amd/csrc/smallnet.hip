// Small per-finger / per-cell pieces around the trunk: encoders, embeddings, first-layer tables,
// the fold of the tile partials and the backward pass through the gripper encoder.
// All of these touch O(B) or O(cells) rows (KBs..MBs) against the trunk's O(B*cells) rows.
#include "common.h"
#include "smallnet.h"

// No implicit a*b+c -> fma contraction in this file: the scheduler update, FPS and ball-query distances must round
// like the reference's separate float32 ops (HIP's __fmul_rn/__fadd_rn are plain * and + and would be contracted).
// Explicit fmaf() calls are unaffected.
#pragma clang fp contract(off)

namespace dgdm {

// ------------------------------------------------------------------------------------------------
// Y[r][n] = act( sum_k X[r][k] * WT[k][n] + bias[n] + rowbias[r / rb_div][n] ) (+ Y[r][n] if accumulate)
// WT is the weight stored [in][out] ("kn"), so a wave reads 64 consecutive outputs per k.
// The k loop is an in-order fmaf chain per output, like a sequential dot product.
// RB rows per workgroup (8 for a handful of rows, 32 otherwise: every weight value loaded from L2 feeds RB fmas); the k loop
// runs four at a time - four weight loads in flight, the activations as one ds_read_b128 per row - with the same ascending-k
// fmaf order per output as a one-at-a-time loop, so the result does not depend on RB or the unrolling.
template <int ACT, int RB>
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ WT,
                                                     const float *__restrict__ bias, const float *__restrict__ rowbias,
                                                     int rb_div, float *__restrict__ Y, int ldy, int rows, int K, int N,
                                                     int accumulate) {
    constexpr int KC = 128;
    __shared__ __attribute__((aligned(16))) float xs[RB][KC];
    const int r0 = blockIdx.x * RB;
    const int n = blockIdx.y * 256 + threadIdx.x;
    const int nn = min(n, N - 1);                       // out-of-range columns compute a copy of the last one and drop it
    float acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int kc = min(KC, K - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < RB * KC; i += 256) {
            const int r = i / KC, k = i - r * KC;
            xs[r][k] = (r0 + r < rows && k < kc) ? X[(size_t)(r0 + r) * ldx + k0 + k] : 0.f;
        }
        __syncthreads();
        const float *wp = WT + (size_t)k0 * N + nn;
        for (int k = 0; k < kc; k += 4) {
            // k + j >= kc only in the last group of a K that is not a multiple of 4: x is zero there, w is read from a valid row
            const float w0 = wp[(size_t)k * N], w1 = wp[(size_t)min(k + 1, kc - 1) * N], w2 = wp[(size_t)min(k + 2, kc - 1) * N],
                        w3 = wp[(size_t)min(k + 3, kc - 1) * N];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float4 x = *reinterpret_cast<const float4 *>(&xs[r][k]);
                acc[r] = fmaf(x.w, w3, fmaf(x.z, w2, fmaf(x.y, w1, fmaf(x.x, w0, acc[r]))));
            }
        }
    }
    if (n >= N) return;
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        if (r0 + r >= rows) break;
        float v = acc[r] + bn;
        if (rowbias) v += rowbias[(size_t)((r0 + r) / rb_div) * N + n];
        if (accumulate) v += Y[(size_t)(r0 + r) * ldy + n];
        if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
        if (ACT == ACT_SILU) v = v / (1.f + expf(-v));
        Y[(size_t)(r0 + r) * ldy + n] = v;
    }
}

template <int RB>
static void linear_launch(const float *X, int ldx, const float *WT, const float *bias, const float *rowbias, int rb_div, float *Y, int ldy,
                          int rows, int K, int N, int act, bool accumulate, hipStream_t s) {
    dim3 grid((rows + RB - 1) / RB, (N + 255) / 256);
    if (act == ACT_NONE)
        hipLaunchKernelGGL((linear_kernel<ACT_NONE, RB>), grid, dim3(256), 0, s, X, ldx, WT, bias, rowbias, rb_div, Y, ldy, rows, K, N, (int)accumulate);
    else if (act == ACT_RELU)
        hipLaunchKernelGGL((linear_kernel<ACT_RELU, RB>), grid, dim3(256), 0, s, X, ldx, WT, bias, rowbias, rb_div, Y, ldy, rows, K, N, (int)accumulate);
    else
        hipLaunchKernelGGL((linear_kernel<ACT_SILU, RB>), grid, dim3(256), 0, s, X, ldx, WT, bias, rowbias, rb_div, Y, ldy, rows, K, N, (int)accumulate);
}

int linear(const float *X, int ldx, const float *WT, const float *bias, const float *rowbias, int rb_div, float *Y, int ldy,
           int rows, int K, int N, int act, bool accumulate, hipStream_t s) {
    if (rows <= 0) return DGDM_OK;
    if (rows >= 64) linear_launch<32>(X, ldx, WT, bias, rowbias, rb_div, Y, ldy, rows, K, N, act, accumulate, s);
    else linear_launch<8>(X, ldx, WT, bias, rowbias, rb_div, Y, ldy, rows, K, N, act, accumulate, s);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// ------------------------------------------------------------------------------------------------
// get_embedder(d, 4): [x, sin(2^k x), cos(2^k x)]_{k=0..3}  (dynamics/profile_forward_2d.py:10-56)
// pose = cat(embed(ori) [9], embed(pos) [18])  (profile_forward_2d.py:149-151) -> out[rows][27]
__global__ void pose_embed_kernel(const float *__restrict__ ori, const float *__restrict__ pos, float *__restrict__ out, int rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float o = ori[r], px = pos[2 * r], py = pos[2 * r + 1];
    float *e = out + (size_t)r * 27;
    e[0] = o;
    e[9] = px; e[10] = py;
    float f = 1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e[1 + 2 * k] = sinf(o * f);
        e[2 + 2 * k] = cosf(o * f);
        e[11 + 4 * k] = sinf(px * f); e[12 + 4 * k] = sinf(py * f);
        e[13 + 4 * k] = cosf(px * f); e[14 + 4 * k] = cosf(py * f);
        f *= 2.f;
    }
}

__global__ void tile_table_kernel(const float *__restrict__ T, int rows, int W, float4 *__restrict__ out, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 of the output
    if (e >= total) return;
    const int lane = (int)(e & 63), n = lane & 31, h = lane >> 5;
    const int64_t t = e >> 6;
    const int q = (int)(t & 3);
    const int wb = W / 32;
    const int o = (int)((t >> 2) % wb);
    const int64_t g = (t >> 2) / wb;
    const int64_t r = min((int64_t)32 * g + n, (int64_t)rows - 1);
    out[e] = *reinterpret_cast<const float4 *>(T + r * W + 32 * o + 8 * q + 4 * h);
}

int tile_table(const float *T, int rows, int W, float *out, hipStream_t s) {
    const int64_t total = (int64_t)((rows + 31) / 32) * (W / 32) * 4 * 64;
    hipLaunchKernelGGL(tile_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, T, rows, W, reinterpret_cast<float4 *>(out), total);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pose_embed(const float *ori, const float *pos, float *out, int rows, hipStream_t s) {
    if (rows <= 0) return DGDM_OK;
    hipLaunchKernelGGL(pose_embed_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, ori, pos, out, rows);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// timestep_embedding(t, dim): [cos(t f_i) | sin(t f_i)], f_i from the host (profile_forward_2d.py:58-76)
__global__ void time_embed_kernel(const float *__restrict__ t, float t_scalar, const float *__restrict__ freqs, float *__restrict__ out,
                                  int rows, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * half) return;
    const int r = i / half, j = i - r * half;
    const float a = (t ? t[r] : t_scalar) * freqs[j];
    out[(size_t)r * 2 * half + j] = cosf(a);
    out[(size_t)r * 2 * half + half + j] = sinf(a);
}

int time_embed(const float *t_dev, float t_scalar, const float *freqs, float *out, int rows, int half, hipStream_t s) {
    if (rows <= 0) return DGDM_OK;
    hipLaunchKernelGGL(time_embed_kernel, dim3((rows * half + 255) / 256), dim3(256), 0, s, t_dev, t_scalar, freqs, out, rows, half);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// out[g][n] = (a ? a[idx ? idx[g] : g][n] : 0) + (b ? b[n] : 0)
__global__ void gather_add_kernel(const float *__restrict__ a, const int *__restrict__ idx, const float *__restrict__ b,
                                  float *__restrict__ out, int groups, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * N) return;
    const int g = i / N, n = i - g * N;
    float v = b ? b[n] : 0.f;
    if (a) v += a[(size_t)(idx ? idx[g] : g) * N + n];
    out[i] = v;
}

int gather_add(const float *a, const int *idx, const float *b, float *out, int groups, int N, hipStream_t s) {
    if (groups <= 0) return DGDM_OK;
    hipLaunchKernelGGL(gather_add_kernel, dim3((groups * N + 255) / 256), dim3(256), 0, s, a, idx, b, out, groups, N);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward tail of cond_fn for one finger of one chain (what autograd does after the trunk,
// generator/diffusion.py:498,504): fold the tile partials over cells, then go back through
// W1'[:, ctrl part], gripper_encoder.2, ReLU, gripper_encoder.0 (profile_forward_2d.py:103-107,148).
template <int W1>
__global__ __launch_bounds__(256) void dyn_post_kernel(const float *__restrict__ partial, int tiles_per_b,
                                                       const float *__restrict__ w1c /*[W1][256]*/,
                                                       const float *__restrict__ g2w /*[256][256]*/,
                                                       const float *__restrict__ g0w /*[256][L]*/,
                                                       const float *__restrict__ V /*[rows][256] relu(g0 x + b)*/,
                                                       float *__restrict__ grad /*[rows][L]*/, int L) {
    __shared__ float da[W1];
    __shared__ float v1[256];
    __shared__ float v2[256];
    const int row = blockIdx.x, t = threadIdx.x;
    for (int f = t; f < W1; f += 256) {
        const float *src = partial + (size_t)row * tiles_per_b * W1 + f;
        float acc = 0.f;
        for (int ct = 0; ct < tiles_per_b; ++ct) acc += src[(size_t)ct * W1];
        da[f] = acc;
    }
    __syncthreads();
    float acc = 0.f;
    for (int f = 0; f < W1; ++f) acc = fmaf(w1c[(size_t)f * 256 + t], da[f], acc);
    v1[t] = acc;
    __syncthreads();
    acc = 0.f;
    for (int f = 0; f < 256; ++f) acc = fmaf(g2w[(size_t)f * 256 + t], v1[f], acc);
    v2[t] = V[(size_t)row * 256 + t] > 0.f ? acc : 0.f;
    __syncthreads();
    if (t < L) {
        acc = 0.f;
        for (int f = 0; f < 256; ++f) acc = fmaf(g0w[(size_t)f * L + t], v2[f], acc);
        grad[(size_t)row * L + t] = acc;
    }
}

int dyn_post(int W1, const float *partial, int tiles_per_b, const float *w1c, const float *g2w, const float *g0w, const float *V,
             float *grad, int rows, int L, hipStream_t s) {
    if (rows <= 0) return DGDM_OK;
    DGDM_REQUIRE(L <= 256, DGDM_EINVAL, "params_ch %d > 256 not supported", L);
    if (W1 == 256)
        hipLaunchKernelGGL(dyn_post_kernel<256>, dim3(rows), dim3(256), 0, s, partial, tiles_per_b, w1c, g2w, g0w, V, grad, L);
    else
        hipLaunchKernelGGL(dyn_post_kernel<512>, dim3(rows), dim3(256), 0, s, partial, tiles_per_b, w1c, g2w, g0w, V, grad, L);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// ------------------------------------------------------------------------------------------------
// Float64 versions of the per-finger / per-cell / per-object stages (smallnet.h).  Sizes are KBs..MBs: a thread per output column,
// RB rows per workgroup, X staged in LDS as doubles.
template <int ACT, int RB>
__global__ __launch_bounds__(256) void linear64_kernel(const float *__restrict__ Xf, const double *__restrict__ Xd, int ldx,
                                                       const double *__restrict__ WT, const double *__restrict__ bias,
                                                       const double *__restrict__ rowbias, int rb_div, double *__restrict__ Yd,
                                                       float *__restrict__ Yf, int ldy, int rows, int K, int N) {
    constexpr int KC = 64;
    __shared__ double xs[RB][KC];
    const int r0 = blockIdx.x * RB;
    const int n = blockIdx.y * 256 + threadIdx.x;
    const int nn = min(n, N - 1);
    double acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0;
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int kc = min(KC, K - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < RB * KC; i += 256) {
            const int r = i / KC, k = i - r * KC;
            double v = 0.0;
            if (r0 + r < rows && k < kc) v = Xd ? Xd[(size_t)(r0 + r) * ldx + k0 + k] : (double)Xf[(size_t)(r0 + r) * ldx + k0 + k];
            xs[r][k] = v;
        }
        __syncthreads();
        const double *wp = WT + (size_t)k0 * N + nn;
        for (int k = 0; k < kc; k += 8) {       // eight weight loads in flight per step (the loop is L2-latency bound otherwise)
            double w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = wp[(size_t)min(k + j, kc - 1) * N];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k + j < kc) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r] = fma(xs[r][k + j], w[j], acc[r]);
                }
            }
        }
    }
    if (n >= N) return;
    const double bn = bias ? bias[n] : 0.0;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        if (r0 + r >= rows) break;
        double v = acc[r] + bn;
        if (rowbias) v += rowbias[(size_t)((r0 + r) / rb_div) * N + n];
        if (ACT == ACT_RELU) v = fmax(v, 0.0);
        if (ACT == ACT_SILU) v = v / (1.0 + exp(-v));
        if (Yd) Yd[(size_t)(r0 + r) * ldy + n] = v;
        if (Yf) Yf[(size_t)(r0 + r) * ldy + n] = (float)v;
    }
}

int linear64(const float *Xf, const double *Xd, int ldx, const double *WT, const double *bias, const double *rowbias, int rb_div,
             double *Yd, float *Yf, int ldy, int rows, int K, int N, int act, hipStream_t s) {
    if (rows <= 0) return DGDM_OK;
    constexpr int RB = 8;
    dim3 grid((rows + RB - 1) / RB, (N + 255) / 256);
    if (act == ACT_NONE)
        hipLaunchKernelGGL((linear64_kernel<ACT_NONE, RB>), grid, dim3(256), 0, s, Xf, Xd, ldx, WT, bias, rowbias, rb_div, Yd, Yf, ldy, rows, K, N);
    else if (act == ACT_RELU)
        hipLaunchKernelGGL((linear64_kernel<ACT_RELU, RB>), grid, dim3(256), 0, s, Xf, Xd, ldx, WT, bias, rowbias, rb_div, Yd, Yf, ldy, rows, K, N);
    else
        hipLaunchKernelGGL((linear64_kernel<ACT_SILU, RB>), grid, dim3(256), 0, s, Xf, Xd, ldx, WT, bias, rowbias, rb_div, Yd, Yf, ldy, rows, K, N);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

__global__ void gather_add64_kernel(const double *__restrict__ a, const int *__restrict__ idx, const double *__restrict__ b,
                                    double *__restrict__ out, int groups, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * N) return;
    const int g = i / N, n = i - g * N;
    out[i] = a[(size_t)idx[g] * N + n] + b[n];
}

int gather_add64(const double *a, const int *idx, const double *b, double *out, int groups, int N, hipStream_t s) {
    if (groups <= 0) return DGDM_OK;
    hipLaunchKernelGGL(gather_add64_kernel, dim3((groups * N + 255) / 256), dim3(256), 0, s, a, idx, b, out, groups, N);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

template <int W1>
__global__ __launch_bounds__(256) void dyn_post64_kernel(const float *__restrict__ partial, int tiles_per_b,
                                                         const double *__restrict__ w1c /*[W1][256]*/,
                                                         const double *__restrict__ g2w /*[256][256]*/,
                                                         const double *__restrict__ g0w /*[256][L]*/,
                                                         const double *__restrict__ V /*[rows][256] relu(g0 x + b)*/,
                                                         float *__restrict__ grad /*[rows][L]*/, int L) {
    __shared__ double da[W1];
    __shared__ double v1[256];
    __shared__ double v2[256];
    const int row = blockIdx.x, t = threadIdx.x;
    for (int f = t; f < W1; f += 256) {
        const float *src = partial + (size_t)row * tiles_per_b * W1 + f;
        double acc = 0.0;
        for (int ct = 0; ct < tiles_per_b; ++ct) acc += (double)src[(size_t)ct * W1];
        da[f] = acc;
    }
    __syncthreads();
    // sum_f w[f][t] x[f] with eight weight loads in flight per step (F a multiple of 8)
    auto dot = [&](const double *__restrict__ w, int ld, const double *x, int F) {
        double acc = 0.0;
        for (int f = 0; f < F; f += 8) {
            double wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = w[(size_t)(f + j) * ld + t];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fma(wv[j], x[f + j], acc);
        }
        return acc;
    };
    v1[t] = dot(w1c, 256, da, W1);
    __syncthreads();
    const double a2 = dot(g2w, 256, v1, 256);
    v2[t] = V[(size_t)row * 256 + t] > 0.0 ? a2 : 0.0;
    __syncthreads();
    if (t < L) grad[(size_t)row * L + t] = (float)dot(g0w, L, v2, 256);
}

int dyn_post64(int W1, const float *partial, int tiles_per_b, const double *w1c, const double *g2w, const double *g0w, const double *V64,
               float *grad, int rows, int L, hipStream_t s) {
    if (rows <= 0) return DGDM_OK;
    DGDM_REQUIRE(L <= 256, DGDM_EINVAL, "params_ch %d > 256 not supported", L);
    if (W1 == 256)
        hipLaunchKernelGGL(dyn_post64_kernel<256>, dim3(rows), dim3(256), 0, s, partial, tiles_per_b, w1c, g2w, g0w, V64, grad, L);
    else
        hipLaunchKernelGGL(dyn_post64_kernel<512>, dim3(rows), dim3(256), 0, s, partial, tiles_per_b, w1c, g2w, g0w, V64, grad, L);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// ------------------------------------------------------------------------------------------------
// a13/a14: guidance combine + DDIM(eta=0, clip) step, elementwise (see dgdm_hip.h)
__global__ void ddim_step_kernel(const float *__restrict__ x, const float *__restrict__ eps, const float *__restrict__ grad,
                                 int n_grad, float *__restrict__ out, int64_t n, float sa, float sb, float sap, float sbp,
                                 float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float e = eps[i];
    if (grad) {
        float g = grad[i];
        if (n_grad > 1) {                       // grad = 0.0; grad += g_k; grad /= n   (diffusion.py:640-644)
            g = 0.f + g;
            for (int k = 1; k < n_grad; ++k) g += grad[(size_t)k * n + i];
            g = g / (float)n_grad;
        }
        e = sub_rn(e, mul_rn(mul_rn(sb, g), scale));           // eps - (sqrt(1-abar) * grad) * scale, unfused
    }
    float x0 = __fdiv_rn(sub_rn(x[i], mul_rn(sb, e)), sa);        // (x - sqrt(1-abar) eps) / sqrt(abar)
    x0 = fminf(fmaxf(x0, -1.f), 1.f);
    out[i] = add_rn(mul_rn(sap, x0), mul_rn(sbp, e));
}

__global__ void add_noise_kernel(const float *__restrict__ x0, const float *__restrict__ noise, float *__restrict__ out, int64_t n,
                                 float sa, float sb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = add_rn(mul_rn(sa, x0[i]), mul_rn(sb, noise[i]));
}

}  // namespace dgdm

extern "C" int dgdm_ddim_guided_step(const float *x_dev, const float *eps_dev, const float *grad_dev, int n_grad, float *x_next_dev,
                                     int64_t n, float sqrt_abar_t, float sqrt_1m_abar_t, float sqrt_abar_prev,
                                     float sqrt_1m_abar_prev, float guidance_scale, void *stream) {
    DGDM_REQUIRE(x_dev && eps_dev && x_next_dev && n >= 0, DGDM_EINVAL, "dgdm_ddim_guided_step: null argument");
    if (n == 0) return DGDM_OK;
    dgdm::prof_begin((hipStream_t)stream, DGDM_STAGE_DDIM);
    hipLaunchKernelGGL(dgdm::ddim_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, eps_dev,
                       grad_dev, n_grad, x_next_dev, n, sqrt_abar_t, sqrt_1m_abar_t, sqrt_abar_prev, sqrt_1m_abar_prev, guidance_scale);
    dgdm::prof_end((hipStream_t)stream, DGDM_STAGE_DDIM, (double)n * 4.0 * (3 + n_grad));
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

extern "C" int dgdm_ddim_add_noise(const float *x0_dev, const float *noise_dev, float *out_dev, int64_t n, float sqrt_abar,
                                   float sqrt_1m_abar, void *stream) {
    DGDM_REQUIRE(x0_dev && noise_dev && out_dev && n >= 0, DGDM_EINVAL, "dgdm_ddim_add_noise: null argument");
    if (n == 0) return DGDM_OK;
    hipLaunchKernelGGL(dgdm::add_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x0_dev, noise_dev,
                       out_dev, n, sqrt_abar, sqrt_1m_abar);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
