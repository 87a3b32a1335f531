// Microbenchmark: how accurate is a float32 contraction emulated on the 16-bit matrix pipes?
//
// D[32 x 32] = W[32 x K] * X[K x 32] + bias with float32 W, X (K = 256, the trunk's layer shape), computed
//   (a) on v_mfma_f32_32x32x2_f32 - the exact-float32 k-ordered fma chain the trunk uses today,
//   (b) split-f16: w = wh + 2^-11 wl', x = xh + 2^-11 xl' (f16 pieces, lo scaled by 2^11 into the normal range), three
//       v_mfma_f32_32x32x16_f16 per K-step: hh into one accumulator, hl' + l'h into a second one; D = acc_hh + 2^-11 acc_cross,
//   (c) split-bf16: three bf16 pieces each, six v_mfma_f32_32x32x16_bf16 per K-step (hh | hm + mh | mm + hl + lh),
// against a float64 evaluation on the host.  Prints, per method, the rms and the MEAN (bias) of the error in units of the rms
// output and the worst tile.  Build: hipcc --offload-arch=gfx950 -O3 split_mfma.hip -o split_mfma.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 256, TILES = 2048;

__device__ inline int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// W [tile][32][K], X [tile][K][32] (row n contiguous: X[k][n]), out [tile][32][32]
__global__ void k_f32(const float *W, const float *X, const float *bias, float *out) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5, t = blockIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = bias[crow(r, h)];
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[((size_t)t * 32 + i) * K + k + h], X[((size_t)t * K + k + h) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[((size_t)t * 32 + crow(r, h)) * 32 + i] = acc[r];
}

__global__ void k_f16(const float *W, const float *X, const float *bias, float *out, int mode) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5, t = blockIdx.x;
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = bias[crow(r, h)]; a1[r] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 wh, wl, xh, xl;
        for (int j = 0; j < 8; ++j) {
            const float w = W[((size_t)t * 32 + i) * K + k0 + 8 * h + j], x = X[((size_t)t * K + k0 + 8 * h + j) * 32 + i];
            wh[j] = (_Float16)w; wl[j] = (_Float16)((w - (float)wh[j]) * 2048.f);
            xh[j] = (_Float16)x; xl[j] = (_Float16)((x - (float)xh[j]) * 2048.f);
        }
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, a1, 0, 0, 0);

    }

    for (int r = 0; r < 16; ++r) out[((size_t)t * 32 + crow(r, h)) * 32 + i] = a0[r] + a1[r] * (1.f / 2048.f);
}

__global__ void k_bf16(const float *W, const float *X, const float *bias, float *out) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5, t = blockIdx.x;
    f32x16 a0, a1, a2;
    for (int r = 0; r < 16; ++r) { a0[r] = bias[crow(r, h)]; a1[r] = 0.f; a2[r] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 wp[3], xp[3];
        for (int j = 0; j < 8; ++j) {
            float w = W[((size_t)t * 32 + i) * K + k0 + 8 * h + j], x = X[((size_t)t * K + k0 + 8 * h + j) * 32 + i];
            for (int p = 0; p < 3; ++p) {
                wp[p][j] = (__bf16)w; w -= (float)wp[p][j];
                xp[p][j] = (__bf16)x; x -= (float)xp[p][j];
            }
        }
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[0], xp[0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[0], xp[1], a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[1], xp[0], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[1], xp[1], a2, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[0], xp[2], a2, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[2], xp[0], a2, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out[((size_t)t * 32 + crow(r, h)) * 32 + i] = a0[r] + (a1[r] + a2[r]);
}

// (d) the form the trunk can afford: the six terms of a K-step on TWO alternating accumulators that run through all K-steps (a third
//     of the registers of (c)), added once at the end; small terms first within a K-step
__global__ void k_bf16_2acc(const float *W, const float *X, const float *bias, float *out) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5, t = blockIdx.x;
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = bias[crow(r, h)]; a1[r] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 wp[3], xp[3];
        for (int j = 0; j < 8; ++j) {
            float w = W[((size_t)t * 32 + i) * K + k0 + 8 * h + j], x = X[((size_t)t * K + k0 + 8 * h + j) * 32 + i];
            for (int p = 0; p < 3; ++p) {
                wp[p][j] = (__bf16)w; w -= (float)wp[p][j];
                xp[p][j] = (__bf16)x; x -= (float)xp[p][j];
            }
        }
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[2], xp[0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[0], xp[2], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[1], xp[1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[1], xp[0], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[0], xp[1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[0], xp[0], a1, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out[((size_t)t * 32 + crow(r, h)) * 32 + i] = a0[r] + a1[r];
}

// (e) the form a register-resident trunk can afford with f16: ONE accumulator per output, both operands pre-scaled by powers of two so
//     that their low pieces stay in f16's normal range (weights: one scale per layer, max |w_s| in [2^12, 2^13); activations: one scale per
//     row, max |x_s| in [2^12, 2^13)), three MFMAs per K-step (lh, hl, hh), the scale taken out of the result.  mode 1: hi pieces rounded
//     toward zero (v_cvt_pkrtz), lo pieces to nearest.
__global__ void k_f16_1acc(const float *W, const float *X, const float *bias, float *out, float wscale, int rtz) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5, t = blockIdx.x;
    // per-row (column i of X) scale from the row's maximum
    float m = 0.f;
    for (int k = 0; k < K; ++k) m = fmaxf(m, fabsf(X[((size_t)t * K + k) * 32 + i]));
    const int e = (int)((__float_as_uint(m) >> 23) & 0xff) - 127;
    const float xs = m > 0.f ? __uint_as_float((uint32_t)(12 - e + 127) << 23) : 1.f;
    f32x16 a0;
    for (int r = 0; r < 16; ++r) a0[r] = bias[crow(r, h)] * wscale * xs;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 wh, wl, xh, xl;
        for (int j = 0; j < 8; ++j) {
            const float w = W[((size_t)t * 32 + i) * K + k0 + 8 * h + j] * wscale, x = X[((size_t)t * K + k0 + 8 * h + j) * 32 + i] * xs;
            if (rtz) {
                wh[j] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, (_Float16)w)));      // weights: host side, nearest
                xh[j] = __builtin_amdgcn_cvt_pkrtz(x, 0.f)[0];
            } else { wh[j] = (_Float16)w; xh[j] = (_Float16)x; }
            wl[j] = (_Float16)(w - (float)wh[j]);
            xl[j] = (_Float16)(x - (float)xh[j]);
        }
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, a0, 0, 0, 0);
    }
    const float un = 1.f / (wscale * xs);
    for (int r = 0; r < 16; ++r) out[((size_t)t * 32 + crow(r, h)) * 32 + i] = a0[r] * un;
}

int main() {
    std::mt19937 gen(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> W((size_t)TILES * 32 * K), X((size_t)TILES * K * 32), bias(32);
    for (auto &v : W) v = nd(gen) * 0.0884f;                       // He-init scale sqrt(2/256)
    for (auto &v : X) { v = nd(gen); v = v > 0.f ? v : 0.f; }      // post-ReLU activations
    for (auto &v : bias) v = nd(gen) * 0.1f;
    std::vector<double> ref((size_t)TILES * 32 * 32);
    for (int t = 0; t < TILES; ++t)
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double s = bias[m];
                for (int k = 0; k < K; ++k) s += (double)W[((size_t)t * 32 + m) * K + k] * (double)X[((size_t)t * K + k) * 32 + n];
                ref[((size_t)t * 32 + m) * 32 + n] = s;
            }
    float *dW, *dX, *db, *dout;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&db, 128); hipMalloc(&dout, ref.size() * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bias.data(), 128, hipMemcpyHostToDevice);
    std::vector<float> out(ref.size());
    double rms_ref = 0;
    for (double v : ref) rms_ref += v * v;
    rms_ref = std::sqrt(rms_ref / ref.size());
    auto report = [&](const char *name) {
        hipDeviceSynchronize();
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        double se = 0, me = 0, worst = 0, near0 = 0;
        long flips = 0;
        for (size_t i = 0; i < out.size(); ++i) {
            const double e = (double)out[i] - ref[i];
            se += e * e; me += e;
            worst = std::max(worst, std::fabs(e));
            flips += ((out[i] > 0.f) != (ref[i] > 0.0));
        }
        (void)near0;
        printf("%-28s rms err / rms out %.3e   mean err / rms out %+.3e   max |err| / rms out %.3e   sign flips vs float64 %ld of %zu\n", name,
               std::sqrt(se / out.size()) / rms_ref, me / out.size() / rms_ref, worst / rms_ref, flips, out.size());
    };
    // the float32 rounding of the exact result, for scale
    for (size_t i = 0; i < out.size(); ++i) out[i] = (float)ref[i];
    hipMemcpy(dout, out.data(), out.size() * 4, hipMemcpyHostToDevice);
    report("float32(exact)");
    hipLaunchKernelGGL(k_f32, dim3(TILES), dim3(64), 0, 0, dW, dX, db, dout); report("mfma f32 32x32x2 chain");
    hipLaunchKernelGGL(k_f16, dim3(TILES), dim3(64), 0, 0, dW, dX, db, dout, 0); report("split-f16, 3 mfma / K-step");
    {
        float wmax = 0.f;
        for (float v : W) wmax = std::max(wmax, std::fabs(v));
        int e; std::frexp(wmax, &e);                               // wmax = f * 2^e, f in [0.5, 1)
        const float wscale = std::ldexp(1.f, 13 - e);              // max |w_s| in [2^12, 2^13)
        hipLaunchKernelGGL(k_f16_1acc, dim3(TILES), dim3(64), 0, 0, dW, dX, db, dout, wscale, 0); report("split-f16 scaled, 1 acc, rne");
        hipLaunchKernelGGL(k_f16_1acc, dim3(TILES), dim3(64), 0, 0, dW, dX, db, dout, wscale, 1); report("split-f16 scaled, 1 acc, rtz hi");
    }
    hipLaunchKernelGGL(k_bf16, dim3(TILES), dim3(64), 0, 0, dW, dX, db, dout); report("split-bf16, 6 mfma / K-step");
    hipLaunchKernelGGL(k_bf16_2acc, dim3(TILES), dim3(64), 0, 0, dW, dX, db, dout); report("split-bf16, 6 mfma, 2 acc");
    return 0;
}
