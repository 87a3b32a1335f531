// Microbenchmark: cycles per MFMA for dependent chains with 1 / 2 / 4 alternating accumulators, optionally with VALU fillers
// between MFMAs.  One wave per SIMD (launch 256 threads x #CUs).  Build: hipcc --offload-arch=gfx950 -O3 mfma_dep.hip -o mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int FILL, bool BF16>
__global__ __launch_bounds__(256, 1) void k(float *out, long long *cyc, float a0, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
    float a = a0 + threadIdx.x, b = a0 * 2.f;
    i32x4 av = {(int)threadIdx.x, 1, 2, 3}, bv = {4, 5, 6, (int)threadIdx.x};
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = a0 + i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < FILL; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[q % 8]) : "v"(b));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// the f16x3 trunk's group: six f16 MFMAs on two accumulators + the splitting of one input pair (trunk_f16l.hip item()); VARIANT 0: as the
// kernel has it, 1: without the splitting, 2: the splitting's results do not feed the MFMAs, 3: 0 + the four A operands of every group
// read from LDS one group ahead (ds_read_b128), 4: 1 + those reads, 5: 4 with the reads' results unused (MFMAs on constant operands)
typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int VARIANT>
__global__ __launch_bounds__(256, 1) void kgroup(float *out, long long *cyc, float a0, int iters) {
    f32x16 acc[8], yp;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
    for (int r = 0; r < 16; ++r) yp[r] = a0 * r - 3.f;
    u32x4 w[4], P[2][2];
    for (int j = 0; j < 4; ++j) w[j] = u32x4{threadIdx.x + j, 1u, 2u, 3u};
    for (int j = 0; j < 4; ++j) P[j / 2][j % 2] = u32x4{4u, 5u, threadIdx.x, 7u + j};
    const float f = a0 * 4096.f;
    unsigned mk = 0;
    __shared__ __attribute__((aligned(16))) u32x4 lbuf[4096];            // 64 KiB
    for (int i = threadIdx.x; i < 4096; i += 256) lbuf[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 wn[4], sink = {0u, 0u, 0u, 0u};
    for (int j = 0; j < 4; ++j) wn[j] = w[j];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int pp = g & 3, par = (g >> 2) & 1;
            if (VARIANT >= 3) {
                if (VARIANT != 5) { for (int j = 0; j < 4; ++j) w[j] = wn[j]; }
                else { for (int j = 0; j < 4; ++j) sink ^= wn[j]; }
                for (int j = 0; j < 4; ++j) wn[j] = lbuf[(((it & 7) * 8 + g) * 4 + j) * 64 + lane];
            }
            if (VARIANT != 1 && VARIANT < 4) {
                float lo = yp[2 * g], hi = yp[2 * g + 1];
                mk |= (lo > 0.f ? 1u : 0u) << (2 * g);
                mk |= (hi > 0.f ? 1u : 0u) << (2 * g + 1);
                asm("v_max_f32 %0, 0, %1" : "=v"(lo) : "v"(lo));
                asm("v_max_f32 %0, 0, %1" : "=v"(hi) : "v"(hi));
                const hf32x2 v = {lo * f, hi * f};
                const hf16x2 h = __builtin_convertvector(v, hf16x2);
                const hf32x2 d = v - __builtin_convertvector(h, hf32x2);
                const unsigned a = __builtin_bit_cast(unsigned, h), b = __builtin_bit_cast(unsigned, __builtin_convertvector(d, hf16x2));
                if (VARIANT == 0) { P[par ^ 1][0][g % 4] = a; P[par ^ 1][1][g % 4] = b; }
                else { yp[2 * g] = __builtin_bit_cast(float, a); yp[2 * g + 1] = __builtin_bit_cast(float, b); }
            }
            const hf16x8 xh = __builtin_bit_cast(hf16x8, P[par][0]), xl = __builtin_bit_cast(hf16x8, P[par][1]);
            f32x16 &A = acc[2 * pp], &B = acc[2 * pp + 1];
            A = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, w[1]), xh, A, 0, 0, 0);
            B = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, w[3]), xh, B, 0, 0, 0);
            A = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, w[0]), xl, A, 0, 0, 0);
            B = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, w[2]), xl, B, 0, 0, 0);
            A = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, w[0]), xh, A, 0, 0, 0);
            B = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, w[2]), xh, B, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = (float)mk + (float)(sink.x ^ sink.y ^ sink.z ^ sink.w);
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 16; ++r) s += yp[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int VARIANT>
void run_group(const char *name) {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kgroup<VARIANT>), dim3(256), dim3(256), 0, 0, out, cyc, 1.f, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-60s: %.1f ticks per group of 6 MFMAs\n", name, (double)c / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}

template <int NACC, int FILL, bool BF16>
void run(const char *name) {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 200;
    hipLaunchKernelGGL((k<NACC, FILL, BF16>), dim3(256), dim3(256), 0, 0, out, cyc, 1.f, iters);
    hipLaunchKernelGGL((k<NACC, FILL, BF16>), dim3(256), dim3(256), 0, 0, out, cyc, 1.f, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s nacc=%d fill=%d : %.1f clk-counter ticks per MFMA\n", name, NACC, FILL, (double)c / (iters * 16.0 * NACC));
    hipFree(out); hipFree(cyc);
}

int main() {
    run_group<1>("f16x3 group, MFMAs only");
    run_group<2>("f16x3 group + input splitting (results unused by the MFMAs)");
    run_group<0>("f16x3 group + input splitting feeding the next block");
    run_group<4>("f16x3 group, MFMAs + 4 ds_read_b128 feeding the next group");
    run_group<5>("f16x3 group, MFMAs + 4 ds_read_b128 (results unused)");
    run_group<3>("f16x3 group + splitting + 4 ds_read_b128");
    // s_memtime counts at a fixed 100 MHz; print a calibration with a known dependent chain first
    run<1, 0, false>("f32 32x32x2"); run<2, 0, false>("f32 32x32x2"); run<4, 0, false>("f32 32x32x2");
    run<1, 1, false>("f32 32x32x2"); run<1, 2, false>("f32 32x32x2"); run<2, 2, false>("f32 32x32x2"); run<1, 4, false>("f32 32x32x2"); run<2, 4, false>("f32 32x32x2");
    run<1, 8, false>("f32 32x32x2"); run<2, 8, false>("f32 32x32x2"); run<1, 12, false>("f32 32x32x2"); run<2, 12, false>("f32 32x32x2");
    run<1, 0, true>("bf16 32x32x16"); run<2, 0, true>("bf16 32x32x16"); run<4, 0, true>("bf16 32x32x16");
    run<1, 2, true>("bf16 32x32x16"); run<2, 2, true>("bf16 32x32x16"); run<1, 4, true>("bf16 32x32x16"); run<2, 4, true>("bf16 32x32x16");
    run<2, 5, true>("bf16 32x32x16"); run<2, 6, true>("bf16 32x32x16"); run<2, 8, true>("bf16 32x32x16");
    return 0;
}
