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
    // s_memtime counts at a fixed 100 MHz; print a calibration with a known dependent chain first
    run<1, 0, false>("f32 32x32x2"); run<2, 0, false>("f32 32x32x2"); run<4, 0, false>("f32 32x32x2");
    run<1, 1, false>("f32 32x32x2"); run<1, 2, false>("f32 32x32x2"); run<2, 2, false>("f32 32x32x2"); run<1, 4, false>("f32 32x32x2"); run<2, 4, false>("f32 32x32x2");
    run<1, 8, false>("f32 32x32x2"); run<2, 8, false>("f32 32x32x2"); run<1, 12, false>("f32 32x32x2"); run<2, 12, false>("f32 32x32x2");
    run<1, 0, true>("bf16 32x32x16"); run<2, 0, true>("bf16 32x32x16"); run<4, 0, true>("bf16 32x32x16");
    run<1, 2, true>("bf16 32x32x16"); run<2, 2, true>("bf16 32x32x16"); run<1, 4, true>("bf16 32x32x16"); run<2, 4, true>("bf16 32x32x16");
    run<2, 5, true>("bf16 32x32x16"); run<2, 6, true>("bf16 32x32x16"); run<2, 8, true>("bf16 32x32x16");
    return 0;
}
