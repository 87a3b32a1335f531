// LDS read bandwidth per CU for the access shapes of the shared weight stream: every lane of a wave reads 16 (or 8) consecutive bytes,
// the wave 1 KiB (512 B) linear; WAVES waves per CU read at once.  hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));

template <int WIDTH>
__global__ __launch_bounds__(1024, 1) void k(float *out, long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[16384];           // 64 KiB
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    v4 s = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (WIDTH == 16) { const v4 v = *reinterpret_cast<const v4 *>(buf + ((it & 3) * 16 + e) * 256 + lane * 4); s += v; }
            else { const v2 v = *reinterpret_cast<const v2 *>(buf + ((it & 7) * 16 + e) * 128 + lane * 2); s.x += v.x; s.y += v.y; }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int WIDTH>
void run(int waves) {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<WIDTH>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("ds_read_b%d, %2d waves per CU: %.1f ticks per instruction per wave, %.1f B/tick/CU\n", WIDTH * 8, waves, (double)c / (iters * 16.0),
           (double)iters * 16 * 64 * WIDTH * waves / c);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4, 8, 16}) run<16>(w);
    for (int w : {1, 2, 4, 8, 16}) run<8>(w);
    return 0;
}
