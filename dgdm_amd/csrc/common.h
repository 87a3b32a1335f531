// Shared host/device helpers for libdgdm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/dgdm_hip.h"

#define DGDM_MAX_CHAINS 256

namespace dgdm {

// a * b, a + b, a - b each rounded on its own.  HIP's __fmul_rn / __fadd_rn are plain operators defined in HIP's headers: inlined,
// they carry the `contract` flag of the default -ffp-contract=fast whatever `#pragma clang fp contract(off)` says at the call site,
// and the backend fuses them with a neighbouring operation when it likes the pattern (seen in the ISA of sqdist_rows_kernel: the sum of
// squares became v_pk_mul + 2 x v_pk_fma, one rounding instead of five - 20 % of the squared distances off by an ulp from torch's, and a
// ball-query decision flipped for a point within 1e-7 of the radius).  Inline assembly is opaque to the optimiser: these are what the
// code that must reproduce the reference's separate float32 roundings uses.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float mul_rn(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sub_rn(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
#endif

void set_error(const char *fmt, ...);

#define DGDM_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            dgdm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DGDM_EHIP;                                                             \
        }                                                                                 \
    } while (0)

#define DGDM_REQUIRE(cond, code, ...)                                                     \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            dgdm::set_error(__VA_ARGS__);                                                 \
            return (code);                                                                \
        }                                                                                 \
    } while (0)

// Device buffer owned by a model/guidance handle.
struct DevBuf {
    void  *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        if (p && bytes >= n) return DGDM_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (n == 0) return DGDM_OK;
        DGDM_HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
        return DGDM_OK;
    }
    int upload(const void *host, size_t n) {
        int rc = alloc(n);
        if (rc) return rc;
        DGDM_HIP_CHECK(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
        return DGDM_OK;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// state_dict lookup
struct StateDict {
    std::map<std::string, const DgdmTensor *> m;
    StateDict(const DgdmTensor *t, int n) { for (int i = 0; i < n; ++i) m[t[i].name] = &t[i]; }
    const float *f32(const std::string &k, int64_t numel) const;   // nullptr + error when missing / wrong size
    bool has(const std::string &k) const { return m.count(k) != 0; }
};

// Linear weight [out][in] with an optional eval-mode BatchNorm folded in (double precision fold):
//   y = s*(W x + b - mean) + beta,  s = gamma / sqrt(var + eps)
struct Folded {
    int out = 0, in = 0;
    std::vector<float> w;   // [out][in]
    std::vector<float> b;   // [out]
};
int fold_linear(const StateDict &sd, const std::string &lin, const std::string &bn /* "" = none */, int out, int in,
                Folded *dst);
// The same fold kept in double precision (nothing rounded to float32): weights of the stages that run in float64 because they are
// evaluated once per object / per finger instead of once per replicated row (DESIGN_HISTORY.md 4.9)
struct Folded64 {
    int out = 0, in = 0;
    std::vector<double> w;  // [out][in]
    std::vector<double> b;  // [out]
};
int fold_linear64(const StateDict &sd, const std::string &lin, const std::string &bn /* "" = none */, int out, int in, Folded64 *dst);
std::vector<double> transpose64(const double *src, int rows, int cols);   // -> [cols][rows]
// A-operand image of W [M][K] (row-major doubles, M a multiple of 32, K of 4) for v_mfma_f64_16x16x4_f64 (pointnet64.hip):
//   entry e = ks * (M/32) + mp, lane (m = l & 15, kq = l >> 4), slot j in {0, 1}  =  W[f(2 mp + j, m)][kq * (K/4) + ks],
//   f(mt, m) = 16 mt + 4 (m & 3) + (m >> 2)   (so that lane group rb of the C/D layout holds four CONSECUTIVE features)
std::vector<double> pack_mfma64(const double *src, int M, int K);

// ---- MFMA-chain weight image (see mfma_chain.h).  src is [M][K] row-major, M,K multiples of 32.
//   img[((op*KB + o)*4 + q)*64 + lane][0..3] = src[32 op + (lane&31)][32 o + 8 q + 4 (lane>>5) + 0..3]
std::vector<float> pack_chain(const float *src, int M, int K);
uint16_t f32_to_bf16(float x);                                              // round to nearest even
std::vector<uint16_t> pack_chain_bf16(const float *src, int M, int K);     // A-operand image for trunk_bf16.hip
std::vector<float> transpose(const float *src, int rows, int cols);   // -> [cols][rows]

// profiling (dgdm_prof_*): HIP events on the launch stream around one stage; `work` = its algorithmic FLOPs (MFMA-bound stages)
// or bytes (HBM-bound stages), 0 when the host cannot know it
constexpr int PROF_BUCKETS = DGDM_STAGE_COUNT;
void prof_begin(hipStream_t s, int stage);
void prof_end(hipStream_t s, int stage, double work);

}  // namespace dgdm
