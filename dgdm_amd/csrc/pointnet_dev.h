// Device helpers shared by pointnet.hip and pointnet64.hip: the reference's float32 distance forms (index decisions are the
// float32 ones in every build mode) and sa1's ball query.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

// No implicit a*b+c -> fma contraction: FPS and ball-query distances must round like the reference's separate float32 ops: mul_rn /
// add_rn (common.h).  torch.sum(v ** 2, -1) is ((x^2 + y^2) + z^2) and torch.matmul's K = 3 dot product is the fma chain
// fma(z z', fma(y y', x x')) (both checked bitwise against torch CPU, here and on the GPU box's host).
#pragma clang fp contract(off)

namespace dgdm {

__device__ __forceinline__ float sq3(float x, float y, float z) {
    return add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z));
}

// square_distance(src = centre, dst = candidate) in the reference's expanded form
__device__ __forceinline__ float sqdist_expanded(float cx, float cy, float cz, float cn, float px, float py, float pz, float pn) {
    const float dot = fmaf(cz, pz, fmaf(cy, py, mul_rn(cx, px)));
    return add_rn(add_rn(mul_rn(-2.f, dot), cn), pn);
}

// query_ball_point (pointnet2_utils.py:95-115) for one centre by one wave: the first 32 in-radius indices in index order,
// padded with the first one (:112-114), into nbr[0..32) (LDS).  Shared by sa1_kernel and the index test hook.
__device__ __forceinline__ void ball_first32(const float *__restrict__ xyz, int N, int p, float cx, float cy, float cz, float cn, float r2,
                                             int *nbr, int lane) {
    int cnt = 0;
    for (int base = 0; base < N && cnt < 32; base += 64) {
        const int k = base + lane;
        bool in = false;
        if (k < N) {
            const float x = xyz[3 * k], y = xyz[3 * k + 1], z = xyz[3 * k + 2];
            in = !(sqdist_expanded(cx, cy, cz, cn, x, y, z, sq3(x, y, z)) > r2);
        }
        const unsigned long long m = __ballot(in);
        const int rank = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (in && rank < 32) nbr[rank] = k;
        cnt += __popcll(m);
    }
    cnt = min(cnt, 32);
    __builtin_amdgcn_wave_barrier();
    if (lane >= cnt && lane < 32) nbr[lane] = (cnt > 0) ? nbr[0] : p;   // pad with the first (:112-114)
}

}  // namespace dgdm
