// PointNet++ SSG object encoder (dynamics/models/pointnet2.py:11-32, pointnet2_utils.py:27-210)
// restructured around what Diffusion.cond_fn actually feeds it: R replicas of ONE cloud that differ
// only in the two random FPS start indices (s1 for sa1, s2 for sa2) drawn per row
// (pointnet2_utils.py:83 via generator/diffusion.py:491-496).
//
// Facts used (DESIGN_HISTORY.md §4 derives them; all are exact, no approximation):
//  * sa1 samples npoint = 512 of N points: FPS from start s1 yields an ordering fps1[s1][0..511] of
//    point ids ("variant" s1).  The sa1 feature of a centre depends on the centre POINT only
//    (ball query scans the original order), so F1[p] is computed once per cloud.
//  * sa2 works on the cloud re-ordered by fps1[s1].  Its ball query keeps the first 64 in-radius
//    neighbours IN THAT ORDER, so the sa2 feature of a centre depends on (variant, centre point):
//    L2[v][c] = max over those <=64 neighbours k of Y[c][k], where Y[c][k] is the two-layer pointwise
//    MLP on the pair (centre c, neighbour k) - a function of the two POINTS only.
//  * sa3 (group_all) applies a pointwise layer to [xyz_c | L2[v][c]] and takes the max over the 128
//    centres FPS selected from start s2: Z[v][c] once per (variant, point), then per row only
//    FPS(128) on the re-ordered cloud and a max over 128 rows of Z[v].
// The first sa2 conv is linear in [xyz_k - xyz_c | F1[k]] (pointnet2_utils.py:136-140), so its
// feature part U[k] = W[:,3:] F1[k] + b is computed once per point.
//
// Distances replicate the reference's float32 operation order, without fma contraction:
//   FPS   : (dx*dx + dy*dy) + dz*dz                         (pointnet2_utils.py:88)
//   balls : ((-2 * (a.b)) + |a|^2) + |b|^2  >  r^2           (pointnet2_utils.py:45-47,110)
#include "common.h"
#include "mfma_chain.h"
#include <type_traits>
#include "pointnet.h"
#include "pointnet_dev.h"

// No implicit a*b+c -> fma contraction in this file: the scheduler update, FPS and ball-query distances must round
// like the reference's separate float32 ops (HIP's __fmul_rn/__fadd_rn are plain * and + and would be contracted).
// Explicit fmaf() calls are unaffected.
#pragma clang fp contract(off)

namespace dgdm {

// wave-wide argmax, first index wins ties (torch.max semantics, pointnet2_utils.py:91); every lane receives the result.
// Two DPP reductions (v_max_f32 / v_min_u32 with data-parallel-primitive operands: no LDS crossbar traffic, unlike __shfl_xor,
// which is ds_bpermute_b32): the maximum, then the smallest index among the lanes that hold it.
template <class T, class Op>
__device__ __forceinline__ T dpp_reduce(T v, Op op) {
    auto step = [&](auto ctrl, auto rmask) {
        constexpr int C = decltype(ctrl)::value, M = decltype(rmask)::value;
        const int o = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), C, M, 0xf, false);
        v = op(v, __builtin_bit_cast(T, o));
    };
    step(std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});     // quad_perm [1,0,3,2]: lane ^ 1
    step(std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});     // quad_perm [2,3,0,1]: lane ^ 2
    step(std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});    // row_half_mirror: the other quad of the 8
    step(std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});    // row_mirror: the other half of the row of 16
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});    // row_bcast15 into rows 1 and 3
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});    // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));   // lane 63 has seen all 64
}

__device__ __forceinline__ void wave_argmax(float &v, int &i) {
    const float m = dpp_reduce(v, [](float a, float b) { return fmaxf(a, b); });
    const unsigned cand = (v == m) ? (unsigned)i : 0xffffffffu;
    i = (int)dpp_reduce(cand, [](unsigned a, unsigned b) { return a < b ? a : b; });
    v = m;
}

// farthest_point_sample (pointnet2_utils.py:71-92) by one wave over M <= 64*PPL points whose
// coordinates sit in LDS (SoA lx/ly/lz, M entries).  Writes npoint indices to out (LDS or global).
// With TIES the return value tells whether the selection was ever order-dependent: two candidates
// with DIFFERENT coordinates shared the maximal distance, or every remaining distance was 0 (all
// distinct coordinates used up, so the reference falls back to position 0 of its own ordering).
// A sequence without such an event is the same for every ordering of the cloud, up to which of
// several bit-identical points represents a coordinate.
template <int PPL, bool TIES>
__device__ bool fps_wave(const float *lx, const float *ly, const float *lz, int M, int start, int npoint, int *out, int lane) {
    float px[PPL], py[PPL], pz[PPL], dist[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int j = lane + 64 * i;
        const bool ok = j < M;
        px[i] = ok ? lx[j] : 0.f; py[i] = ok ? ly[j] : 0.f; pz[i] = ok ? lz[j] : 0.f;
        dist[i] = ok ? 1e10f : -1.f;                 // padding can never be the farthest point
    }
    int far = start;
    bool ambiguous = false;
    for (int it = 0; it < npoint; ++it) {
        if (lane == 0) out[it] = far;
        const float cx = lx[far], cy = ly[far], cz = lz[far];
        float bv = -2.f;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const float d = sq3(px[i] - cx, py[i] - cy, pz[i] - cz);
            if (d < dist[i]) dist[i] = d;
            if (dist[i] > bv) { bv = dist[i]; bi = lane + 64 * i; }
        }
        wave_argmax(bv, bi);
        far = bi;
        if (TIES && it + 1 < npoint) {
            const float wx = lx[far], wy = ly[far], wz = lz[far];
            bool t = bv == 0.f;
#pragma unroll
            for (int i = 0; i < PPL; ++i) t = t || (dist[i] == bv && (px[i] != wx || py[i] != wy || pz[i] != wz));
            ambiguous = ambiguous || t;
        }
    }
    return TIES ? (__ballot(ambiguous) != 0ull) : false;
}

// ------------------------------------------------------------------------------------------------ T1
// fps[obj][v][0..npoint) for v = 0..nv-1 (start index v) on cloud xyz[obj] [N][3]; flags[obj][v] (optional) = selection was order-dependent.
// All objects of a set_objects call go in one launch: the 512 dependent iterations are latency bound, so the more waves the better.
__global__ __launch_bounds__(256) void fps_table_kernel(const float *__restrict__ xyz, int N, int nv, int npoint, int *__restrict__ out,
                                                        int *__restrict__ flags) {
    extern __shared__ float lds[];
    float *lx = lds, *ly = lds + N, *lz = lds + 2 * N;
    // blockIdx.y = object: clouds, tables and flags of the objects lie back to back
    xyz += (size_t)blockIdx.y * N * 3;
    out += (size_t)blockIdx.y * nv * npoint;
    if (flags) flags += (size_t)blockIdx.y * nv;
    for (int i = threadIdx.x; i < N; i += blockDim.x) { lx[i] = xyz[3 * i]; ly[i] = xyz[3 * i + 1]; lz[i] = xyz[3 * i + 2]; }
    __syncthreads();
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= nv) return;
    bool amb;
    if (flags) {
        amb = N <= 512 ? fps_wave<8, true>(lx, ly, lz, N, v, npoint, out + (size_t)v * npoint, lane)
                       : fps_wave<16, true>(lx, ly, lz, N, v, npoint, out + (size_t)v * npoint, lane);
        if (lane == 0) flags[v] = amb ? 1 : 0;
    } else {
        if (N <= 512) fps_wave<8, false>(lx, ly, lz, N, v, npoint, out + (size_t)v * npoint, lane);
        else fps_wave<16, false>(lx, ly, lz, N, v, npoint, out + (size_t)v * npoint, lane);
    }
}

// ------------------------------------------------------------------------------------------------ T2
// sa1 feature of every point p as a centre: first 32 in-radius (r=0.2) points in index order, padded with
// the first; Conv(3->64)+BN+ReLU, Conv(64->128)+BN+ReLU, max  (pointnet2.py:17, pointnet2_utils.py:95-146,203-208)
__global__ __launch_bounds__(128) void sa1_kernel(const float *xyz, int N, float r2, const float *__restrict__ w0t /*[3][64]*/,
                                                  const float *__restrict__ b0, const float *__restrict__ w1 /*[128][64]*/,
                                                  const float *__restrict__ b1, float *F1 /*[N][128]*/) {
    __shared__ int nbr[32];
    __shared__ __attribute__((aligned(16))) float h1[32][64];
    const int t = threadIdx.x, lane = t & 63;
    xyz += (size_t)blockIdx.y * 3 * N; F1 += (size_t)blockIdx.y * N * 128;      // object of a batched launch
    float wrow[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wrow[k] = w1[t * 64 + k];
    const float bias1 = b1[t];
    for (int p = blockIdx.x; p < N; p += gridDim.x) {
        const float cx = xyz[3 * p], cy = xyz[3 * p + 1], cz = xyz[3 * p + 2];
        const float cn = sq3(cx, cy, cz);
        if (t < 64) ball_first32(xyz, N, p, cx, cy, cz, cn, r2, nbr, lane);      // wave 0
        __syncthreads();
        for (int i = t; i < 32 * 64; i += 128) {          // layer 0 on the relative coordinates
            const int s = i >> 6, c = i & 63, k = nbr[s];
            const float dx = xyz[3 * k] - cx, dy = xyz[3 * k + 1] - cy, dz = xyz[3 * k + 2] - cz;
            const float v = fmaf(w0t[128 + c], dz, fmaf(w0t[64 + c], dy, fmaf(w0t[c], dx, 0.f))) + b0[c];
            h1[s][c] = fmaxf(v, 0.f);
        }
        __syncthreads();
        float best = 0.f;                                  // ReLU outputs are >= 0 and the group is never empty
        for (int s = 0; s < 32; ++s) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 64; k += 4) {
                const float4 h = *reinterpret_cast<const float4 *>(&h1[s][k]);
                acc = fmaf(wrow[k + 3], h.w, fmaf(wrow[k + 2], h.z, fmaf(wrow[k + 1], h.y, fmaf(wrow[k], h.x, acc))));
            }
            best = fmaxf(best, fmaxf(acc + bias1, 0.f));
        }
        F1[(size_t)p * 128 + t] = best;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ T4
// Y[pair][256] = ReLU(W2b' ReLU(U[k] + Vx (xyz_k - xyz_c)) + b2b') for every ordered pair (centre c, point k) with k inside
// c's r = 0.4 ball - the only pairs sa2's ball query can ever select, whatever the ordering (crowd_kernel / nbr_fill_kernel
// build the list: pairs[off[c] + j] = (c << 16 | k), j-th in-radius point of c in index order).
// One wave = 32 consecutive pairs.  Layer 128 -> 256 on the MFMA chain.
__global__ __launch_bounds__(256, 1) void pair_kernel(const float *__restrict__ xyz, int N, const float *__restrict__ U /*[N][128]*/,
                                                      const float *__restrict__ vx /*[3][128]*/, const float4 *__restrict__ Wimg,
                                                      const float *__restrict__ bias, const int *__restrict__ pairs,
                                                      const int *__restrict__ off /*[N+1]*/, float *__restrict__ Y, uint32_t *__restrict__ Y16) {
    const int lane = threadIdx.x & 63, n = lane & 31, h4 = (lane >> 5) * 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total = off[N];
    const int tile = blockIdx.x * 4 + wave;
    if (tile * 32 >= total) return;
    const int p = min(tile * 32 + n, total - 1);
    const int ck = pairs[p], c = ck >> 16, k = ck & 0xffff;
    const float dx = xyz[3 * k] - xyz[3 * c], dy = xyz[3 * k + 1] - xyz[3 * c + 1], dz = xyz[3 * k + 2] - xyz[3 * c + 2];
    f32x16 in[4], out[8];
    const float *urow = U + (size_t)k * 128;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 u = feat4(urow, o, q, h4);
            const float4 a = feat4(vx, o, q, h4), b = feat4(vx + 128, o, q, h4), d = feat4(vx + 256, o, q, h4);
            in[o][4 * q + 0] = fmaxf(fmaf(d.x, dz, fmaf(b.x, dy, fmaf(a.x, dx, u.x))), 0.f);
            in[o][4 * q + 1] = fmaxf(fmaf(d.y, dz, fmaf(b.y, dy, fmaf(a.y, dx, u.y))), 0.f);
            in[o][4 * q + 2] = fmaxf(fmaf(d.z, dz, fmaf(b.z, dy, fmaf(a.z, dx, u.z))), 0.f);
            in[o][4 * q + 3] = fmaxf(fmaf(d.w, dz, fmaf(b.w, dy, fmaf(a.w, dx, u.w))), 0.f);
        }
    }
    chain_layer<4, 8, CHAIN_BIAS>(Wimg, bias, in, out, lane);
    if (Y16) {
        // bf16 mode (the only consumer, z16_kernel's contraction, rounds to bf16 and rounding commutes with l2's max): bf16
        // operand-order rows, the lane's 8 dwords of a block contiguous, instead of the float32 rows
        if (tile * 32 + n < total) {
            uint4 *d16 = reinterpret_cast<uint4 *>(Y16 + (size_t)p * 128) + (h4 >> 1);
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                uint32_t pk[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) pk[d] = pack_bf16(fmaxf(out[o][2 * d], 0.f), fmaxf(out[o][2 * d + 1], 0.f));
                d16[4 * o] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                d16[4 * o + 1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
        }
        return;
    }
    if (tile * 32 + n < total) {
        float *dst = Y + (size_t)p * 256;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = fmaxf(out[o][4 * q + 0], 0.f); v.y = fmaxf(out[o][4 * q + 1], 0.f);
                v.z = fmaxf(out[o][4 * q + 2], 0.f); v.w = fmaxf(out[o][4 * q + 3], 0.f);
                *reinterpret_cast<float4 *>(dst + 32 * o + 8 * q + h4) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ T5
// A centre whose r=0.4 ball holds at most 64 points keeps ALL of them whatever the ordering, so its sa2 feature (and
// with it the sa3 feature Z) is the same for every variant and is computed once, in slot 0.  Only "crowded" centres
// (more than 64 in-radius points: the ball query truncates, and which 64 survive depends on the variant's order) need
// one value per variant.  crowded[c] in {0,1}; clist = the crowded centres, *ncr their number.
__global__ __launch_bounds__(1024) void crowd_kernel(const float *xyz, int N, float r2, int *crowded,
                                                     int *clist, int *ncr, int *off /*[N+1]*/,
                                                     int *ncr_copy /* optional second home of the count (host readback pool) */) {
    __shared__ int wcount[16], wsum[16];
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    // blockIdx.y: object of a batched launch (pooled per-object arrays at their natural strides; 0 for a single object)
    const size_t ob = blockIdx.y;
    xyz += ob * 3 * N; crowded += ob * N; clist += ob * (N + 1); ncr += ob * (N + 1); off += ob * (N + 1);
    if (ncr_copy) ncr_copy += ob;
    bool cr = false;
    int cnt = 0;
    if (c < N) {
        const float cx = xyz[3 * c], cy = xyz[3 * c + 1], cz = xyz[3 * c + 2];
        const float cn = sq3(cx, cy, cz);
        for (int k = 0; k < N; ++k) {
            const float x = xyz[3 * k], y = xyz[3 * k + 1], z = xyz[3 * k + 2];
            cnt += !(sqdist_expanded(cx, cy, cz, cn, x, y, z, sq3(x, y, z)) > r2);
        }
        cr = cnt > 64;
        crowded[c] = cr ? 1 : 0;
    }
    const unsigned long long m = __ballot(cr);
    // exclusive prefix sum of the in-radius counts: off[c] = first row of centre c in the pair list, off[N] = number of pairs
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    int base = 0, sbase = 0;
    for (int w = 0; w < wave; ++w) { base += wcount[w]; sbase += wsum[w]; }
    if (cr) clist[base + __popcll(m & ((1ull << lane) - 1ull))] = c;
    if (c < N) off[c] = sbase + incl - cnt;
    if (c == 0) {
        int tot = 0, stot = 0;
        for (int w = 0; w < 16; ++w) { tot += wcount[w]; stot += wsum[w]; }
        *ncr = tot;
        if (ncr_copy) *ncr_copy = tot;
        off[N] = stot;
    }
}

// Pair list and lookup for T4/T5.  One wave per centre c: rank[c][k] = position of point k among c's in-radius points in
// index order (or -1), pairs[off[c] + rank] = (c << 16 | k).
__global__ __launch_bounds__(256) void nbr_fill_kernel(const float *xyz, int N, float r2, const int *off,
                                                       int *pairs, short *rank /*[N][N]*/) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= N) return;
    const size_t ob = blockIdx.y;                      // object of a batched launch (crowd_kernel)
    xyz += ob * 3 * N; off += ob * (N + 1); pairs += ob * N * N; rank += ob * N * N;
    const float cx = xyz[3 * c], cy = xyz[3 * c + 1], cz = xyz[3 * c + 2];
    const float cn = sq3(cx, cy, cz);
    const int o = off[c];
    int seen = 0;
    for (int k0 = 0; k0 < N; k0 += 64) {
        const int k = k0 + lane;
        bool in = false;
        if (k < N) {
            const float x = xyz[3 * k], y = xyz[3 * k + 1], z = xyz[3 * k + 2];
            in = !(sqdist_expanded(cx, cy, cz, cn, x, y, z, sq3(x, y, z)) > r2);
        }
        const unsigned long long m = __ballot(in);
        const int r = seen + __popcll(m & ((1ull << lane) - 1ull));
        if (k < N) rank[(size_t)c * N + k] = in ? (short)r : (short)-1;
        if (in) pairs[o + r] = (c << 16) | k;
        seen += __popcll(m);
    }
}

// sa2's ball query for one centre by one wave: the first 64 in-radius candidates in the order `perm` lists the points
// (query_ball_point scans the re-ordered cloud, pointnet2_utils.py:95-115).  rks = the centre's row of the rank table in LDS
// (rank >= 0 <=> in the ball; the value is the candidate's row in the centre's block of the pair list).  Writes those rows to
// sel[0..cnt) and returns cnt <= 64.  Shared by l2_kernel and the index test hook.
__device__ __forceinline__ int l2_select(const short *rks, const int *__restrict__ perm, int M, int *sel, int lane) {
    int cnt = 0;
    for (int base = 0; base < M && cnt < 64; base += 64) {
        const int j = base + lane;
        const int r = j < M ? rks[perm[j]] : -1;
        const bool in = r >= 0;
        const unsigned long long m = __ballot(in);
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (in && pos < 64) sel[pos] = r;
        cnt += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    return min(cnt, 64);
}

// L2[slot][c][256] = max over the first 64 in-radius (r=0.4) positions of variant vlist[slot] of Y[(c, point)].
// mode 0: slot 0, every centre (blockIdx.y*4 + wave).  mode 1: slots >= 1, crowded centres only (clist[blockIdx.y]); the
// waves of a workgroup then share the centre so its Y slab stays cache resident.
template <bool BF16>
__global__ __launch_bounds__(256) void l2_kernel(const float *__restrict__ xyz, int N, float r2, const int *__restrict__ fps1 /*[N][512]*/,
                                                 const int *__restrict__ vlist, int nv, const float *__restrict__ Y,
                                                 float *__restrict__ L2 /*[nv][N][256]*/, int mode, const int *__restrict__ clist,
                                                 const int *__restrict__ ncr, const int *__restrict__ off, const short *__restrict__ rank) {
    __shared__ int sel[4][64];
    __shared__ short rks[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int c, slot;
    if (mode == 0) {
        c = blockIdx.y * 4 + wave;
        slot = 0;
        if (c >= N) return;
    } else {
        // 1-D grid.  Workgroups go round-robin to the 8 XCDs, so workgroup L runs on XCD L % 8: give every XCD whole centres
        // (crowded centre 8 k + xcd, all of its variant groups in a row), so that the centre's block of Y rows - which each of
        // its ~511 variants gathers 64 rows from - is served by that XCD's L2 instead of the memory-side cache.
        const int nvg = (nv - 1 + 3) / 4;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int ci = (j / nvg) * 8 + xcd;
        if (ci >= *ncr) return;
        c = clist[ci];
        slot = 1 + (j % nvg) * 4 + wave;
        if (slot >= nv) return;
    }
    const int *perm = fps1 + (size_t)vlist[slot] * 512;
    // rank[c][k] >= 0 <=> point k lies in c's ball (nbr_fill_kernel made that decision with the reference's distance formula),
    // and its value is the point's row in c's block of the pair list: the centre's row of the table, copied to LDS, replaces
    // three coordinate gathers and the distance arithmetic per candidate
    const short *rk = rank + (size_t)c * N;
    for (int i = lane; i < N; i += 64) rks[wave][i] = rk[i];
    __builtin_amdgcn_wave_barrier();
    const int cnt = l2_select(rks[wave], perm, 512, sel[wave], lane);       // sa1 always hands 512 centres on (pointnet2.py:17)
    if (BF16) {
        // Y and L2 are bf16 operand-order rows (128 dwords): two dwords per lane, v_pk_max_u16 (values are >= 0)
        const uint2 *slab = reinterpret_cast<const uint2 *>(reinterpret_cast<const uint32_t *>(Y) + (size_t)off[c] * 128) + lane;
        uint2 best = make_uint2(0u, 0u);
        int i = 0;
        for (; i + 8 <= cnt; i += 8) {                  // eight rows in flight: the gather is latency bound
            uint2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = slab[(size_t)sel[wave][i + k] * 64];
#pragma unroll
            for (int k = 0; k < 8; ++k) { best.x = pkmax_u16(best.x, v[k].x); best.y = pkmax_u16(best.y, v[k].y); }
        }
        for (; i < cnt; ++i) {
            const uint2 a = slab[(size_t)sel[wave][i] * 64];
            best.x = pkmax_u16(best.x, a.x); best.y = pkmax_u16(best.y, a.y);
        }
        reinterpret_cast<uint2 *>(reinterpret_cast<uint32_t *>(L2) + ((size_t)slot * N + c) * 128)[lane] = best;
        return;
    }
    const float *slab = Y + (size_t)off[c] * 256 + lane * 4;
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);       // Y >= 0 (ReLU); an empty ball cannot happen for a centre of the set,
                                                         // and a centre outside the variant's set is never read
    int i = 0;
    for (; i + 8 <= cnt; i += 8) {                      // eight rows in flight: the gather is latency bound
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(slab + (size_t)sel[wave][i + k] * 256);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            best.x = fmaxf(best.x, v[k].x); best.y = fmaxf(best.y, v[k].y); best.z = fmaxf(best.z, v[k].z); best.w = fmaxf(best.w, v[k].w);
        }
    }
    for (; i < cnt; ++i) {
        const float4 a = *reinterpret_cast<const float4 *>(slab + (size_t)sel[wave][i] * 256);
        best.x = fmaxf(best.x, a.x); best.y = fmaxf(best.y, a.y); best.z = fmaxf(best.z, a.z); best.w = fmaxf(best.w, a.w);
    }
    *reinterpret_cast<float4 *>(L2 + ((size_t)slot * N + c) * 256 + lane * 4) = best;
}

// T5 for the crowded centres, all variants at once.  l2_kernel's mode 1 makes every (variant, crowded centre) pair gather its 64
// rows of Y from L2: 511 variants x 64 KB per centre, ~5 GB of L2 traffic per object - the bound of the whole table build.  Here
// one workgroup owns a crowded centre: its block of Y ([K][256] floats, K = points in the ball, 65..~190 on the shipped clouds) is
// staged in LDS once (in feature chunks of F floats when K KB do not fit), the first-64 selections of all variants are computed
// once (l2_select, kept as bytes: K <= 255) and every variant's max is taken out of LDS.  Same operands, same max: bit-identical
// to l2_kernel.  Centres with K > 255 (none on 512-point clouds so far) are left to l2_kernel (kmax tells the host).
constexpr int L2C_YBYTES = 120 * 1024, L2C_THREADS = 1024;     // 16 waves per CU (one workgroup per CU by LDS): the per-variant row reads are LDS-latency-bound

template <bool BF16, int LPR>
__device__ __forceinline__ void l2c_reduce(const uint32_t *ych, const unsigned char *selv, int cnt, uint32_t *dst, int lane) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    constexpr int RPW = 64 / LPR;
    const int sg = lane / LPR, fl = lane % LPR;
    u4 best = {0u, 0u, 0u, 0u};                                   // Y >= 0 (ReLU): 0 is the identity of both max flavours
    auto vmax4 = [](u4 a, u4 b) {
        u4 o;
        if (BF16) { o.x = pkmax_u16(a.x, b.x); o.y = pkmax_u16(a.y, b.y); o.z = pkmax_u16(a.z, b.z); o.w = pkmax_u16(a.w, b.w); }
        else {
            o.x = __float_as_uint(fmaxf(__uint_as_float(a.x), __uint_as_float(b.x))); o.y = __float_as_uint(fmaxf(__uint_as_float(a.y), __uint_as_float(b.y)));
            o.z = __float_as_uint(fmaxf(__uint_as_float(a.z), __uint_as_float(b.z))); o.w = __float_as_uint(fmaxf(__uint_as_float(a.w), __uint_as_float(b.w)));
        }
        return o;
    };
    // selv is padded to 64 entries with its last member, so lane groups that run past cnt re-read a member (idempotent)
    // lane group sg takes entries j + 4 sg .. j + 4 sg + 3: one dword of four selection bytes per step
    for (int j = 0; j < cnt; j += 4 * RPW) {
        const uint32_t four = *reinterpret_cast<const uint32_t *>(selv + min(j + 4 * sg, 60));
        u4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const u4 *>(ych + ((size_t)((four >> (8 * e)) & 255u) * LPR + fl) * 4);
        if (BF16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) best = vmax4(best, v[e]);
        } else {
            // Y >= 0: the unsigned maximum is the float maximum, and three operands go into one v_max3_u32 (bit-identical to the fmaxf chain)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                best.x = max(max(best.x, v[e].x), v[e + 1].x); best.y = max(max(best.y, v[e].y), v[e + 1].y);
                best.z = max(max(best.z, v[e].z), v[e + 1].z); best.w = max(max(best.w, v[e].w), v[e + 1].w);
            }
        }
    }
    if (RPW > 1) {
#pragma unroll
        for (int o = 32; o >= LPR; o >>= 1) {
            u4 t;
            t.x = __shfl_xor(best.x, o); t.y = __shfl_xor(best.y, o); t.z = __shfl_xor(best.z, o); t.w = __shfl_xor(best.w, o);
            best = vmax4(best, t);
        }
    }
    if (sg == 0) *reinterpret_cast<u4 *>(dst + fl * 4) = best;
}

// one crowded centre c, the variants of share blockIdx.y
template <bool BF16>
__device__ __forceinline__ void l2c_centre(int c, int N, const int *__restrict__ fps1 /*[N][512]*/, int nv, const uint32_t *__restrict__ Y,
                                           uint32_t *__restrict__ L2 /*[nv][N][W]*/, const int *__restrict__ off, const short *__restrict__ rank,
                                           int (*selw)[64]) {
    constexpr int W = BF16 ? 128 : 256;                           // dwords per row of Y / L2
    extern __shared__ uint32_t l2c_lds[];
    short *rks = reinterpret_cast<short *>(l2c_lds);              // [1024]
    unsigned char *sel = reinterpret_cast<unsigned char *>(l2c_lds + 512);        // [512][64]
    unsigned char *cnts = sel + 512 * 64;                         // [512]
    uint32_t *ych = l2c_lds + 512 + 512 * 16 + 128;               // [K][F]
    const int K = off[c + 1] - off[c];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = L2C_THREADS / 64;
    // blockIdx.y: this workgroup's share of the variants (an object has ~40 crowded centres: one workgroup per centre leaves five CUs
    // of six idle, and the reduction is bound by the LDS bandwidth of the CU it runs on; staging the block of Y once per share is cheap)
    const int vlo = 1 + (int)(((int64_t)(nv - 1) * blockIdx.y) / gridDim.y), vhi = 1 + (int)(((int64_t)(nv - 1) * (blockIdx.y + 1)) / gridDim.y);
#ifdef DGDM_L2C_CLOCKS
    long long tk[8]; int nk = 0;
#define L2C_STAMP() do { if (nk < 8) tk[nk++] = __builtin_readcyclecounter(); } while (0)
#else
#define L2C_STAMP() do {} while (0)
#endif
    L2C_STAMP();
    for (int i = threadIdx.x; i < N; i += L2C_THREADS) rks[i] = rank[(size_t)c * N + i];
    __syncthreads();
    L2C_STAMP();
    if (K > 255) {
        // a ball with more points than a byte can index (never on the shipped 512-point clouds): gather from global memory as
        // l2_kernel does, W / 64 dwords per lane
        constexpr int D = W / 64;
        for (int v = vlo + wave; v < vhi; v += nwave) {
            const int cnt = l2_select(rks, fps1 + (size_t)v * 512, 512, selw[wave], lane);
            uint32_t best[D];
#pragma unroll
            for (int d = 0; d < D; ++d) best[d] = 0u;
            for (int i = 0; i < cnt; ++i) {
                const uint32_t *row = Y + ((size_t)off[c] + selw[wave][i]) * W + lane * D;
#pragma unroll
                for (int d = 0; d < D; ++d)
                    best[d] = BF16 ? pkmax_u16(best[d], row[d]) : __float_as_uint(fmaxf(__uint_as_float(best[d]), __uint_as_float(row[d])));
            }
            uint32_t *dst = L2 + ((size_t)v * N + c) * W + lane * D;
#pragma unroll
            for (int d = 0; d < D; ++d) dst[d] = best[d];
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // ---- the first-64 selection of every variant (slot v reads the cloud in the order fps1[v])
    // (the 512 candidates of the next variant are loaded while the current one is scanned: alone in its CU's LDS, the workgroup
    //  has only its own 8 waves to hide the L2 latency of those loads)
    {
        int pv[8], pn[8];
        const int v0 = vlo + wave;
#pragma unroll
        for (int i = 0; i < 8; ++i) pn[i] = v0 < vhi ? fps1[(size_t)v0 * 512 + 64 * i + lane] : 0;
        for (int v = v0; v < vhi; v += nwave) {
#pragma unroll
            for (int i = 0; i < 8; ++i) pv[i] = pn[i];
            const int vn = v + nwave;
            if (vn < vhi) {
#pragma unroll
                for (int i = 0; i < 8; ++i) pn[i] = fps1[(size_t)vn * 512 + 64 * i + lane];
            }
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {                                 // l2_select on the registers
                if (cnt >= 64) break;
                const int r = rks[pv[i]];
                const bool in = r >= 0;
                const unsigned long long m = __ballot(in);
                const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                if (in && pos < 64) selw[wave][pos] = r;
                cnt += __popcll(m);
            }
            cnt = min(cnt, 64);
            __builtin_amdgcn_wave_barrier();
            sel[(size_t)v * 64 + lane] = (unsigned char)selw[wave][lane < cnt ? lane : max(cnt - 1, 0)];   // a crowded centre has cnt = 64
            if (lane == 0) cnts[v] = (unsigned char)cnt;
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- feature chunks of F dwords: F = 4 * LPR, the widest that holds K rows in the Y area
    int lpr = W / 4;
    while ((size_t)K * lpr * 16 > (size_t)L2C_YBYTES) lpr >>= 1;
    const uint32_t *Yc = Y + (size_t)off[c] * W;
    for (int f0 = 0; f0 < W; f0 += 4 * lpr) {
        __syncthreads();                                          // selections written / previous chunk consumed
        L2C_STAMP();
        const int pieces = K * lpr;
        for (int i = threadIdx.x; i < pieces; i += L2C_THREADS)
            *reinterpret_cast<uint4 *>(ych + (size_t)i * 4) = *reinterpret_cast<const uint4 *>(Yc + (size_t)(i / lpr) * W + f0 + (i % lpr) * 4);
        __syncthreads();
        L2C_STAMP();
        for (int v = vlo + wave; v < vhi; v += nwave) {
            uint32_t *dst = L2 + ((size_t)v * N + c) * W + f0;
            const unsigned char *sv = sel + (size_t)v * 64;
            const int cnt = cnts[v];
            switch (lpr) {
                case 64: if (!BF16) l2c_reduce<BF16, (BF16 ? 32 : 64)>(ych, sv, cnt, dst, lane); break;
                case 32: l2c_reduce<BF16, 32>(ych, sv, cnt, dst, lane); break;
                case 16: l2c_reduce<BF16, 16>(ych, sv, cnt, dst, lane); break;
                default: l2c_reduce<BF16, 8>(ych, sv, cnt, dst, lane); break;
            }
        }
    }
#ifdef DGDM_L2C_CLOCKS
    __syncthreads();
    L2C_STAMP();
    if (threadIdx.x == 0 && blockIdx.x == 3 && blockIdx.y == 0) {
        printf("l2c c %d K %d lpr %d share %d..%d:", c, K, lpr, vlo, vhi);
        for (int i = 1; i < nk; ++i) printf(" %lld", tk[i] - tk[i - 1]);
        printf("\n");
    }
#endif
}

// Grid (centres' stride, variant shares): a workgroup takes the crowded centres blockIdx.x, blockIdx.x + gridDim.x, ... and of each the
// variants of share blockIdx.y.  The number of crowded centres is device data (0 .. N): the host launches 64 x 4 workgroups - one per
// CU, ONE round for any count - instead of one workgroup per possible centre (N x shares workgroups of 16 waves and 150 KB of LDS,
// each of which needs a whole free CU to find out that it has nothing to do: 3072 of them cost more than the 240 that had work).
template <bool BF16>
__global__ __launch_bounds__(L2C_THREADS, 1) void l2c_kernel(int N, const int *__restrict__ fps1 /*[N][512]*/, int nv, const uint32_t *__restrict__ Y,
                                                             uint32_t *__restrict__ L2 /*[nv][N][W]*/, const int *__restrict__ clist,
                                                             const int *__restrict__ ncr, const int *__restrict__ off,
                                                             const short *__restrict__ rank) {
    __shared__ int selw[L2C_THREADS / 64][64];
    const int n = *ncr;
#ifdef DGDM_L2C_CLOCKS
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) printf("l2c ncr %d pairs %d\n", n, off[N]);
#endif
    for (int ci = blockIdx.x; ci < n; ci += gridDim.x) {
        l2c_centre<BF16>(clist[ci], N, fps1, nv, Y, L2, off, rank, selw);
        __syncthreads();                                          // the LDS areas are reused by the next centre
    }
}

// ------------------------------------------------------------------------------------------------ T6
// Z[row][256] = ReLU(W3'[:,3:] L2[row] + W3'[:,0:3] xyz_c + b3'),  row = slot*N + c   (sa3, pointnet2.py:19)
// mode 0: slot 0, all N centres.  mode 1: slots 1..nv-1, crowded centres only (work item k -> slot 1 + k / ncr, centre clist[k % ncr]).
__global__ __launch_bounds__(256, 1) void z_kernel(const float *__restrict__ xyz, int N, int nv, const float *__restrict__ L2,
                                                   const float4 *__restrict__ Wimg, const float *__restrict__ w3x /*[3][256]*/,
                                                   const float *__restrict__ bias, float *__restrict__ Z, uint32_t *__restrict__ Z16,
                                                   int mode, const int *__restrict__ clist, const int *__restrict__ ncr) {
    const int lane = threadIdx.x & 63, n = lane & 31, h4 = (lane >> 5) * 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int ncrv = mode ? *ncr : N;
    const int64_t items = mode ? (int64_t)(nv - 1) * ncrv : N;
    if (tile * 32 >= items) return;
    const int64_t item = min(tile * 32 + n, items - 1);
    const int c = mode ? clist[item % ncrv] : (int)item;
    const int64_t row = mode ? (1 + item / ncrv) * N + c : c;
    const int64_t rows = items;
    const float x = xyz[3 * c], y = xyz[3 * c + 1], z = xyz[3 * c + 2];
    f32x16 in[8], out[8];
    const float *lrow = L2 + (size_t)row * 256;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = feat4(lrow, o, q, h4);
            in[o][4 * q + 0] = v.x; in[o][4 * q + 1] = v.y; in[o][4 * q + 2] = v.z; in[o][4 * q + 3] = v.w;
            const float4 b = feat4(bias, o, q, h4);
            const float4 a0 = feat4(w3x, o, q, h4), a1 = feat4(w3x + 256, o, q, h4), a2 = feat4(w3x + 512, o, q, h4);
            out[o][4 * q + 0] = fmaf(a2.x, z, fmaf(a1.x, y, fmaf(a0.x, x, b.x)));
            out[o][4 * q + 1] = fmaf(a2.y, z, fmaf(a1.y, y, fmaf(a0.y, x, b.y)));
            out[o][4 * q + 2] = fmaf(a2.z, z, fmaf(a1.z, y, fmaf(a0.z, x, b.z)));
            out[o][4 * q + 3] = fmaf(a2.w, z, fmaf(a1.w, y, fmaf(a0.w, x, b.w)));
        }
    }
    chain_layer<8, 8, CHAIN_KEEP>(Wimg, nullptr, in, out, lane);
    if (tile * 32 + n < rows) {
        float *dst = Z + (size_t)row * 256;   // rows of one tile are distinct (item -> row is injective)
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = fmaxf(out[o][4 * q + 0], 0.f); v.y = fmaxf(out[o][4 * q + 1], 0.f);
                v.z = fmaxf(out[o][4 * q + 2], 0.f); v.w = fmaxf(out[o][4 * q + 3], 0.f);
                *reinterpret_cast<float4 *>(dst + 32 * o + 8 * q + h4) = v;
            }
        }
        if (Z16) {
            // the same row in bf16 operand order for the bf16 trunk (mfma_chain.h): the lane's 8 dwords of block o are contiguous
            uint4 *d16 = reinterpret_cast<uint4 *>(Z16 + (size_t)row * 128) + (h4 >> 1);
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                uint32_t pk[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) pk[d] = pack_bf16(fmaxf(out[o][2 * d], 0.f), fmaxf(out[o][2 * d + 1], 0.f));
                d16[4 * o] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                d16[4 * o + 1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
        }
    }
}

// T6 in bf16 mode: the sa3 contraction on v_mfma_f32_32x32x16_bf16.  One wave = two 32-row tiles (each weight entry feeds two
// MFMAs).  L2_16 rows are already the B operand (operand order); W3'[:, 3:] comes as a pack_chain_bf16 image through the
// buffer-load ring; the coordinate part W3'[:, 0:3] xyz_c + b3' is the float32 accumulator start.  Writes the float32 rows
// (orientation sweep, M0) and their bf16 operand-order copy (xobj gathers).  Items as in z_kernel.
typedef __bf16 zbf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256, 1) void z16_kernel(const float *__restrict__ xyz, int N, int nv, const uint32_t *__restrict__ L2_16,
                                                     const float4 *__restrict__ Wimg16, const float *__restrict__ w3x /*[3][256]*/,
                                                     const float *__restrict__ bias, float *__restrict__ Z, uint32_t *__restrict__ Z16,
                                                     int mode, const int *__restrict__ clist, const int *__restrict__ ncr) {
    const int lane = threadIdx.x & 63, n = lane & 31, h4 = (lane >> 5) * 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * 2;
    const int ncrv = mode ? *ncr : N;
    const int64_t items = mode ? (int64_t)(nv - 1) * ncrv : N;
    if (tile0 * 32 >= items) return;
    int64_t row[2];
    float cx[2], cy[2], cz[2];
    bool live[2];
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 in[2][8][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t idx = (tile0 + t) * 32 + n;
        live[t] = idx < items;
        const int64_t item = min(idx, items - 1);
        const int c = mode ? clist[item % ncrv] : (int)item;
        row[t] = mode ? (1 + item / ncrv) * N + c : c;
        cx[t] = xyz[3 * c]; cy[t] = xyz[3 * c + 1]; cz[t] = xyz[3 * c + 2];
        const u4 *src = reinterpret_cast<const u4 *>(L2_16 + (size_t)row[t] * 128) + (h4 >> 1);
#pragma unroll
        for (int o = 0; o < 8; ++o) { in[t][o][0] = src[4 * o]; in[t][o][1] = src[4 * o + 1]; }
    }
    const wrsrc_t rs = weight_rsrc(Wimg16, 128 * 1024);
    float4 ring[CONT_DEPTH];
    ring_fill(rs, lane * 16, 0, ring);
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = feat4(bias, o, q, h4);
            const float4 a0 = feat4(w3x, o, q, h4), a1 = feat4(w3x + 256, o, q, h4), a2 = feat4(w3x + 512, o, q, h4);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[t][4 * q + 0] = fmaf(a2.x, cz[t], fmaf(a1.x, cy[t], fmaf(a0.x, cx[t], b.x)));
                acc[t][4 * q + 1] = fmaf(a2.y, cz[t], fmaf(a1.y, cy[t], fmaf(a0.y, cx[t], b.y)));
                acc[t][4 * q + 2] = fmaf(a2.z, cz[t], fmaf(a1.z, cy[t], fmaf(a0.z, cx[t], b.z)));
                acc[t][4 * q + 3] = fmaf(a2.w, cz[t], fmaf(a1.w, cy[t], fmaf(a0.w, cx[t], b.w)));
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = o * 16 + i;
            const float4 a = ring[e % CONT_DEPTH];
            ring[e % CONT_DEPTH] = wload(rs, lane * 16, (e + CONT_DEPTH) * 1024);       // past the image: clipped to zero, unused
            const zbf16x8 av = __builtin_bit_cast(zbf16x8, a);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(zbf16x8, in[0][i / 2][i % 2]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(zbf16x8, in[1][i / 2][i % 2]), acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!live[t]) continue;                    // rows of one tile are distinct (item -> row is injective)
            float *dst = Z + (size_t)row[t] * 256;
            uint32_t pk[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = fmaxf(acc[t][4 * q + 0], 0.f); v.y = fmaxf(acc[t][4 * q + 1], 0.f);
                v.z = fmaxf(acc[t][4 * q + 2], 0.f); v.w = fmaxf(acc[t][4 * q + 3], 0.f);
                *reinterpret_cast<float4 *>(dst + 32 * o + 8 * q + h4) = v;
                pk[2 * q] = pack_bf16(v.x, v.y); pk[2 * q + 1] = pack_bf16(v.z, v.w);
            }
            uint4 *d16 = reinterpret_cast<uint4 *>(Z16 + (size_t)row[t] * 128) + (h4 >> 1);
            d16[4 * o] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            d16[4 * o + 1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ T7
// Non-crowded centres contribute the same Z row (slot 0) to every reference row whose sa2 FPS starts at point q, so their
// part of the sa3 max is a per-object table:  M0[q][256] = max over the non-crowded centres c of fps2[q] of Z[0][c].
// cl2[q][0..cnt2[q]) lists the crowded centres of fps2[q] (point ids), padded to 128 with its last entry so the per-row
// loop can run in groups of four without a tail.  One wave per start point q.  max is exact, so the split is bit-neutral.
__global__ __launch_bounds__(256) void m0_kernel(const int *__restrict__ fps2 /*[N][128]*/, const int *__restrict__ crowded, int N,
                                                 const float *__restrict__ Z0 /*[N][256]*/, float *__restrict__ M0, int *__restrict__ cl2,
                                                 int *__restrict__ cnt2, const uint32_t *__restrict__ Z0_16, uint32_t *__restrict__ M0_16,
                                                 const int *__restrict__ clist, const int *__restrict__ ncr, int *__restrict__ cl2s,
                                                 unsigned short *__restrict__ cl2o, int bf16) {
    // position of every crowded centre in clist (ascending point ids, crowd_kernel): cl2s = cl2 in those terms
    __shared__ short slot_of[1024];
    for (int i = threadIdx.x; i < *ncr; i += blockDim.x) slot_of[clist[i]] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= N) return;
    const int idA = fps2[(size_t)q * 128 + lane], idB = fps2[(size_t)q * 128 + 64 + lane];
    const bool crA = crowded[idA] != 0, crB = crowded[idB] != 0;
    const unsigned long long mA = __ballot(crA), mB = __ballot(crB);
    const int nA = __popcll(mA), cnt = nA + __popcll(mB);
    int *dst = cl2 + (size_t)q * 128;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (crA) dst[__popcll(mA & below)] = idA;
    if (crB) dst[nA + __popcll(mB & below)] = idB;
    // pad with the last crowded id (which lane holds it: the highest set bit of mB, else of mA)
    int last = 0;
    if (cnt > 0) last = mB ? __shfl(idB, 63 - __clzll(mB)) : __shfl(idA, 63 - __clzll(mA));
    for (int j = cnt + lane; j < 128; j += 64) dst[j] = last;
    if (lane == 0) cnt2[q] = cnt;
    {
        int *ds = cl2s + (size_t)q * 128;
        if (crA) ds[__popcll(mA & below)] = slot_of[idA];
        if (crB) ds[nA + __popcll(mB & below)] = slot_of[idB];
        const int ls = cnt > 0 ? slot_of[last] : 0;
        for (int j = cnt + lane; j < 128; j += 64) ds[j] = ls;
        // the same list as byte offsets into xobj_rows_kernel's LDS slab (slot * lanes-per-row * 16 B; 0 when the slab does not fit), in TWO
        // orders (max does not care): members with an even slot first / with an odd slot first.  With 128-byte slab entries (lpr = 8:
        // objects with more than 256 crowded centres) a slot's parity is the half of the 64 banks its entry lies in, and a ds_read_b128's
        // 16-lane service group holds quarter-entries of two lane groups (0 & 3, 1 & 2 of every four): lane groups 0, 1 walk the
        // even-first list and 2, 3 the odd-first one, so the two meet in the same bank half only where the halves of their lists overlap
        // (the unsorted list: a 2-way conflict on every other read, +44 % LDS cycles in profiles/r06's first collection)
        const int sc = xobj_rows_lpr(*ncr, bf16 != 0) * 16;
        const int sA = crA ? slot_of[idA] : 0, sB = crB ? slot_of[idB] : 0;
        const unsigned long long mAe = __ballot(crA && !(sA & 1)), mAo = __ballot(crA && (sA & 1)), mBe = __ballot(crB && !(sB & 1)), mBo = __ballot(crB && (sB & 1));
        const int ne = __popcll(mAe) + __popcll(mBe), no = cnt - ne;
        unsigned short *dz = cl2o + (size_t)q * 256;          // [2][128]
        if (crA) {
            const int pe = (sA & 1) ? ne + __popcll(mAo & below) : __popcll(mAe & below);                       // even-first position
            dz[pe] = (unsigned short)(sA * sc);
            dz[128 + ((sA & 1) ? pe - ne : no + pe)] = (unsigned short)(sA * sc);                                // odd-first position
        }
        if (crB) {
            const int pe = (sB & 1) ? ne + __popcll(mAo) + __popcll(mBo & below) : __popcll(mAe) + __popcll(mBe & below);
            dz[pe] = (unsigned short)(sB * sc);
            dz[128 + ((sB & 1) ? pe - ne : no + pe)] = (unsigned short)(sB * sc);
        }
        for (int j = cnt + lane; j < 128; j += 64) { dz[j] = (unsigned short)(ls * sc); dz[128 + j] = (unsigned short)(ls * sc); }
    }
    // the non-crowded centres as a compact list (max is order-independent), padded to a multiple of eight with its first entry: eight
    // row loads in flight per wave instead of one behind a branch (the loop was a chain of 128 L2 round trips: 92 us per object)
    __shared__ int nlist[4][136];
    int *nl = nlist[threadIdx.x >> 6];
    const int nnA = 64 - nA, nn = 128 - cnt;
    if (!crA) nl[__popcll(~mA & below)] = idA;
    if (!crB) nl[nnA + __popcll(~mB & below)] = idB;
    __builtin_amdgcn_wave_barrier();
    if (nn > 0 && lane < 8) nl[nn + lane] = nl[0];
    __builtin_amdgcn_wave_barrier();
    const float *zt = Z0 + lane * 4;
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);      // Z >= 0 (ReLU)
    for (int i = 0; i < nn; i += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(zt + (size_t)nl[i + k] * 256);
#pragma unroll
        for (int k = 0; k < 8; ++k) { best.x = fmaxf(best.x, v[k].x); best.y = fmaxf(best.y, v[k].y); best.z = fmaxf(best.z, v[k].z); best.w = fmaxf(best.w, v[k].w); }
    }
    *reinterpret_cast<float4 *>(M0 + (size_t)q * 256 + lane * 4) = best;
    if (M0_16) {                                        // the same table over the bf16 rows (two dwords per lane)
        uint2 b16 = make_uint2(0u, 0u);
        for (int i = 0; i < nn; i += 8) {
            uint2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = reinterpret_cast<const uint2 *>(Z0_16 + (size_t)nl[i + k] * 128)[lane];
#pragma unroll
            for (int k = 0; k < 8; ++k) { b16.x = pkmax_u16(b16.x, v[k].x); b16.y = pkmax_u16(b16.y, v[k].y); }
        }
        reinterpret_cast<uint2 *>(M0_16 + (size_t)q * 128)[lane] = b16;
    }
}

// ------------------------------------------------------------------------------------------------ per row
// xobj[row][256] = max over the 128 centres FPS picks (start s2) on the cloud re-ordered by variant s1 of
// Z[slot(s1)][point].  One wave per row.   (sa2's FPS + sa3's max, pointnet2_utils.py:132,208)
// The centres come from the per-object table fps2[start point] whenever that sequence is order-independent.
// Work item -> row: workgroups are dealt round-robin to the 8 XCDs, so block b takes the (b/8)-th block of XCD (b%8)'s
// contiguous share of the (chain, s1)-sorted rows; rows that gather from the same Z[variant] slab then meet in one L2.
__device__ __forceinline__ int64_t xcd_contiguous(int64_t block, int64_t nblocks) {
    const int64_t q = nblocks / 8, rem = nblocks % 8, x = block % 8, i = block / 8;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + i;
}

// One row of a feature table per wave-load: float32 rows are 1 KiB (a float4 per lane), bf16 operand-order rows 512 B (a uint2).
template <bool BF16> struct RowOps;
template <> struct RowOps<false> {
    typedef float4 V;
    static __device__ __forceinline__ V zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }       // features are >= 0 (ReLU)
    static __device__ __forceinline__ V load(const XobjChain &ch, bool m0, size_t row, int lane) {
        return reinterpret_cast<const float4 *>((m0 ? ch.M0 : ch.Z) + row * 256)[lane];
    }
    static __device__ __forceinline__ V vmax(V a, V b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }
    static __device__ __forceinline__ void store(const XobjParams &p, size_t row, int lane, V v) { reinterpret_cast<float4 *>(p.xobj + row * 256)[lane] = v; }
};
template <> struct RowOps<true> {
    typedef uint2 V;
    static __device__ __forceinline__ V zero() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ V load(const XobjChain &ch, bool m0, size_t row, int lane) {
        return reinterpret_cast<const uint2 *>((m0 ? ch.M0_16 : ch.Z16) + row * 128)[lane];
    }
    static __device__ __forceinline__ V vmax(V a, V b) { return make_uint2(pkmax_u16(a.x, b.x), pkmax_u16(a.y, b.y)); }
    static __device__ __forceinline__ void store(const XobjParams &p, size_t row, int lane, V v) { reinterpret_cast<uint2 *>(p.xobj16 + row * 128)[lane] = v; }
};

// The common case on its own: rows whose 128-centre sequence comes from the fps2 table, reduced to M0[q] plus the crowded
// centres.  No LDS (the per-row FPS of xobj_kernel needs 26 KB per workgroup, which caps it at 24 waves per CU) and eight
// gathers in flight per wave instead of four: the kernel is bound by L2 latency.  BF16: gathers from the bf16 copies of Z / M0
// and writes bf16 operand-order rows for the bf16 trunk - half the bytes, bit-identical after the trunk's own rounding.
template <bool BF16>
__global__ __launch_bounds__(256) void xobj_fast_kernel(const XobjParams p) {
    typedef RowOps<BF16> R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t w = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wave;
    if (w >= p.total_rows) return;
    if (p.order) w = (w / p.R) * p.R + (p.order[w] & 0x3fffff);      // row id (its s2 sits in bits 22..)
    const int chain = (int)(w / p.R);
    const int64_t r = w - (int64_t)chain * p.R;
    const XobjChain ch = p.chains[chain];
    auto leave = [&]() {                               // one lane records the row for xobj_kernel (rare: tie-flagged start points)
        if (lane == 0) p.todo[atomicAdd(p.todo_count, 1)] = w;
    };
    if (!ch.M0 || !ch.fps2) { leave(); return; }
    const int *st = p.starts + (size_t)chain * 2 * p.R;
    const int s1 = __builtin_amdgcn_readfirstlane(st[2 * r]), s2 = __builtin_amdgcn_readfirstlane(st[2 * r + 1]);
    const int slot = ch.slot_of_start ? ch.slot_of_start[s1] : s1;
    const int q = __builtin_amdgcn_readfirstlane(ch.fps1[(size_t)s1 * 512 + s2]);      // start point of sa2's FPS
    if (ch.flags[q] != 0) { leave(); return; }                                          // xobj_kernel's row
    const int cnt = __builtin_amdgcn_readfirstlane(ch.cnt2[q]);
    const int rowA = slot * ch.N + ch.cl2[(size_t)q * 128 + lane], rowB = slot * ch.N + ch.cl2[(size_t)q * 128 + 64 + lane];
    typename R::V best = R::load(ch, true, (size_t)q, lane);
    for (int i = 0; i < cnt; i += 8) {               // groups of eight never straddle lane 63|64; the list is padded to 128
        const int src = i < 64 ? rowA : rowB, j = i & 63;
        typename R::V v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = R::load(ch, false, (size_t)__builtin_amdgcn_readlane(src, j + k), lane);
#pragma unroll
        for (int k = 0; k < 8; ++k) best = R::vmax(best, v[k]);
    }
    R::store(p, (size_t)w, lane, best);
}

template <bool BF16>
__device__ __forceinline__ void xobj_row(const XobjParams &p, int64_t w, float (*coords)[3][512], int (*centres)[128], int lane, int wave) {
    typedef RowOps<BF16> R;
    const int chain = (int)(w / p.R);
    const int64_t r = w - (int64_t)chain * p.R;
    const XobjChain ch = p.chains[chain];
    const int *st = p.starts + (size_t)chain * 2 * p.R;
    const int s1 = __builtin_amdgcn_readfirstlane(st[2 * r]), s2 = __builtin_amdgcn_readfirstlane(st[2 * r + 1]);
    const int slot = ch.slot_of_start ? ch.slot_of_start[s1] : s1;
    const int *perm = ch.fps1 + (size_t)s1 * 512;
    int idA, idB;                                       // the 128 centre POINT ids, two per lane
    const int q = __builtin_amdgcn_readfirstlane(perm[s2]);      // start point of sa2's FPS
    if (p.use_table && ch.fps2 && ch.flags[q] == 0) {
        // the FPS(128) sequence from start point q is the same for every ordering of the cloud (fps_wave TIES clear)
        if (ch.M0) {
            // non-crowded centres are already folded into M0[q]; only the crowded ones differ per variant
            const int cnt = __builtin_amdgcn_readfirstlane(ch.cnt2[q]);
            const int rowA = slot * ch.N + ch.cl2[(size_t)q * 128 + lane], rowB = slot * ch.N + ch.cl2[(size_t)q * 128 + 64 + lane];
            typename R::V best = R::load(ch, true, (size_t)q, lane);
            for (int i = 0; i < cnt; i += 4) {           // groups of four never straddle lane 63|64; the list is padded
                const int src = i < 64 ? rowA : rowB, j = i & 63;
                const int c0 = __shfl(src, j), c1 = __shfl(src, j + 1), c2 = __shfl(src, j + 2), c3 = __shfl(src, j + 3);
                const typename R::V a = R::load(ch, false, (size_t)c0, lane), b = R::load(ch, false, (size_t)c1, lane);
                const typename R::V d = R::load(ch, false, (size_t)c2, lane), e = R::load(ch, false, (size_t)c3, lane);
                best = R::vmax(R::vmax(best, R::vmax(a, b)), R::vmax(d, e));
            }
            R::store(p, (size_t)w, lane, best);
            return;
        }
        idA = ch.fps2[(size_t)q * 128 + lane];
        idB = ch.fps2[(size_t)q * 128 + 64 + lane];
    } else {
        // order-dependent selection (exact distance tie) or no table: run FPS on the cloud as re-ordered by variant s1
        float *lx = coords[wave][0], *ly = coords[wave][1], *lz = coords[wave][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = lane + 64 * i, pid = perm[j];
            lx[j] = ch.xyz[3 * pid]; ly[j] = ch.xyz[3 * pid + 1]; lz[j] = ch.xyz[3 * pid + 2];
        }
        __builtin_amdgcn_wave_barrier();
        fps_wave<8, false>(lx, ly, lz, 512, s2, 128, centres[wave], lane);
        __builtin_amdgcn_wave_barrier();
        idA = perm[centres[wave][lane]];
        idB = perm[centres[wave][64 + lane]];
    }
    // row of Z for a centre: its own variant's slot when the centre is crowded, slot 0 otherwise (see crowd_kernel)
    const int rowA = (ch.crowded[idA] ? slot * ch.N : 0) + idA, rowB = (ch.crowded[idB] ? slot * ch.N : 0) + idB;
    typename R::V best = R::zero();
#pragma unroll 4
    for (int i = 0; i < 64; i += 2) {
        const int c0 = __shfl(rowA, i), c1 = __shfl(rowA, i + 1), c2 = __shfl(rowB, i), c3 = __shfl(rowB, i + 1);
        const typename R::V a = R::load(ch, false, (size_t)c0, lane), b = R::load(ch, false, (size_t)c1, lane);
        const typename R::V d = R::load(ch, false, (size_t)c2, lane), e = R::load(ch, false, (size_t)c3, lane);
        best = R::vmax(R::vmax(best, R::vmax(a, b)), R::vmax(d, e));
    }
    R::store(p, (size_t)w, lane, best);
}

template <bool BF16>
__global__ __launch_bounds__(256) void xobj_kernel(const XobjParams p) {
    __shared__ float coords[4][3][512];
    __shared__ int centres[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (p.skip_fast) {                                  // only the rows the table kernels left out (few: a bounded grid walks the list)
        const int n = *p.todo_count;
        for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) xobj_row<BF16>(p, p.todo[i], coords, centres, lane, wave);
        return;
    }
    int64_t w = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wave;
    if (w >= p.total_rows) return;
    if (p.order) w = (w / p.R) * p.R + (p.order[w] & 0x3fffff);      // row id (its s2 sits in bits 22..)
    xobj_row<BF16>(p, w, coords, centres, lane, wave);
}

// ------------------------------------------------------------------------------------------------ per (chain, s1) group
// All rows of a chain that drew the same sa1 start s1 reduce rows of the SAME slab Z[s1][crowded centres], and rows that also drew
// the same s2 are the same row (q = fps1[s1][s2] is all a row's embedding depends on).  One workgroup per (chain, s1): the host
// sorts a chain's rows by (s1, s2), so equal rows are neighbours; the group's rows are cut into runs of equal s2 once, then for
// each feature chunk (lpr * 16 B per centre, 1 .. 8 of them) the slab chunk [ncr][F] goes from HBM/L2 to LDS (<= 64 KiB: two
// workgroups per CU) and every lane group of LPR lanes (16 B per lane = one row's chunk) walks the slot list of ITS run's row on
// its own: 64 / LPR rows in flight per wave, no cross-lane traffic, the result stored to every row of the run.  The slot list
// cl2o[q] holds slab byte offsets (16 bits, eight per 16-byte load); a slab read is one `ds_read_b128` per lane group and member,
// two members per `v_max3_u32` (features are >= 0, so the unsigned maximum IS the float maximum: bit-identical to
// xobj_fast_kernel's fmaxf).  Row metadata comes from two loads: the sorted row list carries s2 beside the row id, and
// pcf[s1][s2] = (start point q, list length, tie flag) is a per-object table.
// What bounds it: 4 LDS cycles and ~3.5 vector instructions per member KiB and CU; the slab reads from HBM (every Z row of a
// crowded centre once per launch); a workgroup's set-up round trips.
// a pointer that came out of a struct in memory is a generic one to the compiler (flat loads, which also count as LDS traffic in
// the wait counters): say that it points to global memory
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T *as_global(T *p) {
    return (__attribute__((address_space(1))) T *)p;
}

constexpr int XR_WAVES = 8;
constexpr int XR_PASS = 1024;                                  // rows of a group taken per pass (a group of the 4-call bench launch holds ~280)
constexpr int XR_SLAB_BYTES = 64 * 1024;
constexpr int XR_LDS_BYTES = XR_SLAB_BYTES + XR_PASS * 10 + 16; // slab + per row: id (int), start point, list length | flags, run heads (u16 each) + their count

// pcf[s1][s2] = q | cnt2[q] << 10 | (flags[q] != 0) << 18 with q = fps1[s1][s2]: what xobj_rows_kernel needs to know about a row
__global__ __launch_bounds__(256) void pcf_kernel(const int *__restrict__ fps1, const int *__restrict__ cnt2, const int *__restrict__ flags, int N,
                                                  int *__restrict__ pcf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * 512) return;
    const int q = fps1[i];
    pcf[i] = q | (cnt2[q] << 10) | (flags[q] ? 1 << 18 : 0);
}

template <bool BF16, int LPR>
__device__ __forceinline__ void xobj_rows_body(const XobjParams &p, const XobjChain &ch, int chain, int s1, unsigned char *lds) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u4 *gu4;
    constexpr int W = BF16 ? 128 : 256, F = 4 * LPR, RPW = 64 / LPR, NG = XR_WAVES * RPW, NT = 64 * XR_WAVES, RPT = XR_PASS / NT, NCHUNK = W / F;
    int *rid = reinterpret_cast<int *>(lds + XR_SLAB_BYTES);                     // [XR_PASS] row ids of the pass
    unsigned short *qs = reinterpret_cast<unsigned short *>(rid + XR_PASS);      // [XR_PASS] their sa2 start points q = fps1[s1][s2]
    unsigned short *cfs = qs + XR_PASS, *lead = cfs + XR_PASS;                   // [XR_PASS] cnt2[q] | flags[q] << 8 | run head << 9;  [XR_PASS] first rows of the runs
    int *nlead = reinterpret_cast<int *>(lead + XR_PASS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *goff = p.group_off + (size_t)chain * (ch.N + 1);
    const int g0 = goff[s1], gn = goff[s1 + 1] - g0;
    if (gn == 0) return;                                               // no row of this chain drew s1 (workgroup-uniform)
    const auto *Zg = as_global((BF16 ? ch.Z16 : reinterpret_cast<const uint32_t *>(ch.Z)) + ((size_t)(ch.slot_of_start ? ch.slot_of_start[s1] : s1) * ch.N) * W);
    const auto *clist = as_global(ch.clist);
    const auto *ord = as_global(p.order + (size_t)chain * p.R + g0);
    const int sg = lane / LPR, fl = lane % LPR;
    const auto *M0 = as_global((BF16 ? ch.M0_16 : reinterpret_cast<const uint32_t *>(ch.M0)) + fl * 4);
    auto *out = as_global((BF16 ? p.xobj16 : reinterpret_cast<uint32_t *>(p.xobj)) + (size_t)chain * p.R * W + fl * 4);
    const unsigned char *mine = lds + fl * 16;                         // this lane's 16 bytes of a slab entry
    const auto *pcf = as_global(ch.pcf + (size_t)s1 * 512);
    // 2 x 16 x (8 offsets) per start point: the even-slots-first and the odd-slots-first order (m0_kernel)
    const gu4 lists = reinterpret_cast<gu4>(as_global(ch.cl2o)) + (LPR == 8 ? ((sg >> 1) & 1) * 16 : 0);
    const int pieces = ch.ncr * LPR;
    auto mx3 = [](u4 a, u4 b, u4 c) {
        u4 o;
        if (BF16) { o.x = pkmax_u16(pkmax_u16(a.x, b.x), c.x); o.y = pkmax_u16(pkmax_u16(a.y, b.y), c.y); o.z = pkmax_u16(pkmax_u16(a.z, b.z), c.z); o.w = pkmax_u16(pkmax_u16(a.w, b.w), c.w); }
        else { o.x = max(max(a.x, b.x), c.x); o.y = max(max(a.y, b.y), c.y); o.z = max(max(a.z, b.z), c.z); o.w = max(max(a.w, b.w), c.w); }
        return o;
    };
    for (int kb = 0; kb < gn; kb += XR_PASS) {
        const int np = min(XR_PASS, gn - kb);
        // ---- this pass's rows, RPT per thread: (row id | s2 << 22) of the row and of the one before it (a run starts where the
        //      draws differ); requested together with chunk 0's centre ids
        int o_v[RPT], op_v[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int k = tid + NT * u;
            o_v[u] = 0; op_v[u] = -1;
            if (k < np) { o_v[u] = ord[kb + k]; if (k) op_v[u] = ord[kb + k - 1]; }
        }
        int n = 0;
        for (int chunk = 0; chunk < NCHUNK; ++chunk) {
            const int f0 = chunk * F;
            // ---- stage the slab chunk: piece i = (centre i / LPR, 16-byte part i % LPR) at byte 16 i; 64 KiB = NT * 8 pieces at most
            int ci[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = tid + NT * k;
                ci[k] = i < pieces ? clist[i / LPR] : 0;
            }
            int pc_v[RPT];
#pragma unroll
            for (int u = 0; u < RPT; ++u) pc_v[u] = chunk == 0 ? pcf[(unsigned)o_v[u] >> 22] : 0;      // start point, list length, tie flag
            u4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = tid + NT * k;
                if (i < pieces) v[k] = *reinterpret_cast<gu4>(Zg + (size_t)ci[k] * W + f0 + (i % LPR) * 4);
            }
            __syncthreads();                                    // the slab's and the row arrays' previous readers are through
            if (chunk == 0) {
                if (tid == 0) *nlead = 0;
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    const int k = tid + NT * u;
                    if (k < np) {
                        const bool head = ((unsigned)o_v[u] >> 22) != ((unsigned)op_v[u] >> 22);       // op = -1 for the pass's first row
                        rid[k] = o_v[u] & 0x3fffff; qs[k] = (unsigned short)(pc_v[u] & 1023);
                        cfs[k] = (unsigned short)(((pc_v[u] >> 10) & 511) | (head ? 512 : 0));          // 256: order-dependent sequence, xobj_kernel's row
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = tid + NT * k;
                if (i < pieces) *reinterpret_cast<u4 *>(lds + (size_t)i * 16) = v[k];
            }
            __syncthreads();
            if (chunk == 0) {
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    const int k = tid + NT * u;
                    if (k < np && (cfs[k] & 512)) lead[atomicAdd(nlead, 1)] = (unsigned short)k;
                }
                __syncthreads();
                n = *nlead;
            }
            // ---- one run per lane group at a time; the next run's M0 piece and first sixteen offsets are requested a run ahead, a
            //      run's further offsets two iterations (16 slab reads) ahead
            int li = wave * RPW + sg;
            bool have = li < n;
            int kk = have ? lead[li] : 0;
            int q = qs[kk], cf = cfs[kk];
            u4 m0 = *reinterpret_cast<gu4>(M0 + (size_t)q * W + f0);
            u4 cur = lists[q * 32], nx1 = lists[q * 32 + 1];
            while (have) {
                const int li2 = li + NG;
                const bool have2 = li2 < n;
                const int kk2 = have2 ? lead[li2] : kk;
                const int q2 = qs[kk2], cf2 = cfs[kk2];
                const u4 m02 = *reinterpret_cast<gu4>(M0 + (size_t)q2 * W + f0);
                const u4 cur2 = lists[q2 * 32], nx12 = lists[q2 * 32 + 1];
                const int cnt = cf & 255, slow = cf & 256;
                u4 best = m0;
                if (!slow) {
                    const gu4 lp = lists + q * 32;
                    // the list is padded with its last entry to 128: reading past cnt repeats a member
                    for (int j = 0; j < cnt; j += 8) {
                        u4 nx2 = nx1;
                        if (j + 16 < cnt) nx2 = lp[(j >> 3) + 2];
                        const u4 a0 = *reinterpret_cast<const u4 *>(mine + (cur.x & 0xffffu)), a1 = *reinterpret_cast<const u4 *>(mine + (cur.x >> 16));
                        const u4 a2 = *reinterpret_cast<const u4 *>(mine + (cur.y & 0xffffu)), a3 = *reinterpret_cast<const u4 *>(mine + (cur.y >> 16));
                        if (j + 4 < cnt) {
                            const u4 b0 = *reinterpret_cast<const u4 *>(mine + (cur.z & 0xffffu)), b1 = *reinterpret_cast<const u4 *>(mine + (cur.z >> 16));
                            const u4 b2 = *reinterpret_cast<const u4 *>(mine + (cur.w & 0xffffu)), b3 = *reinterpret_cast<const u4 *>(mine + (cur.w >> 16));
                            best = mx3(best, b0, b1); best = mx3(best, b2, b3);
                        }
                        best = mx3(best, a0, a1); best = mx3(best, a2, a3);
                        cur = nx1; nx1 = nx2;
                    }
                }
                int k2 = kk;
                do {                                                           // every row of the run
                    const int r = rid[k2];
                    if (!slow) *reinterpret_cast<__attribute__((address_space(1))) u4 *>(out + (size_t)r * W + f0) = best;
                    else if (chunk == 0 && fl == 0) p.todo[atomicAdd(p.todo_count, 1)] = (int)((int64_t)chain * p.R + r);
                    ++k2;
                } while (k2 < np && !(cfs[k2] & 512));
                li = li2; have = have2; kk = kk2; q = q2; cf = cf2; m0 = m02; cur = cur2; nx1 = nx12;
            }
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(64 * XR_WAVES, 4) void xobj_rows_kernel(const XobjParams p) {
    extern __shared__ unsigned char xr_lds[];
    // work item -> (chain, s1): N items per chain, the chains in the host's order (most crowded centres first: their workgroups
    // run up to eight chunks and should not be the last to start)
    const int N = p.group_N;
    const int chain = p.chain_of_rank[blockIdx.x / N], s1 = blockIdx.x % N;
    const XobjChain ch = p.chains[chain];
    switch (ch.lpr) {
        case 64: if (!BF16) xobj_rows_body<BF16, (BF16 ? 32 : 64)>(p, ch, chain, s1, xr_lds); break;
        case 32: xobj_rows_body<BF16, 32>(p, ch, chain, s1, xr_lds); break;
        case 16: xobj_rows_body<BF16, 16>(p, ch, chain, s1, xr_lds); break;
        default: xobj_rows_body<BF16, 8>(p, ch, chain, s1, xr_lds); break;
    }
}

int pn_pcf(const int *fps1, const int *cnt2, const int *flags, int N, int *pcf, hipStream_t s) {
    hipLaunchKernelGGL(pcf_kernel, dim3((unsigned)((N * 512 + 255) / 256)), dim3(256), 0, s, fps1, cnt2, flags, N, pcf);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_xobj_groups(const XobjParams &p, hipStream_t s) {
    if (p.total_items <= 0) return DGDM_OK;
    static bool attr_set = false;
    if (!attr_set) {
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)xobj_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, XR_LDS_BYTES));
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)xobj_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, XR_LDS_BYTES));
        attr_set = true;
    }
    DGDM_HIP_CHECK(hipMemsetAsync(p.todo_count, 0, sizeof(int), s));
    if (p.xobj16) hipLaunchKernelGGL(xobj_rows_kernel<true>, dim3((unsigned)p.total_items), dim3(64 * XR_WAVES), XR_LDS_BYTES, s, p);
    else hipLaunchKernelGGL(xobj_rows_kernel<false>, dim3((unsigned)p.total_items), dim3(64 * XR_WAVES), XR_LDS_BYTES, s, p);
    DGDM_HIP_CHECK(hipGetLastError());
    // the rows it recorded (tie-flagged start points) run their own FPS
    XobjParams q = p;
    q.skip_fast = 1;
    const int64_t cap = std::min<int64_t>(std::min<int64_t>(p.total_rows, p.todo_capacity), 8192);
    if (p.xobj16) hipLaunchKernelGGL(xobj_kernel<true>, dim3((unsigned)((cap + 3) / 4)), dim3(256), 0, s, q);
    else hipLaunchKernelGGL(xobj_kernel<false>, dim3((unsigned)((cap + 3) / 4)), dim3(256), 0, s, q);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// ------------------------------------------------------------------------------------------------ T8: the embedding table X
// A reference row's embedding depends on its draws (s1, s2) only through s1 and the start POINT q = fps1[s1][s2] of sa2's FPS:
//   X[s1][q] = max( M0[q], max over the crowded centres c of fps2[q] of Z[s1][c] ).
// 512 x 512 rows per object, built once per object by the same slab-in-LDS reduction xobj_group_kernel does per denoise step
// (one workgroup per (s1, feature chunk): the variant's crowded rows staged once, then ALL 512 start points reduced from LDS);
// afterwards a cond_fn call needs no gather at all - the trunk reads row X[s1 * N + q] directly.  One chain uses 36 000 rows per
// step, 180 000 over the 5 steps, of these 262 144 - so the table costs about what 1.5 chains' worth of per-step gathers did, and
// an object usually serves 12 objectives.  Objects without crowded centres need no table: X[s1][q] = M0[q].
constexpr int XG_LDS_BYTES = 78 * 1024;

template <bool BF16, int LPR>
__device__ __forceinline__ void xtab_body(const XtabObj &o, int ncr, int s1, int chunk, uint32_t *slab) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    constexpr int W = BF16 ? 128 : 256, F = 4 * LPR, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f0 = chunk * F, N = o.N;
    const uint32_t *Zs = (BF16 ? o.Z16 : reinterpret_cast<const uint32_t *>(o.Z)) + ((size_t)s1 * N) * W + f0;
    const int pieces = ncr * LPR;
    for (int i0 = threadIdx.x; i0 < pieces; i0 += 256 * 8) {
        u4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + 256 * k;
            if (i < pieces) v[k] = *reinterpret_cast<const u4 *>(Zs + (size_t)o.clist[i / LPR] * W + (i % LPR) * 4);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + 256 * k;
            if (i < pieces) *reinterpret_cast<u4 *>(slab + (size_t)i * 4) = v[k];
        }
    }
    __syncthreads();
    const int sg = lane / LPR, fl = lane % LPR;
    const uint32_t *M0 = (BF16 ? o.M0_16 : reinterpret_cast<const uint32_t *>(o.M0)) + f0 + fl * 4;
    uint32_t *out = (BF16 ? o.X16 : reinterpret_cast<uint32_t *>(o.X)) + ((size_t)s1 * N) * W + f0 + fl * 4;
    auto vmax4 = [](u4 a, u4 b) {
        u4 r;
        if (BF16) { r.x = pkmax_u16(a.x, b.x); r.y = pkmax_u16(a.y, b.y); r.z = pkmax_u16(a.z, b.z); r.w = pkmax_u16(a.w, b.w); }
        else {
            r.x = __float_as_uint(fmaxf(__uint_as_float(a.x), __uint_as_float(b.x))); r.y = __float_as_uint(fmaxf(__uint_as_float(a.y), __uint_as_float(b.y)));
            r.z = __float_as_uint(fmaxf(__uint_as_float(a.z), __uint_as_float(b.z))); r.w = __float_as_uint(fmaxf(__uint_as_float(a.w), __uint_as_float(b.w)));
        }
        return r;
    };
    // start points q = wave, wave + 4, ...: 64 of them per pass, their counts / tie flags one per lane, then four at a time
    for (int kb = 0; kb < N; kb += 256) {
        const int myq = kb + wave + 4 * lane;
        const int cnt_v = myq < N ? o.cnt2[myq] : 0;
        const int slow_v = myq < N ? o.flags[myq] : 1;
        const int nmine = min(64, (N - kb - wave + 3) / 4);
        for (int i0 = 0; i0 < nmine; i0 += 4) {
            int SA[4], SB[4];
            u4 m0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qn = kb + wave + 4 * min(i0 + u, nmine - 1);
                SA[u] = o.cl2s[(size_t)qn * 128 + lane]; SB[u] = o.cl2s[(size_t)qn * 128 + 64 + lane];
                m0[u] = *reinterpret_cast<const u4 *>(M0 + (size_t)qn * W);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if (i >= nmine) break;
                const int q = kb + wave + 4 * i;
                const int cnt = __builtin_amdgcn_readlane(cnt_v, i), slow = __builtin_amdgcn_readlane(slow_v, i);
                if (slow) continue;                                     // xtab_slow_kernel's row
                const int sa = SA[u], sb = SB[u];
                u4 best = m0[u];
                for (int j = 0; j < cnt; j += 4 * RPW) {
                    u4 v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int jj = j + e * RPW;
                        int slot;
                        if (RPW == 1) slot = jj < 64 ? __builtin_amdgcn_readlane(sa, jj & 63) : __builtin_amdgcn_readlane(sb, jj & 63);
                        else slot = __shfl(jj < 64 ? sa : sb, (jj & 63) + sg);
                        v[e] = *reinterpret_cast<const u4 *>(slab + ((size_t)slot * LPR + fl) * 4);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) best = vmax4(best, v[e]);
                }
                if (RPW > 1 && cnt > 0) {
#pragma unroll
                    for (int x = 32; x >= LPR; x >>= 1) {
                        u4 t;
                        t.x = __shfl_xor(best.x, x); t.y = __shfl_xor(best.y, x); t.z = __shfl_xor(best.z, x); t.w = __shfl_xor(best.w, x);
                        best = vmax4(best, t);
                    }
                }
                if (sg == 0) *reinterpret_cast<u4 *>(out + (size_t)q * W) = best;
            }
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(256, 2) void xtab_kernel(const XtabObj o) {
    extern __shared__ uint32_t xt_slab[];
    constexpr int W = BF16 ? 128 : 256, MAXCH = W / 32;
    const int ncr = *o.ncr;
    if (ncr == 0) return;                                               // the object's table is M0 itself
    int lpr = W / 4;
    while (lpr > 8 && (size_t)ncr * lpr * 16 > (size_t)XG_LDS_BYTES) lpr >>= 1;
    const int nchunk = (W / 4) / lpr;
    const int s1 = blockIdx.x / MAXCH, chunk = blockIdx.x % MAXCH;
    if (chunk >= nchunk) return;
    switch (lpr) {
        case 64: if (!BF16) xtab_body<BF16, (BF16 ? 32 : 64)>(o, ncr, s1, chunk, xt_slab); break;
        case 32: xtab_body<BF16, 32>(o, ncr, s1, chunk, xt_slab); break;
        case 16: xtab_body<BF16, 16>(o, ncr, s1, chunk, xt_slab); break;
        default: xtab_body<BF16, 8>(o, ncr, s1, chunk, xt_slab); break;
    }
}

// Rows X[s1][q] of the start points q whose FPS(128) sequence depends on the ordering (exact distance ties: flags[q]): FPS on the
// cloud as re-ordered by variant s1, started at q's position, as the reference does for every row.  One workgroup per s1.
template <bool BF16>
__global__ __launch_bounds__(256) void xtab_slow_kernel(const XtabObj o) {
    typedef RowOps<BF16> R;
    __shared__ float coords[4][3][512];
    __shared__ int centres[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s1 = blockIdx.x, N = o.N;
    if (*o.ncr == 0) return;
    const int *perm = o.fps1 + (size_t)s1 * 512;
    XobjChain ch{};
    ch.Z = o.Z; ch.Z16 = o.Z16;
    int seen = 0;
    for (int q = 0; q < N; ++q) {
        if (o.flags[q] == 0) continue;                                  // uniform
        if ((seen++ & 3) != wave) continue;
        // position of q in the variant's order (any of them if the point repeats: the sequence of POINTS is the same)
        int s2 = -1;
        for (int base = 0; base < 512 && s2 < 0; base += 64) {
            const unsigned long long m = __ballot(perm[base + lane] == q);
            if (m) s2 = base + __ffsll((long long)m) - 1;
        }
        if (s2 < 0) continue;                                           // q is not among this variant's 512 centres: no row refers to it
        float *lx = coords[wave][0], *ly = coords[wave][1], *lz = coords[wave][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = lane + 64 * i, pid = perm[j];
            lx[j] = o.xyz[3 * pid]; ly[j] = o.xyz[3 * pid + 1]; lz[j] = o.xyz[3 * pid + 2];
        }
        __builtin_amdgcn_wave_barrier();
        fps_wave<8, false>(lx, ly, lz, 512, s2, 128, centres[wave], lane);
        __builtin_amdgcn_wave_barrier();
        const int idA = perm[centres[wave][lane]], idB = perm[centres[wave][64 + lane]];
        const int rowA = (o.crowded[idA] ? s1 * N : 0) + idA, rowB = (o.crowded[idB] ? s1 * N : 0) + idB;
        typename R::V best = R::zero();
#pragma unroll 4
        for (int i = 0; i < 64; i += 2) {
            const int c0 = __shfl(rowA, i), c1 = __shfl(rowA, i + 1), c2 = __shfl(rowB, i), c3 = __shfl(rowB, i + 1);
            const typename R::V a = R::load(ch, false, (size_t)c0, lane), b = R::load(ch, false, (size_t)c1, lane);
            const typename R::V d = R::load(ch, false, (size_t)c2, lane), e = R::load(ch, false, (size_t)c3, lane);
            best = R::vmax(R::vmax(best, R::vmax(a, b)), R::vmax(d, e));
        }
        if (BF16) reinterpret_cast<uint2 *>(o.X16 + ((size_t)s1 * N + q) * 128)[lane] = *reinterpret_cast<uint2 *>(&best);
        else reinterpret_cast<float4 *>(o.X + ((size_t)s1 * N + q) * 256)[lane] = *reinterpret_cast<float4 *>(&best);
        __builtin_amdgcn_wave_barrier();
    }
}

int pn_xtab(const XtabObj &o, bool bf16, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)xtab_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, XG_LDS_BYTES));
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)xtab_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, XG_LDS_BYTES));
        attr_set = true;
    }
    if (bf16) {
        hipLaunchKernelGGL(xtab_kernel<true>, dim3(o.N * 4), dim3(256), XG_LDS_BYTES, s, o);
        hipLaunchKernelGGL(xtab_slow_kernel<true>, dim3(o.N), dim3(256), 0, s, o);
    } else {
        hipLaunchKernelGGL(xtab_kernel<false>, dim3(o.N * 8), dim3(256), XG_LDS_BYTES, s, o);
        hipLaunchKernelGGL(xtab_slow_kernel<false>, dim3(o.N), dim3(256), 0, s, o);
    }
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

__global__ __launch_bounds__(256) void xidx_kernel(const XidxChain *__restrict__ chains, const int *__restrict__ starts, int64_t R, int64_t total,
                                                   int *__restrict__ idx) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= total) return;
    const int chain = (int)(w / R);
    const XidxChain ch = chains[chain];
    const int s1 = starts[2 * w], s2 = starts[2 * w + 1];
    const int q = ch.fps1[(size_t)s1 * 512 + s2];
    idx[w] = ch.m0_only ? q : s1 * ch.N + q;
}

int pn_xidx(const XidxChain *chains_dev, const int *starts, int64_t R, int nchain, int *idx, hipStream_t s) {
    const int64_t total = R * nchain;
    if (total <= 0) return DGDM_OK;
    hipLaunchKernelGGL(xidx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, chains_dev, starts, R, total, idx);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// ------------------------------------------------------------------------------------------------ the index functions on their own
// dynamics/models/pointnet2_utils.py:27-115 for arbitrary batches of clouds (the reference's names, mirrored in
// dgdm_amd/dynamics/models/pointnet2_utils.py).  The guided path does not call these - it reads the per-object tables above - but they are
// the same device code (fps_wave, the expanded distance form) with per-cloud inputs.
__global__ __launch_bounds__(256) void fps_rows_kernel(const float *__restrict__ xyz /*[B][N][3]*/, const int *__restrict__ start /*[B]*/, int B, int N,
                                                       int npoint, int *__restrict__ out /*[B][npoint]*/) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    float *lx = lds + (size_t)wave * 3 * N, *ly = lx + N, *lz = ly + N;
    if (b < B) {
        const float *c = xyz + (size_t)b * N * 3;
        for (int i = lane; i < N; i += 64) { lx[i] = c[3 * i]; ly[i] = c[3 * i + 1]; lz[i] = c[3 * i + 2]; }
    }
    __builtin_amdgcn_wave_barrier();
    if (b >= B) return;
    if (N <= 512) fps_wave<8, false>(lx, ly, lz, N, start[b], npoint, out + (size_t)b * npoint, lane);
    else fps_wave<16, false>(lx, ly, lz, N, start[b], npoint, out + (size_t)b * npoint, lane);
}

// first `nsample` indices k (ascending) with square_distance(centre, xyz[k]) <= r2, padded with the first one; a group without any member
// is filled with N, which is what the reference's masked assignment leaves there (group_first is N then)
__global__ __launch_bounds__(256) void ball_rows_kernel(const float *__restrict__ xyz /*[B][N][3]*/, const float *__restrict__ centres /*[B][S][3]*/, int B,
                                                        int N, int S, float r2, int nsample, int *__restrict__ out /*[B][S][nsample]*/) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= (int64_t)B * S) return;
    const int b = (int)(g / S);
    const float *c = xyz + (size_t)b * N * 3;
    const float cx = centres[3 * g], cy = centres[3 * g + 1], cz = centres[3 * g + 2];
    const float cn = sq3(cx, cy, cz);
    int *dst = out + (size_t)g * nsample;
    int cnt = 0, first = N;
    for (int base = 0; base < N && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool in = false;
        if (k < N) {
            const float x = c[3 * k], y = c[3 * k + 1], z = c[3 * k + 2];
            in = !(sqdist_expanded(cx, cy, cz, cn, x, y, z, sq3(x, y, z)) > r2);
        }
        const unsigned long long m = __ballot(in);
        if (m && first == N) first = base + __ffsll((long long)m) - 1;
        const int rank = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (in && rank < nsample) dst[rank] = k;
        cnt += __popcll(m);
    }
    cnt = min(cnt, nsample);
    for (int i = cnt + lane; i < nsample; i += 64) dst[i] = first;
}

__global__ void sqdist_rows_kernel(const float *__restrict__ src /*[B][S][3]*/, const float *__restrict__ dst /*[B][N][3]*/, int B, int S, int N,
                                   float *__restrict__ out /*[B][S][N]*/) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * S * N) return;
    const int n = (int)(i % N);
    const int64_t bs = i / N;
    const int b = (int)(bs / S);
    const float *a = src + 3 * bs, *p = dst + ((size_t)b * N + n) * 3;
    out[i] = sqdist_expanded(a[0], a[1], a[2], sq3(a[0], a[1], a[2]), p[0], p[1], p[2], sq3(p[0], p[1], p[2]));
}

__global__ void index_rows_kernel(const float *__restrict__ points /*[B][N][C]*/, const int *__restrict__ idx /*[B][M]*/, int B, int N, int M, int C,
                                  float *__restrict__ out /*[B][M][C]*/) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * M * C) return;
    const int c = (int)(i % C);
    const int64_t bm = i / C;
    const int b = (int)(bm / M);
    // an index outside [0, N) - query_ball_point's marker N for a group without members, where the reference's indexing raises - reads
    // nothing and yields NaN
    const int id = idx[bm];
    out[i] = (id >= 0 && id < N) ? points[((size_t)b * N + id) * C + c] : __builtin_nanf("");
}

int pn_fps_rows(const float *xyz, const int *start, int B, int N, int npoint, int *out, hipStream_t s) {
    hipLaunchKernelGGL(fps_rows_kernel, dim3((B + 3) / 4), dim3(256), (size_t)4 * 3 * N * sizeof(float), s, xyz, start, B, N, npoint, out);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
int pn_ball_rows(const float *xyz, const float *centres, int B, int N, int S, float r2, int nsample, int *out, hipStream_t s) {
    hipLaunchKernelGGL(ball_rows_kernel, dim3((unsigned)(((int64_t)B * S + 3) / 4)), dim3(256), 0, s, xyz, centres, B, N, S, r2, nsample, out);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
int pn_sqdist_rows(const float *src, const float *dst, int B, int S, int N, float *out, hipStream_t s) {
    const int64_t n = (int64_t)B * S * N;
    hipLaunchKernelGGL(sqdist_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, B, S, N, out);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
int pn_index_rows(const float *points, const int *idx, int B, int N, int M, int C, float *out, hipStream_t s) {
    const int64_t n = (int64_t)B * M * C;
    hipLaunchKernelGGL(index_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, points, idx, B, N, M, C, out);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// ------------------------------------------------------------------------------------------------ host side
int pn_fps_table(const float *xyz, int N, int nv, int npoint, int *out, int *flags, hipStream_t s, int nobj) {
    hipLaunchKernelGGL(fps_table_kernel, dim3((nv + 3) / 4, nobj), dim3(256), (size_t)3 * N * sizeof(float), s, xyz, N, nv, npoint, out, flags);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_sa1(const float *xyz, int N, const PnWeights &w, float *F1, hipStream_t s, int nobj) {
    hipLaunchKernelGGL(sa1_kernel, dim3(std::min(N, 1024), nobj), dim3(128), 0, s, xyz, N, w.r1sq, w.sa1_w0t, w.sa1_b0, w.sa1_w1, w.sa1_b1, F1);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_pairs(const float *xyz, int N, const float *U, const PnWeights &w, const int *pairs, const int *off, float *Y, uint32_t *Y16, hipStream_t s) {
    const int tiles = N * ((N + 31) / 32);     // worst case (every point inside every ball); surplus workgroups leave at once
    hipLaunchKernelGGL(pair_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, xyz, N, U, w.sa2_vx, w.sa2_w1_img, w.sa2_b1, pairs, off, Y, Y16);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_crowd(const float *xyz, int N, const PnWeights &w, int *crowded, int *clist, int *ncr, int *off, int *pairs, short *rank, hipStream_t s,
             int *ncr_copy, int nobj) {
    // nobj > 1: the arrays are pools [nobj][...] at their natural strides (clist and off: N + 1 entries per object, ncr = clist + N)
    hipLaunchKernelGGL(crowd_kernel, dim3(1, nobj), dim3(1024), 0, s, xyz, N, w.r2sq, crowded, clist, ncr, off, ncr_copy);
    hipLaunchKernelGGL(nbr_fill_kernel, dim3((N + 3) / 4, nobj), dim3(256), 0, s, xyz, N, w.r2sq, off, pairs, rank);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_l2c(int N, const int *fps1, int nv, const float *Y, float *L2, const int *clist, const int *ncr, const int *off, const short *rank, bool bf16,
           hipStream_t s) {
    static bool attr_set = false;
    const size_t lds = (size_t)(512 + 512 * 16 + 128) * 4 + L2C_YBYTES;
    if (!attr_set) {
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)l2c_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DGDM_HIP_CHECK(hipFuncSetAttribute((const void *)l2c_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    // 64 x 4 = 256 workgroups, one per CU (150 KB of LDS each): ONE round whatever the number of crowded centres is (0 .. N: a few
    // objects of a batch have every centre crowded and carry most of the build's work), and 160 busy CUs at the typical 40
    static const int split = []() { const char *e = getenv("DGDM_L2C_SPLIT"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 32 ? 32 : v); }();
    static const int gx = []() { const char *e = getenv("DGDM_L2C_GRID"); const int v = e ? atoi(e) : 64; return v < 1 ? 1 : v; }();
    const dim3 grid(std::min(N, gx), std::min(split, std::max(nv - 1, 1)));
    if (bf16) hipLaunchKernelGGL(l2c_kernel<true>, grid, dim3(L2C_THREADS), lds, s, N, fps1, nv, reinterpret_cast<const uint32_t *>(Y),
                                 reinterpret_cast<uint32_t *>(L2), clist, ncr, off, rank);
    else hipLaunchKernelGGL(l2c_kernel<false>, grid, dim3(L2C_THREADS), lds, s, N, fps1, nv, reinterpret_cast<const uint32_t *>(Y),
                            reinterpret_cast<uint32_t *>(L2), clist, ncr, off, rank);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_l2(const float *xyz, int N, const PnWeights &w, const int *fps1, const int *vlist, int nv, const float *Y, float *L2,
          const int *clist, const int *ncr, const int *off, const short *rank, bool bf16, hipStream_t s, int crowded_mode) {
    const dim3 g0(1, (N + 3) / 4), g1((unsigned)(((nv - 1 + 3) / 4) * ((N + 7) / 8) * 8));   // g1: worst case (every centre crowded)
    if (crowded_mode == 1 && N <= 1024 && nv <= 512) {            // slot 0 here, slots >= 1 of the crowded centres by l2c_kernel
        if (bf16) hipLaunchKernelGGL(l2_kernel<true>, g0, dim3(256), 0, s, xyz, N, w.r2sq, fps1, vlist, nv, Y, L2, 0, clist, ncr, off, rank);
        else hipLaunchKernelGGL(l2_kernel<false>, g0, dim3(256), 0, s, xyz, N, w.r2sq, fps1, vlist, nv, Y, L2, 0, clist, ncr, off, rank);
        DGDM_HIP_CHECK(hipGetLastError());
        return nv > 1 ? pn_l2c(N, fps1, nv, Y, L2, clist, ncr, off, rank, bf16, s) : DGDM_OK;
    }
    if (bf16) {
        hipLaunchKernelGGL(l2_kernel<true>, g0, dim3(256), 0, s, xyz, N, w.r2sq, fps1, vlist, nv, Y, L2, 0, clist, ncr, off, rank);
        if (nv > 1) hipLaunchKernelGGL(l2_kernel<true>, g1, dim3(256), 0, s, xyz, N, w.r2sq, fps1, vlist, nv, Y, L2, 1, clist, ncr, off, rank);
    } else {
        hipLaunchKernelGGL(l2_kernel<false>, g0, dim3(256), 0, s, xyz, N, w.r2sq, fps1, vlist, nv, Y, L2, 0, clist, ncr, off, rank);
        if (nv > 1) hipLaunchKernelGGL(l2_kernel<false>, g1, dim3(256), 0, s, xyz, N, w.r2sq, fps1, vlist, nv, Y, L2, 1, clist, ncr, off, rank);
    }
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_z(const float *xyz, int N, int nv, const PnWeights &w, const float *L2, float *Z, uint32_t *Z16, const int *clist, const int *ncr,
         hipStream_t s) {
    const int64_t t0 = (N + 31) / 32;
    hipLaunchKernelGGL(z_kernel, dim3((unsigned)((t0 + 3) / 4)), dim3(256), 0, s, xyz, N, nv, L2, w.sa3_w_img, w.sa3_wx, w.sa3_b, Z, Z16, 0, clist, ncr);
    if (nv > 1) {      // sized for the worst case (every centre crowded); surplus workgroups leave at once
        const int64_t t1 = ((int64_t)(nv - 1) * N + 31) / 32;
        hipLaunchKernelGGL(z_kernel, dim3((unsigned)((t1 + 3) / 4)), dim3(256), 0, s, xyz, N, nv, L2, w.sa3_w_img, w.sa3_wx, w.sa3_b, Z, Z16, 1, clist, ncr);
    }
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_z16(const float *xyz, int N, int nv, const PnWeights &w, const uint32_t *L2_16, float *Z, uint32_t *Z16, const int *clist, const int *ncr,
           hipStream_t s) {
    const int64_t t0 = (N + 63) / 64;
    hipLaunchKernelGGL(z16_kernel, dim3((unsigned)((t0 + 3) / 4)), dim3(256), 0, s, xyz, N, nv, L2_16, w.sa3_w_img16, w.sa3_wx, w.sa3_b, Z, Z16, 0, clist, ncr);
    if (nv > 1) {      // sized for the worst case (every centre crowded); surplus workgroups leave at once
        const int64_t t1 = ((int64_t)(nv - 1) * N + 63) / 64;
        hipLaunchKernelGGL(z16_kernel, dim3((unsigned)((t1 + 3) / 4)), dim3(256), 0, s, xyz, N, nv, L2_16, w.sa3_w_img16, w.sa3_wx, w.sa3_b, Z, Z16, 1, clist, ncr);
    }
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_m0(const int *fps2, const int *crowded, int N, const float *Z0, float *M0, int *cl2, int *cnt2, const uint32_t *Z0_16, uint32_t *M0_16,
          const int *clist, const int *ncr, int *cl2s, unsigned short *cl2o, hipStream_t s) {
    hipLaunchKernelGGL(m0_kernel, dim3((N + 3) / 4), dim3(256), 0, s, fps2, crowded, N, Z0, M0, cl2, cnt2, Z0_16, M0_16, clist, ncr, cl2s, cl2o,
                       M0_16 ? 1 : 0);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// ------------------------------------------------------------------------------------------------ index test hook
// The index decisions of the pipeline, written out for bit-exact comparison with the reference's farthest_point_sample /
// query_ball_point outputs (tests/test_gpu_indices.py).  Same device functions as the production kernels.
__global__ __launch_bounds__(64) void debug_ball1_kernel(const float *__restrict__ xyz, int N, float r2, int *__restrict__ out /*[N][32]*/) {
    __shared__ int nbr[32];
    const int p = blockIdx.x, lane = threadIdx.x;
    const float cx = xyz[3 * p], cy = xyz[3 * p + 1], cz = xyz[3 * p + 2];
    ball_first32(xyz, N, p, cx, cy, cz, sq3(cx, cy, cz), r2, nbr, lane);
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) out[(size_t)p * 32 + lane] = nbr[lane];
}

// out[c][0..cnt[c]) = point ids of the first 64 in-radius (r = 0.4) candidates of centre POINT c when the cloud is scanned in
// the order perm[0..M) (l2_kernel's selection: rank-table row in LDS, l2_select, pair list)
__global__ __launch_bounds__(64) void debug_ball2_kernel(int N, const int *__restrict__ perm, int M, const int *__restrict__ off,
                                                         const int *__restrict__ pairs, const short *__restrict__ rank,
                                                         int *__restrict__ out /*[N][64]*/, int *__restrict__ cnt_out /*[N]*/) {
    __shared__ int sel[64];
    __shared__ short rks[1024];
    const int c = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < N; i += 64) rks[i] = rank[(size_t)c * N + i];
    __builtin_amdgcn_wave_barrier();
    const int cnt = l2_select(rks, perm, M, sel, lane);
    out[(size_t)c * 64 + lane] = lane < cnt ? (pairs[off[c] + sel[lane]] & 0xffff) : -1;
    if (lane == 0) cnt_out[c] = cnt;
}

int pn_debug_indices(const float *xyz, int N, const PnWeights &w, const int *perm, int M, int *ball1, int *ball2, int *ball2_cnt, int *crowded,
                     hipStream_t s) {
    DevBuf clist, off, pairs, rank;
    int rc;
    if ((rc = clist.alloc((size_t)(N + 1) * 4)) || (rc = off.alloc((size_t)(N + 1) * 4)) || (rc = pairs.alloc((size_t)N * N * 4)) ||
        (rc = rank.alloc((size_t)N * N * 2)))
        return rc;
    hipLaunchKernelGGL(debug_ball1_kernel, dim3(N), dim3(64), 0, s, xyz, N, w.r1sq, ball1);
    if ((rc = pn_crowd(xyz, N, w, crowded, clist.as<int>(), clist.as<int>() + N, off.as<int>(), pairs.as<int>(), rank.as<short>(), s))) return rc;
    hipLaunchKernelGGL(debug_ball2_kernel, dim3(N), dim3(64), 0, s, N, perm, M, off.as<int>(), pairs.as<int>(), rank.as<short>(), ball2, ball2_cnt);
    DGDM_HIP_CHECK(hipGetLastError());
    DGDM_HIP_CHECK(hipStreamSynchronize(s));     // temporaries die here
    return DGDM_OK;
}

template <bool BF16>
static int xobj_launch(XobjParams p, bool all_fast, hipStream_t s) {
    p.skip_fast = 0;
    if (p.use_table && p.todo) {
        DGDM_HIP_CHECK(hipMemsetAsync(p.todo_count, 0, sizeof(int), s));
        hipLaunchKernelGGL(xobj_fast_kernel<BF16>, dim3((unsigned)((p.total_rows + 3) / 4)), dim3(256), 0, s, p);
        DGDM_HIP_CHECK(hipGetLastError());
        if (all_fast) return DGDM_OK;               // no chain without tables, no start point with an order-dependent sequence
        // the rows it recorded (a fraction of a percent: tie-flagged start points, or every row of a chain without tables): a
        // bounded grid walks the list
        p.skip_fast = 1;
        const int64_t cap = std::min<int64_t>(std::min<int64_t>(p.total_rows, p.todo_capacity), 65536);
        hipLaunchKernelGGL(xobj_kernel<BF16>, dim3((unsigned)((cap + 3) / 4)), dim3(256), 0, s, p);
        DGDM_HIP_CHECK(hipGetLastError());
        return DGDM_OK;
    }
    hipLaunchKernelGGL(xobj_kernel<BF16>, dim3((unsigned)((p.total_rows + 3) / 4)), dim3(256), 0, s, p);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int pn_xobj(const XobjParams &p, bool all_fast, hipStream_t s) {
    if (p.total_rows <= 0) return DGDM_OK;
    return p.xobj16 ? xobj_launch<true>(p, all_fast, s) : xobj_launch<false>(p, all_fast, s);
}

}  // namespace dgdm
