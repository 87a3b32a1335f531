// Fused dynamics-trunk forward + input-gradient backward (the dominant kernel of the path).
//
// Reference dataflow being replaced: the 8 x (Linear -> BatchNorm1d(eval) -> ReLU) + Linear(256,3)
// trunk of ProfileForward{2,3}DModel (dynamics/profile_forward_2d.py:109-135,154-155;
// profile_forward_3d.py:39-65,84-85) evaluated on the R = B*G*P*P replicated rows that
// Diffusion.cond_fn builds (generator/diffusion.py:478-500), the objective of
// deltas_to_objective (:430-471) and torch.autograd.grad of its sum w.r.t. the fingers (:498,504).
//
// One wave = one tile of 32 rows, all layers, forward and backward, in registers (mfma_chain.h).
// A tile is one finger b of one chain against 32 consecutive pose cells, so the sum over cells that
// autograd performs when it folds the replicated rows back onto x becomes a sum over the tile's 32
// MFMA columns; each tile writes its partial d/dz1 vector and dyn_post_kernel adds the tiles in a
// fixed order (deterministic, no atomics).
//
// First layer: Linear is linear in the concatenation [x_object | x_ctrl | x_pose | time_emb]
// (profile_forward_2d.py:154), so with BatchNorm folded
//     z1[row(c,b)] = Atab[chain,b] + Ptab[c]            (2-D; Atab carries object+time+bias terms)
//     z1[row]      = W1o' * xobj[row] + Atab + Ptab     (3-D; the PointNet++ embedding differs per row
//                                                        because of the per-row random FPS starts)
// The backward pass needs no activations, only the ReLU sign bits (kept in LDS, 1 bit per unit),
// because the dynamics weights are frozen (generator/train.py:91-92).
#include "common.h"
#include <algorithm>
#include "mfma_chain.h"
#include "trunk.h"

namespace dgdm {

template <int KIND, bool ROWS, bool FWD_ONLY>
__global__ __launch_bounds__(256, 1) void trunk_kernel(const TrunkParams p) {
    constexpr int W1B = (KIND == 3) ? 16 : 8;                 // first-layer width in 32-feature blocks
    constexpr int W1 = W1B * 32;
    constexpr int NSLOT = (KIND == 3) ? 8 + 7 * 4 : 8 * 4;
    __shared__ uint32_t smask[FWD_ONLY ? 1 : NSLOT][256];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= p.ntiles) return;                             // wave-uniform; the kernel has no barrier
    const int n = lane & 31;
    const int h4 = (lane >> 5) * 4;

    // ---- which rows
    int chain, b, c;
    bool valid;
    int64_t r;                                                // reference row index inside the chain
    const float *arow;
    const float4 *ptile = nullptr;                            // table mode: this tile's cells in the tiled pose table, + lane
    if (ROWS) {
        chain = 0; b = 0;
        c = tile * 32 + n;
        valid = c < p.C;
        r = valid ? c : p.C - 1;
        arow = p.Atab + (size_t)r * W1;
    } else {
        const int per_chain = p.B * p.tiles_per_b;
        chain = tile / per_chain;
        const int rem = tile - chain * per_chain;
        b = rem / p.tiles_per_b;
        c = (rem - b * p.tiles_per_b) * 32 + n;
        valid = c < p.C;
        const int cc = valid ? c : p.C - 1;
        r = (int64_t)cc * p.B + b;
        arow = p.Atab + (size_t)(chain * p.B + b) * W1;
        ptile = reinterpret_cast<const float4 *>(p.PtabT) + (size_t)(rem - b * p.tiles_per_b) * W1B * 4 * 64 + lane;
    }

    f32x16 cur[8], nxt[8];
    uint32_t m[4];
    int slot = 0;
    const wrsrc_t rsF = weight_rsrc(p.Wfwd, p.fwd_bytes);
    float4 ring[CONT_DEPTH];
    int woff = 0;                                             // byte offset of the next pass inside the forward stream
    ring_fill(rsF, lane * 16, 0, ring);

    if (KIND == 2) {
        // ---- layer 1: table lookups
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = feat4(arow, o, q, h4);
                if (!ROWS) {
                    const float4 w = ptile[(o * 4 + q) * 64];
                    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                }
                cur[o][4 * q + 0] = v.x; cur[o][4 * q + 1] = v.y; cur[o][4 * q + 2] = v.z; cur[o][4 * q + 3] = v.w;
            }
        }
        relu_mask<8>(cur, m);
        if (!FWD_ONLY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) smask[slot + i][tid] = m[i];
        }
        slot += 4;
    } else {
        // ---- 3-D layers 1 and 2, streamed over the 16 blocks of the 512-wide layer 1
        const float *xrow = (!ROWS && p.xtab) ? p.xtab[chain] + (size_t)p.xidx[(size_t)chain * p.xstride + r] * 256
                                              : p.xobj + ((size_t)chain * p.xstride + r) * 256;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = feat4(xrow, o, q, h4);
                nxt[o][4 * q + 0] = v.x; nxt[o][4 * q + 1] = v.y; nxt[o][4 * q + 2] = v.z; nxt[o][4 * q + 3] = v.w;
            }
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = feat4(p.b2, o, q, h4);
                cur[o][4 * q + 0] = b4.x; cur[o][4 * q + 1] = b4.y; cur[o][4 * q + 2] = b4.z; cur[o][4 * q + 3] = b4.w;
            }
        }
        for (int blk = 0; blk < 16; blk += 2) {
            uint32_t bits2 = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kb = blk + e;
                f32x16 z = chain_block_cont<8>(rsF, woff, ring, nxt, lane);
                woff += 32 * 1024;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = *reinterpret_cast<const float4 *>(arow + 32 * kb + 8 * q + h4);
                    if (!ROWS) {
                        const float4 w = ptile[(kb * 4 + q) * 64];
                        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                    }
                    z[4 * q + 0] += v.x; z[4 * q + 1] += v.y; z[4 * q + 2] += v.z; z[4 * q + 3] += v.w;
                }
                bits2 |= relu_bits(z) << (16 * e);
                chain_accumulate_cont<8>(rsF, woff, ring, z, cur, lane);
                woff += 32 * 1024;
            }
            if (!FWD_ONLY) smask[blk / 2][tid] = bits2;
        }
        slot = 8;
        relu_mask<8>(cur, m);
        if (!FWD_ONLY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) smask[slot + i][tid] = m[i];
        }
        slot += 4;
    }

    // ---- 256 -> 256 layers
    for (int l = 0; l < p.n_mid; ++l) {
        chain_layer_cont<8, 8, CHAIN_BIAS>(rsF, woff, ring, p.bf[l], cur, nxt, lane);
        woff += 256 * 1024;
        relu_mask<8>(nxt, m);
        if (!FWD_ONLY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) smask[slot + i][tid] = m[i];
        }
        slot += 4;
#pragma unroll
        for (int o = 0; o < 8; ++o) cur[o] = nxt[o];
    }

    // ---- output layer (256 -> 3) on the VALU, objective gradient
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w0 = feat4(p.Wout, o, q, h4);
            const float4 w1 = feat4(p.Wout + 256, o, q, h4);
            const float4 w2 = feat4(p.Wout + 512, o, q, h4);
            const float x0 = cur[o][4 * q + 0], x1 = cur[o][4 * q + 1], x2 = cur[o][4 * q + 2], x3 = cur[o][4 * q + 3];
            s0 = fmaf(w0.w, x3, fmaf(w0.z, x2, fmaf(w0.y, x1, fmaf(w0.x, x0, s0))));
            s1 = fmaf(w1.w, x3, fmaf(w1.z, x2, fmaf(w1.y, x1, fmaf(w1.x, x0, s1))));
            s2 = fmaf(w2.w, x3, fmaf(w2.z, x2, fmaf(w2.y, x1, fmaf(w2.x, x0, s2))));
        }
    }
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    const float d0 = s0 + p.bout[0], d1 = s1 + p.bout[1], d2 = s2 + p.bout[2];

    if (FWD_ONLY) {
        if (valid && lane < 32) {
            float *dst = p.logits + ((size_t)chain * p.R + r) * 3;
            dst[0] = d0; dst[1] = d1; dst[2] = d2;
        }
        return;
    } else {
        const wrsrc_t rsB = weight_rsrc(p.Wbwd, p.bwd_bytes);
        woff = 0;
        ring_fill(rsB, lane * 16, 0, ring);                   // in flight while the output layer and the objective run on the VALU
        const TrunkObjective ob = p.obj[chain];
        float g0 = ob.lin[0] + 2.f * ob.quad[0] * d0;
        float g1 = ob.lin[1] + 2.f * ob.quad[1] * d1;
        float g2 = ob.lin[2] + 2.f * ob.quad[2] * d2;
        if (ob.use_rowcoef) g0 = p.rowcoef[(size_t)chain * p.R + r];
        if (!valid) { g0 = 0.f; g1 = 0.f; g2 = 0.f; }

        slot -= 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = smask[slot + i][tid];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w0 = feat4(p.Wout, o, q, h4);
                const float4 w1 = feat4(p.Wout + 256, o, q, h4);
                const float4 w2 = feat4(p.Wout + 512, o, q, h4);
                cur[o][4 * q + 0] = fmaf(g2, w2.x, fmaf(g1, w1.x, g0 * w0.x));
                cur[o][4 * q + 1] = fmaf(g2, w2.y, fmaf(g1, w1.y, g0 * w0.y));
                cur[o][4 * q + 2] = fmaf(g2, w2.z, fmaf(g1, w1.z, g0 * w0.z));
                cur[o][4 * q + 3] = fmaf(g2, w2.w, fmaf(g1, w1.w, g0 * w0.w));
            }
        }
        apply_mask<8>(cur, m);

        // ---- backward through the 256 -> 256 layers
        for (int l = p.n_mid - 1; l >= 0; --l) {
            chain_layer_cont<8, 8, CHAIN_ZERO>(rsB, woff, ring, nullptr, cur, nxt, lane);
            woff += 256 * 1024;
            slot -= 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = smask[slot + i][tid];
            apply_mask<8>(nxt, m);
#pragma unroll
            for (int o = 0; o < 8; ++o) cur[o] = nxt[o];
        }

        float *dst = p.partial + (size_t)tile * W1;
        if (KIND == 2) {
            // cur = d/dz1 of every row of the tile; fold the 32 cells
#pragma unroll
            for (int o = 0; o < 8; ++o) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v;
                    v.x = rows_sum(cur[o][4 * q + 0]); v.y = rows_sum(cur[o][4 * q + 1]);
                    v.z = rows_sum(cur[o][4 * q + 2]); v.w = rows_sum(cur[o][4 * q + 3]);
                    if (n == ROWS_SUM_LANE) *reinterpret_cast<float4 *>(dst + 32 * o + 8 * q + h4) = v;
                }
            }
        } else {
            // 3-D: one more layer back (256 -> 512), block by block, straight into the fold
            for (int blk = 0; blk < 16; blk += 2) {
                const uint32_t bits2 = smask[blk / 2][tid];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int kb = blk + e;
                    f32x16 g = chain_block_cont<8>(rsB, woff, ring, cur, lane);
                    woff += 32 * 1024;
                    apply_bits(g, (bits2 >> (16 * e)) & 0xffffu);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v;
                        v.x = rows_sum(g[4 * q + 0]); v.y = rows_sum(g[4 * q + 1]);
                        v.z = rows_sum(g[4 * q + 2]); v.w = rows_sum(g[4 * q + 3]);
                        if (n == ROWS_SUM_LANE) *reinterpret_cast<float4 *>(dst + 32 * kb + 8 * q + h4) = v;
                    }
                }
            }
        }
    }
}

template <int KIND, bool ROWS, bool FWD_ONLY>
static int launch(const TrunkParams &p, hipStream_t s) {
    const int grid = (p.ntiles + 3) / 4;
    if (grid == 0) return DGDM_OK;
    hipLaunchKernelGGL((trunk_kernel<KIND, ROWS, FWD_ONLY>), dim3(grid), dim3(256), 0, s, p);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int trunk_launch(int kind, bool rows_mode, bool fwd_only, const TrunkParams &p, hipStream_t s) {
    // algorithmic FLOPs of what this launch executes (DESIGN_HISTORY.md §5): MFMA layers only
    // real rows only (the last cell tile of a finger is padded to 32: 1125 cells -> 36 tiles = 1152 issued rows)
    const double rows = rows_mode ? (double)p.R : (double)(p.ntiles / std::max(1, p.tiles_per_b)) * p.C;
    const double mid = 2.0 * 256 * 256 * p.n_mid;
    double per_row = (kind == 3) ? (2.0 * 256 * 512 * 2 + mid) : mid;
    if (!fwd_only) per_row += (kind == 3) ? (2.0 * 256 * 512 + mid) : mid;
    prof_begin(s, DGDM_STAGE_TRUNK);
    int rc;
    if (kind == 2) {
        if (rows_mode) rc = fwd_only ? launch<2, true, true>(p, s) : DGDM_EINVAL;
        else rc = fwd_only ? launch<2, false, true>(p, s) : launch<2, false, false>(p, s);
    } else {
        if (rows_mode) rc = fwd_only ? launch<3, true, true>(p, s) : DGDM_EINVAL;
        else rc = fwd_only ? launch<3, false, true>(p, s) : launch<3, false, false>(p, s);
    }
    prof_end(s, DGDM_STAGE_TRUNK, rows * per_row);
    return rc;
}

}  // namespace dgdm
