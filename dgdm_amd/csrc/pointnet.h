#pragma once
#include "common.h"

namespace dgdm {

// PointNet2 weights, BatchNorm folded (device pointers)
struct PnWeights {
    float r1sq, r2sq;             // float32(0.2**2), float32(0.4**2): what `sqrdists > radius ** 2` compares against
    const float *sa1_w0t;         // [3][64]    sa1.mlp_convs.0 (kn)
    const float *sa1_b0;          // [64]
    const float *sa1_w1;          // [128][64]  sa1.mlp_convs.1 row-major
    const float *sa1_b1;          // [128]
    const float *sa2_wf_t;        // [128][128] sa2.mlp_convs.0[:, 3:] (kn)  -> U = F1 * wf_t + b
    const float *sa2_b0;          // [128]
    const float *sa2_vx;          // [3][128]   sa2.mlp_convs.0[:, 0:3] (kn)
    const float4 *sa2_w1_img;     // chain image of sa2.mlp_convs.1 [256 x 128]
    const float *sa2_b1;          // [256]
    const float4 *sa3_w_img;      // chain image of sa3.mlp_convs.0[:, 3:] [256 x 256]
    const float4 *sa3_w_img16;    // the same as a pack_chain_bf16 image (bf16 mode, z16_kernel)
    const float *sa3_wx;          // [3][256]   sa3.mlp_convs.0[:, 0:3] (kn)
    const float *sa3_b;           // [256]
};

// The same weights unrounded (float64 folds) for the float64 table build (pointnet64.hip)
struct PnWeights64 {
    const double *sa1_w0t, *sa1_b0, *sa1_w1, *sa1_b1;     // [3][64], [64], [128][64], [128]
    const double *sa2_wf_t, *sa2_b0, *sa2_vx;             // [128][128] (kn), [128], [3][128]
    const double *sa2_w1_img, *sa2_b1;                    // pack_mfma64 image of sa2.mlp_convs.1 [256 x 128], [256]
    const double *sa3_w_img, *sa3_wx, *sa3_b;             // pack_mfma64 image of sa3.mlp_convs.0[:, 3:] [256 x 256], [3][256], [256]
};

struct XobjChain {
    const float *xyz;             // [N][3]
    const int   *fps1;            // [N][512]
    const int   *slot_of_start;   // [N] start index -> table slot, or null (slot = start)
    const float *Z;               // [nv][N][256]
    const uint32_t *Z16, *M0_16;  // bf16 operand-order copies of Z and M0 ([.][128 dwords], mfma_chain.h) or null
    const int   *fps2;            // [N][128] FPS(128) sequences by start POINT, or null
    const int   *flags;           // [N] 1 = the sequence from this start point is order-dependent: run FPS for the row
    const int   *crowded;         // [N] 1 = more than 64 points in the centre's r=0.4 ball: Z depends on the variant
    const float *M0;              // [N][256] max of Z[0][c] over the non-crowded centres of fps2[q], or null (see m0_kernel)
    const int   *cl2;             // [N][128] crowded centres of fps2[q], padded with the last one
    const int   *cnt2;            // [N] their number
    int          N;
    // (chain, s1)-group kernel (xobj_group_kernel): the object's crowded centres, their slot in that list per fps2 sequence
    const int   *clist;           // [ncr] crowded centre point ids (crowd_kernel)
    const int   *cl2s;            // [N][128] cl2 as positions in clist (m0_kernel), padded like cl2
    const unsigned short *cl2o;   // [N][2][128] the same as byte offsets into xobj_rows_kernel's LDS slab (position * lpr * 16), even / odd positions first
    int          ncr;             // number of crowded centres (host copy)
    const int   *pcf;             // [N][512] q | cnt2[q] << 10 | (flags[q] != 0) << 18 with q = fps1[s1][s2] (pcf_kernel)
    int          lpr;             // lanes per row of the group kernel: 64 / 32 / 16 / 8  <=>  1 / 2 / 4 / 8 feature chunks
};

struct XobjParams {
    const XobjChain *chains;      // device array [nchain]
    const int       *starts;      // [nchain][R][2]  (s1, s2) per reference row
    const int       *order;       // [nchain][R] row ids of each chain sorted by (s1, s2), each with its s2 in bits 22.., or null (natural order)
    float           *xobj;        // [nchain][R][256]
    uint32_t        *xobj16;      // when set: gather from Z16 / M0_16 and write bf16 operand-order rows [nchain][R][128] here instead
    int64_t          R, total_rows;
    int              use_table;   // 0: always run the per-row FPS (test hook)
    int              skip_fast;   // set by pn_xobj: xobj_kernel handles only the rows listed in todo
    int             *todo;        // [total_rows] rows xobj_fast_kernel left to xobj_kernel, or null (then xobj_kernel does everything)
    int             *todo_count;  // device counter of that list
    int64_t          todo_capacity;
    const int       *group_off;   // [nchain][N+1] offsets of the s1-groups in `order` (rows of a chain sorted by s1), or null
    int              nchain, total_items;      // group kernel: total_items = nchain * group_N work items (chain rank, s1)
    int              group_N;                   // points per object = sa1 start values
    unsigned char    chain_of_rank[DGDM_MAX_CHAINS];   // the chains by descending number of crowded centres
};

// Per-object table of finished embeddings (xtab_kernel): X[s1][q] = max(M0[q], max over the crowded centres of fps2[q] of Z[s1][centre]) -
// everything a reference row contributes to the trunk depends on its two FPS draws only through (s1, q = fps1[s1][s2]).
struct XtabObj {
    const float *xyz; const int *fps1;            // [N][3], [N][512]
    const float *Z, *M0; const uint32_t *Z16, *M0_16;
    const int *clist, *ncr;                       // crowded centres (device count)
    const int *cl2s, *cnt2, *flags, *crowded;
    float *X; uint32_t *X16;                      // [N][N][256] float32 or [N][N][128] bf16 operand-order dwords (one of them)
    int N;
};
// builds X (or X16) for one object; start points whose FPS(128) sequence is order-dependent (flags) get their rows from a per-row FPS
int pn_xtab(const XtabObj &o, bool bf16, hipStream_t s);
// per reference row the row of X its embedding is: idx[chain][r] = s1 * N + q (or q alone for chains whose object has no crowded centre:
// their table is M0 itself)
struct XidxChain { const int *fps1; int N; int m0_only; };
int pn_xidx(const XidxChain *chains_dev, const int *starts, int64_t R, int nchain, int *idx, hipStream_t s);

// the index functions for arbitrary batches of clouds (C-ABI dgdm_farthest_point_sample / dgdm_query_ball_point / dgdm_square_distance / dgdm_index_points)
int pn_fps_rows(const float *xyz, const int *start, int B, int N, int npoint, int *out, hipStream_t s);
int pn_ball_rows(const float *xyz, const float *centres, int B, int N, int S, float r2, int nsample, int *out, hipStream_t s);
int pn_sqdist_rows(const float *src, const float *dst, int B, int S, int N, float *out, hipStream_t s);
int pn_index_rows(const float *points, const int *idx, int B, int N, int M, int C, float *out, hipStream_t s);

int pn_fps_table(const float *xyz, int N, int nv, int npoint, int *out, int *flags, hipStream_t s, int nobj = 1);
int pn_sa1(const float *xyz, int N, const PnWeights &w, float *F1, hipStream_t s, int nobj = 1);      // nobj > 1: pools [nobj][N][..]
// crowded/clist/ncr: centres whose ball holds > 64 points; off [N+1], pairs [<= N*N], rank [N][N]: the in-radius pair list (T4/T5)
int pn_crowd(const float *xyz, int N, const PnWeights &w, int *crowded, int *clist, int *ncr, int *off, int *pairs, short *rank, hipStream_t s,
             int *ncr_copy = nullptr, int nobj = 1);
// Y16 (optional): write bf16 operand-order rows there INSTEAD of the float32 rows (bf16 mode)
int pn_pairs(const float *xyz, int N, const float *U, const PnWeights &w, const int *pairs, const int *off, float *Y, uint32_t *Y16, hipStream_t s);
// crowded_mode 1: the variants >= 1 of the crowded centres by l2c_kernel (needs vlist = identity: slot v = start index v, and K <= 255
// points per ball); 0: l2_kernel throughout
int pn_l2(const float *xyz, int N, const PnWeights &w, const int *fps1, const int *vlist, int nv, const float *Y, float *L2,
          const int *clist, const int *ncr, const int *off, const short *rank, bool bf16 /* Y and L2 are bf16 operand-order rows */, hipStream_t s,
          int crowded_mode = 0);
// bf16 mode T6: bf16 contraction from L2_16 rows; writes the float32 rows and their bf16 copy
int pn_z16(const float *xyz, int N, int nv, const PnWeights &w, const uint32_t *L2_16, float *Z, uint32_t *Z16, const int *clist, const int *ncr,
           hipStream_t s);
// Z16 (optional): the same rows again in bf16 operand order
int pn_z(const float *xyz, int N, int nv, const PnWeights &w, const float *L2, float *Z, uint32_t *Z16, const int *clist, const int *ncr,
         hipStream_t s);
// ---- float64 table build (pointnet64.hip): same inputs, same float32 outputs (each rounded ONCE from a float64 accumulation)
// F1_64 [N][128] doubles (sa1 features); U is then linear64(F1_64) -> U64 [N][128] doubles
int pn_sa1_64(const float *xyz, int N, float r1sq, const PnWeights64 &w, double *F1_64, hipStream_t s, int nobj = 1);
int pn_pairs64(const float *xyz, int N, const double *U64, const PnWeights64 &w, const int *pairs, const int *off, float *Y, hipStream_t s);
int pn_z64(const float *xyz, int N, int nv, const PnWeights64 &w, const float *L2, float *Z, const int *clist, const int *ncr, hipStream_t s);
int pn_m0(const int *fps2, const int *crowded, int N, const float *Z0, float *M0, int *cl2, int *cnt2, const uint32_t *Z0_16, uint32_t *M0_16,
          const int *clist, const int *ncr, int *cl2s, unsigned short *cl2o, hipStream_t s);
// lanes per row for an object with `ncr` crowded centres (0 = the group kernel cannot hold its slab: use the per-row kernels):
// the largest feature chunk (lpr * 16 B per centre) whose slab fits 64 KiB, so that a slab offset is a 16-bit number
__host__ __device__ inline int xobj_rows_lpr(int ncr, bool bf16) {
    for (int lpr = bf16 ? 32 : 64; lpr >= 8; lpr >>= 1)
        if (ncr * lpr * 16 <= 65536) return lpr;
    return 0;
}
// one workgroup per (chain, s1): per feature chunk the variant's crowded Z rows staged once in LDS, all rows of the group reduced from there
int pn_xobj_groups(const XobjParams &p, hipStream_t s);
// the row-metadata table of xobj_rows_kernel: pcf [N][512]
int pn_pcf(const int *fps1, const int *cnt2, const int *flags, int N, int *pcf, hipStream_t s);
// index test hook (dgdm_debug_pointnet_indices): sa1's 32-neighbour lists [N][32], sa2's first-64 lists [N][64] + counts for the
// candidate order perm[0..M), crowded flags [N]; synchronises
int pn_debug_indices(const float *xyz, int N, const PnWeights &w, const int *perm, int M, int *ball1, int *ball2, int *ball2_cnt, int *crowded,
                     hipStream_t s);
// all_fast: every chain has its tables and no start point with an order-dependent FPS(128) sequence (then one kernel does it all)
int pn_xobj(const XobjParams &p, bool all_fast, hipStream_t s);

}  // namespace dgdm
