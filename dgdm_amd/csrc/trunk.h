#pragma once
#include "common.h"

namespace dgdm {

struct TrunkObjective {   // device copy of DgdmObjective (object index not needed on the device)
    float lin[3];
    float quad[3];
    int   use_rowcoef;
    int   pad;
};

struct TrunkParams {
    // 256 -> 256 layers (trunk layers 2..8 for 2-D, 3..8 for 3-D), BatchNorm folded
    const float  *bf[8];      // folded biases
    int           n_mid;
    // continuous weight streams (mfma_chain.h stream_cont), in consumption order:
    //   forward : [3-D: 16 x (W1o block (32 entries) | W2' column block (32 entries))] then the n_mid images of W'
    //   backward: the n_mid images of W'^T, LAST layer first, [3-D: then the 16 blocks of W2'^T]
    const float4 *Wfwd, *Wbwd;
    unsigned      fwd_bytes, bwd_bytes;
    const float  *Wout;       // [3][256] row-major
    const float  *bout;       // [3]
    // 3-D only
    const float  *b2;         // folded bias of layer 2
    const float  *xobj;       // [nchain][R][256]  PointNet++ embedding per reference row
    const uint32_t *xobj16;   // bf16 trunk only: the same rows in bf16 operand order [nchain][R][128 dwords] (mfma_chain.h), or null
    // table mode, 3-D: instead of materialised rows, the per-object embedding tables (pointnet.h xtab): row r of chain c is
    // xtab[c] + xidx[c * R + r] * 256 floats (xtab16: * 128 dwords).  Null: read xobj / xobj16.
    const float *const *xtab;
    const uint32_t *const *xtab16;
    const int    *xidx;
    int64_t       xstride;    // rows per chain in xobj / xobj16 / xidx (>= R: a launch may read one denoise step's rows out of buffers that hold all steps')
    // first-layer tables
    const float  *Atab;       // table mode: [nchain*B][W1]; rows mode: [rows][W1]
    const float  *Ptab;       // [C][W1]  (table mode)
    const float  *PtabT;      // the same, tiled per 32 cells in operand layout (smallnet.h tile_table)
    const float  *Pmax;       // [C] largest magnitude of a cell's row of Ptab (trunk_f16l.hip: the f16 scale of 3-D layer 2's input); may be null elsewhere
    const TrunkObjective *obj;// [nchain]
    const float  *rowcoef;    // [nchain][R] or null
    float        *partial;    // [ntiles][W1]
    float        *logits;     // fwd-only: [nchain][R][3]
    int           B, C, tiles_per_b, ntiles;
#ifdef DGDM_TRUNK_CLOCKS
    long long    *clk;        // experiment build only: [waves][8] phase time stamps (scripts/README: -DDGDM_TRUNK_CLOCKS)
#endif
    int64_t       R;          // rows per chain (B*C in table mode, rows in rows mode)
};

// kind: 2 | 3.  rows_mode: first-layer pre-activations given per row (general forward API).
int trunk_launch(int kind, bool rows_mode, bool fwd_only, const TrunkParams &p, hipStream_t s);

// float32 contractions as three f16 MFMA products on two-way split, power-of-two scaled operands (trunk_f16l.hip): table mode, forward +
// backward.  p.Wfwd / p.Wbwd point at the f16 streams (DgdmDynamics::fill_trunk_f16), sc carries the weight matrices' scale exponents.
struct TrunkF16Scales {
    int ew_mid[8];            // 256 -> 256 stack layer l (the same for W and its transpose)
    int ew_l1, ew_l2;         // 3-D: layer 1's object-embedding columns; layer 2 (the same for its transpose, the last layer back)
    float l1_norm1;           // 3-D: largest absolute row sum of layer 1's object-embedding columns (bounds a row of layer 1 from its input)
};
// (the weight stream is shared by the workgroup's four waves through LDS: trunk_f16l.hip)
int trunk_f16l_launch(int kind, const TrunkParams &p, const TrunkF16Scales &sc, hipStream_t s);

// bf16-contraction variant (trunk_bf16.hip): table mode, forward + backward only.  p.Wfwd / p.Wbwd point at the bf16 streams
// (DgdmDynamics::fill_trunk_bf16); everything else in TrunkParams means the same.
int trunk_bf16_launch(int kind, const TrunkParams &p, hipStream_t s);

}  // namespace dgdm
