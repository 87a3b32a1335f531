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
    const float4 *Wf[8];      // chain images of W'
    const float  *bf[8];      // folded biases
    const float4 *Wb[8];      // chain images of W'^T
    int           n_mid;
    const float  *Wout;       // [3][256] row-major
    const float  *bout;       // [3]
    // 3-D only
    const float4 *W1o;        // image of W1'[:, object part]   [512 x 256]
    const float4 *W2f;        // image of W2'                   [256 x 512]
    const float  *b2;         // folded bias of layer 2
    const float4 *W2b;        // image of W2'^T                 [512 x 256]
    const float  *xobj;       // [nchain][R][256]  PointNet++ embedding per reference row
    // first-layer tables
    const float  *Atab;       // table mode: [nchain*B][W1]; rows mode: [rows][W1]
    const float  *Ptab;       // [C][W1]  (table mode)
    const TrunkObjective *obj;// [nchain]
    const float  *rowcoef;    // [nchain][R] or null
    float        *partial;    // [ntiles][W1]
    float        *logits;     // fwd-only: [nchain][R][3]
    int           B, C, tiles_per_b, ntiles;
    int64_t       R;          // rows per chain (B*C in table mode, rows in rows mode)
};

// kind: 2 | 3.  rows_mode: first-layer pre-activations given per row (general forward API).
int trunk_launch(int kind, bool rows_mode, bool fwd_only, const TrunkParams &p, hipStream_t s);

}  // namespace dgdm
