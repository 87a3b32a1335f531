#pragma once
#include "common.h"

namespace dgdm {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

// Y = act(X * WT + bias + rowbias[row / rb_div]) (+ Y).  WT stored [K][N].
int linear(const float *X, int ldx, const float *WT, const float *bias, const float *rowbias, int rb_div, float *Y, int ldy,
           int rows, int K, int N, int act, bool accumulate, hipStream_t s);
// out[((g * (W/32) + o) * 4 + q) * 64 + lane][0..3] = T[min(32 g + (lane & 31), rows - 1)][32 o + 8 q + 4 (lane >> 5) + 0..3]:
// a [rows][W] table re-tiled so that the trunk kernels read the 32 rows of a tile in their MFMA operand layout with one
// coalesced 1 KiB load per (block o, quarter q) instead of 64 scattered 16-byte pieces
int tile_table(const float *T, int rows, int W, float *out, hipStream_t s);
int pose_embed(const float *ori, const float *pos, float *out /*[rows][27]*/, int rows, hipStream_t s);
int time_embed(const float *t_dev /*or null*/, float t_scalar, const float *freqs, float *out, int rows, int half, hipStream_t s);
int gather_add(const float *a, const int *idx, const float *b, float *out, int groups, int N, hipStream_t s);
int dyn_post(int W1, const float *partial, int tiles_per_b, const float *w1c, const float *g2w, const float *g0w, const float *V,
             float *grad, int rows, int L, hipStream_t s);

}  // namespace dgdm
