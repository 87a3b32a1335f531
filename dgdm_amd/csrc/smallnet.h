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


// ---- the same small stages in float64 (DESIGN_HISTORY.md 4.9).  Everything that is evaluated once per finger, per pose cell or per object
// instead of once per replicated row costs nothing in double precision, and what it feeds - the first trunk layer's tables - then
// carries one float32 rounding instead of the error of a 256..795-term float32 dot product.
// Y = act(X * WT + bias + rowbias[row / rb_div]): X float32 (Xf) or float64 (Xd) rows, WT [K][N] / bias / rowbias float64; the result
// as float64 (Yd) and/or rounded once to float32 (Yf); either may be null.
int linear64(const float *Xf, const double *Xd, int ldx, const double *WT, const double *bias, const double *rowbias, int rb_div,
             double *Yd, float *Yf, int ldy, int rows, int K, int N, int act, hipStream_t s);
// out[g][n] = a[idx[g]][n] + b[n]  (float64)
int gather_add64(const double *a, const int *idx, const double *b, double *out, int groups, int N, hipStream_t s);
// dyn_post with float64 sums (tile partials are float32 inputs); V64 = the float64 hidden layer of the gripper encoder
int dyn_post64(int W1, const float *partial, int tiles_per_b, const double *w1c, const double *g2w, const double *g0w, const double *V64,
               float *grad, int rows, int L, hipStream_t s);

}  // namespace dgdm
