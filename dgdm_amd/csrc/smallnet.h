#pragma once
#include "common.h"

namespace dgdm {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

// Y = act(X * WT + bias + rowbias[row / rb_div]) (+ Y).  WT stored [K][N].
int linear(const float *X, int ldx, const float *WT, const float *bias, const float *rowbias, int rb_div, float *Y, int ldy,
           int rows, int K, int N, int act, bool accumulate, hipStream_t s);
int pose_embed(const float *ori, const float *pos, float *out /*[rows][27]*/, int rows, hipStream_t s);
int time_embed(const float *t_dev /*or null*/, float t_scalar, const float *freqs, float *out, int rows, int half, hipStream_t s);
int gather_add(const float *a, const int *idx, const float *b, float *out, int groups, int N, hipStream_t s);
int dyn_post(int W1, const float *partial, int tiles_per_b, const float *w1c, const float *g2w, const float *g0w, const float *V,
             float *grad, int rows, int L, hipStream_t s);

}  // namespace dgdm
