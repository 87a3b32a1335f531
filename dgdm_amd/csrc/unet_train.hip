// Training of the eps-net on gfx950: Diffusion.get_stats / training_step (generator/diffusion.py:126-177) for the
// ConditionalUnet1D of generator/train.py:80 (generator/diffusion_utils.py:123-285), torch.optim.Adam (diffusion.py:711-714) and the
// EMA copy (diffusers EMAModel.step, diffusion.py:716-724).  SURVEY.md 8(f) rank 4.
//
// The sampling kernel (unet.hip) keeps one sample's activations in LDS; training cannot: every weight gradient is a contraction over
// (sample, position) - 43 008 rows at batch 1024, L = 42 - so the training path is GEMM-shaped and runs layer by layer with the
// activations resident in HBM (1.5 GB at batch 2048: nothing on a 288 GB part).
//
// Layout.  Activations are channels-last, one row per (sample, position), with zero padding rows around every sample: level 1
// (L positions) has 4 padding rows on either side, level 2 (L/2 positions) has 2, so that level-1 row 2r + j is position 2q + j of the
// sample whose level-2 row r is position q.  Then EVERY convolution of the network is a plain GEMM over rows whose A operand row is a
// contiguous WINDOW of the input buffer:
//     Conv1d k (stride 1)      Y[r]      = W_f [k Cin -> Cout] . X[r - k/2 .. r + k/2]          window stride Cin   (rows overlap)
//     its input gradient       dX[r]     = W_b [k Cout -> Cin] . dY[r - k/2 .. r + k/2]         taps reversed
//     Conv1d k=3 stride 2      Y2[r]     = W_f . X1[2r - 1 .. 2r + 1]                            window stride 2 Cin
//     ConvTranspose1d k=4 s=2  Y1[2r]    = (W_3, W_1) . X2[r - 1, r],  Y1[2r + 1] = (W_2, W_0) . X2[r, r + 1]
//     every weight gradient    dW_f[kk][n] = sum_r window(r)[kk] * dY[r][n]                     contraction over the rows, split + ordered sum
// (padding rows hold zeros in every activation and gradient buffer, which IS the convolutions' zero padding), cat((x, h), dim=1)
// is two windows accumulated into one output, and the Linear layers are the k = 1 case on one row per sample.
// Two kernels carry all of it on v_mfma_f32_32x32x2_f32 (exact float32, as the reference trains): rowgemm_kernel (128 x 128 output tile,
// windows -> LDS transposed, weight image [K][N] -> LDS) and colgemm_kernel (contraction over rows, both operands copied to LDS as they
// lie).  GroupNorm + Mish (+ FiLM, + residual add) is one workgroup per sample forward and backward; per-sample partial sums of the
// GroupNorm / FiLM / bias gradients and the split weight-gradient tiles are added in a fixed order (float64), so a step is reproducible
// bit for bit.  No atomics.
#include "common.h"
#include "mfma_chain.h"
#include "train_gemm.h"
#include <cmath>
#include <cstring>
#include <memory>
#include <algorithm>
#include <array>

namespace dgdm {
namespace {

// torch.nn.functional.mish and its derivative (torch: grad * (tanh(sp) + x * sigmoid(x) * (1 - tanh(sp)^2)), sp = softplus(x))
__device__ __forceinline__ void mish_both(float x, float &y, float &dy) {
    const float e = expf(fminf(x, 20.f));
    const float nn = e * (e + 2.f);
    const float th = nn / (nn + 2.f);
    const float sg = 1.f / (1.f + expf(-x));
    y = x * th;
    dy = th + x * sg * (1.f - th * th);
}
__device__ __forceinline__ float mish_f(float x) {
    const float e = expf(fminf(x, 20.f));
    const float nn = e * (e + 2.f);
    return x * (nn / (nn + 2.f));
}
__global__ void mish_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = mish_f(x[i]);
}
// dx (+)= dy * mish'(x)
__global__ void mish_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ dx, int64_t n, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float y, d;
    mish_both(x[i], y, d);
    const float v = dy[i] * d;
    dx[i] = accumulate ? dx[i] + v : v;
}

// ---- GroupNorm(8) -> Mish (-> FiLM) (+ residual), one workgroup per sample (diffusion_utils.py:57-72, 108-120).  x, y, res are
// [rows][C] with the sample's valid rows at r0 = s * rp + pad; film [S][2C] = (scale | bias) of cond_encoder; stats [S][8][2] = mean, rstd.
struct GnArgs {
    const float *x; float *y; const float *res; const float *gamma, *beta; const float *film; float *stats;
    int C, rp, pad, lv; float eps;
};
template <int C>
__device__ __forceinline__ float group_total(float v, float (*red)[256], int tid) {       // sum over the group's channels and the row lanes
    constexpr int CG = C / 8, RL = 256 / C;
#pragma unroll
    for (int o = CG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (RL > 1) {
        __syncthreads();
        red[0][tid] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RL; ++k) t += red[0][(tid % C) + k * C];
        v = t;
    }
    return v;
}
template <int C>
__global__ __launch_bounds__(256) void gn_act_kernel(const GnArgs a) {
    __shared__ float red[1][256];
    constexpr int CG = C / 8, RL = 256 / C;
    const int tid = threadIdx.x, c = tid % C, rl = tid / C, s = blockIdx.x;
    const int64_t r0 = (int64_t)s * a.rp + a.pad;
    const float cnt = (float)(CG * a.lv);
    float s1 = 0.f;
    for (int p = rl; p < a.lv; p += RL) s1 += a.x[(r0 + p) * C + c];
    const float mean = group_total<C>(s1, red, tid) / cnt;
    float s2 = 0.f;
    for (int p = rl; p < a.lv; p += RL) { const float d = a.x[(r0 + p) * C + c] - mean; s2 = fmaf(d, d, s2); }
    const float var = group_total<C>(s2, red, tid) / cnt;
    const float rstd = 1.f / sqrtf(var + a.eps);
    if (rl == 0 && (c % CG) == 0) { a.stats[((int64_t)s * 8 + c / CG) * 2 + 0] = mean; a.stats[((int64_t)s * 8 + c / CG) * 2 + 1] = rstd; }
    const float ga = a.gamma[c] * rstd, be = a.beta[c];
    const float fs = a.film ? a.film[(int64_t)s * 2 * C + c] : 1.f, fb = a.film ? a.film[(int64_t)s * 2 * C + C + c] : 0.f;
    for (int p = rl; p < a.lv; p += RL) {
        const int64_t o = (r0 + p) * C + c;
        float v = mish_f((a.x[o] - mean) * ga + be);
        if (a.film) v = fs * v + fb;
        if (a.res) v += a.res[o];
        a.y[o] = v;
    }
}
// backward of the same: dy -> dx, per-sample partials spart[s] = [dgamma (C) | dbeta (C)], dfilm[s] = [dscale (C) | dbias (C)]
struct GnBwdArgs {
    const float *dy; const float *x; float *dx; const float *gamma, *beta; const float *film; const float *stats; float *spart; float *dfilm;
    int C, rp, pad, lv;
};
template <int C>
__global__ __launch_bounds__(256) void gn_act_bwd_kernel(const GnBwdArgs a) {
    __shared__ float red[4][256];
    constexpr int CG = C / 8, RL = 256 / C;
    const int tid = threadIdx.x, c = tid % C, rl = tid / C, s = blockIdx.x, grp = c / CG;
    const int64_t r0 = (int64_t)s * a.rp + a.pad;
    const float mean = a.stats[((int64_t)s * 8 + grp) * 2 + 0], rstd = a.stats[((int64_t)s * 8 + grp) * 2 + 1];
    const float gam = a.gamma[c], be = a.beta[c];
    const float fs = a.film ? a.film[(int64_t)s * 2 * C + c] : 1.f;
    float dga = 0.f, dbe = 0.f, dsc = 0.f, dbi = 0.f;
    for (int p = rl; p < a.lv; p += RL) {
        const int64_t o = (r0 + p) * C + c;
        const float xh = (a.x[o] - mean) * rstd, g = fmaf(gam, xh, be), d = a.dy[o];
        float y, dm;
        mish_both(g, y, dm);
        dsc = fmaf(d, y, dsc); dbi += d;
        const float dg = d * fs * dm;
        dga = fmaf(dg, xh, dga); dbe += dg;
    }
    if (RL > 1) {       // add the row lanes (fixed order)
        red[0][tid] = dga; red[1][tid] = dbe; red[2][tid] = dsc; red[3][tid] = dbi;
        __syncthreads();
        dga = dbe = dsc = dbi = 0.f;
#pragma unroll
        for (int k = 0; k < RL; ++k) { dga += red[0][c + k * C]; dbe += red[1][c + k * C]; dsc += red[2][c + k * C]; dbi += red[3][c + k * C]; }
        __syncthreads();
    }
    if (rl == 0) {
        a.spart[(int64_t)s * 2 * C + c] = dga; a.spart[(int64_t)s * 2 * C + C + c] = dbe;
        if (a.film) { a.dfilm[(int64_t)s * 2 * C + c] = dsc; a.dfilm[(int64_t)s * 2 * C + C + c] = dbi; }
    }
    // group means of dxh = dg * gamma and of dxh * xh:  sum_c gamma_c dbeta_c, sum_c gamma_c dgamma_c
    float m1 = gam * dbe, m2 = gam * dga;
#pragma unroll
    for (int o = CG / 2; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
    const float cnt = (float)(CG * a.lv);
    m1 /= cnt; m2 /= cnt;
    for (int p = rl; p < a.lv; p += RL) {
        const int64_t o = (r0 + p) * C + c;
        const float xh = (a.x[o] - mean) * rstd, g = fmaf(gam, xh, be);
        float y, dm;
        mish_both(g, y, dm);
        const float dxh = a.dy[o] * fs * dm * gam;
        a.dx[o] = rstd * (dxh - m1 - xh * m2);
    }
}

// noisy sample (DDIMScheduler.add_noise, diffusion.py:144-148) into the padded level-1 layout, SinusoidalPosEmb of the timestep
__global__ void prep_kernel(const float *__restrict__ x0, const float *__restrict__ noise, const float *__restrict__ sa, const float *__restrict__ sb,
                            const int64_t *__restrict__ t, const float *__restrict__ freqs, int S, int L, int rp, int pad, int dsed, float *__restrict__ X0,
                            float *__restrict__ emb) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int W = L + dsed;
    if (e >= (int64_t)S * W) return;
    const int s = (int)(e / W), k = (int)(e - (int64_t)s * W);
    if (k < L) {
        X0[(int64_t)s * rp + pad + k] = add_rn(mul_rn(sa[s], x0[(int64_t)s * L + k]), mul_rn(sb[s], noise[(int64_t)s * L + k]));
    } else {
        const int j = k - L, half = dsed / 2;
        const float arg = (float)t[s] * freqs[j % half];
        emb[(int64_t)s * dsed + j] = j < half ? sinf(arg) : cosf(arg);
    }
}
// F.mse_loss(noise_pred, noise) (diffusion.py:164): per-sample sums of squares, d loss / d pred = 2 (pred - noise) / (S_total L)
__global__ void loss_kernel(const float *__restrict__ pred, const float *__restrict__ noise, int S, int L, int rp, int pad, float inv, float *__restrict__ dpred,
                            float *__restrict__ part, float *__restrict__ pred_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    float a = 0.f;
    for (int k = 0; k < L; ++k) {
        const float p = pred[(int64_t)s * rp + pad + k], d = p - noise[(int64_t)s * L + k];
        a = fmaf(d, d, a);
        dpred[(int64_t)s * rp + pad + k] = d * inv;
        if (pred_out) pred_out[(int64_t)s * L + k] = p;
    }
    part[s] = a;
}
__global__ void loss_finish_kernel(const float *__restrict__ part, int S, double denom, float *__restrict__ loss) {
    if (threadIdx.x || blockIdx.x) return;
    double a = 0.0;
    for (int s = 0; s < S; ++s) a += (double)part[s];
    *loss = (float)(a / denom);
}

// diffusers EMAModel.step: ema.mul_(decay); ema.add_(param, alpha = 1 - decay)
__global__ void ema_kernel(float *__restrict__ e, const float *__restrict__ p, int64_t n, float decay, float one_minus) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    e[i] = add_rn(mul_rn(e[i], decay), mul_rn(one_minus, p[i]));
}
}  // namespace
}  // namespace dgdm

using namespace dgdm;

struct DgdmUnetTrainer {
    // ---- parameters: one flat buffer in the state_dict's own tensor layouts; `named` maps the reference's keys to it
    struct Named { std::string name; size_t off; int64_t numel; };
    std::vector<Named> named;
    size_t n_params = 0;
    DevBuf P, G, M1, V, E /* EMA copy */, IMG, descs_dev, ws, wpart, freqs, loss_dev;
    std::vector<ImgDesc> descs;
    size_t n_img = 0;
    int max_img_elems = 0;
    float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, wd = 0.f;
    int64_t adam_steps = 0;
    int L = 0, d0 = 128, d1 = 256, dsed = 32, ks = 5;
    // ---- graph
    struct T { float *v = nullptr, *g = nullptr; int C = 0, lvl = 0; bool gw = false; };     // lvl 0: one row per sample; 1, 2: padded positions
    enum { S1 = 0, DOWN = 1, UP = 2 };
    struct Conv { int kind = S1, k = 1, cin = 0, cout = 0, nparts = 1, pc[2] = {0, 0}; size_t w = 0, b = 0; int F[2] = {-1, -1}, B[2] = {-1, -1}; };
    struct Res {
        Conv c1, c2, rc, cond; bool has_rc = false; size_t g1 = 0, g2 = 0; int cout = 0, lvl = 1;
        T t_c1, t_a, t_c2, t_rc, t_film, t_out; float *st1 = nullptr, *st2 = nullptr, *sp = nullptr;
    };
    Res res[8];
    Conv se1, se3, down, up, fconv, oconv;
    size_t fg = 0;
    T t_x0, t_emb, t_h1, t_a1, t_gf, t_mg, t_ds, t_us, t_cf, t_af, t_pred;
    float *f_st = nullptr, *f_sp = nullptr, *lpart = nullptr, *cpart = nullptr;
    int64_t S_ws = 0, M[3] = {0, 0, 0};
    RowMask mk[3];
    int64_t wpart_floats = 0;

    float *p(size_t o) const { return P.as<float>() + o; }
    float *gr(size_t o) const { return G.as<float>() + o; }
    size_t add_param(const std::string &name, int64_t numel) { named.push_back({name, n_params, numel}); const size_t o = n_params; n_params += (size_t)numel; return o; }
    int add_img(size_t w_off, int Kblk, std::initializer_list<int> taps, int N, int s_kc, int s_n, int base);
    void make_conv(Conv &c, const std::string &name, int kind, int k, int cin, int cout, int nparts = 1);
    int reserve(int S);
    int rowgemm(const float *A, int64_t a_rs, int img, float *C, int64_t c_rs, const float *add, int64_t add_rs, const float *bias, int64_t Mrows, int lvl,
                hipStream_t s) const;
    int colgemm(const float *A, int64_t a_rs, int img, const float *D, int64_t d_rs, int64_t Mrows, hipStream_t s);
    int bias_grad(const float *D, int64_t rs, int64_t Mrows, int N, size_t b_off, hipStream_t s);
    int conv_fwd(const Conv &c, T *const *parts, T &y, hipStream_t s) const;
    int conv_bwd(const Conv &c, T *const *parts, T &y, hipStream_t s);
    int gn_fwd(const T &x, T &y, const T *res_t, size_t gb, const T *film, float *stats, hipStream_t s) const;
    int gn_bwd(T &x, const T &y, size_t gb, const T *film, T *dfilm, const float *stats, float *spart, int S, hipStream_t s);
    int res_fwd(Res &r, T *const *parts, hipStream_t s);
    int res_bwd(Res &r, T *const *parts, int S, hipStream_t s);
    int run(const float *x0, const float *noise, const float *sa, const float *sb, const int64_t *t, int S, int64_t S_total, bool backward, float *pred_out,
            float *loss_host, hipStream_t s);
    int adam(float lr, hipStream_t s);
    int repack(hipStream_t s);
    int copy_state(int which, DgdmTensor *t, int n, bool to_device);
};

int DgdmUnetTrainer::add_img(size_t w_off, int Kblk, std::initializer_list<int> taps, int N, int s_kc, int s_n, int base) {
    ImgDesc d{};
    d.src = (int64_t)w_off + base; d.dst = (int64_t)n_img;
    d.Kblk = Kblk; d.ntaps = (int)taps.size();
    int i = 0;
    for (int t : taps) d.taps[i++] = t;
    d.K = Kblk * d.ntaps; d.Kp = round_up(d.K, KC); d.N = N; d.Np = round_up(N, TN); d.s_kc = s_kc; d.s_n = s_n;
    n_img += (size_t)d.Kp * d.Np;
    max_img_elems = std::max(max_img_elems, d.Kp * d.Np);
    descs.push_back(d);
    return (int)descs.size() - 1;
}

// Conv1d weight [cout][cin][k] (Linear: k = 1), Downsample1d's Conv1d(k = 3, stride 2, padding 1), Upsample1d's ConvTranspose1d
// weight [cin][cout][4] (stride 2, padding 1): the forward / input-gradient images of the file header
void DgdmUnetTrainer::make_conv(Conv &c, const std::string &name, int kind, int k, int cin, int cout, int nparts) {
    c.kind = kind; c.k = k; c.cin = cin; c.cout = cout; c.nparts = nparts;
    c.w = add_param(name + ".weight", (int64_t)cin * cout * k);
    c.b = add_param(name + ".bias", cout);
    if (kind == S1) {
        const int pcin = cin / nparts;
        for (int p = 0; p < nparts; ++p) {
            c.pc[p] = pcin;
            const int base = p * pcin * k;
            if (k == 5) { c.F[p] = add_img(c.w, pcin, {0, 1, 2, 3, 4}, cout, k, cin * k, base); c.B[p] = add_img(c.w, cout, {4, 3, 2, 1, 0}, pcin, cin * k, k, base); }
            else { c.F[p] = add_img(c.w, pcin, {0}, cout, k, cin * k, base); c.B[p] = add_img(c.w, cout, {0}, pcin, cin * k, k, base); }
        }
    } else if (kind == DOWN) {
        c.pc[0] = cin;
        c.F[0] = add_img(c.w, cin, {0, 1, 2}, cout, 3, cin * 3, 0);
        c.B[0] = add_img(c.w, cout, {1}, cin, cin * 3, 3, 0);              // even input rows 2m: tap 1 of output m
        c.B[1] = add_img(c.w, cout, {2, 0}, cin, cin * 3, 3, 0);           // odd rows 2m + 1: tap 2 of output m, tap 0 of output m + 1
    } else {
        c.pc[0] = cin;
        c.F[0] = add_img(c.w, cin, {3, 1}, cout, cout * 4, 4, 0);          // even outputs 2m: inputs (m - 1, m)
        c.F[1] = add_img(c.w, cin, {2, 0}, cout, cout * 4, 4, 0);          // odd outputs 2m + 1: inputs (m, m + 1)
        c.B[0] = add_img(c.w, cout, {0, 1, 2, 3}, cin, 4, cout * 4, 0);    // input m: outputs 2m - 1 .. 2m + 2
    }
}

int DgdmUnetTrainer::reserve(int S) {
    if (S <= S_ws) return DGDM_OK;
    const int rp1 = L + 8, rp2 = L / 2 + 4;
    M[0] = S; M[1] = (int64_t)S * rp1; M[2] = (int64_t)S * rp2;
    mk[0] = RowMask{0, 0, 0}; mk[1] = RowMask{rp1, 4, L}; mk[2] = RowMask{rp2, 2, L / 2};
    std::vector<std::pair<float **, int64_t>> want;
    auto tensor = [&](T &t, int C, int lvl, bool grad = true) {
        t.C = C; t.lvl = lvl;
        want.push_back({&t.v, M[lvl] * C + 2 * GUARD});
        if (grad) want.push_back({&t.g, M[lvl] * C + 2 * GUARD}); else t.g = nullptr;
    };
    auto plain = [&](float *&q, int64_t n) { want.push_back({&q, n}); };
    tensor(t_x0, 1, 1, false); tensor(t_emb, dsed, 0, false); tensor(t_h1, 4 * dsed, 0); tensor(t_a1, 4 * dsed, 0); tensor(t_gf, dsed, 0); tensor(t_mg, dsed, 0);
    for (int i = 0; i < 8; ++i) {
        Res &r = res[i];
        tensor(r.t_c1, r.cout, r.lvl); tensor(r.t_a, r.cout, r.lvl); tensor(r.t_c2, r.cout, r.lvl); tensor(r.t_out, r.cout, r.lvl);
        if (r.has_rc) tensor(r.t_rc, r.cout, r.lvl, false);      // its gradient is t_out's
        tensor(r.t_film, 2 * r.cout, 0);
        plain(r.st1, (int64_t)S * 16); plain(r.st2, (int64_t)S * 16); plain(r.sp, (int64_t)S * 2 * r.cout);
    }
    tensor(t_ds, d0, 2); tensor(t_us, d0, 1); tensor(t_cf, d0, 1); tensor(t_af, d0, 1); tensor(t_pred, 1, 1);
    plain(f_st, (int64_t)S * 16); plain(f_sp, (int64_t)S * 2 * d0); plain(lpart, S);
    plain(cpart, ((M[1] + CS_ROWS - 1) / CS_ROWS) * 512);
    int64_t total = 0;
    for (auto &w : want) total += (w.second + 63) / 64 * 64;
    int rc = ws.alloc((size_t)total * sizeof(float));
    if (rc) return rc;
    DGDM_HIP_CHECK(hipMemset(ws.p, 0, (size_t)total * sizeof(float)));      // padding rows and guards are zero and stay zero
    float *q = ws.as<float>();
    for (auto &w : want) { *w.first = q; q += (w.second + 63) / 64 * 64; }
    // tensors: skip the front guard
    auto fix = [&](T &t) { t.v += GUARD; if (t.g) t.g += GUARD; };
    for (T *t : {&t_x0, &t_emb, &t_h1, &t_a1, &t_gf, &t_mg, &t_ds, &t_us, &t_cf, &t_af, &t_pred}) fix(*t);
    for (int i = 0; i < 8; ++i) { Res &r = res[i]; for (T *t : {&r.t_c1, &r.t_a, &r.t_c2, &r.t_out, &r.t_film}) fix(*t); if (r.has_rc) fix(r.t_rc); }
    // weight-gradient partials: the largest (tiles x splits) product
    wpart_floats = (int64_t)512 * TM * TN + (int64_t)64 * TM * TN;
    if ((rc = wpart.alloc((size_t)wpart_floats * sizeof(float)))) return rc;
    S_ws = S;
    return DGDM_OK;
}

int DgdmUnetTrainer::rowgemm(const float *A, int64_t a_rs, int img, float *C, int64_t c_rs, const float *add, int64_t add_rs, const float *bias, int64_t Mrows,
                             int lvl, hipStream_t s) const {
    const ImgDesc &d = descs[img];
    RowGemm g{};
    g.A = A; g.a_rs = a_rs; g.B = IMG.as<float>() + d.dst; g.Kp = d.Kp; g.Np = d.Np; g.C = C; g.c_rs = c_rs; g.N = d.N; g.add = add; g.add_rs = add_rs;
    g.bias = bias; g.M = Mrows; g.mk = mk[lvl]; g.scalar_a = (a_rs & 3) != 0 || (reinterpret_cast<uintptr_t>(A) & 15) != 0;
    hipLaunchKernelGGL(rowgemm_kernel, dim3((unsigned)((Mrows + TM - 1) / TM), (unsigned)(d.Np / TN)), dim3(256), 0, s, g);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// dW in the layout of image `img` = sum over the rows of window(r) (x) D[r], scattered into the gradient of the image's tensor
int DgdmUnetTrainer::colgemm(const float *A, int64_t a_rs, int img, const float *D, int64_t d_rs, int64_t Mrows, hipStream_t s) {
    const ImgDesc &d = descs[img];
    const int kt = (d.Kp + TM - 1) / TM, nt = d.Np / TN;
    int64_t splits = std::min<int64_t>(std::max<int64_t>(1, Mrows / 256), std::max(1, 512 / (kt * nt)));
    int64_t per = ((Mrows + splits - 1) / splits + KC - 1) / KC * KC;
    splits = (Mrows + per - 1) / per;
    ColGemm g{};
    g.A = A; g.a_rs = a_rs; g.D = D; g.d_rs = d_rs; g.part = wpart.as<float>(); g.ldp = nt * TN; g.split_stride = (int64_t)kt * TM * g.ldp;
    g.M = Mrows; g.m_per_split = per;
    g.scalar_a = (a_rs & 3) != 0 || (reinterpret_cast<uintptr_t>(A) & 15) != 0;
    g.scalar_d = (d_rs & 3) != 0 || (reinterpret_cast<uintptr_t>(D) & 15) != 0;
    DGDM_REQUIRE(splits * g.split_stride <= wpart_floats, DGDM_EINVAL, "unet trainer: weight-gradient partials do not fit");
    hipLaunchKernelGGL(colgemm_kernel, dim3(kt, nt, (unsigned)splits), dim3(256), 0, s, g);
    DGDM_HIP_CHECK(hipGetLastError());
    const int64_t n = (int64_t)d.K * d.N;
    hipLaunchKernelGGL(wgrad_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wpart.as<float>(), (int)splits, g.split_stride, g.ldp,
                       descs_dev.as<ImgDesc>(), img, G.as<float>());
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int DgdmUnetTrainer::bias_grad(const float *D, int64_t rs, int64_t Mrows, int N, size_t b_off, hipStream_t s) {
    const int64_t blocks = (Mrows + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks, (N + 63) / 64), dim3(256), 0, s, D, rs, Mrows, N, (int64_t)CS_ROWS, cpart);
    DGDM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(rows_sum_kernel, rows_sum_grid(N), dim3(256), 0, s, cpart, blocks, N, gr(b_off));
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int DgdmUnetTrainer::conv_fwd(const Conv &c, T *const *parts, T &y, hipStream_t s) const {
    int rc;
    if (c.kind == S1) {
        for (int p = 0; p < c.nparts; ++p) {
            const T &x = *parts[p];
            if ((rc = rowgemm(x.v - (int64_t)(c.k / 2) * x.C, x.C, c.F[p], y.v, c.cout, p ? y.v : nullptr, c.cout, p ? nullptr : this->p(c.b), M[y.lvl], y.lvl, s)))
                return rc;
        }
    } else if (c.kind == DOWN) {       // y (level 2) row m <- x (level 1) rows 2m - 1 .. 2m + 1
        const T &x = *parts[0];
        if ((rc = rowgemm(x.v - x.C, 2 * x.C, c.F[0], y.v, c.cout, nullptr, 0, this->p(c.b), M[2], 2, s))) return rc;
    } else {                           // y (level 1) rows 2m, 2m + 1 <- x (level 2) rows (m - 1, m), (m, m + 1)
        const T &x = *parts[0];
        if ((rc = rowgemm(x.v - x.C, x.C, c.F[0], y.v, 2 * c.cout, nullptr, 0, this->p(c.b), M[2], 2, s))) return rc;
        if ((rc = rowgemm(x.v, x.C, c.F[1], y.v + c.cout, 2 * c.cout, nullptr, 0, this->p(c.b), M[2], 2, s))) return rc;
    }
    return DGDM_OK;
}

// input gradients (accumulated into x.g when it already holds one), weight and bias gradients
int DgdmUnetTrainer::conv_bwd(const Conv &c, T *const *parts, T &y, hipStream_t s) {
    int rc;
    if (c.kind == S1) {
        for (int p = 0; p < c.nparts; ++p) {
            T &x = *parts[p];
            if (x.g) {
                if ((rc = rowgemm(y.g - (int64_t)(c.k / 2) * c.cout, c.cout, c.B[p], x.g, x.C, x.gw ? x.g : nullptr, x.C, nullptr, M[y.lvl], y.lvl, s))) return rc;
                x.gw = true;
            }
            if ((rc = colgemm(x.v - (int64_t)(c.k / 2) * x.C, x.C, c.F[p], y.g, c.cout, M[y.lvl], s))) return rc;
        }
        return bias_grad(y.g, c.cout, M[y.lvl], c.cout, c.b, s);
    }
    T &x = *parts[0];
    if (c.kind == DOWN) {
        if ((rc = rowgemm(y.g, c.cout, c.B[0], x.g, 2 * x.C, x.gw ? x.g : nullptr, 2 * x.C, nullptr, M[2], 2, s))) return rc;
        if ((rc = rowgemm(y.g, c.cout, c.B[1], x.g + x.C, 2 * x.C, x.gw ? x.g + x.C : nullptr, 2 * x.C, nullptr, M[2], 2, s))) return rc;
        x.gw = true;
        if ((rc = colgemm(x.v - x.C, 2 * x.C, c.F[0], y.g, c.cout, M[2], s))) return rc;
        return bias_grad(y.g, c.cout, M[2], c.cout, c.b, s);
    }
    if ((rc = rowgemm(y.g - c.cout, 2 * c.cout, c.B[0], x.g, x.C, x.gw ? x.g : nullptr, x.C, nullptr, M[2], 2, s))) return rc;
    x.gw = true;
    if ((rc = colgemm(y.g - c.cout, 2 * c.cout, c.B[0], x.v, x.C, M[2], s))) return rc;      // dW[ci][co][k] in the layout of the input-gradient image
    return bias_grad(y.g, c.cout, M[1], c.cout, c.b, s);
}

int DgdmUnetTrainer::gn_fwd(const T &x, T &y, const T *res_t, size_t gb, const T *film, float *stats, hipStream_t s) const {
    GnArgs a{};
    a.x = x.v; a.y = y.v; a.res = res_t ? res_t->v : nullptr; a.gamma = p(gb); a.beta = p(gb) + x.C; a.film = film ? film->v : nullptr; a.stats = stats;
    a.C = x.C; a.rp = mk[x.lvl].rp; a.pad = mk[x.lvl].pad; a.lv = mk[x.lvl].lv; a.eps = 1e-5f;
    if (x.C == 128) hipLaunchKernelGGL(gn_act_kernel<128>, dim3((unsigned)M[0]), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gn_act_kernel<256>, dim3((unsigned)M[0]), dim3(256), 0, s, a);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int DgdmUnetTrainer::gn_bwd(T &x, const T &y, size_t gb, const T *film, T *dfilm, const float *stats, float *spart, int S, hipStream_t s) {
    GnBwdArgs a{};
    a.dy = y.g; a.x = x.v; a.dx = x.g; a.gamma = p(gb); a.beta = p(gb) + x.C; a.film = film ? film->v : nullptr; a.stats = stats; a.spart = spart;
    a.dfilm = dfilm ? dfilm->g : nullptr; a.C = x.C; a.rp = mk[x.lvl].rp; a.pad = mk[x.lvl].pad; a.lv = mk[x.lvl].lv;
    if (x.C == 128) hipLaunchKernelGGL(gn_act_bwd_kernel<128>, dim3((unsigned)S), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gn_act_bwd_kernel<256>, dim3((unsigned)S), dim3(256), 0, s, a);
    DGDM_HIP_CHECK(hipGetLastError());
    x.gw = true;
    if (dfilm) dfilm->gw = true;
    hipLaunchKernelGGL(rows_sum_kernel, rows_sum_grid(2 * x.C), dim3(256), 0, s, spart, (int64_t)S, 2 * x.C, gr(gb));       // dgamma | dbeta
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

// ConditionalResidualBlock1D.forward (diffusion_utils.py:101-120)
int DgdmUnetTrainer::res_fwd(Res &r, T *const *parts, hipStream_t s) {
    int rc;
    T *mg[1] = {&t_mg};
    if ((rc = conv_fwd(r.c1, parts, r.t_c1, s))) return rc;
    if ((rc = conv_fwd(r.cond, mg, r.t_film, s))) return rc;
    if ((rc = gn_fwd(r.t_c1, r.t_a, nullptr, r.g1, &r.t_film, r.st1, s))) return rc;
    T *a[1] = {&r.t_a};
    if ((rc = conv_fwd(r.c2, a, r.t_c2, s))) return rc;
    if (r.has_rc && (rc = conv_fwd(r.rc, parts, r.t_rc, s))) return rc;
    return gn_fwd(r.t_c2, r.t_out, r.has_rc ? &r.t_rc : parts[0], r.g2, nullptr, r.st2, s);
}

int DgdmUnetTrainer::res_bwd(Res &r, T *const *parts, int S, hipStream_t s) {
    int rc;
    // out = act2(c2) + R(x): the gradient of `out` goes to c2 through GroupNorm/Mish, and to x through R
    if ((rc = gn_bwd(r.t_c2, r.t_out, r.g2, nullptr, nullptr, r.st2, r.sp, S, s))) return rc;
    T *a[1] = {&r.t_a};
    r.t_a.gw = false;
    if ((rc = conv_bwd(r.c2, a, r.t_c2, s))) return rc;
    if ((rc = gn_bwd(r.t_c1, r.t_a, r.g1, &r.t_film, &r.t_film, r.st1, r.sp, S, s))) return rc;
    T *mg[1] = {&t_mg};
    if ((rc = conv_bwd(r.cond, mg, r.t_film, s))) return rc;
    if ((rc = conv_bwd(r.c1, parts, r.t_c1, s))) return rc;
    if (r.has_rc) {
        r.t_rc.g = r.t_out.g;          // d out / d R(x) = identity: the residual convolution's output gradient IS out's
        return conv_bwd(r.rc, parts, r.t_rc, s);
    }
    // identity residual: x.g += out.g
    T &x = *parts[0];
    const int64_t n = M[x.lvl] * x.C;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, r.t_out.g, x.g, n, 1);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int DgdmUnetTrainer::repack(hipStream_t s) {
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)((max_img_elems + 255) / 256), (unsigned)descs.size()), dim3(256), 0, s, P.as<float>(), IMG.as<float>(),
                       descs_dev.as<ImgDesc>());
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

int DgdmUnetTrainer::adam(float lr, hipStream_t s) {
    ++adam_steps;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)adam_steps), bc2 = 1.0 - std::pow((double)beta2, (double)adam_steps);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, s, P.as<float>(), G.as<float>(), M1.as<float>(), V.as<float>(),
                       (int64_t)n_params, beta1, beta2, eps, wd, (float)((double)lr / bc1), (float)std::sqrt(bc2));
    DGDM_HIP_CHECK(hipGetLastError());
    return repack(s);
}

// Diffusion.get_stats (diffusion.py:126-166) on S samples: noisy input, eps-net forward, MSE loss; backward = what loss.backward() leaves
// in the parameters' .grad (training_step + Lightning's automatic optimization)
int DgdmUnetTrainer::run(const float *x0, const float *noise, const float *sa, const float *sb, const int64_t *t, int S, int64_t S_total, bool backward,
                         float *pred_out, float *loss_host, hipStream_t s) {
    int rc = reserve(S);
    if (rc) return rc;
    if (S != M[0]) {       // a smaller batch than the workspace was sized for: same buffers, fewer rows (padding rows of the unused tail stay zero)
        const int rp1 = L + 8, rp2 = L / 2 + 4;
        M[0] = S; M[1] = (int64_t)S * rp1; M[2] = (int64_t)S * rp2;
    }
    {
        const int64_t n = (int64_t)S * (L + dsed);
        hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x0, noise, sa, sb, t, freqs.as<float>(), S, L, mk[1].rp, mk[1].pad, dsed,
                           t_x0.v, t_emb.v);
        DGDM_HIP_CHECK(hipGetLastError());
    }
    auto mish = [&](const T &x, T &y) {
        const int64_t n = M[0] * x.C;
        hipLaunchKernelGGL(mish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x.v, y.v, n);
        return hipGetLastError() == hipSuccess ? DGDM_OK : DGDM_EHIP;
    };
    auto one = [](T &x) { return std::array<T *, 2>{&x, nullptr}; };
    // ---- forward (diffusion_utils.py:238-285)
    { auto pa = one(t_emb); if ((rc = conv_fwd(se1, pa.data(), t_h1, s))) return rc; }
    if ((rc = mish(t_h1, t_a1))) return rc;
    { auto pa = one(t_a1); if ((rc = conv_fwd(se3, pa.data(), t_gf, s))) return rc; }
    if ((rc = mish(t_gf, t_mg))) return rc;
    std::array<T *, 2> in[8] = {one(t_x0), one(res[0].t_out), one(t_ds), one(res[2].t_out), one(res[3].t_out), one(res[4].t_out),
                                {&res[5].t_out, &res[3].t_out}, one(res[6].t_out)};
    for (int i = 0; i < 8; ++i) {
        if ((rc = res_fwd(res[i], in[i].data(), s))) return rc;
        if (i == 1) { auto pa = one(res[1].t_out); if ((rc = conv_fwd(down, pa.data(), t_ds, s))) return rc; }
    }
    { auto pa = one(res[7].t_out); if ((rc = conv_fwd(up, pa.data(), t_us, s))) return rc; }
    { auto pa = one(t_us); if ((rc = conv_fwd(fconv, pa.data(), t_cf, s))) return rc; }
    if ((rc = gn_fwd(t_cf, t_af, nullptr, fg, nullptr, f_st, s))) return rc;
    { auto pa = one(t_af); if ((rc = conv_fwd(oconv, pa.data(), t_pred, s))) return rc; }
    hipLaunchKernelGGL(loss_kernel, dim3((S + 255) / 256), dim3(256), 0, s, t_pred.v, noise, S, L, mk[1].rp, mk[1].pad, (float)(2.0 / ((double)S_total * L)), t_pred.g,
                       lpart, pred_out);
    DGDM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1), 0, s, lpart, S, (double)S_total * L, loss_dev.as<float>());
    DGDM_HIP_CHECK(hipGetLastError());
    if (backward) {
        for (T *q : {&t_h1, &t_a1, &t_gf, &t_mg, &t_ds, &t_us, &t_cf, &t_af}) q->gw = false;
        for (int i = 0; i < 8; ++i) for (T *q : {&res[i].t_c1, &res[i].t_a, &res[i].t_c2, &res[i].t_out, &res[i].t_film}) q->gw = false;
        { auto pa = one(t_af); if ((rc = conv_bwd(oconv, pa.data(), t_pred, s))) return rc; }
        if ((rc = gn_bwd(t_cf, t_af, fg, nullptr, nullptr, f_st, f_sp, S, s))) return rc;
        { auto pa = one(t_us); if ((rc = conv_bwd(fconv, pa.data(), t_cf, s))) return rc; }
        { auto pa = one(res[7].t_out); if ((rc = conv_bwd(up, pa.data(), t_us, s))) return rc; }
        for (int i = 7; i >= 0; --i) {
            if ((rc = res_bwd(res[i], in[i].data(), S, s))) return rc;
            if (i == 2) { auto pa = one(res[1].t_out); if ((rc = conv_bwd(down, pa.data(), t_ds, s))) return rc; }
        }
        auto mish_bwd = [&](T &x, const T &y) {      // x.g = y.g * mish'(x.v)
            const int64_t n = M[0] * x.C;
            hipLaunchKernelGGL(mish_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y.g, x.v, x.g, n, 0);
            x.gw = true;
            return hipGetLastError() == hipSuccess ? DGDM_OK : DGDM_EHIP;
        };
        if ((rc = mish_bwd(t_gf, t_mg))) return rc;
        { auto pa = one(t_a1); if ((rc = conv_bwd(se3, pa.data(), t_gf, s))) return rc; }
        if ((rc = mish_bwd(t_h1, t_a1))) return rc;
        { auto pa = one(t_emb); if ((rc = conv_bwd(se1, pa.data(), t_h1, s))) return rc; }
    }
    if (loss_host) {
        DGDM_HIP_CHECK(hipMemcpyAsync(loss_host, loss_dev.p, sizeof(float), hipMemcpyDeviceToHost, s));
        DGDM_HIP_CHECK(hipStreamSynchronize(s));
    }
    return DGDM_OK;
}

// which: 0 parameters, 1 gradients, 2 / 3 Adam's exp_avg / exp_avg_sq, 4 the EMA copy
int DgdmUnetTrainer::copy_state(int which, DgdmTensor *t, int n, bool to_device) {
    DevBuf *src = which == 0 ? &P : which == 1 ? &G : which == 2 ? &M1 : which == 3 ? &V : &E;
    std::vector<float> host(n_params);
    DGDM_HIP_CHECK(hipDeviceSynchronize());
    DGDM_HIP_CHECK(hipMemcpy(host.data(), src->p, n_params * sizeof(float), hipMemcpyDeviceToHost));
    std::map<std::string, DgdmTensor *> by;
    for (int i = 0; i < n; ++i) by[t[i].name] = &t[i];
    for (const Named &nm : named) {
        auto it = by.find(nm.name);
        if (it == by.end()) { set_error("state_dict key '%s' missing", nm.name.c_str()); return DGDM_EKEY; }
        if (it->second->dtype != 0 || it->second->numel != nm.numel) {
            set_error("state_dict key '%s': expected %lld float32 values, got %lld", nm.name.c_str(), (long long)nm.numel, (long long)it->second->numel);
            return DGDM_EKEY;
        }
        float *user = const_cast<float *>(static_cast<const float *>(it->second->data));
        if (to_device) memcpy(&host[nm.off], user, (size_t)nm.numel * sizeof(float));
        else memcpy(user, &host[nm.off], (size_t)nm.numel * sizeof(float));
    }
    if (to_device) {
        DGDM_HIP_CHECK(hipMemcpy(src->p, host.data(), n_params * sizeof(float), hipMemcpyHostToDevice));
        if (which == 0) {
            int rc = repack(0);
            if (rc) return rc;
            DGDM_HIP_CHECK(hipDeviceSynchronize());
        }
    }
    return DGDM_OK;
}

extern "C" int dgdm_unet_trainer_create(DgdmUnetTrainer **out, const DgdmTensor *state_dict, int n_tensors, int num_points, const int32_t *down_dims,
                                        int n_down, int dsed, int kernel_size, int n_groups, float beta1, float beta2, float eps, float weight_decay) {
    DGDM_REQUIRE(out && state_dict && down_dims, DGDM_EINVAL, "dgdm_unet_trainer_create: null argument");
    DGDM_REQUIRE(n_down == 2 && kernel_size == 5 && n_groups == 8, DGDM_EINVAL,
                 "dgdm_unet_trainer_create: the U-Net of generator/train.py:80 (two levels, kernel 5, 8 groups) is what is built, got %d levels, kernel %d, %d groups",
                 n_down, kernel_size, n_groups);
    DGDM_REQUIRE(down_dims[0] == 128 && down_dims[1] == 256, DGDM_EINVAL, "dgdm_unet_trainer_create: down_dims must be [128, 256] (GroupNorm kernels), got [%d, %d]",
                 down_dims[0], down_dims[1]);
    DGDM_REQUIRE(num_points >= 2 && num_points % 2 == 0 && num_points <= 512, DGDM_EINVAL, "dgdm_unet_trainer_create: num_points %d (must be even: the up path "
                 "returns 2 * (L / 2) positions)", num_points);
    DGDM_REQUIRE(dsed >= 4 && dsed % 4 == 0, DGDM_EINVAL, "dgdm_unet_trainer_create: diffusion_step_embed_dim %d", dsed);
    std::unique_ptr<DgdmUnetTrainer> m(new DgdmUnetTrainer());
    m->L = num_points; m->d0 = down_dims[0]; m->d1 = down_dims[1]; m->dsed = dsed; m->ks = kernel_size;
    m->beta1 = beta1; m->beta2 = beta2; m->eps = eps; m->wd = weight_decay;
    const int d0 = m->d0, d1 = m->d1;
    using UT = DgdmUnetTrainer;
    m->make_conv(m->se1, "diffusion_step_encoder.1", UT::S1, 1, dsed, 4 * dsed);
    m->make_conv(m->se3, "diffusion_step_encoder.3", UT::S1, 1, 4 * dsed, dsed);
    const struct { const char *name; int cin, cout, lvl, nparts; } spec[8] = {
        {"down_modules.0.0", 1, d0, 1, 1}, {"down_modules.0.1", d0, d0, 1, 1}, {"down_modules.1.0", d0, d1, 2, 1}, {"down_modules.1.1", d1, d1, 2, 1},
        {"mid_modules.0", d1, d1, 2, 1}, {"mid_modules.1", d1, d1, 2, 1}, {"up_modules.0.0", 2 * d1, d0, 2, 2}, {"up_modules.0.1", d0, d0, 2, 1}};
    for (int i = 0; i < 8; ++i) {
        UT::Res &r = m->res[i];
        const std::string n = spec[i].name;
        r.cout = spec[i].cout; r.lvl = spec[i].lvl;
        m->make_conv(r.c1, n + ".blocks.0.block.0", UT::S1, kernel_size, spec[i].cin, spec[i].cout, spec[i].nparts);
        r.g1 = m->add_param(n + ".blocks.0.block.1.weight", spec[i].cout); m->add_param(n + ".blocks.0.block.1.bias", spec[i].cout);
        m->make_conv(r.c2, n + ".blocks.1.block.0", UT::S1, kernel_size, spec[i].cout, spec[i].cout);
        r.g2 = m->add_param(n + ".blocks.1.block.1.weight", spec[i].cout); m->add_param(n + ".blocks.1.block.1.bias", spec[i].cout);
        m->make_conv(r.cond, n + ".cond_encoder.1", UT::S1, 1, dsed, 2 * spec[i].cout);
        r.has_rc = spec[i].cin != spec[i].cout;
        if (r.has_rc) m->make_conv(r.rc, n + ".residual_conv", UT::S1, 1, spec[i].cin, spec[i].cout, spec[i].nparts);
    }
    m->make_conv(m->down, "down_modules.0.2.conv", UT::DOWN, 3, d0, d0);
    m->make_conv(m->up, "up_modules.0.2.conv", UT::UP, 4, d0, d0);
    m->make_conv(m->fconv, "final_conv.0.block.0", UT::S1, kernel_size, d0, d0);
    m->fg = m->add_param("final_conv.0.block.1.weight", d0); m->add_param("final_conv.0.block.1.bias", d0);
    m->make_conv(m->oconv, "final_conv.1", UT::S1, 1, d0, 1);
    int rc;
    for (DevBuf *b : {&m->P, &m->G, &m->M1, &m->V, &m->E}) {
        if ((rc = b->alloc(m->n_params * sizeof(float)))) return rc;
        DGDM_HIP_CHECK(hipMemset(b->p, 0, m->n_params * sizeof(float)));
    }
    if ((rc = m->IMG.alloc(m->n_img * sizeof(float)))) return rc;
    if ((rc = m->descs_dev.upload(m->descs.data(), m->descs.size() * sizeof(ImgDesc)))) return rc;
    if ((rc = m->loss_dev.alloc(64))) return rc;
    {   // SinusoidalPosEmb (diffusion_utils.py:32-34): exp(arange(half) * -(log(10000) / (half - 1))) in float32
        const int half = dsed / 2;
        std::vector<float> fr(half);
        const float e = -(float)(std::log(10000.0) / (half - 1));
        for (int i = 0; i < half; ++i) fr[i] = expf((float)i * e);
        if ((rc = m->freqs.upload(fr.data(), fr.size() * sizeof(float)))) return rc;
    }
    if ((rc = m->copy_state(0, const_cast<DgdmTensor *>(state_dict), n_tensors, true))) return rc;
    DGDM_HIP_CHECK(hipMemcpy(m->E.p, m->P.p, m->n_params * sizeof(float), hipMemcpyDeviceToDevice));     // EMAModel starts as a copy of the model
    *out = m.release();
    return DGDM_OK;
}

extern "C" void dgdm_unet_trainer_destroy(DgdmUnetTrainer *m) { delete m; }

extern "C" int dgdm_unet_trainer_forward_backward(DgdmUnetTrainer *m, const float *x0_dev, const float *noise_dev, const float *sqrt_abar_dev,
                                                  const float *sqrt_1m_abar_dev, const int64_t *timesteps_dev, int samples, int64_t total_samples, int backward,
                                                  float *pred_dev, float *loss_host, void *stream) {
    DGDM_REQUIRE(m && x0_dev && noise_dev && sqrt_abar_dev && sqrt_1m_abar_dev && timesteps_dev, DGDM_EINVAL, "dgdm_unet_trainer_forward_backward: null argument");
    DGDM_REQUIRE(samples >= 1 && samples <= total_samples && samples <= (1 << 20), DGDM_EINVAL, "dgdm_unet_trainer_forward_backward: %d of %lld samples", samples,
                 (long long)total_samples);
    return m->run(x0_dev, noise_dev, sqrt_abar_dev, sqrt_1m_abar_dev, timesteps_dev, samples, total_samples, backward != 0, pred_dev, loss_host, (hipStream_t)stream);
}

extern "C" int dgdm_unet_trainer_step(DgdmUnetTrainer *m, const float *x0_dev, const float *noise_dev, const float *sqrt_abar_dev, const float *sqrt_1m_abar_dev,
                                      const int64_t *timesteps_dev, int samples, float lr, float *pred_dev, float *loss_host, void *stream) {
    int rc = dgdm_unet_trainer_forward_backward(m, x0_dev, noise_dev, sqrt_abar_dev, sqrt_1m_abar_dev, timesteps_dev, samples, samples, 1, pred_dev, nullptr, stream);
    if (rc) return rc;
    if ((rc = m->adam(lr, (hipStream_t)stream))) return rc;
    if (loss_host) {
        DGDM_HIP_CHECK(hipMemcpyAsync(loss_host, m->loss_dev.p, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
        DGDM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    }
    return DGDM_OK;
}

extern "C" int64_t dgdm_unet_trainer_gradient_count(const DgdmUnetTrainer *m) { return m ? (int64_t)m->n_params : -1; }

extern "C" int dgdm_unet_trainer_gradients(DgdmUnetTrainer *m, float *flat_dev, int64_t numel, int to_trainer, float scale, void *stream) {
    DGDM_REQUIRE(m && flat_dev && numel == (int64_t)m->n_params, DGDM_EINVAL, "dgdm_unet_trainer_gradients: expected %lld values", m ? (long long)m->n_params : 0LL);
    DGDM_HIP_CHECK(hipMemcpyAsync(to_trainer ? m->G.p : (void *)flat_dev, to_trainer ? (const void *)flat_dev : m->G.p, (size_t)numel * sizeof(float),
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (to_trainer && scale != 1.f) {
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m->G.as<float>(), numel, scale);
        DGDM_HIP_CHECK(hipGetLastError());
    }
    return DGDM_OK;
}

extern "C" int dgdm_unet_trainer_apply(DgdmUnetTrainer *m, float lr, void *stream) {
    DGDM_REQUIRE(m, DGDM_EINVAL, "dgdm_unet_trainer_apply: null handle");
    return m->adam(lr, (hipStream_t)stream);
}

extern "C" int dgdm_unet_trainer_ema_step(DgdmUnetTrainer *m, float decay, float one_minus_decay, void *stream) {
    DGDM_REQUIRE(m, DGDM_EINVAL, "dgdm_unet_trainer_ema_step: null handle");
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)((m->n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m->E.as<float>(), m->P.as<float>(),
                       (int64_t)m->n_params, decay, one_minus_decay);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

extern "C" int dgdm_unet_trainer_export(DgdmUnetTrainer *m, int which, DgdmTensor *tensors, int n_tensors) {
    DGDM_REQUIRE(m && tensors && which >= 0 && which <= 4, DGDM_EINVAL, "dgdm_unet_trainer_export: bad argument");
    return m->copy_state(which, tensors, n_tensors, false);
}

extern "C" int dgdm_unet_trainer_import(DgdmUnetTrainer *m, int which, const DgdmTensor *tensors, int n_tensors, int64_t adam_steps) {
    DGDM_REQUIRE(m && tensors && which >= 0 && which <= 4, DGDM_EINVAL, "dgdm_unet_trainer_import: bad argument");
    if (adam_steps >= 0) m->adam_steps = adam_steps;
    return m->copy_state(which, const_cast<DgdmTensor *>(tensors), n_tensors, true);
}

extern "C" int64_t dgdm_unet_trainer_steps(const DgdmUnetTrainer *m) { return m ? m->adam_steps : -1; }
