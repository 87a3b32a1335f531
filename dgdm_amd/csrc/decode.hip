// Finger-geometry decode: the step right after the sampler.
//
// The sampler's output is one scalar per control point in [-1, 1].  Before simulation the reference turns it into geometry
// on the host, gripper by gripper:
//   2-D: y = 0.03 p - 0.015 m over x = linspace(-0.12, 0.12, L/2) per finger (dynamics/sim_test_mj.py:257-262), then
//        scipy.interpolate.CubicSpline(x, y) (not-a-knot) evaluated on linspace(x_0, x_last, num_points)
//        (assets/finger_sampler.py:7-12,39-51; prepare_finger uses num_points = 200, sim_test_mj.py:90-97);
//   3-D: y = 0.05 p - 0.05 m (dynamics/sim_test_mj_3d.py:236-237) on the 7 x 3 control net x = linspace(-0.12, 0.12, 7),
//        z = linspace(0, 0.12, 3), control point (i, j) = y[3 i + j] (assets/finger_3d.py:77-81), then a geomdl B-spline surface of
//        degree (3, 2) with clamped uniform knot vectors evaluated on sample_size x sample_size parameters in [0, 1]^2, u-major
//        (assets/finger_3d.py:13-28,60-68; save_3d_gripper uses sample_size = 25).
// Both maps are LINEAR in the control values with the abscissae fixed, so each is one constant matrix (built here on the host in
// double precision) applied to every finger of the batch by one small kernel; nothing leaves the device between the last DDIM
// step and the geometry.  geomdl is not in this image: the surface follows the published Cox-de Boor recursion and geomdl's
// documented knot-vector generator, and is checked against scipy.interpolate.BSpline (tests/test_decode.py).
#include "common.h"
#include <cmath>
#include <map>
#include <memory>
#include <mutex>

namespace dgdm {
namespace {

// ---- not-a-knot cubic spline through (x_i, y_i), as scipy.interpolate.CubicSpline builds it: solve for the knot slopes s_i,
// then on [x_i, x_i+1]: S = y_i + s_i t + c2 t^2 + c3 t^3.  Returns E [npts][n] with S(xq_p) = sum_k E[p][k] y_k.
std::vector<double> cubic_spline_matrix(const std::vector<double> &x, const std::vector<double> &xq) {
    const int n = (int)x.size(), np = (int)xq.size();
    std::vector<double> h(n - 1);
    for (int i = 0; i + 1 < n; ++i) h[i] = x[i + 1] - x[i];
    // A s = B y  (B maps y to the right-hand side through the secant slopes delta_i = (y_i+1 - y_i) / h_i)
    std::vector<double> A((size_t)n * n, 0.0), B((size_t)n * n, 0.0);
    auto add_delta = [&](int row, int i, double coef) {      // B[row] += coef * delta_i
        B[(size_t)row * n + i + 1] += coef / h[i];
        B[(size_t)row * n + i] -= coef / h[i];
    };
    for (int i = 1; i + 1 < n; ++i) {
        A[(size_t)i * n + i - 1] = h[i];
        A[(size_t)i * n + i] = 2.0 * (h[i - 1] + h[i]);
        A[(size_t)i * n + i + 1] = h[i - 1];
        add_delta(i, i - 1, 3.0 * h[i]);
        add_delta(i, i, 3.0 * h[i - 1]);
    }
    {   // not-a-knot at both ends (scipy _cubic.py: bc_type='not-a-knot', n > 3)
        const double d0 = x[2] - x[0];
        A[0] = h[1]; A[1] = d0;
        add_delta(0, 0, (h[0] + 2.0 * d0) * h[1] / d0);
        add_delta(0, 1, h[0] * h[0] / d0);
        const double d1 = x[n - 1] - x[n - 3];
        A[(size_t)(n - 1) * n + n - 1] = h[n - 3]; A[(size_t)(n - 1) * n + n - 2] = d1;
        add_delta(n - 1, n - 3, h[n - 2] * h[n - 2] / d1);
        add_delta(n - 1, n - 2, (2.0 * d1 + h[n - 2]) * h[n - 3] / d1);
    }
    // S = A^-1 B by Gauss-Jordan with partial pivoting (n = 7)
    std::vector<double> S(B);
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)piv * n + c])) piv = r;
        for (int k = 0; k < n; ++k) { std::swap(A[(size_t)c * n + k], A[(size_t)piv * n + k]); std::swap(S[(size_t)c * n + k], S[(size_t)piv * n + k]); }
        const double inv = 1.0 / A[(size_t)c * n + c];
        for (int k = 0; k < n; ++k) { A[(size_t)c * n + k] *= inv; S[(size_t)c * n + k] *= inv; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = A[(size_t)r * n + c];
            if (f == 0.0) continue;
            for (int k = 0; k < n; ++k) { A[(size_t)r * n + k] -= f * A[(size_t)c * n + k]; S[(size_t)r * n + k] -= f * S[(size_t)c * n + k]; }
        }
    }
    std::vector<double> E((size_t)np * n, 0.0);
    for (int p = 0; p < np; ++p) {
        int i = n - 2;
        for (int k = 0; k + 1 < n; ++k)
            if (xq[p] < x[k + 1]) { i = k; break; }
        const double t = xq[p] - x[i], hi = h[i];
        // S(t) = y_i + s_i t + ((delta - s_i)/h - tt) t^2 + (tt/h) t^3,  tt = (s_i + s_i+1 - 2 delta)/h
        for (int k = 0; k < n; ++k) {
            const double si = S[(size_t)i * n + k], sj = S[(size_t)(i + 1) * n + k];
            const double delta = ((k == i + 1) ? 1.0 : 0.0) / hi - ((k == i) ? 1.0 : 0.0) / hi;
            const double tt = (si + sj - 2.0 * delta) / hi;
            const double c2 = (delta - si) / hi - tt, c3 = tt / hi;
            E[(size_t)p * n + k] = ((k == i) ? 1.0 : 0.0) + si * t + c2 * t * t + c3 * t * t * t;
        }
    }
    return E;
}

// geomdl.knotvector.generate(degree, n): degree zeros, linspace(0, 1, n - degree + 1), degree ones
std::vector<double> clamped_knots(int degree, int n) {
    std::vector<double> kv(degree, 0.0);
    const int m = n - degree + 1;
    for (int i = 0; i < m; ++i) kv.push_back(m == 1 ? 0.0 : (double)i / (double)(m - 1));
    for (int i = 0; i < degree; ++i) kv.push_back(1.0);
    return kv;
}

// B-spline basis N_k,degree(u), k < n, by the Cox-de Boor recursion (the half-open convention with the last knot closed)
std::vector<double> bspline_basis(int degree, int n, const std::vector<double> &kv, double u) {
    const int m = (int)kv.size();
    std::vector<double> N(m - 1, 0.0);
    int span = -1;
    for (int i = 0; i + 1 < m; ++i)
        if (kv[i] <= u && u < kv[i + 1]) span = i;
    if (span < 0) span = n - 1;                          // u == last knot: last non-empty span
    N[span] = 1.0;
    for (int d = 1; d <= degree; ++d) {
        for (int i = 0; i + d + 1 <= m - 1; ++i) {
            const double a = (kv[i + d] > kv[i]) ? (u - kv[i]) / (kv[i + d] - kv[i]) * N[i] : 0.0;
            const double b = (kv[i + d + 1] > kv[i + 1]) ? (kv[i + d + 1] - u) / (kv[i + d + 1] - kv[i + 1]) * N[i + 1] : 0.0;
            N[i] = a + b;
        }
    }
    N.resize(n);
    return N;
}

struct DecodeTable {
    DevBuf mat;          // [npts][K] float: weights of the K control values
    DevBuf fixed;        // [npts][F] float: the coordinates that do not depend on the sample (2-D: x; 3-D: x, z)
    int npts = 0, K = 0;
};

std::mutex g_mu;
std::map<std::pair<int, int>, std::unique_ptr<DecodeTable>> g_tables;     // (kind * 65536 + K, npts) -> table

int get_table(int kind, int K, int n, DecodeTable **out) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(kind * 65536 + K, n);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) { *out = it->second.get(); return DGDM_OK; }
    std::unique_ptr<DecodeTable> t(new DecodeTable());
    std::vector<float> mat, fixed;
    if (kind == 2) {
        std::vector<double> x(K), xq(n);
        for (int i = 0; i < K; ++i) x[i] = -0.12 + 0.24 * (double)i / (double)(K - 1);       // np.linspace(-0.12, 0.12, K)
        for (int p = 0; p < n; ++p) xq[p] = (n == 1) ? x[0] : x[0] + (x[K - 1] - x[0]) * (double)p / (double)(n - 1);
        const std::vector<double> E = cubic_spline_matrix(x, xq);
        mat.assign(E.begin(), E.end());
        fixed.assign(xq.begin(), xq.end());
        t->npts = n; t->K = K;
    } else {
        const int nu = 7, nv = 3, du = 3, dv = 2;
        const std::vector<double> ku = clamped_knots(du, nu), kv = clamped_knots(dv, nv);
        mat.resize((size_t)n * n * nu * nv);
        fixed.resize((size_t)n * n * 2);
        for (int a = 0; a < n; ++a) {
            const double u = (n == 1) ? 0.0 : (double)a / (double)(n - 1);
            const std::vector<double> Nu = bspline_basis(du, nu, ku, u);
            for (int b = 0; b < n; ++b) {
                const double v = (n == 1) ? 0.0 : (double)b / (double)(n - 1);
                const std::vector<double> Nv = bspline_basis(dv, nv, kv, v);
                const size_t p = (size_t)a * n + b;                                       // u-major, as geomdl's evalpts
                double xs = 0.0, zs = 0.0;
                for (int i = 0; i < nu; ++i)
                    for (int j = 0; j < nv; ++j) {
                        const double w = Nu[i] * Nv[j];
                        mat[p * nu * nv + i * nv + j] = (float)w;
                        xs += w * (-0.12 + 0.24 * (double)i / 6.0);
                        zs += w * (0.12 * (double)j / 2.0);
                    }
                fixed[2 * p] = (float)xs; fixed[2 * p + 1] = (float)zs;
            }
        }
        t->npts = n * n; t->K = nu * nv;
    }
    int rc;
    if ((rc = t->mat.upload(mat.data(), mat.size() * sizeof(float)))) return rc;
    if ((rc = t->fixed.upload(fixed.data(), fixed.size() * sizeof(float)))) return rc;
    *out = t.get();
    g_tables[key] = std::move(t);
    return DGDM_OK;
}

// out[b][finger][p][:]: 2-D (x_p, y), 3-D (x_p, y, z_p) with y = sum_k mat[p][k] (scale * s[b][finger*K + k] + offset)
template <int DIM>
__global__ void decode_kernel(const float *__restrict__ samples, int B, int K, int npts, const float *__restrict__ mat,
                              const float *__restrict__ fixed, float scale, float offset, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)B * 2 * npts) return;
    const int p = (int)(e % npts);
    const int64_t bf = e / npts;                       // b * 2 + finger
    const float *s = samples + bf * K;
    const float *m = mat + (size_t)p * K;
    float y = 0.f;
    for (int k = 0; k < K; ++k) y = fmaf(m[k], fmaf(scale, s[k], offset), y);
    float *o = out + e * DIM;
    if (DIM == 2) {
        o[0] = fixed[p]; o[1] = y;
    } else {
        o[0] = fixed[2 * p]; o[1] = y; o[2] = fixed[2 * p + 1];
    }
}

}  // namespace
}  // namespace dgdm

extern "C" int dgdm_finger_decode_2d(const float *samples_dev, int batch, int num_ctrl, int num_points, float scale, float offset,
                                     float *curve_dev, void *stream) {
    using namespace dgdm;
    DGDM_REQUIRE(samples_dev && curve_dev && batch >= 0, DGDM_EINVAL, "dgdm_finger_decode_2d: null argument");
    DGDM_REQUIRE(num_ctrl >= 8 && num_ctrl % 2 == 0 && num_points >= 1, DGDM_EINVAL,
                 "dgdm_finger_decode_2d: %d control values (need an even number >= 8: not-a-knot needs 4 knots per finger), %d points", num_ctrl, num_points);
    if (batch == 0) return DGDM_OK;
    DecodeTable *t = nullptr;
    int rc;
    if ((rc = get_table(2, num_ctrl / 2, num_points, &t))) return rc;
    const int64_t n = (int64_t)batch * 2 * t->npts;
    hipLaunchKernelGGL(decode_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, samples_dev, batch, t->K, t->npts,
                       t->mat.as<float>(), t->fixed.as<float>(), scale, offset, curve_dev);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}

extern "C" int dgdm_finger_decode_3d(const float *samples_dev, int batch, int num_ctrl, int sample_size, float scale, float offset,
                                     float *surface_dev, void *stream) {
    using namespace dgdm;
    DGDM_REQUIRE(samples_dev && surface_dev && batch >= 0, DGDM_EINVAL, "dgdm_finger_decode_3d: null argument");
    DGDM_REQUIRE(num_ctrl == 42 && sample_size >= 1, DGDM_EINVAL,
                 "dgdm_finger_decode_3d: %d control values (the reference's net is 2 fingers x 7 x 3 = 42), sample_size %d", num_ctrl, sample_size);
    if (batch == 0) return DGDM_OK;
    DecodeTable *t = nullptr;
    int rc;
    if ((rc = get_table(3, 21, sample_size, &t))) return rc;
    const int64_t n = (int64_t)batch * 2 * t->npts;
    hipLaunchKernelGGL(decode_kernel<3>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, samples_dev, batch, t->K, t->npts,
                       t->mat.as<float>(), t->fixed.as<float>(), scale, offset, surface_dev);
    DGDM_HIP_CHECK(hipGetLastError());
    return DGDM_OK;
}
