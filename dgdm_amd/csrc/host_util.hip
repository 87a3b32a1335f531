// Host-side utilities of libdgdm_hip.so: error reporting, state_dict access, BatchNorm folding,
// weight-image packing, profiling hooks, objective table, convergence row coefficients.
#include "common.h"
#include "blob.h"
#include <cmath>
#include <cstdarg>
#include <cstring>

namespace dgdm {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

const float *StateDict::f32(const std::string &k, int64_t numel) const {
    auto it = m.find(k);
    if (it == m.end()) { set_error("state_dict key '%s' missing", k.c_str()); return nullptr; }
    if (it->second->dtype != 0 || it->second->numel != numel) {
        set_error("state_dict key '%s': expected %lld float32 values, got %lld (dtype %d)", k.c_str(), (long long)numel,
                  (long long)it->second->numel, it->second->dtype);
        return nullptr;
    }
    return static_cast<const float *>(it->second->data);
}

int fold_linear(const StateDict &sd, const std::string &lin, const std::string &bn, int out, int in, Folded *dst) {
    const float *w = sd.f32(lin + ".weight", (int64_t)out * in);
    const float *b = sd.f32(lin + ".bias", out);
    if (!w || !b) return DGDM_EKEY;
    dst->out = out; dst->in = in;
    dst->w.resize((size_t)out * in);
    dst->b.resize(out);
    if (bn.empty()) {
        memcpy(dst->w.data(), w, sizeof(float) * out * in);
        memcpy(dst->b.data(), b, sizeof(float) * out);
        return DGDM_OK;
    }
    const float *g = sd.f32(bn + ".weight", out), *be = sd.f32(bn + ".bias", out);
    const float *mu = sd.f32(bn + ".running_mean", out), *var = sd.f32(bn + ".running_var", out);
    if (!g || !be || !mu || !var) return DGDM_EKEY;
    for (int o = 0; o < out; ++o) {
        const double s = (double)g[o] / std::sqrt((double)var[o] + 1e-5);      // eps of nn.BatchNorm{1,2}d
        for (int i = 0; i < in; ++i) dst->w[(size_t)o * in + i] = (float)(s * (double)w[(size_t)o * in + i]);
        dst->b[o] = (float)(s * ((double)b[o] - (double)mu[o]) + (double)be[o]);
    }
    return DGDM_OK;
}

int fold_linear64(const StateDict &sd, const std::string &lin, const std::string &bn, int out, int in, Folded64 *dst) {
    const float *w = sd.f32(lin + ".weight", (int64_t)out * in);
    const float *b = sd.f32(lin + ".bias", out);
    if (!w || !b) return DGDM_EKEY;
    dst->out = out; dst->in = in;
    dst->w.resize((size_t)out * in);
    dst->b.resize(out);
    const float *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
    if (!bn.empty()) {
        g = sd.f32(bn + ".weight", out); be = sd.f32(bn + ".bias", out);
        mu = sd.f32(bn + ".running_mean", out); var = sd.f32(bn + ".running_var", out);
        if (!g || !be || !mu || !var) return DGDM_EKEY;
    }
    for (int o = 0; o < out; ++o) {
        const double s = g ? (double)g[o] / std::sqrt((double)var[o] + 1e-5) : 1.0;      // eps of nn.BatchNorm{1,2}d
        for (int i = 0; i < in; ++i) dst->w[(size_t)o * in + i] = s * (double)w[(size_t)o * in + i];
        dst->b[o] = g ? s * ((double)b[o] - (double)mu[o]) + (double)be[o] : (double)b[o];
    }
    return DGDM_OK;
}

std::vector<double> transpose64(const double *src, int rows, int cols) {
    std::vector<double> t((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = src[(size_t)r * cols + c];
    return t;
}

std::vector<double> pack_mfma64(const double *src, int M, int K) {
    const int MP = M / 32, KS = K / 4;
    std::vector<double> img((size_t)M * K);
    for (int ks = 0; ks < KS; ++ks)
        for (int mp = 0; mp < MP; ++mp)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 2; ++j) {
                    const int m = lane & 15, kq = lane >> 4, f = 16 * (2 * mp + j) + 4 * (m & 3) + (m >> 2);
                    img[(((size_t)ks * MP + mp) * 64 + lane) * 2 + j] = src[(size_t)f * K + kq * KS + ks];
                }
    return img;
}

std::vector<float> pack_chain(const float *src, int M, int K) {
    const int MB = M / 32, KB = K / 32;
    std::vector<float> img((size_t)M * K);
    for (int op = 0; op < MB; ++op)
        for (int o = 0; o < KB; ++o)
            for (int q = 0; q < 4; ++q)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    const float *s = src + (size_t)(32 * op + i) * K + 32 * o + 8 * q + 4 * h;
                    float *d = img.data() + ((((size_t)op * KB + o) * 4 + q) * 64 + lane) * 4;
                    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
                }
    return img;
}

uint16_t f32_to_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);      // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// Image of W [M x K] (row-major, M and K multiples of 32) for v_mfma_f32_32x32x16_bf16 with the weights as the A operand:
// entry ((op * KB + ib) * 2 + s), lane (i = l & 31, h = l >> 5), slot j  =  bf16(W[32 op + i][32 ib + rho(8 s + j, h)]),
// rho(r, h) = (r & 3) + 8 (r >> 2) + 4 h.  One entry = 64 lanes x 8 bf16 = 1 KiB.
std::vector<uint16_t> pack_chain_bf16(const float *src, int M, int K) {
    const int MB = M / 32, KB = K / 32;
    std::vector<uint16_t> img((size_t)M * K);
    for (int op = 0; op < MB; ++op)
        for (int ib = 0; ib < KB; ++ib)
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    uint16_t *d = img.data() + ((((size_t)op * KB + ib) * 2 + s) * 64 + lane) * 8;
                    for (int j = 0; j < 8; ++j) {
                        const int r = 8 * s + j, f = (r & 3) + 8 * (r >> 2) + 4 * h;
                        d[j] = f32_to_bf16(src[(size_t)(32 * op + i) * K + 32 * ib + f]);
                    }
                }
    return img;
}

std::vector<float> transpose(const float *src, int rows, int cols) {
    std::vector<float> t((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = src[(size_t)r * cols + c];
    return t;
}

// ---------------------------------------------------------------- profiling: HIP events around the launches of each stage
struct ProfRec { hipEvent_t a, b; double work; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof[PROF_BUCKETS];

void prof_begin(hipStream_t s, int bucket) {
    if (!g_prof_on) return;
    ProfRec r{};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_prof[bucket].push_back(r);
}

void prof_end(hipStream_t s, int bucket, double work) {
    if (!g_prof_on || g_prof[bucket].empty()) return;
    g_prof[bucket].back().work = work;
    (void)hipEventRecord(g_prof[bucket].back().b, s);
}

}  // namespace dgdm

using namespace dgdm;

extern "C" int dgdm_version(void) { return 100; }
extern "C" const char *dgdm_last_error(void) { return g_err.c_str(); }

extern "C" int dgdm_device_init(int ordinal) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible"); return DGDM_ENODEVICE; }
    DGDM_REQUIRE(ordinal >= 0 && ordinal < n, DGDM_EINVAL, "device ordinal %d out of range (%d devices)", ordinal, n);
    hipDeviceProp_t prop;
    DGDM_HIP_CHECK(hipGetDeviceProperties(&prop, ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", ordinal, prop.gcnArchName);
        return DGDM_ENODEVICE;
    }
    DGDM_HIP_CHECK(hipSetDevice(ordinal));
    return DGDM_OK;
}

extern "C" int dgdm_prof_enable(int on) {
    g_prof_on = on != 0;
    for (auto &b : g_prof) {
        for (auto &r : b) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        b.clear();
    }
    return DGDM_OK;
}

extern "C" int dgdm_prof_read_stage(int stage, int64_t *launches, double *total_ms, double *total_work) {
    DGDM_REQUIRE(stage >= 0 && stage < PROF_BUCKETS, DGDM_EINVAL, "dgdm_prof_read_stage: stage %d outside 0..%d", stage, PROF_BUCKETS - 1);
    double ms = 0, fl = 0;
    int64_t n = 0;
    for (auto &r : g_prof[stage]) {
        DGDM_HIP_CHECK(hipEventSynchronize(r.b));
        float t = 0;
        DGDM_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        ms += t; fl += r.work; ++n;
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof[stage].clear();
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = fl;
    return DGDM_OK;
}

extern "C" int dgdm_prof_read(int64_t *launches, double *total_ms, double *total_flops) {
    return dgdm_prof_read_stage(DGDM_STAGE_TRUNK, launches, total_ms, total_flops);
}

// generator/diffusion.py:430-471 in gradient form
extern "C" int dgdm_objective_from_name(const char *name, DgdmObjective *out) {
    DGDM_REQUIRE(name && out, DGDM_EINVAL, "dgdm_objective_from_name: null argument");
    struct Row { const char *n; float l0, l1, l2; };
    static const Row rows[] = {
        {"rotate_clockwise", -1, 0, 0}, {"rotate_counterclockwise", 1, 0, 0}, {"shift_up", 0, -1, 0}, {"shift_down", 0, 1, 0},
        {"shift_left", 0, 0, -1}, {"shift_right", 0, 0, 1}, {"clockwise_up", -1, -1, 0}, {"clockwise_down", -1, 1, 0},
        {"clockwise_left", -1, 0, -1}, {"clockwise_right", -1, 0, 1}, {"counterclockwise_up", 1, -1, 0},
        {"counterclockwise_down", 1, 1, 0}, {"counterclockwise_left", 1, 0, -1}, {"counterclockwise_right", 1, 0, 1},
    };
    const int obj = out->object;
    memset(out, 0, sizeof *out);
    out->object = obj;
    if (!strcmp(name, "rotate")) { out->quad[0] = 1.f; return DGDM_OK; }          // objective = dtheta^2
    if (!strcmp(name, "convergence")) { out->use_rowcoef = 1; return DGDM_OK; }
    for (const Row &r : rows)
        if (!strcmp(name, r.n)) { out->lin[0] = r.l0; out->lin[1] = r.l1; out->lin[2] = r.l2; return DGDM_OK; }
    set_error("opt obj not supported: %s", name);
    return DGDM_EOBJECTIVE;
}

// Python slice a[start:stop] on a sequence of length n (None encoded by has_* = false): adds `sign` to coef[base + i]
static void add_slice(float *coef, int64_t base, int64_t n, bool has_start, int64_t start, bool has_stop, int64_t stop, float sign) {
    int64_t lo = 0, hi = n;
    if (has_start) { lo = start < 0 ? start + n : start; lo = lo < 0 ? 0 : (lo > n ? n : lo); }
    if (has_stop) { hi = stop < 0 ? stop + n : stop; hi = hi < 0 ? 0 : (hi > n ? n : hi); }
    for (int64_t i = lo; i < hi; ++i) coef[base + i] += sign;
}

// dynamics/metrics.py:32-38 applied to an index vector
static void add_slicer(float *coef, int64_t base, int64_t n, int64_t lower, int64_t upper, float sign) {
    if (lower < 0) {
        add_slice(coef, base, n, true, lower, false, 0, sign);
        add_slice(coef, base, n, false, 0, true, upper, sign);
    } else if (upper > n) {
        add_slice(coef, base, n, true, lower, false, 0, sign);
        add_slice(coef, base, n, false, 0, true, upper - n, sign);
    } else {
        add_slice(coef, base, n, true, lower, true, upper, sign);
    }
}

extern "C" int dgdm_convergence_rowcoef(const int64_t *centers, int n_centers, int grid_size, int num_pos, int64_t total_rows,
                                        int64_t sub_batch_size, float *rowcoef) {
    DGDM_REQUIRE(centers && rowcoef && n_centers >= 0 && total_rows >= 0, DGDM_EINVAL, "dgdm_convergence_rowcoef: bad argument");
    const int64_t pp = (int64_t)num_pos * num_pos, cells = (int64_t)grid_size * pp, half = (int64_t)(grid_size / 2) * pp;
    for (int64_t i = 0; i < total_rows; ++i) rowcoef[i] = 0.f;
    const int64_t sb = sub_batch_size > 0 ? sub_batch_size : (total_rows > 0 ? total_rows : 1);
    for (int64_t s0 = 0; s0 < total_rows; s0 += sb) {                 // cond_fn :495 (3-D) or the single call (2-D)
        const int64_t n_call = std::min(sb, total_rows - s0);
        for (int i = 0; i < n_centers; ++i) {                         // deltas_to_objective :447-451
            int64_t a0 = (int64_t)i * cells, a1 = a0 + cells;         // deltas[i*cells:(i+1)*cells] clamps like a slice
            a0 = std::min(a0, n_call); a1 = std::min(a1, n_call);
            const int64_t n = a1 - a0, c = centers[i] * pp;
            add_slicer(rowcoef, s0 + a0, n, c - half, c, +1.f);       // left_delta
            add_slicer(rowcoef, s0 + a0, n, c, c + half, -1.f);       // right_delta = slicer(-delta_theta, ...)
        }
    }
    return DGDM_OK;
}
