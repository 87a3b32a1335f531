// Model handles: weight packing/upload and the plain forward entry points of the C-ABI.
#include "common.h"
#include <climits>
#include "blob.h"
#include "models.h"
#include <algorithm>
#include <cmath>
#include <cstring>

using namespace dgdm;

// ================================================================================================ U-Net
namespace {

// MFMA image of a convolution (csrc/unet.hip conv_mfma): [Cout/16][ntaps][Cin/16][64 lanes][4];
// lane (i = l & 15, q = l >> 4), component c  ->  W(co = 16 mt + i, ci = 16 g + 4 q + c, tap taps[t]): K-step c of a group takes
// channel 4 q + c from lane group q, so a lane's four B values are four CONSECUTIVE channels - one ds_read_b128 (unet.hip).
// `at(co, ci, k)` reads the torch tensor: Conv1d is [Cout][Cin][KW], ConvTranspose1d is [Cin][Cout][KW].
template <class At>
std::vector<float> conv_image(int cout, int cin, const std::vector<int> &taps, At at) {
    const int mt = cout / 16, g = cin / 16, nt = (int)taps.size();
    std::vector<float> o((size_t)cout * cin * nt);
    for (int m = 0; m < mt; ++m)
        for (int t = 0; t < nt; ++t)
            for (int gg = 0; gg < g; ++gg)
                for (int lane = 0; lane < 64; ++lane)
                    for (int c = 0; c < 4; ++c)
                        o[((((size_t)m * nt + t) * g + gg) * 64 + lane) * 4 + c] = at(16 * m + (lane & 15), 16 * gg + 4 * (lane >> 4) + c, taps[t]);
    return o;
}

// bf16 image for conv_mfma_bf16: [Cout/16][ntaps][Cin/32][64 lanes][8]; lane (i = l & 15, kg = l >> 4), slot j ->
// bf16(W(co = 16 mt + i, ci = 32 g + 4 kg + j (j < 4) | 32 g + 16 + 4 kg + j - 4 (j >= 4), tap taps[t])): a lane's eight K values are
// two runs of four consecutive channels 16 apart, so both of its ds_read_b128 have the conflict-free bank pattern of the float32 path.  Appended to `dst`; returns the element offset.
template <class At>
size_t conv_image16(std::vector<uint16_t> &dst, int cout, int cin, const std::vector<int> &taps, At at) {
    const size_t off = dst.size();
    const int mt = cout / 16, g = cin / 32, nt = (int)taps.size();
    dst.resize(off + (size_t)cout * cin * nt);
    for (int m = 0; m < mt; ++m)
        for (int t = 0; t < nt; ++t)
            for (int gg = 0; gg < g; ++gg)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j)
                        dst[off + ((((size_t)m * nt + t) * g + gg) * 64 + lane) * 8 + j] =
                            f32_to_bf16(at(16 * m + (lane & 15), 32 * gg + (j < 4 ? 4 * (lane >> 4) + j : 16 + 4 * (lane >> 4) + j - 4), taps[t]));
    return off;
}

// Two-piece f16 image for conv_mfma_f16x3: every OUTPUT CHANNEL's weights times its own exact power of two 2^e_co (max |w 2^e_co| over the
// channel's row in [2^12, 2^13): f16 has five exponent bits, and a checkpoint whose per-channel gains span more than 2^16 inside one
// convolution would otherwise push its small rows under f16's subnormal floor) as h = f16(.) and l = f16(. - h); conv_image16's lane layout,
// the two pieces of an entry side by side, channel groups outside the taps: [Cout/16][Cin/32][ntaps][h | l][64 lanes][8], followed by the
// Cout float32 factors 2^-e_co the kernel multiplies its sums by (it finds them right behind the image: 4 Cout Cin ntaps bytes in).
// Appended to `dst`; returns the element offset.  *ew: always 0 (the per-matrix exponent of the first version, kept in the parameter block).
template <class At>
size_t conv_image_f16x3(std::vector<uint16_t> &dst, int cout, int cin, const std::vector<int> &taps, At at, int *ew) {
    const size_t off = dst.size();
    const int mt = cout / 16, g = cin / 32, nt = (int)taps.size();
    std::vector<int> er(cout, 0);
    for (int co = 0; co < cout; ++co) {
        float mx = 0.f;
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < nt; ++t) mx = std::max(mx, std::fabs(at(co, ci, taps[t])));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) { std::frexp(mx, &e); er[co] = std::min(std::max(13 - e, -100), 100); }     // mx = f 2^e, f in [0.5, 1)
    }
    *ew = 0;
    auto bits = [](float x) { const _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; };
    dst.resize(off + (size_t)2 * cout * cin * nt + (size_t)2 * cout);
    for (int m = 0; m < mt; ++m)
        for (int t = 0; t < nt; ++t)
            for (int gg = 0; gg < g; ++gg)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 16 * m + (lane & 15);
                        const float w = std::ldexp(at(co, 32 * gg + (j < 4 ? 4 * (lane >> 4) + j : 16 + 4 * (lane >> 4) + j - 4), taps[t]), er[co]);
                        const float h = (float)(_Float16)w;
                        const size_t entry = (((size_t)m * g + gg) * nt + t) * 2;             // channel-group-major: the kernel's K order
                        dst[off + (entry * 64 + lane) * 8 + j] = bits(h);
                        dst[off + ((entry + 1) * 64 + lane) * 8 + j] = bits(w - h);
                    }
    for (int co = 0; co < cout; ++co) {
        const float u = std::ldexp(1.0f, -er[co]);
        memcpy(&dst[off + (size_t)2 * cout * cin * nt + 2 * co], &u, 4);
    }
    return off;
}

size_t conv1d_image_f16x3(std::vector<uint16_t> &dst, const float *w, int cout, int cin, int kw, int *ew) {
    std::vector<int> taps(kw);
    for (int k = 0; k < kw; ++k) taps[k] = k;
    return conv_image_f16x3(dst, cout, cin, taps, [=](int co, int ci, int k) { return w[((size_t)co * cin + ci) * kw + k]; }, ew);
}

size_t conv1d_image16(std::vector<uint16_t> &dst, const float *w, int cout, int cin, int kw) {
    std::vector<int> taps(kw);
    for (int k = 0; k < kw; ++k) taps[k] = k;
    return conv_image16(dst, cout, cin, taps, [=](int co, int ci, int k) { return w[((size_t)co * cin + ci) * kw + k]; });
}

std::vector<float> conv1d_image(const float *w, int cout, int cin, int kw) {
    std::vector<int> taps(kw);
    for (int k = 0; k < kw; ++k) taps[k] = k;
    return conv_image(cout, cin, taps, [=](int co, int ci, int k) { return w[((size_t)co * cin + ci) * kw + k]; });
}

struct ResOff { size_t c0w, c0b, g0w, g0b, c1w, c1b, g1w, g1b, cw, cb, rw, rb; bool has_res; int cin, cout;
                size_t c0w16 = 0, c1w16 = 0, rw16 = 0;         // element offsets of the bf16 images (MFMA convolutions only)
                size_t c0wh = 0, c1wh = 0, rwh = 0; int c0e = 0, c1e = 0, re = 0; };      // ... of the f16x3 images and their scale exponents

int pack_res(const StateDict &sd, Blob &bl, std::vector<uint16_t> &b16, std::vector<uint16_t> &bh, const std::string &p, int cin, int cout, int cond, int kw,
             ResOff *o) {
    auto vec = [&](const std::string &k, int64_t n, size_t *off) -> int {
        const float *d = sd.f32(k, n);
        if (!d) return DGDM_EKEY;
        *off = bl.add(d, (size_t)n);
        return DGDM_OK;
    };
    o->cin = cin; o->cout = cout;
    const float *w;
    if (!(w = sd.f32(p + ".blocks.0.block.0.weight", (int64_t)cout * cin * kw))) return DGDM_EKEY;
    if (cin == 1) {                                   // [tap][cout] for the VALU input convolution
        std::vector<float> tw((size_t)kw * cout);
        for (int co = 0; co < cout; ++co)
            for (int k = 0; k < kw; ++k) tw[(size_t)k * cout + co] = w[(size_t)co * kw + k];
        o->c0w = bl.add(tw);
    } else {
        o->c0w = bl.add(conv1d_image(w, cout, cin, kw));
        o->c0w16 = conv1d_image16(b16, w, cout, cin, kw);
        o->c0wh = conv1d_image_f16x3(bh, w, cout, cin, kw, &o->c0e);
    }
    if (!(w = sd.f32(p + ".blocks.1.block.0.weight", (int64_t)cout * cout * kw))) return DGDM_EKEY;
    o->c1w = bl.add(conv1d_image(w, cout, cout, kw));
    o->c1w16 = conv1d_image16(b16, w, cout, cout, kw);
    o->c1wh = conv1d_image_f16x3(bh, w, cout, cout, kw, &o->c1e);
    int rc;
    if ((rc = vec(p + ".blocks.0.block.0.bias", cout, &o->c0b))) return rc;
    if ((rc = vec(p + ".blocks.0.block.1.weight", cout, &o->g0w))) return rc;
    if ((rc = vec(p + ".blocks.0.block.1.bias", cout, &o->g0b))) return rc;
    if ((rc = vec(p + ".blocks.1.block.0.bias", cout, &o->c1b))) return rc;
    if ((rc = vec(p + ".blocks.1.block.1.weight", cout, &o->g1w))) return rc;
    if ((rc = vec(p + ".blocks.1.block.1.bias", cout, &o->g1b))) return rc;
    if (!(w = sd.f32(p + ".cond_encoder.1.weight", (int64_t)2 * cout * cond))) return DGDM_EKEY;
    o->cw = bl.add(transpose(w, 2 * cout, cond));
    if ((rc = vec(p + ".cond_encoder.1.bias", 2 * cout, &o->cb))) return rc;
    o->has_res = cin != cout;
    if (o->has_res) {
        if (!(w = sd.f32(p + ".residual_conv.weight", (int64_t)cout * cin))) return DGDM_EKEY;
        o->rw = cin == 1 ? bl.add(w, (size_t)cout) : bl.add(conv1d_image(w, cout, cin, 1));
        if (cin != 1) { o->rw16 = conv1d_image16(b16, w, cout, cin, 1); o->rwh = conv1d_image_f16x3(bh, w, cout, cin, 1, &o->re); }
        if ((rc = vec(p + ".residual_conv.bias", cout, &o->rb))) return rc;
    }
    return DGDM_OK;
}

}  // namespace

extern "C" int dgdm_unet1d_create(DgdmUnet1d **out, const DgdmTensor *tensors, int n_tensors, const int32_t *down_dims, int n_down,
                                  int dsed, int kernel_size, int n_groups) {
    DGDM_REQUIRE(out && tensors && down_dims, DGDM_EINVAL, "dgdm_unet1d_create: null argument");
    DGDM_REQUIRE(n_down == 2, DGDM_EINVAL, "only two-level U-Nets (len(down_dims) == 2, as generator/train.py:80 builds) are supported, got %d", n_down);
    DGDM_REQUIRE(kernel_size == 5, DGDM_EINVAL, "kernel_size %d unsupported (reference uses 5)", kernel_size);
    const int d0 = down_dims[0], d1 = down_dims[1];
    DGDM_REQUIRE(d0 % n_groups == 0 && d1 % n_groups == 0 && dsed % 2 == 0 && dsed >= 4, DGDM_EINVAL, "bad U-Net dims");
    DGDM_REQUIRE(d0 % 128 == 0 && d1 % 128 == 0, DGDM_EINVAL, "U-Net widths must be multiples of 128 for the MFMA tiling and its chunked accumulation (got %d, %d)", d0, d1);
    StateDict sd(tensors, n_tensors);
    std::unique_ptr<DgdmUnet1d> m(new DgdmUnet1d());
    Blob &bl = m->blob;
    ResOff ro[8];
    std::vector<uint16_t> b16;                       // bf16 images of every MFMA convolution (unet.hip conv_mfma_bf16)
    std::vector<uint16_t> bh;                        // two-piece f16 images of the same (conv_mfma_f16x3)
    const struct { const char *name; int cin, cout; } spec[8] = {
        {"down_modules.0.0", 1, d0}, {"down_modules.0.1", d0, d0}, {"down_modules.1.0", d0, d1}, {"down_modules.1.1", d1, d1},
        {"mid_modules.0", d1, d1}, {"mid_modules.1", d1, d1}, {"up_modules.0.0", 2 * d1, d0}, {"up_modules.0.1", d0, d0}};
    int rc;
    for (int i = 0; i < 8; ++i)
        if ((rc = pack_res(sd, bl, b16, bh, spec[i].name, spec[i].cin, spec[i].cout, dsed, kernel_size, &ro[i]))) return rc;
    const float *w, *b;
    // diffusion_step_encoder
    std::vector<float> fr(dsed / 2);
    {   // SinusoidalPosEmb (diffusion_utils.py:32-34): exp(arange(half) * -(log(10000)/(half-1))) in float32
        const int half = dsed / 2;
        const float e = -(float)(std::log(10000.0) / (half - 1));
        for (int i = 0; i < half; ++i) fr[i] = expf((float)i * e);
    }
    const size_t o_fr = bl.add(fr);
    if (!(w = sd.f32("diffusion_step_encoder.1.weight", (int64_t)4 * dsed * dsed)) || !(b = sd.f32("diffusion_step_encoder.1.bias", 4 * dsed))) return DGDM_EKEY;
    const size_t o_s1w = bl.add(transpose(w, 4 * dsed, dsed)), o_s1b = bl.add(b, 4 * dsed);
    if (!(w = sd.f32("diffusion_step_encoder.3.weight", (int64_t)4 * dsed * dsed)) || !(b = sd.f32("diffusion_step_encoder.3.bias", dsed))) return DGDM_EKEY;
    const size_t o_s3w = bl.add(transpose(w, dsed, 4 * dsed)), o_s3b = bl.add(b, dsed);
    if (!(w = sd.f32("down_modules.0.2.conv.weight", (int64_t)d0 * d0 * 3)) || !(b = sd.f32("down_modules.0.2.conv.bias", d0))) return DGDM_EKEY;
    const size_t o_dw = bl.add(conv1d_image(w, d0, d0, 3)), o_db = bl.add(b, d0);
    const size_t o_dw16 = conv1d_image16(b16, w, d0, d0, 3);
    int e_dw = 0, e_uwe = 0, e_uwo = 0, e_fw = 0;
    const size_t o_dwh = conv1d_image_f16x3(bh, w, d0, d0, 3, &e_dw);
    if (!(w = sd.f32("up_modules.0.2.conv.weight", (int64_t)d0 * d0 * 4)) || !(b = sd.f32("up_modules.0.2.conv.bias", d0))) return DGDM_EKEY;
    // ConvTranspose1d weight is [Cin][Cout][4]; even outputs use taps (1, 3), odd outputs taps (2, 0)
    const float *wt = w;
    auto atT = [=](int co, int ci, int k) { return wt[((size_t)ci * d0 + co) * 4 + k]; };
    const size_t o_uwe = bl.add(conv_image(d0, d0, std::vector<int>{1, 3}, atT)), o_uwo = bl.add(conv_image(d0, d0, std::vector<int>{2, 0}, atT)),
                 o_ub = bl.add(b, d0);
    const size_t o_uwe16 = conv_image16(b16, d0, d0, std::vector<int>{1, 3}, atT), o_uwo16 = conv_image16(b16, d0, d0, std::vector<int>{2, 0}, atT);
    const size_t o_uweh = conv_image_f16x3(bh, d0, d0, std::vector<int>{1, 3}, atT, &e_uwe), o_uwoh = conv_image_f16x3(bh, d0, d0, std::vector<int>{2, 0}, atT, &e_uwo);
    if (!(w = sd.f32("final_conv.0.block.0.weight", (int64_t)d0 * d0 * kernel_size)) || !(b = sd.f32("final_conv.0.block.0.bias", d0))) return DGDM_EKEY;
    const size_t o_fw = bl.add(conv1d_image(w, d0, d0, kernel_size)), o_fb = bl.add(b, d0);
    const size_t o_fw16 = conv1d_image16(b16, w, d0, d0, kernel_size);
    const size_t o_fwh = conv1d_image_f16x3(bh, w, d0, d0, kernel_size, &e_fw);
    const float *gw, *gb;
    if (!(gw = sd.f32("final_conv.0.block.1.weight", d0)) || !(gb = sd.f32("final_conv.0.block.1.bias", d0))) return DGDM_EKEY;
    const size_t o_fgw = bl.add(gw, d0), o_fgb = bl.add(gb, d0);
    if (!(w = sd.f32("final_conv.1.weight", d0)) || !(b = sd.f32("final_conv.1.bias", 1))) return DGDM_EKEY;
    const size_t o_ow = bl.add(w, d0), o_ob = bl.add(b, 1);
    if ((rc = bl.upload())) return rc;

    UnetParams &p = m->p;
    memset(&p, 0, sizeof p);
    p.d0 = d0; p.d1 = d1; p.dsed = dsed; p.groups = n_groups; p.cmax = std::max(d0, d1);
    p.freqs = bl.at(o_fr);
    p.se1_wt = bl.at(o_s1w); p.se1_b = bl.at(o_s1b); p.se3_wt = bl.at(o_s3w); p.se3_b = bl.at(o_s3b);
    for (int i = 0; i < 8; ++i) {
        UnetRes &r = p.res[i];
        r.cin = ro[i].cin; r.cout = ro[i].cout;
        r.c0_w = bl.at(ro[i].c0w); r.c0_b = bl.at(ro[i].c0b); r.g0_w = bl.at(ro[i].g0w); r.g0_b = bl.at(ro[i].g0b);
        r.c1_w = bl.at(ro[i].c1w); r.c1_b = bl.at(ro[i].c1b); r.g1_w = bl.at(ro[i].g1w); r.g1_b = bl.at(ro[i].g1b);
        r.cond_wt = bl.at(ro[i].cw); r.cond_b = bl.at(ro[i].cb);
        r.res_w = ro[i].has_res ? bl.at(ro[i].rw) : nullptr;
        r.res_b = ro[i].has_res ? bl.at(ro[i].rb) : nullptr;
    }
    p.down_w = bl.at(o_dw); p.down_b = bl.at(o_db); p.up_w_even = bl.at(o_uwe); p.up_w_odd = bl.at(o_uwo); p.up_b = bl.at(o_ub);
    p.fin_w = bl.at(o_fw); p.fin_b = bl.at(o_fb); p.fin_gw = bl.at(o_fgw); p.fin_gb = bl.at(o_fgb);
    p.out_w = bl.at(o_ow); p.out_b = bl.at(o_ob);
    if ((rc = m->p_dev.upload(&p, sizeof p))) return rc;
    {   // the same parameter block with the MFMA convolutions pointing at their bf16 images
        if ((rc = m->w16.upload(b16.data(), b16.size() * sizeof(uint16_t)))) return rc;
        auto at16 = [&](size_t off) { return reinterpret_cast<const float *>(static_cast<const uint16_t *>(m->w16.p) + off); };
        UnetParams q = p;
        q.bf16 = 1;
        for (int i = 0; i < 8; ++i) {
            if (ro[i].cin != 1) q.res[i].c0_w = at16(ro[i].c0w16);
            q.res[i].c1_w = at16(ro[i].c1w16);
            if (ro[i].has_res && ro[i].cin != 1) q.res[i].res_w = at16(ro[i].rw16);
        }
        q.down_w = at16(o_dw16); q.up_w_even = at16(o_uwe16); q.up_w_odd = at16(o_uwo16); q.fin_w = at16(o_fw16);
        if ((rc = m->p16_dev.upload(&q, sizeof q))) return rc;
    }
    {   // ... and at their two-piece f16 images, with the scale exponents
        if ((rc = m->wf16.upload(bh.data(), bh.size() * sizeof(uint16_t)))) return rc;
        auto ath = [&](size_t off) { return reinterpret_cast<const float *>(static_cast<const uint16_t *>(m->wf16.p) + off); };
        UnetParams q = p;
        q.bf16 = 2;
        for (int i = 0; i < 8; ++i) {
            if (ro[i].cin != 1) { q.res[i].c0_w = ath(ro[i].c0wh); q.res[i].c0_e = ro[i].c0e; }
            q.res[i].c1_w = ath(ro[i].c1wh); q.res[i].c1_e = ro[i].c1e;
            if (ro[i].has_res && ro[i].cin != 1) { q.res[i].res_w = ath(ro[i].rwh); q.res[i].res_e = ro[i].re; }
        }
        q.down_w = ath(o_dwh); q.up_w_even = ath(o_uweh); q.up_w_odd = ath(o_uwoh); q.fin_w = ath(o_fwh);
        q.down_e = e_dw; q.up_e_even = e_uwe; q.up_e_odd = e_uwo; q.fin_e = e_fw;
        if ((rc = m->pf16_dev.upload(&q, sizeof q))) return rc;
        m->pf16 = q;
    }
    if (const char *e = getenv("DGDM_UNET_BATCHED_MIN")) m->batched_min = atoi(e);
    *out = m.release();
    return DGDM_OK;
}

extern "C" void dgdm_unet1d_destroy(DgdmUnet1d *m) { delete m; }

extern "C" int dgdm_unet1d_forward(DgdmUnet1d *m, const float *sample_dev, const int32_t *timestep_dev, float *eps_dev, int B, int L,
                                   void *stream) {
    DGDM_REQUIRE(m && sample_dev && timestep_dev && eps_dev && B >= 0 && L > 0, DGDM_EINVAL, "dgdm_unet1d_forward: bad argument");
    // useful multiply-adds of one forward (82.0 M per sample at L = 42, 27.4 M at L = 14; the convolutions
    // scale with L, the step encoder / FiLM linears are a 0.3 M constant)
    const double macs = 0.3e6 + (82.0e6 - 0.3e6) * (double)L / 42.0;
    prof_begin((hipStream_t)stream, DGDM_STAGE_UNET);
    // the f16x3 form needs 12 KB of LDS beside one sample's activations: where that does not fit (L = 44, 46) the float32 MFMA chain runs
    const int mode = (m->mode == 2 && !unet_f16x3_fits(m->p, L)) ? 0 : m->mode;
    if (mode == 2 && m->batched_min > 0 && B >= m->batched_min && unet_batched_samples(m->pf16, L) > 0) {
        // large batches: layer by layer, several samples per workgroup (unet.hip "batched form"; the same bits as the per-sample kernel)
        int rc = DGDM_OK;
        if (m->bws_B < B || m->bws_L != L) {
            const size_t n = unet_batched_ws_floats(m->pf16, B, L);
            DGDM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
            if ((rc = m->bws.alloc(n * sizeof(float)))) return rc;
            DGDM_HIP_CHECK(hipMemsetAsync(m->bws.p, 0, n * sizeof(float), (hipStream_t)stream));      // the halo rows stay zero: no launch writes them
            m->bws_B = B; m->bws_L = L;
        }
        rc = unet_launch_batched(m->pf16, m->pf16_dev.as<UnetParams>(), m->bws.as<float>(), m->bws_B, sample_dev, timestep_dev, eps_dev, B, L, (hipStream_t)stream);
        prof_end((hipStream_t)stream, DGDM_STAGE_UNET, 2.0 * macs * B);
        return rc;
    }
    const int rc = unet_launch(m->p, (mode == 1 ? m->p16_dev : mode == 2 ? m->pf16_dev : m->p_dev).as<UnetParams>(), mode == 2, sample_dev, timestep_dev, eps_dev, B, L, (hipStream_t)stream);
    prof_end((hipStream_t)stream, DGDM_STAGE_UNET, 2.0 * macs * B);
    return rc;
}

extern "C" int dgdm_unet1d_effective_form(const DgdmUnet1d *m, int B, int L) {
    if (!m || B < 0 || L <= 0) return -1;
    const int mode = (m->mode == 2 && !unet_f16x3_fits(m->p, L)) ? 0 : m->mode;
    const bool batched = mode == 2 && m->batched_min > 0 && B >= m->batched_min && unet_batched_samples(m->pf16, L) > 0;
    return (mode == 1 ? DGDM_DTYPE_BF16 : mode == 2 ? DGDM_DTYPE_F32_F16X3 : DGDM_DTYPE_F32_MFMA) + (batched ? 16 : 0);
}

extern "C" int dgdm_unet1d_set_contraction_dtype(DgdmUnet1d *m, int dtype) {
    DGDM_REQUIRE(m, DGDM_EINVAL, "dgdm_unet1d_set_contraction_dtype: null handle");
    // DGDM_DTYPE_F32 (the default) = DGDM_DTYPE_F32_F16X3: float32-grade convolutions as three f16 MFMA products; DGDM_DTYPE_F32_MFMA: the float32 MFMA chain of rounds 1-3; DGDM_DTYPE_BF16: operands rounded to bf16
    DGDM_REQUIRE(dtype >= DGDM_DTYPE_F32 && dtype <= DGDM_DTYPE_F32_F16X3, DGDM_EINVAL, "contraction dtype %d unsupported", dtype);
    m->mode = dtype == DGDM_DTYPE_BF16 ? 1 : dtype == DGDM_DTYPE_F32_MFMA ? 0 : 2;
    return DGDM_OK;
}

// ================================================================================================ dynamics
namespace {

// columns [c0, c0+n) of a row-major [rows][cols] matrix -> row-major [rows][n]
std::vector<float> cols(const std::vector<float> &w, int rows, int ncols, int c0, int n) {
    std::vector<float> o((size_t)rows * n);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < n; ++c) o[(size_t)r * n + c] = w[(size_t)r * ncols + c0 + c];
    return o;
}

std::vector<double> cols64(const std::vector<double> &w, int rows, int ncols, int c0, int n) {
    std::vector<double> o((size_t)rows * n);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < n; ++c) o[(size_t)r * n + c] = w[(size_t)r * ncols + c0 + c];
    return o;
}

// ---- two-way f16 split streams (csrc/trunk_f16l.hip): w 2^ew = h + l, both f16, max |w 2^ew| in [2^12, 2^13)
struct Split2 {
    std::vector<float> p[2];
    int K = 0, ew = 0;
    static float f16(float x) { const _Float16 h = (_Float16)x; return (float)h; }
    static uint16_t bits(float x) { const _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }      // exact: x is an f16 value
    Split2(const float *src, int M, int K_) : K(K_) {
        float mx = 0.f;
        for (size_t i = 0; i < (size_t)M * K_; ++i) mx = std::max(mx, std::fabs(src[i]));
        int e = 0;
        if (mx > 0.f) std::frexp(mx, &e);           // mx = f 2^e, f in [0.5, 1)
        ew = 13 - e;
        for (auto &v : p) v.resize((size_t)M * K_);
        for (size_t i = 0; i < (size_t)M * K_; ++i) {
            const float w = std::ldexp(src[i], ew), h = f16(w);
            p[0][i] = h; p[1][i] = f16(w - h);
        }
    }
    // the two 1 KiB entries [h l] of (output block op, input block ib, K-step s): lane (i, hh), slot j = P[32 op + i][32 ib + rho(8 s + j, hh)]
    void emit(std::vector<uint16_t> &dst, int op, int ib, int s) const {
        for (int q = 0; q < 2; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, hh = lane >> 5;
                for (int j = 0; j < 8; ++j) {
                    const int rr = 8 * s + j, f = (rr & 3) + 8 * (rr >> 2) + 4 * hh;
                    dst.push_back(bits(p[q][(size_t)(32 * op + i) * K + 32 * ib + f]));
                }
            }
    }
};
// 256 -> 256 layer (trunk_f16l.hip stream_layer): 16 K-steps x 4 output-block pairs x [A.h A.l B.h B.l]; returns the matrix' scale exponent
int f16_layer_stream(std::vector<uint16_t> &dst, const float *w /*[256][256]*/) {
    const Split2 sp(w, 256, 256);
    for (int ks = 0; ks < 16; ++ks)
        for (int pp = 0; pp < 4; ++pp)
            for (int blk = 2 * pp; blk < 2 * pp + 2; ++blk) sp.emit(dst, blk, ks / 2, ks % 2);
    return sp.ew;
}

std::vector<float> tfreqs(int half) {
    // timestep_embedding (profile_forward_2d.py:68-71): exp(-log(10000) * arange(half, f32) / half), float32 ops
    std::vector<float> f(half);
    const float l = -(float)std::log(10000.0);
    for (int i = 0; i < half; ++i) f[i] = expf(l * (float)i / (float)half);
    return f;
}

// ---- Equilibration of the dynamics trunk (exact).  Hidden unit j of trunk layer l is carried as (its true value) x 2^e[l][j]: row j of
// the folded layer l (weights and bias) is multiplied by 2^e[l][j] and column j of layer l + 1 (of the output layer after the last one) by
// 2^-e[l][j].  ReLU commutes with a positive factor and a power of two changes no rounding, so in exact arithmetic AND in float32 (no
// overflow / underflow) the network computes the same function bit for bit - every float32 form of the trunk (MFMA chain, bf16-rounded
// operands, six-product split), the float64 tables of layer 1 and the Jacobian of the finger part see the same scaled fold and return
// what they returned before.  What it buys: the f16 pieces of trunk_f16l.hip have ONE power-of-two scale per weight matrix; a
// checkpoint whose BatchNorm-folded per-channel gains span more than ~2^16 inside a matrix would leave its small rows below f16's
// subnormal floor (an absolute error where float32 has a relative one).  e[l][j] lifts every row's largest |w| into the binade of the
// matrix' largest (after the previous layer's column factors), so rows - and, through the column factors, the next layer's inputs - are
// balanced whatever the checkpoint's gains.  The BIAS counts as a weight on a constant input of the layer's input scale A (an
// estimate: 1 for layer 1, then the largest scaled bias / (largest scaled weight x A) of the layer before): a BatchNorm channel with a
// tiny gamma / sigma but an ordinary beta - a near-dead channel, common in trained nets - has an ordinary activation relu(beta + tiny),
// and lifting its WEIGHTS to the top binade would make that activation 2^24 .. 2^60 times its neighbours', which share one f16 scale per
// tile row in trunk_f16l.hip: everything else in the row would be pushed to or below the f16 floor.  With the bias in the row's
// magnitude such a row is left where its bias is ordinary, and only its (irrelevant) tiny weights fall under the floor, as without
// equilibration.  tests/test_gpu_range.py (gains on weight and bias together, on gamma alone, near-dead channels).
struct TrunkEquil {
    std::vector<std::vector<int>> e;         // [layer 0 .. 7][unit]
    // units that are provably dead - a zero weight row and a bias <= 0, e.g. a BatchNorm channel with gamma = 0: relu gives exactly 0 for
    // every input.  Their COLUMN of the next layer multiplies that zero and is dropped from it (bit-neutral for every float32 form: the
    // products were +-0, the gradient into the unit is masked); left in place such a column - any size, it never mattered to the
    // function - could set the next matrix' f16 scale and push the live weights under f16's floor.
    std::vector<std::vector<char>> dead;
    int W1 = 0;
    static std::string lin(int l) { return "linears." + std::to_string(3 * l); }
    static std::string bn(int l) { return "linears." + std::to_string(3 * l + 1); }
    int out_of(int l) const { return l == 0 ? W1 : 256; }
    int build(const StateDict &sd, int W1_, int IN1) {
        W1 = W1_;
        e.assign(8, {});
        dead.assign(8, {});
        double A = 1.0;                                      // magnitude of the layer's inputs in scaled units (encoder outputs: O(1))
        for (int l = 0; l < 8; ++l) {
            const int out = out_of(l), in = l == 0 ? IN1 : out_of(l - 1);
            Folded64 f;
            const int rc = fold_linear64(sd, lin(l), bn(l), out, in, &f);
            if (rc) return rc;
            std::vector<int> re(out, INT_MIN);
            std::vector<double> mw(out, 0.0);                // a row's largest |w| after the previous layer's column factors
            std::vector<char> zero_row(out, 1);
            int top = INT_MIN;
            for (int j = 0; j < out; ++j) {
                for (int k = 0; k < in; ++k) {
                    const double w = std::fabs(f.w[(size_t)j * in + k]);
                    if (l && dead[l - 1][k]) continue;
                    if (std::isfinite(w)) mw[j] = std::max(mw[j], l ? std::ldexp(w, -e[l - 1][k]) : w);
                }
                zero_row[j] = mw[j] == 0.0;
                const double b = std::fabs(f.b[j]);
                const double mx = std::max(mw[j], (A > 0.0 && std::isfinite(b)) ? b / A : 0.0);
                if (mx > 0.0) { std::frexp(mx, &re[j]); top = std::max(top, re[j]); }
            }
            dead[l].assign(out, 0);
            for (int j = 0; j < out; ++j) dead[l][j] = zero_row[j] && f.b[j] <= 0.0;
            e[l].assign(out, 0);
            double An = 0.0;
            for (int j = 0; j < out; ++j) {
                if (re[j] != INT_MIN && !dead[l][j]) e[l][j] = std::min(top - re[j], 60);      // >= 0: rows are only ever scaled up, to the top row's binade
                if (!dead[l][j] && std::isfinite(f.b[j])) An = std::max(An, std::ldexp(std::max(std::fabs(f.b[j]), mw[j] * A), e[l][j]));
            }
            A = An;
        }
        return DGDM_OK;
    }
    // trunk layer l (0 .. 7; 8 = the output layer) folded and scaled
    template <class F>
    int apply(F *f, int l) const {
        for (int j = 0; j < f->out; ++j) {
            const int ej = l < 8 ? e[l][j] : 0;
            for (int k = 0; k < f->in; ++k)
                f->w[(size_t)j * f->in + k] = (l && dead[l - 1][k]) ? 0 : std::ldexp(f->w[(size_t)j * f->in + k], ej - (l ? e[l - 1][k] : 0));
            f->b[j] = std::ldexp(f->b[j], ej);
        }
        return DGDM_OK;
    }
    int fold(const StateDict &sd, int l, int out, int in, Folded *dst) const {
        const int rc = l < 8 ? fold_linear(sd, lin(l), bn(l), out, in, dst) : fold_linear(sd, "output", "", out, in, dst);
        return rc ? rc : apply(dst, l);
    }
    int fold64(const StateDict &sd, int l, int out, int in, Folded64 *dst) const {
        const int rc = l < 8 ? fold_linear64(sd, lin(l), bn(l), out, in, dst) : fold_linear64(sd, "output", "", out, in, dst);
        return rc ? rc : apply(dst, l);
    }
};

}  // namespace

extern "C" int dgdm_dynamics_create(DgdmDynamics **out, int kind, const DgdmTensor *tensors, int n_tensors, int params_ch, int object_ch) {
    DGDM_REQUIRE(out && tensors, DGDM_EINVAL, "dgdm_dynamics_create: null argument");
    if (kind != 2 && kind != 3) { set_error("model type not supported: kind %d", kind); return DGDM_EMODE; }
    DGDM_REQUIRE(params_ch > 0 && params_ch <= 256, DGDM_EINVAL, "params_ch %d unsupported", params_ch);
    StateDict sd(tensors, n_tensors);
    std::unique_ptr<DgdmDynamics> m(new DgdmDynamics());
    m->kind = kind; m->L = params_ch; m->object_ch = object_ch; m->W1 = kind == 3 ? 512 : 256;
    const int W = 256, W1 = m->W1, IN1 = 3 * W + 27;
    Blob &bl = m->blob;
    int rc;
    TrunkEquil eq;
    if ((rc = eq.build(sd, W1, IN1))) return rc;
    if (getenv("DGDM_NO_EQUILIBRATION")) for (int l = 0; l < 8; ++l) { std::fill(eq.e[l].begin(), eq.e[l].end(), 0); std::fill(eq.dead[l].begin(), eq.dead[l].end(), 0); }      // test hook: the trunk as the checkpoint scales it
    {
        std::vector<float> u(W1);
        for (int j = 0; j < W1; ++j) u[j] = std::ldexp(1.0f, eq.e[0][j]);
        if ((rc = m->z1_unit.upload(u.data(), u.size() * sizeof(float)))) return rc;
    }
    Folded g0, g2, l1, lout;
    if ((rc = fold_linear(sd, "gripper_encoder.0", "", W, params_ch, &g0))) return rc;
    if ((rc = fold_linear(sd, "gripper_encoder.2", "", W, W, &g2))) return rc;
    if ((rc = eq.fold(sd, 0, W1, IN1, &l1))) return rc;
    if ((rc = eq.fold(sd, 8, 3, W, &lout))) return rc;
    DynOff &o = m->off;
    std::vector<float> fwd, bwd_tail;                             // continuous trunk weight streams (csrc/trunk.h)
    std::vector<uint16_t> fwd16, bwd16_tail, sa3_img16;           // the same in bf16 (csrc/trunk_bf16.hip); sa3 image for z16_kernel
    const size_t E16 = 64 * 8;                                    // bf16 values per entry
    o.g0_wt = bl.add(transpose(g0.w.data(), W, params_ch)); o.g0_b = bl.add(g0.b); o.g0_w = bl.add(g0.w);
    o.g2_wt = bl.add(transpose(g2.w.data(), W, W)); o.g2_b = bl.add(g2.b); o.g2_w = bl.add(g2.w);
    // first trunk layer: concat order [x_object | x_ctrl | x_pose | time_emb]  (profile_forward_2d.py:154, _3d.py:84)
    const std::vector<float> w1o = cols(l1.w, W1, IN1, 0, W), w1c = cols(l1.w, W1, IN1, W, W), w1p = cols(l1.w, W1, IN1, 2 * W, 27),
                             w1t = cols(l1.w, W1, IN1, 2 * W + 27, W);
    o.w1c_wt = bl.add(transpose(w1c.data(), W1, W)); o.w1c_w = bl.add(w1c);
    o.w1p_wt = bl.add(transpose(w1p.data(), W1, 27));
    o.w1t_wt = bl.add(transpose(w1t.data(), W1, W));
    o.b1 = bl.add(l1.b);
    o.wout = bl.add(lout.w); o.bout = bl.add(lout.b);
    int first_mid = 1;
    if (kind == 2) {
        o.w1o_wt = bl.add(transpose(w1o.data(), W1, W));
        Folded t0, t2, e0, e2;
        if ((rc = fold_linear(sd, "time_encoder.0", "", W, W / 2, &t0))) return rc;
        if ((rc = fold_linear(sd, "time_encoder.2", "", W, W, &t2))) return rc;
        if ((rc = fold_linear(sd, "object_encoder.0", "", W, object_ch, &e0))) return rc;
        if ((rc = fold_linear(sd, "object_encoder.2", "", W, W, &e2))) return rc;
        o.te0_wt = bl.add(transpose(t0.w.data(), W, W / 2)); o.te0_b = bl.add(t0.b);
        o.te2_wt = bl.add(transpose(t2.w.data(), W, W)); o.te2_b = bl.add(t2.b);
        o.oe0_wt = bl.add(transpose(e0.w.data(), W, object_ch)); o.oe0_b = bl.add(e0.b);
        o.oe2_wt = bl.add(transpose(e2.w.data(), W, W)); o.oe2_b = bl.add(e2.b);
        o.tfreq = bl.add(tfreqs(W / 4));
        m->thalf = W / 4;
    } else {
        Folded l2;
        if ((rc = eq.fold(sd, 1, W, W1, &l2))) return rc;
        o.b2 = bl.add(l2.b);
        // forward stream head: per 32-feature block of the 512-wide layer 1, its W1o' rows (32 entries) and then the matching
        // column block of W2' (8 output blocks x 4 entries)
        const std::vector<float> i1 = pack_chain(w1o.data(), W1, W), i2 = pack_chain(l2.w.data(), W, W1);
        const size_t E = 64 * 4;                                  // floats per entry (64 lanes x float4)
        for (int blk = 0; blk < 16; ++blk) {
            fwd.insert(fwd.end(), i1.begin() + (size_t)blk * 32 * E, i1.begin() + (size_t)(blk + 1) * 32 * E);
            for (int op = 0; op < 8; ++op)
                fwd.insert(fwd.end(), i2.begin() + (size_t)((op * 16 + blk) * 4) * E, i2.begin() + (size_t)((op * 16 + blk) * 4 + 4) * E);
        }
        bwd_tail = pack_chain(transpose(l2.w.data(), W, W1).data(), W1, W);     // 16 blocks x 32 entries, consumed in order
        {   // bf16 head: z(0), z(1), l2(0), z(2), l2(1), ..., z(15), l2(14), l2(15), 16 entries each (trunk_bf16.hip front3d)
            const std::vector<uint16_t> j1 = pack_chain_bf16(w1o.data(), W1, W), j2 = pack_chain_bf16(l2.w.data(), W, W1);
            auto put_z = [&](int kb) { fwd16.insert(fwd16.end(), j1.begin() + (size_t)kb * 16 * E16, j1.begin() + (size_t)(kb + 1) * 16 * E16); };
            auto put_l2 = [&](int kb) {
                for (int op = 0; op < 8; ++op)
                    fwd16.insert(fwd16.end(), j2.begin() + (size_t)((op * 16 + kb) * 2) * E16, j2.begin() + (size_t)((op * 16 + kb) * 2 + 2) * E16);
            };
            put_z(0);
            for (int kb = 1; kb < 16; ++kb) { put_z(kb); put_l2(kb - 1); }
            put_l2(15);
            bwd16_tail = pack_chain_bf16(transpose(l2.w.data(), W, W1).data(), W1, W);   // 16 blocks x 16 entries
        }
        o.tfreq = bl.add(tfreqs(W / 2));
        m->thalf = W / 2;
        first_mid = 2;
        // PointNet++ (pointnet2.py:17-19); Conv2d 1x1 weights are [out][in][1][1]
        Folded a0, a1, b0, b1, c0;
        if ((rc = fold_linear(sd, "object_encoder.sa1.mlp_convs.0", "object_encoder.sa1.mlp_bns.0", 64, 3, &a0))) return rc;
        if ((rc = fold_linear(sd, "object_encoder.sa1.mlp_convs.1", "object_encoder.sa1.mlp_bns.1", 128, 64, &a1))) return rc;
        if ((rc = fold_linear(sd, "object_encoder.sa2.mlp_convs.0", "object_encoder.sa2.mlp_bns.0", 128, 131, &b0))) return rc;
        if ((rc = fold_linear(sd, "object_encoder.sa2.mlp_convs.1", "object_encoder.sa2.mlp_bns.1", 256, 128, &b1))) return rc;
        if ((rc = fold_linear(sd, "object_encoder.sa3.mlp_convs.0", "object_encoder.sa3.mlp_bns.0", 256, 259, &c0))) return rc;
        o.sa1_w0t = bl.add(transpose(a0.w.data(), 64, 3)); o.sa1_b0 = bl.add(a0.b);
        o.sa1_w1 = bl.add(a1.w); o.sa1_b1 = bl.add(a1.b);
        o.sa2_wf_t = bl.add(transpose(cols(b0.w, 128, 131, 3, 128).data(), 128, 128)); o.sa2_b0 = bl.add(b0.b);
        o.sa2_vx = bl.add(transpose(cols(b0.w, 128, 131, 0, 3).data(), 128, 3));
        o.sa2_w1_img = bl.add(pack_chain(b1.w.data(), 256, 128)); o.sa2_b1 = bl.add(b1.b);
        o.sa3_w_img = bl.add(pack_chain(cols(c0.w, 256, 259, 3, 256).data(), 256, 256));
        sa3_img16 = pack_chain_bf16(cols(c0.w, 256, 259, 3, 256).data(), 256, 256);
        o.sa3_wx = bl.add(transpose(cols(c0.w, 256, 259, 0, 3).data(), 256, 3)); o.sa3_b = bl.add(c0.b);
    }
    m->n_mid = 8 - first_mid;
    std::vector<std::vector<float>> bwd_imgs;
    std::vector<std::vector<uint16_t>> bwd16_imgs;
    for (int i = 0; i < m->n_mid; ++i) {
        const int li = 3 * (first_mid + i);
        Folded f;
        if ((rc = eq.fold(sd, li / 3, W, W, &f))) return rc;
        const std::vector<float> fi = pack_chain(f.w.data(), W, W);
        fwd.insert(fwd.end(), fi.begin(), fi.end());
        o.bf[i] = bl.add(f.b);
        bwd_imgs.push_back(pack_chain(transpose(f.w.data(), W, W).data(), W, W));
        const std::vector<uint16_t> f16 = pack_chain_bf16(f.w.data(), W, W);
        fwd16.insert(fwd16.end(), f16.begin(), f16.end());
        bwd16_imgs.push_back(pack_chain_bf16(transpose(f.w.data(), W, W).data(), W, W));
    }
    {   // output layer padded to one 32-row block (forward) / its transpose padded to K = 32 (backward, consumed first)
        std::vector<float> wo((size_t)32 * W, 0.f), wot((size_t)W * 32, 0.f);
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < W; ++k) { wo[(size_t)j * W + k] = lout.w[(size_t)j * W + k]; wot[(size_t)k * 32 + j] = lout.w[(size_t)j * W + k]; }
        const std::vector<uint16_t> o16 = pack_chain_bf16(wo.data(), 32, W), ot16 = pack_chain_bf16(wot.data(), W, 32);
        fwd16.insert(fwd16.end(), o16.begin(), o16.end());
        std::vector<uint16_t> bwd16(ot16);
        for (int i = m->n_mid - 1; i >= 0; --i) bwd16.insert(bwd16.end(), bwd16_imgs[i].begin(), bwd16_imgs[i].end());
        bwd16.insert(bwd16.end(), bwd16_tail.begin(), bwd16_tail.end());
        m->fwd16_bytes = fwd16.size() * 2; m->bwd16_bytes = bwd16.size() * 2;
        fwd16.insert(fwd16.end(), bwd16.begin(), bwd16.end());
        m->sa3_16_offset = fwd16.size() * 2;
        fwd16.insert(fwd16.end(), sa3_img16.begin(), sa3_img16.end());
        if ((rc = m->w16.upload(fwd16.data(), fwd16.size() * 2))) return rc;
    }
    {   // two-way f16 split streams (trunk_f16l.hip), in consumption order
        std::vector<uint16_t> fs, bs;
        TrunkF16Scales &sc = m->f16_scales;
        sc = TrunkF16Scales{};
        Folded l2;
        if (kind == 3) {
            if ((rc = eq.fold(sd, 1, W, W1, &l2))) return rc;
            const std::vector<float> w1o3 = cols(l1.w, W1, IN1, 0, W);
            const Split2 s1(w1o3.data(), W1, W);
            const Split2 s2(l2.w.data(), W, W1);
            sc.ew_l1 = s1.ew;
            sc.ew_l2 = s2.ew;                             // (the transposed image of the last layer back has the same largest entry: checked below)
            // the largest absolute row sum of layer 1's embedding columns: |W1o x| <= l1_norm1 max |x| bounds a row of layer 1 before it exists,
            // which is what fixes the row's f16 scale for layer 2 (trunk_f16l.hip, front)
            float n1 = 0.f;
            for (int j = 0; j < W1; ++j) {
                double a = 0.0;
                for (int k = 0; k < W; ++k) a += std::fabs((double)w1o3[(size_t)j * W + k]);
                n1 = std::max(n1, (float)(a * (1.0 + 1e-6)));
            }
            sc.l1_norm1 = n1;
            for (int kb = 0; kb < 16; ++kb) {
                for (int ks = 0; ks < 16; ++ks) s1.emit(fs, kb, ks / 2, ks % 2);                    // layer-1 block kb: 16 K-steps x [h l]
                for (int pp = 0; pp < 4; ++pp)                                                      // layer 2, input block kb: 2 K-steps x 4 pairs x [A.h A.l B.h B.l]
                    for (int sx = 0; sx < 2; ++sx)
                        for (int blk = 2 * pp; blk < 2 * pp + 2; ++blk) s2.emit(fs, blk, kb, sx);
            }
        }
        std::vector<std::vector<uint16_t>> back;
        for (int i = 0; i < m->n_mid; ++i) {
            const int li = 3 * (first_mid + i);
            Folded f;
            if ((rc = eq.fold(sd, li / 3, W, W, &f))) return rc;
            sc.ew_mid[i] = f16_layer_stream(fs, f.w.data());
            back.emplace_back();
            const int et = f16_layer_stream(back.back(), transpose(f.w.data(), W, W).data());
            DGDM_REQUIRE(et == sc.ew_mid[i], DGDM_EINVAL, "f16 split: a matrix and its transpose disagree about their scale");
        }
        for (int i = m->n_mid - 1; i >= 0; --i) bs.insert(bs.end(), back[i].begin(), back[i].end());               // last layer first
        if (kind == 3) {
            const std::vector<float> w2t = transpose(l2.w.data(), W, W1);                                          // [512][256]
            const Split2 st(w2t.data(), W1, W);
            DGDM_REQUIRE(st.ew == sc.ew_l2, DGDM_EINVAL, "f16 split: layer 2 and its transpose disagree about their scale");
            for (int kb = 0; kb < 16; ++kb)
                for (int ks = 0; ks < 16; ++ks) st.emit(bs, kb, ks / 2, ks % 2);
        }
        m->fwdh_bytes = fs.size() * 2; m->bwdh_bytes = bs.size() * 2;
        fs.insert(fs.end(), bs.begin(), bs.end());
        if ((rc = m->wf16.upload(fs.data(), fs.size() * 2))) return rc;
    }
    std::vector<float> bwd;
    for (int i = m->n_mid - 1; i >= 0; --i) bwd.insert(bwd.end(), bwd_imgs[i].begin(), bwd_imgs[i].end());     // last layer first
    bwd.insert(bwd.end(), bwd_tail.begin(), bwd_tail.end());
    o.wfwd = bl.add(fwd); o.fwd_floats = fwd.size();
    o.wbwd = bl.add(bwd); o.bwd_floats = bwd.size();
    if ((rc = bl.upload())) return rc;
    {   // float64 folds (unrounded) of everything outside the per-row trunk: encoders, first-layer tables, PointNet++ table build
        Blob64 &b6 = m->blob64;
        DynOff64 &q = m->off64;
        Folded64 h0, h2, k1;
        if ((rc = fold_linear64(sd, "gripper_encoder.0", "", W, params_ch, &h0))) return rc;
        if ((rc = fold_linear64(sd, "gripper_encoder.2", "", W, W, &h2))) return rc;
        if ((rc = eq.fold64(sd, 0, W1, IN1, &k1))) return rc;
        q.g0_wt = b6.add(transpose64(h0.w.data(), W, params_ch)); q.g0_b = b6.add(h0.b); q.g0_w = b6.add(h0.w);
        q.g2_wt = b6.add(transpose64(h2.w.data(), W, W)); q.g2_b = b6.add(h2.b); q.g2_w = b6.add(h2.w);
        const std::vector<double> v1o = cols64(k1.w, W1, IN1, 0, W), v1c = cols64(k1.w, W1, IN1, W, W), v1p = cols64(k1.w, W1, IN1, 2 * W, 27),
                                  v1t = cols64(k1.w, W1, IN1, 2 * W + 27, W);
        q.w1c_wt = b6.add(transpose64(v1c.data(), W1, W)); q.w1c_w = b6.add(v1c);
        q.w1p_wt = b6.add(transpose64(v1p.data(), W1, 27));
        q.w1t_wt = b6.add(transpose64(v1t.data(), W1, W));
        q.b1 = b6.add(k1.b);
        if (kind == 2) {
            q.w1o_wt = b6.add(transpose64(v1o.data(), W1, W));
            Folded64 t0, t2, e0, e2;
            if ((rc = fold_linear64(sd, "time_encoder.0", "", W, W / 2, &t0))) return rc;
            if ((rc = fold_linear64(sd, "time_encoder.2", "", W, W, &t2))) return rc;
            if ((rc = fold_linear64(sd, "object_encoder.0", "", W, object_ch, &e0))) return rc;
            if ((rc = fold_linear64(sd, "object_encoder.2", "", W, W, &e2))) return rc;
            q.te0_wt = b6.add(transpose64(t0.w.data(), W, W / 2)); q.te0_b = b6.add(t0.b);
            q.te2_wt = b6.add(transpose64(t2.w.data(), W, W)); q.te2_b = b6.add(t2.b);
            q.oe0_wt = b6.add(transpose64(e0.w.data(), W, object_ch)); q.oe0_b = b6.add(e0.b);
            q.oe2_wt = b6.add(transpose64(e2.w.data(), W, W)); q.oe2_b = b6.add(e2.b);
        } else {
            Folded64 a0, a1, b0, b1, c0;
            if ((rc = fold_linear64(sd, "object_encoder.sa1.mlp_convs.0", "object_encoder.sa1.mlp_bns.0", 64, 3, &a0))) return rc;
            if ((rc = fold_linear64(sd, "object_encoder.sa1.mlp_convs.1", "object_encoder.sa1.mlp_bns.1", 128, 64, &a1))) return rc;
            if ((rc = fold_linear64(sd, "object_encoder.sa2.mlp_convs.0", "object_encoder.sa2.mlp_bns.0", 128, 131, &b0))) return rc;
            if ((rc = fold_linear64(sd, "object_encoder.sa2.mlp_convs.1", "object_encoder.sa2.mlp_bns.1", 256, 128, &b1))) return rc;
            if ((rc = fold_linear64(sd, "object_encoder.sa3.mlp_convs.0", "object_encoder.sa3.mlp_bns.0", 256, 259, &c0))) return rc;
            q.sa1_w0t = b6.add(transpose64(a0.w.data(), 64, 3)); q.sa1_b0 = b6.add(a0.b);
            q.sa1_w1 = b6.add(a1.w); q.sa1_b1 = b6.add(a1.b);
            q.sa2_wf_t = b6.add(transpose64(cols64(b0.w, 128, 131, 3, 128).data(), 128, 128)); q.sa2_b0 = b6.add(b0.b);
            q.sa2_vx = b6.add(transpose64(cols64(b0.w, 128, 131, 0, 3).data(), 128, 3));
            q.sa2_w1_img = b6.add(pack_mfma64(b1.w.data(), 256, 128)); q.sa2_b1 = b6.add(b1.b);
            q.sa3_w_img = b6.add(pack_mfma64(cols64(c0.w, 256, 259, 3, 256).data(), 256, 256));
            q.sa3_wx = b6.add(transpose64(cols64(c0.w, 256, 259, 0, 3).data(), 256, 3)); q.sa3_b = b6.add(c0.b);
        }
        if ((rc = b6.upload())) return rc;
    }
    *out = m.release();
    return DGDM_OK;
}

extern "C" void dgdm_dynamics_destroy(DgdmDynamics *m) { delete m; }

void DgdmDynamics::fill_trunk(TrunkParams *p) const {
    memset(p, 0, sizeof *p);
    for (int i = 0; i < n_mid; ++i) p->bf[i] = blob.at(off.bf[i]);
    p->n_mid = n_mid;
    p->Wout = blob.at(off.wout); p->bout = blob.at(off.bout);
    p->Wfwd = blob.at4(off.wfwd); p->fwd_bytes = (unsigned)(off.fwd_floats * 4);
    p->Wbwd = blob.at4(off.wbwd); p->bwd_bytes = (unsigned)(off.bwd_floats * 4);
    if (kind == 3) p->b2 = blob.at(off.b2);
}

void DgdmDynamics::fill_trunk_bf16(TrunkParams *p) const {     // after fill_trunk: swaps the two weight streams only
    p->Wfwd = reinterpret_cast<const float4 *>(static_cast<const char *>(w16.p)); p->fwd_bytes = (unsigned)fwd16_bytes;
    p->Wbwd = reinterpret_cast<const float4 *>(static_cast<const char *>(w16.p) + fwd16_bytes); p->bwd_bytes = (unsigned)bwd16_bytes;
}

void DgdmDynamics::fill_trunk_f16(TrunkParams *p, TrunkF16Scales *sc) const {     // after fill_trunk: swaps the two weight streams only
    p->Wfwd = reinterpret_cast<const float4 *>(static_cast<const char *>(wf16.p)); p->fwd_bytes = (unsigned)fwdh_bytes;
    p->Wbwd = reinterpret_cast<const float4 *>(static_cast<const char *>(wf16.p) + fwdh_bytes); p->bwd_bytes = (unsigned)bwdh_bytes;
    *sc = f16_scales;
}

PnWeights DgdmDynamics::pn() const {
    PnWeights w{};
    w.r1sq = (float)(0.2 * 0.2);     // `radius ** 2` is a Python double; torch compares it with float32 distances as float32
    w.r2sq = (float)(0.4 * 0.4);
    w.sa1_w0t = blob.at(off.sa1_w0t); w.sa1_b0 = blob.at(off.sa1_b0); w.sa1_w1 = blob.at(off.sa1_w1); w.sa1_b1 = blob.at(off.sa1_b1);
    w.sa2_wf_t = blob.at(off.sa2_wf_t); w.sa2_b0 = blob.at(off.sa2_b0); w.sa2_vx = blob.at(off.sa2_vx);
    w.sa2_w1_img = blob.at4(off.sa2_w1_img); w.sa2_b1 = blob.at(off.sa2_b1);
    w.sa3_w_img = blob.at4(off.sa3_w_img);
    w.sa3_w_img16 = reinterpret_cast<const float4 *>(static_cast<const char *>(w16.p) + sa3_16_offset);
    w.sa3_wx = blob.at(off.sa3_wx); w.sa3_b = blob.at(off.sa3_b);
    return w;
}

PnWeights64 DgdmDynamics::pn64() const {
    PnWeights64 w{};
    w.sa1_w0t = blob64.at(off64.sa1_w0t); w.sa1_b0 = blob64.at(off64.sa1_b0); w.sa1_w1 = blob64.at(off64.sa1_w1); w.sa1_b1 = blob64.at(off64.sa1_b1);
    w.sa2_wf_t = blob64.at(off64.sa2_wf_t); w.sa2_b0 = blob64.at(off64.sa2_b0); w.sa2_vx = blob64.at(off64.sa2_vx);
    w.sa2_w1_img = blob64.at(off64.sa2_w1_img); w.sa2_b1 = blob64.at(off64.sa2_b1);
    w.sa3_w_img = blob64.at(off64.sa3_w_img); w.sa3_wx = blob64.at(off64.sa3_wx); w.sa3_b = blob64.at(off64.sa3_b);
    return w;
}

int DgdmDynamics::gripper_forward64(const float *x, int ldx, double *V64, double *genc64, int rows, hipStream_t s) const {
    int rc;
    if ((rc = linear64(x, nullptr, ldx, blob64.at(off64.g0_wt), blob64.at(off64.g0_b), nullptr, 1, V64, nullptr, 256, rows, L, 256, ACT_RELU, s))) return rc;
    return linear64(nullptr, V64, 256, blob64.at(off64.g2_wt), blob64.at(off64.g2_b), nullptr, 1, genc64, nullptr, 256, rows, 256, 256, ACT_NONE, s);
}

int DgdmDynamics::time_part64(float t_scalar, float *tmp /*768 floats*/, double *tmp64 /*512*/, double *out64, hipStream_t s) const {
    int rc;
    // the sinusoidal features are float32 values in the reference (profile_forward_2d.py:58-76 on a float32 tensor): same kernel as before
    if ((rc = time_embed(nullptr, t_scalar, blob.at(off.tfreq), tmp, 1, thalf, s))) return rc;
    if (kind == 2) {       // time_encoder: Linear -> SiLU -> Linear (profile_forward_2d.py:92-96,153); the 3-D model feeds the raw embedding (_3d.py:83)
        if ((rc = linear64(tmp, nullptr, 2 * thalf, blob64.at(off64.te0_wt), blob64.at(off64.te0_b), nullptr, 1, tmp64, nullptr, 256, 1, 2 * thalf, 256, ACT_SILU, s))) return rc;
        if ((rc = linear64(nullptr, tmp64, 256, blob64.at(off64.te2_wt), blob64.at(off64.te2_b), nullptr, 1, tmp64 + 256, nullptr, 256, 1, 256, 256, ACT_NONE, s))) return rc;
        return linear64(nullptr, tmp64 + 256, 256, blob64.at(off64.w1t_wt), blob64.at(off64.b1), nullptr, 1, out64, nullptr, W1, 1, 256, W1, ACT_NONE, s);
    }
    return linear64(tmp, nullptr, 256, blob64.at(off64.w1t_wt), blob64.at(off64.b1), nullptr, 1, out64, nullptr, W1, 1, 256, W1, ACT_NONE, s);
}

int DgdmDynamics::object_part_2d64(const float *obj, double *tmp64 /*n*512*/, double *out64, int n, hipStream_t s) const {
    int rc;
    double *h = tmp64, *e = tmp64 + (size_t)n * 256;
    if ((rc = linear64(obj, nullptr, object_ch, blob64.at(off64.oe0_wt), blob64.at(off64.oe0_b), nullptr, 1, h, nullptr, 256, n, object_ch, 256, ACT_RELU, s))) return rc;
    if ((rc = linear64(nullptr, h, 256, blob64.at(off64.oe2_wt), blob64.at(off64.oe2_b), nullptr, 1, e, nullptr, 256, n, 256, 256, ACT_NONE, s))) return rc;
    return linear64(nullptr, e, 256, blob64.at(off64.w1o_wt), nullptr, nullptr, 1, out64, nullptr, W1, n, 256, W1, ACT_NONE, s);
}

// gripper encoder forward on `rows` fingers: V = relu(g0 x + b), GENC = g2 V + b   (profile_forward_2d.py:103-107,148)
int DgdmDynamics::gripper_forward(const float *x, int ldx, float *V, float *genc, int rows, hipStream_t s) const {
    int rc;
    if ((rc = linear(x, ldx, blob.at(off.g0_wt), blob.at(off.g0_b), nullptr, 1, V, 256, rows, L, 256, ACT_RELU, false, s))) return rc;
    return linear(V, 256, blob.at(off.g2_wt), blob.at(off.g2_b), nullptr, 1, genc, 256, rows, 256, 256, ACT_NONE, false, s);
}

// out[rows][W1] = W1'[:, time part] * time_feature(t) + b1'.  t_dev per row, or the scalar t_scalar for one row.
int DgdmDynamics::time_part(const float *t_dev, float t_scalar, float *tmp /*rows*768*/, float *out, int rows, hipStream_t s) const {
    int rc;
    float *emb = tmp, *h = tmp + (size_t)rows * 256, *enc = tmp + (size_t)rows * 512;
    if ((rc = time_embed(t_dev, t_scalar, blob.at(off.tfreq), emb, rows, thalf, s))) return rc;
    const float *feat = emb;
    if (kind == 2) {       // time_encoder: Linear -> SiLU -> Linear (profile_forward_2d.py:92-96,153); the 3-D model feeds the raw embedding (_3d.py:83)
        if ((rc = linear(emb, 2 * thalf, blob.at(off.te0_wt), blob.at(off.te0_b), nullptr, 1, h, 256, rows, 2 * thalf, 256, ACT_SILU, false, s))) return rc;
        if ((rc = linear(h, 256, blob.at(off.te2_wt), blob.at(off.te2_b), nullptr, 1, enc, 256, rows, 256, 256, ACT_NONE, false, s))) return rc;
        feat = enc;
    }
    return linear(feat, 256, blob.at(off.w1t_wt), blob.at(off.b1), nullptr, 1, out, W1, rows, 256, W1, ACT_NONE, false, s);
}

// 2-D object encoder + first-layer object part: out[n][W1] = W1'[:, object part] * oenc(obj)
int DgdmDynamics::object_part_2d(const float *obj, float *tmp /*n*512*/, float *out, int n, bool accumulate, hipStream_t s) const {
    int rc;
    float *h = tmp, *e = tmp + (size_t)n * 256;
    if ((rc = linear(obj, object_ch, blob.at(off.oe0_wt), blob.at(off.oe0_b), nullptr, 1, h, 256, n, object_ch, 256, ACT_RELU, false, s))) return rc;
    if ((rc = linear(h, 256, blob.at(off.oe2_wt), blob.at(off.oe2_b), nullptr, 1, e, 256, n, 256, 256, ACT_NONE, false, s))) return rc;
    return linear(e, 256, blob.at(off.w1o_wt), nullptr, nullptr, 1, out, W1, n, 256, W1, ACT_NONE, accumulate, s);
}

// ProfileForward2DModel.forward on arbitrary rows
extern "C" int dgdm_dyn2d_forward(DgdmDynamics *m, const float *x_ctrl, const float *x_ori, const float *x_pos, const float *t,
                                  const float *object, float *logits, int rows, void *stream) {
    DGDM_REQUIRE(m && x_ctrl && x_ori && x_pos && t && object && logits && rows >= 0, DGDM_EINVAL, "dgdm_dyn2d_forward: bad argument");
    if (m->kind != 2) { set_error("model type not supported: dgdm_dyn2d_forward on a 3-D model"); return DGDM_EMODE; }
    if (rows == 0) return DGDM_OK;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // workspace: V, GENC [rows][256] | pose [rows][27 -> 32] | tmp [rows][768] | z1 [rows][256]
    const size_t need = (size_t)rows * (256 + 256 + 32 + 768 + 256) * sizeof(float);
    if ((rc = m->ws.alloc(need))) return rc;
    float *V = m->ws.as<float>(), *genc = V + (size_t)rows * 256, *pose = genc + (size_t)rows * 256, *tmp = pose + (size_t)rows * 32,
          *z1 = tmp + (size_t)rows * 768;
    if ((rc = m->time_part(t, 0.f, tmp, z1, rows, s))) return rc;                          // z1 = W1t' te + b1'
    if ((rc = m->object_part_2d(object, tmp, z1, rows, true, s))) return rc;               // += W1o' oenc
    if ((rc = m->gripper_forward(x_ctrl, m->L, V, genc, rows, s))) return rc;
    if ((rc = linear(genc, 256, m->blob.at(m->off.w1c_wt), nullptr, nullptr, 1, z1, 256, rows, 256, 256, ACT_NONE, true, s))) return rc;
    if ((rc = pose_embed(x_ori, x_pos, pose, rows, s))) return rc;
    if ((rc = linear(pose, 27, m->blob.at(m->off.w1p_wt), nullptr, nullptr, 1, z1, 256, rows, 27, 256, ACT_NONE, true, s))) return rc;
    TrunkParams p;
    m->fill_trunk(&p);
    p.Atab = z1; p.logits = logits; p.C = rows; p.R = rows; p.xstride = rows; p.B = 1; p.tiles_per_b = 1; p.ntiles = (rows + 31) / 32;
    return trunk_launch(2, true, true, p, s);
}
