// Diffusion.cond_fn / get_convergence_centers (generator/diffusion.py:473-539) for batches of chains,
// and the PointNet++-backed forward entry points.
#ifndef DGDM_DEFAULT_F16X3
#endif
#include "common.h"
#include "models.h"
#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <atomic>
#include <chrono>
#include <thread>

using namespace dgdm;

namespace {

// torch.linspace(start, end, steps) for float32 on CPU: symmetric evaluation around the midpoint
std::vector<float> linspace_f32(float start, float end, int steps) {
    std::vector<float> v(steps);
    if (steps == 1) { v[0] = start; return v; }
    const float step = (end - start) / (float)(steps - 1);
    const int half = steps / 2;
    for (int i = 0; i < steps; ++i) v[i] = i < half ? start + step * (float)i : end - step * (float)(steps - i - 1);
    return v;
}

// experiment hook (DGDM_HOST_TIMING): host wall-clock stamps of the calls' phases on stderr
static inline void host_stamp(const char *tag) {
    static const bool on = getenv("DGDM_HOST_TIMING") != nullptr;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "host %9.3f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), tag);
}

struct ObjectTables {       // 3-D, per object
    // slices of the guidance handle's pools (the FPS tables of all objects are built by one launch each)
    const float *xyz = nullptr;   // [N][3]
    const int   *fps1 = nullptr;  // [N][512]
    const int   *fps2 = nullptr;  // [N][128] FPS(128) sequence by start point
    const int   *flags = nullptr; // [N] that sequence is order-dependent (exact distance tie / coordinates exhausted)
    DevBuf Z;               // [N][N][256]
    bool   fast_ok = false; // no flag set: rows can take their centres from fps2 instead of running FPS
    int *crowded = nullptr, *clist = nullptr;   // [N] int, [N + 1] int (clist[N] = count): centres whose ball query truncates (pointnet.hip crowd_kernel); pool slices
    DevBuf M0, cl2, cnt2;   // [N][256] float, [N][128] int, [N] int: variant-independent part of the sa3 max (pointnet.hip m0_kernel)
    DevBuf cl2s;            // [N][128] cl2 as positions in clist (xtab_kernel)
    DevBuf cl2o;            // [N][2][128] u16: cl2 as byte offsets into xobj_rows_kernel's LDS slab (scaled for the build's mode: has16), even / odd slots first
    DevBuf pcf;             // [N][512] per (s1, s2): sa2's start point, its list length, its tie flag (pointnet.hip pcf_kernel)
    DevBuf X, X16;          // [N][N][256] float32 / [N][N][128] bf16 dwords: the finished embedding per (s1, start point) (pointnet.hip xtab_kernel)
    bool   has_x = false, has_x16 = false;
    int    ncr = 0;         // number of crowded centres (read back by set_objects)
    DevBuf Z16, M0_16;      // bf16 operand-order copies, built when the handle is in bf16 mode at set_objects time
    bool   has16 = false;
};

}  // namespace

struct DgdmGuidance {
    DgdmDynamics *m = nullptr;
    DgdmGuidanceConfig cfg{};
    int C = 0, G = 0, B = 0, tiles_per_b = 0, sweep_tiles_per_b = 0;
    int64_t R = 0, Rs = 0;                       // rows per chain: cond_fn grid, orientation sweep
    DevBuf ptab, ptab_sweep;                     // [C][W1], [G][W1]
    DevBuf ptab_t, ptab_sweep_t;                 // the same tables tiled for the trunk kernels (smallnet.h tile_table)
    DevBuf pmax;                                 // [C] largest magnitude of a cell's row of ptab (trunk_f16l.hip: f16 scale of 3-D layer 2's input)
    DevBuf objpart;                              // 2-D: [max_objects][W1] doubles
    DevBuf objtmp;                               // 2-D: scratch of set_objects (the object encoder's hidden layer, [n_objects][512] doubles)
    std::vector<std::unique_ptr<ObjectTables>> tables;   // 3-D
    bool bf16 = false;                           // contractions of the trunk on bf16 MFMA (dgdm_guidance_set_contraction_dtype)
    bool f32_mfma = false;                       // float32 mode on the k-ordered float32 MFMA chain (trunk.hip) instead of the f16x3 form (trunk_f16l.hip)
#ifndef DGDM_NBUILD
#define DGDM_NBUILD 3
#endif
    static constexpr int NBUILD = 16;            // most objects whose tables can be built concurrently (own stream + 0.5 GB of temporaries each)
    // how many are: DGDM_NBUILD at compile time, the environment variable of that name at run time (experiments: bench.py --extra records them)
    int nbuild = []() { const char *e = getenv("DGDM_NBUILD"); const int v = e ? atoi(e) : DGDM_NBUILD; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
    DevBuf pool_crowded, pool_clist, pool_off, pool_pairs, pool_rank, pool_F1, pool_U;   // [n_objects] x the light build stages' outputs (built by one launch per stage)
    DevBuf pool_xyz, pool_fps1, pool_fps2, pool_flags, pool_ncr;   // [n_objects] x per-object FPS tables (ObjectTables point into these), crowded-centre counts
    DevBuf tmpY[NBUILD], tmpL2[NBUILD], vlist;      // 3-D table-build temporaries of the heavy stages (per build stream)
    hipStream_t bstream[NBUILD] = {}, fstream = nullptr;    // fstream: sa2's FPS table, beside the builds
    hipEvent_t bev[NBUILD] = {}, bstart = nullptr, fstart = nullptr, fdone = nullptr, ldone = nullptr;      // ldone: the batched light build stages
    // V, genc, chainbias, timepart: float64 (smallnet.h linear64: per-finger / per-chain quantities are evaluated in double precision,
    // so the A table carries one float32 rounding); ttmp64: scratch of the time encoder
    DevBuf V, genc, atab, chainbias, timepart, ttmp, ttmp64, partial, objdev, objidx, xobj, xobj16, starts, order, xchains, todo, groupoff;
    DevBuf loopx[2], loopeps, loopgrad, loopxrep, loopts;      // workspace of dgdm_guided_chains_run
    DevBuf xidx, xidxchains, xtabptrs;       // embedding-table path: row index per reference row, per-chain lookup info, per-chain table base pointers
    bool xtab_enabled = true;       // test hook: modes 1-3 read materialised rows (per-step gather kernels) instead of the embedding table
    int xtab_policy = 0;            // 0: build the embedding tables once the objects have served more than XTAB_AFTER cond_fn calls; 1: at set_objects (test hook mode 5)
    int grads_since_set = 0;
    static constexpr int XTAB_AFTER = 5;
    int l2_gather_mode = 0;         // test hook (mode 4): build the sa2 features of the crowded centres with l2_kernel's global gathers
    int xobj_mode = 0;              // test hook: 0 = group kernel where possible, 2 = per-row table kernel (xobj_fast_kernel)
    int n_objects = 0;
    bool force_slow_xobj = false;   // test hook: always run the per-row FPS kernel
    void *pinned = nullptr; size_t pinned_bytes = 0; hipEvent_t pinned_ev = nullptr;
    int64_t todo_capacity = 0;
    // uploads of the start indices go through their own stream (the copy engine works beside the kernels of the launch stream):
    // up_ready orders the consumers behind the copy, up_consumed the next copy behind the last reader of the device buffers
    // set_objects' read-back (tie flags, crowded-centre counts) lands in pinned memory behind ro_ev and is only waited for when its
    // values are first needed - the host converts and uploads the first step's start indices while the tables are still being built
    int *ro_host = nullptr; size_t ro_ints = 0; hipEvent_t ro_ev = nullptr; bool ro_pending = false;
    int finish_objects();
    hipStream_t cstream = nullptr; hipEvent_t up_ready = nullptr, up_consumed = nullptr; bool consumed_recorded = false;
    ~DgdmGuidance() {
        if (pinned) (void)hipHostFree(pinned);
        if (pinned_ev) (void)hipEventDestroy(pinned_ev);
        if (ro_host) (void)hipHostFree(ro_host);
        if (ro_ev) (void)hipEventDestroy(ro_ev);
        if (cstream) (void)hipStreamDestroy(cstream);
        if (up_ready) (void)hipEventDestroy(up_ready);
        if (up_consumed) (void)hipEventDestroy(up_consumed);
        for (int i = 0; i < NBUILD; ++i) {
            if (bstream[i]) (void)hipStreamDestroy(bstream[i]);
            if (bev[i]) (void)hipEventDestroy(bev[i]);
        }
        if (bstart) (void)hipEventDestroy(bstart);
        if (ldone) (void)hipEventDestroy(ldone);
        if (fstart) (void)hipEventDestroy(fstart);
        if (fdone) (void)hipEventDestroy(fdone);
        if (fstream) (void)hipStreamDestroy(fstream);
    }
    int build_pose_table(const std::vector<float> &ori, const std::vector<float> &pos, DevBuf *dst, DevBuf *dst_tiled, hipStream_t s);
    int common_pre(const float *x_dev, float t_scaled, const int *objidx_host, int n_chains, hipStream_t s);
    // starts of `n_calls` classifier calls per chain: call k of chain c at starts_host + k * call_stride + c * 2 * rows; on the device the
    // chain's rows of all calls form one run of n_calls * rows rows (row k * rows + r)
    int upload_starts(const int64_t *starts_host, int n_chains, int64_t rows, hipStream_t s, bool need_order = true, int n_calls = 1,
                      int64_t call_stride = 0);
    int ensure_rows(int n_chains, int64_t rows_per_chain);      // grows the per-row device buffers and the pinned staging area
    // true + p filled when every chain's object has its embedding table in the wanted format: then no per-step gather runs at all
    int use_xtab(const int *objidx_host, int n_chains, int64_t rows, bool want16, dgdm::TrunkParams *p, bool *ok, hipStream_t s);
    int build_object(int oi, int slot, hipStream_t s);
    int build_xtab(int oi, hipStream_t s);
    int run_xobj(const int *objidx_host, int n_chains, int64_t rows, bool want16, bool *used16, hipStream_t s);
    // the embeddings of `n_calls` cond_fn calls at once: afterwards call k reads rows [k * R, (k + 1) * R) of every chain
    struct Embedded { bool tab = false, used16 = false; int64_t rows_per_chain = 0; };
    int embed(const int *objidx_host, int n_chains, const int64_t *starts_host, int n_calls, int64_t call_stride, Embedded *e, hipStream_t s);
};

int DgdmGuidance::build_pose_table(const std::vector<float> &ori, const std::vector<float> &pos, DevBuf *dst, DevBuf *dst_tiled, hipStream_t s) {
    const int n = (int)ori.size(), W1 = m->W1;
    DevBuf d_ori, d_pos, d_emb;
    int rc;
    if ((rc = d_ori.upload(ori.data(), sizeof(float) * n))) return rc;
    if ((rc = d_pos.upload(pos.data(), sizeof(float) * 2 * n))) return rc;
    if ((rc = d_emb.alloc(sizeof(float) * 27 * n))) return rc;
    if ((rc = dst->alloc(sizeof(float) * (size_t)W1 * n))) return rc;
    if ((rc = pose_embed(d_ori.as<float>(), d_pos.as<float>(), d_emb.as<float>(), n, s))) return rc;
    if ((rc = linear64(d_emb.as<float>(), nullptr, 27, m->blob64.at(m->off64.w1p_wt), nullptr, nullptr, 1, nullptr, dst->as<float>(), W1, n, 27, W1, ACT_NONE, s))) return rc;
    if ((rc = dst_tiled->alloc(sizeof(float) * (size_t)W1 * ((n + 31) / 32) * 32))) return rc;
    if ((rc = tile_table(dst->as<float>(), n, W1, dst_tiled->as<float>(), s))) return rc;
    DGDM_HIP_CHECK(hipStreamSynchronize(s));     // temporaries die here
    return DGDM_OK;
}

extern "C" int dgdm_guidance_create(DgdmGuidance **out, DgdmDynamics *model, const DgdmGuidanceConfig *cfg) {
    DGDM_REQUIRE(out && model && cfg, DGDM_EINVAL, "dgdm_guidance_create: null argument");
    DGDM_REQUIRE(cfg->batch > 0 && cfg->grid_size > 0 && cfg->num_pos > 0 && cfg->max_chains > 0 && cfg->num_train_timesteps > 0,
                 DGDM_EINVAL, "dgdm_guidance_create: bad config");
    DGDM_REQUIRE(cfg->max_chains <= DGDM_MAX_CHAINS, DGDM_EINVAL, "max_chains %d > %d", cfg->max_chains, DGDM_MAX_CHAINS);
    if (model->kind == 3) {
        DGDM_REQUIRE(cfg->sub_batch_size > 0, DGDM_EINVAL, "3-D guidance needs sub_batch_size (it partitions the FPS start draws)");
        DGDM_REQUIRE(cfg->num_object_points > 0 && cfg->num_object_points <= 1024, DGDM_EINVAL, "object clouds of %d points unsupported (1..1024)", cfg->num_object_points);
    } else {
        DGDM_REQUIRE(2 * cfg->num_object_points == model->object_ch, DGDM_EINVAL, "2*num_object_points (%d) != object_ch (%d)", 2 * cfg->num_object_points, model->object_ch);
    }
    std::unique_ptr<DgdmGuidance> g(new DgdmGuidance());
    g->m = model; g->cfg = *cfg;
    g->B = cfg->batch; g->G = cfg->grid_size;
    const int P = cfg->num_pos;
    g->C = g->G * P * P;
    g->R = (int64_t)g->B * g->C; g->Rs = (int64_t)g->B * g->G;
    g->tiles_per_b = (g->C + 31) / 32; g->sweep_tiles_per_b = (g->G + 31) / 32;
    // pose grid: torch.meshgrid(linspace(ori), linspace(-1,1,P), linspace(-1,1,P)) 'ij' -> cell = (g*P + px)*P + py  (diffusion.py:478)
    const std::vector<float> lo = linspace_f32(cfg->ori_lo, cfg->ori_hi, g->G), lp = linspace_f32(-1.f, 1.f, P);
    std::vector<float> ori(g->C), pos(2 * (size_t)g->C);
    for (int gi = 0; gi < g->G; ++gi)
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) {
                const int c = (gi * P + a) * P + b;
                ori[c] = lo[gi]; pos[2 * c] = lp[a]; pos[2 * c + 1] = lp[b];
            }
    int rc;
    if ((rc = g->build_pose_table(ori, pos, &g->ptab, &g->ptab_t, nullptr))) return rc;
    {   // per-cell magnitude bound of the pose table (built once: the pose grid is fixed)
        std::vector<float> tab((size_t)g->C * model->W1), mx(g->C, 0.f);
        DGDM_HIP_CHECK(hipMemcpy(tab.data(), g->ptab.p, tab.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int c = 0; c < g->C; ++c)
            for (int j = 0; j < model->W1; ++j) mx[c] = std::max(mx[c], std::fabs(tab[(size_t)c * model->W1 + j]));
        if ((rc = g->pmax.upload(mx.data(), mx.size() * sizeof(float)))) return rc;
    }
    std::vector<float> pos0(2 * (size_t)g->G, 0.f);                       // get_convergence_centers: pos = 0 (:511)
    if ((rc = g->build_pose_table(lo, pos0, &g->ptab_sweep, &g->ptab_sweep_t, nullptr))) return rc;
    const int W1 = model->W1, nc = cfg->max_chains;
    const size_t rows = (size_t)nc * g->B;
    if ((rc = g->V.alloc(rows * 256 * 8)) || (rc = g->genc.alloc(rows * 256 * 8)) || (rc = g->atab.alloc(rows * W1 * 4)) ||
        (rc = g->chainbias.alloc((size_t)nc * W1 * 8)) || (rc = g->timepart.alloc((size_t)W1 * 8)) || (rc = g->ttmp.alloc(768 * 4)) ||
        (rc = g->ttmp64.alloc(512 * 8)) ||
        (rc = g->partial.alloc(rows * g->tiles_per_b * W1 * 4)) || (rc = g->objdev.alloc(sizeof(TrunkObjective) * nc)) ||
        (rc = g->objidx.alloc(sizeof(int) * nc)))
        return rc;
    if (model->kind == 2) {
        if ((rc = g->objpart.alloc((size_t)std::max(1, cfg->max_objects) * W1 * 8))) return rc;
    } else {
        if ((rc = g->starts.alloc((size_t)nc * g->R * 2 * sizeof(int))) ||
            (rc = g->order.alloc((size_t)nc * g->R * sizeof(int))) || (rc = g->xchains.alloc(sizeof(XobjChain) * nc)) ||
            (rc = g->todo.alloc(((size_t)nc * g->R + 1) * sizeof(int))))
            return rc;
        g->todo_capacity = (int64_t)nc * g->R;
        g->pinned_bytes = ((size_t)nc * g->R * 3 + (size_t)nc * (cfg->num_object_points + 1)) * sizeof(int);
        if ((rc = g->groupoff.alloc((size_t)nc * (cfg->num_object_points + 1) * sizeof(int))) || (rc = g->xidx.alloc((size_t)nc * g->R * sizeof(int))) ||
            (rc = g->xidxchains.alloc(sizeof(XidxChain) * nc)) || (rc = g->xtabptrs.alloc(sizeof(void *) * nc)))
            return rc;
        DGDM_HIP_CHECK(hipHostMalloc(&g->pinned, g->pinned_bytes, hipHostMallocDefault));
        DGDM_HIP_CHECK(hipEventCreateWithFlags(&g->pinned_ev, hipEventDisableTiming));
    }
    *out = g.release();
    return DGDM_OK;
}

extern "C" void dgdm_guidance_destroy(DgdmGuidance *g) { delete g; }

extern "C" int dgdm_guidance_set_contraction_dtype(DgdmGuidance *g, int dtype) {
    DGDM_REQUIRE(g, DGDM_EINVAL, "dgdm_guidance_set_contraction_dtype: null handle");
    DGDM_REQUIRE(dtype >= DGDM_DTYPE_F32 && dtype <= DGDM_DTYPE_F32_F16X3, DGDM_EINVAL,
                 "contraction dtype %d unsupported (0 = f32 (default form), 1 = bf16, 2 = f32 on the float32 MFMA, 3 = f32 as 3 f16 products = the default form)", dtype);
    g->bf16 = dtype == DGDM_DTYPE_BF16;
    g->f32_mfma = dtype == DGDM_DTYPE_F32_MFMA;
    return DGDM_OK;
}

extern "C" int dgdm_guidance_debug_fps_path(DgdmGuidance *g, int force_per_row, int32_t *out_fast_ok) {
    DGDM_REQUIRE(g, DGDM_EINVAL, "dgdm_guidance_debug_fps_path: null handle");
    g->force_slow_xobj = force_per_row == 1;            // 1: every row runs its own FPS; 2: per-row table kernel; 0: default (group kernel)
    g->xobj_mode = force_per_row == 2 ? 2 : 0;
    g->l2_gather_mode = force_per_row == 4 ? 1 : 0;     // takes effect at the next dgdm_guidance_set_objects
    g->xtab_enabled = force_per_row == 0 || force_per_row == 4 || force_per_row == 5;      // modes 1-3 read materialised rows; 3 = the group gather kernel
    g->xtab_policy = force_per_row == 5 ? 1 : 0;         // 5: the next set_objects builds the embedding tables right away
    if (force_per_row == 3) g->xobj_mode = 0;
    if (out_fast_ok) {
        int rc = g->finish_objects();
        if (rc) return rc;
        for (int i = 0; i < g->n_objects && i < (int)g->tables.size(); ++i) out_fast_ok[i] = g->tables[i]->fast_ok ? 1 : 0;
    }
    return DGDM_OK;
}
// d objective / d (scaled z1) -> d objective / d z1: x 2^e_j per column (the trunk is equilibrated by exact powers of two, models_api.hip TrunkEquil)
__global__ void partials_true_units_kernel(float *p, const float *unit, size_t n, int W1) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] *= unit[i % W1];
}
// Test hook: the per-tile partial sums of d objective / d z1 the last dgdm_dyn{2,3}d_guidance_grad call left behind
// ([n_chains * B * tiles_per_b][W1], tile = (chain * B + b) * tiles_per_b + cell tile; a tile is 32 consecutive pose cells of one finger).
extern "C" int dgdm_guidance_debug_partials(DgdmGuidance *g, int n_chains, float *out_dev, int32_t *tiles_per_finger, int32_t *width, void *stream) {
    DGDM_REQUIRE(g && n_chains > 0 && n_chains <= g->cfg.max_chains, DGDM_EINVAL, "dgdm_guidance_debug_partials: bad argument");
    if (tiles_per_finger) *tiles_per_finger = g->tiles_per_b;
    if (width) *width = g->m->W1;
    if (out_dev) {
        const size_t n = (size_t)n_chains * g->B * g->tiles_per_b * g->m->W1;
        DGDM_HIP_CHECK(hipMemcpyAsync(out_dev, g->partial.p, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        hipLaunchKernelGGL(partials_true_units_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out_dev, g->m->z1_unit.as<float>(), n, g->m->W1);
        DGDM_HIP_CHECK(hipGetLastError());
    }
    return DGDM_OK;
}
extern "C" int64_t dgdm_guidance_rows(const DgdmGuidance *g) { return g ? g->R : 0; }
extern "C" int64_t dgdm_guidance_starts_per_call(const DgdmGuidance *g) { return (g && g->m->kind == 3) ? 2 * g->R : 0; }

// ------------------------------------------------------------------------------------------------ objects
int DgdmGuidance::build_object(int oi, int slot, hipStream_t s) {
    const int N = cfg.num_object_points;
    ObjectTables &t = *tables[oi];
    const PnWeights w = m->pn();
    int rc;
    if ((rc = t.Z.alloc((size_t)N * N * 256 * 4)) ||
        (rc = t.M0.alloc((size_t)N * 256 * 4)) || (rc = t.cl2.alloc((size_t)N * 128 * sizeof(int))) || (rc = t.cnt2.alloc((size_t)N * sizeof(int))) ||
        (rc = t.cl2s.alloc((size_t)N * 128 * sizeof(int))) || (rc = t.cl2o.alloc((size_t)N * 256 * sizeof(unsigned short))) ||
        (rc = t.pcf.alloc((size_t)N * 512 * sizeof(int))))
        return rc;
    t.has16 = bf16;
    if (bf16 && ((rc = t.Z16.alloc((size_t)N * N * 128 * 4)) || (rc = t.M0_16.alloc((size_t)N * 128 * 4)))) return rc;
    uint32_t *z16 = bf16 ? t.Z16.as<uint32_t>() : nullptr;
    DevBuf &tY = tmpY[slot], &tL2 = tmpL2[slot];
    if ((rc = tY.alloc((size_t)N * N * 256 * 4)) || (rc = tL2.alloc((size_t)N * N * 256 * 4))) return rc;
    const float *xyz = t.xyz;                                                                                  // T1: set_objects, batched
    // crowded flags / pair lists (crowd_kernel, nbr_fill_kernel), T2 (sa1) and T3 (U): set_objects, one launch per stage for all objects
    const int *off = pool_off.as<int>() + (size_t)oi * (N + 1), *pairs = pool_pairs.as<int>() + (size_t)oi * N * N;
    const short *rank = pool_rank.as<short>() + (size_t)oi * N * N;
    const char *U = static_cast<const char *>(pool_U.p) + (size_t)oi * N * 128 * (bf16 ? 4 : 8);      // float32 rows in bf16 mode, float64 otherwise
    if (bf16) {
        // bf16 mode: the sa3 contraction (T6) runs on the bf16 matrix pipe and rounds its input, so T4 writes and T5 reduces bf16
        // rows (the temporaries tY / tL2 are simply used at half size); the stages in front of it stay float32
        if ((rc = pn_pairs(xyz, N, reinterpret_cast<const float *>(U), w, pairs, off, tY.as<float>(), tY.as<uint32_t>(), s))) return rc;   // T4
        if ((rc = pn_l2(xyz, N, w, t.fps1, vlist.as<int>(), N, tY.as<float>(), tL2.as<float>(), t.clist, t.clist + N,
                        off, rank, true, s, l2_gather_mode ? 0 : 1))) return rc;                   // T5
        if ((rc = pn_z16(xyz, N, N, w, tL2.as<uint32_t>(), t.Z.as<float>(), z16, t.clist, t.clist + N, s))) return rc;          // T6
    } else {
        // float32 mode (the parity path): every contraction of the build accumulates in float64 and each table entry is rounded once
        // (pointnet64.hip); the max stages (T5, T7) are exact as they are
        const PnWeights64 w64 = m->pn64();
        if ((rc = pn_pairs64(xyz, N, reinterpret_cast<const double *>(U), w64, pairs, off, tY.as<float>(), s))) return rc;                 // T4
        if ((rc = pn_l2(xyz, N, w, t.fps1, vlist.as<int>(), N, tY.as<float>(), tL2.as<float>(), t.clist, t.clist + N,
                        off, rank, false, s, l2_gather_mode ? 0 : 1))) return rc;                  // T5
        if ((rc = pn_z64(xyz, N, N, w64, tL2.as<float>(), t.Z.as<float>(), t.clist, t.clist + N, s))) return rc;                // T6
    }
    DGDM_HIP_CHECK(hipStreamWaitEvent(s, fdone, 0));          // fps2 (sa2's FPS table) is built beside the other stages, on its own stream
    if ((rc = pn_m0(t.fps2, t.crowded, N, t.Z.as<float>(), t.M0.as<float>(), t.cl2.as<int>(), t.cnt2.as<int>(), z16,
                    bf16 ? t.M0_16.as<uint32_t>() : nullptr, t.clist, t.clist + N, t.cl2s.as<int>(), t.cl2o.as<unsigned short>(), s))) return rc;          // T7
    if ((rc = pn_pcf(t.fps1, t.cnt2.as<int>(), t.flags, N, t.pcf.as<int>(), s))) return rc;
    t.has_x = t.has_x16 = false;
    return xtab_policy == 1 ? build_xtab(oi, s) : DGDM_OK;       // eager only on request: see guidance_grad for when it pays
}

// T8: the embedding table of object oi in the format the trunk of the object's build mode reads (float32 for the parity path, bf16
// operand-order rows when the object was built in bf16 mode)
int DgdmGuidance::build_xtab(int oi, hipStream_t s) {
    const int N = cfg.num_object_points;
    ObjectTables &t = *tables[oi];
    if (N < 128) return DGDM_OK;
    const bool b16 = t.has16;
    int rc;
    if (b16) { if ((rc = t.X16.alloc((size_t)N * N * 128 * 4))) return rc; }
    else if ((rc = t.X.alloc((size_t)N * N * 256 * 4))) return rc;
    XtabObj xo{};
    xo.xyz = t.xyz; xo.fps1 = t.fps1; xo.Z = t.Z.as<float>(); xo.M0 = t.M0.as<float>();
    xo.Z16 = b16 ? t.Z16.as<uint32_t>() : nullptr; xo.M0_16 = b16 ? t.M0_16.as<uint32_t>() : nullptr;
    xo.clist = t.clist; xo.ncr = t.clist + N; xo.cl2s = t.cl2s.as<int>(); xo.cnt2 = t.cnt2.as<int>(); xo.flags = t.flags;
    xo.crowded = t.crowded; xo.X = b16 ? nullptr : t.X.as<float>(); xo.X16 = b16 ? t.X16.as<uint32_t>() : nullptr; xo.N = N;
    if ((rc = pn_xtab(xo, b16, s))) return rc;
    (b16 ? t.has_x16 : t.has_x) = true;
    return DGDM_OK;
}

extern "C" int dgdm_guidance_set_objects(DgdmGuidance *g, const float *objects_dev, int n_objects, void *stream) {
    DGDM_REQUIRE(g && objects_dev && n_objects > 0, DGDM_EINVAL, "dgdm_guidance_set_objects: bad argument");
    DGDM_REQUIRE(n_objects <= std::max(1, g->cfg.max_objects), DGDM_EINVAL, "%d objects > max_objects %d", n_objects, g->cfg.max_objects);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    host_stamp("set_objects: enter");
    prof_begin(s, DGDM_STAGE_TABLES);
    if (g->m->kind == 2) {
        // (a grow-only member, not a local: a local's hipFree - and the synchronize that had to precede it - stalled the host behind
        //  the previous batch's chains at every call)
        if ((rc = g->objtmp.alloc((size_t)n_objects * 512 * 8))) return rc;
        if ((rc = g->m->object_part_2d64(objects_dev, g->objtmp.as<double>(), g->objpart.as<double>(), n_objects, s))) return rc;
        prof_end(s, DGDM_STAGE_TABLES, 0.0);
    } else {
        while ((int)g->tables.size() < n_objects) g->tables.emplace_back(new ObjectTables());
        const int N = g->cfg.num_object_points;
        if (g->vlist.bytes < (size_t)N * sizeof(int)) {
            std::vector<int> v(N);
            for (int i = 0; i < N; ++i) v[i] = i;
            if ((rc = g->vlist.upload(v.data(), sizeof(int) * N))) return rc;
        }
        // objects are independent: build them round-robin on a few side streams so the latency-bound stages (FPS: 512
        // dependent iterations on 128 workgroups) of one object overlap the bandwidth/MFMA-bound stages of the others
        const int nb = std::min<int>(g->nbuild, n_objects);
        if (!g->bstart) DGDM_HIP_CHECK(hipEventCreateWithFlags(&g->bstart, hipEventDisableTiming));
        if (!g->ldone) DGDM_HIP_CHECK(hipEventCreateWithFlags(&g->ldone, hipEventDisableTiming));
        for (int i = 0; i < nb; ++i) {
            if (!g->bstream[i]) DGDM_HIP_CHECK(hipStreamCreateWithFlags(&g->bstream[i], hipStreamNonBlocking));
            if (!g->bev[i]) DGDM_HIP_CHECK(hipEventCreateWithFlags(&g->bev[i], hipEventDisableTiming));
        }
        // T1 for every object at once, on the caller's stream: fps1[obj][start][512], fps2[obj][start point][128] + tie flags
        const size_t no = (size_t)n_objects;
        if ((rc = g->pool_xyz.alloc(no * N * 3 * 4)) || (rc = g->pool_fps1.alloc(no * N * 512 * sizeof(int))) ||
            (rc = g->pool_fps2.alloc(no * N * 128 * sizeof(int))) || (rc = g->pool_flags.alloc(no * N * sizeof(int))) ||
            (rc = g->pool_ncr.alloc(no * sizeof(int))) || (rc = g->pool_crowded.alloc(no * N * sizeof(int))) ||
            (rc = g->pool_clist.alloc(no * (N + 1) * sizeof(int))) || (rc = g->pool_off.alloc(no * (N + 1) * sizeof(int))) ||
            (rc = g->pool_pairs.alloc(no * N * N * sizeof(int))) || (rc = g->pool_rank.alloc(no * N * N * sizeof(short))) ||
            (rc = g->pool_F1.alloc(no * N * 128 * 8)) || (rc = g->pool_U.alloc(no * N * 128 * 8)))
            return rc;
        if (!g->fstream) {
            DGDM_HIP_CHECK(hipStreamCreateWithFlags(&g->fstream, hipStreamNonBlocking));
            DGDM_HIP_CHECK(hipEventCreateWithFlags(&g->fstart, hipEventDisableTiming));
            DGDM_HIP_CHECK(hipEventCreateWithFlags(&g->fdone, hipEventDisableTiming));
        }
        DGDM_HIP_CHECK(hipMemcpyAsync(g->pool_xyz.p, objects_dev, no * N * 3 * 4, hipMemcpyDeviceToDevice, s));
        // sa2's FPS table (fps2) is only read by the last build stage (m0): it runs on its own stream beside sa1's table and the builds
        DGDM_HIP_CHECK(hipEventRecord(g->fstart, s));
        DGDM_HIP_CHECK(hipStreamWaitEvent(g->fstream, g->fstart, 0));
        if ((rc = pn_fps_table(g->pool_xyz.as<float>(), N, N, 128, g->pool_fps2.as<int>(), g->pool_flags.as<int>(), g->fstream, n_objects))) return rc;
        DGDM_HIP_CHECK(hipEventRecord(g->fdone, g->fstream));
        if ((rc = pn_fps_table(g->pool_xyz.as<float>(), N, N, 512, g->pool_fps1.as<int>(), nullptr, s, n_objects))) return rc;
        for (int i = 0; i < n_objects; ++i) {
            ObjectTables &t = *g->tables[i];
            t.xyz = g->pool_xyz.as<float>() + (size_t)i * N * 3; t.fps1 = g->pool_fps1.as<int>() + (size_t)i * N * 512;
            t.fps2 = g->pool_fps2.as<int>() + (size_t)i * N * 128; t.flags = g->pool_flags.as<int>() + (size_t)i * N;
            t.crowded = g->pool_crowded.as<int>() + (size_t)i * N; t.clist = g->pool_clist.as<int>() + (size_t)i * (N + 1);
        }
        // the light, latency-bound stages - crowded flags + pair lists, T2 (sa1 features), T3 (U = sa2's first layer on the features) -
        // for ALL objects in one launch each (per object they are 1 .. 512 small workgroups: 130-500 us of a build stream each, a third
        // of its time), on the first build stream, beside the FPS tables; the heavy per-object stages (T4 .. T7) follow on the build streams
        {
            hipStream_t ls = g->bstream[0];
            DGDM_HIP_CHECK(hipStreamWaitEvent(ls, g->fstart, 0));          // the objects' coordinates are in the pool
            const PnWeights w = g->m->pn();
            if ((rc = pn_crowd(g->pool_xyz.as<float>(), N, w, g->pool_crowded.as<int>(), g->pool_clist.as<int>(), g->pool_clist.as<int>() + N,
                               g->pool_off.as<int>(), g->pool_pairs.as<int>(), g->pool_rank.as<short>(), ls, g->pool_ncr.as<int>(), n_objects)))
                return rc;
            if (g->bf16) {
                if ((rc = pn_sa1(g->pool_xyz.as<float>(), N, w, g->pool_F1.as<float>(), ls, n_objects))) return rc;                                   // T2
                if ((rc = linear(g->pool_F1.as<float>(), 128, w.sa2_wf_t, w.sa2_b0, nullptr, 1, g->pool_U.as<float>(), 128, n_objects * N, 128, 128,
                                 ACT_NONE, false, ls))) return rc;                                                                                 // T3
            } else {
                const PnWeights64 w64 = g->m->pn64();
                if ((rc = pn_sa1_64(g->pool_xyz.as<float>(), N, w.r1sq, w64, g->pool_F1.as<double>(), ls, n_objects))) return rc;                      // T2
                if ((rc = linear64(nullptr, g->pool_F1.as<double>(), 128, w64.sa2_wf_t, w64.sa2_b0, nullptr, 1, g->pool_U.as<double>(), nullptr, 128,
                                   n_objects * N, 128, 128, ACT_NONE, ls))) return rc;                                                             // T3
            }
            DGDM_HIP_CHECK(hipEventRecord(g->ldone, ls));
        }
        DGDM_HIP_CHECK(hipEventRecord(g->bstart, s));
        for (int i = 0; i < nb; ++i) {
            DGDM_HIP_CHECK(hipStreamWaitEvent(g->bstream[i], g->bstart, 0));
            DGDM_HIP_CHECK(hipStreamWaitEvent(g->bstream[i], g->ldone, 0));
        }
        for (int i = 0; i < n_objects; ++i)
            if ((rc = g->build_object(i, i % nb, g->bstream[i % nb]))) return rc;
        for (int i = 0; i < nb; ++i) {
            DGDM_HIP_CHECK(hipEventRecord(g->bev[i], g->bstream[i]));
            DGDM_HIP_CHECK(hipStreamWaitEvent(s, g->bev[i], 0));
        }
        DGDM_HIP_CHECK(hipStreamWaitEvent(s, g->fdone, 0));
        // which objects may use the table of FPS(128) sequences (no order-dependent selection anywhere); crowded-centre counts:
        // copied to pinned memory behind an event, read by finish_objects() when first needed
        prof_end(s, DGDM_STAGE_TABLES, 0.0);
        const size_t need = no * N + no;
        if (need > g->ro_ints) {
            if (g->ro_host) DGDM_HIP_CHECK(hipHostFree(g->ro_host));
            g->ro_host = nullptr;
            DGDM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&g->ro_host), need * sizeof(int), hipHostMallocDefault));
            g->ro_ints = need;
        }
        if (!g->ro_ev) DGDM_HIP_CHECK(hipEventCreateWithFlags(&g->ro_ev, hipEventDisableTiming));
        DGDM_HIP_CHECK(hipMemcpyAsync(g->ro_host, g->pool_flags.p, sizeof(int) * no * N, hipMemcpyDeviceToHost, s));
        DGDM_HIP_CHECK(hipMemcpyAsync(g->ro_host + no * N, g->pool_ncr.p, sizeof(int) * no, hipMemcpyDeviceToHost, s));
        DGDM_HIP_CHECK(hipEventRecord(g->ro_ev, s));
        g->ro_pending = true;
    }
    g->n_objects = n_objects;
    g->grads_since_set = 0;
    host_stamp("set_objects: return");
    return DGDM_OK;
}

// ------------------------------------------------------------------------------------------------ shared front end
// V/genc from x, timepart(t), chain bias (2-D: object part + time part), A table.
int DgdmGuidance::common_pre(const float *x_dev, float t_scaled, const int *objidx_host, int n_chains, hipStream_t s) {
    const int rows = n_chains * B, W1 = m->W1;
    int rc;
    if ((rc = m->gripper_forward64(x_dev, m->L, V.as<double>(), genc.as<double>(), rows, s))) return rc;
    if ((rc = m->time_part64(t_scaled, ttmp.as<float>(), ttmp64.as<double>(), timepart.as<double>(), s))) return rc;
    if (m->kind == 2) {
        for (int i = 0; i < n_chains; ++i)
            DGDM_REQUIRE(objidx_host[i] >= 0 && objidx_host[i] < n_objects, DGDM_EINVAL, "chain %d refers to object %d of %d", i, objidx_host[i], n_objects);
        DGDM_HIP_CHECK(hipMemcpyAsync(objidx.p, objidx_host, sizeof(int) * n_chains, hipMemcpyHostToDevice, s));   // pageable: staged before return
        if ((rc = gather_add64(objpart.as<double>(), objidx.as<int>(), timepart.as<double>(), chainbias.as<double>(), n_chains, W1, s))) return rc;
        return linear64(nullptr, genc.as<double>(), 256, m->blob64.at(m->off64.w1c_wt), nullptr, chainbias.as<double>(), B, nullptr, atab.as<float>(), W1, rows, 256, W1,
                        ACT_NONE, s);
    }
    return linear64(nullptr, genc.as<double>(), 256, m->blob64.at(m->off64.w1c_wt), timepart.as<double>(), nullptr, 1, nullptr, atab.as<float>(), W1, rows, 256, W1,
                    ACT_NONE, s);
}

// Reference draw order per chain: for each sub-batch i, sa1's torch.randint(rows_i) then sa2's  ->  device [chain][row][2]
int DgdmGuidance::finish_objects() {
    if (!ro_pending) return DGDM_OK;
    DGDM_HIP_CHECK(hipEventSynchronize(ro_ev));
    const int N = cfg.num_object_points;
    for (int i = 0; i < n_objects; ++i) {
        bool ok = N >= 128;
        for (int k = 0; k < N; ++k) ok = ok && ro_host[(size_t)i * N + k] == 0;
        tables[i]->fast_ok = ok;
        tables[i]->ncr = ro_host[(size_t)n_objects * N + i];
    }
    ro_pending = false;
    return DGDM_OK;
}

int DgdmGuidance::ensure_rows(int n_chains, int64_t rows_per_chain) {
    const int N = cfg.num_object_points;
    const size_t nr = (size_t)n_chains * rows_per_chain;
    int rc;
    // (the embedding rows themselves - xobj / xobj16, 1 KiB / 512 B per row - are allocated by run_xobj, the only path that writes them:
    // with the embedding tables X[s1][q] in place a call gathers nothing)
    if ((rc = starts.alloc(nr * 2 * sizeof(int))) || (rc = order.alloc(nr * sizeof(int))) ||
        (rc = todo.alloc((nr + 1) * sizeof(int))) || (rc = xidx.alloc(nr * sizeof(int))))
        return rc;
    todo_capacity = (int64_t)nr;
    const size_t need = (nr * 3 + (size_t)n_chains * (N + 1)) * sizeof(int);
    if (need > pinned_bytes) {
        if (pinned) { DGDM_HIP_CHECK(hipEventSynchronize(pinned_ev)); DGDM_HIP_CHECK(hipHostFree(pinned)); pinned = nullptr; pinned_bytes = 0; }
        DGDM_HIP_CHECK(hipHostMalloc(&pinned, need, hipHostMallocDefault));
        pinned_bytes = need;
    }
    return DGDM_OK;
}

int DgdmGuidance::upload_starts(const int64_t *starts_host, int n_chains, int64_t rows, hipStream_t s, bool need_order, int n_calls,
                                int64_t call_stride) {
    const int N = cfg.num_object_points;
    const int64_t sb = cfg.sub_batch_size;
    const int64_t rt = rows * n_calls;                         // rows per chain on the device
    int rc;
    DGDM_REQUIRE(!need_order || rt < ((int64_t)1 << 22), DGDM_EINVAL, "%lld rows per chain in one gather launch (limit 4194303)", (long long)rt);
    if ((rc = ensure_rows(n_chains, rt))) return rc;
    DGDM_HIP_CHECK(hipEventSynchronize(pinned_ev));            // previous copy out of the staging buffer has finished
    int *dst = static_cast<int *>(pinned);
    int *ord = dst + (size_t)n_chains * 2 * rt;
    int *goff = ord + (size_t)n_chains * rt;                   // [n_chains][N+1]: where each s1-group starts in the chain's sorted rows
    // per chain: int64 draws -> (s1, s2) int32 pairs in row order; rows sorted by s1 (counting sort) so that the rows of one
    // variant gather from the same Z slab; group offsets.  Chains are independent: a few host threads share them.
    std::atomic<int> bad{0};
    auto work = [&](int c0, int c1) {
        std::vector<int> cnt(N + 1), cnt2(513), tmp;
        for (int c = c0; c < c1; ++c) {
            int *d = dst + (size_t)c * 2 * rt;
            for (int k = 0; k < n_calls; ++k) {
                const int64_t *src = starts_host + (size_t)k * call_stride + (size_t)c * 2 * rows;
                int *dk = d + 2 * (size_t)k * rows;
                for (int64_t r0 = 0; r0 < rows; r0 += sb) {
                    const int64_t n = std::min(sb, rows - r0);
                    const int64_t *s1 = src + 2 * r0, *s2 = s1 + n;
                    for (int64_t i = 0; i < n; ++i) {
                        const int64_t a = s1[i], b = s2[i];
                        if (!(a >= 0 && a < N && b >= 0 && b < 512)) { bad.store(1); return; }
                        dk[2 * (r0 + i)] = (int)a; dk[2 * (r0 + i) + 1] = (int)b;
                    }
                }
            }
            if (!need_order) continue;                      // embedding-table path: rows are looked up where they are
            // rows sorted by (s1, s2): two stable counting sorts, s2 first.  Rows of one variant s1 gather from the same Z slab, and
            // rows that share both draws are the same row (xobj_rows_kernel computes a run of them once)
            int *o = ord + (size_t)c * rt;
            tmp.resize((size_t)rt);
            std::fill(cnt2.begin(), cnt2.end(), 0);
            for (int64_t r = 0; r < rt; ++r) ++cnt2[d[2 * r + 1] + 1];
            for (int k = 0; k < 512; ++k) cnt2[k + 1] += cnt2[k];
            for (int64_t r = 0; r < rt; ++r) tmp[cnt2[d[2 * r + 1]]++] = (int)r;
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int64_t r = 0; r < rt; ++r) ++cnt[d[2 * r] + 1];
            for (int k = 0; k < N; ++k) cnt[k + 1] += cnt[k];
            memcpy(goff + (size_t)c * (N + 1), cnt.data(), sizeof(int) * (N + 1));
            for (int64_t i = 0; i < rt; ++i) { const int r = tmp[i]; o[cnt[d[2 * r]]++] = r | (d[2 * r + 1] << 22); }      // row id | s2 << 22
        }
    };
    static const bool host_timing = getenv("DGDM_HOST_TIMING") != nullptr;      // experiment hook: the host's conversion + sort time per call
    const auto ht0 = std::chrono::steady_clock::now();
    const int hw = (int)std::max(2u, std::thread::hardware_concurrency());
    const int nthreads = (int)std::min<int64_t>(std::min(16, hw / 2), std::max<int64_t>(1, std::min<int64_t>(n_chains, (int64_t)n_chains * rt / 65536)));
    if (nthreads <= 1) {
        work(0, n_chains);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(work, (int)((int64_t)n_chains * t / nthreads), (int)((int64_t)n_chains * (t + 1) / nthreads));
        for (auto &th : pool) th.join();
    }
    if (host_timing)
        fprintf(stderr, "upload_starts: %d chains x %lld rows, %d threads, need_order %d: host %.2f ms\n", n_chains, (long long)rt, nthreads, (int)need_order,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ht0).count());
    DGDM_REQUIRE(!bad.load(), DGDM_EINVAL, "FPS start out of range (sa1 must be in [0, %d), sa2 in [0, 512))", N);
    if (!cstream) {
        DGDM_HIP_CHECK(hipStreamCreateWithFlags(&cstream, hipStreamNonBlocking));
        DGDM_HIP_CHECK(hipEventCreateWithFlags(&up_ready, hipEventDisableTiming));
        DGDM_HIP_CHECK(hipEventCreateWithFlags(&up_consumed, hipEventDisableTiming));
    }
    // the copy may start as soon as the previous readers of the device buffers (the last gather / index kernel on the launch stream)
    // are done - not behind everything queued on the launch stream since
    if (consumed_recorded) DGDM_HIP_CHECK(hipStreamWaitEvent(cstream, up_consumed, 0));
    DGDM_HIP_CHECK(hipMemcpyAsync(starts.p, pinned, (size_t)n_chains * rt * 2 * sizeof(int), hipMemcpyHostToDevice, cstream));
    if (need_order) {
        DGDM_HIP_CHECK(hipMemcpyAsync(order.p, ord, (size_t)n_chains * rt * sizeof(int), hipMemcpyHostToDevice, cstream));
        DGDM_HIP_CHECK(hipMemcpyAsync(groupoff.p, goff, (size_t)n_chains * (N + 1) * sizeof(int), hipMemcpyHostToDevice, cstream));
    }
    DGDM_HIP_CHECK(hipEventRecord(pinned_ev, cstream));
    DGDM_HIP_CHECK(hipEventRecord(up_ready, cstream));
    DGDM_HIP_CHECK(hipStreamWaitEvent(s, up_ready, 0));
    return DGDM_OK;
}

int DgdmGuidance::use_xtab(const int *objidx_host, int n_chains, int64_t rows, bool want16, TrunkParams *p, bool *ok, hipStream_t s) {
    *ok = false;
    if (!xtab_enabled || force_slow_xobj || xobj_mode != 0) return DGDM_OK;
    std::vector<XidxChain> xc(n_chains);
    std::vector<const void *> base(n_chains);
    for (int i = 0; i < n_chains; ++i) {
        DGDM_REQUIRE(objidx_host[i] >= 0 && objidx_host[i] < n_objects, DGDM_EINVAL, "chain %d refers to object %d of %d", i, objidx_host[i], n_objects);
        const ObjectTables &t = *tables[objidx_host[i]];
        if (!(want16 ? t.has_x16 : t.has_x)) return DGDM_OK;          // built in the other format (or not at all): gather kernels
        xc[i].fps1 = t.fps1; xc[i].N = cfg.num_object_points; xc[i].m0_only = t.ncr == 0;
        // an object without crowded centres has X[s1][q] = M0[q]: its table is M0 itself (xtab_kernel wrote nothing)
        base[i] = want16 ? (t.ncr == 0 ? (const void *)t.M0_16.p : (const void *)t.X16.p) : (t.ncr == 0 ? (const void *)t.M0.p : (const void *)t.X.p);
    }
    DGDM_HIP_CHECK(hipMemcpyAsync(xidxchains.p, xc.data(), sizeof(XidxChain) * n_chains, hipMemcpyHostToDevice, s));      // pageable: staged before return
    DGDM_HIP_CHECK(hipMemcpyAsync(xtabptrs.p, base.data(), sizeof(void *) * n_chains, hipMemcpyHostToDevice, s));
    int rc;
    if ((rc = pn_xidx(xidxchains.as<XidxChain>(), starts.as<int>(), rows, n_chains, xidx.as<int>(), s))) return rc;
    p->xidx = xidx.as<int>();
    if (want16) p->xtab16 = xtabptrs.as<const uint32_t *>();
    else p->xtab = xtabptrs.as<const float *>();
    *ok = true;
    return DGDM_OK;
}

int DgdmGuidance::run_xobj(const int *objidx_host, int n_chains, int64_t rows, bool want16, bool *used16, hipStream_t s) {
    std::vector<XobjChain> ch(n_chains);
    for (int i = 0; i < n_chains; ++i) {
        DGDM_REQUIRE(objidx_host[i] >= 0 && objidx_host[i] < n_objects, DGDM_EINVAL, "chain %d refers to object %d of %d", i, objidx_host[i], n_objects);
        const ObjectTables &t = *tables[objidx_host[i]];
        ch[i].xyz = t.xyz; ch[i].fps1 = t.fps1; ch[i].slot_of_start = nullptr; ch[i].Z = t.Z.as<float>();
        ch[i].fps2 = t.fps2; ch[i].flags = t.flags; ch[i].crowded = t.crowded; ch[i].N = cfg.num_object_points;
        ch[i].M0 = t.M0.as<float>(); ch[i].cl2 = t.cl2.as<int>(); ch[i].cnt2 = t.cnt2.as<int>();
        ch[i].Z16 = t.has16 ? t.Z16.as<uint32_t>() : nullptr; ch[i].M0_16 = t.has16 ? t.M0_16.as<uint32_t>() : nullptr;
        want16 = want16 && t.has16;          // bf16 rows only if every chain's object was built with its bf16 tables
    }
    {   // grow-only buffers (a run's embeds have the same size, or shrink at its tail: no reallocation under kernels in flight)
        const size_t nrows = (size_t)n_chains * rows;
        int rc = want16 ? xobj16.alloc(nrows * 512) : xobj.alloc(nrows * 256 * 4);
        if (rc) return rc;
    }
    if (used16) *used16 = want16;
    XobjParams xp{};
    xp.chains = xchains.as<XobjChain>(); xp.starts = starts.as<int>(); xp.order = order.as<int>(); xp.xobj = xobj.as<float>();
    xp.xobj16 = want16 ? xobj16.as<uint32_t>() : nullptr;
    xp.R = rows; xp.total_rows = rows * n_chains; xp.use_table = force_slow_xobj ? 0 : 1;
    xp.todo = todo.as<int>(); xp.todo_count = todo.as<int>() + todo_capacity; xp.todo_capacity = todo_capacity;
    bool all_fast = true;
    for (int i = 0; i < n_chains; ++i) all_fast = all_fast && tables[objidx_host[i]]->fast_ok;
    // group kernel: every chain needs its tables and a slab chunk that fits LDS (it handles tie-flagged start points itself)
    bool groups = !force_slow_xobj && xobj_mode == 0 && cfg.num_object_points >= 128;
    for (int i = 0; i < n_chains && groups; ++i) {
        const ObjectTables &t = *tables[objidx_host[i]];
        const int lpr = xobj_rows_lpr(t.ncr, want16);
        groups = lpr > 0 && want16 == t.has16;     // the slot offsets are scaled for the build's mode
        ch[i].clist = t.clist; ch[i].cl2s = t.cl2s.as<int>(); ch[i].cl2o = t.cl2o.as<unsigned short>(); ch[i].pcf = t.pcf.as<int>(); ch[i].ncr = t.ncr; ch[i].lpr = lpr;
    }
    DGDM_HIP_CHECK(hipMemcpyAsync(xchains.p, ch.data(), sizeof(XobjChain) * n_chains, hipMemcpyHostToDevice, s));     // pageable: staged before return
    if (groups) {
        xp.group_off = groupoff.as<int>(); xp.nchain = n_chains; xp.group_N = cfg.num_object_points; xp.total_items = n_chains * xp.group_N; xp.use_table = 1;
        std::vector<int> rank(n_chains);
        for (int i = 0; i < n_chains; ++i) rank[i] = i;
        std::stable_sort(rank.begin(), rank.end(), [&](int a, int b) { return ch[a].ncr > ch[b].ncr; });
        for (int i = 0; i < n_chains; ++i) xp.chain_of_rank[i] = (unsigned char)rank[i];
        return pn_xobj_groups(xp, s);
    }
    return pn_xobj(xp, all_fast, s);
}

// The PointNet++ embeddings of the rows of `n_calls` consecutive cond_fn calls (3-D).  They depend on the FPS start draws and the
// objects only - not on x - so a whole denoise loop's worth can be made before its first step: one upload and ONE gather launch whose
// (chain, s1) groups hold the rows of all the calls (the variant's slab is staged once for five times the rows), or - when the objects'
// embedding tables exist - one index kernel.  Call k then reads rows [k * R, (k + 1) * R) of every chain.
int DgdmGuidance::embed(const int *oidx, int n_chains, const int64_t *starts_host, int n_calls, int64_t call_stride, Embedded *e, hipStream_t s) {
    DGDM_REQUIRE(starts_host, DGDM_EINVAL, "3-D guidance needs the FPS start indices");
    int rc;
    prof_begin(s, DGDM_STAGE_XOBJ);
    // The table X[s1][q] of an object costs about what 7 cond_fn calls spend gathering its rows (it holds all 262 144 (s1, q)
    // pairs; one call touches 36 000 of them): it pays as soon as the objects serve more than one 5-step chain - the
    // reference's validation sweep runs 12 objectives x 5 steps (+ the multi-object chains) on the same objects - and it would
    // cost 2 % when every pair brings its own object (bench.py).  So it is built when the call count says the objects are being
    // reused: when the calls since set_objects pass XTAB_AFTER.
    bool tab = xtab_enabled && !force_slow_xobj && xobj_mode == 0;
    if (tab) {
        const bool crosses = grads_since_set <= XTAB_AFTER && grads_since_set + n_calls > XTAB_AFTER;
        grads_since_set += n_calls;
        if (crosses)
            for (int i = 0; i < n_objects; ++i) {
                ObjectTables &t = *tables[i];
                if (!(t.has16 ? t.has_x16 : t.has_x) && (rc = build_xtab(i, s))) return rc;
            }
    }
    for (int i = 0; i < n_chains && tab; ++i) {
        DGDM_REQUIRE(oidx[i] >= 0 && oidx[i] < n_objects, DGDM_EINVAL, "chain %d refers to object %d of %d", i, oidx[i], n_objects);
        tab = bf16 ? tables[oidx[i]]->has_x16 : tables[oidx[i]]->has_x;
    }
    const int64_t rt = R * n_calls;
    host_stamp("embed: enter");
    if ((rc = upload_starts(starts_host, n_chains, R, s, !tab, n_calls, call_stride))) return rc;     // host work: overlaps a table build still in flight
    host_stamp("embed: starts converted, sorted, copy queued");
    if ((rc = finish_objects())) return rc;
    host_stamp("embed: tables finished");
    TrunkParams scratch{};
    if (tab && (rc = use_xtab(oidx, n_chains, rt, bf16, &scratch, &tab, s))) return rc;
    e->used16 = false;
    if (!tab && (rc = run_xobj(oidx, n_chains, rt, bf16, &e->used16, s))) return rc;
    e->tab = tab; e->rows_per_chain = rt;
    DGDM_HIP_CHECK(hipEventRecord(up_consumed, s));          // the index buffers may be overwritten once these kernels are through
    consumed_recorded = true;
    prof_end(s, DGDM_STAGE_XOBJ, 0.0);
    return DGDM_OK;
}

// cond_fn for n_chains chains.  3-D: `emb` = the embeddings made by DgdmGuidance::embed for a run of calls, `call` = which of them this is;
// emb == nullptr: this call's own (starts_host = its draws).
static int guidance_grad(DgdmGuidance *g, int kind, const float *x_dev, int timestep, const DgdmObjective *objectives, const float *rowcoef_dev,
                         const int64_t *starts_host, int n_chains, float *grad_dev, hipStream_t s, const DgdmGuidance::Embedded *emb = nullptr,
                         int call = 0) {
    DGDM_REQUIRE(g && x_dev && objectives && grad_dev, DGDM_EINVAL, "guidance_grad: null argument");
    if (g->m->kind != kind) { set_error("model type not supported: %d-D entry point on a %d-D model", kind, g->m->kind); return DGDM_EMODE; }
    DGDM_REQUIRE(n_chains > 0 && n_chains <= g->cfg.max_chains, DGDM_EINVAL, "n_chains %d outside 1..%d", n_chains, g->cfg.max_chains);
    DGDM_REQUIRE(g->n_objects > 0, DGDM_EINVAL, "dgdm_guidance_set_objects has not been called");
    std::vector<int> oidx(n_chains);
    std::vector<TrunkObjective> tob(n_chains);
    for (int i = 0; i < n_chains; ++i) {
        oidx[i] = objectives[i].object;
        for (int j = 0; j < 3; ++j) { tob[i].lin[j] = objectives[i].lin[j]; tob[i].quad[j] = objectives[i].quad[j]; }
        tob[i].use_rowcoef = objectives[i].use_rowcoef; tob[i].pad = 0;
        DGDM_REQUIRE(!tob[i].use_rowcoef || rowcoef_dev, DGDM_EINVAL, "chain %d uses rowcoef but rowcoef_dev is null", i);
    }
    const float t_scaled = (float)timestep / (float)g->cfg.num_train_timesteps;      // timesteps.float() / T  (diffusion.py:487,496)
    int rc;
    prof_begin(s, DGDM_STAGE_GUIDE_MISC);
    if ((rc = g->common_pre(x_dev, t_scaled, oidx.data(), n_chains, s))) return rc;
    DGDM_HIP_CHECK(hipMemcpyAsync(g->objdev.p, tob.data(), sizeof(TrunkObjective) * n_chains, hipMemcpyHostToDevice, s));
    prof_end(s, DGDM_STAGE_GUIDE_MISC, 0.0);
    TrunkParams p;
    g->m->fill_trunk(&p);
    if (kind == 3) {
        DgdmGuidance::Embedded own;
        if (!emb) {
            if ((rc = g->embed(oidx.data(), n_chains, starts_host, 1, 0, &own, s))) return rc;
            emb = &own;
            call = 0;
        }
        const size_t row0 = (size_t)call * g->R;
        p.xstride = emb->rows_per_chain;
        if (emb->tab) {
            p.xidx = g->xidx.as<int>() + row0;
            if (g->bf16) p.xtab16 = g->xtabptrs.as<const uint32_t *>();
            else p.xtab = g->xtabptrs.as<const float *>();
        } else {
            p.xobj = g->xobj.as<float>() + row0 * 256;
            p.xobj16 = emb->used16 ? g->xobj16.as<uint32_t>() + row0 * 128 : nullptr;
        }
    }
    p.Atab = g->atab.as<float>(); p.Ptab = g->ptab.as<float>(); p.PtabT = g->ptab_t.as<float>(); p.Pmax = g->pmax.as<float>(); p.obj = g->objdev.as<TrunkObjective>(); p.rowcoef = rowcoef_dev;
    p.partial = g->partial.as<float>();
    p.B = g->B; p.C = g->C; p.tiles_per_b = g->tiles_per_b; p.ntiles = n_chains * g->B * g->tiles_per_b; p.R = g->R;
    if (kind != 3) p.xstride = g->R;
    if (g->bf16) {
        g->m->fill_trunk_bf16(&p);       // only the two weight streams differ
#ifdef DGDM_TRUNK_CLOCKS
        static long long *dclk = nullptr;
        if (!dclk) (void)hipMalloc(&dclk, (size_t)1 << 26);
        p.clk = dclk;
#endif
        if ((rc = trunk_bf16_launch(kind, p, s))) return rc;
    } else if (g->f32_mfma) {
        if ((rc = trunk_launch(kind, false, false, p, s))) return rc;
    } else {
        TrunkF16Scales sc;
        g->m->fill_trunk_f16(&p, &sc);   // only the two weight streams differ (+ their scale exponents)
        if ((rc = trunk_f16l_launch(kind, p, sc, s))) return rc;
    }
#ifdef DGDM_TRUNK_CLOCKS
    if (g->bf16) {      // mean cycles per phase over all waves (experiment build)
        const size_t nw = (size_t)(p.ntiles + 1) / 2;
        std::vector<long long> h(nw * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), p.clk, nw * 64, hipMemcpyDeviceToHost);
        double ph[6] = {0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < nw; ++w)
            for (int i = 0; i < 6; ++i) ph[i] += (double)(h[w * 8 + i + 1] - h[w * 8 + i]);
        fprintf(stderr, "PHASES(mean cycles, %zu waves) first %.0f fwdmid %.0f wout+obj %.0f woutT %.0f bwdmid %.0f final %.0f\n", nw, ph[0] / nw, ph[1] / nw, ph[2] / nw,
                ph[3] / nw, ph[4] / nw, ph[5] / nw);
    }
#endif
    prof_begin(s, DGDM_STAGE_GUIDE_MISC);
    rc = dyn_post64(g->m->W1, g->partial.as<float>(), g->tiles_per_b, g->m->blob64.at(g->m->off64.w1c_w), g->m->blob64.at(g->m->off64.g2_w),
                    g->m->blob64.at(g->m->off64.g0_w), g->V.as<double>(), grad_dev, n_chains * g->B, g->m->L, s);
    prof_end(s, DGDM_STAGE_GUIDE_MISC, 0.0);
    return rc;
}

extern "C" int dgdm_dyn2d_guidance_grad(DgdmGuidance *g, const float *x_dev, int timestep, const DgdmObjective *objectives,
                                        const float *rowcoef_dev, int n_chains, float *grad_dev, void *stream) {
    return guidance_grad(g, 2, x_dev, timestep, objectives, rowcoef_dev, nullptr, n_chains, grad_dev, (hipStream_t)stream);
}

extern "C" int dgdm_dyn3d_guidance_grad(DgdmGuidance *g, const float *x_dev, int timestep, const DgdmObjective *objectives,
                                        const float *rowcoef_dev, const int64_t *starts_host, int n_chains, float *grad_dev, void *stream) {
    return guidance_grad(g, 3, x_dev, timestep, objectives, rowcoef_dev, starts_host, n_chains, grad_dev, (hipStream_t)stream);
}

// ================================================================================================ the denoise loop as one call
namespace {
__global__ void fill_i32_kernel(int *p, int v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void repeat_rows_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t n_src, int reps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_src * reps) dst[i] = src[i % n_src];
}
}  // namespace

extern "C" int dgdm_guided_chains_run(DgdmUnet1d *unet, DgdmGuidance *g, const float *noise_dev, int n_chains, int n_grad,
                                      const DgdmObjective *objectives, const float *rowcoef_dev, const int64_t *starts_host,
                                      const int32_t *timesteps, const float *coef, const float *scales, int n_steps, float *x_out_dev,
                                      void *stream) {
    DGDM_REQUIRE(unet && g && noise_dev && objectives && timesteps && coef && scales && x_out_dev, DGDM_EINVAL, "dgdm_guided_chains_run: null argument");
    DGDM_REQUIRE(n_chains > 0 && n_grad > 0 && n_chains * n_grad <= g->cfg.max_chains && n_steps > 0, DGDM_EINVAL,
                 "dgdm_guided_chains_run: %d chains x %d gradients outside 1..%d", n_chains, n_grad, g->cfg.max_chains);
    hipStream_t s = (hipStream_t)stream;
    const int B = g->B, L = g->m->L, kind = g->m->kind;
    const size_t per_chain = (size_t)B * L, nx = per_chain * n_chains, ng = nx * n_grad;
    DGDM_REQUIRE(kind == 2 || starts_host, DGDM_EINVAL, "3-D guidance needs the FPS start indices");
    int rc;
    if ((rc = g->loopx[0].alloc(nx * 4)) || (rc = g->loopx[1].alloc(nx * 4)) || (rc = g->loopeps.alloc(nx * 4)) || (rc = g->loopgrad.alloc(ng * 4)) ||
        (rc = g->loopxrep.alloc(ng * 4)) || (rc = g->loopts.alloc((size_t)n_chains * B * sizeof(int))))
        return rc;
    // every chain starts from the same noise (generator/diffusion.py:570)
    hipLaunchKernelGGL(repeat_rows_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, s, noise_dev, g->loopx[0].as<float>(), (int64_t)per_chain, n_chains);
    bool same_scale = true;
    for (int c = 1; c < n_chains; ++c) same_scale = same_scale && scales[c] == scales[0];
    const int64_t spc = kind == 3 ? 2 * g->R : 0;
    // 3-D: the embeddings of the rows depend on the draws and the objects, not on x, so those of ALL the steps are made before the
    // first one, in one upload and one gather launch: the (chain, s1) groups then hold n_steps times the rows per slab staged (and
    // per slab read from HBM), and more of them are equal rows computed once.  The host's conversion and sort of the draws runs
    // while the objects' tables are still being built (embed() waits for them only after the host work).  (A gather launch holds
    // fewer than 2^22 rows per chain - the sorted row list packs s2 beside the row id - so a longer run is embedded in groups of
    // `cpe` calls, each before its first step.)
    DgdmGuidance::Embedded emb;
    std::vector<int> oidx(n_chains * n_grad);
    const int64_t call_stride = (int64_t)n_chains * n_grad * spc;
    int cpe = kind == 3 ? (int)std::min<int64_t>(n_steps, std::max<int64_t>(1, (((int64_t)1 << 22) - 1) / std::max<int64_t>(1, g->R))) : n_steps;
    if (const char *e = getenv("DGDM_EMBED_CALLS")) cpe = std::max(1, std::min(cpe, atoi(e)));      // test hook: calls per gather launch
    if (kind == 3)
        for (int i = 0; i < n_chains * n_grad; ++i) oidx[i] = objectives[i].object;
    for (int si = 0; si < n_steps; ++si) {
        float *x = g->loopx[si & 1].as<float>(), *xn = (si + 1 == n_steps) ? x_out_dev : g->loopx[(si + 1) & 1].as<float>();
        const int t = timesteps[si];
        if (kind == 3 && si % cpe == 0 &&
            (rc = g->embed(oidx.data(), n_chains * n_grad, starts_host + (size_t)si * call_stride, std::min(cpe, n_steps - si), call_stride, &emb, s))) return rc;
        // eps-net: at the first step all chains hold the same B fingers - one evaluation, replicated (same input, same bits)
        const int nb = (si == 0 && n_chains > 1) ? B : n_chains * B;
        hipLaunchKernelGGL(fill_i32_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, g->loopts.as<int>(), t, nb);
        if ((rc = dgdm_unet1d_forward(unet, x, g->loopts.as<int>(), g->loopeps.as<float>(), nb, L, s))) return rc;
        if (nb != n_chains * B) {
            // replicate in place from the back so that the source block [0, per_chain) is read before it could be overwritten: separate buffer instead
            hipLaunchKernelGGL(repeat_rows_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, s, g->loopeps.as<float>(), g->loopgrad.as<float>(), (int64_t)per_chain, n_chains);
            DGDM_HIP_CHECK(hipMemcpyAsync(g->loopeps.p, g->loopgrad.p, nx * 4, hipMemcpyDeviceToDevice, s));
        }
        // cond_fn for the n_grad * n_chains gradient chains (object-major: gradient j of chain k is chain j * n_chains + k), x repeated
        const float *xg = x;
        if (n_grad > 1) {
            hipLaunchKernelGGL(repeat_rows_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, x, g->loopxrep.as<float>(), (int64_t)nx, n_grad);
            xg = g->loopxrep.as<float>();
        }
        if ((rc = guidance_grad(g, kind, xg, t, objectives, rowcoef_dev, nullptr, n_chains * n_grad, g->loopgrad.as<float>(), s,
                                kind == 3 ? &emb : nullptr, si % cpe))) return rc;
        const float *cf = coef + 4 * si;
        if (same_scale) {
            if ((rc = dgdm_ddim_guided_step(x, g->loopeps.as<float>(), g->loopgrad.as<float>(), n_grad, xn, (int64_t)nx, cf[0], cf[1], cf[2], cf[3], scales[0], s))) return rc;
        } else {
            DGDM_REQUIRE(n_grad == 1, DGDM_EINVAL, "per-chain guidance scales with averaged gradients are not supported");
            for (int c = 0; c < n_chains; ++c)
                if ((rc = dgdm_ddim_guided_step(x + c * per_chain, g->loopeps.as<float>() + c * per_chain, g->loopgrad.as<float>() + c * per_chain, 1,
                                                xn + c * per_chain, (int64_t)per_chain, cf[0], cf[1], cf[2], cf[3], scales[c], s))) return rc;
        }
    }
    DGDM_HIP_CHECK(hipGetLastError());
    host_stamp("chains_run: all steps queued");
    return DGDM_OK;
}

extern "C" int dgdm_guidance_orientation_sweep(DgdmGuidance *g, const float *x_dev, const int32_t *object_of_chain, const int64_t *starts_host,
                                               int n_chains, float *logits_dev, void *stream) {
    DGDM_REQUIRE(g && x_dev && object_of_chain && logits_dev, DGDM_EINVAL, "dgdm_guidance_orientation_sweep: null argument");
    DGDM_REQUIRE(n_chains > 0 && n_chains <= g->cfg.max_chains, DGDM_EINVAL, "n_chains %d outside 1..%d", n_chains, g->cfg.max_chains);
    DGDM_REQUIRE(g->n_objects > 0, DGDM_EINVAL, "dgdm_guidance_set_objects has not been called");
    hipStream_t s = (hipStream_t)stream;
    const int kind = g->m->kind;
    int rc;
    if ((rc = g->common_pre(x_dev, 0.f, object_of_chain, n_chains, s))) return rc;       // timesteps = zeros (:515,521)
    TrunkParams p;
    g->m->fill_trunk(&p);
    if (kind == 3) {
        DGDM_REQUIRE(starts_host, DGDM_EINVAL, "3-D sweep needs the FPS start indices");
        // the sweep stays float32: table rows when the float32 tables exist, the gather kernels otherwise (e.g. a bf16-mode handle)
        bool tab = g->xtab_enabled && !g->force_slow_xobj && g->xobj_mode == 0;
        for (int i = 0; i < n_chains && tab; ++i) tab = object_of_chain[i] >= 0 && object_of_chain[i] < g->n_objects && g->tables[object_of_chain[i]]->has_x;
        if ((rc = g->upload_starts(starts_host, n_chains, g->Rs, s, !tab))) return rc;
        if ((rc = g->finish_objects())) return rc;
        if (tab && (rc = g->use_xtab(object_of_chain, n_chains, g->Rs, false, &p, &tab, s))) return rc;
        if (!tab) {
            if ((rc = g->run_xobj(object_of_chain, n_chains, g->Rs, false, nullptr, s))) return rc;
            p.xobj = g->xobj.as<float>();
        }
        DGDM_HIP_CHECK(hipEventRecord(g->up_consumed, s));
        g->consumed_recorded = true;
    }
    p.Atab = g->atab.as<float>(); p.Ptab = g->ptab_sweep.as<float>(); p.PtabT = g->ptab_sweep_t.as<float>(); p.logits = logits_dev;
    p.B = g->B; p.C = g->G; p.tiles_per_b = g->sweep_tiles_per_b; p.ntiles = n_chains * g->B * g->sweep_tiles_per_b; p.R = g->Rs; p.xstride = g->Rs;
    return trunk_launch(kind, false, true, p, s);
}

// ================================================================================================ PointNet++ on arbitrary rows
namespace {

struct CloudGroup { std::vector<int> rows; };

// Runs the table pipeline for every distinct cloud among `rows` clouds and writes emb_dev [rows][256].
int pointnet_rows(DgdmDynamics *m, const float *xyz_dev /*[rows][3][N]*/, const int64_t *s1, const int64_t *s2, float *emb_dev, int rows,
                  int N, hipStream_t s) {
    DGDM_REQUIRE(N > 0 && N <= 1024, DGDM_EINVAL, "clouds of %d points unsupported (1..1024)", N);
    std::vector<float> host((size_t)rows * 3 * N);
    DGDM_HIP_CHECK(hipMemcpyAsync(host.data(), xyz_dev, host.size() * 4, hipMemcpyDeviceToHost, s));
    DGDM_HIP_CHECK(hipStreamSynchronize(s));
    // group bit-identical clouds (cond_fn hands 512 replicas of one cloud per call, diffusion.py:491)
    std::unordered_map<std::string, int> seen;
    std::vector<CloudGroup> groups;
    std::vector<int> rep;
    for (int r = 0; r < rows; ++r) {
        std::string key(reinterpret_cast<const char *>(host.data() + (size_t)r * 3 * N), (size_t)3 * N * 4);
        auto it = seen.find(key);
        int gid;
        if (it == seen.end()) {
            gid = (int)groups.size();
            seen.emplace(std::move(key), gid);
            groups.emplace_back();
            rep.push_back(r);
        } else {
            gid = it->second;
        }
        groups[gid].rows.push_back(r);
    }
    const PnWeights w = m->pn();
    DevBuf xyz, fps1, F1, U, Y, L2, Z, vlist, slotmap, starts, chains, out, crowded, clist, poff, pairs, prank;
    int rc;
    const PnWeights64 w64 = m->pn64();
    if ((rc = xyz.alloc((size_t)N * 12)) || (rc = fps1.alloc((size_t)N * 512 * 4)) || (rc = F1.alloc((size_t)N * 128 * 8)) || (rc = U.alloc((size_t)N * 128 * 8)) ||
        (rc = Y.alloc((size_t)N * N * 1024)) || (rc = chains.alloc(sizeof(XobjChain))) || (rc = crowded.alloc((size_t)N * 4)) ||
        (rc = clist.alloc((size_t)(N + 1) * 4)) || (rc = poff.alloc((size_t)(N + 1) * 4)) || (rc = pairs.alloc((size_t)N * N * 4)) ||
        (rc = prank.alloc((size_t)N * N * 2)))
        return rc;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const std::vector<int> &rws = groups[gi].rows;
        std::vector<float> pts((size_t)N * 3);
        const float *src = host.data() + (size_t)rep[gi] * 3 * N;
        for (int i = 0; i < N; ++i) { pts[3 * i] = src[i]; pts[3 * i + 1] = src[N + i]; pts[3 * i + 2] = src[2 * N + i]; }
        std::vector<int> slot_of(N, -1), vl, st(2 * rws.size());
        for (size_t k = 0; k < rws.size(); ++k) {
            const int64_t a = s1[rws[k]], b = s2[rws[k]];
            DGDM_REQUIRE(a >= 0 && a < N && b >= 0 && b < 512, DGDM_EINVAL, "FPS start out of range");
            if (slot_of[a] < 0) { slot_of[a] = (int)vl.size(); vl.push_back((int)a); }
            st[2 * k] = (int)a; st[2 * k + 1] = (int)b;
        }
        const int nv = (int)vl.size();
        if ((rc = xyz.upload(pts.data(), pts.size() * 4)) || (rc = vlist.upload(vl.data(), vl.size() * 4)) ||
            (rc = slotmap.upload(slot_of.data(), slot_of.size() * 4)) || (rc = starts.upload(st.data(), st.size() * 4)) ||
            (rc = L2.alloc((size_t)nv * N * 1024)) || (rc = Z.alloc((size_t)nv * N * 1024)) || (rc = out.alloc(rws.size() * 1024)))
            return rc;
        const float *x = xyz.as<float>();
        if ((rc = pn_fps_table(x, N, N, 512, fps1.as<int>(), nullptr, s))) return rc;
        if ((rc = pn_sa1_64(x, N, w.r1sq, w64, F1.as<double>(), s))) return rc;
        if ((rc = linear64(nullptr, F1.as<double>(), 128, w64.sa2_wf_t, w64.sa2_b0, nullptr, 1, U.as<double>(), nullptr, 128, N, 128, 128, ACT_NONE, s))) return rc;
        if ((rc = pn_crowd(x, N, w, crowded.as<int>(), clist.as<int>(), clist.as<int>() + N, poff.as<int>(), pairs.as<int>(), prank.as<short>(), s))) return rc;
        if ((rc = pn_pairs64(x, N, U.as<double>(), w64, pairs.as<int>(), poff.as<int>(), Y.as<float>(), s))) return rc;
        if ((rc = pn_l2(x, N, w, fps1.as<int>(), vlist.as<int>(), nv, Y.as<float>(), L2.as<float>(), clist.as<int>(), clist.as<int>() + N,
                        poff.as<int>(), prank.as<short>(), false, s))) return rc;
        if ((rc = pn_z64(x, N, nv, w64, L2.as<float>(), Z.as<float>(), clist.as<int>(), clist.as<int>() + N, s))) return rc;
        XobjChain ch{};
        ch.xyz = x; ch.fps1 = fps1.as<int>(); ch.slot_of_start = slotmap.as<int>(); ch.Z = Z.as<float>(); ch.fps2 = nullptr; ch.flags = nullptr; ch.crowded = crowded.as<int>(); ch.N = N;
        DGDM_HIP_CHECK(hipMemcpyAsync(chains.p, &ch, sizeof ch, hipMemcpyHostToDevice, s));
        XobjParams xp{};
        xp.chains = chains.as<XobjChain>(); xp.starts = starts.as<int>(); xp.xobj = out.as<float>(); xp.R = (int64_t)rws.size(); xp.total_rows = xp.R;
        if ((rc = pn_xobj(xp, false, s))) return rc;
        // scatter the group's rows back
        bool contiguous = true;
        for (size_t k = 1; k < rws.size(); ++k) contiguous = contiguous && rws[k] == rws[k - 1] + 1;
        if (contiguous) {
            DGDM_HIP_CHECK(hipMemcpyAsync(emb_dev + (size_t)rws[0] * 256, out.p, rws.size() * 1024, hipMemcpyDeviceToDevice, s));
        } else {
            for (size_t k = 0; k < rws.size(); ++k)
                DGDM_HIP_CHECK(hipMemcpyAsync(emb_dev + (size_t)rws[k] * 256, out.as<float>() + k * 256, 1024, hipMemcpyDeviceToDevice, s));
        }
        DGDM_HIP_CHECK(hipStreamSynchronize(s));     // per-group buffers are reused by the next group
    }
    return DGDM_OK;
}

}  // namespace

extern "C" int dgdm_pointnet2_forward(DgdmDynamics *m, const float *xyz_dev, const int64_t *start_sa1_host, const int64_t *start_sa2_host,
                                      float *emb_dev, int rows, int N, void *stream) {
    DGDM_REQUIRE(m && xyz_dev && start_sa1_host && start_sa2_host && emb_dev && rows >= 0, DGDM_EINVAL, "dgdm_pointnet2_forward: bad argument");
    if (m->kind != 3) { set_error("model type not supported: PointNet++ belongs to the 3-D model"); return DGDM_EMODE; }
    if (rows == 0) return DGDM_OK;
    return pointnet_rows(m, xyz_dev, start_sa1_host, start_sa2_host, emb_dev, rows, N, (hipStream_t)stream);
}

extern "C" int dgdm_dyn3d_forward(DgdmDynamics *m, const float *x_ctrl, const float *x_ori, const float *x_pos, const float *t,
                                  const float *xyz, const int64_t *start_sa1_host, const int64_t *start_sa2_host, float *logits, int rows,
                                  int N, void *stream) {
    DGDM_REQUIRE(m && x_ctrl && x_ori && x_pos && t && xyz && start_sa1_host && start_sa2_host && logits && rows >= 0, DGDM_EINVAL,
                 "dgdm_dyn3d_forward: bad argument");
    if (m->kind != 3) { set_error("model type not supported: dgdm_dyn3d_forward on a 2-D model"); return DGDM_EMODE; }
    if (rows == 0) return DGDM_OK;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // workspace: V, GENC [rows][256] | pose [rows][32] | tmp [rows][768] | z1 [rows][512] | xobj [rows][256]
    const size_t need = (size_t)rows * (256 + 256 + 32 + 768 + 512 + 256) * sizeof(float);
    if ((rc = m->ws.alloc(need))) return rc;
    float *V = m->ws.as<float>(), *genc = V + (size_t)rows * 256, *pose = genc + (size_t)rows * 256, *tmp = pose + (size_t)rows * 32,
          *z1 = tmp + (size_t)rows * 768, *xo = z1 + (size_t)rows * 512;
    if ((rc = pointnet_rows(m, xyz, start_sa1_host, start_sa2_host, xo, rows, N, s))) return rc;
    if ((rc = m->time_part(t, 0.f, tmp, z1, rows, s))) return rc;
    // x_ctrl[:, 1, :]  (profile_forward_3d.py:78): rows of length L at stride 3L, offset L
    if ((rc = m->gripper_forward(x_ctrl + m->L, 3 * m->L, V, genc, rows, s))) return rc;
    if ((rc = linear(genc, 256, m->blob.at(m->off.w1c_wt), nullptr, nullptr, 1, z1, 512, rows, 256, 512, ACT_NONE, true, s))) return rc;
    if ((rc = pose_embed(x_ori, x_pos, pose, rows, s))) return rc;
    if ((rc = linear(pose, 27, m->blob.at(m->off.w1p_wt), nullptr, nullptr, 1, z1, 512, rows, 27, 512, ACT_NONE, true, s))) return rc;
    TrunkParams p;
    m->fill_trunk(&p);
    p.Atab = z1; p.xobj = xo; p.logits = logits; p.C = rows; p.R = rows; p.xstride = rows; p.B = 1; p.tiles_per_b = 1; p.ntiles = (rows + 31) / 32;
    return trunk_launch(3, true, true, p, s);
}
