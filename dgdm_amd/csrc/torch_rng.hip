// The torch CPU generator, replayed outside torch (host code only).
//
// farthest_point_sample draws its start index with torch.randint(0, N, (rows,)) on the CPU default generator
// (dynamics/models/pointnet2_utils.py:83), twice per classifier call and sub-batch - 72 000 draws per cond_fn call at the shipped 3-D
// grid, 11.5 M per batch of 32 chains.  The draws do not depend on anything the GPU computes, so they can be produced ahead of the
// launches by a worker thread - but torch.randint holds the GIL-free kernel to one serial stream at ~3.5 ns per draw and cannot skip.
// This file restates the generator (at::mt19937 = MT19937 with the reference seeding; CPUGeneratorImpl::random() = one 32-bit output;
// random_from_to = `value % range + base`, ATen/core/MT19937RNGEngine.h, ATen/native/cpu/DistributionTemplates.h) on the generator's
// own state blob (torch.get_rng_state() / Generator.get_state(): CPUGeneratorImplState, 5056 bytes), so that
//   * a stream can be advanced from any thread without the interpreter (ctypes releases the GIL), ~3x faster than torch.randint
//     (the 624-word state refresh and the tempering are written as plain loops the compiler vectorises),
//   * draws can be SKIPPED without being materialised (a rank replaying the global stream past other ranks' chains: state refresh only),
//   * the blob can be handed back to torch (set_rng_state): the generator continues exactly as if torch.randint had made the draws.
// Pinned bit for bit against torch.randint / manual_seed in tests/test_host_logic.py.
#include "common.h"
#include <cstring>

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UMASK = 0x80000000u, LMASK = 0x7fffffffu;
constexpr size_t BLOB_BYTES = 5056;      // sizeof(at::CPUGeneratorImplState)
// CPUGeneratorImplStateLegacy: uint64 the_initial_seed; int left; int seeded; uint64 next; uint64 state[624]; double normal_x, normal_y,
// normal_rho; int normal_is_valid;   then float next_float_normal_sample; bool is_next_float_normal_sample_valid
constexpr size_t OFF_SEED = 0, OFF_LEFT = 8, OFF_SEEDED = 12, OFF_NEXT = 16, OFF_STATE = 24, OFF_TAIL = 24 + 8 * MT_N;

struct Engine {
    uint32_t st[MT_N];
    int left;
    uint32_t next;
    void load(const uint8_t *b) {
        int32_t l;
        uint64_t n;
        memcpy(&l, b + OFF_LEFT, 4);
        memcpy(&n, b + OFF_NEXT, 8);
        left = l; next = (uint32_t)n;
        const uint64_t *s = reinterpret_cast<const uint64_t *>(b + OFF_STATE);
        for (int i = 0; i < MT_N; ++i) st[i] = (uint32_t)s[i];
    }
    void store(uint8_t *b) const {
        const int32_t l = left;
        const uint64_t n = next;
        memcpy(b + OFF_LEFT, &l, 4);
        memcpy(b + OFF_NEXT, &n, 8);
        uint64_t *s = reinterpret_cast<uint64_t *>(b + OFF_STATE);
        for (int i = 0; i < MT_N; ++i) s[i] = st[i];
    }
    static inline uint32_t twist(uint32_t u, uint32_t v) { return (((u & UMASK) | (v & LMASK)) >> 1) ^ ((v & 1u) ? MATRIX_A : 0u); }
    void refresh() {                       // mt19937::next_state
        for (int i = 0; i < MT_N - MT_M; ++i) st[i] = st[i + MT_M] ^ twist(st[i], st[i + 1]);
        for (int i = MT_N - MT_M; i < MT_N - 1; ++i) st[i] = st[i + MT_M - MT_N] ^ twist(st[i], st[i + 1]);
        st[MT_N - 1] = st[MT_M - 1] ^ twist(st[MT_N - 1], st[0]);
    }
    static inline uint32_t temper(uint32_t y) {
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    // n draws of `value % range` (range >= 1) into out (or nowhere).  mt19937::operator(): if (--left == 0) refresh; y = st[next++].
    template <class T>
    void draw(int64_t n, uint32_t range, T *out) {
        const bool pow2 = (range & (range - 1)) == 0;
        const uint32_t mask = range - 1;
        while (n > 0) {
            if (left <= 1) { refresh(); left = MT_N + 1; next = 0; }      // transient 625: the draw that triggers the refresh does not count down
            const int m = (int)(n < (int64_t)(left - 1) ? n : (int64_t)(left - 1));
            if (out) {
                const uint32_t *s = st + next;
                if (pow2) for (int i = 0; i < m; ++i) out[i] = (T)(temper(s[i]) & mask);
                else for (int i = 0; i < m; ++i) out[i] = (T)(temper(s[i]) % range);
                out += m;
            }
            next += (uint32_t)m; left -= m; n -= m;
        }
    }
};

bool blob_ok(const uint8_t *b, int64_t bytes) {
    if (!b || bytes != (int64_t)BLOB_BYTES) { dgdm::set_error("torch generator state blob: expected %zu bytes (torch.get_rng_state()), got %lld", BLOB_BYTES, (long long)bytes); return false; }
    int32_t l;
    uint64_t n;
    memcpy(&l, b + OFF_LEFT, 4);
    memcpy(&n, b + OFF_NEXT, 8);
    if (l < 1 || l > MT_N || n > (uint64_t)MT_N) { dgdm::set_error("torch generator state blob is not an mt19937 state (left %d, next %llu)", l, (unsigned long long)n); return false; }
    return true;
}

}  // namespace

// torch.Generator().manual_seed(seed) -> state blob  (CPUGeneratorImpl::set_current_seed: engine_ = mt19937(seed), cached normals dropped)
extern "C" int dgdm_torch_rng_seed(uint8_t *state, int64_t state_bytes, uint64_t seed) {
    DGDM_REQUIRE(state && state_bytes == (int64_t)BLOB_BYTES, DGDM_EINVAL, "dgdm_torch_rng_seed: state blob must be %zu bytes", BLOB_BYTES);
    memset(state, 0, BLOB_BYTES);
    memcpy(state + OFF_SEED, &seed, 8);
    const int32_t seeded = 1;
    memcpy(state + OFF_SEEDED, &seeded, 4);
    Engine e;
    e.st[0] = (uint32_t)(seed & 0xffffffffu);
    for (int j = 1; j < MT_N; ++j) e.st[j] = 1812433253u * (e.st[j - 1] ^ (e.st[j - 1] >> 30)) + (uint32_t)j;
    e.left = 1; e.next = 0;
    e.store(state);
    (void)OFF_TAIL;
    return DGDM_OK;
}

// n x torch.randint(0, high, ...) draws (high < 2^32) on the blob: into out (int64, as torch hands them over) or skipped (out == NULL)
extern "C" int dgdm_torch_rng_randint(uint8_t *state, int64_t state_bytes, uint32_t high, int64_t n, int64_t *out) {
    if (!blob_ok(state, state_bytes)) return DGDM_EINVAL;
    DGDM_REQUIRE(high >= 1 && n >= 0, DGDM_EINVAL, "dgdm_torch_rng_randint: bad range / count");
    Engine e;
    e.load(state);
    e.draw<int64_t>(n, high, out);
    e.store(state);
    return DGDM_OK;
}

// The draws of n_calls consecutive classifier calls over `rows` rows each (generator/diffusion.py:495-498 -> pointnet2_utils.py:83): per
// call and sub-batch of n <= sub_batch_size rows, sa1's n draws in [0, num_points) then sa2's n draws in [0, 512), laid out per call as
// [sub-batch 0: sa1 x n0, sa2 x n0 | sub-batch 1: ...] - the int64 layout dgdm_dyn3d_guidance_grad takes.  out == NULL: skipped.
// out_call_stride: distance (in elements) between the outputs of consecutive calls, 0 = 2 * rows (contiguous) - so that the calls of one
// chain can be written straight into a [step][chain][2 * rows] array.
extern "C" int dgdm_torch_rng_fps_starts(uint8_t *state, int64_t state_bytes, int num_points, int64_t sub_batch_size, int64_t rows, int64_t n_calls,
                                         int64_t *out, int64_t out_call_stride) {
    if (!blob_ok(state, state_bytes)) return DGDM_EINVAL;
    DGDM_REQUIRE(num_points >= 1 && sub_batch_size >= 1 && rows >= 0 && n_calls >= 0, DGDM_EINVAL, "dgdm_torch_rng_fps_starts: bad argument");
    DGDM_REQUIRE(out_call_stride == 0 || out_call_stride >= 2 * rows, DGDM_EINVAL, "dgdm_torch_rng_fps_starts: call stride smaller than a call");
    const int64_t stride = out_call_stride ? out_call_stride : 2 * rows;
    Engine e;
    e.load(state);
    if (num_points == 512 && (!out || stride == 2 * rows)) {      // one range throughout: the calls are one flat run of draws
        e.draw<int64_t>(2 * rows * n_calls, 512u, out);
    } else {
        for (int64_t c = 0; c < n_calls; ++c) {
            int64_t *o = out ? out + c * stride : nullptr;
            for (int64_t r0 = 0; r0 < rows; r0 += sub_batch_size) {
                const int64_t n = rows - r0 < sub_batch_size ? rows - r0 : sub_batch_size;
                e.draw<int64_t>(n, (uint32_t)num_points, o ? o + 2 * r0 : nullptr);
                e.draw<int64_t>(n, 512u, o ? o + 2 * r0 + n : nullptr);
            }
        }
    }
    e.store(state);
    return DGDM_OK;
}
