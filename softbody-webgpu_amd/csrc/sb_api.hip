// sb_api.hip -- the C ABI of include/softbody.h: host driver of the HIP substep.
//
// Plays the role src/engineWorker.ts plays in the reference (device + buffers + subtick loop +
// upload / read-back), minus the canvas/render/Worker plumbing.  No CPU fallback: every entry
// point needs a live HIP device and fails loudly otherwise.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <new>
#include <numeric>
#include <stdexcept>
#include <unordered_map>

#include "sb_engine.h"
#include "sb_edit.h"

extern "C" sb_status sb_halo_set_layout(sb_engine *e, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *);

static thread_local std::string g_create_error;
void sb_set_create_error(const char *msg) { g_create_error = msg ? msg : ""; } // for sb_partition.cpp

#define SB_FAIL(e, code, ...)                                   \
    do {                                                        \
        char _buf[512];                                         \
        snprintf(_buf, sizeof _buf, __VA_ARGS__);               \
        if (e) (e)->err = _buf; else g_create_error = _buf;     \
        return (code);                                          \
    } while (0)

#define SB_HIP(e, call)                                                                       \
    do {                                                                                      \
        hipError_t _r = (call);                                                               \
        if (_r != hipSuccess) {                                                               \
            (void)hipGetLastError(); /* reported here: must not resurface in a later launch check */ \
            SB_FAIL(e, _r == hipErrorOutOfMemory ? SB_ERR_OOM : SB_ERR_HIP, "%s failed: %s",  \
                    #call, hipGetErrorString(_r));                                            \
        }                                                                                     \
    } while (0)

// Device arrays of a scene come from a per-engine pool: an upload that replaces a scene (the reference uploads after every
// edit, engineWorker.ts:569-601) finds last scene's blocks there instead of paying ~30 hipFree + hipMalloc pairs, each a device
// synchronisation (28 ms of a 128 ms re-upload of 1 M particles).  A block serves a request of up to its own size and at least
// half of it; new blocks get an eighth of head-room so that a slightly larger scene still fits.
template <typename T>
static sb_status dev_alloc(sb_engine *e, T **p, size_t n)
{
    *p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    size_t best = SIZE_MAX;
    for (size_t k = 0; k < e->pool_free.size(); k++) {
        const size_t have = e->pool_free[k].second;
        if (have >= bytes && have / 2 <= bytes && (best == SIZE_MAX || have < e->pool_free[best].second)) best = k;
    }
    std::pair<void *, size_t> blk;
    if (best != SIZE_MAX) {
        blk = e->pool_free[best];
        e->pool_free.erase(e->pool_free.begin() + (ptrdiff_t)best);
    } else {
        const size_t want = bytes + (bytes >= (1u << 20) ? bytes / 8 : 0);
        void *q = nullptr;
        SB_HIP(e, hipMalloc(&q, want));
        blk = {q, want};
    }
    e->pool_used.push_back(blk);
    e->device_bytes += bytes;
    *p = (T *)blk.first;
    return SB_OK;
}
// blocks the new scene did not take: kept while they are small change next to the scene itself, else returned
static void pool_trim(sb_engine *e)
{
    size_t idle = 0;
    for (const auto &b : e->pool_free) idle += b.second;
    while (!e->pool_free.empty() && idle > e->device_bytes / 4) {
        size_t big = 0;
        for (size_t k = 1; k < e->pool_free.size(); k++)
            if (e->pool_free[k].second > e->pool_free[big].second) big = k;
        idle -= e->pool_free[big].second;
        (void)hipFree(e->pool_free[big].first);
        e->pool_free.erase(e->pool_free.begin() + (ptrdiff_t)big);
    }
}

#define SB_TRY(x)                          \
    do {                                   \
        sb_status _s = (x);                \
        if (_s != SB_OK) return _s;        \
    } while (0)

// Waiting for the engine's stream where the wait is SHORT (the end of a call's launches, the looks of the spatial hash, a run of
// blocked launches under the hash): poll for `spin_us` before falling back on the blocking wait.  hipStreamSynchronize parks the
// thread, and being woken costs 20-50 us on a good day on this stack and several hundred on a bad one (r04: the lattice on the
// floor, 32 looks and runs per 1000 substeps, read 28.5 us per substep on one box and 39-44 on another with every wait behind a
// run parked) -- 10 % of the driver's 20-substep protocol at 1 M particles, and as much as the kernels of a small scene's frame.
// Callers pass what the work in flight should take (default: a third of a millisecond; 0.2 s at most); SB_WAIT_SPIN_US overrides
// every site (0: always park).
static int64_t sb_spin_override()
{
    static const int64_t v = [] { const char *s = getenv("SB_WAIT_SPIN_US"); return s ? (int64_t)atoll(s) : (int64_t)-1; }();
    return v;
}
// (three phases: busy polling for the first eight milliseconds, then a look every ~50 us between short sleeps -- a long grid-mode call
// is found finished within a tenth of a millisecond without a core spinning for its 30 ms -- then the parking wait)
template <typename Query, typename Park>
static hipError_t sb_wait_phases(Query query, Park park, int64_t spin_us)
{
    if (sb_spin_override() >= 0) spin_us = sb_spin_override();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = query();
        if (q != hipErrorNotReady) return q;
        const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (waited > spin_us) break;
        if (waited > 8000) {
            std::this_thread::sleep_for(std::chrono::microseconds(40));
        } else {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    return park();
}
static hipError_t sb_stream_wait(hipStream_t stream, int64_t spin_us = 330)
{
    return sb_wait_phases([&] { return hipStreamQuery(stream); }, [&] { return hipStreamSynchronize(stream); }, spin_us);
}
static hipError_t sb_event_wait(hipEvent_t ev, int64_t spin_us = 330)
{
    return sb_wait_phases([&] { return hipEventQuery(ev); }, [&] { return hipEventSynchronize(ev); }, spin_us);
}

// SB_UPLOAD_TIMING=1: where sb_write_buffers spends its time (stderr), for tuning the host side of an upload
struct SbStageTimer {
    bool on;
    std::chrono::steady_clock::time_point t0;
    SbStageTimer() : on(getenv("SB_UPLOAD_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[sb upload] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// The plan of an upload lives in a few hundred megabytes of host arrays; handing them back to the kernel (munmap) took 40 ms of
// a 140 ms upload.  They are moved here and destroyed on a side thread, which the next upload or sb_destroy joins.
struct SbUploadTrash {
    std::vector<float> px, py;
    SbTiling tl;
    SbBlocking bl;
    SbHostBeams hb;
    std::vector<uint32_t> v[8];
};
static void reap_join(sb_engine *e)
{
    if (e->reaper.joinable()) e->reaper.join();
}

// The blocked plan of an SB_COLLIDE_GRID engine (e->hy) costs what a blocked plan costs -- 65 ms of host time per million
// particles -- and only a scene that is ever quiet uses it.  The upload therefore only STARTS it, on a side thread, from copies of
// the positions and the engine's own beam records (which stay put until the next upload that plans, and that one joins first).
struct SbHybridPending {
    std::thread th;
    SbBlocking bl;
    std::vector<float> px, py;
    uint32_t target = 0, depth = 0;
};
static void hybrid_pending_drop(sb_engine *e)
{
    if (!e->hy_pending) return;
    if (e->hy_pending->th.joinable()) e->hy_pending->th.join();
    delete e->hy_pending;
    e->hy_pending = nullptr;
}

static void free_scene(sb_engine *e)
{
    hybrid_pending_drop(e);
    for (void *p : e->allocs) (void)hipFree(p); // (what does not go through the pool: the fine-grained mailbox)
    e->allocs.clear();
    e->pool_free.insert(e->pool_free.end(), e->pool_used.begin(), e->pool_used.end());
    e->pool_used.clear();
    e->device_bytes = 0;
    e->loaded = false;
    e->n_ghost_p = e->n_send_p = e->n_ghost_b = e->n_send_b = e->n_ghost_b_copies = 0;
    e->d_ghost_p = e->d_send_p = e->d_send_b = e->d_send_p_off = e->d_send_b_off = e->d_ghost_p_off = e->d_ghost_b_off = nullptr;
    e->d_ghost_b = nullptr;
    for (void *p : e->mapped) (void)hipIpcCloseMemHandle(p);
    e->mapped.clear();
    e->mailbox = nullptr; // was in allocs
    e->bk = SbBlockedDev{};
    e->hy = SbBlockedDev{};
    e->n_peers = e->peer_seq = e->send_floats = e->recv_floats = 0;
    if (e->dev_err) *e->dev_err = 0;
}

static inline uint32_t beam_stride(const sb_engine *e)
{
    return e->opt.layout == SB_LAYOUT_V1 ? SB_BEAM_STRIDE_V1 : SB_BEAM_STRIDE_V2;
}
static inline uint32_t map_isz(const sb_engine *e) { return e->opt.layout == SB_LAYOUT_V1 ? 2 : 4; }
static inline uint32_t map_get(const sb_engine *e, const uint8_t *m, size_t id)
{
    if (e->opt.layout == SB_LAYOUT_V1) {
        uint16_t v;
        memcpy(&v, m + 2 * id, 2);
        return v;
    }
    uint32_t v;
    memcpy(&v, m + 4 * id, 4);
    return v;
}
static inline void map_set(const sb_engine *e, uint8_t *m, size_t id, uint32_t val)
{
    if (e->opt.layout == SB_LAYOUT_V1) {
        uint16_t v = (uint16_t)val;
        memcpy(m + 2 * id, &v, 2);
    } else {
        memcpy(m + 4 * id, &val, 4);
    }
}
static inline uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

// material dictionary: beams that share (length, spring, damp, yield, limit) share one table row; mode 2 = rows with
// the rest length, mode 1 = rows without it (arbitrary rest lengths travel per beam), 0 = more rows than `cap`
struct SbMatDict {
    uint32_t mode = 0;
    std::vector<float> table; // [rows][6] = length, spring, damp, yield, limit, 1/length
    sbt::uvec<uint32_t> of_slot;
};
static void build_material_dictionary(SbMatDict &d, const SbHostBeams &hb, uint32_t cap)
{
    const uint32_t B = (uint32_t)hb.size();
    struct Key { uint32_t w[5]; bool operator==(const Key &o) const { return memcmp(w, o.w, sizeof w) == 0; } };
    struct KeyHash { size_t operator()(const Key &k) const { size_t h = 1469598103934665603ull; for (uint32_t x : k.w) h = (h ^ x) * 1099511628211ull; return h; } };
    // a few host threads over contiguous slot ranges, each with its own small dictionary (provisional row numbers); the rows
    // are then numbered by the first slot that uses them, i.e. exactly as one thread walking the slots in order would
    struct Local {
        std::unordered_map<Key, uint32_t, KeyHash> dict;
        std::vector<Key> keys;
        std::vector<uint32_t> first, remap;
        uint32_t lo = 0, hi = 0;
        bool ok = true;
    };
    d.of_slot.resize(B);
    d.mode = 0;
    d.table.clear();
    for (int mode = 2; mode >= 1 && cap; mode--) {
        std::vector<Local> loc(16); // (parallel_tiles uses at most 16 workers)
        sbt::parallel_tiles(B, [&](uint32_t w, uint32_t s0, uint32_t s1) {
            Local &l = loc[w];
            l.lo = s0;
            l.hi = s1;
            bool have_last = false;
            Key last_key{};
            uint32_t last_row = 0;
            for (uint32_t s = s0; s < s1; s++) {
                Key k;
                const float *f = hb[s].f; // length, target, last, spring, damp, yield, limit
                float row[5] = {mode == 2 ? f[0] : 0.0f, f[3], f[4], f[5], f[6]};
                memcpy(k.w, row, sizeof row);
                if (have_last && k == last_key) { // runs of one material are the rule: no hash lookup
                    d.of_slot[s] = last_row;
                    continue;
                }
                auto it = l.dict.find(k);
                if (it == l.dict.end()) {
                    if (l.dict.size() >= cap) {
                        l.ok = false;
                        return;
                    }
                    it = l.dict.emplace(k, (uint32_t)l.keys.size()).first;
                    l.keys.push_back(k);
                    l.first.push_back(s);
                }
                d.of_slot[s] = it->second;
                last_key = k;
                last_row = it->second;
                have_last = true;
            }
        });
        bool ok = true;
        std::unordered_map<Key, uint32_t, KeyHash> first_of; // key -> lowest slot that uses it
        for (const Local &l : loc) {
            ok = ok && l.ok;
            if (!ok) break;
            for (size_t j = 0; j < l.keys.size(); j++) {
                auto it = first_of.emplace(l.keys[j], l.first[j]).first;
                it->second = std::min(it->second, l.first[j]);
            }
            if (first_of.size() > cap) ok = false;
        }
        if (!ok) continue;
        std::vector<std::pair<uint32_t, Key>> rows;
        for (const auto &kv : first_of) rows.emplace_back(kv.second, kv.first);
        std::sort(rows.begin(), rows.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        std::unordered_map<Key, uint32_t, KeyHash> row_of;
        std::vector<float> table;
        for (const auto &r : rows) {
            row_of.emplace(r.second, (uint32_t)row_of.size());
            float row[5];
            memcpy(row, r.second.w, sizeof row);
            table.insert(table.end(), row, row + 5);
            table.push_back(mode == 2 ? 1.0f / row[0] : 0.0f); // 1 / length: one IEEE divide per material
        }
        for (Local &l : loc) {
            l.remap.resize(l.keys.size());
            for (size_t j = 0; j < l.keys.size(); j++) l.remap[j] = row_of[l.keys[j]];
        }
        sbt::parallel_tiles(B, [&](uint32_t w, uint32_t s0, uint32_t s1) {
            const Local &l = loc[w];
            for (uint32_t s = s0; s < s1; s++) d.of_slot[s] = l.remap[d.of_slot[s]];
        });
        d.mode = (uint32_t)mode;
        d.table.swap(table);
        return;
    }
}

// ---- pinned staging (see sb_engine.h).  stage_put: host -> device; `fill(off, len, out)` writes bytes [off, off + len) of the
// source into `out` (pinned); it is called once per chunk, in order, and may itself use several threads.
#define SB_STAGE_CHUNK ((size_t)16 << 20)
static sb_status stage_init(sb_engine *e)
{
    if (e->stage[0]) return SB_OK;
    uint8_t *p = nullptr;
    SB_HIP(e, hipHostMalloc((void **)&p, 2 * SB_STAGE_CHUNK, hipHostMallocDefault));
    e->stage[0] = p;
    e->stage[1] = p + SB_STAGE_CHUNK;
    for (int k = 0; k < 2; k++) SB_HIP(e, hipEventCreateWithFlags(&e->stage_done[k], hipEventDisableTiming));
    return SB_OK;
}
template <typename F>
static sb_status stage_put(sb_engine *e, void *dst, size_t bytes, F fill)
{
    SB_TRY(stage_init(e));
    for (size_t off = 0; off < bytes; off += SB_STAGE_CHUNK) {
        const int b = e->stage_cur;
        e->stage_cur ^= 1;
        if (e->stage_busy[b]) SB_HIP(e, hipEventSynchronize(e->stage_done[b]));
        const size_t len = std::min(SB_STAGE_CHUNK, bytes - off);
        fill(off, len, e->stage[b]);
        SB_HIP(e, hipMemcpyAsync((uint8_t *)dst + off, e->stage[b], len, hipMemcpyHostToDevice, e->stream));
        SB_HIP(e, hipEventRecord(e->stage_done[b], e->stream));
        e->stage_busy[b] = true;
    }
    return SB_OK;
}
// a host array as it is
static sb_status stage_put_bytes(sb_engine *e, void *dst, const void *src, size_t bytes)
{
    return stage_put(e, dst, bytes, [&](size_t off, size_t len, uint8_t *out) {
        sbt::parallel_ranges(len, (size_t)1 << 20, [&](size_t a, size_t b) { memcpy(out + a, (const uint8_t *)src + off + a, b - a); });
    });
}
// device -> host, `drain(off, len, in)` consumes bytes [off, off + len) from `in` (pinned); chunk k is drained while k + 1 flies
template <typename F>
static sb_status stage_get(sb_engine *e, const void *src, size_t bytes, F drain)
{
    SB_TRY(stage_init(e));
    for (int k = 0; k < 2; k++)
        if (e->stage_busy[k]) {
            SB_HIP(e, hipEventSynchronize(e->stage_done[k]));
            e->stage_busy[k] = false;
        }
    const size_t n = (bytes + SB_STAGE_CHUNK - 1) / SB_STAGE_CHUNK;
    auto issue = [&](size_t k) -> hipError_t {
        const size_t off = k * SB_STAGE_CHUNK, len = std::min(SB_STAGE_CHUNK, bytes - off);
        hipError_t r = hipMemcpyAsync(e->stage[k & 1], (const uint8_t *)src + off, len, hipMemcpyDeviceToHost, e->stream);
        return r != hipSuccess ? r : hipEventRecord(e->stage_done[k & 1], e->stream);
    };
    if (n) SB_HIP(e, issue(0));
    for (size_t k = 0; k < n; k++) {
        SB_HIP(e, hipEventSynchronize(e->stage_done[k & 1]));
        if (k + 1 < n) SB_HIP(e, issue(k + 1));
        const size_t off = k * SB_STAGE_CHUNK;
        drain(off, std::min(SB_STAGE_CHUNK, bytes - off), e->stage[k & 1]);
    }
    return SB_OK;
}

static sb_status stage_get_bytes(sb_engine *e, void *dst, const void *src, size_t bytes)
{
    return stage_get(e, src, bytes, [&](size_t off, size_t len, const uint8_t *in) {
        sbt::parallel_ranges(len, (size_t)1 << 20, [&](size_t a, size_t b) { memcpy((uint8_t *)dst + off + a, in + a, b - a); });
    });
}

template <typename T, typename A>
static sb_status dev_upload(sb_engine *e, T **p, const std::vector<T, A> &v)
{
    SB_TRY(dev_alloc(e, p, v.size()));
    if (!v.empty()) SB_TRY(stage_put_bytes(e, *p, v.data(), v.size() * sizeof(T)));
    return SB_OK;
}

// Beam state of a blocked plan (the engine's, or the one beside a tiled layout), one entry per beam in owner order, from the
// records of an upload: buffer 0 = uploaded, buffer 1 = the same targets (a tile that never yields never stores its targets:
// both buffers must hold them) and zeroed lengths; the plastic flags -- a tile starts unyielded when every beam it owns is
// uploaded with target_length == length, bit for bit.  Gathered straight into the pinned staging chunks.
static sb_status blocked_state_to_device(sb_engine *e, SbBlockedDev &k, const SbHostBeams &hb)
{
    const uint32_t B = k.nbeams, T = k.ntiles;
    float *dst[4] = {k.d_target[0], k.d_last[0], k.d_strain, k.d_stress};
    const int field[4] = {1, 2, 7, 8};
    const uint32_t *slot_of = k.h_beam_slot.data();
    for (int a = 0; a < 4 && B; a++) {
        const int fld = field[a];
        SB_TRY(stage_put(e, dst[a], (size_t)B * 4, [&](size_t off, size_t len, uint8_t *out) {
            float *o = (float *)out;
            const size_t g_first = off / 4, n = len / 4;
            sbt::parallel_ranges(n, 1 << 16, [&](size_t i0, size_t i1) {
                for (size_t i = i0; i < i1; i++) o[i] = hb[slot_of[g_first + i]].f[fld];
            });
        }));
    }
    if (B) SB_HIP(e, hipMemcpyAsync(k.d_target[1], k.d_target[0], (size_t)B * 4, hipMemcpyDeviceToDevice, e->stream));
    SB_HIP(e, hipMemsetAsync(k.d_last[1], 0, std::max<size_t>(B, 1) * 4, e->stream));
    std::vector<uint32_t> pl(std::max<uint32_t>(T, 1), 0u);
    sbt::parallel_ranges(T, 16, [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; t++)
            for (uint32_t g = k.h_tile_b0[t]; g < k.h_tile_b0[t + 1] && !pl[t]; g++) {
                const float *f = hb[slot_of[g]].f;
                pl[t] = memcmp(&f[0], &f[1], 4) != 0;
            }
    });
    k.pristine = std::none_of(pl.begin(), pl.begin() + T, [](uint32_t x) { return x != 0; });
    for (int b = 0; b < 2; b++) SB_TRY(stage_put_bytes(e, k.d_plastic[b], pl.data(), std::max<size_t>(T, 1) * 4));
    k.cur = 0;
    return SB_OK;
}

// device side of the temporally blocked plan; on return blockK is 0 if the scene cannot use it (more material rows
// than the 8 spare bits of an entry word address) and the caller falls back to the single-substep tiling
// `hybrid`: the plan goes into e->hy, BESIDE the tiled layout of an SB_COLLIDE_GRID engine (which stays the layout every other
// entry point sees), with beam-state, strain/stress and break-flag arrays of its own; `tiled_copy_of_slot` / `tiled_copy_slot`
// are that layout's maps (beam slot -> one of its copies; copy -> beam slot).
static sb_status upload_blocked(sb_engine *e, const SbBlocking &bl, const SbHostBeams &hb, uint32_t &blockK, SbStageTimer &tm,
                                bool hybrid = false, const std::vector<uint32_t> *tiled_copy_of_slot = nullptr,
                                const std::vector<uint32_t> *tiled_copy_slot = nullptr)
{
    const uint32_t B = (uint32_t)hb.size(), T = bl.ntiles;
    SbMatDict md;
    build_material_dictionary(md, hb, 1u << (32u - 2u * SB_BK_LBITS));
    bool plain_yield = true; // sb_beam_group applies sign(strain) as a copied sign bit, exact for yield_strain >= 0
    for (size_t r = 0; r < md.table.size(); r += 6) plain_yield = plain_yield && md.table[r + 3] >= 0.0f;
    // ... and the force scale early, as a factor of spring and damp: exact while they are zero or ordinary numbers (SbBeamMat::sd)
    for (size_t r = 0; r < md.table.size(); r += 6)
        for (int f = 1; f <= 2; f++) {
            const float v = std::fabs(md.table[r + f]);
            plain_yield = plain_yield && (v == 0.0f || (v >= SB_BK_SD_MIN && v <= SB_BK_SD_MAX));
        }
    if (md.mode == 0 || !plain_yield) {
        blockK = 0;
        return SB_OK;
    }
    tm.mark("  material dictionary");
    SbBlockedDev &k = hybrid ? e->hy : e->bk;
    k.K = blockK;
    k.plan_K = bl.K;
    k.cap = bl.max_region;
    k.dummy_word = (SB_BK_MAXP * SB_BK_T) | ((SB_BK_MAXP * SB_BK_T + 1u) << SB_BK_LBITS); // the two dummy LDS records, material row 0
    k.cur = 0;
    k.entries = bl.ent_la.size();
    k.halo_entries = bl.ent_state.size();
    k.halo_particles = bl.halo_idx.size();
    k.fixed_depth = e->opt.block_substeps != 0;
    {
        uint32_t first = 0, n_first = 0; // what a long call's launches look like (sbk_split_call prices the candidates)
        sbk_split_call(960u, blockK, k.fixed_depth, &first, &n_first);
        k.k_long = first ? first : blockK;
    }
    for (uint32_t d = 1; d <= blockK; d++) {
        k.entries_at[d] = bl.sum_entries_at[d];
        k.region_at[d] = bl.sum_region_at[d];
    }
    k.ntiles = T;
    k.nbeams = B;
    k.mat_mode = md.mode;
    k.nmat = (uint32_t)(md.table.size() / 6);
    // the dynamic part of the kernel's LDS: material rows, and with per-entry rest lengths (mode 1) one float per entry
    k.lds_bytes = (size_t)k.nmat * 8 * sizeof(float) + (md.mode == 1 ? (size_t)SB_BK_MAXB * SB_BK_T * sizeof(float) : 0);
    if (!hybrid) {
        e->ntiles = T;
        e->nhalo = (uint32_t)bl.halo_idx.size();
        e->tile_cap_own = bl.max_own;
        e->tile_cap_all = bl.max_region;
        e->mat_mode = md.mode;
        e->lbits = SB_BK_LBITS;
        e->nmat = k.nmat;
        e->nbeam = B;
        e->lds_bytes = k.lds_bytes;
        e->h_copy_of_slot.assign(bl.g_of_slot.begin(), bl.g_of_slot.end());
    }
    sbt::uvec<uint32_t> words(bl.ent_la.size());
    sbt::uvec<float> lengths;
    if (md.mode == 1) lengths.resize(words.size());
    sbt::parallel_ranges(words.size(), 1 << 16, [&](size_t j0, size_t j1) {
        for (size_t j = j0; j < j1; j++) {
            const uint32_t s = bl.ent_slot[j];
            words[j] = bl.ent_la[j] | (bl.ent_lb[j] << SB_BK_LBITS) | (md.of_slot[s] << (2u * SB_BK_LBITS));
            if (md.mode == 1) lengths[j] = hb[s].f[0];
        }
    });
    tm.mark("  entry words");
    SB_TRY(dev_upload(e, &k.d_tile_p0, bl.tile_p0));
    SB_TRY(dev_upload(e, &k.d_tile_h0, bl.tile_h0));
    SB_TRY(dev_upload(e, &k.d_halo_idx, bl.halo_idx));
    SB_TRY(dev_upload(e, &k.d_ring_cnt, bl.ring_cnt));
    SB_TRY(dev_upload(e, &k.d_tile_b0, bl.tile_b0));
    SB_TRY(dev_upload(e, &k.d_tile_e0, bl.tile_e0));
    SB_TRY(dev_upload(e, &k.d_tile_s0, bl.tile_s0));
    SB_TRY(dev_upload(e, &k.d_ent_word, words));
    SB_TRY(dev_upload(e, &k.d_ent_state, bl.ent_state));
    SB_TRY(dev_upload(e, &k.d_lvl_cnt, bl.lvl_cnt));
    SB_TRY(dev_upload(e, &k.d_tile_n0, bl.tile_n0));
    SB_TRY(dev_upload(e, &k.d_tile_nb, bl.tile_nb));
    SB_TRY(dev_upload(e, &k.d_slot_e0, bl.slot_e0));
    SB_TRY(dev_upload(e, &k.d_slot_ent, bl.slot_ent));
    SB_TRY(dev_upload(e, &k.d_ent_length, lengths));
    SB_TRY(dev_upload(e, &k.d_mat, md.table));
    uint32_t *d_beam_slot = nullptr;
    SB_TRY(dev_upload(e, &d_beam_slot, bl.beam_slot));
    if (!hybrid) {
        e->d_mat = k.d_mat;
        e->beams.slot = d_beam_slot;
    }
    // what an upload of the same topology needs again (rewrite_scene_state): the words as they are now, the two host maps
    SB_TRY(dev_alloc(e, &k.d_ent_word0, words.size()));
    if (!words.empty()) SB_HIP(e, hipMemcpyAsync(k.d_ent_word0, k.d_ent_word, words.size() * 4, hipMemcpyDeviceToDevice, e->stream));
    k.h_beam_slot.assign(bl.beam_slot.begin(), bl.beam_slot.end());
    k.h_tile_b0.assign(bl.tile_b0.begin(), bl.tile_b0.end());
    tm.mark("  plan arrays to device");
    float **dst[4] = {&k.d_target[0], &k.d_last[0], &k.d_strain, &k.d_stress};
    for (int a = 0; a < 4; a++) SB_TRY(dev_alloc(e, dst[a], (size_t)B + (a >= 2 ? SB_BK_T : 0u))); // (strain / stress: one dump element per thread behind the last beam, sb_blocked.hip)
    SB_TRY(dev_alloc(e, &k.d_target[1], B));
    SB_TRY(dev_alloc(e, &k.d_last[1], B));
    for (int b = 0; b < 2; b++) SB_TRY(dev_alloc(e, &k.d_plastic[b], T));
    SB_TRY(blocked_state_to_device(e, k, hb));
    tm.mark("  beam state to device");
    sbk_preload_blocked(k, hybrid); // (or the first launch of each kernel variant resolves it inside somebody's timed call)
    tm.mark("  kernels resolved");
    if (hybrid) {
        // break flags of its own (merged into the tiled layout's on the way back), the maps between the two layouts, the
        // running state of a tracked run
        SB_TRY(dev_alloc(e, &k.d_broken, (B + 31) / 32));
        for (int b = 0; b < 2; b++) SB_TRY(dev_alloc(e, &k.d_broken_new[b], (B + 31) / 32));
        SB_HIP(e, hipMemsetAsync(k.d_broken, 0, std::max<size_t>((B + 31) / 32, 1) * 4, e->stream));
        for (int b = 0; b < 2; b++) SB_HIP(e, hipMemsetAsync(k.d_broken_new[b], 0, std::max<size_t>((B + 31) / 32, 1) * 4, e->stream));
        sbt::uvec<uint32_t> copy_of_g(B), g_of_copy(tiled_copy_slot->size());
        sbt::parallel_ranges(B, 1 << 16, [&](size_t g0, size_t g1) {
            for (size_t g = g0; g < g1; g++) copy_of_g[g] = (*tiled_copy_of_slot)[bl.beam_slot[g]];
        });
        sbt::parallel_ranges(g_of_copy.size(), 1 << 16, [&](size_t c0, size_t c1) {
            for (size_t c = c0; c < c1; c++) g_of_copy[c] = (*tiled_copy_slot)[c] == 0xFFFFFFFFu ? 0xFFFFFFFFu : bl.g_of_slot[(*tiled_copy_slot)[c]];
        });
        SB_TRY(dev_upload(e, &k.d_copy_of_g, copy_of_g));
        SB_TRY(dev_upload(e, &k.d_g_of_copy, g_of_copy));
        SB_TRY(dev_alloc(e, &k.d_q, 2));
        SB_TRY(dev_alloc(e, &k.d_hslots, 3 * 65 * 4));
        SB_HIP(e, hipMemsetAsync(k.d_hslots, 0, 3 * 65 * 16, e->stream));
        SB_HIP(e, hipMemsetAsync(k.d_q, 0, 2 * sizeof(SbHybridCtl), e->stream));
        k.qpar = k.seq = k.run_launches = k.k_prev = 0;
        k.synced_delete_gen = 0;
        k.slow_chunk = 0;
        k.slow_left = 0;
        return SB_OK;
    }
    k.d_broken = nullptr; // (set by the caller once the engine's mask exists)
    e->beams.strain = k.d_strain;
    e->beams.stress = k.d_stress;
    e->beams.target = k.d_target[0];
    e->beams.last = k.d_last[0];
    // buffer A holds whatever accelerations were uploaded; buffer B is all zeros (engineWorker.ts:593)
    SB_TRY(dev_alloc(e, &e->d_acc_flag[0], T));
    SB_TRY(dev_alloc(e, &e->d_acc_flag[1], T));
    SB_HIP(e, hipMemset(e->d_acc_flag[0], 0x01, std::max<size_t>(T, 1) * 4));
    SB_HIP(e, hipMemset(e->d_acc_flag[1], 0x00, std::max<size_t>(T, 1) * 4));
    return SB_OK;
}

extern "C" {

uint32_t sb_abi_version(void) { return SB_ABI_VERSION; }

const char *sb_last_error(const sb_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

void sb_default_options(sb_options *o)
{
    memset(o, 0, sizeof *o);
    o->struct_size = sizeof *o;
    o->bounds_size = 1000.0f;   // engineWorker.ts:39
    o->particle_radius = 10.0f; // engineWorker.ts:40
    o->subticks = 64;           // engineWorker.ts:41
    o->max_particles = 65536;   // engineMapping.ts:362
    o->max_beams = 65536;       // engineMapping.ts:363
    o->layout = SB_LAYOUT_V1;
    o->collision_mode = SB_COLLIDE_GRID; // the bits of the reference's all-pairs scan (compute.wgsl:142-170), not its O(P^2) cost
    o->path = SB_PATH_AUTO;
    o->device_ordinal = 0;
}

sb_status sb_create(const sb_options *opts, sb_engine **out)
{
    sb_engine *none = nullptr;
    if (!opts || !out) SB_FAIL(none, SB_ERR_INVALID, "sb_create: null argument");
    *out = nullptr;
    if (opts->struct_size != sizeof(sb_options))
        SB_FAIL(none, SB_ERR_INVALID, "sb_create: sb_options.struct_size %u != %zu", opts->struct_size, sizeof(sb_options));
    if (opts->layout != SB_LAYOUT_V1 && opts->layout != SB_LAYOUT_V2)
        SB_FAIL(none, SB_ERR_INVALID, "sb_create: unknown layout %u", opts->layout);
    if (opts->layout == SB_LAYOUT_V1 && (opts->max_particles > 65536 || opts->max_beams > 65536))
        SB_FAIL(none, SB_ERR_INVALID, "sb_create: v1 layout holds at most 65536 particles/beams (u16 indices)");
    if (opts->max_particles == 0) SB_FAIL(none, SB_ERR_INVALID, "sb_create: max_particles is 0");
    if (opts->collision_mode > SB_COLLIDE_GRID) SB_FAIL(none, SB_ERR_INVALID, "sb_create: unknown collision_mode");
    if (opts->path > SB_PATH_TILED) SB_FAIL(none, SB_ERR_INVALID, "sb_create: unknown path");
    if (!(opts->particle_radius > 0.f) || !(opts->bounds_size > 0.f) || opts->subticks == 0)
        SB_FAIL(none, SB_ERR_INVALID, "sb_create: radius, bounds and subticks must be positive");
    int ndev = 0;
    hipError_t r = hipGetDeviceCount(&ndev);
    if (r != hipSuccess || ndev <= 0)
        SB_FAIL(none, SB_ERR_NO_DEVICE, "no HIP device available (%s); this engine has no CPU fallback",
                r == hipSuccess ? "device count 0" : hipGetErrorString(r));
    if (opts->device_ordinal < 0 || opts->device_ordinal >= ndev)
        SB_FAIL(none, SB_ERR_NO_DEVICE, "device_ordinal %d out of range (%d devices)", opts->device_ordinal, ndev);
    sb_engine *e = new sb_engine();
    e->opt = *opts;
    e->device = opts->device_ordinal;
    e->subticks = (opts->subticks + 1) / 2 * 2; // engineWorker.ts:90
    e->prm.bounds_size = opts->bounds_size;
    e->prm.particle_radius = opts->particle_radius;
    e->prm.time_step = 1.0f / (float)e->subticks; // engineWorker.ts:331
    {
        const float dt2 = e->prm.time_step * e->prm.time_step;
        uint32_t bits;
        memcpy(&bits, &dt2, 4);
        const bool pow2 = (bits & 0x007fffffu) == 0u && (bits >> 23) > 1u && (bits >> 23) < 253u; // normal, exact reciprocal
        e->prm.inv_dt2 = pow2 ? 1.0f / dt2 : 0.0f;
    }
    // the all-pairs scan sums contacts in ascending slot order (compute.wgsl:144); only the
    // atomic path keeps particles in slot order, so it serves that mode
    e->path = opts->path != SB_PATH_AUTO ? opts->path
              : (opts->collision_mode == SB_COLLIDE_ALLPAIRS ? SB_PATH_ATOMIC : SB_PATH_TILED);
    if (e->path == SB_PATH_TILED && opts->collision_mode == SB_COLLIDE_ALLPAIRS) {
        g_create_error = "SB_PATH_TILED does not implement SB_COLLIDE_ALLPAIRS (use SB_COLLIDE_GRID, same bits, or SB_PATH_ATOMIC)";
        delete e;
        return SB_ERR_UNSUPPORTED;
    }
    if ((r = hipSetDevice(e->device)) != hipSuccess || (r = hipStreamCreate(&e->stream)) != hipSuccess ||
        (r = hipEventCreate(&e->ev0)) != hipSuccess || (r = hipEventCreate(&e->ev1)) != hipSuccess ||
        (r = hipHostMalloc((void **)&e->dev_err, 1024, hipHostMallocMapped)) != hipSuccess) {
        g_create_error = std::string("HIP init failed: ") + hipGetErrorString(r);
        delete e;
        return SB_ERR_HIP;
    }
    *e->dev_err = 0;
    *out = e;
    return SB_OK;
}

sb_status sb_destroy(sb_engine *e)
{
    if (!e) return SB_ERR_INVALID;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    reap_join(e);
    free_scene(e);
    for (const auto &b : e->pool_free) (void)hipFree(b.first);
    e->pool_free.clear();
    if (e->dev_err) (void)hipHostFree(e->dev_err);
    if (e->stage[0]) (void)hipHostFree(e->stage[0]);
    for (int k = 0; k < 2; k++)
        if (e->stage_done[k]) (void)hipEventDestroy(e->stage_done[k]);
    for (hipEvent_t m : e->marks)
        if (m) (void)hipEventDestroy(m);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return SB_OK;
}

// Test hook (tests/test_gpu_grid_schedule.py; read once per process): the count of executed substeps an upload starts from.  The
// device counts in 32 bits, the host picks the slot set by count % 3: SB_GRID_EXECUTED0=4294967200 puts the wrap a hundred
// substeps behind every upload (r04: the host count is 64 bits wide since; it used to wrap out of step with the sets).
static inline uint32_t sb_grid_executed0()
{
    static const uint32_t v = [] { const char *s = getenv("SB_GRID_EXECUTED0"); return s ? (uint32_t)strtoull(s, nullptr, 10) : 0u; }();
    return v;
}

// the caller's beam slots (those of the latest upload) -> the engine's own (sb_engine.h h_user_slot)
static inline uint32_t sb_user_beams(const sb_engine *e) { return e->h_user_slot.empty() ? e->B : (uint32_t)e->h_user_slot.size(); }
static inline uint32_t sb_user_slot(const sb_engine *e, size_t u) { return e->h_user_slot.empty() ? (uint32_t)u : e->h_user_slot[u]; }

// An upload of the SAME topology (same counts, same mapping, every beam between the same two particles with the same rest
// length and material) as the scene on the device -- the editor moved or nudged something, or a caller steps from a saved
// state again: everything the plan is made of (bisection, rings, entry lists, material rows, endpoint words, the hash's
// arrays) is still right, and only state has to travel: particles, the beams' target / last lengths and strain / stress,
// flags and masks back to "just uploaded".  10x cheaper than planning again (the reference re-uploads everything on every
// edit: engineWorker.ts:497-507,580-597).  *kept = false: something differs, nothing was touched that the full path does
// not write again.  SB_KEEP_PLAN=0 turns it off.
static sb_status rewrite_scene_state(sb_engine *e, const uint8_t *md, const uint8_t *mp, const uint8_t *pd, const uint8_t *bd, bool *kept)
{
    *kept = false;
    static const bool off = [] { const char *v = getenv("SB_KEEP_PLAN"); return v && atoi(v) == 0; }();
    const uint32_t maxP = e->opt.max_particles, maxB = e->opt.max_beams, bstride = beam_stride(e);
    const uint32_t P = rd_u32(md + 4), B = rd_u32(md + 24);
    const size_t map_bytes = (size_t)(maxP + (size_t)maxB) * map_isz(e), isz = map_isz(e);
    const uint32_t Bu = e->loaded ? sb_user_beams(e) : 0u; // beams of the upload before this one, in the caller's slots
    if (off || !e->loaded || P != e->P || B > Bu || e->h_beams.size() != e->B || e->h_pidx.size() != P || e->h_mapping.size() != map_bytes)
        return SB_OK;
    if (Bu - B > Bu / 8u) return SB_OK; // (an edit, not another scene)
    if (e->n_ghost_p || e->n_send_p || e->n_ghost_b || e->n_send_b || e->n_peers || e->mailbox) return SB_OK; // (ghost zones: configured per upload)
    SbStageTimer tm;
    std::atomic<bool> same{true};
    // the particle slots must map as before; the beam slots may map differently when beams were removed (the caller's slots are
    // renumbered then: engineMapping.ts:500-518 writes the beams that are left in order, slot i at data index i)
    const size_t cmp_bytes = B == Bu ? map_bytes : (size_t)maxP * isz;
    sbt::parallel_ranges(cmp_bytes, (size_t)1 << 20, [&](size_t a, size_t b) {
        if (memcmp(mp + a, e->h_mapping.data() + a, b - a) != 0) same.store(false, std::memory_order_relaxed);
    });
    if (!same.load()) return SB_OK;
    if (B != Bu && memcmp(mp + (size_t)maxP * isz, e->h_mapping.data() + (size_t)maxP * isz, (size_t)B * isz) != 0) {
        // a different map for the beams that are left: it has to be a partial injection like any other (sb_write_buffers_impl)
        std::vector<uint8_t> seen((size_t)maxB, 0);
        for (uint32_t u = 0; u < B; u++) {
            const uint32_t idx = map_get(e, mp, (size_t)maxP + u);
            if (idx >= maxB || seen[idx]) return SB_OK; // (the full path says what is wrong with it)
            seen[idx] = 1;
        }
    }
    const bool v1 = e->opt.layout == SB_LAYOUT_V1;
    // one record of the new upload against one beam of the engine: endpoints and static parameters must match what the plan was
    // made for; the state fields are taken over on a match (if the upload turns out not to fit, the full path replaces every
    // record anyway)
    auto take = [&](size_t u, uint32_t s) -> bool {
        const uint8_t *rec = bd + (size_t)map_get(e, mp, (size_t)maxP + u) * bstride;
        uint32_t a, b;
        float f[9];
        if (v1) {
            const uint32_t pair = rd_u32(rec);
            a = pair & 0xffffu;
            b = pair >> 16;
            memcpy(f, rec + 4, sizeof f);
        } else {
            a = rd_u32(rec);
            b = rd_u32(rec + 4);
            memcpy(f, rec + 8, sizeof f);
        }
        SbHostBeam &h = e->h_beams[s];
        if (a != h.da || b != h.db || memcmp(&f[0], &h.f[0], 4) != 0 || memcmp(&f[3], &h.f[3], 16) != 0) return false;
        h.f[1] = f[1];
        h.f[2] = f[2];
        h.f[7] = f[7];
        h.f[8] = f[8];
        return true;
    };
    std::vector<uint32_t> user_slot; // (stays empty when the caller's slots are the engine's)
    if (B == Bu) {
        sbt::parallel_ranges(B, 1 << 15, [&](size_t s0, size_t s1) {
            if (!same.load(std::memory_order_relaxed)) return;
            for (size_t s = s0; s < s1; s++)
                if (!take(s, sb_user_slot(e, s))) {
                    same.store(false, std::memory_order_relaxed);
                    return;
                }
        });
        if (!same.load()) return SB_OK;
        user_slot = e->h_user_slot;
    } else {
        // Beams were removed: the records that are left are a subsequence of the old ones (sb_edit.h)
        auto same_key = [&](size_t u, size_t o) {
            const uint8_t *rec = bd + (size_t)map_get(e, mp, (size_t)maxP + u) * bstride;
            const uint32_t a = v1 ? rd_u32(rec) & 0xffffu : rd_u32(rec), b = v1 ? rd_u32(rec) >> 16 : rd_u32(rec + 4);
            const SbHostBeam &h = e->h_beams[sb_user_slot(e, o)];
            return h.da == a && h.db == b;
        };
        std::vector<uint32_t> old_of_new;
        if (!sbe::match_subsequence(B, Bu, (size_t)1 << 15, same_key, [&](size_t u, size_t o) { return take(u, sb_user_slot(e, o)); }, old_of_new))
            return SB_OK;
        user_slot.resize(B);
        for (uint32_t u = 0; u < B; u++) user_slot[u] = sb_user_slot(e, old_of_new[u]);
    }
    tm.mark("same topology: mapping + beam records");
    std::vector<float2> hp(P), hv(P), ha(P);
    {
        struct Box { float minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY; };
        std::vector<Box> boxes(256);
        std::atomic<uint32_t> nbox{0};
        sbt::parallel_ranges(P, 1 << 16, [&](size_t i0, size_t i1) {
            Box bx;
            for (size_t i = i0; i < i1; i++) {
                float q[6];
                memcpy(q, pd + (size_t)e->h_pidx[i] * SB_PARTICLE_STRIDE, SB_PARTICLE_STRIDE);
                hp[i] = make_float2(q[0], q[1]);
                hv[i] = make_float2(q[2], q[3]);
                ha[i] = make_float2(q[4], q[5]);
                if (std::isfinite(q[0]) && std::isfinite(q[1])) {
                    bx.minx = std::min(bx.minx, q[0]); bx.maxx = std::max(bx.maxx, q[0]);
                    bx.miny = std::min(bx.miny, q[1]); bx.maxy = std::max(bx.maxy, q[1]);
                }
            }
            boxes[nbox.fetch_add(1) % boxes.size()] = bx; // (P / 65536 ranges: fewer than 256 up to 16 M particles; beyond, a box may be lost
        });                                              //  -- which only ever keeps a frame that a fresh upload would have moved)
        if (e->opt.collision_mode == SB_COLLIDE_GRID && e->d_grid_ctl) {
            Box all;
            for (const Box &b : boxes) {
                all.minx = std::min(all.minx, b.minx); all.maxx = std::max(all.maxx, b.maxx);
                all.miny = std::min(all.miny, b.miny); all.maxy = std::max(all.maxy, b.maxy);
            }
            // the hash keeps its frame: fine while the scene is still inside it (outside, particles are clamped into edge
            // cells -- correct, but slow: plan again)
            if (all.minx <= all.maxx && !(all.minx >= e->grid.x0 && all.maxx <= e->grid.x0 + e->grid.width && all.miny >= e->grid.y0 &&
                                          all.maxy <= e->grid.y0 + e->grid.height))
                return SB_OK;
        }
    }
    // ---- from here on the scene on the device is rewritten
    e->h_metadata.assign(md, md + SB_METADATA_BYTES);
    if (B != Bu) e->h_mapping.assign(mp, mp + map_bytes);
    e->cur = 0;
    e->substeps_done = 0;
    if (e->dev_err) *e->dev_err = 0;
    SB_TRY(stage_put_bytes(e, e->part[0].pos, hp.data(), P * sizeof(float2)));
    SB_TRY(stage_put_bytes(e, e->part[0].vel, hv.data(), P * sizeof(float2)));
    SB_TRY(stage_put_bytes(e, e->part[0].acc, ha.data(), P * sizeof(float2)));
    SB_HIP(e, hipMemsetAsync(e->part[1].pos, 0, std::max<size_t>(P, 1) * sizeof(float2), e->stream));
    SB_HIP(e, hipMemsetAsync(e->part[1].vel, 0, std::max<size_t>(P, 1) * sizeof(float2), e->stream));
    SB_HIP(e, hipMemsetAsync(e->part[1].acc, 0, std::max<size_t>(P, 1) * sizeof(float2), e->stream));
    tm.mark("particles to device");
    const uint32_t nc = e->nbeam;
    if (e->bk.K) {
        SB_TRY(blocked_state_to_device(e, e->bk, e->h_beams));
        e->beams.target = e->bk.d_target[0];
        e->beams.last = e->bk.d_last[0];
        if (e->delete_gen && e->bk.entries)
            SB_HIP(e, hipMemcpyAsync(e->bk.d_ent_word, e->bk.d_ent_word0, (size_t)e->bk.entries * 4, hipMemcpyDeviceToDevice, e->stream));
    } else {
        float *dst[4] = {e->beams.target, e->beams.last, e->beams.strain, e->beams.stress};
        const int field[4] = {1, 2, 7, 8};
        const uint32_t *slot_of = e->h_slot_of_copy.data();
        for (int a = 0; a < 4 && nc; a++) {
            const int fld = field[a];
            SB_TRY(stage_put(e, dst[a], (size_t)nc * 4, [&](size_t off, size_t len, uint8_t *out) {
                float *o = (float *)out;
                const size_t c_first = off / 4, n = len / 4;
                sbt::parallel_ranges(n, 1 << 16, [&](size_t i0, size_t i1) {
                    for (size_t i = i0; i < i1; i++) {
                        const uint32_t sl = slot_of[c_first + i];
                        o[i] = sl == 0xFFFFFFFFu ? 0.0f : e->h_beams[sl].f[fld];
                    }
                });
            }));
        }
        if (e->delete_gen && e->live_words)
            SB_HIP(e, hipMemcpyAsync(e->path == SB_PATH_TILED ? e->beams.pair : e->beams.ia, e->d_live0, e->live_words * 4,
                                     hipMemcpyDeviceToDevice, e->stream));
    }
    if (e->hy.K) {
        SbBlockedDev &h = e->hy; // (its beam state is borrowed from the tiled layout at the start of every run)
        if (h.synced_delete_gen && h.entries)
            SB_HIP(e, hipMemcpyAsync(h.d_ent_word, h.d_ent_word0, (size_t)h.entries * 4, hipMemcpyDeviceToDevice, e->stream));
        h.synced_delete_gen = 0;
    }
    e->hy.slow_chunk = e->hy.slow_left = 0; // (also while its plan is still on the side thread: that one only depends on the topology)
    tm.mark("beam state to device");
    if (e->d_acc_flag[0]) { // buffer A holds whatever accelerations were uploaded; buffer B is all zeros (engineWorker.ts:593)
        SB_HIP(e, hipMemsetAsync(e->d_acc_flag[0], 0x01, std::max<size_t>(e->ntiles, 1) * 4, e->stream));
        SB_HIP(e, hipMemsetAsync(e->d_acc_flag[1], 0x00, std::max<size_t>(e->ntiles, 1) * 4, e->stream));
    }
    if (e->d_forces) SB_HIP(e, hipMemsetAsync(e->d_forces, 0, std::max<size_t>(P, 1) * sizeof(int2), e->stream));
    SB_HIP(e, hipMemsetAsync(e->d_broken, 0, std::max<size_t>((nc + 31) / 32, 1) * 4, e->stream));
    SB_HIP(e, hipMemsetAsync(e->d_dead_gen, 0, std::max<size_t>(e->B, 1) * 4, e->stream));
    e->delete_gen = 0;
    // The engine's beams that are not part of the scene any more (removed by this upload or by one before it) die like beams a
    // delete pass removes -- every copy / entry of them, in a pass of its own that is the upload's: generation 1; the passes of
    // the frames that follow count from 2, and none of the caller's slots ever carries 1.
    if (!user_slot.empty()) {
        std::vector<uint8_t> live(e->B, 0);
        for (uint32_t s : user_slot) live[s] = 1;
        const uint32_t *slot_of = e->bk.K ? e->bk.h_beam_slot.data() : e->h_slot_of_copy.data();
        const size_t n_of = e->bk.K ? e->bk.h_beam_slot.size() : e->h_slot_of_copy.size();
        if (n_of != nc) SB_FAIL(e, SB_ERR_STATE, "plan-keeping upload: %zu copies in the host map, %u on the device", n_of, nc);
        std::vector<uint32_t> mask((nc + 31) / 32, 0u);
        sbt::parallel_ranges(mask.size(), 1 << 12, [&](size_t w0, size_t w1) {
            for (size_t w = w0; w < w1; w++) {
                uint32_t bits = 0u;
                for (uint32_t k = 0; k < 32u && w * 32 + k < nc; k++) {
                    const uint32_t sl = slot_of[w * 32 + k];
                    if (sl != 0xFFFFFFFFu && !live[sl]) bits |= 1u << k;
                }
                mask[w] = bits;
            }
        });
        SB_TRY(stage_put_bytes(e, e->d_broken, mask.data(), mask.size() * 4));
        if (e->bk.K) sbk_launch_delete_blocked(e);
        else sbk_launch_delete(e);
        e->uploads_edited += B != Bu ? 1u : 0u;
    }
    e->h_user_slot.swap(user_slot);
    if (e->opt.collision_mode == SB_COLLIDE_GRID && e->d_grid_ctl) { // the hash: no build yet, same frame
        for (int k = 0; k < 2; k++) SB_HIP(e, hipMemsetAsync(e->d_head[k], 0, e->grid_heads * 8, e->stream));
        SB_HIP(e, hipMemsetAsync(e->d_blk_max, 0, e->grid_slots * 4, e->stream));
        SB_HIP(e, hipMemsetAsync(e->d_grid_outside, 0, 16, e->stream));
        SB_HIP(e, hipMemsetAsync(e->d_nl_count, 0, std::max<size_t>(P, 1) * 4, e->stream));
        SB_HIP(e, hipMemcpyAsync(e->d_grid_ctl, e->grid_ctl0, sizeof e->grid_ctl0, hipMemcpyHostToDevice, e->stream));
        e->grid_par = 0;
        e->grid_force = true;
        e->grid_classic_left = e->grid_classic_chunk = e->grid_calm = 0;
        e->grid_executed = sb_grid_executed0();
    }
    memcpy(&e->consts, md + 48, sizeof(SbConsts));
    SB_HIP(e, hipStreamSynchronize(e->stream));
    e->uploads_kept++;
    tm.mark("flags, masks, hash + final sync");
    *kept = true;
    return SB_OK;
}

static sb_status sb_write_buffers_impl(sb_engine *e, const void *metadata, size_t metadata_bytes, const void *mapping,
                           size_t mapping_bytes, const void *particles, size_t particles_bytes,
                           const void *beams, size_t beams_bytes)
{
    if (!e) return SB_ERR_INVALID;
    if (!metadata || !mapping || !particles || (!beams && e->opt.max_beams))
        SB_FAIL(e, SB_ERR_INVALID, "sb_write_buffers: null buffer");
    const uint32_t maxP = e->opt.max_particles, maxB = e->opt.max_beams;
    const uint32_t bstride = beam_stride(e);
    if (metadata_bytes < SB_METADATA_BYTES) SB_FAIL(e, SB_ERR_INVALID, "metadata buffer is %zu bytes, need 112", metadata_bytes);
    if (mapping_bytes < (size_t)(maxP + (size_t)maxB) * map_isz(e))
        SB_FAIL(e, SB_ERR_INVALID, "mapping buffer is %zu bytes, need %zu", mapping_bytes, (size_t)(maxP + (size_t)maxB) * map_isz(e));
    if (particles_bytes < (size_t)maxP * SB_PARTICLE_STRIDE)
        SB_FAIL(e, SB_ERR_INVALID, "particle buffer is %zu bytes, need %zu", particles_bytes, (size_t)maxP * SB_PARTICLE_STRIDE);
    if (beams_bytes < (size_t)maxB * bstride)
        SB_FAIL(e, SB_ERR_INVALID, "beam buffer is %zu bytes, need %zu", beams_bytes, (size_t)maxB * bstride);
    const uint8_t *md = (const uint8_t *)metadata, *mp = (const uint8_t *)mapping;
    const uint8_t *pd = (const uint8_t *)particles, *bd = (const uint8_t *)beams;
    const uint32_t P = rd_u32(md + 4), B = rd_u32(md + 24);
    if (rd_u32(md + 40) != maxP || rd_u32(md + 44) != maxB)
        SB_FAIL(e, SB_ERR_INVALID, "metadata max_particles/max_beams (%u/%u) differ from the engine capacity (%u/%u)",
                rd_u32(md + 40), rd_u32(md + 44), maxP, maxB);
    if (P > maxP || B > maxB) SB_FAIL(e, SB_ERR_INVALID, "metadata counts (%u/%u) exceed capacity (%u/%u)", P, B, maxP, maxB);

    SB_HIP(e, hipSetDevice(e->device));
    SB_HIP(e, hipStreamSynchronize(e->stream));
    reap_join(e);
    {
        bool kept = false;
        const sb_status st = rewrite_scene_state(e, md, mp, pd, bd, &kept);
        if (st != SB_OK) { // (past its point of no return the device holds half of each scene: nothing may step or read it)
            e->loaded = false;
            return st;
        }
        if (kept) return SB_OK;
    }
    free_scene(e);

    SbStageTimer tm;
    // ---- host shadows (copy semantics)
    e->h_metadata.assign(md, md + SB_METADATA_BYTES);
    e->h_mapping.assign(mp, mp + (size_t)(maxP + (size_t)maxB) * map_isz(e));
    e->P = P;
    e->B = B;
    e->h_user_slot.clear();
    e->cur = 0;
    e->substeps_done = 0;

    tm.mark("sync + free + shadows");
    // ---- particles: slot -> data index, must be a partial injection
    std::vector<uint32_t> slot_index(P), internal_of_index(maxP, 0xFFFFFFFFu);
    for (uint32_t s = 0; s < P; s++) {
        uint32_t idx = map_get(e, mp, s);
        if (idx >= maxP) SB_FAIL(e, SB_ERR_INVALID, "particle slot %u maps to data index %u >= max_particles", s, idx);
        if (internal_of_index[idx] != 0xFFFFFFFFu)
            SB_FAIL(e, SB_ERR_INVALID, "particle data index %u is mapped by two slots (%u and %u)", idx, internal_of_index[idx], s);
        internal_of_index[idx] = s; // provisional: slot
        slot_index[s] = idx;
    }
    // ---- beams: slot -> record; endpoints must be active particles
    SbHostBeams hb(B);
    {
        // a few host threads over the beam slots; the first offence (lowest slot of its chunk) is reported
        std::vector<uint8_t> seen((size_t)maxB, 0);
        struct Bad { uint32_t slot = 0xFFFFFFFFu, kind = 0, idx = 0, a = 0, b = 0; };
        std::vector<Bad> bad(64);
        std::atomic<uint32_t> nchunk{0};
        sbt::parallel_ranges(B, 1 << 15, [&](size_t s0, size_t s1) {
            Bad &mine = bad[nchunk.fetch_add(1) % bad.size()];
            for (size_t s = s0; s < s1; s++) {
                const uint32_t idx = map_get(e, mp, (size_t)maxP + s);
                Bad here;
                here.slot = (uint32_t)s;
                here.idx = idx;
                if (idx >= maxB) here.kind = 1;
                else if (__atomic_fetch_or(&seen[idx], 1, __ATOMIC_RELAXED)) here.kind = 2;
                if (!here.kind) {
                    const uint8_t *rec = bd + (size_t)idx * bstride;
                    uint32_t a, b;
                    const uint8_t *f;
                    if (e->opt.layout == SB_LAYOUT_V1) { // engineMapping.ts:183-186, compute.wgsl:99-100
                        uint32_t pair = rd_u32(rec);
                        a = pair & 0xffffu;
                        b = pair >> 16;
                        f = rec + 4;
                    } else {
                        a = rd_u32(rec);
                        b = rd_u32(rec + 4);
                        f = rec + 8;
                    }
                    here.a = a;
                    here.b = b;
                    if (a >= maxP || b >= maxP || internal_of_index[a] == 0xFFFFFFFFu || internal_of_index[b] == 0xFFFFFFFFu) here.kind = 3;
                    else {
                        SbHostBeam &h = hb[s];
                        h.a = internal_of_index[a]; // slot of endpoint A (re-indexed below)
                        h.b = internal_of_index[b];
                        h.da = a;
                        h.db = b;
                        memcpy(h.f, f, 9 * sizeof(float));
                    }
                }
                if (here.kind && here.slot < mine.slot) mine = here;
            }
        });
        Bad first;
        for (const Bad &c : bad)
            if (c.kind && c.slot < first.slot) first = c;
        if (first.kind == 1) SB_FAIL(e, SB_ERR_INVALID, "beam slot %u maps to data index %u >= max_beams", first.slot, first.idx);
        if (first.kind == 2) SB_FAIL(e, SB_ERR_INVALID, "beam data index %u is mapped by two slots", first.idx);
        if (first.kind == 3)
            SB_FAIL(e, SB_ERR_INVALID, "beam slot %u (data index %u) references particle data index %u/%u that no particle slot maps to",
                    first.slot, first.idx, first.a, first.b);
    }

    tm.mark("validate + beam records");
    // ---- internal particle order
    std::vector<float> px(P), py(P);
    sbt::parallel_ranges(P, 1 << 16, [&](size_t s0, size_t s1) {
        for (size_t s = s0; s < s1; s++) {
            memcpy(&px[s], pd + (size_t)slot_index[s] * SB_PARTICLE_STRIDE, 4);
            memcpy(&py[s], pd + (size_t)slot_index[s] * SB_PARTICLE_STRIDE + 4, 4);
        }
    });
    SbTiling tl;
    SbBlocking bl;
    uint32_t blockK = 0; // > 0: the temporally blocked plan is in use (sb_blocking.h)
    uint32_t plan_target = 0; // tile size the plan's bisection was made for
    uint32_t tile_target_used = 0; // ... and the single-substep tiling's
    std::vector<uint32_t> order; // internal -> slot
    if (e->path == SB_PATH_TILED) {
        uint32_t target = e->opt.tile_particles ? e->opt.tile_particles : 1024;
        const bool want_blocked = e->opt.collision_mode == SB_COLLIDE_OFF && e->opt.block_substeps != 1 && P && B;
        if (want_blocked && !e->opt.tile_particles) {
            // The blocked kernel keeps two tiles per CU resident (128 VGPRs), i.e. `slots` tiles at a time, and a launch
            // runs in whole rounds of them: 1026 tiles on 512 slots take three rounds where 1024 take two (a slab with its
            // ghost columns ran 27 % slower than the same slab without, for 5 % more particles).  So the tile size follows
            // the scene: the fewest rounds whose tiles stay within ~1100 particles, and then tiles that fill those rounds.
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device);
            const uint64_t slots = (uint64_t)std::max(cus, 1) * 2u;
            const uint64_t own_cap = SB_BK_OWNP * SB_BK_T; // (the kernel's own-particle slots)
            const uint64_t rounds = std::max<uint64_t>(1, (P + slots * own_cap - 1) / (slots * own_cap));
            target = (uint32_t)std::min<uint64_t>(own_cap, std::max<uint64_t>(256u, (P + slots * rounds - 1) / (slots * rounds)));
        }
        if (want_blocked) {
            // the plan is made as deep as asked for (default SB_BK_KPLAN) while every tile's region still fits the kernel's
            // per-thread register arrays and 12-bit local indices.  One breadth-first search to the depth asked for tells the
            // region and entry sizes of EVERY smaller depth (they are prefixes) -- and a launch of k substeps only ever loads the
            // depth-k prefix of the plan, so a plan that does not fit at full depth is kept as it is and the launches stay at the
            // deepest depth that does (r02 re-made the plan for that depth: 65 ms per million particles; before that once per candidate)
            // (by class: the kernel keeps own and halo items in slots of their own, sb_blocked.hip)
            auto fits = [&](uint32_t d) {
                return bl.max_own <= SB_BK_OWNP * SB_BK_T && bl.halo_at[d] <= SB_BK_HALOP * SB_BK_T && bl.max_ownb <= SB_BK_OWNB * SB_BK_T &&
                       bl.halo_entries_at[d] <= SB_BK_HALOB * SB_BK_T && bl.region_at[d] <= (1u << SB_BK_LBITS) - 2u;
            };
            blockK = std::min<uint32_t>(e->opt.block_substeps ? e->opt.block_substeps : SB_BK_KPLAN, SB_BK_KMAX);
            sb_build_blocking(bl, px, py, hb, target, blockK);
            if (bl.max_own > SB_BK_OWNP * SB_BK_T || bl.max_ownb > SB_BK_OWNB * SB_BK_T) {
                // the tile size owns more than the kernel's own slots hold (the automatic one on a scene with four or more beams
                // per particle; an explicit sb_options.tile_particles above 1024 -- it is an upper bound, include/softbody.h:
                // until r04 such a value silently lost the blocked kernel): smaller tiles, once
                const double shrink = std::min((double)(SB_BK_OWNP * SB_BK_T) / bl.max_own, (double)(SB_BK_OWNB * SB_BK_T) / std::max(bl.max_ownb, 1u));
                target = std::max<uint32_t>(128u, (uint32_t)(target * shrink * 0.97));
                sb_build_blocking(bl, px, py, hb, target, blockK);
            }
            plan_target = target;
            if (!fits(blockK)) { // the plan stays (a launch of k substeps loads the depth-k prefix of it): launches just stay shallower
                uint32_t fit = 0;
                for (uint32_t d = 1; d < blockK; d++)
                    if (fits(d)) fit = d;
                blockK = fit;
            }
        }
        if (blockK) {
            order = bl.order;
        } else {
            if (!e->opt.tile_particles) {
                // the single-substep kernel's own default: 1024 particles, four workgroups per CU.  A scene a few per cent
                // above a whole number of rounds of those slots (a 1000-column slab with its ghost columns: 1036 tiles) would
                // run one more, almost empty round -- tiles of up to 1100 particles that fit the rounds instead: 29.6 -> 26.7 us
                // with the hash on, 20.3 -> 18.9 without (smaller tiles in more rounds measured worse than either)
                int cus = 256;
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device);
                const uint64_t slots = (uint64_t)std::max(cus, 1) * 4u, full = ((uint64_t)P + 1023u) / 1024u / slots;
                target = 1024;
                if (full >= 1 && (uint64_t)P > slots * full * 1024u && (uint64_t)P <= slots * full * 1100u)
                    target = (uint32_t)(((uint64_t)P + slots * full - 1) / (slots * full));
            }
            sb_build_tiling(tl, px, py, hb, target);
            tile_target_used = target;
            order = tl.order;
        }
    } else {
        order.resize(P);
        std::iota(order.begin(), order.end(), 0u);
    }
    std::vector<uint32_t> internal_of_slot(P);
    e->h_pslot = order;
    e->h_pidx.resize(P);
    sbt::parallel_ranges(P, 1 << 16, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            internal_of_slot[order[i]] = (uint32_t)i;
            e->h_pidx[i] = slot_index[order[i]];
        }
    });

    tm.mark("tiling / blocking plan");
    // ---- upload particles (A = data, B = zero: engineWorker.ts:588,593)
    {
        std::vector<float2> hp(P), hv(P), ha(P);
        sbt::parallel_ranges(P, 1 << 16, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                float q[6];
                memcpy(q, pd + (size_t)e->h_pidx[i] * SB_PARTICLE_STRIDE, SB_PARTICLE_STRIDE);
                hp[i] = make_float2(q[0], q[1]);
                hv[i] = make_float2(q[2], q[3]);
                ha[i] = make_float2(q[4], q[5]);
            }
        });
        for (int k = 0; k < 2; k++) {
            SB_TRY(dev_alloc(e, &e->part[k].pos, P));
            SB_TRY(dev_alloc(e, &e->part[k].vel, P));
            SB_TRY(dev_alloc(e, &e->part[k].acc, P));
        }
        SB_TRY(stage_put_bytes(e, e->part[0].pos, hp.data(), P * sizeof(float2)));
        SB_TRY(stage_put_bytes(e, e->part[0].vel, hv.data(), P * sizeof(float2)));
        SB_TRY(stage_put_bytes(e, e->part[0].acc, ha.data(), P * sizeof(float2)));
        SB_HIP(e, hipMemset(e->part[1].pos, 0, std::max<size_t>(P, 1) * sizeof(float2)));
        SB_HIP(e, hipMemset(e->part[1].vel, 0, std::max<size_t>(P, 1) * sizeof(float2)));
        SB_HIP(e, hipMemset(e->part[1].acc, 0, std::max<size_t>(P, 1) * sizeof(float2)));
        SB_TRY(dev_alloc(e, &e->d_pidx, P));
        SB_TRY(dev_alloc(e, &e->d_pslot, P));
        SB_TRY(stage_put_bytes(e, e->d_pidx, e->h_pidx.data(), (size_t)P * 4));
        SB_TRY(stage_put_bytes(e, e->d_pslot, e->h_pslot.data(), (size_t)P * 4));
        e->d_islot = nullptr;
        if (e->opt.collision_mode == SB_COLLIDE_GRID && e->path == SB_PATH_TILED) { // (active slots are 0 .. P - 1)
            SB_TRY(dev_alloc(e, &e->d_islot, P));
            SB_TRY(stage_put(e, e->d_islot, (size_t)P * 4, [&](size_t off, size_t len, uint8_t *out) {
                uint32_t *o = (uint32_t *)out;
                const size_t first = off / 4, n = len / 4;
                sbt::parallel_ranges(n, 1 << 16, [&](size_t i0, size_t i1) {
                    for (size_t i = i0; i < i1; i++) o[i] = internal_of_slot[first + i];
                });
            }));
        }
    }

    tm.mark("particles to device");
    // ---- beam copies
    std::vector<uint32_t> c_ia, c_ib, c_pair, c_slot;
    std::vector<float> mat_table;
    e->h_copy_of_slot.assign(B, 0);
    e->bk = SbBlockedDev{};
    e->beams = SbBeamArrays{};
    if (blockK) {
        SB_TRY(upload_blocked(e, bl, hb, blockK, tm));
        if (!blockK) // no dictionary: the single-substep tiling after all, on the SAME bisection (the particles are already on
            sb_build_tiling(tl, px, py, hb, plan_target); // the device in the blocked plan's order, which follows its tile size)
        if (!blockK) tile_target_used = plan_target;
    }
    if (blockK) {
    } else if (e->path == SB_PATH_TILED) {
        e->ntiles = tl.ntiles;
        e->nhalo = (uint32_t)tl.halo_idx.size();
        c_slot = tl.copy_slot;
        e->h_copy_of_slot = tl.copy_of_slot;
        e->tile_cap_own = tl.max_own;
        e->tile_cap_all = tl.max_all;
        SB_TRY(dev_alloc(e, &e->d_tile_p0, tl.tile_p0.size()));
        SB_TRY(dev_alloc(e, &e->d_tile_b0, tl.tile_b0.size()));
        SB_TRY(dev_alloc(e, &e->d_tile_h0, tl.tile_h0.size()));
        SB_TRY(dev_alloc(e, &e->d_halo_idx, tl.halo_idx.size()));
        SB_HIP(e, hipMemcpy(e->d_tile_p0, tl.tile_p0.data(), tl.tile_p0.size() * 4, hipMemcpyHostToDevice));
        SB_HIP(e, hipMemcpy(e->d_tile_b0, tl.tile_b0.data(), tl.tile_b0.size() * 4, hipMemcpyHostToDevice));
        SB_HIP(e, hipMemcpy(e->d_tile_h0, tl.tile_h0.data(), tl.tile_h0.size() * 4, hipMemcpyHostToDevice));
        // buffer A holds whatever accelerations were uploaded; buffer B is all zeros (engineWorker.ts:593)
        SB_TRY(dev_alloc(e, &e->d_acc_flag[0], tl.ntiles));
        SB_TRY(dev_alloc(e, &e->d_acc_flag[1], tl.ntiles));
        SB_HIP(e, hipMemset(e->d_acc_flag[0], 0x01, std::max<size_t>(tl.ntiles, 1) * 4));
        SB_HIP(e, hipMemset(e->d_acc_flag[1], 0x00, std::max<size_t>(tl.ntiles, 1) * 4));
        if (!tl.halo_idx.empty())
            SB_HIP(e, hipMemcpy(e->d_halo_idx, tl.halo_idx.data(), tl.halo_idx.size() * 4, hipMemcpyHostToDevice));
        if (tl.max_all > 65535)
            SB_FAIL(e, SB_ERR_UNSUPPORTED, "a tile addresses %u particles (> 65535, 16-bit local indices): lower tile_particles", tl.max_all);
        // ---- material dictionary: beams that share (length, spring, damp, yield, limit) share one
        // table row, and the row number rides in the spare bits of the endpoint word.  Lossless;
        // falls back to (spring, damp, yield, limit) rows + per-copy length, then to per-copy arrays.
        uint32_t lbits = 1;
        while ((1u << lbits) <= tl.max_all) lbits++; // all-ones local index stays free for the dead marker
        const uint32_t mbits = lbits >= 16 ? 0 : 32 - 2 * lbits;
        const uint32_t mat_cap = mbits == 0 ? 0 : std::min<uint32_t>(1u << std::min(mbits, 12u), 2048u);
        SbMatDict mdict;
        build_material_dictionary(mdict, hb, mat_cap);
        e->mat_mode = mdict.mode;
        mat_table.swap(mdict.table);
        if (e->mat_mode == 0) lbits = 16;
        e->lbits = lbits;
        e->nmat = (uint32_t)(mat_table.size() / 6);
        c_pair.resize(tl.copy_la.size());
        sbt::parallel_ranges(c_pair.size(), 1 << 16, [&](size_t c0, size_t c1) {
            for (size_t c = c0; c < c1; c++) {
                uint32_t w = 0xFFFFFFFFu;
                if (tl.copy_slot[c] != 0xFFFFFFFFu) {
                    w = tl.copy_la[c] | (tl.copy_lb[c] << lbits);
                    if (e->mat_mode) w |= mdict.of_slot[tl.copy_slot[c]] << (2 * lbits);
                }
                c_pair[c] = w;
            }
        });
        e->lds_bytes = (size_t)tl.max_all * sizeof(float2) + (size_t)tl.max_all * sizeof(int2) + (size_t)e->nmat * 6 * sizeof(float);
        if (e->lds_bytes > 160 * 1024)
            SB_FAIL(e, SB_ERR_UNSUPPORTED, "tile needs %zu bytes of LDS (> 160 KiB): lower tile_particles or use SB_PATH_ATOMIC", e->lds_bytes);
        // SB_COLLIDE_GRID: what is left of a quarter of the CU's LDS (four workgroups per CU) behind the kernel's own arrays is
        // the area the workgroup makes its tile's neighbour lists in (sb_lists_cooperative: 12 bytes per record and cell)
        e->lds_coop_off = e->lds_coop = 0;
        if (e->opt.collision_mode == SB_COLLIDE_GRID) {
            static const bool coop_off = [] { const char *v = getenv("SB_GRID_COOP"); return v && atoi(v) == 0; }();
            const size_t quarter = 160 * 1024 / (SB_GRID_WAVES / 2), fixed = 1024 /* the kernel's static LDS */ + 512 /* allocation granule */;
            const size_t off = (e->lds_bytes + 15) & ~(size_t)15;
            if (!coop_off && off + fixed + 8 * 1024 <= quarter) {
                e->lds_coop_off = (uint32_t)off;
                e->lds_coop = (uint32_t)((quarter - fixed - off) & ~(size_t)15);
            }
        }
    } else {
        c_ia.resize(B);
        c_ib.resize(B);
        c_slot.resize(B);
        for (uint32_t s = 0; s < B; s++) {
            c_ia[s] = internal_of_slot[hb[s].a];
            c_ib[s] = internal_of_slot[hb[s].b];
            c_slot[s] = s;
            e->h_copy_of_slot[s] = s;
        }
    }
    const uint32_t nc = blockK ? e->nbeam : (uint32_t)c_slot.size();
    e->nbeam = nc;
    e->h_slot_of_copy = c_slot; // (empty with a blocked plan, which keeps its own maps: SbBlockedDev::h_beam_slot)
    e->d_live0 = nullptr;
    e->live_words = 0;
    if (!blockK) {
        float *SbBeamArrays::*fields[9] = {&SbBeamArrays::length, &SbBeamArrays::target, &SbBeamArrays::last,
                                           &SbBeamArrays::spring, &SbBeamArrays::damp,   &SbBeamArrays::yield,
                                           &SbBeamArrays::limit,  &SbBeamArrays::strain, &SbBeamArrays::stress};
        sbt::uvec<float> tmp(nc);
        for (int k = 0; k < 9; k++) {
            // the material table replaces the static parameter arrays (and the length array in mode 2)
            const bool is_static = k == 0 || (k >= 3 && k <= 6);
            if (e->path == SB_PATH_TILED && is_static && (e->mat_mode == 2 || (e->mat_mode == 1 && k != 0))) {
                e->beams.*fields[k] = nullptr;
                continue;
            }
            sbt::parallel_ranges(nc, 1 << 16, [&](size_t c0, size_t c1) {
                for (size_t c = c0; c < c1; c++) tmp[c] = c_slot[c] == 0xFFFFFFFFu ? 0.0f : hb[c_slot[c]].f[k];
            });
            SB_TRY(dev_alloc(e, &(e->beams.*fields[k]), nc));
            if (nc) SB_TRY(stage_put_bytes(e, e->beams.*fields[k], tmp.data(), (size_t)nc * 4));
        }
        SB_TRY(dev_alloc(e, &e->beams.slot, nc));
        if (nc) SB_TRY(stage_put_bytes(e, e->beams.slot, c_slot.data(), (size_t)nc * 4));
        if (e->path == SB_PATH_TILED) {
            SB_TRY(dev_alloc(e, &e->beams.pair, nc));
            if (nc) SB_TRY(stage_put_bytes(e, e->beams.pair, c_pair.data(), (size_t)nc * 4));
            SB_TRY(dev_alloc(e, &e->d_mat, mat_table.size()));
            if (!mat_table.empty())
                SB_HIP(e, hipMemcpy(e->d_mat, mat_table.data(), mat_table.size() * 4, hipMemcpyHostToDevice));
        } else {
            SB_TRY(dev_alloc(e, &e->beams.ia, nc));
            SB_TRY(dev_alloc(e, &e->beams.ib, nc));
            if (nc) SB_TRY(stage_put_bytes(e, e->beams.ia, c_ia.data(), (size_t)nc * 4));
            if (nc) SB_TRY(stage_put_bytes(e, e->beams.ib, c_ib.data(), (size_t)nc * 4));
        }
        // the one array delete passes write into, as uploaded (rewrite_scene_state puts it back)
        uint32_t *live = e->path == SB_PATH_TILED ? e->beams.pair : e->beams.ia;
        SB_TRY(dev_alloc(e, &e->d_live0, nc));
        if (nc) SB_HIP(e, hipMemcpyAsync(e->d_live0, live, (size_t)nc * 4, hipMemcpyDeviceToDevice, e->stream));
        e->live_words = nc;
    }
    tm.mark("beams to device");
    // ---- spatial hash: covers the uploaded bounding box plus a margin; particles that later
    // leave it are clamped into edge cells (still a superset of the contacts, sb_physics.h)
    e->grid = SbGrid{};
    e->ncell = 0;
    e->d_grid_ctl = nullptr;
    if (e->opt.collision_mode == SB_COLLIDE_GRID) {
        float minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
        for (uint32_t s = 0; s < P; s++) {
            if (!std::isfinite(px[s]) || !std::isfinite(py[s])) continue;
            minx = std::min(minx, px[s]); maxx = std::max(maxx, px[s]);
            miny = std::min(miny, py[s]); maxy = std::max(maxy, py[s]);
        }
        if (!(minx <= maxx)) minx = maxx = miny = maxy = 0.f;
        // skin: how far a particle may drift before the hash is rebuilt (SbGridCtl, sb_physics.h)
        const float skin = e->opt.grid_skin < 0.f ? 0.f : (e->opt.grid_skin > 0.f ? e->opt.grid_skin : 0.4f * e->prm.particle_radius);
        float cell = e->prm.particle_radius * 2.0f * 1.015625f + 2.0f * skin;
        const float S = e->prm.bounds_size;
        float mx = std::max(0.25f * (maxx - minx), 16.f * cell), my = std::max(0.25f * (maxy - miny), 16.f * cell);
        float x0 = std::max(0.f, minx - mx), x1 = std::min(S, maxx + mx);
        float y0 = std::max(0.f, miny - my), y1 = std::min(S, maxy + my);
        if (!(x1 > x0)) x1 = x0 + cell;
        if (!(y1 > y0)) y1 = y0 + cell;
        // at most 16384 cells per side and 2^27 cells in all: coarser cells are still a valid broad phase
        for (;;) {
            double nxd = std::ceil((double)(x1 - x0) / cell), nyd = std::ceil((double)(y1 - y0) / cell);
            if (nxd <= 16384.0 && nyd <= 16384.0 && nxd * nyd <= 134217728.0) break;
            cell *= 1.25f;
        }
        e->grid.x0 = x0;
        e->grid.y0 = y0;
        e->grid.width = x1 - x0;
        e->grid.height = y1 - y0;
        e->grid.two_r = e->prm.particle_radius * 2.0f;
        e->grid.cell_min = cell;
        e->grid.nx_cap = std::max(1u, (uint32_t)std::ceil((double)(x1 - x0) / cell));
        e->grid.ny_cap = std::max(1u, (uint32_t)std::ceil((double)(y1 - y0) / cell));
        e->ncell = e->grid.nx_cap * e->grid.ny_cap; // capacity: the skin only ever makes cells larger than cell_min
        e->grid.bounds = S;
        e->grid.wide_side = std::max(1u, (uint32_t)std::floor(std::sqrt((double)e->ncell)));
        const size_t n1 = (size_t)e->ncell + 1;
        for (int k = 0; k < 2; k++) { // two hash buffers (SbGrid: the lagged schedule pushes the next one while the current one is still scanned)
            SB_TRY(dev_alloc(e, &e->d_head[k], n1));
            SB_HIP(e, hipMemset(e->d_head[k], 0, n1 * 8)); // build number 0: "never written" (the first build is number 1)
            SB_TRY(dev_alloc(e, &e->d_rec[k], P));
            SB_TRY(dev_alloc(e, &e->d_cell_of[k], P));
        }
        SB_TRY(dev_alloc(e, &e->d_grid_ctl, 2));
        SB_TRY(dev_alloc(e, &e->d_blk_max, 3 * SB_GRID_SLOTS * 4)); // float4[3][SB_GRID_SLOTS]: max | sample dx | sample dy | -
        SB_HIP(e, hipMemset(e->d_blk_max, 0, 3 * SB_GRID_SLOTS * 16));
        SB_TRY(dev_alloc(e, &e->d_grid_outside, 4));
        SB_HIP(e, hipMemset(e->d_grid_outside, 0, 16));
        SB_TRY(dev_alloc(e, &e->d_grid_nonempty, 1 + 1024)); // (the hybrid look's answer and the per-workgroup minima behind it)
        SbGridCtl ctl[2] = {};
        for (int k = 0; k < 2; k++) {
            ctl[k].skin_min = skin;
            // grid_skin given explicitly: that skin, fixed.  Default: adaptive between 0.4 r and 1.6 r (SbGridCtl).
            ctl[k].skin_max = e->opt.grid_skin > 0.f ? skin : 4.0f * skin;
            ctl[k].geo.skin = skin;
            ctl[k].geo.cell = cell;
            ctl[k].geo.nx = e->grid.nx_cap;
            ctl[k].geo.ny = e->grid.ny_cap;
            ctl[k].geo.x0 = x0;
            ctl[k].geo.y0 = y0;
            ctl[k].geo.wide = 0;
            const float reach = e->grid.two_r + 2.0f * skin;
            ctl[k].geo.reach2 = reach * reach * 1.001f;
            ctl[k].pgeo = ctl[k].geo;
            ctl[k].since = 1000; // "the hash before the first one lasted long": start lean
            ctl[k].executed = sb_grid_executed0();
        }
        SB_HIP(e, hipMemcpy(e->d_grid_ctl, ctl, sizeof ctl, hipMemcpyHostToDevice));
        memcpy(e->grid_ctl0, ctl, sizeof ctl);
        e->grid_heads = n1;
        e->grid_slots = 3 * SB_GRID_SLOTS * 4;
        e->grid_par = 0;
        e->grid_force = true; // no hash yet: the first substep starts with a forced helper launch
        e->grid_classic_left = e->grid_classic_chunk = e->grid_calm = 0;
        e->grid_executed = sb_grid_executed0();
        e->grid.head = e->d_head[0];
        e->grid.rec = e->d_rec[0];
        e->grid.cell_of = e->d_cell_of[0];
        e->grid.head1 = e->d_head[1];
        e->grid.rec1 = e->d_rec[1];
        e->grid.cell_of1 = e->d_cell_of[1];
        SB_TRY(dev_alloc(e, &e->d_nl_count, P));
        SB_TRY(dev_alloc(e, &e->d_nl, (size_t)SB_NL_CAP * std::max<size_t>(P, 1)));
        SB_HIP(e, hipMemset(e->d_nl_count, 0, std::max<size_t>(P, 1) * 4));
        e->grid.nl_count = e->d_nl_count;
        e->grid.nl = e->d_nl;
        e->grid.nl_stride = P;
    }
    tm.mark("spatial hash arrays");
    // ---- accumulators and masks, zeroed (engineWorker.ts:591-592)
    if (e->path == SB_PATH_ATOMIC) {
        SB_TRY(dev_alloc(e, &e->d_forces, P));
        SB_HIP(e, hipMemset(e->d_forces, 0, std::max<size_t>(P, 1) * sizeof(int2)));
    }
    SB_TRY(dev_alloc(e, &e->d_broken, (nc + 31) / 32));
    SB_HIP(e, hipMemset(e->d_broken, 0, std::max<size_t>((nc + 31) / 32, 1) * 4));
    if (e->bk.K) e->bk.d_broken = e->d_broken;
    // ---- SB_COLLIDE_GRID: a blocked plan BESIDE the tiled layout, on the same bisection, for the stretches of a run in which
    // nothing is within reach of anything (every neighbour list empty: the collision loop of compute.wgsl:142-170, which the
    // reference always runs, is then a no-op and K substeps can go out of LDS and registers as with collisions off).
    // hybrid_substeps below decides substep run by substep run; every other entry point only ever sees the tiled layout.
    e->hy = SbBlockedDev{};
    e->h_tile_p0.assign(tl.tile_p0.begin(), tl.tile_p0.end());
    SB_TRY(dev_alloc(e, &e->d_dead_gen, B));
    SB_HIP(e, hipMemset(e->d_dead_gen, 0, std::max<size_t>(B, 1) * 4));
    e->delete_gen = 0;
    memcpy(&e->consts, md + 48, sizeof(SbConsts));
    SB_HIP(e, hipDeviceSynchronize());
    e->h_beams.swap(hb);
    e->loaded = true;
    pool_trim(e);
    tm.mark("masks + final sync");
    {
        static const bool hybrid_off = [] { const char *v = getenv("SB_HYBRID"); return v && atoi(v) == 0; }();
        if (!hybrid_off && e->path == SB_PATH_TILED && e->opt.collision_mode == SB_COLLIDE_GRID && e->opt.block_substeps != 1 && P && B &&
            tl.ntiles) {
            auto *p = new SbHybridPending;
            p->px = px;
            p->py = py;
            p->target = tile_target_used;
            p->depth = std::min<uint32_t>(e->opt.block_substeps ? e->opt.block_substeps : SB_BK_KPLAN, SB_BK_KMAX);
            const SbHostBeams *beams = &e->h_beams;
            p->th = std::thread([p, beams] { sb_build_blocking(p->bl, p->px, p->py, *beams, p->target, p->depth); });
            e->hy_pending = p;
        }
    }
    {
        auto *trash = new SbUploadTrash;
        trash->px.swap(px);
        trash->py.swap(py);
        std::swap(trash->tl, tl);
        std::swap(trash->bl, bl);
        trash->hb.swap(hb); // (the previous scene's records)
        std::vector<uint32_t> *big[8] = {&slot_index, &internal_of_index, &order, &internal_of_slot, &c_ia, &c_ib, &c_pair, &c_slot};
        for (int k = 0; k < 8; k++) trash->v[k].swap(*big[k]);
        e->reaper = std::thread([trash] { delete trash; });
    }
    return SB_OK;
}

sb_status sb_write_user_input(sb_engine *e, const void *bytes32)
{
    if (!e || !bytes32) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_write_user_input before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    memcpy(e->h_metadata.data() + SB_USER_INPUT_OFFSET, bytes32, SB_USER_INPUT_BYTES);
    memcpy((uint8_t *)&e->consts + 32, bytes32, SB_USER_INPUT_BYTES); // rides in the kernarg of later launches
    return SB_OK;
}

sb_status sb_set_physics_constants(sb_engine *e, const float c8[8])
{
    if (!e || !c8) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_set_physics_constants before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    memcpy(e->h_metadata.data() + 48, c8, 32);
    memcpy(&e->consts, c8, 32);
    return SB_OK;
}

sb_status sb_get_physics_constants(sb_engine *e, float c8[8])
{
    if (!e || !c8) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_get_physics_constants before sb_write_buffers");
    memcpy(c8, e->h_metadata.data() + 48, 32);
    return SB_OK;
}

// m substeps, one launch each, with the collision loop served by the spatial hash (SbGridCtl, sb_physics.h).  In the lagged
// schedule nothing but the substep kernels is launched; what the host owes that schedule is one look when the launches have
// been issued: if a substep moved somebody farther than its predecessor predicted, the tail of that launch -- itself still a
// correct substep -- raised `abort` and every launch behind it returned at once.  The host then takes the buffers back to what
// the device really did, orders a hash of the current state from the helper launch, and runs a stretch in the classic schedule
// (helper launch in front of every substep: violent scenes cost what they cost in r03), which doubles with every abort and
// decays while the scene stays calm.
// an abort has been seen (grid_substeps, or the hybrid's look): a hash of the current state from the helper, then a stretch in
// the classic schedule that doubles with every abort
static void grid_recover(sb_engine *e)
{
    e->grid_aborts++;
    e->grid_calm = 0;
    e->grid_force = true; // (the forced helper launch publishes a state without the flag)
    e->grid_classic_chunk = std::min(std::max(16u, 2u * e->grid_classic_chunk), 1024u);
    e->grid_classic_left = e->grid_classic_chunk;
}

static sb_status grid_substeps(sb_engine *e, uint32_t m, bool aux_on_last)
{
    if (!(e->opt.collision_mode == SB_COLLIDE_GRID && e->P && e->d_grid_ctl)) {
        for (uint32_t i = 0; i < m; i++) sbk_launch_substep(e, aux_on_last && i + 1 == m);
        return SB_OK;
    }
    while (m) {
        const uint32_t sched = sbk_grid_mode(e);
        const bool stretch = sched == SB_GRID_CLASSIC && e->grid_classic_left != 0u;
        const uint32_t chunk = stretch ? std::min(m, e->grid_classic_left) : m;
        const uint64_t exec0 = e->grid_executed;
        for (uint32_t i = 0; i < chunk; i++) sbk_launch_substep(e, aux_on_last && i + 1 == m);
        SbGridCtl *pin = (SbGridCtl *)(e->dev_err + 96); // (pinned; hybrid_substeps uses words 16 .. 95)
        if (sched == SB_GRID_CLASSIC) { // (its decisions never abort: the helper serves whatever they order)
            e->grid_classic_substeps += chunk;
            m -= chunk;
            if (!stretch) continue; // (the atomic path, SB_GRID_MODE=classic: always)
            e->grid_classic_left -= chunk;
            if (e->grid_classic_left == 0u) { // end of a stretch: is the scene still one that wears a hash out in a few substeps?
                SB_HIP(e, hipMemcpyAsync(pin, e->d_grid_ctl + e->grid_par, sizeof(SbGridCtl), hipMemcpyDeviceToHost, e->stream));
                SB_HIP(e, sb_stream_wait(e->stream));
                if (pin[0].short_lived != 0u) {
                    e->grid_classic_chunk = std::min(2u * e->grid_classic_chunk, 1024u);
                    e->grid_classic_left = e->grid_classic_chunk;
                }
            }
            continue;
        }
        SB_HIP(e, hipMemcpyAsync(pin, e->d_grid_ctl, 2 * sizeof(SbGridCtl), hipMemcpyDeviceToHost, e->stream));
        SB_HIP(e, sb_stream_wait(e->stream, std::min<int64_t>(400 + (int64_t)chunk * (int64_t)(10u + e->P / 25000u), 200000)));
        if (!(pin[0].abort | pin[1].abort)) {
            m -= chunk;
            if (pin[e->grid_par].short_lived != 0u) { // hashes last four substeps or less: the classic schedule serves such a scene better
                e->grid_classic_chunk = std::max(64u, e->grid_classic_chunk);
                e->grid_classic_left = e->grid_classic_chunk;
                e->grid_calm = 0;
                continue;
            }
            e->grid_calm += chunk;
            if (e->grid_calm >= 1024u && e->grid_classic_chunk) { // calm for a while: the next abort starts with a shorter stretch
                e->grid_classic_chunk /= 2u;
                e->grid_calm = 0;
            }
            continue;
        }
        // the launches that ran are the ones that counted themselves; the rest returned at once
        // (the device counts in 32 bits: how far the two blocks are AHEAD of where this call started, wrap or no wrap)
        const int32_t ahead = std::max((int32_t)(pin[0].executed - (uint32_t)exec0), (int32_t)(pin[1].executed - (uint32_t)exec0));
        const uint32_t ran = std::min<uint32_t>((uint32_t)std::max(ahead, 0), chunk), undone = chunk - ran;
        static const bool debug = getenv("SB_GRID_DEBUG") != nullptr;
        if (debug) fprintf(stderr, "[sb grid] abort after %u of %u substeps (classic stretch %u)\n", ran, chunk, std::max(16u, 2u * e->grid_classic_chunk));
        e->cur ^= undone & 1u;
        e->grid_par ^= undone & 1u;
        e->substeps_done -= undone;
        e->grid_executed = exec0 + ran;
        grid_recover(e);
        m -= ran;
    }
    return SB_OK;
}

// n substeps.  strain/stress are pure outputs (render inputs in the reference, render.wgsl:82): only the last
// substep before control returns to the caller can ever be observed, so only it stores them.
// SB_COLLIDE_GRID with a blocked plan beside the tiling (e->hy).  The reference always runs its collision loop
// (compute.wgsl:142-170); while every neighbour list of the spatial hash is empty and the hash's displacement bound stays
// within its skin that loop is a no-op, and K substeps can go out of LDS and registers exactly as with collisions off.  So:
//   look   (stream sync + a few words back): lists all empty? no rebuild pending? at least half the skin left?
//   no  -> substep by substep for a stretch that doubles while the answer stays no (a pile never pays more than a few looks),
//          then look again;
//   yes -> beam state into the blocked layout, a run of tracked launches (k_substep_blocked<TRACK>: the launch measures what
//          its particles move, the launch behind it -- k_hybrid_validate behind the last one -- adds that to the bound in its
//          prologue; one over the skin raises a flag on the device and everything queued behind returns at once, having
//          written nothing), state back into the tiled layout, the hash's
//          bookkeeping (bound, drift, age) brought up to date; a launch that went over is simply not counted -- the buffers it
//          read are intact (everything is double-buffered) -- and its substeps are redone one by one, where the hash gets rebuilt.
// Every other entry point sees the tiled layout only.  Bit-exact by construction: a launch counts only if no contact could
// have happened during it.
// the side thread's plan goes to the device (first quiet look); afterwards e->hy.K says whether the scene can use it
static sb_status hybrid_materialise(sb_engine *e)
{
    SbHybridPending *p = e->hy_pending;
    if (!p) return SB_OK;
    if (p->th.joinable()) p->th.join();
    const SbBlocking &hbl = p->bl;
    uint32_t hk = p->depth;
    auto fits = [&](uint32_t d) {
        return hbl.max_own <= SB_BK_OWNP * SB_BK_T && hbl.halo_at[d] <= SB_BK_HALOP * SB_BK_T && hbl.max_ownb <= SB_BK_OWNB * SB_BK_T &&
               hbl.halo_entries_at[d] <= SB_BK_HALOB * SB_BK_T && hbl.region_at[d] <= (1u << SB_BK_LBITS) - 2u;
    };
    if (!fits(hk)) { // (launches of k substeps load the depth-k prefix of the plan: they just stay shallower)
        uint32_t fit = 0;
        for (uint32_t d = 2; d < hk; d++)
            if (fits(d)) fit = d;
        hk = fit;
    }
    sb_status st = SB_OK;
    // the two plans must agree on the particle order and on the tiles (same bisection of the same positions)
    if (hk && hbl.order == e->h_pslot && hbl.tile_p0 == e->h_tile_p0) {
        SbStageTimer tm;
        st = upload_blocked(e, hbl, e->h_beams, hk, tm, true, &e->h_copy_of_slot, &e->h_slot_of_copy);
        if (st != SB_OK || !hk) e->hy = SbBlockedDev{}; // (a plan that is not whole is no plan: the engine stays on single substeps)
        tm.mark("blocked plan beside the tiling");
    }
    delete p;
    e->hy_pending = nullptr;
    return st;
}

static sb_status hybrid_substeps(sb_engine *e, uint32_t n)
{
    SbBlockedDev &h = e->hy;
    static const uint32_t fail_every = [] { const char *v = getenv("SB_HYBRID_FAIL_EVERY"); return v ? (uint32_t)atoi(v) : 0u; }();
    while (n) {
        if (h.slow_left || n < 2u) {
            const uint32_t m = h.slow_left ? std::min(n, h.slow_left) : n;
            SB_TRY(grid_substeps(e, m, m == n));
            h.slow_left -= std::min(h.slow_left, m);
            n -= m;
            continue;
        }
        // ---- look
        struct Look { SbGridCtl ctl; float min_d2; } *look = (Look *)(e->dev_err + 16); // (pinned; words 0..15 are the error and stamp words)
        sbk_launch_grid_settle(e); // the next substep's decision ahead of time: the bound then covers the substep just done
        SB_HIP(e, hipMemcpyAsync(&look->ctl, e->d_grid_ctl + e->grid_par, sizeof(SbGridCtl), hipMemcpyDeviceToHost, e->stream));
        sbk_launch_lists_min_d2(e); // (asked only here, by kernels of its own)
        SB_HIP(e, hipMemcpyAsync(&look->min_d2, e->d_grid_nonempty, 4, hipMemcpyDeviceToHost, e->stream));
        SB_HIP(e, sb_stream_wait(e->stream));
        const SbGridCtl ctl = look->ctl;
        auto force_rebuild = [&]() -> sb_status { // the next substep starts with a forced helper launch and makes the lists
            e->grid_force = true;
            h.slow_left = std::min<uint32_t>(n, 1u);
            return SB_OK;
        };
        if (ctl.abort != 0u) { // (the settling launch found the lists not known to be valid any more: as in grid_substeps)
            grid_recover(e);
            h.slow_left = std::min<uint32_t>(n, 1u);
            continue;
        }
        if (e->grid_force || ctl.builds == 0u || ctl.fresh != 0u || ctl.need_build != 0u || ctl.pushing != 0u) {
            h.slow_left = std::min<uint32_t>(n, ctl.pushing ? 2u : 1u); // no hash yet, or lists on order: a substep or two make them
            continue;
        }
        const float skin = ctl.geo.skin;
        // Nobody within 2r + 2 skin of anybody (every list empty): no contact while the hash's bound stays inside its skin.
        // Somebody listed, the closest such pair d apart: two particles approach each other by at most twice the displacement
        // bound (it is measured against a common drift), so no LISTED pair touches either while the bound grows by less than
        // gap = (d - 2r) / 2 from here -- the run's budget is the smaller of the two.  A pair AT 2r or closer: not quiet.
        float gap = INFINITY;
        if (look->min_d2 < INFINITY) gap = 0.5f * ((float)(std::sqrt((double)look->min_d2) * (1.0 - 1.0e-6)) - e->grid.two_r);
        if (!(gap > 0.0f)) gap = 0.0f; // (NaN, or in contact)
        const float budget = std::min(skin - ctl.accum, gap);
        static const bool debug = getenv("SB_HYBRID_DEBUG") != nullptr;
        if (debug)
            fprintf(stderr, "[sb hybrid] look: n %u builds %u accum %g skin %g since %u closest listed pair %g gap %g budget %g K %u pending %d\n", n,
                    ctl.builds, ctl.accum, skin, ctl.since, std::sqrt((double)look->min_d2), gap, budget, h.K, e->hy_pending ? 1 : 0);
        // (an engine with ghost zones runs blocked too since r04: a call never spans a refresh -- the exchanger steps up to the
        // next one -- and everything a refresh touches is the tiled layout, which the run starts from and ends in; the forced hash
        // a refresh orders costs the call one single substep in front of its run)
        if (!(gap >= 0.15f * skin)) { // somebody (nearly) touching
            h.slow_chunk = std::min<uint32_t>(std::max<uint32_t>(16u, 2u * h.slow_chunk), 1024u);
            h.slow_left = std::min(n, h.slow_chunk);
            continue;
        }
        if (!(budget >= 0.15f * skin)) { // quiet, but little of the skin left: a fresh hash is cheaper than a run that fails
            SB_TRY(force_rebuild());
            continue;
        }
        // How long will the budget last?  The last tracked run measured how fast the bound grows (h.rate, per substep; the
        // hash's own accum / since is no guide right after a build, while its drift estimate is still settling).  A run costs a
        // dozen launches' worth of fixed work (two conversions, two syncs, the rebuild that follows), so one that the budget
        // would end within 32 substeps is not started -- two blobs flying at each other use the skin up in ten -- and a longer
        // one is cut to 0.8 of the prediction, so that it ends with its last launch validated rather than with one refused.
        // An old measurement fades (halved at every look that turns a run down), so a scene that calms down is tried again; a run
        // that ends in a refusal after three validated launches has still paid for itself (a refusal costs about nine single
        // substeps' time, a blocked substep saves half of one).
        const float lasts = h.rate > 0.0f ? budget / h.rate : 1.0e9f;
        if (!(lasts >= 32.0f)) {
            h.rate *= 0.5f;
            h.slow_chunk = std::min<uint32_t>(std::max<uint32_t>(16u, 2u * h.slow_chunk), 1024u);
            h.slow_left = std::min(n, h.slow_chunk);
            continue;
        }
        h.slow_chunk = 0;
        if (!h.K) { // quiet for the first time: now the plan is worth having on the device
            SB_TRY(hybrid_materialise(e));
            if (!h.K) { // (the scene cannot use one after all: more material rows than an entry word holds, a negative yield ...)
                return grid_substeps(e, n, true);
            }
        }
        // ---- a run of tracked launches
        // (48 launches a run at most: a run costs a look, two conversions and three waits whatever its length -- half a microsecond per
        // substep of a 1 M-particle scene at 48 -- but runs of up to 240 launches measured SLOWER on the quiet lattice, 13.5 against
        // 13.1 - 13.2 us per substep: r04)
        const uint32_t chunk = std::min<uint32_t>(std::min<uint32_t>(n, 48u * h.K), (uint32_t)std::min(0.8f * lasts, 1.0e6f));
        uint32_t ks[64], count = 0, k_hi = 0, n_hi = 0;
        const uint32_t L = sbk_split_call(chunk, h.K, false, &k_hi, &n_hi);
        for (uint32_t i = 0; i < L && count < 64u; i++) ks[count++] = i < n_hi ? k_hi : k_hi - 1u;
        uint32_t planned = 0;
        for (uint32_t i = 0; i < count; i++) planned += ks[i];
        const bool aux_last = planned == n;
        SbHybridCtl q{};
        q.D = ctl.accum;
        q.Cx = ctl.Cx;
        q.Cy = ctl.Cy;
        q.cx = ctl.cx;
        q.cy = ctl.cy;
        q.skin = std::min(skin, ctl.accum + gap); // (the validation's limit for the bound: the skin, or the gap of the closest listed pair)
        q.fail_at = 0xFFFFFFFFu;
        if (fail_every && (h.launches_ok + h.launches_failed + count) / fail_every != (h.launches_ok + h.launches_failed) / fail_every)
            q.fail_at = fail_every - 1u - (uint32_t)((h.launches_ok + h.launches_failed) % fail_every); // (tests: a roll-back every so many launches)
        SbHybridCtl *pin = (SbHybridCtl *)(e->dev_err + 64);
        *pin = q;
        SB_HIP(e, hipMemcpyAsync(h.d_q + h.qpar, pin, sizeof q, hipMemcpyHostToDevice, e->stream)); // (the block the run's first launch reads)
        sbk_hybrid_to_blocked(e);
        const uint32_t cur0 = e->cur, bcur0 = h.cur;
        const uint64_t done0 = e->substeps_done;
        sbk_hybrid_launch(e, ks, count, aux_last);
        SB_HIP(e, hipMemcpyAsync(pin, h.d_q + h.qpar, sizeof q, hipMemcpyDeviceToHost, e->stream)); // (... and the one its last validation wrote)
        // (the run is a few hundred microseconds to a few milliseconds of kernels, known in advance: poll for twice that)
        SB_HIP(e, sb_stream_wait(e->stream, std::min<int64_t>(400 + (int64_t)planned * (int64_t)(20u + e->P / 25000u), 200000)));
        q = *pin;
        const uint32_t done = std::min(q.done, count);
        // the host's idea of the buffers follows what the device really did
        e->cur = cur0 ^ (done & 1u);
        h.cur = bcur0 ^ (done & 1u);
        e->substeps_done = done0 + q.substeps;
        sbk_hybrid_to_tiled(e, done == count && aux_last);
        h.launches_ok += done;
        h.substeps_blocked += q.substeps;
        if (q.substeps) h.rate = std::max(q.D - ctl.accum, 0.0f) / (float)q.substeps;
        // the hash's bookkeeping, as if the tails of single substeps had kept it all along
        SbGridCtl upd = ctl;
        upd.accum = q.D;
        upd.Cx = q.Cx;
        upd.Cy = q.Cy;
        upd.cx = q.cx;
        upd.cy = q.cy;
        upd.since = ctl.since + q.substeps;
        upd.settled = 1u; // (the next substep adopts this block as it stands)
        look->ctl = upd;
        SB_HIP(e, hipMemcpyAsync(e->d_grid_ctl + e->grid_par, &look->ctl, sizeof(SbGridCtl), hipMemcpyHostToDevice, e->stream));
        SB_HIP(e, sb_stream_wait(e->stream)); // (`look` is reused by the next look)
        if (!(upd.accum + 2.0f * h.rate <= skin)) e->grid_force = true; // the run used the skin up: single substeps start on a fresh hash
        n -= q.substeps;
        if (done < count) { // over the budget (or told to fail, by a test): a fresh hash, then look again -- after a stretch of
            h.launches_failed += 1; // single substeps that doubles with every refusal in a row (a scene that keeps using its budget up
            if (done >= 2u) h.fail_streak = 0;          // (a run that got somewhere and then met its budget: the ordinary end of a run)
            else if (done != q.fail_at) h.fail_streak++; // within a launch or two is cheaper substep by substep; a test's refusals do not count)
            SB_TRY(force_rebuild());
            if (h.fail_streak > 1u) h.slow_left = std::min<uint32_t>(n, 8u << std::min<uint32_t>(h.fail_streak, 7u));
        } else {
            h.fail_streak = 0;
        }
    }
    return SB_OK;
}

static sb_status launch_substeps(sb_engine *e, uint32_t n)
{
    if (e->bk.K) sbk_launch_blocked(e, n, true);
    else if (e->hy.K || e->hy_pending) return hybrid_substeps(e, n);
    else return grid_substeps(e, n, true);
    return SB_OK;
}

sb_status sb_step(sb_engine *e, uint32_t n)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_step before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    SB_TRY(launch_substeps(e, n));
    SB_HIP(e, hipGetLastError());
    return SB_OK;
}

sb_status sb_delete_pass(sb_engine *e)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_delete_pass before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    sbk_launch_halo_clear_ghost_flags(e); // (an engine with a halo: ghost copies die on their owner's word, sb_halo_delete_ghosts)
    if (e->bk.K) sbk_launch_delete_blocked(e);
    else sbk_launch_delete(e);
    SB_HIP(e, hipGetLastError());
    return SB_OK;
}

sb_status sb_halo_delete_ghosts(sb_engine *e)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_halo_delete_ghosts before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    if (e->bk.K) sbk_launch_delete_blocked(e);
    else sbk_launch_delete(e);
    SB_HIP(e, hipGetLastError());
    return SB_OK;
}

sb_status sb_frame(sb_engine *e)
{
    if (!e) return SB_ERR_INVALID;
    SB_TRY(sb_step(e, e->subticks)); // engineWorker.ts:655-661
    return sb_delete_pass(e);       // engineWorker.ts:663-664
}

sb_status sb_sync(sb_engine *e)
{
    if (!e) return SB_ERR_INVALID;
    SB_HIP(e, hipSetDevice(e->device));
    SB_HIP(e, sb_stream_wait(e->stream));
    if (e->dev_err && *e->dev_err) {
        const uint32_t what = *e->dev_err;
        *e->dev_err = 0;
        SB_FAIL(e, SB_ERR_HIP, "peer exchange: neighbour(s) 0x%x did not signal within %u ms", what, e->peer_timeout_ms);
    }
    return SB_OK;
}

sb_status sb_step_timed(sb_engine *e, uint32_t n, float *ms)
{
    if (!e || !ms) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_step_timed before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    SB_HIP(e, hipEventRecord(e->ev0, e->stream));
    SB_TRY(launch_substeps(e, n));
    SB_HIP(e, hipGetLastError());
    SB_HIP(e, hipEventRecord(e->ev1, e->stream));
    SB_HIP(e, sb_event_wait(e->ev1));
    SB_HIP(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
    return SB_OK;
}

sb_status sb_mark(sb_engine *e, uint32_t slot)
{
    if (!e) return SB_ERR_INVALID;
    if (slot >= SB_MAX_MARKS) SB_FAIL(e, SB_ERR_INVALID, "sb_mark: slot %u >= %d", slot, SB_MAX_MARKS);
    SB_HIP(e, hipSetDevice(e->device));
    if (e->marks.size() <= slot) e->marks.resize((size_t)slot + 1, nullptr);
    if (!e->marks[slot]) SB_HIP(e, hipEventCreate(&e->marks[slot]));
    SB_HIP(e, hipEventRecord(e->marks[slot], e->stream));
    return SB_OK;
}

sb_status sb_mark_elapsed(sb_engine *e, uint32_t a, uint32_t b, float *ms)
{
    if (!e || !ms) return SB_ERR_INVALID;
    if (a >= e->marks.size() || b >= e->marks.size() || !e->marks[a] || !e->marks[b])
        SB_FAIL(e, SB_ERR_STATE, "sb_mark_elapsed: mark %u or %u was never recorded", a, b);
    SB_HIP(e, hipSetDevice(e->device));
    SB_HIP(e, sb_event_wait(e->marks[b]));
    SB_HIP(e, hipEventElapsedTime(ms, e->marks[a], e->marks[b]));
    return SB_OK;
}

// per-slot delete generation -> host; returns the number of dead slots
static sb_status fetch_dead(sb_engine *e, sbt::uvec<uint32_t> &dead, uint32_t *ndead)
{
    *ndead = 0;
    if (!(e->B && e->delete_gen)) { // no delete pass has run since the upload: nobody is dead (and nobody needs the array)
        dead.clear();
        return SB_OK;
    }
    dead.resize(e->B);
    SB_TRY(stage_get_bytes(e, dead.data(), e->d_dead_gen, (size_t)e->B * 4));
    uint32_t n = 0;
    for (uint32_t g : dead) n += g != 0;
    *ndead = n;
    return SB_OK;
}

sb_status sb_get_counts(sb_engine *e, uint32_t *particles, uint32_t *beams)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_get_counts before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    SB_HIP(e, hipStreamSynchronize(e->stream));
    sbt::uvec<uint32_t> dead;
    uint32_t nd = 0;
    SB_TRY(fetch_dead(e, dead, &nd));
    if (particles) *particles = e->P;
    if (beams) *beams = e->B - nd;
    return SB_OK;
}

static sb_status sb_load_buffers_impl(sb_engine *e, void *metadata, size_t metadata_bytes, void *mapping, size_t mapping_bytes,
                          void *particles, size_t particles_bytes, void *beams, size_t beams_bytes)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_load_buffers before sb_write_buffers");
    const uint32_t maxP = e->opt.max_particles, maxB = e->opt.max_beams, P = e->P, B = e->B;
    const uint32_t bstride = beam_stride(e);
    if (metadata && metadata_bytes < SB_METADATA_BYTES) SB_FAIL(e, SB_ERR_INVALID, "metadata buffer too small");
    if (mapping && mapping_bytes < (size_t)(maxP + (size_t)maxB) * map_isz(e)) SB_FAIL(e, SB_ERR_INVALID, "mapping buffer too small");
    if (particles && particles_bytes < (size_t)maxP * SB_PARTICLE_STRIDE) SB_FAIL(e, SB_ERR_INVALID, "particle buffer too small");
    if (beams && beams_bytes < (size_t)maxB * bstride) SB_FAIL(e, SB_ERR_INVALID, "beam buffer too small");
    SB_HIP(e, hipSetDevice(e->device));
    SB_HIP(e, hipStreamSynchronize(e->stream)); // engineWorker.ts:554
    SbStageTimer tm;

    sbt::uvec<uint32_t> dead;
    uint32_t nd = 0;
    SB_TRY(fetch_dead(e, dead, &nd));

    if (metadata) {
        memcpy(metadata, e->h_metadata.data(), SB_METADATA_BYTES);
        uint32_t bc = B - nd;
        memcpy((uint8_t *)metadata + 24, &bc, 4); // metadata.beam_i_c, compute.wgsl:238
    }
    if (mapping) {
        // replay the per-frame stable in-place compactions of the beam slots, pass by pass
        // (compute.wgsl:221-233 intent); slots past the live count keep stale values
        // (over the CALLER's slots, those of the latest upload: beams an upload removed are in none of them -- sb_engine.h h_user_slot)
        uint8_t *m = (uint8_t *)mapping;
        memcpy(m, e->h_mapping.data(), e->h_mapping.size());
        if (nd) {
            const uint32_t Bu = sb_user_beams(e);
            std::vector<uint32_t> gens;
            for (uint32_t u = 0; u < Bu; u++)
                if (dead[sb_user_slot(e, u)]) gens.push_back(dead[sb_user_slot(e, u)]);
            std::sort(gens.begin(), gens.end());
            gens.erase(std::unique(gens.begin(), gens.end()), gens.end());
            std::vector<uint32_t> orig(Bu);
            for (uint32_t u = 0; u < Bu; u++) orig[u] = sb_user_slot(e, u);
            uint32_t count = Bu;
            for (uint32_t g : gens) {
                uint32_t w = 0;
                for (uint32_t s = 0; s < count; s++)
                    if (dead[orig[s]] != g) {
                        if (w != s) {
                            map_set(e, m, (size_t)maxP + w, map_get(e, m, (size_t)maxP + s));
                            orig[w] = orig[s];
                        }
                        w++;
                    }
                count = w;
            }
        }
    }
    tm.mark("readback: metadata + mapping");
    if (particles && P) {
        sbt::uvec<float2> hp(P), hv(P), ha(P);
        const SbParticleArrays &c = e->part[e->cur];
        SB_TRY(stage_get_bytes(e, hp.data(), c.pos, P * sizeof(float2)));
        SB_TRY(stage_get_bytes(e, hv.data(), c.vel, P * sizeof(float2)));
        SB_TRY(stage_get_bytes(e, ha.data(), c.acc, P * sizeof(float2)));
        uint8_t *out = (uint8_t *)particles;
        sbt::parallel_ranges(P, 1 << 16, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                float q[6] = {hp[i].x, hp[i].y, hv[i].x, hv[i].y, ha[i].x, ha[i].y};
                memcpy(out + (size_t)e->h_pidx[i] * SB_PARTICLE_STRIDE, q, SB_PARTICLE_STRIDE);
            }
        });
    }
    tm.mark("readback: particles");
    if (beams && B) {
        const uint32_t nc = e->nbeam;
        sbt::uvec<float> t(nc), l(nc), sn(nc), ss(nc);
        SB_TRY(stage_get_bytes(e, t.data(), e->beams.target, (size_t)nc * 4));
        SB_TRY(stage_get_bytes(e, l.data(), e->beams.last, (size_t)nc * 4));
        SB_TRY(stage_get_bytes(e, sn.data(), e->beams.strain, (size_t)nc * 4));
        SB_TRY(stage_get_bytes(e, ss.data(), e->beams.stress, (size_t)nc * 4));
        tm.mark("readback: beam state to host");
        uint8_t *out = (uint8_t *)beams;
        const size_t foff = e->opt.layout == SB_LAYOUT_V1 ? 4 : 8;
        sbt::parallel_ranges(sb_user_beams(e), 1 << 16, [&](size_t s0, size_t s1) {
        for (size_t u = s0; u < s1; u++) {
            // every slot that was active at upload is written, dead ones with their last state
            const uint32_t s = sb_user_slot(e, u);
            uint32_t c = e->h_copy_of_slot[s];
            uint32_t idx = map_get(e, e->h_mapping.data(), (size_t)maxP + u);
            uint8_t *rec = out + (size_t)idx * bstride, *f = rec + foff;
            const SbHostBeam &h = e->h_beams[s];
            if (e->opt.layout == SB_LAYOUT_V1) {
                uint32_t pair = (h.da & 0xffffu) | (h.db << 16);
                memcpy(rec, &pair, 4);
            } else {
                memcpy(rec, &h.da, 4);
                memcpy(rec + 4, &h.db, 4);
            }
            memcpy(f, h.f, 9 * sizeof(float));
            memcpy(f + 4, &t[c], 4);   // target_length
            memcpy(f + 8, &l[c], 4);   // last_length
            memcpy(f + 28, &sn[c], 4); // strain
            memcpy(f + 32, &ss[c], 4); // stress
        }
        });
        tm.mark("readback: beam records");
    }
    return SB_OK;
}

// HBM bytes ONE lean substep launch has to move with the data layout this engine actually holds (the launched
// kernel's own compulsory traffic: what bench.py prices `roofline.achieved` with, and what the PMC counters
// of profiles/ must reproduce).  Counted: every array element the kernel reads or writes once per substep.
// Not counted: accelerations (skipped while they are zero), strain/stress (stored by the last
// substep of a call only), target stores (plastic yield only), neighbour-list traffic of SB_COLLIDE_GRID.
// one launch of a long call of a blocked plan = k_long substeps: every entry word of that depth, the state of every halo entry
// (index + target + last, gathered), the owned states in and out, own particles in and out, halo particles (index + position +
// velocity); per substep
static uint64_t blocked_bytes_model(const sb_engine *e, const SbBlockedDev &bk)
{
    const uint64_t P = e->P, nb = bk.nbeams;
    const uint32_t k = bk.k_long;
    const uint64_t entries = bk.entries_at[k], halo_entries = entries - std::min<uint64_t>(entries, nb),
                   halo_particles = bk.region_at[k] - std::min<uint64_t>(bk.region_at[k], P);
    // (targets: neither read, written nor gathered while no tile has yielded; priced as of the upload)
    const uint64_t per_launch = entries * (4 + (bk.mat_mode == 1 ? 4 : 0)) + halo_entries * (bk.pristine ? 8 : 12) + nb * (bk.pristine ? 8 : 16) + P * 32 +
                                halo_particles * 20 + (uint64_t)bk.ntiles * (8 * 4 + 8 * k) + (uint64_t)bk.nmat * 24;
    return per_launch / k;
}

static uint64_t substep_bytes_model(const sb_engine *e)
{
    const uint64_t P = e->P, nc = e->nbeam;
    if (e->bk.K) return blocked_bytes_model(e, e->bk);
    if (e->path == SB_PATH_TILED) {
        uint64_t per_copy = 4 /* endpoint word */ + 4 /* target */ + 4 + 4 /* last: read, written */;
        if (e->mat_mode <= 1) per_copy += 4;  // per-copy rest length
        if (e->mat_mode == 0) per_copy += 16; // per-copy spring, damp, yield, limit
        return nc * per_copy + P * 32 /* pos, vel: read and written */ + (uint64_t)e->nhalo * 12 /* index + position */ +
               (uint64_t)e->ntiles * (3 * 4 + 2 * 4) /* tile tables, acceleration flags */ + (uint64_t)e->nmat * 24;
    }
    // atomic path: the reference's own schedule (compute.wgsl:98,125: whole records in and out, 4 atomics per beam)
    return nc * (8 + 36 + 16 + 16 + 16) + P * (48 + 16);
}

sb_status sb_get_info(sb_engine *e, const char *key, uint64_t *value)
{
    if (!e || !key || !value) return SB_ERR_INVALID;
    std::string k(key);
    if (k == "path") *value = e->path;
    else if (k == "substep_hbm_bytes") *value = substep_bytes_model(e);
    else if (k == "substeps_per_launch") *value = e->bk.K ? e->bk.k_long : 1; // of a long call (sbk_split_call)
    else if (k == "plan_depth") *value = e->bk.K ? e->bk.K : 1;
    else if (k.rfind("block_redundancy_x1000_", 0) == 0) { // beam evaluations per beam and substep of a launch of <k> substeps, in thousandths (ring redundancy of the blocked plan)
        const int d = atoi(key + 23);
        const SbBlockedDev &bk = e->bk.K ? e->bk : e->hy;
        *value = (bk.K && d >= 1 && d <= (int)bk.K && bk.nbeams) ? (uint64_t)(1000.0 * (double)bk.entries_at[d] / (double)bk.nbeams + 0.5) : 0;
    }
    else if (k == "region_particles") *value = e->bk.K ? e->bk.cap : e->tile_cap_all;
    else if (k == "tiles") *value = e->ntiles;
    else if (k == "beam_copies") *value = e->bk.K ? e->bk.entries_at[e->bk.k_long] : e->nbeam;
    else if (k == "halo_particles") *value = e->bk.K ? e->bk.region_at[e->bk.k_long] - e->P : e->nhalo;
    else if (k == "device_bytes") *value = e->device_bytes;
    else if (k == "substeps_done") *value = e->substeps_done;
    else if (k == "lds_bytes") *value = e->lds_bytes;
    else if (k == "uploads_kept") *value = e->uploads_kept;
    else if (k == "kernels_per_substep") // (the hash's helper launch runs in the classic schedule only: sb_physics.h SbGridCtl)
        *value = (e->path == SB_PATH_TILED ? 1 : 2) + (e->opt.collision_mode == SB_COLLIDE_GRID && sbk_grid_mode(e) == SB_GRID_CLASSIC ? 1 : 0);
    else if (k == "grid_aborts") *value = e->grid_aborts;                   // lagged launches whose tail found the lists not known to be valid any more (host recovery)
    else if (k == "grid_helper_launches") *value = e->grid_helper_launches; // k_grid_build launches since the engine was created
    else if (k == "grid_classic_substeps") *value = e->grid_classic_substeps; // substeps run in the classic schedule (helper launch in front)
    else if (k == "grid_schedule") *value = sbk_grid_mode(e);               // 0 lagged, 1 classic: what the next substep would run
    else if (k == "grid_cells") *value = e->ncell;
    else if (k == "uploads_edited") *value = e->uploads_edited; // plan-keeping uploads that removed beams (rewrite_scene_state)
    else if (k == "grid_builds") {
        *value = 0;
        if (e->d_grid_ctl) {
            SbGridCtl ctl;
            SB_HIP(e, hipStreamSynchronize(e->stream));
            SB_HIP(e, hipMemcpy(&ctl, e->d_grid_ctl + e->grid_par, sizeof ctl, hipMemcpyDeviceToHost));
            *value = ctl.builds;
        }
    }
    else if (k == "grid_wide") { // 1 once the hash switched from the uploaded bounding box to the whole domain
        *value = 0;
        if (e->d_grid_ctl) {
            SbGridCtl ctl;
            SB_HIP(e, hipStreamSynchronize(e->stream));
            SB_HIP(e, hipMemcpy(&ctl, e->d_grid_ctl + e->grid_par, sizeof ctl, hipMemcpyDeviceToHost));
            *value = ctl.geo.wide;
        }
    }
    else if (k == "grid_skin_x1000") { // the skin of the current hash, in thousandths of a unit (it adapts)
        *value = 0;
        if (e->d_grid_ctl) {
            SbGridCtl ctl;
            SB_HIP(e, hipStreamSynchronize(e->stream));
            SB_HIP(e, hipMemcpy(&ctl, e->d_grid_ctl + e->grid_par, sizeof ctl, hipMemcpyDeviceToHost));
            *value = (uint64_t)(ctl.geo.skin * 1000.0f + 0.5f);
        }
    }
    else if (k == "beams_flagged") { // beams flagged by mark_beam_deleted (compute.wgsl:117-121) since the last delete pass
        *value = 0;
        const size_t words = ((size_t)e->nbeam + 31) / 32;
        if (words && e->d_broken) {
            std::vector<uint32_t> h(words);
            SB_HIP(e, hipStreamSynchronize(e->stream));
            SB_HIP(e, hipMemcpy(h.data(), e->d_broken, words * 4, hipMemcpyDeviceToHost));
            uint64_t n = 0;
            for (uint32_t w : h) n += (uint64_t)__builtin_popcount(w);
            *value = n;
        }
    }
    else if (k.rfind("grid_stamp_", 0) == 0) { // -DSB_STAMPS builds: stamps of one mid-grid workgroup of the last substep launch, 10 ns ticks since its start (diagnostic)
        const int i = atoi(key + 11);
        SB_HIP(e, hipStreamSynchronize(e->stream));
        *value = (i >= 0 && i < 40 && e->dev_err) ? e->dev_err[4 + i] : 0; // (i >= 16: the last launch that made its lists together)
    }
    else if (k.rfind("grid_ctl_", 0) == 0) { // word <n> of the SbGridCtl block the next launch reads (sb_physics.h; diagnostic: tools/grid_ctl_dump.py)
        const int i = atoi(key + 9);
        *value = 0;
        if (e->d_grid_ctl && i >= 0 && i < (int)(sizeof(SbGridCtl) / 4)) {
            uint32_t w = 0;
            SB_HIP(e, hipStreamSynchronize(e->stream));
            SB_HIP(e, hipMemcpy(&w, (const uint32_t *)(e->d_grid_ctl + e->grid_par) + i, 4, hipMemcpyDeviceToHost));
            *value = w;
        }
    }
    else if (k == "hybrid") *value = e->hy.K;                         // depth of the blocked plan beside the tiling (0: none, or not on the device yet)
    else if (k == "hybrid_substep_hbm_bytes") *value = e->hy.K ? blocked_bytes_model(e, e->hy) : 0; // of its blocked launches (k = hybrid_substeps_per_launch)
    else if (k == "hybrid_substeps_per_launch") *value = e->hy.K ? e->hy.k_long : 0;
    else if (k == "hybrid_pending") *value = e->hy_pending ? 1 : 0;   // that plan is (being) made on the side thread and has not been needed yet
    else if (k == "hybrid_substeps") *value = e->hy.substeps_blocked; // substeps that ran blocked under SB_COLLIDE_GRID
    else if (k == "hybrid_launches") *value = e->hy.launches_ok;      // tracked launches that were validated
    else if (k == "hybrid_failed") *value = e->hy.launches_failed;    // tracked launches that went over the skin and were redone
    else if (k == "hybrid_validate_launches") *value = e->hy.validate_launches; // k_hybrid_validate launches: one per run (a launch is validated by its successor's prologue)
    else if (k == "material_mode") *value = e->mat_mode;
    else if (k == "materials") *value = e->nmat;
    else if (k == "local_index_bits") *value = e->lbits;
    else SB_FAIL(e, SB_ERR_INVALID, "sb_get_info: unknown key '%s'", key);
    return SB_OK;
}

static sb_status sb_halo_configure_impl(sb_engine *e, const uint32_t *ghost_particles, uint32_t n_gp,
                            const uint32_t *send_particles, uint32_t n_sp, const uint32_t *ghost_beams,
                            uint32_t n_gb, const uint32_t *send_beams, uint32_t n_sb)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_halo_configure before sb_write_buffers");
    if ((n_gp && !ghost_particles) || (n_sp && !send_particles) || (n_gb && !ghost_beams) || (n_sb && !send_beams))
        SB_FAIL(e, SB_ERR_INVALID, "null index list");
    SB_HIP(e, hipSetDevice(e->device));
    SB_HIP(e, hipStreamSynchronize(e->stream));
    const uint32_t maxP = e->opt.max_particles, maxB = e->opt.max_beams;
    std::vector<uint32_t> internal_of_index(maxP, 0xFFFFFFFFu), slot_of_beam(maxB, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < e->P; i++) internal_of_index[e->h_pidx[i]] = i;
    for (uint32_t u = 0; u < sb_user_beams(e); u++) slot_of_beam[map_get(e, e->h_mapping.data(), (size_t)maxP + u)] = sb_user_slot(e, u);
    auto translate = [](const uint32_t *src, uint32_t n, const std::vector<uint32_t> &table, std::vector<uint32_t> &dst) {
        dst.resize(n);
        for (uint32_t k = 0; k < n; k++) {
            if (src[k] >= table.size() || table[src[k]] == 0xFFFFFFFFu) return false;
            dst[k] = table[src[k]];
        }
        return true;
    };
    std::vector<uint32_t> gp, sp, gb, sbm;
    if (!translate(ghost_particles, n_gp, internal_of_index, gp) || !translate(send_particles, n_sp, internal_of_index, sp))
        SB_FAIL(e, SB_ERR_INVALID, "halo list names a particle data index that is not active");
    if (!translate(ghost_beams, n_gb, slot_of_beam, gb) || !translate(send_beams, n_sb, slot_of_beam, sbm))
        SB_FAIL(e, SB_ERR_INVALID, "halo list names a beam data index that is not active");
    // sent beams: any one copy (all copies of a valid beam are identical)
    std::vector<uint32_t> send_copy(n_sb);
    for (uint32_t k = 0; k < n_sb; k++) send_copy[k] = e->h_copy_of_slot[sbm[k]];
    // ghost beams: every copy
    std::vector<uint32_t> pos_of_slot(e->B, 0xFFFFFFFFu);
    for (uint32_t k = 0; k < n_gb; k++) pos_of_slot[gb[k]] = k;
    std::vector<uint32_t> copy_slot(e->nbeam);
    if (e->nbeam) SB_HIP(e, hipMemcpy(copy_slot.data(), e->beams.slot, (size_t)e->nbeam * 4, hipMemcpyDeviceToHost));
    std::vector<uint2> ghost_copies;
    for (uint32_t c = 0; c < e->nbeam; c++) {
        uint32_t s = copy_slot[c];
        if (s != 0xFFFFFFFFu && pos_of_slot[s] != 0xFFFFFFFFu) ghost_copies.push_back(make_uint2(c, pos_of_slot[s]));
    }
    if (e->mailbox) SB_FAIL(e, SB_ERR_STATE, "sb_halo_configure after sb_peer_mailbox (its size follows the lists): upload again to reconfigure");
    // a second configuration replaces the first: release its lists now, not at the next upload
    for (void *old : {(void *)e->d_ghost_p, (void *)e->d_send_p, (void *)e->d_send_b, (void *)e->d_ghost_b, (void *)e->d_send_p_off,
                      (void *)e->d_send_b_off, (void *)e->d_ghost_p_off, (void *)e->d_ghost_b_off}) {
        if (!old) continue;
        auto it = std::find_if(e->pool_used.begin(), e->pool_used.end(), [old](const std::pair<void *, size_t> &b) { return b.first == old; });
        if (it != e->pool_used.end()) { // back to the pool (the lists that replace them are about to ask for the same sizes)
            e->pool_free.push_back(*it);
            e->pool_used.erase(it);
        }
    }
    e->d_ghost_p = e->d_send_p = e->d_send_b = e->d_send_p_off = e->d_send_b_off = e->d_ghost_p_off = e->d_ghost_b_off = nullptr;
    e->d_ghost_b = nullptr;
    SB_TRY(dev_alloc(e, &e->d_ghost_p, n_gp));
    SB_TRY(dev_alloc(e, &e->d_send_p, n_sp));
    SB_TRY(dev_alloc(e, &e->d_send_b, n_sb));
    SB_TRY(dev_alloc(e, &e->d_ghost_b, ghost_copies.size()));
    if (n_gp) SB_HIP(e, hipMemcpy(e->d_ghost_p, gp.data(), (size_t)n_gp * 4, hipMemcpyHostToDevice));
    if (n_sp) SB_HIP(e, hipMemcpy(e->d_send_p, sp.data(), (size_t)n_sp * 4, hipMemcpyHostToDevice));
    if (n_sb) SB_HIP(e, hipMemcpy(e->d_send_b, send_copy.data(), (size_t)n_sb * 4, hipMemcpyHostToDevice));
    if (!ghost_copies.empty())
        SB_HIP(e, hipMemcpy(e->d_ghost_b, ghost_copies.data(), ghost_copies.size() * sizeof(uint2), hipMemcpyHostToDevice));
    e->n_ghost_p = n_gp;
    e->n_send_p = n_sp;
    e->n_ghost_b = n_gb;
    e->n_send_b = n_sb;
    e->n_ghost_b_copies = (uint32_t)ghost_copies.size();
    SB_TRY(dev_alloc(e, &e->d_send_p_off, n_sp));
    SB_TRY(dev_alloc(e, &e->d_send_b_off, n_sb));
    SB_TRY(dev_alloc(e, &e->d_ghost_p_off, n_gp));
    SB_TRY(dev_alloc(e, &e->d_ghost_b_off, n_gb));
    return sb_halo_set_layout(e, nullptr, nullptr, nullptr, nullptr); // default: particles back to back, then beams
}

sb_status sb_halo_set_layout(sb_engine *e, const uint32_t *send_p_off, const uint32_t *send_b_off,
                             const uint32_t *ghost_p_off, const uint32_t *ghost_b_off)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_halo_set_layout before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    SB_HIP(e, hipStreamSynchronize(e->stream));
    uint32_t extent = 0;
    auto put = [&](uint32_t *dst, const uint32_t *src, uint32_t n, uint32_t base, uint32_t stride) -> sb_status {
        if (!n) return SB_OK;
        std::vector<uint32_t> dflt;
        if (!src) {
            dflt.resize(n);
            for (uint32_t k = 0; k < n; k++) dflt[k] = base + stride * k;
            src = dflt.data();
        }
        for (uint32_t k = 0; k < n; k++)
            if (src[k] & 1u) SB_FAIL(e, SB_ERR_INVALID, "halo offsets must be even (8-byte aligned records)");
        for (uint32_t k = 0; k < n; k++) extent = std::max(extent, src[k] + stride);
        SB_HIP(e, hipMemcpy(dst, src, (size_t)n * 4, hipMemcpyHostToDevice));
        return SB_OK;
    };
    if (e->mailbox) SB_FAIL(e, SB_ERR_STATE, "sb_halo_set_layout after sb_peer_mailbox");
    SB_TRY(put(e->d_send_p_off, send_p_off, e->n_send_p, 0, 6));
    SB_TRY(put(e->d_send_b_off, send_b_off, e->n_send_b, 6 * e->n_send_p, 2));
    e->send_floats = extent;
    extent = 0;
    SB_TRY(put(e->d_ghost_p_off, ghost_p_off, e->n_ghost_p, 0, 6));
    SB_TRY(put(e->d_ghost_b_off, ghost_b_off, e->n_ghost_b, 6 * e->n_ghost_p, 2));
    e->recv_floats = extent;
    return SB_OK;
}

// ---- direct peer exchange -------------------------------------------------------------------
static inline size_t mailbox_stride(uint32_t recv_floats) { return ((size_t)recv_floats * 4 + 255) & ~(size_t)255; }
static const size_t kMailboxFlagsBytes = 256;

sb_status sb_peer_mailbox(sb_engine *e, void **local_mailbox, void *ipc_handle, uint64_t *mailbox_bytes)
{
    if (!e || !local_mailbox) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_peer_mailbox before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    if (!e->mailbox) {
        size_t bytes = kMailboxFlagsBytes + 2 * mailbox_stride(e->recv_floats);
        void *q = nullptr;
        // fine-grained: stores arriving over xGMI and the flag polling stay coherent without a kernel boundary
        SB_HIP(e, hipExtMallocWithFlags(&q, bytes, hipDeviceMallocFinegrained));
        e->allocs.push_back(q);
        e->device_bytes += bytes;
        SB_HIP(e, hipMemset(q, 0, bytes));
        SB_HIP(e, hipDeviceSynchronize());
        e->mailbox = q;
    }
    *local_mailbox = e->mailbox;
    if (mailbox_bytes) *mailbox_bytes = kMailboxFlagsBytes + 2 * mailbox_stride(e->recv_floats);
    if (ipc_handle) {
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
        hipIpcMemHandle_t h;
        SB_HIP(e, hipIpcGetMemHandle(&h, e->mailbox));
        memcpy(ipc_handle, &h, sizeof h);
    }
    return SB_OK;
}

sb_status sb_peer_map(sb_engine *e, const void *ipc_handle, void **mapped_mailbox)
{
    if (!e || !ipc_handle || !mapped_mailbox) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_peer_map before sb_write_buffers");
    SB_HIP(e, hipSetDevice(e->device));
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle, sizeof h);
    void *q = nullptr;
    SB_HIP(e, hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess));
    e->mapped.push_back(q);
    *mapped_mailbox = q;
    return SB_OK;
}

sb_status sb_peer_connect(sb_engine *e, uint32_t n_peers, void *const *mailboxes, const uint32_t *peer_recv_floats,
                          const uint32_t *send_begin, const uint32_t *send_len, const uint32_t *dst_begin,
                          const uint32_t *their_slot, uint32_t timeout_ms)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->mailbox) SB_FAIL(e, SB_ERR_STATE, "sb_peer_connect before sb_peer_mailbox");
    if (n_peers > SB_MAX_PEERS) SB_FAIL(e, SB_ERR_INVALID, "at most %d neighbours", SB_MAX_PEERS);
    // The sequence flags in both sides' mailboxes only ever count up.  Re-connecting after exchanges have run would have
    // to restart them on every side at once; a restart on one side alone makes stale mailbox data look fresh.  A new
    // upload (sb_write_buffers) releases the mailbox and is the way to start over.
    if (e->peer_seq != 0)
        SB_FAIL(e, SB_ERR_STATE, "sb_peer_connect on a mailbox that has already exchanged %u times: upload again to start over", e->peer_seq);
    if (n_peers && (!mailboxes || !peer_recv_floats || !send_begin || !send_len || !dst_begin || !their_slot))
        return SB_ERR_INVALID;
    for (uint32_t j = 0; j < n_peers; j++) {
        if (!mailboxes[j]) SB_FAIL(e, SB_ERR_INVALID, "neighbour %u has no mailbox", j);
        if ((uint64_t)send_begin[j] + send_len[j] > e->send_floats)
            SB_FAIL(e, SB_ERR_INVALID, "send segment of neighbour %u exceeds the packed send layout", j);
        if ((uint64_t)dst_begin[j] + send_len[j] > peer_recv_floats[j])
            SB_FAIL(e, SB_ERR_INVALID, "send segment of neighbour %u does not fit its receive layout", j);
        if (their_slot[j] >= 64 || ((send_begin[j] | dst_begin[j]) & 1u))
            SB_FAIL(e, SB_ERR_INVALID, "neighbour %u: bad flag slot or odd segment offset", j);
        for (uint32_t i = 0; i < j; i++)
            if (send_begin[j] < send_begin[i] + send_len[i] && send_begin[i] < send_begin[j] + send_len[j])
                SB_FAIL(e, SB_ERR_INVALID, "send segments of neighbours %u and %u overlap", i, j);
    }
    e->n_peers = n_peers;
    for (uint32_t j = 0; j < n_peers; j++) {
        e->peer_box[j] = mailboxes[j];
        e->peer_stride[j] = (uint32_t)mailbox_stride(peer_recv_floats[j]);
        e->peer_begin[j] = send_begin[j];
        e->peer_len[j] = send_len[j];
        e->peer_dst[j] = dst_begin[j];
        e->peer_slot[j] = their_slot[j];
    }
    e->peer_timeout_ms = timeout_ms ? timeout_ms : 10000;
    return SB_OK;
}

sb_status sb_peer_exchange(sb_engine *e)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded || !e->mailbox) SB_FAIL(e, SB_ERR_STATE, "sb_peer_exchange before sb_peer_connect");
    if (!e->n_peers) return SB_OK;
    SB_HIP(e, hipSetDevice(e->device));
    sbk_launch_peer_exchange(e);
    SB_HIP(e, hipGetLastError());
    return SB_OK;
}

sb_status sb_halo_pack(sb_engine *e, void *device_dst)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_halo_pack before sb_write_buffers");
    if ((e->n_send_p || e->n_send_b) && !device_dst) SB_FAIL(e, SB_ERR_INVALID, "null device buffer");
    SB_HIP(e, hipSetDevice(e->device));
    sbk_launch_halo_pack(e, (float *)device_dst);
    SB_HIP(e, hipGetLastError());
    return SB_OK;
}

sb_status sb_halo_unpack(sb_engine *e, const void *device_src)
{
    if (!e) return SB_ERR_INVALID;
    if (!e->loaded) SB_FAIL(e, SB_ERR_STATE, "sb_halo_unpack before sb_write_buffers");
    if ((e->n_ghost_p || e->n_ghost_b) && !device_src) SB_FAIL(e, SB_ERR_INVALID, "null device buffer");
    SB_HIP(e, hipSetDevice(e->device));
    sbk_launch_halo_unpack(e, (const float *)device_src);
    SB_HIP(e, hipGetLastError());
    return SB_OK;
}

sb_status sb_get_stream(sb_engine *e, void **hip_stream)
{
    if (!e || !hip_stream) return SB_ERR_INVALID;
    *hip_stream = (void *)e->stream;
    return SB_OK;
}

// C++ exceptions must not cross the C boundary: the entry points that allocate host memory run
// their bodies under a catch-all that turns std::bad_alloc (and anything else) into a status.
#define SB_GUARDED(e, call)                                                                   \
    try {                                                                                     \
        return (call);                                                                        \
    } catch (const std::bad_alloc &) {                                                        \
        if (e) (e)->err = "out of host memory";                                               \
        return SB_ERR_OOM;                                                                    \
    } catch (const std::exception &ex) {                                                      \
        if (e) (e)->err = std::string("internal error: ") + ex.what();                        \
        return SB_ERR_INVALID;                                                                \
    }

sb_status sb_write_buffers(sb_engine *e, const void *metadata, size_t metadata_bytes, const void *mapping,
                           size_t mapping_bytes, const void *particles, size_t particles_bytes,
                           const void *beams, size_t beams_bytes)
{
    SB_GUARDED(e, sb_write_buffers_impl(e, metadata, metadata_bytes, mapping, mapping_bytes, particles, particles_bytes,
                                        beams, beams_bytes))
}

sb_status sb_load_buffers(sb_engine *e, void *metadata, size_t metadata_bytes, void *mapping, size_t mapping_bytes,
                          void *particles, size_t particles_bytes, void *beams, size_t beams_bytes)
{
    SB_GUARDED(e, sb_load_buffers_impl(e, metadata, metadata_bytes, mapping, mapping_bytes, particles, particles_bytes,
                                       beams, beams_bytes))
}

sb_status sb_halo_configure(sb_engine *e, const uint32_t *ghost_particles, uint32_t n_gp,
                            const uint32_t *send_particles, uint32_t n_sp, const uint32_t *ghost_beams,
                            uint32_t n_gb, const uint32_t *send_beams, uint32_t n_sb)
{
    SB_GUARDED(e, sb_halo_configure_impl(e, ghost_particles, n_gp, send_particles, n_sp, ghost_beams, n_gb, send_beams, n_sb))
}

} // extern "C"
