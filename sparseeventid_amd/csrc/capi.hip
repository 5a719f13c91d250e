// C ABI glue (include/sparseeventid_amd.h): error state, device pool, profiling, handle management.
#include "common.h"

#include <atomic>
#include <map>

#include <string.h>

namespace sei {

// ---------------------------------------------------------------- errors
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }

// ---------------------------------------------------------------- pool
namespace {
struct PoolBlock {
    size_t size;
    int device;
    hipStream_t stream;
};
typedef std::pair<int, hipStream_t> PoolKey;
// Blocks that work on SEVERAL streams may still be reading when they are freed (hash tables, rule lists, neighbour tables
// of a metadata: built on one stream, read by forward / backward kernels on others).  They are released behind a fence:
// one event per stream the library has been handed on that device, recorded at the time of the free; the blocks return to
// their free list only once every one of those events has completed, i.e. when everything queued before the free -- on
// any stream -- is done.
struct Fence {
    std::vector<hipEvent_t> events;
    // references: one per quarantined block still waiting on this fence, plus ONE held by the owner between
    // pool_fence_begin and pool_fence_end -- a concurrent pool_alloc that reaps the fence's first blocks (their events may
    // already have completed) must not delete it under the owner, who is still adding blocks
    int refs = 1;
};
struct Quarantined {
    void *p;
    PoolBlock b;
    Fence *fence;
};
struct Pool {
    std::mutex mu;
    std::map<PoolKey, std::multimap<size_t, void *>> free_blocks;
    std::map<void *, PoolBlock> live;
    std::vector<Quarantined> quarantine;
    std::vector<PoolKey> streams;        // every (device, stream) the library has launched work on
    std::vector<hipEvent_t> spare_events;
    size_t held = 0, in_use = 0;
} g_pool;

bool fence_done_locked(Fence *f) {
    while (!f->events.empty()) {
        hipError_t e = hipEventQuery(f->events.back());
        if (e == hipErrorNotReady) {
            (void)hipGetLastError();
            return false;
        }
        g_pool.spare_events.push_back(f->events.back());
        f->events.pop_back();
    }
    return true;
}

void fence_unref_locked(Fence *f) {
    if (--f->refs > 0) return;
    for (hipEvent_t ev : f->events) g_pool.spare_events.push_back(ev);
    delete f;
}

// quarantined blocks whose fence has completed go back to their free lists
void reap_quarantine_locked() {
    size_t keep = 0;
    for (size_t i = 0; i < g_pool.quarantine.size(); i++) {
        Quarantined q = g_pool.quarantine[i];
        if (fence_done_locked(q.fence)) {
            g_pool.free_blocks[PoolKey(q.b.device, q.b.stream)].emplace(q.b.size, q.p);
            fence_unref_locked(q.fence);
        } else {
            g_pool.quarantine[keep++] = q;
        }
    }
    g_pool.quarantine.resize(keep);
}

size_t round_size(size_t b) {
    if (b < 256) return 256;
    if (b <= (1u << 20)) {  // powers of two below 1 MiB
        size_t s = 256;
        while (s < b) s <<= 1;
        return s;
    }
    const size_t g = 1u << 20;  // 1 MiB granules above
    return (b + g - 1) / g * g;
}

void release_cached_locked() {
    for (auto &kv : g_pool.free_blocks)
        for (auto &b : kv.second) {
            (void)hipFree(b.second);
            g_pool.held -= b.first;
        }
    g_pool.free_blocks.clear();
}
}  // namespace

void note_stream(hipStream_t st) {
    static thread_local PoolKey last(-1, nullptr);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const PoolKey key(dev, st);
    if (key == last) return;
    last = key;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (const PoolKey &k : g_pool.streams)
        if (k == key) return;
    g_pool.streams.push_back(key);
}

void *pool_alloc(size_t bytes, hipStream_t st) {
    size_t sz = round_size(bytes);
    note_stream(st);
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (!g_pool.quarantine.empty()) reap_quarantine_locked();
    auto &fl = g_pool.free_blocks[PoolKey(dev, st)];
    auto it = fl.lower_bound(sz);
    if (it != fl.end() && it->first <= sz + sz / 4 + (1u << 20)) {
        void *p = it->second;
        size_t got = it->first;
        fl.erase(it);
        g_pool.live[p] = PoolBlock{got, dev, st};
        g_pool.in_use += got;
        return p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
        // release cached blocks (all streams: the allocator call itself synchronises) and retry once
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        reap_quarantine_locked();
        release_cached_locked();
        (void)hipGetLastError();
        e = hipMalloc(&p, sz);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    g_pool.held += sz;
    g_pool.in_use += sz;
    g_pool.live[p] = PoolBlock{sz, dev, st};
    return p;
}

void pool_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.live.find(p);
    if (it == g_pool.live.end()) return;
    const PoolBlock b = it->second;
    g_pool.free_blocks[PoolKey(b.device, b.stream)].emplace(b.size, p);
    g_pool.in_use -= b.size;
    g_pool.live.erase(it);
}

PoolFence pool_fence_begin() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_pool.mu);
    Fence *f = new Fence();
    for (const PoolKey &k : g_pool.streams) {
        if (k.first != dev) continue;
        hipEvent_t ev = nullptr;
        if (!g_pool.spare_events.empty()) {
            ev = g_pool.spare_events.back();
            g_pool.spare_events.pop_back();
        } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        if (hipEventRecord(ev, k.second) != hipSuccess) {   // a stream the caller has destroyed meanwhile: nothing left on it
            (void)hipGetLastError();
            g_pool.spare_events.push_back(ev);
            continue;
        }
        f->events.push_back(ev);
    }
    return f;
}

void pool_free_fenced(void *p, PoolFence fence) {
    if (!p) return;
    Fence *f = (Fence *)fence;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.live.find(p);
    if (it == g_pool.live.end()) return;
    g_pool.quarantine.push_back(Quarantined{p, it->second, f});
    f->refs++;
    g_pool.in_use -= it->second.size;
    g_pool.live.erase(it);
}

void pool_fence_end(PoolFence fence) {
    Fence *f = (Fence *)fence;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    fence_unref_locked(f);  // the owner's reference; the last quarantined block deletes the fence otherwise
}

static std::atomic<long long> g_sync_count{0};
hipError_t sync_stream(hipStream_t st) {
    g_sync_count.fetch_add(1, std::memory_order_relaxed);
    return hipStreamSynchronize(st);
}

int ensure_max_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(dev, kernel);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return 0;
    SEI_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[key] = bytes;
    return 0;
}

// ---------------------------------------------------------------- profiling
namespace {
struct ProfRec {
    hipEvent_t e0, e1;
    double flops, bytes;
};
struct Prof {
    std::mutex mu;
    bool on = false;
    std::vector<ProfRec> recs[PROF_NKIND];
    std::vector<hipEvent_t> spare;
    double ms[PROF_NKIND] = {0}, flops[PROF_NKIND] = {0}, bytes[PROF_NKIND] = {0};
    double launches[PROF_NKIND] = {0};
} g_prof;

hipEvent_t get_event() {
    if (!g_prof.spare.empty()) {
        hipEvent_t e = g_prof.spare.back();
        g_prof.spare.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void drain_locked() {
    for (int k = 0; k < PROF_NKIND; k++) {
        for (auto &r : g_prof.recs[k]) {
            float ms = 0.f;
            if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
                g_prof.ms[k] += ms;
                g_prof.flops[k] += r.flops;
                g_prof.bytes[k] += r.bytes;
                g_prof.launches[k] += 1;
            }
            g_prof.spare.push_back(r.e0);
            g_prof.spare.push_back(r.e1);
        }
        g_prof.recs[k].clear();
    }
}
}  // namespace

ProfScope::ProfScope(int kind_, hipStream_t st_, double flops_, double bytes_)
    : kind(kind_), st(st_), flops(flops_), bytes(bytes_) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    e0 = get_event();
    e1 = get_event();
    if (e0) (void)hipEventRecord(e0, st);
}
ProfScope::~ProfScope() {
    if (!e0 || !e1) return;
    (void)hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.recs[kind].push_back(ProfRec{e0, e1, flops, bytes});
}

}  // namespace sei

static std::atomic<int> g_force_kernel{0};
static std::atomic<int> g_product_mode{0};
static std::atomic<int> g_force_dw{0};
namespace sei {
int forced_conv_kernel() {
#ifdef SEI_EXPERIMENTS  // the sweep tools of a kernel-experiment build may also pin the family through the environment
    if (const char *e = getenv("SEI_CONV_KERNEL")) {
        if (!strcmp(e, "tile")) return CONV_KERNEL_TILE;
        if (!strcmp(e, "wave")) return CONV_KERNEL_WAVE;
        if (!strcmp(e, "band")) return CONV_KERNEL_BAND;
        if (!strcmp(e, "cols")) return CONV_KERNEL_COLS;
        if (!strcmp(e, "rounds")) return CONV_KERNEL_ROUNDS;
    }
#endif
    return g_force_kernel.load(std::memory_order_relaxed);
}
int product_mode() { return g_product_mode.load(std::memory_order_relaxed); }
int forced_dw_kernel() {
#ifdef SEI_EXPERIMENTS
    if (const char *e = getenv("SEI_DW_KERNEL")) {
        if (!strcmp(e, "reg")) return DW_KERNEL_REG;
        if (!strcmp(e, "ring")) return DW_KERNEL_RING;
    }
#endif
    return g_force_dw.load(std::memory_order_relaxed);
}
}  // namespace sei

using namespace sei;

static void free_level(Level *l, PoolFence f) {
    pool_free_fenced(l->coords, f);
    pool_free_fenced(l->ht.keys, f);
    pool_free_fenced(l->ht.vals, f);
    pool_free_fenced(l->ht.filter, f);
    delete l;
}

static void free_tiled_fenced(TiledRules &t, PoolFence f) {
    pool_free_fenced(t.tile_sub_start, f); pool_free_fenced(t.tile_seg, f); pool_free_fenced(t.seg_start, f);
    pool_free_fenced(t.sub_src, f); pool_free_fenced(t.sub_dst, f);
    t = TiledRules();
}

extern "C" {

const char *sei_last_error(void) { return g_error.c_str(); }
int sei_abi_version(void) { return 1; }

int sei_metadata_create(sei_metadata_t **out) {
    if (!out) SEI_FAIL("sei_metadata_create: null out");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        SEI_FAIL("sparseeventid_amd: no HIP device available (this engine has no CPU fallback)");
    }
    *out = new sei_metadata();
    return 0;
}

void sei_metadata_destroy(sei_metadata_t *m) {
    if (!m) return;
    // kernels on other streams than the one that built them may still be reading these blocks (forward / backward on the
    // caller's streams, dW on the library's side stream): they are reusable once everything queued so far is done
    PoolFence f = pool_fence_begin();
    for (auto *rb : m->rulebooks) {
        for (int v = 0; v < SEI_NVARIANT; v++) {
            free_tiled_fenced(rb->fwd[v], f);
            free_tiled_fenced(rb->bwd[v], f);
        }
        pool_free_fenced(rb->nbr_fwd, f);
        pool_free_fenced(rb->nbr_bwd, f);
        delete rb;
    }
    for (auto *l : m->levels) free_level(l, f);
    pool_free_fenced(m->input_row, f);
    pool_free_fenced(m->row_first, f);
    pool_free_fenced(m->row_last, f);
    pool_free_fenced(m->row_count, f);
    pool_free_fenced(m->dup_order, f);
    pool_free_fenced(m->dup_start, f);
    pool_fence_end(f);
    delete m;
}

int sei_input_build(sei_metadata_t *m, const int32_t *coords_dev, int64_t n, const int32_t spatial[3],
                    int32_t batch_size, int32_t mode, sei_stream_t stream, int64_t *n_rows) {
    if (!m || !n_rows || (n > 0 && !coords_dev)) SEI_FAIL("sei_input_build: null argument");
    if (mode < 0 || mode > 4) SEI_FAIL("InputLayer: mode must be 0..4");
    std::lock_guard<std::mutex> lk(m->mu);
    if (!m->levels.empty()) SEI_FAIL("InputLayer: metadata already holds an input level");
    m->input_mode = mode;
    ProfScope ps(PROF_RULES, (hipStream_t)stream, 0.0, 12.0 * n);
    return build_input_level(m, coords_dev, n, spatial, batch_size, (hipStream_t)stream, n_rows);
}

int sei_larcv_to_scn(const float *larcv_dev, int32_t B, int32_t P, int32_t M, int32_t D, int32_t *coords_out_dev,
                     float *features_out_dev, sei_stream_t stream, int64_t *n_points) {
    if (!larcv_dev || !coords_out_dev || !features_out_dev || !n_points) SEI_FAIL("sei_larcv_to_scn: null argument");
    return larcv_compact(larcv_dev, B, P, M, D, coords_out_dev, features_out_dev, n_points, (hipStream_t)stream);
}

int sei_input_features_fwd(sei_metadata_t *m, const float *in, int32_t C, float *out, sei_stream_t stream) {
    if (!m) SEI_FAIL("null metadata");
    note_stream((hipStream_t)stream);
    return input_features_fwd(m, in, C, out, (hipStream_t)stream);
}
int sei_input_features_bwd(sei_metadata_t *m, const float *dout, int32_t C, float *din, sei_stream_t stream) {
    if (!m) SEI_FAIL("null metadata");
    note_stream((hipStream_t)stream);
    return input_features_bwd(m, dout, C, din, (hipStream_t)stream);
}

int sei_level_rows(sei_metadata_t *m, const int32_t spatial[3], int64_t *n_rows) {
    if (!m || !spatial || !n_rows) SEI_FAIL("sei_level_rows: null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    Level *l = find_level(m, spatial);
    if (!l) SEI_FAIL("no level with this spatial size in the metadata");
    if (l->failed) SEI_FAIL("the discovery of this level failed earlier (hash table full or a lost synchronisation)");
    *n_rows = l->n;
    return 0;
}

int sei_level_coords(sei_metadata_t *m, const int32_t spatial[3], int32_t *coords_out_dev, sei_stream_t stream) {
    if (!m || !spatial) SEI_FAIL("sei_level_coords: null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    Level *l = find_level(m, spatial);
    if (!l) SEI_FAIL("no level with this spatial size in the metadata");
    if (l->n)
        SEI_HIP(hipMemcpyAsync(coords_out_dev, l->coords, (size_t)l->n * 4 * sizeof(int32_t), hipMemcpyDeviceToDevice,
                               (hipStream_t)stream));
    return 0;
}

int sei_batch_size(sei_metadata_t *m, int32_t *batch_size) {
    if (!m || !batch_size) SEI_FAIL("sei_batch_size: null argument");
    *batch_size = m->batch_size;
    return 0;
}

static bool same3(const int32_t *a, const int32_t *b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }

int sei_subm_rulebook(sei_metadata_t *m, const int32_t spatial[3], const int32_t filter[3], sei_stream_t stream,
                      sei_rulebook_t **rb_out) {
    std::lock_guard<std::mutex> lk(m->mu);
    for (auto *rb : m->rulebooks)
        if (rb->submanifold && same3(rb->spatial_in, spatial) && same3(rb->size, filter)) {
            *rb_out = rb;
            return 0;
        }
    Level *l = find_level(m, spatial);
    if (!l) SEI_FAIL("SubmanifoldConvolution: no level with this spatial size in the metadata");
    if (l->failed) SEI_FAIL("SubmanifoldConvolution: the discovery of this level failed earlier (hash table full or a lost synchronisation)");
    for (int d = 0; d < 3; d++)
        if (filter[d] < 1 || filter[d] % 2 == 0) SEI_FAIL("SubmanifoldConvolution: filter sizes must be odd");
    sei_rulebook *rb = new sei_rulebook();
    rb->submanifold = true;
    for (int d = 0; d < 3; d++) {
        rb->spatial_in[d] = rb->spatial_out[d] = spatial[d];
        rb->size[d] = filter[d];
        rb->stride[d] = 1;
    }
    rb->lin = rb->lout = l;  // tiled rule lists are built lazily, per tile size the kernels ask for
    // a larger odd filter on the same level (the 5^3 stem next to the 3^3 blocks of level 0) already probes every offset
    // of this one: the tiled lists are then cut from ITS neighbour table, no second probe
    for (auto *o : m->rulebooks)
        if (o->submanifold && o->lin == l && o->size[0] >= filter[0] && o->size[1] >= filter[1] && o->size[2] >= filter[2] &&
            (o->size[0] - filter[0]) % 2 == 0 && (o->size[1] - filter[1]) % 2 == 0 && (o->size[2] - filter[2]) % 2 == 0) {
            // both filters are odd (checked above for every submanifold rulebook), so the size differences are even and
            // NbrView::table_k's centring by (F - f) / 2 is exact; the explicit test keeps that true if even filters ever appear
            rb->donor = o;
            break;
        }
    m->rulebooks.push_back(rb);
    *rb_out = rb;
    return 0;
}

// scn.Convolution's rulebook: finds it, or creates it and BEGINS the discovery of its output level (build_strided_level_begin:
// nothing synchronised yet; *pl says whether a finish is owed).
// Returns CHAIN_BREAK (nothing created) when the input level is itself unfinished AND an input row may yield more than one
// candidate site (overlapping filter, e.g. size 3 / stride 2: 8): the level would be sized by the input's UPPER BOUND times
// that factor, which compounds along a chain (50 k points -> 400 k -> 3.2 M -> 25 M -> 205 M candidates; round-4 advisor).
// The caller then synchronises, finishes what is pending and calls again: the level is sized from the exact row count.
// size == stride (factor 1, every network of the reference) chains freely: its bound never exceeds the input points.
constexpr int CHAIN_BREAK = 2;
static int conv_rulebook_begin(sei_metadata_t *m, const int32_t spatial_in[3], const int32_t size[3], const int32_t stride[3],
                               hipStream_t st, sei_rulebook_t **rb_out, int32_t spatial_out[3], PendingLevel *pl,
                               bool may_break = false) {
    *pl = PendingLevel();
    std::lock_guard<std::mutex> lk(m->mu);
    for (auto *rb : m->rulebooks)
        if (!rb->submanifold && same3(rb->spatial_in, spatial_in) && same3(rb->size, size) && same3(rb->stride, stride)) {
            if (rb->lout && rb->lout->failed) SEI_FAIL("Convolution: the discovery of this rulebook's output level failed earlier (hash table full or a lost synchronisation)");
            *rb_out = rb;
            for (int d = 0; d < 3; d++) spatial_out[d] = rb->spatial_out[d];
            return 0;
        }
    Level *l = find_level(m, spatial_in);
    if (!l) SEI_FAIL("Convolution: no level with this spatial size in the metadata");
    if (l->failed) SEI_FAIL("Convolution: the discovery of the input level failed earlier (hash table full or a lost synchronisation)");
    if (may_break && l->n_dev) {
        int mult = 1;
        for (int d = 0; d < 3; d++)
            if (size[d] >= 1 && stride[d] >= 1) mult *= (size[d] + stride[d] - 1) / stride[d];
        if (mult > 1) return CHAIN_BREAK;
    }
    sei_rulebook *rb = new sei_rulebook();
    rb->submanifold = false;
    for (int d = 0; d < 3; d++) {
        rb->spatial_in[d] = spatial_in[d];
        rb->size[d] = size[d];
        rb->stride[d] = stride[d];
    }
    int rc = build_strided_level_begin(m, l, size, stride, rb, st, pl);
    if (rc) {
        delete rb;
        return rc;
    }
    m->rulebooks.push_back(rb);
    *rb_out = rb;
    for (int d = 0; d < 3; d++) spatial_out[d] = rb->spatial_out[d];
    return 0;
}

int sei_conv_rulebook(sei_metadata_t *m, const int32_t spatial_in[3], const int32_t size[3], const int32_t stride[3],
                      sei_stream_t stream, sei_rulebook_t **rb_out, int32_t spatial_out[3]) {
    PendingLevel pl;
    SEI_TRY(conv_rulebook_begin(m, spatial_in, size, stride, (hipStream_t)stream, rb_out, spatial_out, &pl));
    if (pl.pending && sync_stream((hipStream_t)stream) != hipSuccess) {
        build_strided_level_abort(&pl);   // the half-built level stays registered but marked failed: later lookups report it
        SEI_FAIL("rulebook: hipStreamSynchronize failed");
    }
    return build_strided_level_finish(&pl);
}

extern "C++" {
namespace {
struct LikeItem {
    bool subm;
    int32_t sp[3], size[3], stride[3];
    bool fwd[SEI_NVARIANT], bwd[SEI_NVARIANT], fuse[SEI_NVARIANT];
};

// what the metadata of an earlier batch holds: its rulebooks in creation order (= the order of its forward pass: strided
// rulebooks create the levels later ones need) and which tiled variants of each exist
std::vector<LikeItem> like_items(sei_metadata_t *like) {
    std::vector<LikeItem> items;
    std::lock_guard<std::mutex> lk(like->mu);
    for (auto *rb : like->rulebooks) {
        std::lock_guard<std::mutex> lr(rb->mu);
        LikeItem it;
        it.subm = rb->submanifold;
        for (int d = 0; d < 3; d++) { it.sp[d] = rb->spatial_in[d]; it.size[d] = rb->size[d]; it.stride[d] = rb->stride[d]; }
        for (int v = 0; v < SEI_NVARIANT; v++) {
            it.fwd[v] = rb->fwd[v].built; it.bwd[v] = rb->bwd[v].built; it.fuse[v] = rb->fwd[v].fwd_use;
        }
        items.push_back(it);
    }
    return items;
}

// Every rulebook of the list, the strided ones only BEGUN: a level discovered behind an unfinished one reads that one's row
// count from the device, so the whole chain of levels is queued back to back; the caller synchronises ONCE and finishes
// them (finish_levels).  Stops at the first failure (a level of the template that does not exist for this batch).
// `flush`: synchronises the stream and finishes everything begun so far (the caller's pending input level and `levels`);
// called where a chain must not go on from an upper bound (conv_rulebook_begin: CHAIN_BREAK).
template <class Flush>
int begin_rulebooks(sei_metadata_t *m, const std::vector<LikeItem> &items, hipStream_t st, std::vector<sei_rulebook_t *> *made,
                    std::vector<PendingLevel> *levels, Flush &&flush) {
    made->assign(items.size(), nullptr);
    for (size_t i = 0; i < items.size(); i++) {
        const LikeItem &it = items[i];
        int32_t spo[3];
        if (it.subm) {
            SEI_TRY(sei_subm_rulebook(m, it.sp, it.size, (sei_stream_t)st, &(*made)[i]));
        } else {
            PendingLevel pl;
            int rc = conv_rulebook_begin(m, it.sp, it.size, it.stride, st, &(*made)[i], spo, &pl, true);
            if (rc == CHAIN_BREAK) {
                SEI_TRY(flush());
                rc = conv_rulebook_begin(m, it.sp, it.size, it.stride, st, &(*made)[i], spo, &pl, false);
            }
            SEI_TRY(rc);
            if (pl.pending) levels->push_back(pl);
        }
    }
    return 0;
}

// in order: every begun level gets its host row count (also when a later item failed)
int finish_levels(std::vector<PendingLevel> *levels) {
    int rc = 0;
    for (PendingLevel &pl : *levels) {
        const int r = build_strided_level_finish(&pl);
        if (!rc) rc = r;
    }
    levels->clear();
    return rc;
}
}  // namespace
}  // extern "C++"

static int prebuild_like_impl(sei_metadata_t *m, sei_metadata_t *like, bool forward_only, sei_stream_t stream) {
    if (!m || !like) SEI_FAIL("sei_metadata_prebuild_like: null argument");
    if (m == like) return 0;
    const std::vector<LikeItem> items = like_items(like);
    // First the rulebooks and levels (ONE read-back for all strided levels that do not exist yet -- none after
    // sei_input_build_like), then every tiled variant is begun, ONE read-back serves them all, and they are finished: the
    // forward and backward passes that follow find everything built and synchronise nothing.
    std::vector<sei_rulebook_t *> made;
    std::vector<PendingLevel> levels;
    auto flush = [&]() -> int {
        if (!levels.empty() && sync_stream((hipStream_t)stream) != hipSuccess) SEI_FAIL("rulebook: hipStreamSynchronize failed");
        return finish_levels(&levels);
    };
    int rc0 = begin_rulebooks(m, items, (hipStream_t)stream, &made, &levels, flush);
    if (!levels.empty() && sync_stream((hipStream_t)stream) != hipSuccess && !rc0) {
        set_error("rulebook: hipStreamSynchronize failed");
        rc0 = 1;
    }
    const int rf = finish_levels(&levels);
    if (!rc0) rc0 = rf;
    if (rc0) return rc0;
    std::vector<TiledRequest> want;
    for (size_t i = 0; i < items.size(); i++) {
        const LikeItem &it = items[i];
        sei_rulebook_t *rb = made[i];
        for (int v = 0; v < SEI_NVARIANT; v++) {
            // forward_only: the variants a FORWARD convolution of the template fetched -- not the transposed rules of the
            // strided layers (dX) nor dW's 256-row variants: an evaluation pass between training steps builds nothing it
            // will not walk
            if (it.fwd[v] && (!forward_only || it.fuse[v])) want.push_back(TiledRequest{rb, false, 64 * (v + 1), it.fuse[v]});
            if (it.bwd[v] && !it.subm && !forward_only) want.push_back(TiledRequest{rb, true, 64 * (v + 1), false});
        }
    }
    return build_tiled_batch(want, (hipStream_t)stream);
}

// scn.InputLayer for a batch whose network is known from an EARLIER batch's metadata: the input level and the whole chain
// of strided levels `like` holds are discovered back to back with their row counts on the device, and ONE stream
// synchronisation gives the host all of them (sei_input_build + sei_metadata_prebuild_like: one for the InputLayer and one
// for the levels).  Rulebooks and levels that `like` has and this batch cannot (an event too small for a deep level) are
// skipped silently here: sei_metadata_prebuild_like / the lazy path report them as before.
int sei_input_build_like(sei_metadata_t *m, const int32_t *coords_dev, int64_t n, const int32_t spatial[3],
                         int32_t batch_size, int32_t mode, sei_metadata_t *like, sei_stream_t stream, int64_t *n_rows) {
    if (!m || !n_rows || (n > 0 && !coords_dev)) SEI_FAIL("sei_input_build_like: null argument");
    if (mode < 0 || mode > 4) SEI_FAIL("InputLayer: mode must be 0..4");
    if (!like || like == m) return sei_input_build(m, coords_dev, n, spatial, batch_size, mode, stream, n_rows);
    hipStream_t st = (hipStream_t)stream;
    PendingInput pi;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        if (!m->levels.empty()) SEI_FAIL("InputLayer: metadata already holds an input level");
        m->input_mode = mode;
        ProfScope ps(PROF_RULES, st, 0.0, 12.0 * n);
        const int rc = build_input_level_begin(m, coords_dev, n, spatial, batch_size, st, &pi);
        if (rc) {
            pool_free(pi.slot_of); pool_free(pi.flag); pool_free(pi.err_total); pool_free(pi.new_vals); stage_put(pi.host);
            return rc;
        }
    }
    std::vector<sei_rulebook_t *> made;
    std::vector<PendingLevel> levels;
    int rc = 0;
    bool input_done = false;
    const bool input_pending = pi.pending;
    // synchronises once and gives the host every count that is pending: the input level's, then the levels' in order
    auto flush = [&]() -> int {
        if (((input_pending && !input_done) || !levels.empty()) && sync_stream(st) != hipSuccess) {
            set_error("InputLayer: hipStreamSynchronize failed");
            rc = 1;
        }
        if (!input_done) {
            input_done = true;
            std::lock_guard<std::mutex> lk(m->mu);
            if (rc) {
                pool_free(pi.slot_of); pool_free(pi.flag); pool_free(pi.err_total); stage_put(pi.host);
            } else {
                rc = build_input_level_finish(m, &pi, st, n_rows);
            }
        }
        const int rf = finish_levels(&levels);
        if (!rc) rc = rf;
        return rc;
    };
    if (pi.pending) {
        const std::string keep = sei_last_error();
        (void)begin_rulebooks(m, like_items(like), st, &made, &levels, flush);   // failures: see above
        if (!rc) set_error(keep);   // (an error a mid-chain flush met stays: rc carries it)
    }
    if (rc) {   // a mid-chain flush failed: everything begun has been finished / released by it
        (void)finish_levels(&levels);
        return rc;
    }
    return flush();
}

int sei_metadata_prebuild_like(sei_metadata_t *m, sei_metadata_t *like, sei_stream_t stream) {
    return prebuild_like_impl(m, like, false, stream);
}

int sei_metadata_prebuild_forward_like(sei_metadata_t *m, sei_metadata_t *like, sei_stream_t stream) {
    return prebuild_like_impl(m, like, true, stream);
}

int sei_rulebook_info(const sei_rulebook_t *rb_c, int64_t info[8]) {
    sei_rulebook_t *rb = const_cast<sei_rulebook_t *>(rb_c);
    std::lock_guard<std::mutex> lk(rb->mu);
    info[0] = rb->lout->n; info[1] = rb->lin->n; info[2] = rb->size[0] * rb->size[1] * rb->size[2];
    info[3] = info[4] = info[5] = info[6] = info[7] = 0;
    for (int v = 0; v < SEI_NVARIANT; v++)
        if (rb->fwd[v].built) {
            const TiledRules &t = rb->fwd[v];
            info[3] = t.n_rules; info[4] = t.n_sub; info[5] = t.n_tiles; info[6] = t.n_sub; info[7] = t.tile_rows;
            break;
        }
    return 0;
}

int sei_rulebook_build(sei_rulebook_t *rb, int32_t tile_rows, sei_stream_t stream) {
    const TiledRules *t;
    return get_tiled(rb, false, tile_rows, &t, (hipStream_t)stream);
}

int sei_rulebook_export(sei_rulebook_t *rb, int64_t *start_host, int32_t *rules_in_dev, int32_t *rules_out_dev,
                        sei_stream_t stream) {
    const TiledRules *t;
    SEI_TRY(get_any_tiled(rb, false, &t, (hipStream_t)stream));
    return export_rules(*t, start_host, rules_in_dev, rules_out_dev, (hipStream_t)stream);
}

static double conv_bytes(const TiledRules &t, int nIn, int nOut) {
    return 4.0 * t.n_src * nIn + 4.0 * t.n_dst * nOut + 8.0 * t.n_rules + 4.0 * t.K * nIn * nOut;
}

static int rulebook_K(const sei_rulebook_t *rb) { return rb->size[0] * rb->size[1] * rb->size[2]; }

// The MFMA kernels read weights / biases with 16-byte loads.  torch allocations and the parameter arena
// (distributed.py) are aligned; a caller's odd view is staged through an aligned pool block instead of faulting.
struct AlignedArg {
    const float *p = nullptr;
    float *tmp = nullptr;
    int stage(const float *src, size_t count, hipStream_t st) {
        p = src;
        if (!src || ((uintptr_t)src & 15) == 0) return 0;
        SEI_TRY(pool_alloc_t(&tmp, count, st));
        SEI_HIP(hipMemcpyAsync(tmp, src, count * sizeof(float), hipMemcpyDeviceToDevice, st));
        p = tmp;
        return 0;
    }
    ~AlignedArg() { pool_free(tmp); }
};

// forward convolution; stats_out / parts_out (optional): BatchNorm partial sums of Y from the kernel epilogue
static int conv_forward_impl(sei_rulebook_t *rb, const float *X, const float *W, const float *bias, float *Y, int32_t nIn,
                             int32_t nOut, hipStream_t st, double **stats_out, int64_t *parts_out,
                             const ConvEpilogue *ep = nullptr) {
    if (!rb) SEI_FAIL("null rulebook");
    note_stream(st);
    const TiledRules *t;
    const int64_t n_dst = rb->lout ? rb->lout->n : 0;
    const ConvPlan p = plan_conv(nIn, nOut, rulebook_K(rb), n_dst);
    SEI_TRY(get_tiled(rb, false, p.tile_rows, &t, st, true));
    double *stats = nullptr;
    if (stats_out) {
        const int64_t parts = conv_stats_parts(p, *t);
        if (parts > 0) SEI_TRY(pool_alloc_t(&stats, (size_t)parts * 2 * nOut, st));
        *stats_out = stats;
        *parts_out = parts;
    }
    AlignedArg ab, aw;
    if (p.kind != 0) {
        SEI_TRY(ab.stage(bias, nOut, st));
        bias = ab.p;
    }
    const int kind = nIn == 32 && nOut == 32 ? PROF_CONV_FWD_C32 : PROF_CONV_FWD;
    if (p.kind != 1 && p.kind != 4 && !(p.kind == 2 && p.col_group == -2)) {  // natural weight layout: no copy
        if (p.kind >= 2) {
            SEI_TRY(aw.stage(W, (size_t)t->K * nIn * nOut, st));
            W = aw.p;
        }
        ProfScope ps(kind, st, 2.0 * t->n_rules * nIn * nOut, conv_bytes(*t, nIn, nOut));
        return conv_forward(*t, X, W, bias, Y, nIn, nOut, st, nullptr, stats, ep);
    }
    // the workgroup-tile MFMA kernel reads B operands with wide LDS loads from column-permuted weights (tiny copy, <= 4 MB);
    // the column-owner kernel (kind 4, col_group -1) reads them from a copy packed per column tile
    float *Wp;
    SEI_TRY(pool_alloc_t(&Wp, (size_t)t->K * nIn * nOut * (p.split ? 3 : 1), st));   // split: [natural fp32 | three bf16 parts (x 1.5)]
    int rc;
    if (p.split) {
        rc = conv_transform_weights(W, Wp, t->K, nIn, nOut, false, false, 0, 1, st);
        if (!rc) rc = conv_split_weights(Wp, Wp + (size_t)t->K * nIn * nOut, t->K, nIn, nOut, p.nt, st);
    } else {
        rc = conv_transform_weights(W, Wp, t->K, nIn, nOut, false, false, p.col_group, p.blocks, st);
    }
    if (!rc && p.split) {
        ProfScope ps(kind, st, 2.0 * t->n_rules * nIn * nOut, conv_bytes(*t, nIn, nOut));
        rc = conv_forward(*t, X, Wp + (size_t)t->K * nIn * nOut, bias, Y, nIn, nOut, st, nullptr, stats, ep);
    } else if (!rc) {
        ProfScope ps(kind, st, 2.0 * t->n_rules * nIn * nOut, conv_bytes(*t, nIn, nOut));
        rc = conv_forward(*t, X, Wp, bias, Y, nIn, nOut, st, nullptr, stats, ep);
    }
    pool_free(Wp);
    return rc;
}

int sei_conv_forward(sei_rulebook_t *rb, const float *X, const float *W, const float *bias, float *Y, int32_t nIn,
                     int32_t nOut, sei_stream_t stream) {
    return conv_forward_impl(rb, X, W, bias, Y, nIn, nOut, (hipStream_t)stream, nullptr, nullptr);
}

int sei_conv_bn_forward(sei_rulebook_t *rb, const float *X, const float *W, const float *bias, float *Yconv, int32_t nIn,
                        int32_t nOut, const float *gamma, const float *beta, float *running_mean, float *running_var,
                        float eps, float momentum, int32_t training, float leak, const float *residual, float *Ybn,
                        float *save_mean, float *save_invstd, sei_stream_t stream) {
    if (!rb) SEI_FAIL("null rulebook");
    hipStream_t st = (hipStream_t)stream;
    double *stats = nullptr;
    int64_t parts = 0;
    int rc = conv_forward_impl(rb, X, W, bias, Yconv, nIn, nOut, st, training ? &stats : nullptr, &parts);
    if (!rc)
        rc = bn_forward(Yconv, rb->lout ? rb->lout->n : 0, nOut, gamma, beta, running_mean, running_var, eps, momentum,
                        training, leak, residual, Ybn, save_mean, save_invstd, st, stats, parts);
    pool_free(stats);
    return rc;
}

int sei_conv_bn_inference(sei_rulebook_t *rb, const float *X, const float *W, const float *bias, int32_t nIn, int32_t nOut,
                          const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                          float eps, float leak, const float *residual, float *Y, sei_stream_t stream) {
    if (!rb) SEI_FAIL("null rulebook");
    if (!running_mean || !running_var) SEI_FAIL("sei_conv_bn_inference: eval-mode BatchNorm needs running statistics");
    hipStream_t st = (hipStream_t)stream;
    note_stream(st);
    const int64_t n = rb->lout ? rb->lout->n : 0;
    if (n == 0) return 0;
    const TiledRules *t;
    const ConvPlan p = plan_conv(nIn, nOut, rulebook_K(rb), n);
    SEI_TRY(get_tiled(rb, false, p.tile_rows, &t, st, true));
    float *ms;   // [mean | invstd], 16-byte aligned for the epilogue's float4 loads
    const int Cp = (nOut + 3) & ~3;
    SEI_TRY(pool_alloc_t(&ms, 2 * (size_t)Cp, st));
    int rc = bn_eval_stats(running_mean, running_var, eps, nOut, ms, ms + Cp, st);
    const bool fused = conv_epilogue_supported(nIn, nOut, rulebook_K(rb), n, *t) && (((uintptr_t)residual | (uintptr_t)Y) & 15) == 0;
    if (!rc && fused) {
        AlignedArg ag, ab;
        rc = ag.stage(gamma, nOut, st);
        if (!rc) rc = ab.stage(beta, nOut, st);
        ConvEpilogue ep;
        ep.mean = ms; ep.invstd = ms + Cp; ep.gamma = ag.p; ep.beta = ab.p; ep.res = residual; ep.leak = leak;
        if (!rc) rc = conv_forward_impl(rb, X, W, bias, Y, nIn, nOut, st, nullptr, nullptr, &ep);
    } else if (!rc) {
        // plans without the epilogue (scalar kernels of odd plane counts, the opt-in row-band family): two passes
        float *yc;
        rc = pool_alloc_t(&yc, (size_t)n * nOut, st);
        if (!rc) rc = conv_forward_impl(rb, X, W, bias, yc, nIn, nOut, st, nullptr, nullptr);
        if (!rc)
            rc = bn_forward(yc, n, nOut, gamma, beta, const_cast<float *>(running_mean), const_cast<float *>(running_var), eps,
                            0.f, 0, leak, residual, Y, ms, ms + Cp, st, nullptr, 0);
        pool_free(yc);
    }
    pool_free(ms);
    return rc;
}

static int conv_backward_input_impl(sei_rulebook_t *rb, const float *dY, const float *W, const float *dX_init, float *dX,
                                    int32_t nIn, int32_t nOut, sei_stream_t stream) {
    if (!rb) SEI_FAIL("null rulebook");
    hipStream_t st = (hipStream_t)stream;
    note_stream(st);
    const TiledRules *t;
    // roles swap: contraction over nOut, result has nIn columns
    const int64_t n_dst = rb->lin ? rb->lin->n : 0;
    const ConvPlan p = plan_conv(nOut, nIn, rulebook_K(rb), n_dst);
    SEI_TRY(get_tiled(rb, true, p.tile_rows, &t, st));
    if (dX_init && (p.kind == 0 || t->n_sub == 0))
        SEI_FAIL("conv_backward_input_accumulate: not supported for these plane counts");
    float *Wt;
    SEI_TRY(pool_alloc_t(&Wt, (size_t)t->K * nIn * nOut * (p.split ? 3 : 1), st));
    // Wt[k'][co][ci] = W[K-1-k' (submanifold mirror) or k'][ci][co] (kind 1: columns (ci) permuted for the kernel)
    int rc;
    if (p.split) {   // natural layout first, then its three bf16 parts in the kernel's B layout (contraction over nOut, nIn columns)
        rc = conv_transform_weights(W, Wt, t->K, nIn, nOut, true, rb->submanifold, 0, 1, st);
        if (!rc) rc = conv_split_weights(Wt, Wt + (size_t)t->K * nIn * nOut, t->K, nOut, nIn, p.nt, st);
    } else {
        rc = conv_transform_weights(W, Wt, t->K, nIn, nOut, true, rb->submanifold, (p.kind == 1 || p.kind == 4 || (p.kind == 2 && p.col_group == -2)) ? p.col_group : 0,
                                    p.kind == 1 ? p.blocks : 1, st);
    }
    if (!rc && p.split) {
        ProfScope ps(nIn == 32 && nOut == 32 ? PROF_CONV_BWD_IN_C32 : PROF_CONV_BWD_IN, st, 2.0 * t->n_rules * nIn * nOut,
                     conv_bytes(*t, nOut, nIn));
        rc = conv_forward(*t, dY, Wt + (size_t)t->K * nIn * nOut, nullptr, dX, nOut, nIn, st, dX_init);
    } else if (!rc) {
        ProfScope ps(nIn == 32 && nOut == 32 ? PROF_CONV_BWD_IN_C32 : PROF_CONV_BWD_IN, st, 2.0 * t->n_rules * nIn * nOut,
                     conv_bytes(*t, nOut, nIn));
        rc = conv_forward(*t, dY, Wt, nullptr, dX, nOut, nIn, st, dX_init);
    }
    pool_free(Wt);
    return rc;
}

int sei_conv_backward_input(sei_rulebook_t *rb, const float *dY, const float *W, float *dX, int32_t nIn, int32_t nOut,
                            sei_stream_t stream) {
    return conv_backward_input_impl(rb, dY, W, nullptr, dX, nIn, nOut, stream);
}

int sei_conv_backward_input_accumulate(sei_rulebook_t *rb, const float *dY, const float *W, const float *dX_init,
                                       float *dX, int32_t nIn, int32_t nOut, sei_stream_t stream) {
    if (!dX_init) SEI_FAIL("sei_conv_backward_input_accumulate: null dX_init");
    return conv_backward_input_impl(rb, dY, W, dX_init, dX, nIn, nOut, stream);
}

int sei_conv_backward_weight(sei_rulebook_t *rb, const float *X, const float *dY, float *dW, float *dbias, int32_t nIn,
                             int32_t nOut, sei_stream_t stream) {
    if (!rb) SEI_FAIL("null rulebook");
    hipStream_t st = (hipStream_t)stream;
    const TiledRules *t;
    // dW contracts over rule SLOTS, so padding slots are wasted MFMA steps: it walks the 256-row variant (pads to 0.93 where
    // the forward kernels' 64- / 128-row tiles pad to 0.73 / 0.90), cut from the rulebook's cached neighbour table (no second
    // probe): conv_bwd_weight 96.9 -> 93.6 ms per step for ~0.7 ms of extra tiling
    SEI_TRY(get_tiled(rb, false, 256, &t, st));
    {
        ProfScope ps(PROF_CONV_BWD_W, st, 2.0 * t->n_rules * nIn * nOut,
                     4.0 * t->n_src * nIn + 4.0 * t->n_dst * nOut + 8.0 * t->n_rules + 4.0 * t->K * nIn * nOut);
        SEI_TRY(conv_backward_weight(*t, X, dY, dW, nIn, nOut, st));
    }
    if (dbias) SEI_TRY(colsum(dY, t->n_dst, nOut, dbias, st));
    return 0;
}

// dX and dW of one layer are independent.  With sei_set_backward_streams(1) the combined entry point runs dW on a
// library-owned side stream (one per device and caller stream) next to dX on the caller's stream (fork after everything
// already queued on `stream`, join before returning to it), so that the tail of one kernel -- CUs idle while the last
// workgroups finish -- is filled by the other: 361-363 vs 366 ms per step.  Both rule variants are built on the caller's
// stream BEFORE the fork.  Mode 2 (deferred join) leaves the join to the caller (sei_backward_join, once per backward
// pass): the side stream then holds a backlog of dW kernels that also fills the time the caller's stream spends in the
// HBM-bound BatchNorm-backward kernels between two dX kernels.  The caller owns the lifetimes: X, dY and dW of every call
// must stay allocated until the join (the python layer holds the tensors and drops them after it).
// Default: back to back -- concurrent kernels share the GPU, so their individual durations (HIP
// events, rocprof) stop describing the kernels, and the roofline accounting of bench.py is built on those durations.
namespace {
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t e_in = nullptr, e_done = nullptr;
    bool pending = false;   // mode 2: work queued since the last sei_backward_join
};
std::mutex g_side_mu;
std::map<std::pair<int, hipStream_t>, SideStream> g_side;
std::atomic<int> g_bwd_streams{0};

// the side stream paired with (device, caller stream); the mutex covers the lazy creation only
int side_stream_for(hipStream_t st, SideStream *out) {
    int dev = 0;
    SEI_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideStream &ss = g_side[std::make_pair(dev, st)];
    if (!ss.s) {
        // default priority: a more urgent dW stream measured 339.0 vs 336.7 ms per step, a more urgent main stream no change
        SEI_HIP(hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking));
        SEI_HIP(hipEventCreateWithFlags(&ss.e_in, hipEventDisableTiming));
        SEI_HIP(hipEventCreateWithFlags(&ss.e_done, hipEventDisableTiming));
    }
    *out = ss;
    return 0;
}
}  // namespace


int sei_set_product_mode(int32_t mode) {
    if (mode < 0 || mode > 1) SEI_FAIL("sei_set_product_mode: mode must be 0 (fp32 MFMA) or 1 (three-way bf16 split, opt-in)");
    g_product_mode.store(mode);
    return 0;
}

int sei_set_backward_streams(int32_t mode) {
    if (mode < 0 || mode > 2) SEI_FAIL("sei_set_backward_streams: mode must be 0 (back to back), 1 (join per call) or 2 (deferred join)");
    g_bwd_streams.store(mode);
    return 0;
}

int sei_backward_join(sei_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    SEI_HIP(hipGetDevice(&dev));
    // EVERY side stream of the device with un-joined work is joined into `stream`, not only the one paired with it: a join
    // issued under another current stream than the one the backward kernels ran on (a caller that switched streams
    // between backward() and the optimizer) would otherwise find no pair, do nothing, and let the optimizer read dW
    // buffers the side stream has not written yet
    std::lock_guard<std::mutex> lk(g_side_mu);
    for (auto &kv : g_side) {
        SideStream &ss = kv.second;
        if (kv.first.first != dev || !ss.s || !ss.pending) continue;
        SEI_HIP(hipEventRecord(ss.e_done, ss.s));
        SEI_HIP(hipStreamWaitEvent(st, ss.e_done, 0));
        ss.pending = false;
    }
    return 0;
}

int sei_debug_force_dw_kernel(int32_t kind) {
    if (kind < DW_KERNEL_AUTO || kind > DW_KERNEL_RING) SEI_FAIL("sei_debug_force_dw_kernel: kind must be 0..2");
#ifndef SEI_EXPERIMENTS
    if (kind == DW_KERNEL_RING) SEI_FAIL("sei_debug_force_dw_kernel: the LDS-operand dW kernel exists only in experiment builds (make EXTRA=-DSEI_EXPERIMENTS)");
#endif
    g_force_dw.store(kind);
    return 0;
}

int sei_debug_force_conv_kernel(int32_t kind) {
    if (kind < CONV_KERNEL_AUTO || kind > CONV_KERNEL_ROUNDS) SEI_FAIL("sei_debug_force_conv_kernel: kind must be 0..5");
#ifndef SEI_EXPERIMENTS
    if (kind == CONV_KERNEL_BAND) SEI_FAIL("sei_debug_force_conv_kernel: the row-band kernels exist only in experiment builds (make EXTRA=-DSEI_EXPERIMENTS)");
    if (kind == CONV_KERNEL_COLS) SEI_FAIL("sei_debug_force_conv_kernel: the column-owner kernel exists only in experiment builds (make EXTRA=-DSEI_EXPERIMENTS)");
#endif
    g_force_kernel.store(kind);
    return 0;
}

int sei_conv_backward(sei_rulebook_t *rb, const float *X, const float *dY, const float *W, const float *dX_init, float *dX,
                      float *dW, float *dbias, int32_t nIn, int32_t nOut, sei_stream_t stream) {
    if (!rb) SEI_FAIL("null rulebook");
    hipStream_t st = (hipStream_t)stream;
    if (dbias) SEI_TRY(colsum(dY, rb->lout ? rb->lout->n : 0, nOut, dbias, st));
    if (!dX && !dW) return 0;
    if (!dW) return conv_backward_input_impl(rb, dY, W, dX_init, dX, nIn, nOut, stream);
    if (!dX) return sei_conv_backward_weight(rb, X, dY, dW, nullptr, nIn, nOut, stream);
    if (!g_bwd_streams.load(std::memory_order_relaxed)) {   // default: back to back, no stream, no lock
        SEI_TRY(sei_conv_backward_weight(rb, X, dY, dW, nullptr, nIn, nOut, stream));
        return conv_backward_input_impl(rb, dY, W, dX_init, dX, nIn, nOut, stream);
    }
    SideStream ss;
    SEI_TRY(side_stream_for(st, &ss));
    // rule variants of both kernels on the caller's stream (cached after the first call of a level)
    const TiledRules *tw, *tx;
    SEI_TRY(get_tiled(rb, false, 256, &tw, st));
    {
        const int64_t n_dst = rb->lin ? rb->lin->n : 0;
        const ConvPlan p = plan_conv(nOut, nIn, rulebook_K(rb), n_dst);
        SEI_TRY(get_tiled(rb, true, p.tile_rows, &tx, st));
    }
    SEI_HIP(hipEventRecord(ss.e_in, st));
    SEI_HIP(hipStreamWaitEvent(ss.s, ss.e_in, 0));
    int rc1;
    {
        ProfScope ps(PROF_CONV_BWD_W, ss.s, 2.0 * tw->n_rules * nIn * nOut,
                     4.0 * tw->n_src * nIn + 4.0 * tw->n_dst * nOut + 8.0 * tw->n_rules + 4.0 * tw->K * nIn * nOut);
        rc1 = conv_backward_weight(*tw, X, dY, dW, nIn, nOut, ss.s);
    }
    const int rc2 = conv_backward_input_impl(rb, dY, W, dX_init, dX, nIn, nOut, stream);
    if (g_bwd_streams.load(std::memory_order_relaxed) != 2) {   // mode 2: the caller joins (sei_backward_join)
        SEI_HIP(hipEventRecord(ss.e_done, ss.s));
        SEI_HIP(hipStreamWaitEvent(st, ss.e_done, 0));
    } else {
        int dev = 0;
        SEI_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(g_side_mu);
        g_side[std::make_pair(dev, st)].pending = true;
    }
    return rc1 ? rc1 : rc2;
}

int sei_bn_forward(const float *x, int64_t n, int32_t C, const float *gamma, const float *beta, float *running_mean,
                   float *running_var, float eps, float momentum, int32_t training, float leak, const float *residual,
                   float *y, float *save_mean, float *save_invstd, sei_stream_t stream) {
    return bn_forward(x, n, C, gamma, beta, running_mean, running_var, eps, momentum, training, leak, residual, y,
                      save_mean, save_invstd, (hipStream_t)stream);
}

int sei_bn_backward(const float *x, const float *y, const float *dy, int64_t n, int32_t C, const float *gamma,
                    const float *save_mean, const float *save_invstd, float leak, int32_t training, float *dx,
                    float *dresidual, float *dgamma, float *dbeta, sei_stream_t stream) {
    return bn_backward(x, y, dy, n, C, gamma, save_mean, save_invstd, leak, training, dx, dresidual, dgamma, dbeta,
                       (hipStream_t)stream);
}

int sei_bn_backward_colsum(const float *x, const float *y, const float *dy, int64_t n, int32_t C, const float *gamma,
                           const float *beta, const float *save_mean, const float *save_invstd, float leak,
                           int32_t training, float *dx, float *dresidual, float *dgamma, float *dbeta, float *dx_colsum,
                           sei_stream_t stream) {
    if (!y && dresidual) SEI_FAIL("sei_bn_backward_colsum: y may only be NULL when the forward had no residual");
    return bn_backward(x, y, dy, n, C, gamma, save_mean, save_invstd, leak, training, dx, dresidual, dgamma, dbeta,
                       (hipStream_t)stream, dx_colsum, beta);
}

int sei_leaky_forward(const float *x, float *y, int64_t count, float leak, sei_stream_t stream) {
    return leaky_forward(x, y, count, leak, (hipStream_t)stream);
}
int sei_leaky_backward(const float *y, const float *dy, float *dx, int64_t count, float leak, sei_stream_t stream) {
    return leaky_backward(y, dy, dx, count, leak, (hipStream_t)stream);
}
int sei_add_forward(const float *a, const float *b, float *out, int64_t count, sei_stream_t stream) {
    return add_forward(a, b, out, count, 1.f, false, (hipStream_t)stream);
}
int sei_add_leaky_forward(const float *a, const float *b, float *y, int64_t count, float leak, sei_stream_t stream) {
    return add_forward(a, b, y, count, leak, true, (hipStream_t)stream);
}

int sei_sparse_to_dense_forward(sei_metadata_t *m, const int32_t spatial[3], const float *feats, int32_t C, float *dense,
                                sei_stream_t stream) {
    note_stream((hipStream_t)stream);
    std::lock_guard<std::mutex> lk(m->mu);
    Level *l = find_level(m, spatial);
    if (!l) SEI_FAIL("SparseToDense: no level with this spatial size in the metadata");
    return sparse_to_dense(l, m->batch_size, feats, C, dense, false, (hipStream_t)stream);
}
int sei_sparse_to_dense_backward(sei_metadata_t *m, const int32_t spatial[3], const float *ddense, int32_t C,
                                 float *dfeats, sei_stream_t stream) {
    note_stream((hipStream_t)stream);
    std::lock_guard<std::mutex> lk(m->mu);
    Level *l = find_level(m, spatial);
    if (!l) SEI_FAIL("SparseToDense: no level with this spatial size in the metadata");
    return sparse_to_dense(l, m->batch_size, dfeats, C, const_cast<float *>(ddense), true, (hipStream_t)stream);
}

int sei_pool_tanh_forward(sei_metadata_t *m, const int32_t spatial[3], const float *feats, int32_t C, float *pooled,
                          sei_stream_t stream) {
    note_stream((hipStream_t)stream);
    std::lock_guard<std::mutex> lk(m->mu);
    Level *l = find_level(m, spatial);
    if (!l) SEI_FAIL("pool_tanh: no level with this spatial size in the metadata");
    return pool_tanh(l, m->batch_size, feats, C, pooled, nullptr, nullptr, (hipStream_t)stream);
}
int sei_pool_tanh_backward(sei_metadata_t *m, const int32_t spatial[3], const float *feats, const float *dpooled,
                           int32_t C, float *dfeats, sei_stream_t stream) {
    note_stream((hipStream_t)stream);
    std::lock_guard<std::mutex> lk(m->mu);
    Level *l = find_level(m, spatial);
    if (!l) SEI_FAIL("pool_tanh: no level with this spatial size in the metadata");
    return pool_tanh(l, m->batch_size, feats, C, nullptr, dpooled, dfeats, (hipStream_t)stream);
}

int sei_prof_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
    return 0;
}
int sei_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    drain_locked();
    for (int k = 0; k < PROF_NKIND; k++) g_prof.ms[k] = g_prof.flops[k] = g_prof.bytes[k] = g_prof.launches[k] = 0;
    return 0;
}
int sei_prof_query(int32_t kind, double out[4]) {
    if (kind < 0 || kind >= PROF_NKIND) SEI_FAIL("sei_prof_query: bad kind");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    drain_locked();
    out[0] = g_prof.launches[kind]; out[1] = g_prof.ms[kind]; out[2] = g_prof.flops[kind]; out[3] = g_prof.bytes[kind];
    return 0;
}

int sei_debug_sync_count(int64_t *count, int32_t reset) {
    if (count) *count = (int64_t)g_sync_count.load();
    if (reset) g_sync_count.store(0);
    return 0;
}

int sei_pool_stats(int64_t *bytes_held, int64_t *bytes_in_use) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (bytes_held) *bytes_held = (int64_t)g_pool.held;
    if (bytes_in_use) *bytes_in_use = (int64_t)g_pool.in_use;
    return 0;
}
int sei_pool_trim(void) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    reap_quarantine_locked();
    release_cached_locked();
    return 0;
}

}  // extern "C"
