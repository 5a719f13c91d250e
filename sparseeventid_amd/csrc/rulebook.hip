// Rulebook construction on the device (SCN `getSubmanifoldRuleBook` / `getRuleBook`):
// hash probes -> tile-blocked neighbour table -> per-tile, per-offset compacted sub-chunks of 16 rules.
// Integer / HBM-MALL bound; the sub-chunk layout is what the MFMA conv kernels consume
// ("per-wavefront rulebook segments").
#include "common.h"

#include <stdlib.h>

namespace sei {

// One thread per (k, dst row): nbr[k * n_dst + row] = src row feeding dst row `row` through offset k, or -1 (k-major:
// independent of any tile size, so every tiled variant of a rulebook is cut from ONE probe).  Consecutive lanes =
// consecutive dst rows, same offset: coalesced coord reads / nbr writes, and the probed keys are
// spatial neighbours of each other.
// KP = offsets actually probed.  Submanifold rulebooks (mode 0, odd filter) probe only the first half of the
// offsets plus the centre: a hit (i -> o) at offset k is also the rule (o -> i) at the mirrored offset K-1-k, so
// it is written to both places (the table is preset to -1) -- half the hash probes, which dominate this kernel
// (92 % of the stem's 125 probes per site are misses).
constexpr int PROBE_GROUP = 4;  // consecutive offsets (neighbours along x2: one filter word, one table line) per thread

__global__ void probe_kernel(const int32_t *__restrict__ dst_coords, int64_t n_dst, int K, int KP, ProbeGeom g,
                             const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t mask,
                             const uint32_t *__restrict__ filter, uint64_t fmask, int32_t *__restrict__ nbr) {
    // workgroup b = (row chunk b / KG, offset group b % KG): consecutive workgroups probe the SAME 256 sites under the
    // following offsets, so their hash-table lines are still warm (one offset over all sites, then the next, measured 12 %
    // slower).  [r3] a thread probes PROBE_GROUP consecutive offsets of its site: the coordinates are read once per group
    // instead of once per offset, and the group's filter tests are independent loads in flight together.
    const int KG = (KP + PROBE_GROUP - 1) / PROBE_GROUP;
    const int k0 = (int)(blockIdx.x % (unsigned)KG) * PROBE_GROUP;
    const int64_t row = (int64_t)(blockIdx.x / (unsigned)KG) * blockDim.x + threadIdx.x;
    if (row >= n_dst) return;
    const int4 c = reinterpret_cast<const int4 *>(dst_coords)[row];
    uint64_t key[PROBE_GROUP];
    bool ok[PROBE_GROUP];
#pragma unroll
    for (int j = 0; j < PROBE_GROUP; j++) {
        const int k = k0 + j;
        int a = k / (g.f1 * g.f2), rem = k - a * (g.f1 * g.f2);
        int b = rem / g.f2, d = rem - b * g.f2;
        int q0, q1, q2;
        bool v = k < KP;
        if (g.mode == 0) {
            q0 = c.x - g.f0 / 2 + a; q1 = c.y - g.f1 / 2 + b; q2 = c.z - g.f2 / 2 + d;
        } else if (g.mode == 1) {
            q0 = c.x * g.t0 + a; q1 = c.y * g.t1 + b; q2 = c.z * g.t2 + d;
        } else {
            int p0 = c.x - a, p1 = c.y - b, p2 = c.z - d;
            v = v && p0 >= 0 && p1 >= 0 && p2 >= 0 && (p0 % g.t0) == 0 && (p1 % g.t1) == 0 && (p2 % g.t2) == 0;
            q0 = p0 / g.t0; q1 = p1 / g.t1; q2 = p2 / g.t2;
        }
        v = v && q0 >= 0 && q0 < g.s0 && q1 >= 0 && q1 < g.s1 && q2 >= 0 && q2 < g.s2;
        key[j] = pack_key(v ? q0 : 0, v ? q1 : 0, v ? q2 : 0, c.w);
        ok[j] = v;
    }
    // membership filter first (common.h): most probes are misses and end at one bitmap word; all tests of the group are
    // issued before the first table probe
#ifndef SEI_PROBE_NOFILTER
#pragma unroll
    for (int j = 0; j < PROBE_GROUP; j++) ok[j] = ok[j] && filter_test(filter, fmask, key[j]);
#endif
#pragma unroll
    for (int j = 0; j < PROBE_GROUP; j++) {
        const int k = k0 + j;
        if (k >= KP) break;
        const int res = ok[j] ? hash_find(keys, vals, mask, key[j]) : -1;
        nbr[(int64_t)k * n_dst + row] = res;  // the probed half (and every offset without mirroring) is written whole
        if (KP != K && res >= 0 && k != K - 1 - k)  // mirrored rule: dst row `res` receives from `row` through offset K-1-k
            nbr[(int64_t)(K - 1 - k) * n_dst + res] = (int32_t)row;
    }
}

// Offsets of a filter (f0,f1,f2) inside the neighbour table of a filter (F0,F1,F2) >= it, both odd and centred
// (identity when they are equal).
struct NbrView {
    int f0, f1, f2, F0, F1, F2;
    __host__ __device__ int table_k(int k) const {
        if (f0 == F0 && f1 == F1 && f2 == F2) return k;
        const int a = k / (f1 * f2), r = k - a * (f1 * f2), b = r / f2, d = r - b * f2;
        return ((a + (F0 - f0) / 2) * F1 + (b + (F1 - f1) / 2)) * F2 + (d + (F2 - f2) / 2);
    }
};

constexpr int RULE_STRIPES = 128;

template <bool FILL>
__global__ void tile_rules_kernel(const int32_t *__restrict__ nbr, NbrView vw, int64_t n_dst, int K, int TR, int n_tiles,
                                  int32_t *__restrict__ tile_subs, int32_t *__restrict__ tile_seg,
                                  int32_t *__restrict__ seg_start, unsigned long long *__restrict__ totals,
                                  const int32_t *__restrict__ tile_sub_start, int32_t *__restrict__ sub_src,
                                  uint8_t *__restrict__ sub_dst) {
    extern __shared__ int cnt_k[];  // [K] rules per offset (count pass only)
    const int tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = TR >> 6;
    const int64_t row0 = (int64_t)tile * TR;
    auto nb_at = [&](int k, int r) {  // neighbour of tile row r under offset k (-1 beyond the last row)
        return row0 + r < n_dst ? nbr[(int64_t)vw.table_k(k) * n_dst + row0 + r] : -1;
    };
    if (!FILL) {
        for (int k = wave; k < K; k += nw) {
            // all chunks of the offset are loaded before the first ballot (one dependent load per iteration left this pass
            // latency-bound at ~1.9 TB/s of neighbour-table reads)
            int v[4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++) v[ch] = ch < nw ? nb_at(k, ch * 64 + lane) : -1;
            int c = 0;
#pragma unroll
            for (int ch = 0; ch < 4; ch++) c += __popcll(__ballot(v[ch] >= 0));
            if (lane == 0) cnt_k[k] = c;
        }
        __syncthreads();
        if (wave == 0) {
            int run = 0, rules = 0;
            for (int k0 = 0; k0 < K; k0 += 64) {
                const int k = k0 + lane;
                const int c = k < K ? cnt_k[k] : 0;
                const int ns = (c + SUB_SLOTS - 1) / SUB_SLOTS;
                int incl = ns, csum = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int v = __shfl_up(incl, d), w = __shfl_up(csum, d);
                    if (lane >= d) { incl += v; csum += w; }
                }
                if (k < K) {
                    const int start = run + incl - ns;
                    tile_seg[(int64_t)tile * (K + 1) + k] = start;
                }
                run += __shfl(incl, 63);
                rules += __shfl(csum, 63);
            }
            if (lane == 0) {
                tile_seg[(int64_t)tile * (K + 1) + K] = run;
                tile_subs[tile] = run;
                // striped over RULE_STRIPES addresses: ~100 000 tiles per step adding to ONE address serialise at the L2
                // (the count pass took 180 us per launch, more than the fill pass that writes the rules)
                atomicAdd(&totals[4 + (tile & (RULE_STRIPES - 1))], (unsigned long long)rules);
            }
        }
    } else {
        const int64_t base_sub = tile_sub_start[tile];
        for (int k = wave; k < K; k += nw) {
            const int64_t slot0 = (base_sub + tile_seg[(int64_t)tile * (K + 1) + k]) * SUB_SLOTS;
            int run = 0;
            int v[4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++) v[ch] = ch < nw ? nb_at(k, ch * 64 + lane) : -1;  // loads first, then the dependent part
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                if (ch >= nw) break;
                const int src = v[ch];
                const bool hit = src >= 0;
                const unsigned long long bal = __ballot(hit);
                if (hit) {
                    const int rank = run + __popcll(bal & ((1ull << lane) - 1ull));
                    sub_src[slot0 + rank] = src;
                    sub_dst[slot0 + rank] = (uint8_t)(ch * 64 + lane);
                }
                run += __popcll(bal);
            }
            const int pad = ((run + SUB_SLOTS - 1) / SUB_SLOTS) * SUB_SLOTS - run;
            if (lane < pad) {
                sub_src[slot0 + run + lane] = -1;
                sub_dst[slot0 + run + lane] = 0;
            }
        }
    }
}

// seg_start[k][tile] = tile_seg[tile][k]: the k-major copy the dW kernel walks (one k over many tiles).  Written through
// an LDS tile so that both sides are coalesced (the count pass used to scatter K 4-byte values per tile).
__global__ __launch_bounds__(256) void transpose_seg_kernel(const int32_t *__restrict__ tile_seg, int32_t *__restrict__ seg_start,
                                                            int n_tiles, int K1) {
    __shared__ int t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tile0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int tile = tile0 + j, k = k0 + tx;
        if (tile < n_tiles && k < K1) t[j][tx] = tile_seg[(int64_t)tile * K1 + k];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int k = k0 + j, tile = tile0 + tx;
        if (tile < n_tiles && k < K1) seg_start[(int64_t)k * n_tiles + tile] = t[tx][j];
    }
}

void free_tiled(TiledRules &t) {
    pool_free(t.tile_sub_start); pool_free(t.tile_seg); pool_free(t.seg_start); pool_free(t.sub_src);
    pool_free(t.sub_dst);
    t = TiledRules();
}

// dst rows [n_dst] with coords dst_coords probe the src-level hash.  *nbr_cache: the rulebook's k-major neighbour table,
// probed by the first variant that is built and kept for the later ones (dW cuts its own 256-row variant from it) until the
// rulebook is destroyed.
//
// A tiled variant is built in two halves around ONE device -> host read-back (the sub-chunk total sizes the rule arrays):
//   begin:  probe (if the table does not exist yet), count pass, k-major transpose, scan, totals -> pinned host memory
//   finish: (after the stream has been synchronised) allocate the rule arrays, fill pass.
// get_tiled() runs both with a synchronise of its own in between; sei_metadata_prebuild_like() begins EVERY variant a
// template lists, synchronises once, and finishes them all -- one read-back per batch instead of one per variant.
struct PendingTiled {
    TiledRules t;
    TiledRules *dest = nullptr;
    const int32_t *nbr = nullptr;
    NbrView vw{};
    int32_t *tile_subs = nullptr;
    unsigned long long *totals = nullptr;   // device: [2] = sub-chunk total (scan), [4..] = striped rule counts
    unsigned long long *host = nullptr;     // pinned: the same words after the read-back
    bool empty = false;
};

constexpr int TOTAL_WORDS = 4 + RULE_STRIPES;

// pinned staging for the read-backs: blocks of TOTAL_WORDS words, recycled
namespace {
std::mutex g_stage_mu;
std::vector<unsigned long long *> g_stage_free;
}  // namespace
unsigned long long *stage_get() {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    if (!g_stage_free.empty()) {
        unsigned long long *p = g_stage_free.back();
        g_stage_free.pop_back();
        return p;
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, TOTAL_WORDS * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return (unsigned long long *)p;
}
void stage_put(unsigned long long *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_stage_mu);
    g_stage_free.push_back(p);
}

static int build_tiled_begin(const int32_t *dst_coords, int64_t n_dst, int64_t n_src, const HashTable &src_ht, int K,
                             ProbeGeom g, int tile_rows, PendingTiled *pd, int32_t **nbr_cache, hipStream_t st,
                             const int32_t *donor_nbr = nullptr, const int32_t *donor_size = nullptr) {
    if (K > 4096) SEI_FAIL("rulebook: filter volume too large");
    TiledRules &t = pd->t;
    t = TiledRules();
    t.n_dst = n_dst; t.n_src = n_src; t.K = K; t.tile_rows = tile_rows;
    t.n_tiles = (int)((n_dst + tile_rows - 1) / tile_rows);
    int nt1 = t.n_tiles > 0 ? t.n_tiles : 1;
    SEI_TRY(pool_alloc_t(&t.tile_sub_start, nt1 + 1, st));
    SEI_TRY(pool_alloc_t(&t.tile_seg, (int64_t)(K + 1) * nt1, st));
    SEI_TRY(pool_alloc_t(&t.seg_start, (int64_t)(K + 1) * nt1, st));
    if (n_dst == 0) {
        SEI_HIP(hipMemsetAsync(t.tile_sub_start, 0, 2 * sizeof(int32_t), st));
        SEI_TRY(pool_alloc_t(&t.sub_src, 16, st)); SEI_TRY(pool_alloc_t(&t.sub_dst, 16, st));
        t.built = true;
        pd->empty = true;
        return 0;
    }
    double bytes = 12.0 * n_dst + 8.0 * K * n_dst;
    ProfScope ps(PROF_RULES, st, 0.0, bytes);
    SEI_TRY(pool_alloc_t(&pd->tile_subs, t.n_tiles + 1, st));
    SEI_TRY(pool_alloc_t(&pd->totals, TOTAL_WORDS, st));
    pd->host = stage_get();
    if (!pd->host) SEI_FAIL("rulebook: cannot allocate pinned staging memory");
    SEI_HIP(hipMemsetAsync(pd->totals, 0, TOTAL_WORDS * sizeof(unsigned long long), st));
    NbrView vw{g.f0, g.f1, g.f2, g.f0, g.f1, g.f2};
    if (!*nbr_cache && donor_nbr) {
        // the offsets of this filter are a sub-set of the donor's table (same level, larger centred odd filter)
        vw.F0 = donor_size[0]; vw.F1 = donor_size[1]; vw.F2 = donor_size[2];
    } else if (!*nbr_cache) {
        int32_t *nbr;
        const int64_t tot = (int64_t)K * n_dst;
        SEI_TRY(pool_alloc_t(&nbr, tot, st));
        // submanifold: dst level == src level and the filter is odd in every dimension, so rules come in mirrored pairs
        const bool mirror = g.mode == 0 && (g.f0 & 1) && (g.f1 & 1) && (g.f2 & 1) && K > 1;
        const int KP = mirror ? (K + 1) / 2 : K;
        // only the mirrored half (offsets KP .. K-1) is filled by scattered writes and needs the -1 preset
        if (mirror) SEI_HIP(hipMemsetAsync(nbr + (size_t)KP * n_dst, 0xFF, (size_t)(K - KP) * n_dst * sizeof(int32_t), st));
        const int64_t chunks = (n_dst + 255) / 256;
        const int KG = (KP + PROBE_GROUP - 1) / PROBE_GROUP;
        if (chunks * KG >= (1ll << 31)) SEI_FAIL("rulebook: too many (site, offset) pairs for one launch");
        probe_kernel<<<(unsigned)(chunks * KG), 256, 0, st>>>(dst_coords, n_dst, K, KP, g, src_ht.keys, src_ht.vals,
                                                             src_ht.mask, src_ht.filter, src_ht.fmask, nbr);
        SEI_LAUNCH_CHECK();
        *nbr_cache = nbr;
    }
    pd->nbr = *nbr_cache ? *nbr_cache : donor_nbr;
    pd->vw = vw;
    tile_rules_kernel<false><<<t.n_tiles, tile_rows, K * sizeof(int), st>>>(pd->nbr, vw, n_dst, K, tile_rows, t.n_tiles, pd->tile_subs,
                                                                            t.tile_seg, t.seg_start, pd->totals, nullptr, nullptr,
                                                                            nullptr);
    transpose_seg_kernel<<<dim3((t.n_tiles + 31) / 32, (K + 1 + 31) / 32), 256, 0, st>>>(t.tile_seg, t.seg_start, t.n_tiles, K + 1);
    SEI_LAUNCH_CHECK();
    SEI_TRY(exclusive_scan_i32(pd->tile_subs, t.tile_sub_start, t.n_tiles, (int32_t *)(pd->totals + 2), st));
    SEI_HIP(hipMemcpyAsync(t.tile_sub_start + t.n_tiles, pd->totals + 2, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    SEI_HIP(hipMemcpyAsync(pd->host, pd->totals, TOTAL_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    return 0;
}

// after the stream has been synchronised past build_tiled_begin's read-back
static int build_tiled_finish(PendingTiled *pd, hipStream_t st) {
    TiledRules &t = pd->t;
    if (!pd->empty) {
        const unsigned long long *host = pd->host;
        t.n_rules = 0;
        for (int i = 0; i < RULE_STRIPES; i++) t.n_rules += (int64_t)host[4 + i];
        t.n_sub = (int64_t)(uint32_t)host[2];
        int64_t ns = t.n_sub > 0 ? t.n_sub : 1;
        SEI_TRY(pool_alloc_t(&t.sub_src, (ns + 1) * SUB_SLOTS, st));  // +1: the kernels' clamped index prefetch may touch the sub-chunk after the last
        SEI_TRY(pool_alloc_t(&t.sub_dst, (ns + 1) * SUB_SLOTS, st));
        ProfScope ps(PROF_RULES, st, 0.0, 8.0 * t.n_rules);
        tile_rules_kernel<true><<<t.n_tiles, t.tile_rows, 0, st>>>(pd->nbr, pd->vw, t.n_dst, t.K, t.tile_rows, t.n_tiles, nullptr,
                                                                   t.tile_seg, nullptr, nullptr, t.tile_sub_start, t.sub_src,
                                                                   t.sub_dst);
        SEI_LAUNCH_CHECK();
        pool_free(pd->tile_subs); pool_free(pd->totals);
        stage_put(pd->host);
        pd->tile_subs = nullptr; pd->totals = nullptr; pd->host = nullptr;
        t.built = true;
    }
    *pd->dest = t;
    return 0;
}

static void pending_abort(PendingTiled *pd) {
    free_tiled(pd->t);
    pool_free(pd->tile_subs); pool_free(pd->totals);
    stage_put(pd->host);
}

// begins one variant of a rulebook (rb->mu held by the caller); nothing to do if it is built already
static int tiled_begin_locked(sei_rulebook *rb, bool transposed, int tile_rows, PendingTiled *pd, bool *needed, hipStream_t st) {
    if (rb->submanifold) transposed = false;
    TiledRules &t = (transposed ? rb->bwd : rb->fwd)[variant_of(tile_rows)];
    // a variant another caller has begun (build_tiled_batch releases rb->mu between its halves) is not begun again
    *needed = !t.built && !t.building;
    if (!*needed) return 0;
    pd->dest = &t;
    int K = rb->size[0] * rb->size[1] * rb->size[2];
    ProbeGeom g{rb->submanifold ? 0 : 1, rb->size[0], rb->size[1], rb->size[2], rb->stride[0], rb->stride[1],
                rb->stride[2], rb->lin->spatial[0], rb->lin->spatial[1], rb->lin->spatial[2]};
    if (!transposed) {
        // dst = output sites probing the input-level hash
        const int32_t *dn = nullptr, *ds = nullptr;
        if (rb->submanifold && !rb->nbr_fwd && rb->donor) {
            std::lock_guard<std::mutex> ld(rb->donor->mu);
            dn = rb->donor->nbr_fwd;  // stays valid: rulebooks of one metadata are destroyed together
            ds = rb->donor->size;
        }
        const int rc = build_tiled_begin(rb->lout->coords, rb->lout->n, rb->lin->n, rb->lin->ht, K, g, tile_rows, pd, &rb->nbr_fwd, st, dn, ds);
        t.building = rc == 0;
        return rc;
    }
    // dst = input sites probing the output-level hash
    g.mode = 2;
    g.s0 = rb->lout->spatial[0]; g.s1 = rb->lout->spatial[1]; g.s2 = rb->lout->spatial[2];
    const int rc = build_tiled_begin(rb->lin->coords, rb->lin->n, rb->lout->n, rb->lout->ht, K, g, tile_rows, pd, &rb->nbr_bwd, st);
    t.building = rc == 0;
    return rc;
}

int get_tiled(sei_rulebook *rb, bool transposed, int tile_rows, const TiledRules **out, hipStream_t st, bool forward_use) {
    if (tile_rows < 64 || tile_rows > 256 || tile_rows % 64) SEI_FAIL("rulebook: tile_rows must be 64, 128, 192 or 256");
    std::unique_lock<std::mutex> lk(rb->mu);
    if (rb->submanifold) transposed = false;
    TiledRules &t = (transposed ? rb->bwd : rb->fwd)[variant_of(tile_rows)];
    // a batch build of another thread (prefetch) holds this variant between its two halves: wait for it instead of
    // starting a second build that would write the same slot
    rb->cv.wait(lk, [&] { return !t.building; });
    if (!t.built) {
        PendingTiled pd;
        bool needed = false;
        int rc = tiled_begin_locked(rb, transposed, tile_rows, &pd, &needed, st);
        if (!rc && !pd.empty && sync_stream(st) != hipSuccess) {
            set_error("rulebook: hipStreamSynchronize failed");
            rc = 1;
        }
        if (!rc) rc = build_tiled_finish(&pd, st);
        if (rc) {
            pending_abort(&pd);
            t.building = false;
            return rc;
        }
    }
    if (forward_use) t.fwd_use = true;
    *out = &t;
    return 0;
}

// Every variant in `want` (rulebook, transposed, tile_rows) that is not built yet: begun back to back, ONE stream
// synchronisation, then finished.  (sei_metadata_prebuild_like)
int build_tiled_batch(const std::vector<TiledRequest> &want, hipStream_t st) {
    std::vector<PendingTiled> pend(want.size());
    std::vector<char> live(want.size(), 0);
    int rc = 0;
    bool any = false;
    for (size_t i = 0; i < want.size() && !rc; i++) {
        std::lock_guard<std::mutex> lk(want[i].rb->mu);
        bool needed = false;
        rc = tiled_begin_locked(want[i].rb, want[i].transposed, want[i].tile_rows, &pend[i], &needed, st);
        live[i] = needed ? 1 : 0;
        any = any || (needed && !pend[i].empty);
        if (rc && needed) { pending_abort(&pend[i]); live[i] = 0; }   // begin failed: `building` was never set
    }
    if (!rc && any && sync_stream(st) != hipSuccess) {
        set_error("rulebook: hipStreamSynchronize failed");
        rc = 1;
    }
    for (size_t i = 0; i < want.size(); i++) {
        if (!live[i]) continue;
        std::lock_guard<std::mutex> lk(want[i].rb->mu);
        bool done = false;
        if (!rc) {
            rc = build_tiled_finish(&pend[i], st);
            done = rc == 0;
            if (done && want[i].forward_use) pend[i].dest->fwd_use = true;
        }
        if (!done) {  // this item's finish failed, or an earlier item did: release what its two halves hold
            pending_abort(&pend[i]);
            pend[i].dest->building = false;
        }
        want[i].rb->cv.notify_all();
    }
    return rc;
}

int get_any_tiled(sei_rulebook *rb, bool transposed, const TiledRules **out, hipStream_t st) {
    {
        std::lock_guard<std::mutex> lk(rb->mu);
        if (rb->submanifold) transposed = false;
        TiledRules *v = transposed ? rb->bwd : rb->fwd;
        for (int i = 0; i < SEI_NVARIANT; i++)
            if (v[i].built) {
                *out = &v[i];
                return 0;
            }
    }
    return get_tiled(rb, transposed, 128, out, st);
}

// ------------------------------------------------------------------ strided conv: output-site discovery
// candidate j of input i = j-th output site (row-major over the per-dim candidate ranges) covering i.
__device__ inline bool candidate(int4 p, int j, const ProbeGeom &g, int o0, int o1, int o2, int m1, int m2, int *q0,
                                 int *q1, int *q2) {
    int j0 = j / (m1 * m2), r = j - j0 * (m1 * m2);
    int j1 = r / m2, j2 = r - j1 * m2;
    int l0 = p.x - g.f0 + 1, l1 = p.y - g.f1 + 1, l2 = p.z - g.f2 + 1;
    int lo0 = l0 <= 0 ? 0 : (l0 + g.t0 - 1) / g.t0, lo1 = l1 <= 0 ? 0 : (l1 + g.t1 - 1) / g.t1,
        lo2 = l2 <= 0 ? 0 : (l2 + g.t2 - 1) / g.t2;
    int h0 = min(p.x / g.t0, o0 - 1), h1 = min(p.y / g.t1, o1 - 1), h2 = min(p.z / g.t2, o2 - 1);
    *q0 = lo0 + j0; *q1 = lo1 + j1; *q2 = lo2 + j2;
    return *q0 <= h0 && *q1 <= h1 && *q2 <= h2;
}

// `n_dev` (optional): the row count of the input level as a DEVICE value -- the output-site count of the strided level in
// front of this one when a chain of levels is discovered without a host read-back in between (`n` is then an upper bound that
// sized the grid and the arrays)
__global__ void out_insert_kernel(const int32_t *__restrict__ in_coords, int64_t n, const int32_t *__restrict__ n_dev, int mult,
                                  int m1, int m2, ProbeGeom g, int o0, int o1, int o2, uint64_t *keys, uint32_t *vals,
                                  uint64_t mask, uint32_t *filter, uint64_t fmask, uint32_t *slot_of, int32_t *err) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * mult) return;
    if (n_dev && t >= (int64_t)*n_dev * mult) {   // beyond the real rows: no candidate
        slot_of[t] = 0xFFFFFFFFu;
        return;
    }
    int64_t i = t / mult;
    int j = (int)(t - i * mult);
    int4 p = reinterpret_cast<const int4 *>(in_coords)[i];
    int q0, q1, q2;
    if (!candidate(p, j, g, o0, o1, o2, m1, m2, &q0, &q1, &q2)) {
        slot_of[t] = 0xFFFFFFFFu;
        return;
    }
    const uint64_t key = pack_key(q0, q1, q2, p.w);
    const uint32_t s = hash_insert_min(keys, vals, mask, key, (uint32_t)t);
    filter_set(filter, fmask, key);
    if (s == 0xFFFFFFFFu) *err = 1;  // table full (cannot happen at load <= 0.5)
    slot_of[t] = s;
}

__global__ void out_flag_kernel(const uint32_t *slot_of, const uint32_t *vals, int64_t n, int32_t *flag) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t s = slot_of[t];
    flag[t] = (s != 0xFFFFFFFFu && vals[s] == (uint32_t)t) ? 1 : 0;
}

__global__ void out_assign_kernel(const int32_t *__restrict__ in_coords, int64_t n, int mult, int m1, int m2,
                                  ProbeGeom g, int o0, int o1, int o2, const uint32_t *slot_of, const uint32_t *vals,
                                  const int32_t *pos, int32_t *out_coords, uint32_t *new_vals) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * mult) return;   // candidates beyond the real rows carry slot_of = ~0 (out_insert_kernel) and fall out below
    uint32_t s = slot_of[t];
    if (s == 0xFFFFFFFFu || vals[s] != (uint32_t)t) return;
    int64_t i = t / mult;
    int j = (int)(t - i * mult);
    int4 p = reinterpret_cast<const int4 *>(in_coords)[i];
    int q0, q1, q2;
    candidate(p, j, g, o0, o1, o2, m1, m2, &q0, &q1, &q2);
    int r = pos[t];
    reinterpret_cast<int4 *>(out_coords)[r] = make_int4(q0, q1, q2, p.w);
    new_vals[s] = (uint32_t)r;
}

Level *find_level(sei_metadata *m, const int32_t spatial[3]) {
    for (Level *l : m->levels)
        if (l->spatial[0] == spatial[0] && l->spatial[1] == spatial[1] && l->spatial[2] == spatial[2]) return l;
    return nullptr;
}


// Discovers the output active set of a strided convolution and registers it as a new level -- in two halves around the ONE
// device -> host read-back of the site count:
//   begin : tables and arrays sized for an UPPER BOUND of the input rows (every input row yields at most `mult` candidates),
//           candidates inserted, first-child order by a scan, coordinates assigned; the count stays on the device.  If the
//           input level was itself begun and not finished (`lin->n_dev`), its kernels' row count is read from the device: a
//           CHAIN of levels is discovered back to back with no host synchronisation in between
//           (sei_metadata_prebuild_like: one read-back for all strided levels of a batch instead of one per level).
//   finish: after the stream has been synchronised past the begin's read-back: the count becomes the level's host row count.
// The site set, the row order and therefore every rule list are the same whatever the table capacity was.
int build_strided_level_begin(sei_metadata *m, Level *lin, const int32_t size[3], const int32_t stride[3], sei_rulebook *rb,
                              hipStream_t st, PendingLevel *pl) {
    int32_t osz[3];
    for (int d = 0; d < 3; d++) {
        if (size[d] < 1 || stride[d] < 1) SEI_FAIL("Convolution: bad filter size / stride");
        osz[d] = (lin->spatial[d] - size[d]) / stride[d] + 1;
        if (osz[d] < 1) SEI_FAIL("Convolution: input spatial size smaller than filter");
        rb->spatial_out[d] = osz[d];
    }
    if (find_level(m, osz)) SEI_FAIL("Convolution: a level with the output spatial size already exists in this metadata");
    int m0 = (size[0] + stride[0] - 1) / stride[0], m1 = (size[1] + stride[1] - 1) / stride[1],
        m2 = (size[2] + stride[2] - 1) / stride[2];
    int mult = m0 * m1 * m2;
    const int64_t n = lin->n_dev ? lin->n_ub : lin->n;   // rows of the input level, or their upper bound
    ProbeGeom g{1, size[0], size[1], size[2], stride[0], stride[1], stride[2], lin->spatial[0], lin->spatial[1],
                lin->spatial[2]};
    ProfScope ps(PROF_RULES, st, 0.0, 12.0 * n * mult);
    Level *lo = new Level();
    for (int d = 0; d < 3; d++) lo->spatial[d] = osz[d];
    int64_t cand = n * mult;
    if (cand >= (1ll << 31) - 64) { delete lo; SEI_FAIL("Convolution: too many candidate output sites"); }
    int rc = alloc_table(&lo->ht, cand, st);
    *pl = PendingLevel();
    pl->level = lo;
    if (!rc && n == 0) {
        lo->n = 0;
        rc = pool_alloc_t(&lo->coords, 4, st);
    } else if (!rc) {
        uint32_t *slot_of = nullptr, *new_vals = nullptr;
        int32_t *flag = nullptr;
        rc = pool_alloc_t(&slot_of, cand, st);
        if (!rc) rc = pool_alloc_t(&flag, cand, st);
        if (!rc) rc = pool_alloc_t(&pl->total, 2, st);  // [0] number of output sites, [1] table-full flag
        if (!rc) rc = pool_alloc_t(&new_vals, lo->ht.mask + 1, st);
        if (!rc) rc = pool_alloc_t(&lo->coords, 4 * (cand > 0 ? cand : 1), st);   // <= cand output sites
        if (!rc) {
            pl->host = stage_get();
            if (!pl->host) { set_error("rulebook: cannot allocate pinned staging memory"); rc = 1; }
        }
        if (!rc) {
            hipError_t e = hipMemsetAsync(pl->total, 0, 2 * sizeof(int32_t), st);
            if (e == hipSuccess) e = hipMemsetAsync(new_vals, 0xFF, (lo->ht.mask + 1) * sizeof(uint32_t), st);
            if (e != hipSuccess) { set_error(std::string("hipMemsetAsync: ") + hipGetErrorString(e)); rc = 1; }
        }
        if (!rc) {
            unsigned nb = (unsigned)((cand + 255) / 256);
            out_insert_kernel<<<nb, 256, 0, st>>>(lin->coords, n, lin->n_dev, mult, m1, m2, g, osz[0], osz[1], osz[2], lo->ht.keys,
                                                  lo->ht.vals, lo->ht.mask, lo->ht.filter, lo->ht.fmask, slot_of, pl->total + 1);
            out_flag_kernel<<<nb, 256, 0, st>>>(slot_of, lo->ht.vals, cand, flag);
            rc = exclusive_scan_i32(flag, flag, cand, pl->total, st);
            if (!rc) {
                out_assign_kernel<<<nb, 256, 0, st>>>(lin->coords, n, mult, m1, m2, g, osz[0], osz[1], osz[2], slot_of,
                                                      lo->ht.vals, flag, lo->coords, new_vals);
                if (hipGetLastError() != hipSuccess) { set_error("rulebook: launch failed (strided level)"); rc = 1; }
            }
            if (!rc && hipMemcpyAsync(pl->host, pl->total, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess) {
                set_error("rulebook: read-back of the site count failed"); rc = 1;
            }
        }
        pool_free(slot_of); pool_free(flag);
        if (!rc) {
            pool_free(lo->ht.vals);
            lo->ht.vals = new_vals;
            lo->n_dev = pl->total;   // a level begun behind this one reads its row count from here
            lo->n_ub = cand;
            lo->n = -1;              // unknown on the host until build_strided_level_finish
            pl->pending = true;
        } else {
            pool_free(new_vals);
        }
    }
    if (rc) {
        pool_free(lo->coords); pool_free(lo->ht.keys); pool_free(lo->ht.vals); pool_free(lo->ht.filter);
        pool_free(pl->total); stage_put(pl->host);
        delete lo;
        *pl = PendingLevel();
        return rc;
    }
    m->levels.push_back(lo);
    rb->lin = lin;
    rb->lout = lo;
    return 0;
}

int build_strided_level_finish(PendingLevel *pl) {
    if (!pl->pending) return 0;
    Level *lo = pl->level;
    const int32_t *h = reinterpret_cast<const int32_t *>(pl->host);
    const int32_t total = h[0], full = h[1];
    lo->n = total;
    lo->n_dev = nullptr;
    pool_free(pl->total);
    stage_put(pl->host);
    pl->total = nullptr; pl->host = nullptr; pl->pending = false;
    if (full) {
        lo->failed = true;   // the level and its rulebook are registered already: lookups must not hand them out (round-4 advisor)
        lo->n = 0;
        SEI_FAIL("Convolution: hash table full");
    }
    return 0;
}

void build_strided_level_abort(PendingLevel *pl) {
    if (!pl->pending) return;
    Level *lo = pl->level;
    lo->failed = true;
    lo->n = 0;
    lo->n_dev = nullptr;
    pool_free(pl->total);
    stage_put(pl->host);
    pl->total = nullptr; pl->host = nullptr; pl->pending = false;
}

int build_strided_level(sei_metadata *m, Level *lin, const int32_t size[3], const int32_t stride[3], sei_rulebook *rb,
                        hipStream_t st) {
    PendingLevel pl;
    SEI_TRY(build_strided_level_begin(m, lin, size, stride, rb, st, &pl));
    if (pl.pending && sync_stream(st) != hipSuccess) {
        build_strided_level_abort(&pl);
        SEI_FAIL("rulebook: hipStreamSynchronize failed");
    }
    return build_strided_level_finish(&pl);
}

// ------------------------------------------------------------------ export as k-major lists (tests / inspection)
__global__ void export_count_kernel(const int32_t *tile_sub_start, const int32_t *seg_start, int n_tiles, int K,
                                    const int32_t *sub_src, unsigned long long *per_k) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)K * n_tiles) return;
    int k = (int)(idx / n_tiles), t = (int)(idx - (int64_t)k * n_tiles);
    int64_t c0 = tile_sub_start[t] + seg_start[(int64_t)k * n_tiles + t];
    int64_t c1 = tile_sub_start[t] + seg_start[(int64_t)(k + 1) * n_tiles + t];
    int cnt = 0;
    for (int64_t s = c0 * SUB_SLOTS; s < c1 * SUB_SLOTS; s++) cnt += sub_src[s] >= 0;
    if (cnt) atomicAdd(&per_k[k], (unsigned long long)cnt);
}

// single-thread-per-k sequential fill keeps (tile, slot) order: deterministic, test-only path.
__global__ void export_fill_kernel(const int32_t *tile_sub_start, const int32_t *seg_start, int n_tiles, int K, int TR,
                                   const int32_t *sub_src, const uint8_t *sub_dst, const unsigned long long *start,
                                   int32_t *rin, int32_t *rout) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    int64_t pos = (int64_t)start[k];
    for (int t = 0; t < n_tiles; t++) {
        int64_t c0 = tile_sub_start[t] + seg_start[(int64_t)k * n_tiles + t];
        int64_t c1 = tile_sub_start[t] + seg_start[(int64_t)(k + 1) * n_tiles + t];
        for (int64_t s = c0 * SUB_SLOTS; s < c1 * SUB_SLOTS; s++) {
            int src = sub_src[s];
            if (src >= 0) {
                rin[pos] = src;
                rout[pos] = t * TR + sub_dst[s];
                pos++;
            }
        }
    }
}

int export_rules(const TiledRules &t, int64_t *start_host, int32_t *rin, int32_t *rout, hipStream_t st) {
    unsigned long long *per_k;
    SEI_TRY(pool_alloc_t(&per_k, 2 * (t.K + 1), st));
    SEI_HIP(hipMemsetAsync(per_k, 0, 2 * (t.K + 1) * sizeof(unsigned long long), st));
    int64_t tot = (int64_t)t.K * t.n_tiles;
    if (t.n_sub > 0)
        export_count_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(t.tile_sub_start, t.seg_start, t.n_tiles, t.K,
                                                                          t.sub_src, per_k);
    std::vector<unsigned long long> h(t.K + 1);
    SEI_HIP(hipMemcpyAsync(h.data(), per_k, t.K * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SEI_HIP(sync_stream(st));
    std::vector<unsigned long long> start(t.K + 1);
    start[0] = 0;
    for (int k = 0; k < t.K; k++) start[k + 1] = start[k] + h[k];
    for (int k = 0; k <= t.K; k++) start_host[k] = (int64_t)start[k];
    SEI_HIP(hipMemcpyAsync(per_k + t.K + 1, start.data(), (t.K + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    if (t.n_sub > 0 && rin && rout)
        export_fill_kernel<<<(t.K + 63) / 64, 64, 0, st>>>(t.tile_sub_start, t.seg_start, t.n_tiles, t.K, t.tile_rows,
                                                          t.sub_src, t.sub_dst, per_k + t.K + 1, rin, rout);
    SEI_LAUNCH_CHECK();
    SEI_HIP(sync_stream(st));
    pool_free(per_k);
    return 0;
}

}  // namespace sei
