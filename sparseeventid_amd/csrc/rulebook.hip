// Rulebook construction on the device (SCN `getSubmanifoldRuleBook` / `getRuleBook`):
// hash probes -> dense neighbour table -> per-wave-tile, per-offset compacted rule chunks.
// Integer / HBM-MALL bound; the chunk layout is what the MFMA conv kernels consume
// ("per-wavefront rulebook segments").
#include "common.h"

namespace sei {

int alloc_table(HashTable *ht, int64_t n, hipStream_t st);

__device__ inline int hash_find(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t mask,
                                uint64_t key) {
    uint64_t s = mix64(key) & mask;
    for (;;) {
        uint64_t k = keys[s];
        if (k == key) return (int)vals[s];
        if (k == EMPTY_KEY) return -1;
        s = (s + 1) & mask;
    }
}

struct ProbeGeom {
    int mode;  // 0 submanifold, 1 strided forward (dst = out sites), 2 strided transposed (dst = in sites)
    int f0, f1, f2;     // filter size
    int t0, t1, t2;     // stride
    int s0, s1, s2;     // spatial size of the SRC level
};

// One thread per (dst row, offset k): nbr[row*K + k] = src row feeding dst row through offset k, or -1.
__global__ void probe_kernel(const int32_t *__restrict__ dst_coords, int64_t n_dst, int K, ProbeGeom g,
                             const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t mask,
                             int32_t *__restrict__ nbr) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_dst * K) return;
    int64_t row = t / K;
    int k = (int)(t - row * K);
    int4 c = reinterpret_cast<const int4 *>(dst_coords)[row];
    int a = k / (g.f1 * g.f2), rem = k - a * (g.f1 * g.f2);
    int b = rem / g.f2, d = rem - b * g.f2;
    int q0, q1, q2;
    bool ok = true;
    if (g.mode == 0) {
        q0 = c.x - g.f0 / 2 + a; q1 = c.y - g.f1 / 2 + b; q2 = c.z - g.f2 / 2 + d;
    } else if (g.mode == 1) {
        q0 = c.x * g.t0 + a; q1 = c.y * g.t1 + b; q2 = c.z * g.t2 + d;
    } else {
        int p0 = c.x - a, p1 = c.y - b, p2 = c.z - d;
        ok = p0 >= 0 && p1 >= 0 && p2 >= 0 && (p0 % g.t0) == 0 && (p1 % g.t1) == 0 && (p2 % g.t2) == 0;
        q0 = p0 / g.t0; q1 = p1 / g.t1; q2 = p2 / g.t2;
    }
    ok = ok && q0 >= 0 && q0 < g.s0 && q1 >= 0 && q1 < g.s1 && q2 >= 0 && q2 < g.s2;
    int r = -1;
    if (ok) r = hash_find(keys, vals, mask, pack_key(q0, q1, q2, c.w));
    nbr[t] = r;
}

// One wavefront per tile of 64 dst rows (lane = row).  Counts chunks per tile and records the
// tile-relative chunk offset at which every filter offset k starts.
__global__ void tile_count_kernel(const int32_t *__restrict__ nbr, int64_t n_dst, int K, int n_tiles,
                                  int32_t *__restrict__ tile_chunks, int32_t *__restrict__ seg_start,
                                  unsigned long long *__restrict__ totals /*[0]=rules [1]=subchunks*/) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int64_t row = (int64_t)tile * TILE_ROWS + lane;
    int chunks = 0, rules = 0, subs = 0;
    for (int k = 0; k < K; k++) {
        bool hit = row < n_dst && nbr[row * K + k] >= 0;
        int cnt = __popcll(__ballot(hit));
        if (lane == 0) seg_start[(int64_t)k * n_tiles + tile] = chunks;
        chunks += (cnt + CHUNK_SLOTS - 1) / CHUNK_SLOTS;
        rules += cnt;
        subs += (cnt + 15) / 16;
    }
    if (lane == 0) {
        seg_start[(int64_t)K * n_tiles + tile] = chunks;
        tile_chunks[tile] = chunks;
        atomicAdd(&totals[0], (unsigned long long)rules);
        atomicAdd(&totals[1], (unsigned long long)subs);
    }
}

__global__ void tile_fill_kernel(const int32_t *__restrict__ nbr, int64_t n_dst, int K, int n_tiles,
                                 const int32_t *__restrict__ tile_chunk_start, int32_t *__restrict__ chunk_src,
                                 uint8_t *__restrict__ chunk_dst, uint8_t *__restrict__ chunk_k) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int64_t row = (int64_t)tile * TILE_ROWS + lane;
    int64_t chunk = tile_chunk_start[tile];
    for (int k = 0; k < K; k++) {
        int src = row < n_dst ? nbr[row * K + k] : -1;
        bool hit = src >= 0;
        unsigned long long bal = __ballot(hit);
        int cnt = __popcll(bal);
        if (cnt == 0) continue;
        int rank = __popcll(bal & ((1ull << lane) - 1ull));
        int nch = (cnt + CHUNK_SLOTS - 1) / CHUNK_SLOTS;
        int64_t base = chunk * CHUNK_SLOTS;
        if (hit) {
            chunk_src[base + rank] = src;
            chunk_dst[base + rank] = (uint8_t)lane;
        }
        // padding slots
        int pad = nch * CHUNK_SLOTS - cnt;
        if (lane < pad) {
            chunk_src[base + cnt + lane] = -1;
            chunk_dst[base + cnt + lane] = 0;
        }
        if (lane < nch) chunk_k[chunk + lane] = (uint8_t)k;
        chunk += nch;
    }
}

void free_tiled(TiledRules &t) {
    pool_free(t.tile_chunk_start); pool_free(t.chunk_src); pool_free(t.chunk_dst); pool_free(t.chunk_k);
    pool_free(t.seg_start);
    t = TiledRules();
}

// dst rows [n_dst] with coords dst_coords probe the src-level hash; result in *out.
static int build_tiled(const int32_t *dst_coords, int64_t n_dst, int64_t n_src, const HashTable &src_ht, int K,
                       ProbeGeom g, TiledRules *out, hipStream_t st) {
    if (K > 255) SEI_FAIL("rulebook: filter volume > 255 not supported");
    TiledRules t;
    t.n_dst = n_dst; t.n_src = n_src; t.K = K;
    t.n_tiles = (int)((n_dst + TILE_ROWS - 1) / TILE_ROWS);
    SEI_TRY(pool_alloc_t(&t.tile_chunk_start, t.n_tiles + 1));
    SEI_TRY(pool_alloc_t(&t.seg_start, (int64_t)(K + 1) * (t.n_tiles > 0 ? t.n_tiles : 1)));
    if (n_dst == 0) {
        SEI_HIP(hipMemsetAsync(t.tile_chunk_start, 0, sizeof(int32_t), st));
        SEI_TRY(pool_alloc_t(&t.chunk_src, 1)); SEI_TRY(pool_alloc_t(&t.chunk_dst, 1)); SEI_TRY(pool_alloc_t(&t.chunk_k, 1));
        *out = t;
        return 0;
    }
    int32_t *nbr, *tile_chunks;
    unsigned long long *totals;
    SEI_TRY(pool_alloc_t(&nbr, n_dst * K));
    SEI_TRY(pool_alloc_t(&tile_chunks, t.n_tiles + 1));
    SEI_TRY(pool_alloc_t(&totals, 4));
    SEI_HIP(hipMemsetAsync(totals, 0, 4 * sizeof(unsigned long long), st));
    int64_t tot = n_dst * K;
    probe_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(dst_coords, n_dst, K, g, src_ht.keys, src_ht.vals,
                                                               src_ht.mask, nbr);
    const int WPB = 4;
    int nb = (t.n_tiles + WPB - 1) / WPB;
    tile_count_kernel<<<nb, WPB * 64, 0, st>>>(nbr, n_dst, K, t.n_tiles, tile_chunks, t.seg_start, totals);
    SEI_LAUNCH_CHECK();
    SEI_TRY(exclusive_scan_i32(tile_chunks, t.tile_chunk_start, t.n_tiles, (int32_t *)(totals + 2), st));
    // tile_chunk_start[n_tiles] = total
    SEI_HIP(hipMemcpyAsync(t.tile_chunk_start + t.n_tiles, totals + 2, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    unsigned long long host[3];
    SEI_HIP(hipMemcpyAsync(host, totals, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SEI_HIP(hipStreamSynchronize(st));
    t.n_rules = (int64_t)host[0];
    t.n_sub = (int64_t)host[1];
    t.n_chunks = (int64_t)(uint32_t)host[2];
    int64_t nc = t.n_chunks > 0 ? t.n_chunks : 1;
    SEI_TRY(pool_alloc_t(&t.chunk_src, nc * CHUNK_SLOTS));
    SEI_TRY(pool_alloc_t(&t.chunk_dst, nc * CHUNK_SLOTS));
    SEI_TRY(pool_alloc_t(&t.chunk_k, nc));
    tile_fill_kernel<<<nb, WPB * 64, 0, st>>>(nbr, n_dst, K, t.n_tiles, t.tile_chunk_start, t.chunk_src, t.chunk_dst,
                                              t.chunk_k);
    SEI_LAUNCH_CHECK();
    pool_free(nbr); pool_free(tile_chunks); pool_free(totals);
    *out = t;
    return 0;
}

int build_subm_rules(Level *lv, const int32_t filter[3], TiledRules *out, hipStream_t st) {
    int K = filter[0] * filter[1] * filter[2];
    ProbeGeom g{0, filter[0], filter[1], filter[2], 1, 1, 1, lv->spatial[0], lv->spatial[1], lv->spatial[2]};
    double bytes = 12.0 * lv->n + 8.0 * K * lv->n;
    ProfScope ps(PROF_RULES, st, 0.0, bytes);
    return build_tiled(lv->coords, lv->n, lv->n, lv->ht, K, g, out, st);
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

__device__ inline uint32_t hash_insert_min_dev(uint64_t *keys, uint32_t *vals, uint64_t mask, uint64_t key, uint32_t cand) {
    uint64_t s = mix64(key) & mask;
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long *)&keys[s], (unsigned long long)EMPTY_KEY,
                                            (unsigned long long)key);
        if (prev == EMPTY_KEY || prev == key) {
            atomicMin(&vals[s], cand);
            return (uint32_t)s;
        }
        s = (s + 1) & mask;
    }
}

__global__ void out_insert_kernel(const int32_t *__restrict__ in_coords, int64_t n, int mult, int m1, int m2,
                                  ProbeGeom g, int o0, int o1, int o2, uint64_t *keys, uint32_t *vals, uint64_t mask,
                                  uint32_t *slot_of) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * mult) return;
    int64_t i = t / mult;
    int j = (int)(t - i * mult);
    int4 p = reinterpret_cast<const int4 *>(in_coords)[i];
    int q0, q1, q2;
    if (!candidate(p, j, g, o0, o1, o2, m1, m2, &q0, &q1, &q2)) {
        slot_of[t] = 0xFFFFFFFFu;
        return;
    }
    slot_of[t] = hash_insert_min_dev(keys, vals, mask, pack_key(q0, q1, q2, p.w), (uint32_t)t);
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
    if (t >= n * mult) return;
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

int build_strided(sei_metadata *m, Level *lin, const int32_t size[3], const int32_t stride[3], int32_t batch,
                  sei_rulebook *rb, hipStream_t st) {
    int32_t osz[3];
    for (int d = 0; d < 3; d++) {
        if (size[d] < 1 || stride[d] < 1) SEI_FAIL("Convolution: bad filter size / stride");
        osz[d] = (lin->spatial[d] - size[d]) / stride[d] + 1;
        if (osz[d] < 1) SEI_FAIL("Convolution: input spatial size smaller than filter");
        rb->spatial_out[d] = osz[d];
    }
    if (find_level(m, osz)) SEI_FAIL("Convolution: a level with the output spatial size already exists in this metadata");
    int K = size[0] * size[1] * size[2];
    int m0 = (size[0] + stride[0] - 1) / stride[0], m1 = (size[1] + stride[1] - 1) / stride[1],
        m2 = (size[2] + stride[2] - 1) / stride[2];
    int mult = m0 * m1 * m2;
    int64_t n = lin->n;
    ProbeGeom g{1, size[0], size[1], size[2], stride[0], stride[1], stride[2], lin->spatial[0], lin->spatial[1],
                lin->spatial[2]};
    double bytes = 12.0 * n * mult + 8.0 * K * n;
    ProfScope ps(PROF_RULES, st, 0.0, bytes);
    Level *lo = new Level();
    for (int d = 0; d < 3; d++) lo->spatial[d] = osz[d];
    int64_t cand = n * mult;
    if (cand >= (1ll << 31) - 64) SEI_FAIL("Convolution: too many candidate output sites");
    SEI_TRY(alloc_table(&lo->ht, cand, st));
    if (n == 0) {
        lo->n = 0;
        SEI_TRY(pool_alloc_t(&lo->coords, 4));
    } else {
        uint32_t *slot_of, *new_vals;
        int32_t *flag, *total;
        SEI_TRY(pool_alloc_t(&slot_of, cand));
        SEI_TRY(pool_alloc_t(&flag, cand));
        SEI_TRY(pool_alloc_t(&total, 2));
        SEI_TRY(pool_alloc_t(&new_vals, lo->ht.mask + 1));
        SEI_HIP(hipMemsetAsync(new_vals, 0xFF, (lo->ht.mask + 1) * sizeof(uint32_t), st));
        unsigned nb = (unsigned)((cand + 255) / 256);
        out_insert_kernel<<<nb, 256, 0, st>>>(lin->coords, n, mult, m1, m2, g, osz[0], osz[1], osz[2], lo->ht.keys,
                                              lo->ht.vals, lo->ht.mask, slot_of);
        out_flag_kernel<<<nb, 256, 0, st>>>(slot_of, lo->ht.vals, cand, flag);
        SEI_LAUNCH_CHECK();
        SEI_TRY(exclusive_scan_i32(flag, flag, cand, total, st));
        int32_t host;
        SEI_HIP(hipMemcpyAsync(&host, total, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        SEI_HIP(hipStreamSynchronize(st));
        lo->n = host;
        SEI_TRY(pool_alloc_t(&lo->coords, 4 * (int64_t)(host > 0 ? host : 1)));
        out_assign_kernel<<<nb, 256, 0, st>>>(lin->coords, n, mult, m1, m2, g, osz[0], osz[1], osz[2], slot_of,
                                              lo->ht.vals, flag, lo->coords, new_vals);
        SEI_LAUNCH_CHECK();
        pool_free(lo->ht.vals);
        lo->ht.vals = new_vals;
        pool_free(slot_of); pool_free(flag); pool_free(total);
    }
    m->levels.push_back(lo);
    // forward rules: dst = output sites probing the input hash
    SEI_TRY(build_tiled(lo->coords, lo->n, lin->n, lin->ht, K, g, &rb->fwd, st));
    // transposed rules: dst = input sites probing the output hash
    ProbeGeom gt = g;
    gt.mode = 2;
    gt.s0 = osz[0]; gt.s1 = osz[1]; gt.s2 = osz[2];
    SEI_TRY(build_tiled(lin->coords, lin->n, lo->n, lo->ht, K, gt, &rb->bwd, st));
    return 0;
}

// ------------------------------------------------------------------ export as k-major lists (tests / inspection)
__global__ void export_count_kernel(const int32_t *chunk_src, const uint8_t *chunk_k, int64_t n_chunks,
                                    unsigned long long *per_k) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    int cnt = 0;
    for (int s = 0; s < CHUNK_SLOTS; s++) cnt += chunk_src[c * CHUNK_SLOTS + s] >= 0;
    atomicAdd(&per_k[chunk_k[c]], (unsigned long long)cnt);
}

// single-thread-per-k sequential fill keeps (tile, slot) order: deterministic, test-only path.
__global__ void export_fill_kernel(const int32_t *tile_chunk_start, const int32_t *seg_start, int n_tiles, int K,
                                   const int32_t *chunk_src, const uint8_t *chunk_dst,
                                   const unsigned long long *start, int32_t *rin, int32_t *rout) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    int64_t pos = (int64_t)start[k];
    for (int t = 0; t < n_tiles; t++) {
        int64_t c0 = tile_chunk_start[t] + seg_start[(int64_t)k * n_tiles + t];
        int64_t c1 = tile_chunk_start[t] + seg_start[(int64_t)(k + 1) * n_tiles + t];
        for (int64_t s = c0 * CHUNK_SLOTS; s < c1 * CHUNK_SLOTS; s++) {
            int src = chunk_src[s];
            if (src >= 0) {
                rin[pos] = src;
                rout[pos] = t * TILE_ROWS + chunk_dst[s];
                pos++;
            }
        }
    }
}

int export_rules(const TiledRules &t, int64_t *start_host, int32_t *rin, int32_t *rout, hipStream_t st) {
    unsigned long long *per_k;
    SEI_TRY(pool_alloc_t(&per_k, 2 * (t.K + 1)));
    SEI_HIP(hipMemsetAsync(per_k, 0, 2 * (t.K + 1) * sizeof(unsigned long long), st));
    if (t.n_chunks > 0)
        export_count_kernel<<<(unsigned)((t.n_chunks + 255) / 256), 256, 0, st>>>(t.chunk_src, t.chunk_k, t.n_chunks, per_k);
    std::vector<unsigned long long> h(t.K + 1);
    SEI_HIP(hipMemcpyAsync(h.data(), per_k, t.K * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SEI_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> start(t.K + 1);
    start[0] = 0;
    for (int k = 0; k < t.K; k++) start[k + 1] = start[k] + h[k];
    for (int k = 0; k <= t.K; k++) start_host[k] = (int64_t)start[k];
    SEI_HIP(hipMemcpyAsync(per_k + t.K + 1, start.data(), (t.K + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    if (t.n_chunks > 0 && rin && rout)
        export_fill_kernel<<<(t.K + 63) / 64, 64, 0, st>>>(t.tile_chunk_start, t.seg_start, t.n_tiles, t.K, t.chunk_src,
                                                          t.chunk_dst, per_k + t.K + 1, rin, rout);
    SEI_LAUNCH_CHECK();
    SEI_HIP(hipStreamSynchronize(st));
    pool_free(per_k);
    return 0;
}

}  // namespace sei
