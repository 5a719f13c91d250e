// Coordinate hash tables on the device + scn.InputLayer (resnet.py:40-43,143).
// Open addressing, linear probing, 64-bit packed keys (batch|x0|x1|x2), load factor <= 0.5.
// HBM/MALL-bound integer work: one thread per point, coalesced reads of the COO array.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace sei {

// ------------------------------------------------------------------ exclusive scan (int32)
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

__device__ inline int block_exclusive_scan(int v, int *total, int *lds /*[SCAN_THREADS/64+1]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; w++) {
        int s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ void scan_reduce_kernel(const int32_t *in, int64_t n, int32_t *block_sums) {
    __shared__ int lds[8];
    int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) s += in[base + i];
    int tot;
    block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void scan_blocksums_kernel(int32_t *block_sums, int nb, int32_t *total_out) {
    __shared__ int lds[20];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += blockDim.x) {
        int i = b0 + threadIdx.x;
        int v = i < nb ? block_sums[i] : 0;
        int tot;
        int ex = block_exclusive_scan(v, &tot, lds);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ void scan_apply_kernel(const int32_t *in, int32_t *out, int64_t n, const int32_t *block_sums) {
    __shared__ int lds[8];
    int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot, lds) + block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
}

// short arrays (the per-tile sub-chunk counts of a tiled rulebook variant: ~150 scans per training step, most of them of a
// few thousand elements) in ONE launch of one workgroup instead of three launches: every thread scans a contiguous run
constexpr int SCAN_SMALL_THREADS = 1024, SCAN_SMALL_MAX = SCAN_SMALL_THREADS * 32;
__global__ __launch_bounds__(SCAN_SMALL_THREADS) void scan_small_kernel(const int32_t *in, int32_t *out, int n, int32_t *total_out) {
    __shared__ int lds[SCAN_SMALL_THREADS / 64 + 1];
    const int per = (n + SCAN_SMALL_THREADS - 1) / SCAN_SMALL_THREADS;   // <= 32
    const int b = threadIdx.x * per, e = min(n, b + per);
    int v[32];
    int s = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        v[i] = (i < per && b + i < e) ? in[b + i] : 0;
        s += v[i];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot, lds);
#pragma unroll
    for (int i = 0; i < 32; i++) {
        if (i < per && b + i < e) out[b + i] = ex;
        ex += v[i];
    }
    if (threadIdx.x == 0 && total_out) *total_out = tot;
}

// out[i] = sum_{j<i} in[j]; *total_dev (device int, may be null) = sum of all.  in/out may alias.
int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, int32_t *total_dev, hipStream_t st) {
    if (n == 0) {
        if (total_dev) SEI_HIP(hipMemsetAsync(total_dev, 0, sizeof(int32_t), st));
        return 0;
    }
    if (n <= SCAN_SMALL_MAX) {
        scan_small_kernel<<<1, SCAN_SMALL_THREADS, 0, st>>>(in, out, (int)n, total_dev);
        SEI_LAUNCH_CHECK();
        return 0;
    }
    int nb = (int)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    int32_t *bs;
    SEI_TRY(pool_alloc_t(&bs, nb, st));
    scan_reduce_kernel<<<nb, SCAN_THREADS, 0, st>>>(in, n, bs);
    scan_blocksums_kernel<<<1, 1024, 0, st>>>(bs, nb, total_dev);
    scan_apply_kernel<<<nb, SCAN_THREADS, 0, st>>>(in, out, n, bs);
    SEI_LAUNCH_CHECK();
    pool_free(bs);
    return 0;
}

// ------------------------------------------------------------------ hash insert (hash_insert_min: common.h)
__global__ void input_insert_kernel(const int32_t *__restrict__ coords, int64_t n, int s0, int s1, int s2, int batch,
                                    uint64_t *keys, uint32_t *vals, uint64_t mask, uint32_t *filter, uint64_t fmask,
                                    uint32_t *slot_of, int32_t *err) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = reinterpret_cast<const int4 *>(coords)[i];
    if (c.x < 0 || c.x >= s0 || c.y < 0 || c.y >= s1 || c.z < 0 || c.z >= s2 || c.w < 0 || c.w >= batch) {
        atomicMin((unsigned int *)err, (unsigned int)i);
        slot_of[i] = 0xFFFFFFFFu;
        return;
    }
    const uint64_t key = pack_key(c.x, c.y, c.z, c.w);
    const uint32_t s = hash_insert_min(keys, vals, mask, key, (uint32_t)i);
    filter_set(filter, fmask, key);
    if (s == 0xFFFFFFFFu) atomicMin((unsigned int *)err + 2, 0u);  // table full (cannot happen at load <= 0.5)
    slot_of[i] = s;
}

__global__ void flag_first_kernel(const uint32_t *slot_of, const uint32_t *vals, int64_t n, int32_t *flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = slot_of[i];
    flag[i] = (s != 0xFFFFFFFFu && vals[s] == (uint32_t)i) ? 1 : 0;
}

// representatives write their row's coords + first index; slot value is re-pointed to the row in a
// second array so that concurrent readers of `vals` are not disturbed.
__global__ void assign_rows_kernel(const int32_t *__restrict__ coords, const uint32_t *slot_of, const uint32_t *vals,
                                   const int32_t *pos, int64_t n, int32_t *out_coords, uint32_t *new_vals,
                                   int32_t *row_first) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = slot_of[i];
    if (s == 0xFFFFFFFFu || vals[s] != (uint32_t)i) return;
    int r = pos[i];
    reinterpret_cast<int4 *>(out_coords)[r] = reinterpret_cast<const int4 *>(coords)[i];
    new_vals[s] = (uint32_t)r;
    row_first[r] = (int32_t)i;
}

__global__ void input_rowof_kernel(const uint32_t *slot_of, const uint32_t *new_vals, int64_t n, int32_t *input_row,
                                   int32_t *row_last, int32_t *row_count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = slot_of[i];
    if (s == 0xFFFFFFFFu) {
        input_row[i] = -1;
        return;
    }
    int r = (int)new_vals[s];
    input_row[i] = r;
    atomicMax(&row_last[r], (int32_t)i);
    atomicAdd(&row_count[r], 1);
}

__global__ void iota_kernel(int32_t *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}

static uint64_t table_capacity(int64_t n) {
    uint64_t cap = 1024;
    while (cap < (uint64_t)(2 * n + 1)) cap <<= 1;
    return cap;
}

int alloc_table(HashTable *ht, int64_t n, hipStream_t st) {
    uint64_t cap = table_capacity(n);
    SEI_TRY(pool_alloc_t(&ht->keys, cap, st));
    SEI_TRY(pool_alloc_t(&ht->vals, cap, st));
    ht->mask = cap - 1;
    SEI_HIP(hipMemsetAsync(ht->keys, 0xFF, cap * sizeof(uint64_t), st));
    SEI_HIP(hipMemsetAsync(ht->vals, 0xFF, cap * sizeof(uint32_t), st));
    const uint64_t fbits = cap * 8;  // power of two, >= 8192
    SEI_TRY(pool_alloc_t(&ht->filter, fbits / 32, st));
    ht->fmask = fbits - 1;
    SEI_HIP(hipMemsetAsync(ht->filter, 0, fbits / 8, st));
    return 0;
}

// scn.InputLayer coordinate pass, in two halves around the ONE read-back of (first bad point, row count, table-full flag):
//   begin : table, insert, first-appearance flags, scan, rows assigned -- arrays sized for n rows (rows <= points); the row
//           count stays on the device (Level::n_dev) for strided levels begun behind it
//   finish: after the stream has been synchronised: errors, host row count, the sorted lists of duplicate points.
int build_input_level_begin(sei_metadata *m, const int32_t *coords, int64_t n, const int32_t spatial[3], int32_t batch,
                            hipStream_t st, PendingInput *pi) {
    *pi = PendingInput();
    for (int d = 0; d < 3; d++)
        if (spatial[d] <= 0 || spatial[d] > 65535) SEI_FAIL("InputLayer: spatial size must be in [1,65535]");
    if (batch <= 0 || batch > 65534) SEI_FAIL("InputLayer: batch size must be in [1,65534]");
    if (n >= (1ll << 31) - 64) SEI_FAIL("InputLayer: too many points");
    Level *lv = new Level();
    for (int d = 0; d < 3; d++) lv->spatial[d] = spatial[d];
    m->levels.push_back(lv);
    m->batch_size = batch;
    m->n_input = n;
    pi->level = lv;
    pi->coords = coords;
    pi->n = n;
    if (n == 0) {
        lv->n = 0;
        SEI_TRY(alloc_table(&lv->ht, 0, st));
        SEI_TRY(pool_alloc_t(&lv->coords, 4, st));
        return 0;
    }
    SEI_TRY(alloc_table(&lv->ht, n, st));
    SEI_TRY(pool_alloc_t(&pi->slot_of, n, st));
    SEI_TRY(pool_alloc_t(&pi->flag, n, st));
    SEI_TRY(pool_alloc_t(&pi->err_total, 4, st));
    SEI_TRY(pool_alloc_t(&pi->new_vals, lv->ht.mask + 1, st));
    SEI_TRY(pool_alloc_t(&m->input_row, n, st));
    pi->host = stage_get();
    if (!pi->host) SEI_FAIL("InputLayer: cannot allocate pinned staging memory");
    SEI_HIP(hipMemsetAsync(pi->err_total, 0xFF, 4 * sizeof(int32_t), st));
    SEI_HIP(hipMemsetAsync(pi->new_vals, 0xFF, (lv->ht.mask + 1) * sizeof(uint32_t), st));
    const int T = 256;
    const int nb = (int)((n + T - 1) / T);
    input_insert_kernel<<<nb, T, 0, st>>>(coords, n, spatial[0], spatial[1], spatial[2], batch, lv->ht.keys,
                                          lv->ht.vals, lv->ht.mask, lv->ht.filter, lv->ht.fmask, pi->slot_of, pi->err_total);
    flag_first_kernel<<<nb, T, 0, st>>>(pi->slot_of, lv->ht.vals, n, pi->flag);
    SEI_LAUNCH_CHECK();
    SEI_TRY(exclusive_scan_i32(pi->flag, pi->flag, n, pi->err_total + 1, st));
    SEI_HIP(hipMemcpyAsync(pi->host, pi->err_total, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    // rows <= points: the per-row arrays are sized for n
    SEI_TRY(pool_alloc_t(&lv->coords, 4 * n, st));
    SEI_TRY(pool_alloc_t(&m->row_first, n, st));
    SEI_TRY(pool_alloc_t(&m->row_last, n, st));
    SEI_TRY(pool_alloc_t(&m->row_count, n, st));
    SEI_HIP(hipMemsetAsync(m->row_last, 0xFF, n * sizeof(int32_t), st));
    SEI_HIP(hipMemsetAsync(m->row_count, 0, n * sizeof(int32_t), st));
    assign_rows_kernel<<<nb, T, 0, st>>>(coords, pi->slot_of, lv->ht.vals, pi->flag, n, lv->coords, pi->new_vals, m->row_first);
    input_rowof_kernel<<<nb, T, 0, st>>>(pi->slot_of, pi->new_vals, n, m->input_row, m->row_last, m->row_count);
    SEI_LAUNCH_CHECK();
    // the table maps key -> row from here on (a strided level begun behind this one probes nothing of it, but reads coords)
    pool_free(lv->ht.vals);
    lv->ht.vals = pi->new_vals;
    pi->new_vals = nullptr;
    lv->n_dev = pi->err_total + 1;
    lv->n_ub = n;
    lv->n = -1;
    pi->pending = true;
    return 0;
}

int build_input_level_finish(sei_metadata *m, PendingInput *pi, hipStream_t st, int64_t *n_rows) {
    Level *lv = pi->level;
    if (!pi->pending) {   // empty input
        *n_rows = lv ? lv->n : 0;
        return 0;
    }
    const int64_t n = pi->n;
    const int32_t *host = reinterpret_cast<const int32_t *>(pi->host);
    const int32_t bad = host[0], full = host[2];
    const int64_t nr = host[1];
    lv->n = nr;
    lv->n_dev = nullptr;
    stage_put(pi->host);
    pi->host = nullptr;
    pi->pending = false;
    int rc = 0;
    if (bad != -1) {
        set_error("InputLayer: coordinate of input point " + std::to_string((uint32_t)bad) + " is outside spatial_size / batch_size");
        rc = 1;
    } else if (full != -1) {
        set_error("InputLayer: hash table full");
        rc = 1;
    }
    if (!rc && nr != n) {
        // Several input points share a site (modes 3 / 4 sum them): their contributions are added in INPUT ORDER, not by
        // float atomics.  dup_order = the point indices sorted by row with a STABLE radix sort (equal rows keep their input
        // order), dup_start = exclusive scan of the rows' point counts: row r owns dup_order[dup_start[r] .. +row_count[r]).
        // Batches without duplicates (every DUNE / SBND input) skip all of this.
        const int T = 256;
        const int nb = (int)((n + T - 1) / T);
        int32_t *keys_in = m->input_row, *keys_out = nullptr, *idx_in = nullptr;
        rc = pool_alloc_t(&keys_out, n, st);
        if (!rc) rc = pool_alloc_t(&idx_in, n, st);
        if (!rc) rc = pool_alloc_t(&m->dup_order, n, st);
        if (!rc) rc = pool_alloc_t(&m->dup_start, nr + 1, st);
        if (!rc) {
            iota_kernel<<<nb, T, 0, st>>>(idx_in, n);
            int bits = 1;
            while ((1ll << bits) < nr + 1 && bits < 31) bits++;
            size_t tmp_bytes = 0;
            hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in, keys_out, idx_in, m->dup_order, (int)n, 0, bits, st);
            void *tmp = e == hipSuccess ? pool_alloc(tmp_bytes + 256, st) : nullptr;
            if (e == hipSuccess && tmp)
                e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys_in, keys_out, idx_in, m->dup_order, (int)n, 0, bits, st);
            if (e != hipSuccess || !tmp) { set_error("InputLayer: sorting the duplicate points failed"); rc = 1; }
            if (!rc) rc = exclusive_scan_i32(m->row_count, m->dup_start, nr, nullptr, st);
            pool_free(tmp);
        }
        pool_free(keys_out); pool_free(idx_in);
    }
    pool_free(pi->slot_of); pool_free(pi->flag); pool_free(pi->err_total);
    pi->slot_of = nullptr; pi->flag = nullptr; pi->err_total = nullptr;
    *n_rows = nr;
    return rc;
}

int build_input_level(sei_metadata *m, const int32_t *coords, int64_t n, const int32_t spatial[3], int32_t batch,
                      hipStream_t st, int64_t *n_rows) {
    PendingInput pi;
    int rc = build_input_level_begin(m, coords, n, spatial, batch, st, &pi);
    if (!rc && pi.pending && sync_stream(st) != hipSuccess) {
        set_error("InputLayer: hipStreamSynchronize failed");
        rc = 1;
    }
    if (rc) {   // what begin allocated goes back; the level and the arrays the metadata owns are freed with the metadata
        pool_free(pi.slot_of); pool_free(pi.flag); pool_free(pi.err_total); pool_free(pi.new_vals); stage_put(pi.host);
        return rc;
    }
    return build_input_level_finish(m, &pi, st, n_rows);
}

// ------------------------------------------------------------------ InputLayer features
// mode 0/2: first, 1: last, 3: sum, 4: mean.  One thread per (point, channel).
__global__ void input_feat_fwd_kernel(const float *__restrict__ in, int64_t n, int C, const int32_t *input_row,
                                      const int32_t *row_first, const int32_t *row_last, const int32_t *row_count,
                                      int mode, float *out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    int64_t i = t / C;
    int c = (int)(t - i * C);
    int r = input_row[i];
    float v = in[t];
    if (mode == 3 || mode == 4) {
        // only reached when every row has ONE point (input_feat_sum_kernel otherwise)
        out[(int64_t)r * C + c] = v;
    } else if (mode == 1) {
        if (row_last[r] == i) out[(int64_t)r * C + c] = v;
    } else {
        if (row_first[r] == i) out[(int64_t)r * C + c] = v;
    }
}

// modes 3 / 4 with duplicate input points: one thread per (row, channel) adds the row's points in input order
// (dup_order: stable sort by row) -- a fixed summation order, bit-identical reruns like every other kernel.
__global__ void input_feat_sum_kernel(const float *__restrict__ in, int64_t n_rows, int C, const int32_t *__restrict__ dup_order,
                                      const int32_t *__restrict__ dup_start, const int32_t *__restrict__ row_count, int mode,
                                      float *__restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * C) return;
    int64_t r = t / C;
    int c = (int)(t - r * C);
    const int cnt = row_count[r], s0 = dup_start[r];
    float sum = 0.f;
    for (int j = 0; j < cnt; j++) {
        float v = in[(int64_t)dup_order[s0 + j] * C + c];
        if (mode == 4) v /= (float)cnt;
        sum = j == 0 ? v : sum + v;
    }
    out[t] = sum;
}

__global__ void input_feat_bwd_kernel(const float *__restrict__ dout, int64_t n, int C, const int32_t *input_row,
                                      const int32_t *row_first, const int32_t *row_last, const int32_t *row_count,
                                      int mode, float *din) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    int64_t i = t / C;
    int c = (int)(t - i * C);
    int r = input_row[i];
    float g = dout[(int64_t)r * C + c];
    if (mode == 4)
        g /= (float)row_count[r];
    else if (mode == 1)
        g = row_last[r] == i ? g : 0.f;
    else if (mode != 3)
        g = row_first[r] == i ? g : 0.f;
    din[t] = g;
}

int input_features_fwd(sei_metadata *m, const float *in, int C, float *out, hipStream_t st) {
    if (m->levels.empty()) SEI_FAIL("InputLayer features: metadata has no input level");
    int64_t nr = m->levels[0]->n, n = m->n_input;
    if (nr == 0 || n == 0) return 0;
    if ((m->input_mode == 3 || m->input_mode == 4) && m->dup_order) {
        int64_t totr = nr * C;
        input_feat_sum_kernel<<<(unsigned)((totr + 255) / 256), 256, 0, st>>>(in, nr, C, m->dup_order, m->dup_start, m->row_count,
                                                                              m->input_mode, out);
        SEI_LAUNCH_CHECK();
        return 0;
    }
    int64_t tot = n * C;
    input_feat_fwd_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(in, n, C, m->input_row, m->row_first,
                                                                        m->row_last, m->row_count, m->input_mode, out);
    SEI_LAUNCH_CHECK();
    return 0;
}

int input_features_bwd(sei_metadata *m, const float *dout, int C, float *din, hipStream_t st) {
    if (m->levels.empty()) SEI_FAIL("InputLayer features: metadata has no input level");
    int64_t n = m->n_input;
    if (n == 0) return 0;
    int64_t tot = n * C;
    input_feat_bwd_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(dout, n, C, m->input_row, m->row_first,
                                                                        m->row_last, m->row_count, m->input_mode, din);
    SEI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ larcv batch -> scn tuple on the device
// src/io/data_transforms.py:21-49 (3-D) / :198-253 (2-D): strip the -999 padding of [B,P,M,D+1] and emit
// coords int32 [N,4] + features [N,1] in the reference's order (3-D: batch-major; 2-D: plane-major).
// element e of the OUTPUT order = ((p*B + b)*M + v) for 2-D, ((b*P + p)*M + v) for 3-D.
__global__ void larcv_flag_kernel(const float *__restrict__ in, int B, int P, int M, int D, int plane_major,
                                  int32_t *flag) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t tot = (int64_t)B * P * M;
    if (e >= tot) return;
    int v = (int)(e % M);
    int64_t q = e / M;
    int b, p;
    if (plane_major) { p = (int)(q / B); b = (int)(q % B); } else { b = (int)(q / P); p = (int)(q % P); }
    const float *r = in + (((int64_t)b * P + p) * M + v) * (D + 1);
    flag[e] = r[D] != -999.f ? 1 : 0;
}

__global__ void larcv_emit_kernel(const float *__restrict__ in, int B, int P, int M, int D, int plane_major,
                                  const int32_t *__restrict__ pos, int32_t *__restrict__ coords,
                                  float *__restrict__ feats) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t tot = (int64_t)B * P * M;
    if (e >= tot) return;
    int v = (int)(e % M);
    int64_t q = e / M;
    int b, p;
    if (plane_major) { p = (int)(q / B); b = (int)(q % B); } else { b = (int)(q / P); p = (int)(q % P); }
    const float *r = in + (((int64_t)b * P + p) * M + v) * (D + 1);
    if (r[D] == -999.f) return;
    int o = pos[e];
    int4 c;
    if (D == 3) c = make_int4((int)r[0], (int)r[1], (int)r[2], b);   // (x, y, z, batch)
    else        c = make_int4(p, (int)r[1], (int)r[0], b);           // (plane, y, x, batch)
    reinterpret_cast<int4 *>(coords)[o] = c;
    feats[o] = r[D];
}

int larcv_compact(const float *in, int B, int P, int M, int D, int32_t *coords, float *feats, int64_t *n_out,
                  hipStream_t st) {
    if (D != 2 && D != 3) SEI_FAIL("larcv_compact: D must be 2 or 3");
    int64_t tot = (int64_t)B * P * M;
    if (tot >= (1ll << 31) - 64) SEI_FAIL("larcv_compact: batch too large");
    if (tot == 0) { *n_out = 0; return 0; }
    int32_t *flag, *total;
    SEI_TRY(pool_alloc_t(&flag, tot, st));
    SEI_TRY(pool_alloc_t(&total, 2, st));
    unsigned nb = (unsigned)((tot + 255) / 256);
    int pm = D == 2 ? 1 : 0;
    larcv_flag_kernel<<<nb, 256, 0, st>>>(in, B, P, M, D, pm, flag);
    SEI_LAUNCH_CHECK();
    SEI_TRY(exclusive_scan_i32(flag, flag, tot, total, st));
    larcv_emit_kernel<<<nb, 256, 0, st>>>(in, B, P, M, D, pm, flag, coords, feats);
    SEI_LAUNCH_CHECK();
    int32_t host;
    SEI_HIP(hipMemcpyAsync(&host, total, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SEI_HIP(sync_stream(st));
    *n_out = host;
    pool_free(flag); pool_free(total);
    return 0;
}

}  // namespace sei
