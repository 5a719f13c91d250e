/*
 * oracle/scn_rules.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the integer half of the SparseConvNet-CPU
 * algorithm that the reference calls through `import sparseconvnet as scn`
 * (/root/reference/src/networks/resnet.py:2,
 *  /root/reference/src/networks/sparse_building_blocks.py:3):
 *   - scn.InputLayer        (resnet.py:26-29,40-43,143): coordinate hash build
 *                            + duplicate handling (modes 0-4)
 *   - scn.SubmanifoldConvolution rulebook (resnet.py:30-36,44-50,105-110;
 *                            sparse_building_blocks.py:29-34)
 *   - scn.Convolution output-site discovery + rulebook
 *                            (sparse_building_blocks.py:110-117)
 *
 * The arithmetic itself lives in the third-party `sparseconvnet` package
 * (facebookresearch/SparseConvNet, NO pinned version: requirements.txt is
 * empty; not in /root/reference, not installed).  Its published algorithm is
 * restated here: one hash map coord->row per level, one (in,out) rule list per
 * filter offset k (k row-major over the filter, last dim fastest).
 *
 * PARITY UNPINNED: the reference holds no tests / golden vectors for this path
 * (SURVEY.md 4, 8c).  The oracle is pinned instead by dense-convolution
 * equivalence (oracle/dense_ref.py, tests/test_oracle.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * Coordinates are int32 [N,4] = (x0, x1, x2, batch) exactly as the reference's
 * transform emits them (src/io/data_transforms.py:36-46, batch index last).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t *keys;
    int32_t *vals;
    uint64_t mask;
} hmap_t;

#define EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

static uint64_t pack_key(int32_t x0, int32_t x1, int32_t x2, int32_t b) {
    return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)x0 << 32) |
           ((uint64_t)(uint16_t)x1 << 16) | (uint64_t)(uint16_t)x2;
}

static uint64_t mix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33; return k;
}

static int hmap_init(hmap_t *h, int64_t n) {
    uint64_t cap = 16;
    while (cap < (uint64_t)(2 * n + 1)) cap <<= 1;
    h->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    h->vals = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!h->keys || !h->vals) return -1;
    memset(h->keys, 0xFF, cap * sizeof(uint64_t));
    h->mask = cap - 1;
    return 0;
}

static void hmap_free(hmap_t *h) { free(h->keys); free(h->vals); }

/* returns slot; *found says whether key was present */
static uint64_t hmap_slot(const hmap_t *h, uint64_t key, int *found) {
    uint64_t s = mix64(key) & h->mask;
    for (;;) {
        if (h->keys[s] == key) { *found = 1; return s; }
        if (h->keys[s] == EMPTY_KEY) { *found = 0; return s; }
        s = (s + 1) & h->mask;
    }
}

static int32_t hmap_get(const hmap_t *h, uint64_t key) {
    int f; uint64_t s = hmap_slot(h, key, &f);
    return f ? h->vals[s] : -1;
}

/*
 * scn.InputLayer [upstream semantics, SURVEY.md 2.2]: rows are numbered in
 * first-appearance order over the input list; duplicates map to the same row.
 * row_of[i] = output row of input point i.  first_of[r] = first input index of
 * row r, last_of[r] = last, count[r] = multiplicity (caller combines features
 * per `mode`: 0/2 first, 1 last, 3 sum, 4 mean).
 * Returns number of unique rows, or -(i+1) if point i is out of bounds.
 */
int64_t oracle_input_layer(const int32_t *coords, int64_t n, const int32_t *spatial,
                           int32_t batch_size, int32_t *row_of, int32_t *first_of,
                           int32_t *last_of, int32_t *count, int32_t *out_coords) {
    hmap_t h;
    if (hmap_init(&h, n)) return INT64_MIN;
    int64_t nu = 0;
    for (int64_t i = 0; i < n; i++) {
        const int32_t *c = coords + 4 * i;
        if (c[0] < 0 || c[0] >= spatial[0] || c[1] < 0 || c[1] >= spatial[1] ||
            c[2] < 0 || c[2] >= spatial[2] || c[3] < 0 || c[3] >= batch_size) {
            hmap_free(&h);
            return -(i + 1);
        }
        uint64_t key = pack_key(c[0], c[1], c[2], c[3]);
        int f; uint64_t s = hmap_slot(&h, key, &f);
        if (!f) {
            h.keys[s] = key; h.vals[s] = (int32_t)nu;
            first_of[nu] = (int32_t)i; count[nu] = 0;
            memcpy(out_coords + 4 * nu, c, 4 * sizeof(int32_t));
            nu++;
        }
        int32_t r = h.vals[s];
        row_of[i] = r; last_of[r] = (int32_t)i; count[r]++;
    }
    hmap_free(&h);
    return nu;
}

/*
 * scn.SubmanifoldConvolution rulebook [upstream, SURVEY.md 2.2]: output active
 * set == input active set; for each active site p (row o) and filter offset k
 * (row-major over filter, last dim fastest, centred: q = p - size//2 + off(k)),
 * if q is active (row i) emit rule (i -> o) in list k.
 * rules are written k-major: list k occupies [start[k], start[k+1]).
 * Two-pass (count, then fill).  Returns total rule count.
 */
int64_t oracle_subm_rules(const int32_t *coords, int64_t n, const int32_t *spatial,
                          const int32_t *filter, int64_t *start /*[K+1]*/,
                          int32_t *rules_in, int32_t *rules_out, int64_t cap) {
    hmap_t h;
    if (hmap_init(&h, n)) return -1;
    for (int64_t i = 0; i < n; i++) {
        const int32_t *c = coords + 4 * i;
        int f; uint64_t s = hmap_slot(&h, pack_key(c[0], c[1], c[2], c[3]), &f);
        h.keys[s] = pack_key(c[0], c[1], c[2], c[3]); h.vals[s] = (int32_t)i;
    }
    int K = filter[0] * filter[1] * filter[2];
    int64_t *cnt = (int64_t *)calloc(K + 1, sizeof(int64_t));
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            start[0] = 0;
            for (int k = 0; k < K; k++) { start[k + 1] = start[k] + cnt[k]; cnt[k] = 0; }
            if (start[K] > cap) { int64_t t = start[K]; free(cnt); hmap_free(&h); return t; }
        }
        for (int64_t o = 0; o < n; o++) {
            const int32_t *c = coords + 4 * o;
            int k = 0;
            for (int a = 0; a < filter[0]; a++)
                for (int b = 0; b < filter[1]; b++)
                    for (int d = 0; d < filter[2]; d++, k++) {
                        int32_t q0 = c[0] - filter[0] / 2 + a;
                        int32_t q1 = c[1] - filter[1] / 2 + b;
                        int32_t q2 = c[2] - filter[2] / 2 + d;
                        if (q0 < 0 || q0 >= spatial[0] || q1 < 0 || q1 >= spatial[1] ||
                            q2 < 0 || q2 >= spatial[2]) continue;
                        int32_t i = hmap_get(&h, pack_key(q0, q1, q2, c[3]));
                        if (i < 0) continue;
                        if (pass == 1) {
                            int64_t pos = start[k] + cnt[k];
                            rules_in[pos] = i; rules_out[pos] = (int32_t)o;
                        }
                        cnt[k]++;
                    }
        }
    }
    int64_t total = start[K];
    free(cnt); hmap_free(&h);
    return total;
}

/*
 * scn.Convolution (strided) [upstream, SURVEY.md 2.2]: out spatial =
 * (in - size)/stride + 1; output site q is active iff some active input p has
 * q*stride <= p < q*stride + size with q in [0,outSize); rule offset = row-major
 * index of (p - q*stride).  Output rows are numbered in first-appearance order
 * iterating input rows in order (and, per input, candidate q's row-major).
 * Returns N_out; *total_rules receives the rule count; rules k-major.
 * If N_out > out_cap or rules > cap nothing is written beyond capacity; caller
 * sizes out_cap = n * prod(ceil(size/stride)), cap likewise.
 */
int64_t oracle_conv_rules(const int32_t *coords, int64_t n, const int32_t *spatial,
                          const int32_t *size, const int32_t *stride,
                          int32_t *out_coords, int64_t out_cap, int64_t *start /*[K+1]*/,
                          int32_t *rules_in, int32_t *rules_out, int64_t cap,
                          int64_t *total_rules) {
    int32_t osz[3];
    for (int d = 0; d < 3; d++) osz[d] = (spatial[d] - size[d]) / stride[d] + 1;
    int K = size[0] * size[1] * size[2];
    hmap_t h;
    int64_t mult = 1;
    for (int d = 0; d < 3; d++) mult *= (size[d] + stride[d] - 1) / stride[d];
    if (hmap_init(&h, n * mult)) return -1;
    int64_t *cnt = (int64_t *)calloc(K + 1, sizeof(int64_t));
    int64_t nout = 0;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            start[0] = 0;
            for (int k = 0; k < K; k++) { start[k + 1] = start[k] + cnt[k]; cnt[k] = 0; }
        }
        for (int64_t i = 0; i < n; i++) {
            const int32_t *p = coords + 4 * i;
            int32_t lo[3], hi[3];
            for (int d = 0; d < 3; d++) {
                int32_t l = p[d] - size[d] + 1;            /* q*stride >= p-size+1 */
                lo[d] = l <= 0 ? 0 : (l + stride[d] - 1) / stride[d];
                hi[d] = p[d] / stride[d];
                if (hi[d] > osz[d] - 1) hi[d] = osz[d] - 1;
            }
            for (int32_t q0 = lo[0]; q0 <= hi[0]; q0++)
                for (int32_t q1 = lo[1]; q1 <= hi[1]; q1++)
                    for (int32_t q2 = lo[2]; q2 <= hi[2]; q2++) {
                        uint64_t key = pack_key(q0, q1, q2, p[3]);
                        int f; uint64_t s = hmap_slot(&h, key, &f);
                        if (!f) {
                            h.keys[s] = key; h.vals[s] = (int32_t)nout;
                            if (nout < out_cap) {
                                out_coords[4 * nout + 0] = q0; out_coords[4 * nout + 1] = q1;
                                out_coords[4 * nout + 2] = q2; out_coords[4 * nout + 3] = p[3];
                            }
                            nout++;
                        }
                        int k = ((p[0] - q0 * stride[0]) * size[1] + (p[1] - q1 * stride[1])) * size[2] +
                                (p[2] - q2 * stride[2]);
                        if (pass == 1) {
                            int64_t pos = start[k] + cnt[k];
                            if (pos < cap) { rules_in[pos] = (int32_t)i; rules_out[pos] = h.vals[s]; }
                        }
                        cnt[k]++;
                    }
        }
    }
    *total_rules = start[K];
    free(cnt); hmap_free(&h);
    return nout;
}
