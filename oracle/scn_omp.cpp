/*
 * oracle/scn_omp.cpp -- TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C++/OpenMP restatement of the SparseConvNet-CPU algorithm behind the reference's
 * `import sparseconvnet as scn` (/root/reference/src/networks/resnet.py:2, sparse_building_blocks.py:3):
 *   - rule lists with a host hash map (one per level) probed by all cores            [SCN getSubmanifoldRuleBook]
 *   - per filter offset k:  gather rows -> sgemm -> scatter-add                       [SCN Convolution_*_updateOutput]
 *   - backward: dIn[in] += dOut[out] W[k]^T ;  dW[k] += in[in]^T dOut[out]            [SCN ..._backward]
 * The third-party package is absent (no pinned version, not installed), so this is the TIMED CPU BASELINE of
 * bench.py ("CPU restatement of the reference algorithm (SCN binary unavailable)"), validated against the torch
 * restatement oracle/scn_cpu.py (tests/test_oracle_omp.py) -- which in turn is pinned by dense-convolution
 * equivalence (tests/test_oracle.py).  PARITY UNPINNED by the reference itself (it holds no fixtures for this path).
 *
 * sgemm comes from the OpenBLAS that ships inside the scipy wheel, loaded with dlopen (runtime CPU dispatch, no
 * -march=native here: the library is built in the build container and used on the GPU box's host CPU).  Each OpenMP
 * thread multiplies cache-sized blocks of rules single-threaded; rows of one offset are distinct, so the scatter
 * needs no atomics; dW is summed per thread and reduced in thread order (deterministic for a fixed thread count).
 *
 * Only tests/ and bench.py's cpu_baseline leg may load this library.
 */
#include <dlfcn.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

typedef void (*sgemm_fn)(int order, int ta, int tb, int M, int N, int K, float alpha, const float *A, int lda,
                         const float *B, int ldb, float beta, float *C, int ldc);
sgemm_fn g_sgemm = nullptr;
constexpr int RowMajor = 101, NoTrans = 111, Trans = 112;
constexpr int64_t BLOCK = 2048;  // rules per task: gathered block + product stay in L2

// fallback when no BLAS was loaded (tests on a box without scipy): plain loops
void sgemm_ref(int, int ta, int tb, int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb,
               float beta, float *C, int ldc) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float s = 0.f;
            for (int k = 0; k < K; k++) {
                const float a = ta == Trans ? A[(size_t)k * lda + i] : A[(size_t)i * lda + k];
                const float b = tb == Trans ? B[(size_t)j * ldb + k] : B[(size_t)k * ldb + j];
                s += a * b;
            }
            C[(size_t)i * ldc + j] = alpha * s + (beta == 0.f ? 0.f : beta * C[(size_t)i * ldc + j]);
        }
}

inline sgemm_fn gemm() { return g_sgemm ? g_sgemm : sgemm_ref; }

struct HMap {
    std::vector<uint64_t> keys;
    std::vector<int32_t> vals;
    uint64_t mask;
};
constexpr uint64_t EMPTY = 0xFFFFFFFFFFFFFFFFull;
inline uint64_t pack_key(int32_t x0, int32_t x1, int32_t x2, int32_t b) {
    return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)x0 << 32) | ((uint64_t)(uint16_t)x1 << 16) |
           (uint64_t)(uint16_t)x2;
}
inline uint64_t mix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}
void hmap_build(HMap &h, const int32_t *coords, int64_t n) {
    uint64_t cap = 16;
    while (cap < (uint64_t)(2 * n + 1)) cap <<= 1;
    h.keys.assign(cap, EMPTY);
    h.vals.assign(cap, -1);
    h.mask = cap - 1;
    for (int64_t i = 0; i < n; i++) {  // serial insert: rows are unique, order irrelevant
        const int32_t *c = coords + 4 * i;
        const uint64_t key = pack_key(c[0], c[1], c[2], c[3]);
        uint64_t s = mix64(key) & h.mask;
        while (h.keys[s] != EMPTY && h.keys[s] != key) s = (s + 1) & h.mask;
        h.keys[s] = key;
        h.vals[s] = (int32_t)i;
    }
}
inline int32_t hmap_get(const HMap &h, uint64_t key) {
    uint64_t s = mix64(key) & h.mask;
    for (;;) {
        if (h.keys[s] == key) return h.vals[s];
        if (h.keys[s] == EMPTY) return -1;
        s = (s + 1) & h.mask;
    }
}

}  // namespace

extern "C" {

/* dlopen an OpenBLAS that exports `<prefix>cblas_sgemm` / `<prefix>openblas_set_num_threads`; returns 0 on success */
int omp_load_blas(const char *path, const char *prefix) {
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return 1;
    char name[128];
    snprintf(name, sizeof name, "%scblas_sgemm", prefix);
    sgemm_fn f = (sgemm_fn)dlsym(h, name);
    if (!f) return 2;
    snprintf(name, sizeof name, "%sopenblas_set_num_threads", prefix);
    void (*setn)(int) = (void (*)(int))dlsym(h, name);
    if (setn) setn(1);  // every OpenMP thread multiplies its own blocks
    g_sgemm = f;
    return 0;
}

int omp_threads(void) { return omp_get_max_threads(); }
void omp_set_threads(int n) { omp_set_num_threads(n); }

/*
 * scn.SubmanifoldConvolution rulebook, probed by all cores; output identical to oracle/scn_rules.c
 * (k-major lists, sites in row order within a list).  Returns the rule count (> cap: nothing written).
 */
int64_t omp_subm_rules(const int32_t *coords, int64_t n, const int32_t *spatial, const int32_t *filter, int64_t *start,
                       int32_t *rules_in, int32_t *rules_out, int64_t cap) {
    HMap h;
    hmap_build(h, coords, n);
    const int K = filter[0] * filter[1] * filter[2];
    const int T = omp_get_max_threads();
    std::vector<int64_t> cnt((size_t)T * K, 0);
    auto probe = [&](int64_t o, int k) -> int32_t {
        const int32_t *c = coords + 4 * o;
        const int a = k / (filter[1] * filter[2]), rem = k - a * filter[1] * filter[2];
        const int b = rem / filter[2], d = rem - b * filter[2];
        const int32_t q0 = c[0] - filter[0] / 2 + a, q1 = c[1] - filter[1] / 2 + b, q2 = c[2] - filter[2] / 2 + d;
        if (q0 < 0 || q0 >= spatial[0] || q1 < 0 || q1 >= spatial[1] || q2 < 0 || q2 >= spatial[2]) return -1;
        return hmap_get(h, pack_key(q0, q1, q2, c[3]));
    };
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        for (int64_t o = lo; o < hi; o++)
            for (int k = 0; k < K; k++)
                if (probe(o, k) >= 0) cnt[(size_t)t * K + k]++;
    }
    std::vector<int64_t> pos((size_t)T * K);
    int64_t run = 0;
    for (int k = 0; k < K; k++) {
        start[k] = run;
        for (int t = 0; t < T; t++) {
            pos[(size_t)t * K + k] = run;
            run += cnt[(size_t)t * K + k];
        }
    }
    start[K] = run;
    if (run > cap) return run;
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        std::vector<int64_t> p(pos.begin() + (size_t)t * K, pos.begin() + (size_t)(t + 1) * K);
        for (int64_t o = lo; o < hi; o++)
            for (int k = 0; k < K; k++) {
                const int32_t i = probe(o, k);
                if (i >= 0) {
                    rules_in[p[k]] = i;
                    rules_out[p[k]] = (int32_t)o;
                    p[k]++;
                }
            }
    }
    return run;
}

/* out[n_out, Cout] = sum_k scatter_add(ro_k, x[ri_k] @ W[k]);  out must be zero-initialised by the caller */
void omp_conv_fwd(const float *x, const float *W, const int64_t *start, const int32_t *ri, const int32_t *ro, int K, int Cin,
                  int Cout, float *out) {
    for (int k = 0; k < K; k++) {
        const int64_t a = start[k], b = start[k + 1];
        const int64_t nb = (b - a + BLOCK - 1) / BLOCK;
        const float *Wk = W + (size_t)k * Cin * Cout;
#pragma omp parallel
        {
            std::vector<float> xs((size_t)BLOCK * Cin), ys((size_t)BLOCK * Cout);
#pragma omp for schedule(dynamic, 1)
            for (int64_t blk = 0; blk < nb; blk++) {
                const int64_t r0 = a + blk * BLOCK, m = (b - r0 < BLOCK ? b - r0 : BLOCK);
                for (int64_t r = 0; r < m; r++) memcpy(&xs[(size_t)r * Cin], x + (size_t)ri[r0 + r] * Cin, Cin * sizeof(float));
                gemm()(RowMajor, NoTrans, NoTrans, (int)m, Cout, Cin, 1.f, xs.data(), Cin, Wk, Cout, 0.f, ys.data(), Cout);
                for (int64_t r = 0; r < m; r++) {  // rows of one offset are distinct: no atomics
                    float *o = out + (size_t)ro[r0 + r] * Cout;
                    const float *y = &ys[(size_t)r * Cout];
                    for (int c = 0; c < Cout; c++) o[c] += y[c];
                }
            }
        }
    }
}

/* dx[n_in, Cin] (zero-initialised by the caller; may be null) and dW[K, Cin, Cout] (overwritten) */
void omp_conv_bwd(const float *x, const float *W, const float *dout, const int64_t *start, const int32_t *ri,
                  const int32_t *ro, int K, int Cin, int Cout, float *dx, float *dW) {
    const int T = omp_get_max_threads();
    std::vector<float> part((size_t)T * Cin * Cout);
    for (int k = 0; k < K; k++) {
        const int64_t a = start[k], b = start[k + 1];
        const int64_t nb = (b - a + BLOCK - 1) / BLOCK;
        const float *Wk = W + (size_t)k * Cin * Cout;
        std::fill(part.begin(), part.end(), 0.f);
#pragma omp parallel num_threads(T)
        {
            std::vector<float> xs((size_t)BLOCK * Cin), gs((size_t)BLOCK * Cout), ds((size_t)BLOCK * Cin);
            float *mine = &part[(size_t)omp_get_thread_num() * Cin * Cout];
#pragma omp for schedule(static)
            for (int64_t blk = 0; blk < nb; blk++) {
                const int64_t r0 = a + blk * BLOCK, m = (b - r0 < BLOCK ? b - r0 : BLOCK);
                for (int64_t r = 0; r < m; r++) {
                    memcpy(&xs[(size_t)r * Cin], x + (size_t)ri[r0 + r] * Cin, Cin * sizeof(float));
                    memcpy(&gs[(size_t)r * Cout], dout + (size_t)ro[r0 + r] * Cout, Cout * sizeof(float));
                }
                // dW[k] += xs^T gs
                gemm()(RowMajor, Trans, NoTrans, Cin, Cout, (int)m, 1.f, xs.data(), Cin, gs.data(), Cout, 1.f, mine, Cout);
                if (dx) {
                    gemm()(RowMajor, NoTrans, Trans, (int)m, Cin, Cout, 1.f, gs.data(), Cout, Wk, Cout, 0.f, ds.data(), Cin);
                    for (int64_t r = 0; r < m; r++) {  // input rows of one offset are distinct
                        float *o = dx + (size_t)ri[r0 + r] * Cin;
                        const float *d = &ds[(size_t)r * Cin];
                        for (int c = 0; c < Cin; c++) o[c] += d[c];
                    }
                }
            }
        }
        float *dWk = dW + (size_t)k * Cin * Cout;
        const int64_t per = (int64_t)Cin * Cout;
#pragma omp parallel for
        for (int64_t e = 0; e < per; e++) {
            float s = 0.f;
            for (int t = 0; t < T; t++) s += part[(size_t)t * per + e];  // thread order: deterministic
            dWk[e] = s;
        }
    }
}


/*
 * scn.BatchNormalization, training mode (sparse_building_blocks.py:39,122): statistics over all active rows
 * (biased variance, fp64 accumulation), y = gamma (x - mean) rsqrt(var + eps) + beta, then LeakyReLU(leak)
 * (leak = 1: none -- the reference applies scn.LeakyReLU as a separate module; SCN's own kernel has the slope
 * built in).  save_mean / save_invstd [C] are kept for the backward.
 */
void omp_bn_fwd(const float *x, int64_t n, int C, const float *gamma, const float *beta, float eps, float leak, float *y,
                float *save_mean, float *save_invstd, float *var_out) {
    const int T = omp_get_max_threads();
    std::vector<double> s1((size_t)T * C, 0.0), s2((size_t)T * C, 0.0);
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        double *a = &s1[(size_t)t * C], *b = &s2[(size_t)t * C];
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        for (int64_t r = lo; r < hi; r++) {
            const float *xr = x + (size_t)r * C;
            for (int c = 0; c < C; c++) { a[c] += xr[c]; b[c] += (double)xr[c] * xr[c]; }
        }
    }
    std::vector<float> scale(C), shift(C);
    for (int c = 0; c < C; c++) {
        double a = 0, b = 0;
        for (int t = 0; t < T; t++) { a += s1[(size_t)t * C + c]; b += s2[(size_t)t * C + c]; }
        const double mean = n ? a / n : 0.0, var = n ? b / n - mean * mean : 0.0;
        const double inv = 1.0 / sqrt((var > 0 ? var : 0.0) + eps);
        save_mean[c] = (float)mean; save_invstd[c] = (float)inv; var_out[c] = (float)(var > 0 ? var : 0.0);
        scale[c] = (float)(gamma[c] * inv); shift[c] = (float)(beta[c] - gamma[c] * inv * mean);
    }
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; r++) {
        const float *xr = x + (size_t)r * C;
        float *yr = y + (size_t)r * C;
        for (int c = 0; c < C; c++) {
            const float z = xr[c] * scale[c] + shift[c];
            yr[c] = z > 0.f ? z : z * leak;
        }
    }
}

/* dx, dgamma, dbeta of the above; y is the forward output (its sign selects the LeakyReLU branch) */
void omp_bn_bwd(const float *x, const float *y, const float *dy, int64_t n, int C, const float *gamma,
                const float *save_mean, const float *save_invstd, float leak, float *dx, float *dgamma, float *dbeta) {
    const int T = omp_get_max_threads();
    std::vector<double> s1((size_t)T * C, 0.0), s2((size_t)T * C, 0.0);
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        double *a = &s1[(size_t)t * C], *b = &s2[(size_t)t * C];
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        for (int64_t r = lo; r < hi; r++) {
            const float *xr = x + (size_t)r * C, *yr = y + (size_t)r * C, *gr = dy + (size_t)r * C;
            for (int c = 0; c < C; c++) {
                const float dz = yr[c] > 0.f ? gr[c] : gr[c] * leak;
                a[c] += dz;
                b[c] += (double)dz * ((xr[c] - save_mean[c]) * save_invstd[c]);
            }
        }
    }
    std::vector<float> k1(C), k2(C), k3(C);
    for (int c = 0; c < C; c++) {
        double a = 0, b = 0;
        for (int t = 0; t < T; t++) { a += s1[(size_t)t * C + c]; b += s2[(size_t)t * C + c]; }
        dbeta[c] = (float)a; dgamma[c] = (float)b;
        k1[c] = gamma[c] * save_invstd[c];
        k2[c] = (float)(n ? a / n : 0.0);
        k3[c] = (float)(n ? b / n : 0.0);
    }
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; r++) {
        const float *xr = x + (size_t)r * C, *yr = y + (size_t)r * C, *gr = dy + (size_t)r * C;
        float *dr = dx + (size_t)r * C;
        for (int c = 0; c < C; c++) {
            const float dz = yr[c] > 0.f ? gr[c] : gr[c] * leak;
            const float xh = (xr[c] - save_mean[c]) * save_invstd[c];
            dr[c] = k1[c] * (dz - k2[c] - xh * k3[c]);
        }
    }
}

/* scn.LeakyReLU forward (dy == null) / backward (dx = dy * (y > 0 ? 1 : leak)) */
void omp_leaky(const float *x, const float *dy, int64_t count, float leak, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; i++) out[i] = dy ? (x[i] > 0.f ? dy[i] : dy[i] * leak) : (x[i] > 0.f ? x[i] : x[i] * leak);
}

}  // extern "C"
