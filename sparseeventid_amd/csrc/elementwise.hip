// HBM-bound row-major [N,C] kernels: BatchNormalization (+fused LeakyReLU / residual add),
// LeakyReLU, AddTable, SparseToDense  (sparse_building_blocks.py:39,80,96,122; resnet.py:124-125).
// float4 per lane, per-channel reductions accumulated in fp64 and combined in a fixed order
// (two-stage: per-block partials -> one finalize block), so results are run-to-run deterministic.
#include "common.h"

namespace sei {

#ifndef SEI_RED_BLOCKS_MAX
#define SEI_RED_BLOCKS_MAX 1024
#endif
constexpr int RED_BLOCKS_MAX = SEI_RED_BLOCKS_MAX;

// ------------------------------------------------------------------ column reductions
// MODE 0: (x, x*x)                         -> BN statistics
// MODE 1: (dz, dz*xhat), dz = dy*act'(y)   -> BN backward sums
// MODE 2: (x, 0)                           -> bias gradient
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void colreduce_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ dy, int64_t n, int C,
                                                        const float *__restrict__ mean,
                                                        const float *__restrict__ invstd, float leak,
                                                        double *__restrict__ partial /*[blocks][2][C]*/,
                                                        const float *__restrict__ gamma = nullptr,
                                                        const float *__restrict__ beta = nullptr) {
    extern __shared__ double lds_d[];  // [rl][2][C]
    const int cv = C / VEC;            // column groups
    const int rl = 256 / cv;           // row lanes
    const int col = threadIdx.x % cv, rlane = threadIdx.x / cv;
    double s0[VEC], s1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) s0[v] = s1[v] = 0.0;
    if (rlane < rl) {
        float mu[VEC], is[VEC], sc[VEC], sh[VEC];
        if (MODE == 1) {
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                mu[v] = mean[col * VEC + v];
                is[v] = invstd[col * VEC + v];
                // y == null: the activation mask is recomputed from x exactly as the forward computed z (bn_apply_kernel)
                sc[v] = is[v] * (gamma ? gamma[col * VEC + v] : 1.f);
                sh[v] = beta ? beta[col * VEC + v] : 0.f;
            }
        }
        const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
        const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
        for (int64_t r = r0 + rlane; r < r1; r += rl) {
            float xv[VEC], yv[VEC], dv[VEC];
            const size_t off = (size_t)r * C + col * VEC;
            if (VEC == 4) {
                float4 t = *reinterpret_cast<const float4 *>(x + off);
                xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                if (MODE == 1) {
                    if (y) {
                        float4 u = *reinterpret_cast<const float4 *>(y + off);
                        yv[0] = u.x; yv[1] = u.y; yv[2] = u.z; yv[3] = u.w;
                    }
                    float4 w = *reinterpret_cast<const float4 *>(dy + off);
                    dv[0] = w.x; dv[1] = w.y; dv[2] = w.z; dv[3] = w.w;
                }
            } else {
                xv[0] = x[off];
                if (MODE == 1) {
                    if (y) yv[0] = y[off];
                    dv[0] = dy[off];
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                if (MODE == 0) {
                    double d = (double)xv[v];
                    s0[v] += d;
                    s1[v] += d * d;
                } else if (MODE == 1) {
                    if (!y) yv[v] = (xv[v] - mu[v]) * sc[v] + sh[v];
                    float dz = yv[v] > 0.f ? dv[v] : dv[v] * leak;
                    float xh = (xv[v] - mu[v]) * is[v];
                    s0[v] += (double)dz;
                    s1[v] += (double)dz * (double)xh;
                } else {
                    s0[v] += (double)xv[v];
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            lds_d[(rlane * 2 + 0) * C + col * VEC + v] = s0[v];
            lds_d[(rlane * 2 + 1) * C + col * VEC + v] = s1[v];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        double acc = 0.0;
        for (int q = 0; q < rl; q++) acc += lds_d[q * 2 * C + e];
        partial[(size_t)blockIdx.x * 2 * C + e] = acc;
    }
}

// finalize kinds
//  0: BN stats -> save_mean, save_invstd, running update
//  1: BN backward -> coef[0][C]=sum_dz/n, coef[1][C]=sum_dzxhat/n ; dgamma, dbeta
//  2: colsum -> out[C]
__global__ __launch_bounds__(64) void finalize_kernel(const double *__restrict__ partial, int blocks, int C, int64_t n,
                                                      int kind, float eps, float momentum, float *a0, float *a1,
                                                      float *a2, float *a3) {
    // one wavefront per channel: lane l sums partials l, l+64, ... then a fixed-shape shuffle tree
    const int c = blockIdx.x;
    const int lane = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int b = lane; b < blocks; b += 64) {
        s0 += partial[(size_t)b * 2 * C + c];
        s1 += partial[(size_t)b * 2 * C + C + c];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_down(s0, d);
        s1 += __shfl_down(s1, d);
    }
    if (lane != 0) return;
    if (kind == 0) {
        double mean = s0 / (double)n;
        double var = s1 / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        a0[c] = (float)mean;                             // save_mean
        a1[c] = (float)(1.0 / sqrt(var + (double)eps));  // save_invstd
        if (a2) {                                        // running stats (SCN: r = m*r + (1-m)*batch, unbiased var)
            double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
            a2[c] = momentum * a2[c] + (1.f - momentum) * (float)mean;
            a3[c] = momentum * a3[c] + (1.f - momentum) * (float)unb;
        }
    } else if (kind == 1) {
        a0[c] = (float)(s0 / (double)n);
        a1[c] = (float)(s1 / (double)n);
        if (a2) a2[c] = (float)s1;  // dgamma
        if (a3) a3[c] = (float)s0;  // dbeta
    } else {
        a0[c] = (float)s0;
    }
}

// Many per-tile partial sums (conv epilogue) -> <= STATS_ROWS rows, coalesced: block j sums a contiguous run of rows,
// thread = column; fixed order.  (finalize_kernel reads one column per wave: fed 50 000 rows directly it is a
// latency-bound 8-byte-strided walk.)
constexpr int STATS_ROWS = 512;
constexpr int COMPRESS_THREADS = 1024;
__global__ __launch_bounds__(COMPRESS_THREADS) void stats_compress_kernel(const double *__restrict__ in, int64_t n_rows, int cols,
                                                             double *__restrict__ out) {
    // block b sums rows [r0, r1) of `in` per column.  The 1024 threads are cols x RL (row lanes): row lane q takes rows
    // r0 + q, r0 + q + RL, ... (RL independent chains instead of one latency-bound one), and the RL partial sums of a
    // column are combined in lane order -- a fixed order, bit-identical reruns.
    __shared__ double part[COMPRESS_THREADS];
    const int64_t per = (n_rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n_rows ? r0 + per : n_rows;
    const int RL = cols <= COMPRESS_THREADS / 2 ? COMPRESS_THREADS / cols : 1;
    if (RL > 1) {
        const int c = threadIdx.x % cols, q = threadIdx.x / cols;
        double s = 0.0;
        if (q < RL)
            for (int64_t r = r0 + q; r < r1; r += RL) s += in[r * cols + c];
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < cols) {
            double t = part[threadIdx.x];
            for (int j = 1; j < RL; j++) t += part[j * cols + threadIdx.x];
            out[(size_t)blockIdx.x * cols + threadIdx.x] = t;
        }
        return;
    }
    for (int c = threadIdx.x; c < cols; c += COMPRESS_THREADS) {
        double s = 0.0;
        for (int64_t r = r0; r < r1; r++) s += in[r * cols + c];
        out[(size_t)blockIdx.x * cols + c] = s;
    }
}

static int reduce_launch(int mode, const float *x, const float *y, const float *dy, int64_t n, int C, const float *mean,
                         const float *invstd, float leak, double **partial_out, int *blocks_out, hipStream_t st,
                         const float *gamma = nullptr, const float *beta = nullptr) {
    int vec = (C % 4 == 0) ? 4 : 1;
    int cv = C / vec;
    if (cv > 256) SEI_FAIL("channel count too large for the reduction kernel (max 1024 with C%4==0, else 256)");
    int rl = 256 / cv;
    int64_t want = (n + (int64_t)rl * 8 - 1) / ((int64_t)rl * 8);
    int blocks = (int)(want < 1 ? 1 : (want > RED_BLOCKS_MAX ? RED_BLOCKS_MAX : want));
    double *partial;
    SEI_TRY(pool_alloc_t(&partial, (size_t)blocks * 2 * C, st));
    size_t lds = (size_t)rl * 2 * C * sizeof(double);
#define RL(MODE, VEC) colreduce_kernel<MODE, VEC><<<blocks, 256, lds, st>>>(x, y, dy, n, C, mean, invstd, leak, partial, gamma, beta)
    if (vec == 4) {
        if (mode == 0) RL(0, 4); else if (mode == 1) RL(1, 4); else RL(2, 4);
    } else {
        if (mode == 0) RL(0, 1); else if (mode == 1) RL(1, 1); else RL(2, 1);
    }
#undef RL
    SEI_LAUNCH_CHECK();
    *partial_out = partial;
    *blocks_out = blocks;
    return 0;
}

int colsum(const float *x, int64_t n, int C, float *out, hipStream_t st) {
    if (n == 0) {
        SEI_HIP(hipMemsetAsync(out, 0, C * sizeof(float), st));
        return 0;
    }
    double *partial;
    int blocks;
    SEI_TRY(reduce_launch(2, x, nullptr, nullptr, n, C, nullptr, nullptr, 1.f, &partial, &blocks, st));
    finalize_kernel<<<C, 64, 0, st>>>(partial, blocks, C, n, 2, 0.f, 0.f, out, nullptr, nullptr, nullptr);
    SEI_LAUNCH_CHECK();
    pool_free(partial);
    return 0;
}

// ------------------------------------------------------------------ BN apply
template <int VEC>
__global__ void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ res, float *__restrict__ y,
                                int64_t total_vec, int C, const float *__restrict__ mean,
                                const float *__restrict__ invstd, const float *__restrict__ gamma,
                                const float *__restrict__ beta, float leak) {
    // the host picks the grid so that stride*VEC is a multiple of C: every thread keeps ONE channel group and
    // its parameters live in registers (no per-element modulo / parameter loads)
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int c = (int)((t * VEC) % C);
    float sc[VEC], sh[VEC], mu[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        sc[v] = invstd[c + v] * (gamma ? gamma[c + v] : 1.f);
        sh[v] = beta ? beta[c + v] : 0.f;
        mu[v] = mean[c + v];
    }
    for (; t < total_vec; t += stride) {
        float xv[VEC], rv[VEC], o[VEC];
        if (VEC == 4) {
            float4 a = reinterpret_cast<const float4 *>(x)[t];
            xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
            if (res) {
                float4 b = reinterpret_cast<const float4 *>(res)[t];
                rv[0] = b.x; rv[1] = b.y; rv[2] = b.z; rv[3] = b.w;
            }
        } else {
            xv[0] = x[t];
            if (res) rv[0] = res[t];
        }
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            float z = (xv[v] - mu[v]) * sc[v] + sh[v];
            if (res) z += rv[v];
            o[v] = z > 0.f ? z : z * leak;
        }
        if (VEC == 4)
            reinterpret_cast<float4 *>(y)[t] = make_float4(o[0], o[1], o[2], o[3]);
        else
            y[t] = o[0];
    }
}

template <int VEC>
__global__ void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                    const float *__restrict__ dy, float *__restrict__ dx, float *__restrict__ dres,
                                    int64_t total_vec, int C, const float *__restrict__ mean,
                                    const float *__restrict__ invstd, const float *__restrict__ gamma,
                                    const float *__restrict__ coef /*[2][C] or null (eval)*/, float leak,
                                    double *__restrict__ dxsum /*[blocks][C] or null*/,
                                    const float *__restrict__ beta /*only read when y == null*/) {
    __shared__ double colpart[256 * VEC];
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int c = (int)((t * VEC) % C);  // constant per thread: the grid is sized so that stride*VEC % C == 0
    float cs[VEC];  // a thread sums only a few dozen values of one channel group: fp32 here, fp64 across threads
#pragma unroll
    for (int v = 0; v < VEC; v++) cs[v] = 0.f;
    float gs[VEC], is_[VEC], mu[VEC], c0[VEC], c1[VEC], sc[VEC], sh[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        is_[v] = invstd[c + v];
        gs[v] = (gamma ? gamma[c + v] : 1.f) * is_[v];
        sc[v] = is_[v] * (gamma ? gamma[c + v] : 1.f);  // y == null: mask from z recomputed as bn_apply_kernel computed it
        sh[v] = (!y && beta) ? beta[c + v] : 0.f;
        mu[v] = mean[c + v];
        c0[v] = coef ? coef[c + v] : 0.f;
        c1[v] = coef ? coef[C + c + v] : 0.f;
    }
    for (; t < total_vec; t += stride) {
        float xv[VEC], yv[VEC], dv[VEC], o[VEC], dz[VEC];
        if (VEC == 4) {
            float4 a = reinterpret_cast<const float4 *>(x)[t];
            xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
            if (y) {
                float4 b = reinterpret_cast<const float4 *>(y)[t];
                yv[0] = b.x; yv[1] = b.y; yv[2] = b.z; yv[3] = b.w;
            }
            float4 d = reinterpret_cast<const float4 *>(dy)[t];
            dv[0] = d.x; dv[1] = d.y; dv[2] = d.z; dv[3] = d.w;
        } else {
            xv[0] = x[t]; if (y) yv[0] = y[t]; dv[0] = dy[t];
        }
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            if (!y) yv[v] = (xv[v] - mu[v]) * sc[v] + sh[v];
            dz[v] = yv[v] > 0.f ? dv[v] : dv[v] * leak;
            if (coef) {
                float xh = (xv[v] - mu[v]) * is_[v];
                o[v] = gs[v] * (dz[v] - c0[v] - xh * c1[v]);
            } else {
                o[v] = gs[v] * dz[v];
            }
        }
        if (VEC == 4) {
            reinterpret_cast<float4 *>(dx)[t] = make_float4(o[0], o[1], o[2], o[3]);
            if (dres) reinterpret_cast<float4 *>(dres)[t] = make_float4(dz[0], dz[1], dz[2], dz[3]);
        } else {
            dx[t] = o[0];
            if (dres) dres[t] = dz[0];
        }
        if (dxsum) {
#pragma unroll
            for (int v = 0; v < VEC; v++) cs[v] += o[v];
        }
    }
    if (dxsum) {
        // column sums of dx (= the bias gradient of the convolution that produced x) while dx is in registers:
        // every thread owns one channel group; the block's threads of one channel are added in thread order
        // (fixed order), one row of partial sums per block.  Needs C % VEC == 0 (host checks).
#pragma unroll
        for (int v = 0; v < VEC; v++) colpart[threadIdx.x * VEC + v] = (double)cs[v];
        __syncthreads();
        const int P = C / VEC;  // threads per channel period
        const int64_t base = (int64_t)blockIdx.x * blockDim.x;
        for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
            const int gq = ch / VEC, v = ch - gq * VEC;
            int i0 = (int)(((int64_t)gq - base % P + P) % P);
            double acc = 0.0;
            for (int i = i0; i < (int)blockDim.x; i += P) acc += colpart[i * VEC + v];
            dxsum[(size_t)blockIdx.x * C + ch] = acc;
        }
    }
}

// out[c] = sum over rows b of partial[b][C] (one wave per channel, fixed-shape tree)
__global__ __launch_bounds__(64) void colsum_rows_kernel(const double *__restrict__ partial, int rows, int C, float *out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int b = lane; b < rows; b += 64) s += partial[(size_t)b * C + c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
    if (lane == 0) out[c] = (float)s;
}

__global__ void eval_stats_kernel(const float *rmean, const float *rvar, float eps, int C, float *save_mean,
                                  float *save_invstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    save_mean[c] = rmean[c];
    save_invstd[c] = 1.f / sqrtf(rvar[c] + eps);
}

int bn_eval_stats(const float *rmean, const float *rvar, float eps, int C, float *mean, float *invstd, hipStream_t st) {
    eval_stats_kernel<<<(C + 63) / 64, 64, 0, st>>>(rmean, rvar, eps, C, mean, invstd);
    SEI_LAUNCH_CHECK();
    return 0;
}

static inline unsigned ew_blocks(int64_t total_vec) {
    int64_t b = (total_vec + 255) / 256;
    // grid-stride kernels: 3 workgroups per CU (768 on the 256-CU MI355X) measured best for the streaming BatchNorm /
    // activation passes -- whole-step BatchNorm time 29.7 ms at 8192 blocks, 29.1 at 1536, 28.2 at 1024 and at 512, 27.5 at 768
    // (tools/bench_ab.sh, one box; profiles/r03_elementwise_grid.log): fewer, longer-lived blocks keep each CU on long
    // contiguous runs of its streams
#ifndef SEI_EW_BLOCKS_MAX
#define SEI_EW_BLOCKS_MAX 768
#endif
    return (unsigned)(b > SEI_EW_BLOCKS_MAX ? SEI_EW_BLOCKS_MAX : (b < 1 ? 1 : b));
}

// grid for the per-channel kernels: a multiple of m = C / gcd(C, 256*vec) blocks, so that the grid stride (in
// floats) is a multiple of C and every thread keeps its channel group
static inline unsigned bn_blocks(int64_t total_vec, int C, int vec) {
    int a = C, b = 256 * vec;
    while (b) { int r = a % b; a = b; b = r; }
    const unsigned m = (unsigned)(C / a);
    unsigned blocks = ew_blocks(total_vec);
    blocks = blocks / m * m;
    return blocks < m ? m : blocks;
}

int bn_forward(const float *x, int64_t n, int C, const float *gamma, const float *beta, float *rmean, float *rvar,
               float eps, float momentum, int training, float leak, const float *residual, float *y, float *save_mean,
               float *save_invstd, hipStream_t st, const double *stats, int64_t n_parts) {
    ProfScope ps(PROF_BN, st, 0.0, ((residual ? 16.0 : 12.0) - (training && stats ? 4.0 : 0.0)) * (double)n * C);
    if (training) {
        if (n == 0) return 0;
        if (stats) {
            // per-tile partial sums written by the convolution's epilogue: only the (fixed-order) finalize is left
            const double *src = stats;
            double *small = nullptr;
            int rows = (int)n_parts;
            if (n_parts > 2 * STATS_ROWS) {
                SEI_TRY(pool_alloc_t(&small, (size_t)STATS_ROWS * 2 * C, st));
                stats_compress_kernel<<<STATS_ROWS, COMPRESS_THREADS, 0, st>>>(stats, n_parts, 2 * C, small);
                src = small;
                rows = STATS_ROWS;
            }
            finalize_kernel<<<C, 64, 0, st>>>(src, rows, C, n, 0, eps, momentum, save_mean, save_invstd, rmean, rvar);
            SEI_LAUNCH_CHECK();
            pool_free(small);
        } else {
            double *partial;
            int blocks;
            SEI_TRY(reduce_launch(0, x, nullptr, nullptr, n, C, nullptr, nullptr, 1.f, &partial, &blocks, st));
            finalize_kernel<<<C, 64, 0, st>>>(partial, blocks, C, n, 0, eps, momentum, save_mean, save_invstd,
                                                          rmean, rvar);
            SEI_LAUNCH_CHECK();
            pool_free(partial);
        }
    } else {
        if (!rmean || !rvar) SEI_FAIL("BatchNormalization eval mode needs running statistics");
        eval_stats_kernel<<<(C + 63) / 64, 64, 0, st>>>(rmean, rvar, eps, C, save_mean, save_invstd);
        SEI_LAUNCH_CHECK();
        if (n == 0) return 0;
    }
    if (C % 4 == 0) {
        int64_t tv = n * C / 4;
        bn_apply_kernel<4><<<bn_blocks(tv, C, 4), 256, 0, st>>>(x, residual, y, tv, C, save_mean, save_invstd, gamma, beta, leak);
    } else {
        int64_t tv = n * C;
        bn_apply_kernel<1><<<bn_blocks(tv, C, 1), 256, 0, st>>>(x, residual, y, tv, C, save_mean, save_invstd, gamma, beta, leak);
    }
    SEI_LAUNCH_CHECK();
    return 0;
}

int bn_backward(const float *x, const float *y, const float *dy, int64_t n, int C, const float *gamma,
                const float *save_mean, const float *save_invstd, float leak, int training, float *dx, float *dres,
                float *dgamma, float *dbeta, hipStream_t st, float *dx_colsum, const float *beta) {
    if (!y && dres) SEI_FAIL("bn_backward: y may only be omitted when the forward had no residual");
    // bytes: the reduction reads x, (y), dy; the apply pass reads them again and writes dx (and dresidual)
    ProfScope ps(PROF_BN, st, 0.0, ((y ? 28.0 : 20.0) + (dres ? 4.0 : 0.0)) * (double)n * C);
    if (n == 0) {
        if (dx_colsum) SEI_HIP(hipMemsetAsync(dx_colsum, 0, C * sizeof(float), st));
        if (dgamma) SEI_HIP(hipMemsetAsync(dgamma, 0, C * sizeof(float), st));
        if (dbeta) SEI_HIP(hipMemsetAsync(dbeta, 0, C * sizeof(float), st));
        return 0;
    }
    float *coef;
    SEI_TRY(pool_alloc_t(&coef, 2 * C, st));
    double *partial;
    int blocks;
    SEI_TRY(reduce_launch(1, x, y, dy, n, C, save_mean, save_invstd, leak, &partial, &blocks, st, gamma, beta));
    finalize_kernel<<<C, 64, 0, st>>>(partial, blocks, C, n, 1, 0.f, 0.f, coef, coef + C, dgamma, dbeta);
    SEI_LAUNCH_CHECK();
    pool_free(partial);
    const float *cf = training ? coef : nullptr;
    const int vec = C % 4 == 0 ? 4 : 1;
    const int64_t tv = n * C / vec;
    const unsigned nb = bn_blocks(tv, C, vec);
    double *dxs = nullptr;
    if (dx_colsum) SEI_TRY(pool_alloc_t(&dxs, (size_t)nb * C, st));
    if (vec == 4)
        bn_bwd_apply_kernel<4><<<nb, 256, 0, st>>>(x, y, dy, dx, dres, tv, C, save_mean, save_invstd, gamma, cf, leak, dxs, beta);
    else
        bn_bwd_apply_kernel<1><<<nb, 256, 0, st>>>(x, y, dy, dx, dres, tv, C, save_mean, save_invstd, gamma, cf, leak, dxs, beta);
    if (dx_colsum) {
        if (nb > 2 * STATS_ROWS) {  // coalesced compression first: a wave per channel walking 8192 rows is latency-bound
            double *small;
            SEI_TRY(pool_alloc_t(&small, (size_t)STATS_ROWS * C, st));
            stats_compress_kernel<<<STATS_ROWS, COMPRESS_THREADS, 0, st>>>(dxs, nb, C, small);
            colsum_rows_kernel<<<C, 64, 0, st>>>(small, STATS_ROWS, C, dx_colsum);
            pool_free(small);
        } else {
            colsum_rows_kernel<<<C, 64, 0, st>>>(dxs, (int)nb, C, dx_colsum);
        }
    }
    SEI_LAUNCH_CHECK();
    pool_free(dxs);
    pool_free(coef);
    return 0;
}

// ------------------------------------------------------------------ LeakyReLU / AddTable
__global__ void leaky_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n4, int64_t count, float leak) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t; i < n4; i += stride) {
        float4 a = reinterpret_cast<const float4 *>(x)[i];
        a.x = a.x > 0.f ? a.x : a.x * leak; a.y = a.y > 0.f ? a.y : a.y * leak;
        a.z = a.z > 0.f ? a.z : a.z * leak; a.w = a.w > 0.f ? a.w : a.w * leak;
        reinterpret_cast<float4 *>(y)[i] = a;
    }
    for (int64_t i = n4 * 4 + t; i < count; i += stride) y[i] = x[i] > 0.f ? x[i] : x[i] * leak;
}

__global__ void leaky_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy, float *__restrict__ dx,
                                 int64_t n4, int64_t count, float leak) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t; i < n4; i += stride) {
        float4 a = reinterpret_cast<const float4 *>(y)[i];
        float4 d = reinterpret_cast<const float4 *>(dy)[i];
        d.x = a.x > 0.f ? d.x : d.x * leak; d.y = a.y > 0.f ? d.y : d.y * leak;
        d.z = a.z > 0.f ? d.z : d.z * leak; d.w = a.w > 0.f ? d.w : d.w * leak;
        reinterpret_cast<float4 *>(dx)[i] = d;
    }
    for (int64_t i = n4 * 4 + t; i < count; i += stride) dx[i] = y[i] > 0.f ? dy[i] : dy[i] * leak;
}

__global__ void add_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ o, int64_t n4,
                           int64_t count, float leak, int act) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t; i < n4; i += stride) {
        float4 u = reinterpret_cast<const float4 *>(a)[i];
        float4 v = reinterpret_cast<const float4 *>(b)[i];
        u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
        if (act) {
            u.x = u.x > 0.f ? u.x : u.x * leak; u.y = u.y > 0.f ? u.y : u.y * leak;
            u.z = u.z > 0.f ? u.z : u.z * leak; u.w = u.w > 0.f ? u.w : u.w * leak;
        }
        reinterpret_cast<float4 *>(o)[i] = u;
    }
    for (int64_t i = n4 * 4 + t; i < count; i += stride) {
        float s = a[i] + b[i];
        o[i] = (act && s <= 0.f) ? s * leak : s;
    }
}

int leaky_forward(const float *x, float *y, int64_t count, float leak, hipStream_t st) {
    if (count == 0) return 0;
    int64_t n4 = ((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0) ? count / 4 : 0;
    leaky_fwd_kernel<<<ew_blocks(count / 4 + 1), 256, 0, st>>>(x, y, n4, count, leak);
    SEI_LAUNCH_CHECK();
    return 0;
}
int leaky_backward(const float *y, const float *dy, float *dx, int64_t count, float leak, hipStream_t st) {
    if (count == 0) return 0;
    int64_t n4 = ((uintptr_t)y % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0) ? count / 4 : 0;
    leaky_bwd_kernel<<<ew_blocks(count / 4 + 1), 256, 0, st>>>(y, dy, dx, n4, count, leak);
    SEI_LAUNCH_CHECK();
    return 0;
}
int add_forward(const float *a, const float *b, float *out, int64_t count, float leak, bool act, hipStream_t st) {
    if (count == 0) return 0;
    int64_t n4 = ((uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0 && (uintptr_t)out % 16 == 0) ? count / 4 : 0;
    add_kernel<<<ew_blocks(count / 4 + 1), 256, 0, st>>>(a, b, out, n4, count, leak, act ? 1 : 0);
    SEI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ SparseToDense
// dense [B, C, D0, D1, D2]; thread per (row, channel) with channel fastest across lanes of a
// row group so the feature read is coalesced (the dense side is a 4-byte scatter by nature).
__global__ void sparse_to_dense_kernel(const int32_t *__restrict__ coords, int64_t n, int C, int D0, int D1, int D2,
                                       const float *__restrict__ feats, float *__restrict__ dense, int backward,
                                       float *__restrict__ dfeats) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    int64_t r = t / C;
    int c = (int)(t - r * C);
    int4 p = reinterpret_cast<const int4 *>(coords)[r];
    size_t vol = (size_t)D0 * D1 * D2;
    size_t off = ((size_t)p.w * C + c) * vol + ((size_t)p.x * D1 + p.y) * D2 + p.z;
    if (backward)
        dfeats[t] = dense[off];
    else
        dense[off] = feats[t];
}

// ------------------------------------------------------------------ fused sparse head
// AvgPool3d(tanh(SparseToDense(x))) over the full spatial extent == sum over ACTIVE rows of tanh(x) / V,
// because tanh(0) = 0: the [B,C,D0,D1,D2] dense tensor never has to exist.  out[b][c] += tanh(f[row][c]) / V.
// Fixed summation order (bit-identical reruns, like every other kernel of the engine; no float atomics): the rows
// are cut into POOL_SEGS contiguous segments, block (b, s) sums the rows of event b inside segment s in row order,
// and the final kernel adds the segment sums in segment order.
constexpr int POOL_SEGS = 32;
__global__ void pool_tanh_partial_kernel(const int32_t *__restrict__ coords, int64_t n, int C,
                                         const float *__restrict__ feats, float *__restrict__ partial) {
    const int b = blockIdx.x, s = blockIdx.y;
    const int64_t r0 = n * s / POOL_SEGS, r1 = n * (s + 1) / POOL_SEGS;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int64_t r = r0; r < r1; r++)
            if (coords[4 * r + 3] == b) acc += tanhf(feats[r * C + c]);
        partial[((size_t)b * POOL_SEGS + s) * C + c] = acc;
    }
}

__global__ void pool_tanh_final_kernel(const float *__restrict__ partial, int B, int C, float inv_vol,
                                       float *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    const int b = t / C, c = t - b * C;
    float acc = 0.f;
    for (int s = 0; s < POOL_SEGS; s++) acc += partial[((size_t)b * POOL_SEGS + s) * C + c];
    out[t] = acc * inv_vol;
}

__global__ void pool_tanh_bwd_kernel(const int32_t *__restrict__ coords, int64_t n, int C, float inv_vol,
                                     const float *__restrict__ feats, const float *__restrict__ dout,
                                     float *__restrict__ dfeats) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    int64_t r = t / C;
    int c = (int)(t - r * C);
    int b = coords[4 * r + 3];
    float th = tanhf(feats[t]);
    dfeats[t] = dout[(size_t)b * C + c] * inv_vol * (1.f - th * th);
}

int pool_tanh(const Level *lv, int batch, const float *feats, int C, float *out, const float *dout, float *dfeats,
              hipStream_t st) {
    const double vol = (double)lv->spatial[0] * lv->spatial[1] * lv->spatial[2];
    if (!dout) {
        float *partial;
        SEI_TRY(pool_alloc_t(&partial, (size_t)batch * POOL_SEGS * C, st));
        int threads = C < 64 ? 64 : (C > 1024 ? 1024 : (C + 63) / 64 * 64);
        pool_tanh_partial_kernel<<<dim3(batch, POOL_SEGS), threads, 0, st>>>(lv->coords, lv->n, C, feats, partial);
        pool_tanh_final_kernel<<<(batch * C + 255) / 256, 256, 0, st>>>(partial, batch, C, (float)(1.0 / vol), out);
        pool_free(partial);
        SEI_LAUNCH_CHECK();
        return 0;
    }
    if (lv->n == 0) return 0;
    int64_t tot = lv->n * C;
    unsigned nb = (unsigned)((tot + 255) / 256);
    pool_tanh_bwd_kernel<<<nb, 256, 0, st>>>(lv->coords, lv->n, C, (float)(1.0 / vol), feats, dout, dfeats);
    SEI_LAUNCH_CHECK();
    return 0;
}

int sparse_to_dense(const Level *lv, int batch, const float *feats, int C, float *dense, bool backward,
                    hipStream_t st) {
    size_t vol = (size_t)lv->spatial[0] * lv->spatial[1] * lv->spatial[2];
    if (!backward) SEI_HIP(hipMemsetAsync(dense, 0, (size_t)batch * C * vol * sizeof(float), st));
    if (lv->n == 0) return 0;
    int64_t tot = lv->n * C;
    sparse_to_dense_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(
        lv->coords, lv->n, C, lv->spatial[0], lv->spatial[1], lv->spatial[2], feats, dense, backward ? 1 : 0,
        const_cast<float *>(feats));
    SEI_LAUNCH_CHECK();
    return 0;
}

}  // namespace sei
