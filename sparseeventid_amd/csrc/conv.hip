// Sparse convolution arithmetic for gfx950 (scn.SubmanifoldConvolution / scn.Convolution:
// resnet.py:30-50,105-110; sparse_building_blocks.py:29-34,110-117).
//
// Output-stationary gather -> MFMA -> LDS-accumulate:
//   one wavefront owns a tile of 64 dst rows whose fp32 accumulators live in LDS;
//   it walks the tile's rule chunks (<=32 rules of ONE filter offset k, compacted),
//   gathers the src feature rows straight into MFMA A-operand registers (float4 per lane,
//   coalesced 64 B per rule row), multiplies by W[k] with v_mfma_f32_16x16x4_f32 (exact fp32)
//   and adds the 16xN result rows into the LDS tile (ds_add_f32, rows are wave-private).
//   Each dst row is written to HBM exactly once (no global atomics, run-to-run deterministic).
// dW: per (k, split) workgroup reduces X[src]^T dY[dst] over its rule chunks with the same
//   MFMA (rules are the contraction dimension), partial slabs are summed in a fixed order.
#include "common.h"

namespace sei {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ inline void lds_add(float *p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct ChunkIdx {
    int src0, src1, d0, d1, k;
};

__device__ inline ChunkIdx load_chunk(const int32_t *__restrict__ chunk_src, const uint8_t *__restrict__ chunk_dst,
                                      const uint8_t *__restrict__ chunk_k, int64_t c, int r16) {
    ChunkIdx ci;
    ci.src0 = chunk_src[c * CHUNK_SLOTS + r16];
    ci.src1 = chunk_src[c * CHUNK_SLOTS + 16 + r16];
    ci.d0 = chunk_dst[c * CHUNK_SLOTS + r16];
    ci.d1 = chunk_dst[c * CHUNK_SLOTS + 16 + r16];
    ci.k = chunk_k[c];
    return ci;
}

__device__ inline float4 load_a(const float *__restrict__ X, int src, int CIN, int off) {
    if (src >= 0) return *reinterpret_cast<const float4 *>(X + (size_t)src * CIN + off);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

// NT = 16-column tiles per wave.  colsplit waves share one dst tile (disjoint column ranges).
template <int NT>
__global__ __launch_bounds__(256) void conv_tile_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_chunk_start, const int32_t *__restrict__ chunk_src,
    const uint8_t *__restrict__ chunk_dst, const uint8_t *__restrict__ chunk_k, int n_tiles, int64_t n_dst, int CIN,
    int COUT, int colsplit) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_per_wg = 4 / colsplit;
    const int tw = wave / colsplit, cg = wave - tw * colsplit;
    const int tile = blockIdx.x * tiles_per_wg + tw;
    if (tile >= n_tiles) return;  // no barriers below: waves are independent
    const int col0 = cg * NT * 16;
    constexpr int NC = NT * 16;
    float *acc = lds + (size_t)tw * TILE_ROWS * COUT;
    const int r16 = lane & 15, g = lane >> 4;

    for (int idx = lane; idx < TILE_ROWS * NC; idx += 64) {
        int r = idx / NC, c = idx - r * NC;
        acc[r * COUT + col0 + c] = bias ? bias[col0 + c] : 0.f;
    }

    const int c0 = tile_chunk_start[tile], c1 = tile_chunk_start[tile + 1];
    const int J = CIN >> 4;
    if (c0 < c1) {
        ChunkIdx cur = load_chunk(chunk_src, chunk_dst, chunk_k, c0, r16);
        ChunkIdx nxt = cur;
        if (c0 + 1 < c1) nxt = load_chunk(chunk_src, chunk_dst, chunk_k, c0 + 1, r16);
        float4 a0 = load_a(X, cur.src0, CIN, 4 * g), a1 = load_a(X, cur.src1, CIN, 4 * g);
        for (int c = c0; c < c1; c++) {
            const bool has1 = __any(cur.src1 >= 0);
            const float *Wk = W + (size_t)cur.k * CIN * COUT + col0 + r16;
            f32x4 acc0[NT], acc1[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) {
                acc0[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc1[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int j = 0; j < J; j++) {
                float4 a0n, a1n;
                if (j + 1 < J) {
                    a0n = load_a(X, cur.src0, CIN, 16 * (j + 1) + 4 * g);
                    a1n = load_a(X, cur.src1, CIN, 16 * (j + 1) + 4 * g);
                } else {
                    a0n = load_a(X, c + 1 < c1 ? nxt.src0 : -1, CIN, 4 * g);
                    a1n = load_a(X, c + 1 < c1 ? nxt.src1 : -1, CIN, 4 * g);
                }
                const float *wrow = Wk + (size_t)(16 * j + 4 * g) * COUT;
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w};
                const float av1[4] = {a1.x, a1.y, a1.z, a1.w};
                if (has1) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
#pragma unroll
                        for (int n = 0; n < NT; n++) {
                            float b = wrow[(size_t)e * COUT + 16 * n];
                            acc0[n] = mfma16(av0[e], b, acc0[n]);
                            acc1[n] = mfma16(av1[e], b, acc1[n]);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
#pragma unroll
                        for (int n = 0; n < NT; n++) {
                            float b = wrow[(size_t)e * COUT + 16 * n];
                            acc0[n] = mfma16(av0[e], b, acc0[n]);
                        }
                    }
                }
                a0 = a0n;
                a1 = a1n;
            }
            // D[row = 4g+i][col = r16] belongs to rule slot 4g+i of the sub-chunk
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int s0 = __shfl(cur.src0, 4 * g + i), dd0 = __shfl(cur.d0, 4 * g + i);
                if (s0 >= 0) {
                    float *p = acc + dd0 * COUT + col0 + r16;
#pragma unroll
                    for (int n = 0; n < NT; n++) lds_add(p + 16 * n, acc0[n][i]);
                }
            }
            if (has1) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    int s1 = __shfl(cur.src1, 4 * g + i), dd1 = __shfl(cur.d1, 4 * g + i);
                    if (s1 >= 0) {
                        float *p = acc + dd1 * COUT + col0 + r16;
#pragma unroll
                        for (int n = 0; n < NT; n++) lds_add(p + 16 * n, acc1[n][i]);
                    }
                }
            }
            cur = nxt;
            if (c + 2 < c1) nxt = load_chunk(chunk_src, chunk_dst, chunk_k, c + 2, r16);
        }
    }
    // epilogue: every dst row is written once, as whole float4 row segments
    constexpr int V = NC / 4;
    const int64_t row0 = (int64_t)tile * TILE_ROWS;
    for (int idx = lane; idx < TILE_ROWS * V; idx += 64) {
        int r = idx / V, v = idx - r * V;
        if (row0 + r < n_dst) {
            float4 val = *reinterpret_cast<const float4 *>(acc + r * COUT + col0 + 4 * v);
            *reinterpret_cast<float4 *>(Y + (size_t)(row0 + r) * COUT + col0 + 4 * v) = val;
        }
    }
}

// Any channel counts (stem Cin=1, odd test sizes): same tile/chunk walk, scalar FMA.
__global__ __launch_bounds__(64) void conv_tile_generic_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_chunk_start, const int32_t *__restrict__ chunk_src,
    const uint8_t *__restrict__ chunk_dst, const uint8_t *__restrict__ chunk_k, int n_tiles, int64_t n_dst, int CIN,
    int COUT) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    float *acc = lds;
    for (int idx = lane; idx < TILE_ROWS * COUT; idx += 64) acc[idx] = bias ? bias[idx % COUT] : 0.f;
    __syncthreads();
    const int c0 = tile_chunk_start[tile], c1 = tile_chunk_start[tile + 1];
    const int r16 = lane & 15, g = lane >> 4;
    for (int c = c0; c < c1; c++) {
        const int k = chunk_k[c];
        const float *Wk = W + (size_t)k * CIN * COUT;
        for (int sub = 0; sub < 2; sub++) {
            int src = chunk_src[(int64_t)c * CHUNK_SLOTS + 16 * sub + r16];
            int d = chunk_dst[(int64_t)c * CHUNK_SLOTS + 16 * sub + r16];
            if (src < 0) continue;
            const float *xr = X + (size_t)src * CIN;
            for (int co = g; co < COUT; co += 4) {
                float s = 0.f;
                for (int ci = 0; ci < CIN; ci++) s = fmaf(xr[ci], Wk[(size_t)ci * COUT + co], s);
                lds_add(&acc[d * COUT + co], s);
            }
        }
    }
    __syncthreads();
    const int64_t row0 = (int64_t)tile * TILE_ROWS;
    for (int idx = lane; idx < TILE_ROWS * COUT; idx += 64) {
        int r = idx / COUT;
        if (row0 + r < n_dst) Y[(size_t)row0 * COUT + idx] = acc[idx];
    }
}

__global__ void fill_bias_kernel(float *Y, int64_t n, int C, const float *bias) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * C) Y[t] = bias ? bias[t % C] : 0.f;
}

template <int NT>
static int launch_mfma(const TiledRules &t, const float *X, const float *W, const float *bias, float *Y, int nIn,
                       int nOut, int colsplit, hipStream_t st) {
    static bool attr_set = false;
    size_t lds_bytes = (size_t)(4 / colsplit) * TILE_ROWS * nOut * sizeof(float);
    if (!attr_set) {
        SEI_HIP(hipFuncSetAttribute((const void *)conv_tile_mfma_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr_set = true;
    }
    int tiles_per_wg = 4 / colsplit;
    int nb = (t.n_tiles + tiles_per_wg - 1) / tiles_per_wg;
    conv_tile_mfma_kernel<NT><<<nb, 256, lds_bytes, st>>>(X, W, bias, Y, t.tile_chunk_start, t.chunk_src, t.chunk_dst,
                                                          t.chunk_k, t.n_tiles, t.n_dst, nIn, nOut, colsplit);
    SEI_LAUNCH_CHECK();
    return 0;
}

int conv_forward(const TiledRules &t, bool /*unused*/, const float *X, const float *W, const float *bias, float *Y,
                 int nIn, int nOut, hipStream_t st) {
    if (t.n_dst == 0) return 0;
    if (t.n_chunks == 0) {
        int64_t tot = t.n_dst * nOut;
        fill_bias_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(Y, t.n_dst, nOut, bias);
        SEI_LAUNCH_CHECK();
        return 0;
    }
    bool mfma_ok = (nIn % 16 == 0) && (nOut % 16 == 0);
    int colsplit = 1, nt = nOut / 16;
    if (mfma_ok && nt > 6) {
        if (nOut % 32 == 0 && nOut / 32 <= 6) {
            colsplit = 2;
            nt = nOut / 32;
        } else if (nOut % 64 == 0 && nOut / 64 <= 6) {
            colsplit = 4;
            nt = nOut / 64;
        } else
            mfma_ok = false;
    }
    if (mfma_ok && (size_t)(4 / colsplit) * TILE_ROWS * nOut * sizeof(float) > 160 * 1024) mfma_ok = false;
    if (mfma_ok) {
        switch (nt) {
            case 1: return launch_mfma<1>(t, X, W, bias, Y, nIn, nOut, colsplit, st);
            case 2: return launch_mfma<2>(t, X, W, bias, Y, nIn, nOut, colsplit, st);
            case 3: return launch_mfma<3>(t, X, W, bias, Y, nIn, nOut, colsplit, st);
            case 4: return launch_mfma<4>(t, X, W, bias, Y, nIn, nOut, colsplit, st);
            case 5: return launch_mfma<5>(t, X, W, bias, Y, nIn, nOut, colsplit, st);
            case 6: return launch_mfma<6>(t, X, W, bias, Y, nIn, nOut, colsplit, st);
        }
    }
    size_t lds_bytes = (size_t)TILE_ROWS * nOut * sizeof(float);
    if (lds_bytes > 160 * 1024) SEI_FAIL("conv: nOut too large for the generic kernel");
    static bool attr_set = false;
    if (!attr_set) {
        SEI_HIP(hipFuncSetAttribute((const void *)conv_tile_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr_set = true;
    }
    conv_tile_generic_kernel<<<t.n_tiles, 64, lds_bytes, st>>>(X, W, bias, Y, t.tile_chunk_start, t.chunk_src,
                                                              t.chunk_dst, t.chunk_k, t.n_tiles, t.n_dst, nIn, nOut);
    SEI_LAUNCH_CHECK();
    return 0;
}

// Wt[k][co][ci] = W[flip ? K-1-k : k][ci][co]
__global__ void transform_weights_kernel(const float *__restrict__ W, float *__restrict__ Wt, int K, int nIn, int nOut,
                                         int flip) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t per = (int64_t)nIn * nOut;
    if (t >= K * per) return;
    int k = (int)(t / per);
    int rem = (int)(t - k * per);
    int co = rem / nIn, ci = rem - co * nIn;
    int ks = flip ? K - 1 - k : k;
    Wt[t] = W[(size_t)ks * per + (size_t)ci * nOut + co];
}

int conv_transform_weights(const float *W, float *Wt, int K, int nIn, int nOut, bool flip, hipStream_t st) {
    int64_t tot = (int64_t)K * nIn * nOut;
    transform_weights_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(W, Wt, K, nIn, nOut, flip ? 1 : 0);
    SEI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ dW
constexpr int DW_LIST_CAP = 2048;

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_dw_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ dY, const int32_t *__restrict__ tile_chunk_start,
    const int32_t *__restrict__ seg_start, const int32_t *__restrict__ chunk_src, const uint8_t *__restrict__ chunk_dst,
    int n_tiles, int CIN, int COUT, int tiles_per_split, int S, float *__restrict__ slab) {
    extern __shared__ float lds_f[];
    __shared__ int list_chunk[DW_LIST_CAP];
    __shared__ int list_tile[DW_LIST_CAP];
    __shared__ int scan_tmp[8];
    const int k = blockIdx.x, s = blockIdx.y, cig = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int ci0 = cig * MT * 16;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int t_begin = s * tiles_per_split;
    const int t_end = min(n_tiles, t_begin + tiles_per_split);
    for (int tb = t_begin; tb < t_end; tb += 256) {
        const int t = tb + threadIdx.x;
        int cnt = 0, base = 0;
        if (t < t_end) {
            int a = seg_start[(size_t)k * n_tiles + t], b = seg_start[(size_t)(k + 1) * n_tiles + t];
            cnt = b - a;
            base = tile_chunk_start[t] + a;
        }
        // block exclusive scan of cnt
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) scan_tmp[wave] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
        for (int w = 0; w < 4; w++) {
            int v = scan_tmp[w];
            if (w < wave) wbase += v;
            total += v;
        }
        int pos = wbase + incl - cnt;
        for (int q = 0; q < cnt; q++)
            if (pos + q < DW_LIST_CAP) {
                list_chunk[pos + q] = base + q;
                list_tile[pos + q] = t;
            }
        __syncthreads();
        // a tile holds at most 2 chunks per offset (64 rows / 32 slots) -> total <= 512 <= DW_LIST_CAP
        for (int li = wave; li < total; li += 4) {
            const int64_t c = list_chunk[li];
            const int64_t rowbase = (int64_t)list_tile[li] * TILE_ROWS;
#pragma unroll 2
            for (int q = 0; q < CHUNK_SLOTS / 4; q++) {
                int src = chunk_src[c * CHUNK_SLOTS + 4 * q + g];
                if (!__any(src >= 0)) continue;
                int dl = chunk_dst[c * CHUNK_SLOTS + 4 * q + g];
                float a[MT], b[NT];
                const float *xr = X + (size_t)(src >= 0 ? src : 0) * CIN + ci0 + r16;
                const float *yr = dY + (size_t)(rowbase + dl) * COUT + r16;
#pragma unroll
                for (int m = 0; m < MT; m++) a[m] = src >= 0 ? xr[16 * m] : 0.f;
#pragma unroll
                for (int n = 0; n < NT; n++) b[n] = src >= 0 ? yr[16 * n] : 0.f;
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int n = 0; n < NT; n++) acc[m][n] = mfma16(a[m], b[n], acc[m][n]);
            }
        }
        __syncthreads();
    }
    // cross-wave reduction in a fixed order (deterministic): red[ci][co]
    float *red = lds_f;
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int n = 0; n < NT; n++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        int idx = (16 * m + 4 * g + i) * COUT + 16 * n + r16;
                        if (w == 0)
                            red[idx] = acc[m][n][i];
                        else
                            red[idx] += acc[m][n][i];
                    }
        }
        __syncthreads();
    }
    float *out = slab + ((size_t)(k * S + s) * CIN + ci0) * COUT;
    for (int idx = threadIdx.x; idx < MT * 16 * COUT; idx += 256) out[idx] = red[idx];
}

// generic dW (any channel counts): one workgroup per (k, split); thread per (ci, co) element loop.
__global__ __launch_bounds__(256) void conv_dw_generic_kernel(
    const float *__restrict__ X, const float *__restrict__ dY, const int32_t *__restrict__ tile_chunk_start,
    const int32_t *__restrict__ seg_start, const int32_t *__restrict__ chunk_src, const uint8_t *__restrict__ chunk_dst,
    int n_tiles, int CIN, int COUT, int tiles_per_split, int S, float *__restrict__ slab) {
    const int k = blockIdx.x, s = blockIdx.y;
    const int t_begin = s * tiles_per_split;
    const int t_end = min(n_tiles, t_begin + tiles_per_split);
    float *out = slab + (size_t)(k * S + s) * CIN * COUT;
    for (int e = threadIdx.x; e < CIN * COUT; e += 256) {
        int ci = e / COUT, co = e - ci * COUT;
        float sum = 0.f;
        for (int t = t_begin; t < t_end; t++) {
            int64_t ca = tile_chunk_start[t] + seg_start[(size_t)k * n_tiles + t];
            int64_t cb = tile_chunk_start[t] + seg_start[(size_t)(k + 1) * n_tiles + t];
            for (int64_t sl = ca * CHUNK_SLOTS; sl < cb * CHUNK_SLOTS; sl++) {
                int src = chunk_src[sl];
                if (src >= 0)
                    sum = fmaf(X[(size_t)src * CIN + ci], dY[((size_t)t * TILE_ROWS + chunk_dst[sl]) * COUT + co], sum);
            }
        }
        out[e] = sum;
    }
}

__global__ void slab_reduce_kernel(const float *__restrict__ slab, float *__restrict__ dW, int K, int S, int64_t per) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= K * per) return;
    int k = (int)(t / per);
    int64_t e = t - k * per;
    float sum = 0.f;
    for (int s = 0; s < S; s++) sum += slab[((size_t)k * S + s) * per + e];
    dW[t] = sum;
}

template <int MT, int NT>
static int launch_dw(const TiledRules &t, const float *X, const float *dY, int nIn, int nOut, int tps, int S,
                     float *slab, hipStream_t st) {
    static bool attr_set = false;
    size_t lds_bytes = (size_t)MT * 16 * nOut * sizeof(float);
    if (!attr_set) {
        SEI_HIP(hipFuncSetAttribute((const void *)conv_dw_mfma_kernel<MT, NT>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    dim3 grid(t.K, S, nIn / (16 * MT));
    conv_dw_mfma_kernel<MT, NT><<<grid, 256, lds_bytes, st>>>(X, dY, t.tile_chunk_start, t.seg_start, t.chunk_src,
                                                              t.chunk_dst, t.n_tiles, nIn, nOut, tps, S, slab);
    SEI_LAUNCH_CHECK();
    return 0;
}

#define DW_CASE(MT_, NT_) \
    if (mt == MT_ && nt == NT_) return launch_dw<MT_, NT_>(t, X, dY, nIn, nOut, tps, S, slab, st);

static int dispatch_dw(int mt, int nt, const TiledRules &t, const float *X, const float *dY, int nIn, int nOut, int tps,
                       int S, float *slab, hipStream_t st) {
    DW_CASE(1, 1) DW_CASE(1, 2) DW_CASE(1, 3) DW_CASE(1, 4) DW_CASE(1, 5) DW_CASE(1, 6) DW_CASE(1, 8) DW_CASE(1, 10) DW_CASE(1, 12)
    DW_CASE(2, 1) DW_CASE(2, 2) DW_CASE(2, 3) DW_CASE(2, 4) DW_CASE(2, 5) DW_CASE(2, 6) DW_CASE(2, 8) DW_CASE(2, 10) DW_CASE(2, 12)
    DW_CASE(3, 1) DW_CASE(3, 2) DW_CASE(3, 3) DW_CASE(3, 4) DW_CASE(3, 5) DW_CASE(3, 6) DW_CASE(3, 8) DW_CASE(3, 10) DW_CASE(3, 12)
    DW_CASE(4, 1) DW_CASE(4, 2) DW_CASE(4, 3) DW_CASE(4, 4) DW_CASE(4, 5) DW_CASE(4, 6) DW_CASE(4, 8) DW_CASE(4, 10)
    SEI_FAIL("conv dW: no MFMA instantiation");
}

int conv_backward_weight(const TiledRules &t, const float *X, const float *dY, float *dW, int nIn, int nOut,
                         hipStream_t st) {
    const int64_t per = (int64_t)nIn * nOut;
    if (t.n_chunks == 0 || t.n_dst == 0) {
        SEI_HIP(hipMemsetAsync(dW, 0, (size_t)t.K * per * sizeof(float), st));
        return 0;
    }
    int nt = nOut / 16;
    bool mfma_ok = (nIn % 16 == 0) && (nOut % 16 == 0) &&
                   (nt == 1 || nt == 2 || nt == 3 || nt == 4 || nt == 5 || nt == 6 || nt == 8 || nt == 10 || nt == 12);
    int mt = 0;
    if (mfma_ok) {
        int ct = nIn / 16;
        for (int cand = 4; cand >= 1; cand--)
            if (ct % cand == 0 && cand * nt <= 40) {
                mt = cand;
                break;
            }
        if (!mt) mfma_ok = false;
    }
    // splits: enough workgroups to fill 256 CUs, bounded slab size
    int S = (t.n_tiles + 255) / 256;
    int target = (2048 + t.K - 1) / t.K;
    if (S > target) S = target;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    int tps = (t.n_tiles + S - 1) / S;
    S = (t.n_tiles + tps - 1) / tps;
    float *slab;
    SEI_TRY(pool_alloc_t(&slab, (size_t)t.K * S * per));
    int rc;
    if (mfma_ok)
        rc = dispatch_dw(mt, nt, t, X, dY, nIn, nOut, tps, S, slab, st);
    else {
        dim3 grid(t.K, S, 1);
        conv_dw_generic_kernel<<<grid, 256, 0, st>>>(X, dY, t.tile_chunk_start, t.seg_start, t.chunk_src, t.chunk_dst,
                                                     t.n_tiles, nIn, nOut, tps, S, slab);
        rc = 0;
        if (hipGetLastError() != hipSuccess) rc = 1;
    }
    if (rc) {
        pool_free(slab);
        if (!mfma_ok) SEI_FAIL("conv dW generic launch failed");
        return rc;
    }
    int64_t tot = (int64_t)t.K * per;
    slab_reduce_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(slab, dW, t.K, S, per);
    SEI_LAUNCH_CHECK();
    pool_free(slab);
    return 0;
}

}  // namespace sei
