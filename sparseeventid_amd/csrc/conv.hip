// Sparse convolution for gfx950, weight side (scn.SubmanifoldConvolution / scn.Convolution:
// resnet.py:30-50,105-110; sparse_building_blocks.py:29-34,110-117):
//   * weight re-layout for the forward / dX kernels (conv_fwd.hip: column-permuted panels; conv_wave.hip: natural
//     layout, transposed + offset-flipped for dX);
//   * dW: per (k, split, ci block) workgroup reduces X[src]^T dY[dst] over its sub-chunks with v_mfma_f32_16x16x4_f32
//     (rules are the contraction dimension, 4 per MFMA); register-resident operands, no barrier in the main loop;
//     partial slabs are summed in a fixed order (deterministic).
#include "common.h"

#include <limits.h>
#include <stdlib.h>

namespace sei {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

#ifdef SEI_CLOCK
__device__ unsigned long long g_clk[8][2];   // family 2: conv_dw_mfma_kernel (sei_debug_clock_dw)
#endif

// position of column c inside a column group of NT = width/16 tiles (see load_b in conv_fwd.hip)
__device__ inline int permute_col(int c, int width) {
    if (width == 0) return c;
    const int grp = c / width, cl = c - grp * width;
    const int nt = width / 16, b4 = nt / 4, rem = nt % 4;
    int pos;
    if (cl < 64 * b4) {
        const int b = cl / 64, w = cl % 64;
        pos = 64 * b + 4 * (w % 16) + w / 16;
    } else {
        const int c2 = cl - 64 * b4, j = c2 / 16, r = c2 % 16;
        if (rem == 1 || (rem == 3 && j == 2))
            pos = 64 * b4 + (rem == 3 ? 32 : 0) + r;
        else
            pos = 64 * b4 + 2 * r + j;
    }
    return grp * width + pos;
}

// destination index of element (row, col) of offset ks in the kernel's weight layout: [K][rows][cols] with the
// columns of every group permuted, or (blocks > 1) [block][K][rows][cols / blocks] so that each column block is
// a contiguous panel for the workgroups that own it
__device__ inline size_t weight_index(int ks, int row, int col, int K, int rows, int cols, int col_group, int blocks) {
    if (col_group == -2) {   // [r5] wave kernel at 32 x 32 planes: [k][16-plane slice j][half h][lane = 16 g + r][(e & 1) * 2 + n]:
        // lane (g, r) finds planes 16j+4g+e (e = 2h, 2h+1) of its two columns 2r + n (the wave kernel's natural-layout column
        // assignment: wload_b<2> reads columns 2r, 2r+1) in ONE 16-byte word, a slice in two
        const int j = row >> 4, g = (row >> 2) & 3, e = row & 3, n = col & 1, r = col >> 1;
        return ((((size_t)ks * (rows >> 4) + j) * 2 + (e >> 1)) * 64 + (16 * g + r)) * 4 + (e & 1) * 2 + n;
    }
    if (col_group < 0) {  // column-owner kernel: [k][column tile][16-row group][lane = (row % 16) / 4 * 16 + col % 16][row % 4]
        const int cb = col >> 4, j = row >> 4, ln = ((row & 15) >> 2) * 16 + (col & 15);
        return ((((size_t)ks * (cols >> 4) + cb) * (rows >> 4) + j) * 64 + ln) * 4 + (row & 3);
    }
    const int pc = permute_col(col, col_group);
    if (blocks <= 1) return ((size_t)ks * rows + row) * cols + pc;
    const int w = cols / blocks, b = pc / w;
    return (((size_t)b * K + ks) * rows + row) * w + (pc - b * w);
}

__global__ void transform_weights_kernel(const float *__restrict__ W, float *__restrict__ Wout, int K, int nIn, int nOut,
                                         int transpose, int flip, int col_group, int blocks) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t per = (int64_t)nIn * nOut;
    if (t >= K * per) return;
    int k = (int)(t / per);
    int rem = (int)(t - k * per);
    int ci = rem / nOut, co = rem - ci * nOut;  // coalesced read of W
    int ks = flip ? K - 1 - k : k;              // destination offset index
    float v = W[t];
    if (transpose)
        Wout[weight_index(ks, co, ci, K, nOut, nIn, col_group, blocks)] = v;
    else
        Wout[weight_index(ks, ci, co, K, nIn, nOut, col_group, blocks)] = v;
}

int conv_transform_weights(const float *W, float *Wout, int K, int nIn, int nOut, bool transpose, bool flip, int col_group,
                           int blocks, hipStream_t st) {
    int64_t tot = (int64_t)K * nIn * nOut;
    transform_weights_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(W, Wout, K, nIn, nOut, transpose ? 1 : 0,
                                                                           flip ? 1 : 0, col_group, blocks);
    SEI_LAUNCH_CHECK();
    return 0;
}

// three-way bf16 split of the weights in the B layout of v_mfma_f32_16x16x32_bf16 (conv_fwd.hip: compute_phase_split)
__global__ void weights_split_kernel(const float *__restrict__ Wnat, uint32_t *__restrict__ Ws, int K, int rows, int cols, int nt) {
    // one thread per (block, k, s, n, lane, pair jp): two consecutive input planes of one column -> one word of every part
    const int NS = rows / 32, blocks = cols / (16 * nt);
    const int64_t total = (int64_t)blocks * K * NS * nt * 64 * 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int jp = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int64_t r = t >> 8;
    const int n = (int)(r % nt); r /= nt;
    const int sl = (int)(r % NS); r /= NS;
    const int k = (int)(r % K);
    const int blk = (int)(r / K);
    const int row = 32 * sl + 8 * (lane >> 4) + 2 * jp, col = 16 * (blk * nt + n) + (lane & 15);
    const float x = Wnat[((size_t)k * rows + row) * cols + col], y = Wnat[((size_t)k * rows + row + 1) * cols + col];
    // round-to-nearest-even bf16 of what is left, three times (the same arithmetic as split_pair in conv_fwd.hip)
    auto rne = [](float v) { const unsigned u = __float_as_uint(v); return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; };
    const unsigned x1 = rne(x), y1 = rne(y);
    const float rx = x - __uint_as_float(x1), ry = y - __uint_as_float(y1);
    const unsigned x2 = rne(rx), y2 = rne(ry);
    const float rx2 = rx - __uint_as_float(x2), ry2 = ry - __uint_as_float(y2);
    const unsigned x3 = rne(rx2), y3 = rne(ry2);
    const unsigned part[3] = {y1 | (x1 >> 16), y2 | (x2 >> 16), y3 | (x3 >> 16)};
    // [block][k][s][part][n][lane][4 words]
    const size_t slice = ((size_t)(blk * K + k) * NS + sl) * (size_t)(3 * nt * 256);
#pragma unroll
    for (int q = 0; q < 3; q++) Ws[slice + ((size_t)(q * nt + n) * 64 + lane) * 4 + jp] = part[q];
}

int conv_split_weights(const float *Wnat, float *Ws, int K, int rows, int cols, int nt, hipStream_t st) {
    if (rows % 32 || cols % (16 * nt)) SEI_FAIL("conv_split_weights: plane counts do not match the split layout");
    const int64_t total = (int64_t)(cols / (16 * nt)) * K * (rows / 32) * nt * 64 * 4;
    weights_split_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(Wnat, reinterpret_cast<uint32_t *>(Ws), K, rows, cols, nt);
    SEI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ dW
constexpr int DW_TILES_PER_ITER = 128;             // tiles scanned per list refill
constexpr int DW_LIST_CAP = DW_TILES_PER_ITER * 16;  // <= 16 sub-chunks per (tile, k)

// builds in LDS the list of (sub-chunk, tile) pairs of offset k for tiles [tb, tb+128) in tile order
template <int TPI = DW_TILES_PER_ITER>
__device__ inline int dw_build_list(const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ seg_start,
                                    int n_tiles, int k, int tb, int t_end, int *list_sub, int *list_tile, int *scan_tmp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = tb + threadIdx.x;
    int cnt = 0, base = 0;
    if (threadIdx.x < TPI && t < t_end) {
        int a = seg_start[(size_t)k * n_tiles + t], b = seg_start[(size_t)(k + 1) * n_tiles + t];
        cnt = b - a;
        base = tile_sub_start[t] + a;
    }
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
    for (int q = 0; q < cnt; q++) {
        list_sub[pos + q] = base + q;
        list_tile[pos + q] = t;
    }
    __syncthreads();
    return total;
}

// rule indices of one sub-chunk, one slot per lane r16 (replicated over the 4 lane groups)
struct DwItem {
    int src;       // src row of slot r16, -1 = padding / no item
    int drow;      // global dst row of slot r16 (rows are int32 everywhere in the rulebook)
};

__device__ __forceinline__ DwItem dw_load_item(const int *list_sub, const int *list_tile, int li, int total, int TR,
                                               const int32_t *__restrict__ sub_src,
                                               const uint8_t *__restrict__ sub_dst, int r16) {
    // clamped + select: never branch around a load (hipcc would serialise it with vmcnt(0))
    const bool valid = li < total;
    const int lc = valid ? li : 0;
    const int64_t c = valid ? list_sub[lc] : 0;
    const int rowbase = valid ? list_tile[lc] * TR : 0;
    const int sv = sub_src[c * SUB_SLOTS + r16];
    const int dv = sub_dst[c * SUB_SLOTS + r16];
    DwItem it;
    it.src = valid ? sv : -1;
    it.drow = rowbase + dv;
    return it;
}

template <int MT, int NT>
struct DwOps {
    float a[MT], b[NT];
    int any;  // wave-uniform: some slot of this step holds a rule
};

// Operand values of T 16-channel tiles for one rule row, with as few (wide) loads as possible: the tiles of a
// group of 4 (3, 2) interleave their channels -- tile j of the group, row/column r  <->  channel 4r+j (3r+j, 2r+j) -- so
// lane r16 finds the values of 4 (3, 2) tiles in ONE global_load_dwordx4 (dwordx3, dwordx2).  dw_chan is the inverse map.
template <int T>
__device__ __forceinline__ void dw_load_vec(const float *__restrict__ p, int r16, bool ok, float (&o)[T]) {
    constexpr int B4 = T / 4, REM = T % 4;
#pragma unroll
    for (int q = 0; q < B4; q++) {
        const float4 v = *reinterpret_cast<const float4 *>(p + 64 * q + 4 * r16);
        o[4 * q + 0] = ok ? v.x : 0.f; o[4 * q + 1] = ok ? v.y : 0.f;
        o[4 * q + 2] = ok ? v.z : 0.f; o[4 * q + 3] = ok ? v.w : 0.f;
    }
    if (REM == 3) {   // three tiles in ONE global_load_dwordx3: tile j, row/column r <-> channel 3r+j
        struct F3 { float x, y, z; };
        const F3 v = *reinterpret_cast<const F3 *>(p + 64 * B4 + 3 * r16);
        o[4 * B4 + 0] = ok ? v.x : 0.f; o[4 * B4 + 1] = ok ? v.y : 0.f; o[4 * B4 + 2] = ok ? v.z : 0.f;
    }
    if (REM == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(p + 64 * B4 + 2 * r16);
        o[4 * B4 + 0] = ok ? v.x : 0.f; o[4 * B4 + 1] = ok ? v.y : 0.f;
    }
    if (REM == 1) { const float v = p[64 * B4 + r16]; o[4 * B4] = ok ? v : 0.f; }
}

template <int T>
__device__ __forceinline__ int dw_chan(int t, int r) {  // channel (relative to the block) of row / column r of tile t
    constexpr int B4 = T / 4, REM = T % 4;
    if (t < 4 * B4) return 64 * (t / 4) + 4 * r + (t % 4);
    const int t2 = t - 4 * B4;
    if (REM == 3) return 64 * B4 + 3 * r + t2;
    if (REM == 2) return 64 * B4 + 2 * r + t2;
    return 64 * B4 + r;
}

// operands of MFMA step q (rules 4q..4q+3 of the sub-chunk): lane (g, r16) holds X[src_g][ci0 + dw_chan<MT>(m, r16)],
// dY[dst_g][dw_chan<NT>(n, r16)]
template <int MT, int NT>
__device__ __forceinline__ DwOps<MT, NT> dw_load_ops(const DwItem &it, int q, int g, int r16,
                                                     const float *__restrict__ X, const float *__restrict__ dY, int CIN,
                                                     int COUT, int ci0) {
    DwOps<MT, NT> o;
    const int src = __shfl(it.src, 4 * q + g);
    const int drow = __shfl(it.drow, 4 * q + g);
    const bool ok = src >= 0;
    // row address = base + u32 row * u32 stride: ONE v_mad_u64_u32 per operand (signed 64-bit index math costs four)
    const uint32_t xr = ok ? (uint32_t)src : 0u, yr = ok ? (uint32_t)drow : 0u;
#if defined(SEI_DW_ABL_NOGATHER)  // diagnostic: no operand gathers (results are wrong, timing only)
#pragma unroll
    for (int m = 0; m < MT; m++) o.a[m] = ok ? 1.f + xr : 0.f;
#pragma unroll
    for (int n = 0; n < NT; n++) o.b[n] = ok ? 2.f + yr : 0.f;
#elif defined(SEI_DW_ABL_NOB)      // diagnostic: only the X gathers remain (1 of the 3 vector-memory instructions per step)
    dw_load_vec<MT>(X + ci0 + (uint64_t)xr * (uint32_t)CIN, r16, ok, o.a);
#pragma unroll
    for (int n = 0; n < NT; n++) o.b[n] = ok ? 2.f + yr : 0.f;
#elif defined(SEI_DW_ABL_NOA)      // diagnostic: only the dY gathers remain (2 of the 3)
#pragma unroll
    for (int m = 0; m < MT; m++) o.a[m] = ok ? 1.f + xr : 0.f;
    dw_load_vec<NT>(dY + (uint64_t)yr * (uint32_t)COUT, r16, ok, o.b);
#elif defined(SEI_DW_ABL_CACHED)   // diagnostic: every gather hits the caches
    dw_load_vec<MT>(X + ci0 + (uint64_t)(xr & 1023u) * (uint32_t)CIN, r16, ok, o.a);
    dw_load_vec<NT>(dY + (uint64_t)(yr & 1023u) * (uint32_t)COUT, r16, ok, o.b);
#else
    dw_load_vec<MT>(X + ci0 + (uint64_t)xr * (uint32_t)CIN, r16, ok, o.a);
    dw_load_vec<NT>(dY + (uint64_t)yr * (uint32_t)COUT, r16, ok, o.b);
#endif
    o.any = __any(ok);
    return o;
}

// Waves per SIMD the register allocator is held to (1: its own choice).  Left alone hipcc splits these kernels' registers
// into VGPRs + AGPRs with a total just past an occupancy step: 160 planes (2 x 10 tiles) 180 registers = 2 waves per SIMD,
// held to 3 it fits 163 without scratch and runs 1.365 vs 1.454 ms (level 4, 64 events); 64 planes (4 x 4) 148 = 3 waves,
// held to 4: 127 registers, 1.94 vs 1.99 ms; 192 planes (2 x 12) 252 = 2 waves, held to 3: 160 registers, 0.655 vs 0.726 ms.
// Not every step pays: 128 planes (2 x 8) at 4 waves instead of 3 measured 3 % SLOWER, 96 planes (2 x 6) at 5 needs scratch
// and loses 5 % (profiles/r04_conv_dw_occupancy.log).
constexpr int dw_min_waves(int mt, int nt) {
    if (mt == 2 && nt == 10) return 3;
    if (mt == 2 && nt == 12) return 3;
    if (mt == 4 && nt == 4) return 4;
#if defined(SEI_DW_OCC)   // experiment builds: the steps that did not pay
    if (mt == 2 && nt == 8) return 4;
#if SEI_DW_OCC >= 2
    if (mt == 2 && nt == 6) return 5;
#endif
#endif
    return 1;
}

template <int MT, int NT>
__global__ __launch_bounds__(256, dw_min_waves(MT, NT)) void conv_dw_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ dY, const int32_t *__restrict__ tile_sub_start,
    const int32_t *__restrict__ seg_start, const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst,
    int n_tiles, int TR, int CIN, int COUT, int tiles_per_split, int S, int K, float *__restrict__ slab) {
    extern __shared__ float lds_f[];
    __shared__ int list_sub[DW_LIST_CAP];
    __shared__ int list_tile[DW_LIST_CAP];
    __shared__ int scan_tmp[8];
    // linear id -> (xcd, ci block, k, split group): the K * (CIN / (16 MT)) workgroups that walk the same tile
    // range (and re-read the same dY / X rows) get consecutive ids that differ by 8, i.e. run on ONE XCD at about
    // the same time and share its L2 (speed only)
    const int CB = CIN / (16 * MT);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int cig = j % CB, k = (j / CB) % K, s = (j / (CB * K)) * 8 + xcd;
    if (s >= S) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int ci0 = cig * MT * 16;
    CLK_BEGIN;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int t_begin = s * tiles_per_split;
    const int t_end = min(n_tiles, t_begin + tiles_per_split);
    for (int tb = t_begin; tb < t_end; tb += DW_TILES_PER_ITER) {
        const int total = dw_build_list(tile_sub_start, seg_start, n_tiles, k, tb, t_end, list_sub, list_tile, scan_tmp);
        if (total > 0) {
            // software pipeline: item indices one item ahead, MFMA operands one step ahead
            int li = wave;
            DwItem cur = dw_load_item(list_sub, list_tile, li, total, TR, sub_src, sub_dst, r16);
            DwItem nxt = dw_load_item(list_sub, list_tile, li + 4, total, TR, sub_src, sub_dst, r16);
            DwOps<MT, NT> oc = dw_load_ops<MT, NT>(cur, 0, g, r16, X, dY, CIN, COUT, ci0);
            while (li < total) {
#pragma unroll
                for (int q = 0; q < SUB_SLOTS / 4; q++) {
                    DwOps<MT, NT> on = q + 1 < SUB_SLOTS / 4 ? dw_load_ops<MT, NT>(cur, q + 1, g, r16, X, dY, CIN, COUT, ci0)
                                                             : dw_load_ops<MT, NT>(nxt, 0, g, r16, X, dY, CIN, COUT, ci0);
                    if (oc.any) {
#pragma unroll
                        for (int m = 0; m < MT; m++)
#pragma unroll
                            for (int n = 0; n < NT; n++) {
#ifdef SEI_DW_ABL_NOMFMA  // diagnostic: one VALU fma instead of the MFMA (keeps the operand dependencies)
                                acc[m][n][0] = fmaf(oc.a[m], oc.b[n], acc[m][n][0]);
#else
                                acc[m][n] = mfma16(oc.a[m], oc.b[n], acc[m][n]);
#endif
                            }
                    }
                    oc = on;
                }
                li += 4;
                cur = nxt;
                nxt = dw_load_item(list_sub, list_tile, li + 4, total, TR, sub_src, sub_dst, r16);
            }
        }
        __syncthreads();
    }
    // cross-wave reduction in a fixed order (deterministic): red[ci][co]
    float *red = lds_f;
#ifdef SEI_DW_ABL_NOREDUCE  // diagnostic: only wave 0's accumulators are written (results wrong, timing only)
    for (int w = 0; w < 1; w++) {
#else
    for (int w = 0; w < 4; w++) {
#endif
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int n = 0; n < NT; n++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        int idx = dw_chan<MT>(m, 4 * g + i) * COUT + dw_chan<NT>(n, r16);
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
    CLK_END(2);
}

#ifdef SEI_EXPERIMENTS
// ------------------------------------------------------------------------------------------ dW, operands through LDS
// The register-operand kernel above spends its waves on ISSUING gathers: three vector-memory instructions per 12 MFMAs
// (profiles/r02_conv_dw_ablation.log: 111 TFLOP/s without them, 82-89 with), and every rule's dY row is fetched by each of
// the CIN/32 workgroups that share a (k, split).  Here ONE workgroup computes the whole CIN x COUT block of a (k, split):
//   * two LOADER waves gather the 16 X[src] rows and the 16 dY[dst] rows of a sub-chunk -- WHOLE rows, once -- into a ring
//     slot in LDS with per-lane-address LDS-DMA (global_load_lds_dwordx4: no staging registers, 384-768 B row segments);
//   * CIN/32 COMPUTE waves (wave w owns input planes 32 w .. 32 w + 31: 2 x NT accumulator tiles) read their operands with
//     ds_read_b64 / b128 and issue MFMAs; their instruction stream holds no vector-memory instruction at all.
// Hand-off as in conv_band.hip: "sub-chunks landed" words written by the loaders, a "sub-chunks released" word per
// compute wave, bounded polls, no barrier inside a list (one per 128-tile list refill).  Every compute wave does the same
// work per sub-chunk, so nobody waits for a slower sibling.  Accumulation order: sub-chunks in list (tile) order per
// (k, split), splits summed in order by slab_reduce_kernel -- fixed, run-to-run deterministic.
constexpr int DWR_SYNC_WORDS = 32;   // [0] X sub-chunks landed, [1] dY sub-chunks landed, [8 + w] released by compute wave w
constexpr int DWR_LAG = 2;           // sub-chunks a loader keeps in flight before it publishes the older one
constexpr int DWR_SPIN_CAP = 1 << 22;
// diagnostic builds only (EXTRA=-DSEI_DWR_ABL=n): 1 nobody waits for the ring, 2 no MFMAs, 4 loaders issue no DMA; timing only
#ifndef SEI_DWR_ABL
#define SEI_DWR_ABL 0
#endif
constexpr int DWR_TILES_PER_ITER = 32;                   // tiles per list refill: a short list leaves LDS to the ring
constexpr int DWR_LIST_CAP = DWR_TILES_PER_ITER * 16;
__device__ int g_dw_timeouts;

__device__ __forceinline__ int dwr_poll_ge(volatile int *word, int want, int lane) {
    int v, spins = 0;
    for (;;) {
        v = __builtin_amdgcn_readfirstlane(*word);
        if (v >= want) return v;
        if (++spins > DWR_SPIN_CAP) {
            if (lane == 0) atomicAdd(&g_dw_timeouts, 1);
            return INT_MAX;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// one loader wave: rows `row_of(li, slot r16)` of table T (row length CW floats) -> slot part `part_off` of the ring
template <int CT16>  // CW / 16
__device__ __forceinline__ void dwr_loader(const float *__restrict__ T, float *ring, int slot_floats, int part_off, int D,
                                           volatile int *landed, volatile int *sync, volatile int *hdr, int ncw, int total,
                                           const int *list_sub, const int *list_tile, int TR, bool is_x,
                                           const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int lane) {
    constexpr int CW = 16 * CT16, CPR = CW / 4, P = CT16;  // chunks of 16 B per row, DMA instructions per sub-chunk
    const int r16 = lane & 15;
    int row_p[P], ch_p[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int id = 64 * p + lane;
        row_p[p] = id / CPR;
        ch_p[p] = id - row_p[p] * CPR;
    }
    auto load_rows = [&](int li) {  // lane r16: the table row of rule slot r16 (-1: padding); clamped index, no branch
        const int lc = li < total ? li : total - 1;
        const int64_t c = list_sub[lc];
        const int sv = sub_src[c * SUB_SLOTS + r16];
        if (is_x) return sv;
        const int dv = sub_dst[c * SUB_SLOTS + r16];
        return sv >= 0 ? list_tile[lc] * TR + dv : -1;
    };
    int known_free = D;
    int rows = load_rows(0);
    int slot = 0;
    for (int li = 0; li < total; li++) {
        if (li >= known_free && !(SEI_DWR_ABL & 1)) {
            // nothing more can be issued: publish what has been issued as soon as it has landed, then wait for a slot
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) *landed = li;
            asm volatile("" ::: "memory");
            int spins = 0;
            for (;;) {
                int v = lane < ncw ? sync[8 + lane] : INT_MAX;
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) v = min(v, __shfl_xor(v, d));
                const int mn = __builtin_amdgcn_readfirstlane(v);
                known_free = mn > INT_MAX - D ? INT_MAX : mn + D;
                if (li < known_free) break;
                if (++spins > DWR_SPIN_CAP) {
                    if (lane == 0) atomicAdd(&g_dw_timeouts, 1);
                    known_free = INT_MAX;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        const int cur = rows;
        rows = load_rows(li + 1);  // BEFORE this sub-chunk's DMAs: vmcnt retires in order, the index load must not sit behind them
        float *dst = ring + (size_t)slot * slot_floats + part_off;
        const unsigned long long valid = __ballot(cur >= 0);
        if (is_x && lane == 0) hdr[slot] = __popcll(valid & 0xFFFFull);
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int r = __shfl(cur, row_p[p]);
            const float *src = T + (uint64_t)(uint32_t)(r > 0 ? r : 0) * (uint32_t)CW + 4 * ch_p[p];
            if (SEI_DWR_ABL & 4) { asm volatile("" ::"v"(src)); continue; }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + 256 * p), 16, 0, 0);
        }
        if (li >= DWR_LAG) {
            // everything older than the last LAG sub-chunks (their DMAs + one index load each) has landed
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DWR_LAG * (P + 1)) : "memory");
            if (lane == 0) *landed = li - DWR_LAG + 1;
            asm volatile("" ::: "memory");
        }
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) *landed = total;
    asm volatile("" ::: "memory");
}

#ifndef SEI_DWR_MT
#define SEI_DWR_MT 1
#endif
constexpr int DWR_MT = SEI_DWR_MT;   // 16-row accumulator tiles per compute wave (1: CIN/16 compute waves, 2: CIN/32)

// waves per SIMD the register allocator must leave room for: 2 x NT accumulator tiles + two operand sets + the loaders' tables
constexpr int dwr_occ(int nt) { return SEI_DWR_MT == 1 ? (nt <= 8 ? 4 : 3) : (nt <= 6 ? 4 : (nt <= 8 ? 3 : 2)); }

template <int CT16, int NT>
__global__ __launch_bounds__(64 * (CT16 / DWR_MT + 2), dwr_occ(NT)) void conv_dw_ring_kernel(
    const float *__restrict__ X, const float *__restrict__ dY, const int32_t *__restrict__ tile_sub_start,
    const int32_t *__restrict__ seg_start, const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst,
    int n_tiles, int TR, int tiles_per_split, int S, int K, int D, float *__restrict__ slab) {
    extern __shared__ float lds_f[];
    __shared__ int list_sub[DWR_LIST_CAP];
    __shared__ int list_tile[DWR_LIST_CAP];
    __shared__ int scan_tmp[8];
    __shared__ int sync_w[DWR_SYNC_WORDS];
    __shared__ int hdr_w[16];
    constexpr int CIN = 16 * CT16, COUT = 16 * NT, MT = DWR_MT, NCW = CT16 / MT;
    constexpr int SLOT = SUB_SLOTS * (CIN + COUT);  // floats per ring slot: 16 X rows, then 16 dY rows
    volatile int *sync = sync_w;
    volatile int *hdr = hdr_w;
    float *ring = lds_f;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int k = j % K, s = (j / K) * 8 + xcd;
    if (s >= S) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int ci0 = wave * MT * 16;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int t_begin = s * tiles_per_split;
    const int t_end = min(n_tiles, t_begin + tiles_per_split);
    for (int tb = t_begin; tb < t_end; tb += DWR_TILES_PER_ITER) {
        if (threadIdx.x < DWR_SYNC_WORDS) sync[threadIdx.x] = 0;  // the ring is empty here; made visible by the list's barriers
        if (threadIdx.x < 8 && threadIdx.x >= (blockDim.x >> 6)) scan_tmp[threadIdx.x] = 0;  // dw_build_list sums four waves' entries
        const int total = dw_build_list<DWR_TILES_PER_ITER>(tile_sub_start, seg_start, n_tiles, k, tb, t_end, list_sub, list_tile, scan_tmp);
        if (total > 0) {
            if (wave == NCW) {
                dwr_loader<CT16>(X, ring, SLOT, 0, D, &sync[0], sync, hdr, NCW, total, list_sub, list_tile, TR, true, sub_src,
                                 sub_dst, lane);
            } else if (wave == NCW + 1) {
                dwr_loader<NT>(dY, ring, SLOT, SUB_SLOTS * CIN, D, &sync[1], sync, hdr, NCW, total, list_sub, list_tile, TR, false,
                               sub_src, sub_dst, lane);
            } else {
                int known = 0, slot = 0;
                for (int li = 0; li < total; li++) {
                    if (li >= known && !(SEI_DWR_ABL & 1)) {
                        const int a = dwr_poll_ge(&sync[0], li + 1, lane), b = dwr_poll_ge(&sync[1], li + 1, lane);
                        known = a < b ? a : b;
                    }
                    const float *sx = ring + (size_t)slot * SLOT, *sy = sx + SUB_SLOTS * CIN;
                    const int nv = __builtin_amdgcn_readfirstlane(hdr[slot]);
                    const int steps = (nv + 3) >> 2;  // steps that hold only padding slots are skipped (wave-uniform)
                    // operands one step ahead of the MFMAs that consume them.  The padding mask is applied AFTER the MFMAs
                    // of the step in between (a select right at the load would wait for the LDS read there)
                    float a_c[MT], b_c[NT];
                    dw_load_vec<MT>(sx + g * CIN + ci0, r16, g < nv, a_c);
                    dw_load_vec<NT>(sy + g * COUT, r16, g < nv, b_c);
#pragma unroll 1
                    for (int q = 0; q < steps; q++) {
                        float a_n[MT], b_n[NT];
                        const int rs = 4 * (q + 1) + g, rl = rs < SUB_SLOTS ? rs : SUB_SLOTS - 1;  // clamped: stays inside the slot
                        dw_load_vec<MT>(sx + rl * CIN + ci0, r16, true, a_n);
                        dw_load_vec<NT>(sy + rl * COUT, r16, true, b_n);
                        __builtin_amdgcn_sched_barrier(0);  // the next step's LDS reads are issued before this step's MFMAs
#pragma unroll
                        for (int m = 0; m < MT; m++)
#pragma unroll
                            for (int n = 0; n < NT; n++) {
                                if (SEI_DWR_ABL & 2) asm volatile("" ::"v"(a_c[m]), "v"(b_c[n]));
                                else acc[m][n] = mfma16(a_c[m], b_c[n], acc[m][n]);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                        const bool okn = rs < nv;
#pragma unroll
                        for (int m = 0; m < MT; m++) a_c[m] = okn ? a_n[m] : 0.f;
#pragma unroll
                        for (int n = 0; n < NT; n++) b_c[n] = okn ? b_n[n] : 0.f;
                    }
                    // every LDS read of the slot has been issued before this store; a wave's LDS operations execute in order
                    asm volatile("" ::: "memory");
                    if (lane == 0) sync[8 + wave] = li + 1;
                    asm volatile("" ::: "memory");
                    slot = slot + 1 == D ? 0 : slot + 1;
                }
            }
        }
        __syncthreads();
    }
    // block rows ci0 .. ci0 + 31 of this wave -> slab, through the (now idle) ring for coalesced stores: D regions of
    // 32 x COUT floats fit a ring slot pair only if CIN >= COUT; waves take turns in rounds of R regions
    constexpr int REGION = MT * 16 * COUT;
    const int regions = (D * SLOT) / REGION > 0 ? (D * SLOT) / REGION : 1;
    float *out = slab + (size_t)(k * S + s) * CIN * COUT;
    for (int w0 = 0; w0 < NCW; w0 += regions) {
        if (wave >= w0 && wave < w0 + regions && wave < NCW) {
            float *red = ring + (size_t)(wave - w0) * REGION;
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int n = 0; n < NT; n++)
#pragma unroll
                    for (int i = 0; i < 4; i++) red[dw_chan<MT>(m, 4 * g + i) * COUT + dw_chan<NT>(n, r16)] = acc[m][n][i];
        }
        __syncthreads();
        const int nreg = NCW - w0 < regions ? NCW - w0 : regions;
        for (int idx = threadIdx.x; idx < nreg * REGION; idx += blockDim.x) out[(size_t)w0 * REGION + idx] = ring[idx];
        __syncthreads();
    }
}

#endif  // SEI_EXPERIMENTS

// dW for tiny nIn (stem: nIn = 1): HBM-bound gather of dY rows; lane = output plane, the 64/COUT lane groups
// take different rule slots; all loads of a sub-chunk are issued together; fixed-order reductions.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_dw_smallcin_kernel(
    const float *__restrict__ X, const float *__restrict__ dY, const int32_t *__restrict__ tile_sub_start,
    const int32_t *__restrict__ seg_start, const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst,
    int n_tiles, int TR, int tiles_per_split, int S, float *__restrict__ slab) {
    __shared__ int list_sub[DW_LIST_CAP];
    __shared__ int list_tile[DW_LIST_CAP];
    __shared__ int scan_tmp[8];
    __shared__ float red[4][CIN][64];
    const int k = blockIdx.x, s = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15;
    constexpr int groups = 64 / COUT;  // rule slots handled per step by one wave (COUT in {16, 32, 64})
    constexpr int STEPS = SUB_SLOTS / groups;
    const int co = lane % COUT, grp = lane / COUT;
    float acc[CIN];
#pragma unroll
    for (int c = 0; c < CIN; c++) acc[c] = 0.f;
    const int t_begin = s * tiles_per_split;
    const int t_end = min(n_tiles, t_begin + tiles_per_split);
    for (int tb = t_begin; tb < t_end; tb += DW_TILES_PER_ITER) {
        const int total = dw_build_list(tile_sub_start, seg_start, n_tiles, k, tb, t_end, list_sub, list_tile, scan_tmp);
        if (total > 0) {
            int li = wave;
            DwItem cur = dw_load_item(list_sub, list_tile, li, total, TR, sub_src, sub_dst, r16);
            while (li < total) {
                DwItem nxt = dw_load_item(list_sub, list_tile, li + 4, total, TR, sub_src, sub_dst, r16);
                float dyv[STEPS], xv[STEPS][CIN];
#pragma unroll
                for (int q = 0; q < STEPS; q++) {
                    const int slot = q * groups + grp;
                    const int src = __shfl(cur.src, slot);
                    const int64_t drow = __shfl(cur.drow, slot);
                    const bool ok = src >= 0;
                    float dv = dY[(size_t)(ok ? drow : 0) * COUT + co];
                    dyv[q] = ok ? dv : 0.f;
#pragma unroll
                    for (int ci = 0; ci < CIN; ci++) xv[q][ci] = X[(size_t)(ok ? src : 0) * CIN + ci];
                }
#pragma unroll
                for (int q = 0; q < STEPS; q++)
#pragma unroll
                    for (int ci = 0; ci < CIN; ci++) acc[ci] = fmaf(xv[q][ci], dyv[q], acc[ci]);
                li += 4;
                cur = nxt;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ci++) red[wave][ci][lane] = acc[ci];
    __syncthreads();
    float *out = slab + (size_t)(k * S + s) * CIN * COUT;
    for (int e = threadIdx.x; e < CIN * COUT; e += 256) {
        int ci = e / COUT, c2 = e - ci * COUT;
        float sum = 0.f;
        for (int w = 0; w < 4; w++)
            for (int gq = 0; gq < groups; gq++) sum += red[w][ci][gq * COUT + c2];
        out[e] = sum;
    }
}

// generic dW (any channel counts): one workgroup per (k, split); thread per (ci, co) element loop.
__global__ __launch_bounds__(256) void conv_dw_generic_kernel(
    const float *__restrict__ X, const float *__restrict__ dY, const int32_t *__restrict__ tile_sub_start,
    const int32_t *__restrict__ seg_start, const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst,
    int n_tiles, int TR, int CIN, int COUT, int tiles_per_split, int S, float *__restrict__ slab) {
    const int k = blockIdx.x, s = blockIdx.y;
    const int t_begin = s * tiles_per_split;
    const int t_end = min(n_tiles, t_begin + tiles_per_split);
    float *out = slab + (size_t)(k * S + s) * CIN * COUT;
    for (int e = threadIdx.x; e < CIN * COUT; e += 256) {
        int ci = e / COUT, co = e - ci * COUT;
        float sum = 0.f;
        for (int t = t_begin; t < t_end; t++) {
            int64_t ca = tile_sub_start[t] + seg_start[(size_t)k * n_tiles + t];
            int64_t cb = tile_sub_start[t] + seg_start[(size_t)(k + 1) * n_tiles + t];
            for (int64_t sl = ca * SUB_SLOTS; sl < cb * SUB_SLOTS; sl++) {
                int src = sub_src[sl];
                if (src >= 0)
                    sum = fmaf(X[(size_t)src * CIN + ci], dY[((size_t)t * TR + sub_dst[sl]) * COUT + co], sum);
            }
        }
        out[e] = sum;
    }
}

// dW of a convolution with a FEW output planes (COUT <= 8: the 224 -> 2 / 3 class maps that end the SBND heads,
// torch/sparseresnet.py:300-334): thread = input plane, the rule's COUT partial products in registers.  X rows are read
// coalesced across the input planes, a rule's dY values are broadcast loads; slots are walked four at a time with their
// loads issued together and their FMAs kept in slot order (fixed summation order).  One workgroup per (k, split) with
// SMALL splits: the generic kernel below gave each of 672 (ci, co) threads a serial walk over a whole split of
// uncoalesced loads -- 3.6 ms per launch, 14 ms of the SBND step for four 1 x 1 x 1 layers.
template <int CO>
__global__ __launch_bounds__(256) void conv_dw_narrow_kernel(
    const float *__restrict__ X, const float *__restrict__ dY, const int32_t *__restrict__ tile_sub_start,
    const int32_t *__restrict__ seg_start, const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst,
    int n_tiles, int TR, int CIN, int COUT, int tiles_per_split, int S, float *__restrict__ slab) {
    const int k = blockIdx.x, s = blockIdx.y;
    const int t_begin = s * tiles_per_split;
    const int t_end = min(n_tiles, t_begin + tiles_per_split);
    float *out = slab + (size_t)(k * S + s) * CIN * COUT;
    for (int ci = threadIdx.x; ci < CIN; ci += 256) {
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; c++) acc[c] = 0.f;
        for (int t = t_begin; t < t_end; t++) {
            const int64_t ca = (tile_sub_start[t] + (int64_t)seg_start[(size_t)k * n_tiles + t]) * SUB_SLOTS;
            const int64_t cb = (tile_sub_start[t] + (int64_t)seg_start[(size_t)(k + 1) * n_tiles + t]) * SUB_SLOTS;
            for (int64_t sl = ca; sl < cb; sl += 4) {   // sub-chunks hold 16 slots: groups of four never straddle one
                int src[4];
                float x[4], d[4][CO];
#pragma unroll
                for (int j = 0; j < 4; j++) src[j] = sub_src[sl + j];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int sv = src[j] > 0 ? src[j] : 0;   // clamped, never a branch around a load
                    x[j] = X[(size_t)sv * CIN + ci];
                    const float *dr = dY + ((size_t)t * TR + sub_dst[sl + j]) * COUT;
#pragma unroll
                    for (int c = 0; c < CO; c++) d[j][c] = c < COUT ? dr[c] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (src[j] >= 0) {
#pragma unroll
                        for (int c = 0; c < CO; c++) acc[c] = fmaf(x[j], d[j][c], acc[c]);
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < CO; c++)
            if (c < COUT) out[(size_t)ci * COUT + c] = acc[c];
    }
}

__global__ void slab_reduce_kernel(const float *__restrict__ slab, float *__restrict__ dW, int K, int S, int64_t per) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= K * per) return;
    int k = (int)(t / per);
    int64_t e = t - k * per;
    const float *p = slab + (size_t)k * S * per + e;
    // eight slabs' values are loaded before they are added (one load in flight per thread made this pass latency-bound:
    // 1.9 TB/s); the adds stay in slab order, so the result is the one the plain loop gives, bit for bit
    float sum = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = p[(size_t)(s + j) * per];
#pragma unroll
        for (int j = 0; j < 8; j++) sum += v[j];
    }
    for (; s < S; s++) sum += p[(size_t)s * per];
    dW[t] = sum;
}

template <int MT, int NT>
static int launch_dw(const TiledRules &t, const float *X, const float *dY, int nIn, int nOut, int tps, int S,
                     float *slab, hipStream_t st) {
    size_t lds_bytes = (size_t)MT * 16 * nOut * sizeof(float);
    SEI_TRY(ensure_max_lds((const void *)conv_dw_mfma_kernel<MT, NT>, 128 * 1024));
    // 8 * (ci blocks) * K * ceil(S/8) linear ids, decoded in the kernel
    dim3 grid(8 * (nIn / (16 * MT)) * t.K * ((S + 7) / 8), 1, 1);
    conv_dw_mfma_kernel<MT, NT><<<grid, 256, lds_bytes, st>>>(X, dY, t.tile_sub_start, t.seg_start, t.sub_src, t.sub_dst,
                                                              t.n_tiles, t.tile_rows, nIn, nOut, tps, S, t.K, slab);
    SEI_LAUNCH_CHECK();
    return 0;
}

#ifdef SEI_EXPERIMENTS
template <int CT16, int NT>
static int launch_dw_ring(const TiledRules &t, const float *X, const float *dY, int tps, int S, int D, float *slab, hipStream_t st) {
    constexpr int CIN = 16 * CT16, COUT = 16 * NT;
    const size_t lds_bytes = (size_t)D * SUB_SLOTS * (CIN + COUT) * sizeof(float);
    SEI_TRY(ensure_max_lds((const void *)conv_dw_ring_kernel<CT16, NT>, 140 * 1024));
    dim3 grid(8 * t.K * ((S + 7) / 8), 1, 1);
    conv_dw_ring_kernel<CT16, NT><<<grid, 64 * (CT16 / DWR_MT + 2), lds_bytes, st>>>(X, dY, t.tile_sub_start, t.seg_start, t.sub_src,
                                                                               t.sub_dst, t.n_tiles, t.tile_rows, tps, S, t.K, D, slab);
    SEI_LAUNCH_CHECK();
    return 0;
}

// ring depth of the LDS-operand kernel: what fits beside the sub-chunk lists with three (two, one) workgroups per CU
static int dw_ring_depth(int nIn, int nOut) {
    const size_t slot = (size_t)SUB_SLOTS * (nIn + nOut) * sizeof(float), fixed = 2 * DWR_LIST_CAP * sizeof(int) + 1024;
    for (size_t budget : {(size_t)40 * 1024, (size_t)53 * 1024, (size_t)80 * 1024, (size_t)150 * 1024}) {
        const int d = budget > fixed ? (int)((budget - fixed) / slot) : 0;
        if (d >= DWR_LAG + 1) return d > 8 ? 8 : d;
    }
    return 0;
}

#define DWR_CASE(CT_, NT_) \
    if (nIn == 16 * CT_ && nOut == 16 * NT_) return launch_dw_ring<CT_, NT_>(t, X, dY, tps, S, D, slab, st);

static bool dw_ring_instantiated(int nIn, int nOut) {
    const int ct = nIn / 16, nt = nOut / 16;
    if (nIn % 32 || nOut % 16 || ct < 2 || ct > 12 || nt < 2 || nt > 12) return false;
    return nt == ct || nt == ct + 2 || ct == nt + 2;  // C -> C blocks, C -> C +- 32 down-sampling convolutions
}

static int dispatch_dw_ring(const TiledRules &t, const float *X, const float *dY, int nIn, int nOut, int tps, int S, int D,
                            float *slab, hipStream_t st) {
    DWR_CASE(2, 2) DWR_CASE(4, 4) DWR_CASE(6, 6) DWR_CASE(8, 8) DWR_CASE(10, 10) DWR_CASE(12, 12)
    DWR_CASE(2, 4) DWR_CASE(4, 6) DWR_CASE(6, 8) DWR_CASE(8, 10) DWR_CASE(10, 12)
    DWR_CASE(4, 2) DWR_CASE(6, 4) DWR_CASE(8, 6) DWR_CASE(10, 8) DWR_CASE(12, 10)
    SEI_FAIL("conv dW: no LDS-operand instantiation");
}

#endif  // SEI_EXPERIMENTS

#define DW_CASE(MT_, NT_) \
    if (mt == MT_ && nt == NT_) return launch_dw<MT_, NT_>(t, X, dY, nIn, nOut, tps, S, slab, st);

static int dispatch_dw(int mt, int nt, const TiledRules &t, const float *X, const float *dY, int nIn, int nOut, int tps,
                       int S, float *slab, hipStream_t st) {
    DW_CASE(1, 1) DW_CASE(1, 2) DW_CASE(1, 3) DW_CASE(1, 4) DW_CASE(1, 5) DW_CASE(1, 6) DW_CASE(1, 8) DW_CASE(1, 10) DW_CASE(1, 12)
    DW_CASE(2, 1) DW_CASE(2, 2) DW_CASE(2, 3) DW_CASE(2, 4) DW_CASE(2, 5) DW_CASE(2, 6) DW_CASE(2, 8) DW_CASE(2, 10) DW_CASE(2, 12)
    DW_CASE(3, 1) DW_CASE(3, 2) DW_CASE(3, 3) DW_CASE(3, 4) DW_CASE(3, 5) DW_CASE(3, 6) DW_CASE(3, 8) DW_CASE(3, 10) DW_CASE(3, 12)
    DW_CASE(4, 1) DW_CASE(4, 2) DW_CASE(4, 3) DW_CASE(4, 4) DW_CASE(4, 5) DW_CASE(4, 6) DW_CASE(4, 8) DW_CASE(4, 10) DW_CASE(4, 12)
    DW_CASE(6, 6) DW_CASE(6, 8) DW_CASE(5, 10) DW_CASE(5, 12) DW_CASE(8, 8)
    DW_CASE(1, 14) DW_CASE(2, 14) DW_CASE(1, 16) DW_CASE(2, 16) DW_CASE(1, 7) DW_CASE(2, 7) DW_CASE(1, 9) DW_CASE(1, 11) DW_CASE(1, 13) DW_CASE(1, 15)
    SEI_FAIL("conv dW: no MFMA instantiation");
}

int conv_backward_weight(const TiledRules &t, const float *X, const float *dY, float *dW, int nIn, int nOut,
                         hipStream_t st) {
    const int64_t per = (int64_t)nIn * nOut;
    if (t.n_sub == 0 || t.n_dst == 0) {
        SEI_HIP(hipMemsetAsync(dW, 0, (size_t)t.K * per * sizeof(float), st));
        return 0;
    }
    int nt = nOut / 16;
    bool mfma_ok = (nIn % 16 == 0) && (nOut % 16 == 0) &&
                   nt >= 1 && nt <= 16;
    int mt = 0;
    if (mfma_ok) {
        int ct = nIn / 16;
        // ci block height: <= 24 accumulator tiles (96 registers) keeps 3 waves per SIMD resident, which measured
        // faster than wider register tiles at 2 waves (C=96: 6x6 tiles 1.38 ms -> 3x6 tiles 1.17 ms)
        const int cands[] = {4, 3, 2, 1};
        for (int cand : cands) {
            bool inst = (cand == 1) || (cand * nt <= 24 && (nt <= 6 || nt == 8 || nt == 10 || nt == 12)) ||
                        (cand == 2 && (nt == 7 || nt == 14 || nt == 16));
            if (ct % cand == 0 && inst) {
                mt = cand;
                break;
            }
        }
        if (nt == 6 && ct % 2 == 0) mt = 2;  // 96 planes: 2x6 tiles (one dwordx2 + dwordx4/x2 per step) measured 4 % faster than 3x6
        if (!mt) mfma_ok = false;
#ifdef SEI_EXPERIMENTS  // tools/dw_micro.py
        if (const char *e = getenv("SEI_DW_MT")) {
            int em = atoi(e);
            if (em >= 1 && em <= 8 && ct % em == 0 && em * nt <= 64) mt = em;
        }
#endif
    }
    bool small = !mfma_ok && nIn <= 4 && (nOut == 16 || nOut == 32 || nOut == 64);
    const bool narrow = !mfma_ok && !small && nOut <= 8;   // a few output planes: conv_dw_narrow_kernel
    // splits: a split's dY / X rows (re-read by the K workgroups of that split, which share an XCD) should fit
    // that XCD's 4 MiB L2; bounded slab size; at least enough workgroups to fill 256 CUs
    // 3 MiB of rows per split; 1.5 MiB measured faster in the 16-event micro-benchmark but slower in the 64-event step
    // (conv_bwd_weight 101.7 vs 97.4 ms; profiles/r02_conv_dw_knobs.log): tune at the benchmark's size
    int64_t fit_kb = 3072;
#ifdef SEI_EXPERIMENTS
    if (getenv("SEI_DW_FIT_KB")) fit_kb = atoi(getenv("SEI_DW_FIT_KB"));
#endif
    int64_t rows_fit = (fit_kb << 10) / ((int64_t)(nIn + nOut) * 4);
    int tps = (int)(rows_fit / t.tile_rows);
    if (tps > DW_TILES_PER_ITER) tps = DW_TILES_PER_ITER;
    if (tps < 8) tps = 8;
    if (narrow) tps = 2;   // little work per rule: many small splits (the slab is K * S * nIn * nOut floats, tiny here)
    int S = (t.n_tiles + tps - 1) / tps;
    int64_t smax = (512ll << 20) / ((int64_t)t.K * per * 4);
    if (smax < 1) smax = 1;
    if (S > smax) S = (int)smax;
    if (S < 1) S = 1;
    tps = (t.n_tiles + S - 1) / S;
    S = (t.n_tiles + tps - 1) / tps;
    float *slab;
    SEI_TRY(pool_alloc_t(&slab, (size_t)t.K * S * per, st));
    int rc = 0;
    dim3 grid(t.K, S, 1);
#ifdef SEI_EXPERIMENTS  // the LDS-operand kernel: measured 41-67 vs 84-94 TFLOP/s, kept out of the shipped library
    const int ring_d = mfma_ok && dw_ring_instantiated(nIn, nOut) ? dw_ring_depth(nIn, nOut) : 0;
    const bool use_ring = ring_d > 0 && forced_dw_kernel() == DW_KERNEL_RING;
#else
    const bool use_ring = false;
#endif
    if (use_ring) {
#ifdef SEI_EXPERIMENTS
        rc = dispatch_dw_ring(t, X, dY, nIn, nOut, tps, S, ring_d, slab, st);
#endif
    } else if (mfma_ok)
        rc = dispatch_dw(mt, nt, t, X, dY, nIn, nOut, tps, S, slab, st);
    else if (small) {
#define SM(CIN_, COUT_) conv_dw_smallcin_kernel<CIN_, COUT_><<<grid, 256, 0, st>>>(                              \
    X, dY, t.tile_sub_start, t.seg_start, t.sub_src, t.sub_dst, t.n_tiles, t.tile_rows, tps, S, slab)
#define SMC(CIN_) if (nOut == 16) SM(CIN_, 16); else if (nOut == 32) SM(CIN_, 32); else SM(CIN_, 64)
        if (nIn == 1) { SMC(1); } else if (nIn == 2) { SMC(2); } else if (nIn == 3) { SMC(3); } else { SMC(4); }
#undef SMC
#undef SM
        if (hipGetLastError() != hipSuccess) rc = 1;
    } else if (narrow) {
        if (nOut <= 4)
            conv_dw_narrow_kernel<4><<<grid, 256, 0, st>>>(X, dY, t.tile_sub_start, t.seg_start, t.sub_src, t.sub_dst,
                                                           t.n_tiles, t.tile_rows, nIn, nOut, tps, S, slab);
        else
            conv_dw_narrow_kernel<8><<<grid, 256, 0, st>>>(X, dY, t.tile_sub_start, t.seg_start, t.sub_src, t.sub_dst,
                                                           t.n_tiles, t.tile_rows, nIn, nOut, tps, S, slab);
        if (hipGetLastError() != hipSuccess) rc = 1;
    } else {
        conv_dw_generic_kernel<<<grid, 256, 0, st>>>(X, dY, t.tile_sub_start, t.seg_start, t.sub_src, t.sub_dst,
                                                     t.n_tiles, t.tile_rows, nIn, nOut, tps, S, slab);
        if (hipGetLastError() != hipSuccess) rc = 1;
    }
    if (rc) {
        pool_free(slab);
        if (!mfma_ok) SEI_FAIL("conv dW launch failed");
        return rc;
    }
    int64_t tot = (int64_t)t.K * per;
    slab_reduce_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(slab, dW, t.K, S, per);
    SEI_LAUNCH_CHECK();
    pool_free(slab);
    return 0;
}

int dw_ring_timeouts(int *out, bool reset) {
#ifdef SEI_EXPERIMENTS
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_dw_timeouts), sizeof(int)) != hipSuccess) return 1;
    *out = v;
    if (reset) {
        int z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_dw_timeouts), &z, sizeof(int)) != hipSuccess) return 1;
    }
#else
    (void)reset;
    *out = 0;
#endif
    return 0;
}

}  // namespace sei

#ifdef SEI_CLOCK
extern "C" int sei_debug_clock_dw(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sei::g_clk), sizeof(sei::g_clk)) != hipSuccess) return 1;
    if (reset) {
        static unsigned long long z[8][2] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_clk), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
