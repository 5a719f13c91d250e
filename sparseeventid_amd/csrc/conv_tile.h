// Device helpers shared by the workgroup-tile convolution kernels (conv_fwd.hip): gfx950 only.
#pragma once
#include "common.h"

namespace sei {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Workgroup id -> work item for kernels whose consecutive items share gathered rows.  Workgroups are dealt round-robin
// to the 8 XCDs (blockIdx & 7); every XCD gets CONTIGUOUS runs of XCD_CHUNK items, the runs dealt round-robin to the XCDs:
// neighbours inside a run share that XCD's L2, and every XCD samples all of the batch's events, so they finish together
// (one run per XCD -- round 1's mapping -- leaves them with unequal work when the events differ).  Measured on the
// 64-event step, two devices: runs of 4-16 items 368.8 / 366.5 ms, one run per XCD 370.8 / 368.0, runs of 1 = plain
// round-robin 371.2, runs of 256 370.2 (tools/bench_ab.sh).  `total` items, grid = 8 * ceil(total / 8) workgroups.
#ifndef SEI_XCD_CHUNK
#define SEI_XCD_CHUNK 16
#endif
__device__ inline int xcd_item(int total) {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3, per = (total + 7) >> 3;
    constexpr int CH = SEI_XCD_CHUNK;
    const int full = per / CH;  // complete rounds of 8 runs
    if (i < full * CH) return ((i / CH) * 8 + x) * CH + i % CH;
    const int rem = per - full * CH;  // the tail: one shorter run per XCD
    return full * CH * 8 + x * rem + (i - full * CH);
}

// unconditional load (clamped row) + select: conditional loads make hipcc serialise them with vmcnt(0)
__device__ inline float4 load_a(const float *__restrict__ X, int src, int CIN, int off) {
    int row = src > 0 ? src : 0;
#ifdef SEI_ABLATE_GATHER  // diagnostic: every gather hits the caches (results are wrong, timing only)
    row &= 1023;
#endif
#ifdef SEI_ABLATE_SHAPE1  // diagnostic: the 16 rows of a gather instruction collapse into ONE row (one cache line per instruction)
    row = __builtin_amdgcn_readfirstlane(row);
#endif
#ifdef SEI_ABLATE_NOGATHER  // diagnostic: no gather at all
    float4 v = make_float4(1.f + row, 2.f, 3.f + off, 4.f);
#else
    float4 v = *reinterpret_cast<const float4 *>(X + (size_t)row * CIN + off);
#endif
#ifndef SEI_TILE_NOSELECT
    // Padding slots feed ZEROS to the MFMA.  Their output rows are never written back, so garbage would be equally
    // correct -- but measured on one device (tools/tile_ab.sh, levels 1-3) zeros run 5-7 % faster than row-0 data
    // (51.7 / 66.1 / 71.8 vs 49.2 / 61.9 / 68.5 TFLOP/s): the chip holds a higher clock on the ~10 % all-zero rows.
    if (src < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    return v;
}

// Weight slice global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, asynchronous.
// The slice is contiguous in W; every wave-instruction moves 1 KiB (64 lanes x 16 B) to a
// wave-uniform LDS base + lane*16.  PIECES = 1-KiB pieces per slice, dealt round-robin to the NW waves.
template <int PIECES, int NW = 4>
__device__ __forceinline__ void stage_slice(const float *__restrict__ wslice, float *wdst, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < (PIECES + NW - 1) / NW; q++) {
        const int piece = q * NW + wave;
        if (piece < PIECES) {  // wave-uniform
            const float *gp = wslice + (size_t)piece * 256 + lane * 4;
            float *lp = wdst + (size_t)piece * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                             (__attribute__((address_space(3))) void *)lp, 16, 0, 0);
        }
    }
}

// accumulator tile <- bias, or the rows of `init` (Y = init + conv: gradient accumulation)
template <int TR, int COUT, int NTH = 256>
__device__ __forceinline__ void tile_init(float *acc_tile, const float *__restrict__ bias, const float *__restrict__ init,
                                          int64_t row0, int64_t n_dst, int ldy, int ycol0) {
    constexpr int V = COUT >> 2;
    for (int idx = threadIdx.x; idx < TR * V; idx += NTH) {
        int r = idx / V, v = idx - r * V;
        float4 b = bias ? *reinterpret_cast<const float4 *>(bias + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (init && row0 + r < n_dst) b = *reinterpret_cast<const float4 *>(init + (size_t)(row0 + r) * ldy + ycol0 + 4 * v);
        *reinterpret_cast<float4 *>(acc_tile + (size_t)r * COUT + 4 * v) = b;
    }
}

// every dst row is written once, as whole float4 row segments
template <int TR, int COUT, int NTH = 256>
__device__ __forceinline__ void tile_store(const float *acc_tile, float *__restrict__ Y, int64_t row0, int64_t n_dst, int ldy,
                                           int ycol0, const ConvEpilogue &ep = ConvEpilogue{}) {
    constexpr int V = COUT >> 2;
    if (ep.mean) {  // inference: eval-mode BatchNorm (+ residual) + LeakyReLU on the way out (common.h)
        for (int idx = threadIdx.x; idx < TR * V; idx += NTH) {
            int r = idx / V, v = idx - r * V;
            if (row0 + r < n_dst) {
                const float4 a = *reinterpret_cast<const float4 *>(acc_tile + (size_t)r * COUT + 4 * v);
                *reinterpret_cast<float4 *>(Y + (size_t)(row0 + r) * ldy + ycol0 + 4 * v) =
                    conv_epilogue_apply(ep, a, ycol0 + 4 * v, ep.res ? ep.res + (size_t)(row0 + r) * ldy : nullptr);
            }
        }
        return;
    }
    for (int idx = threadIdx.x; idx < TR * V; idx += NTH) {
        int r = idx / V, v = idx - r * V;
        if (row0 + r < n_dst)
            *reinterpret_cast<float4 *>(Y + (size_t)(row0 + r) * ldy + ycol0 + 4 * v) =
                *reinterpret_cast<const float4 *>(acc_tile + (size_t)r * COUT + 4 * v);
    }
}

// BatchNorm statistics of the output while the tile is still in LDS: column sums over the tile's valid rows in fp64,
// PARTS = max(1, 256 / COUT) row-interleaved partial sums per tile (fixed order: bit-identical reruns); the BN forward
// only finalizes them (no second read of Y).  Layout [tile * PARTS + part][2][ldy].
template <int TR, int COUT, int NTH = 256>
__device__ __forceinline__ void tile_stats(const float *acc_tile, double *__restrict__ stats, int tile, int64_t row0,
                                           int64_t n_dst, int ldy, int ycol0) {
    constexpr int PARTS = 256 / COUT > 0 ? 256 / COUT : 1;
    const int64_t left = n_dst - row0;
    const int rows = left < TR ? (int)left : TR;
    for (int e = threadIdx.x; e < PARTS * COUT; e += NTH) {
        const int part = e / COUT, c = e - part * COUT;
        // four independent fp64 chains (a single chain is latency-bound), combined in a fixed order
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
        int r = part;
        for (; r + 3 * PARTS < rows; r += 4 * PARTS) {
            const double v0 = (double)acc_tile[(size_t)r * COUT + c], v1 = (double)acc_tile[(size_t)(r + PARTS) * COUT + c];
            const double v2 = (double)acc_tile[(size_t)(r + 2 * PARTS) * COUT + c], v3 = (double)acc_tile[(size_t)(r + 3 * PARTS) * COUT + c];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            q0 += v0 * v0; q1 += v1 * v1; q2 += v2 * v2; q3 += v3 * v3;
        }
        for (; r < rows; r += PARTS) {
            const double v = (double)acc_tile[(size_t)r * COUT + c];
            a0 += v;
            q0 += v * v;
        }
        const double s0 = (a0 + a1) + (a2 + a3), s1 = (q0 + q1) + (q2 + q3);
        double *o = stats + ((size_t)tile * PARTS + part) * 2 * ldy + ycol0 + c;
        o[0] = s0;
        o[ldy] = s1;
    }
}

}  // namespace sei
