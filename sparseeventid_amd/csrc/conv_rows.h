// Device helpers shared by the kernels whose waves own whole dst-row bands (conv_wave.hip, conv_band.hip): gfx950 only.
// Column convention of these kernels: MFMA column tile 4q+j at lane r16 is output column 64q + 4*r16 + j of the wave's
// column block (remainder tiles: pairs at 2*r16 + j, a single tile at r16), so that
//   * the B operands of four column tiles are ONE 16-byte load from a weight row in its natural [nIn][nOut] layout, and
//   * a lane's partial sums of four column tiles are 16 contiguous bytes of the accumulator row in LDS.
#pragma once
#include "common.h"
#include "conv_tile.h"

namespace sei {

// Unconditional load of a clamped row, and NO select: a padding slot (src < 0) reads row 0 and its products are
// garbage, but an MFMA output row depends on its own A row only and padding slots are never written back (wrmw).
// (A select on the loaded value would make hipcc wait for the gather right where it is issued.)
template <int ABL = 0>
__device__ __forceinline__ float4 wload_a(const float *__restrict__ X, int src, int CIN, int off) {
    uint32_t row = (uint32_t)(src > 0 ? src : 0);
    if (ABL & 1) row &= 1023u;
    if (ABL & 32) return make_float4(1.f + row, 2.f, 3.f + off, 4.f);  // diagnostic: no gather at all
    return *reinterpret_cast<const float4 *>(X + (uint64_t)row * (uint32_t)CIN + off);
}

// B operands of one input plane for the NT column tiles from a weight row (global memory or an LDS copy of it)
template <int NT>
__device__ __forceinline__ void wload_b(const float *__restrict__ wrow, int r16, float (&b)[NT]) {
    constexpr int B4 = NT / 4, REM = NT % 4;
#pragma unroll
    for (int q = 0; q < B4; q++) {
        const float4 v = *reinterpret_cast<const float4 *>(wrow + 64 * q + 4 * r16);
        b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
    }
    if (REM >= 2) {
        const float2 v = *reinterpret_cast<const float2 *>(wrow + 64 * B4 + 2 * r16);
        b[4 * B4 + 0] = v.x; b[4 * B4 + 1] = v.y;
    }
    if (REM == 1) b[4 * B4] = wrow[64 * B4 + r16];
    if (REM == 3) b[4 * B4 + 2] = wrow[64 * B4 + 32 + r16];
}

// tile[dst rows of the sub-chunk] += partial sums.  D[row = 4g+i][col tile n] of lane (g, r16) belongs to rule slot
// 4g+i; slots [0, nv) are rules (padding sits at the end of a sub-chunk and must not write: its dst is row 0).
// All reads are issued before the adds.
template <int NT>
__device__ __forceinline__ void wrmw(float *acc_tile, int AS, uint32_t d4, int nv, int g, int r16,
                                     const f32x4 (&acc)[NT]) {
    constexpr int B4 = NT / 4, REM = NT % 4;
    float4 t4[4][B4 > 0 ? B4 : 1];
    float2 t2[4];
    float t1[4];
    float *base[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        base[i] = acc_tile + ((d4 >> (8 * i)) & 0xFF) * AS;
#pragma unroll
        for (int q = 0; q < B4; q++) t4[i][q] = *reinterpret_cast<const float4 *>(base[i] + 64 * q + 4 * r16);
        if (REM >= 2) t2[i] = *reinterpret_cast<const float2 *>(base[i] + 64 * B4 + 2 * r16);
        if (REM == 1) t1[i] = base[i][64 * B4 + r16];
        if (REM == 3) t1[i] = base[i][64 * B4 + 32 + r16];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (4 * g + i < nv) {
#pragma unroll
            for (int q = 0; q < B4; q++) {
                float4 v = t4[i][q];
                v.x += acc[4 * q + 0][i]; v.y += acc[4 * q + 1][i]; v.z += acc[4 * q + 2][i]; v.w += acc[4 * q + 3][i];
                *reinterpret_cast<float4 *>(base[i] + 64 * q + 4 * r16) = v;
            }
            if (REM >= 2) {
                float2 v = t2[i];
                v.x += acc[4 * B4 + 0][i]; v.y += acc[4 * B4 + 1][i];
                *reinterpret_cast<float2 *>(base[i] + 64 * B4 + 2 * r16) = v;
            }
            if (REM == 1) base[i][64 * B4 + r16] = t1[i] + acc[4 * B4][i];
            if (REM == 3) base[i][64 * B4 + 32 + r16] = t1[i] + acc[4 * B4 + 2][i];
        }
    }
}

// accumulator band <- bias, or the rows of `init` (Y = init + conv: gradient accumulation); one wave, TR rows x NC columns
template <int NC>
__device__ __forceinline__ void band_init(float *acc_tile, const float *__restrict__ bias, const float *__restrict__ init,
                                          int64_t row0, int TR, int64_t n_dst, int ldy, int col0, int lane) {
    constexpr int V = NC >> 2;
    for (int idx = lane; idx < TR * V; idx += 64) {
        const int r = idx / V, v = idx - r * V;
        float4 b = bias ? *reinterpret_cast<const float4 *>(bias + col0 + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (init && row0 + r < n_dst) b = *reinterpret_cast<const float4 *>(init + (size_t)(row0 + r) * ldy + col0 + 4 * v);
        *reinterpret_cast<float4 *>(acc_tile + (size_t)r * NC + 4 * v) = b;
    }
}

// epilogue of one wave's band: every dst row is written once, as whole float4 row segments; optional BatchNorm
// statistics of the output from the LDS band (see conv_tile.h: tile_stats): [tile * PARTS + part][2][ldy] doubles
template <int NC>
__device__ __forceinline__ void band_store(const float *acc_tile, float *__restrict__ Y, double *__restrict__ stats, int tile,
                                           int64_t row0, int TR, int64_t n_dst, int ldy, int col0, int lane,
                                           const ConvEpilogue &ep = ConvEpilogue{}) {
    constexpr int V = NC >> 2;
    if (ep.mean) {  // inference epilogue (common.h)
        for (int idx = lane; idx < TR * V; idx += 64) {
            const int r = idx / V, v = idx - r * V;
            if (row0 + r < n_dst) {
                const float4 a = *reinterpret_cast<const float4 *>(acc_tile + (size_t)r * NC + 4 * v);
                *reinterpret_cast<float4 *>(Y + (size_t)(row0 + r) * ldy + col0 + 4 * v) =
                    conv_epilogue_apply(ep, a, col0 + 4 * v, ep.res ? ep.res + (size_t)(row0 + r) * ldy : nullptr);
            }
        }
    } else
    for (int idx = lane; idx < TR * V; idx += 64) {
        const int r = idx / V, v = idx - r * V;
        if (row0 + r < n_dst)
            *reinterpret_cast<float4 *>(Y + (size_t)(row0 + r) * ldy + col0 + 4 * v) =
                *reinterpret_cast<const float4 *>(acc_tile + (size_t)r * NC + 4 * v);
    }
    if (stats) {
        constexpr int PARTS = 64 / NC > 0 ? 64 / NC : 1;
        const int64_t left = n_dst - row0;
        const int rows = left < TR ? (int)left : TR;
        for (int e = lane; e < PARTS * NC; e += 64) {
            const int part = e / NC, c = e - part * NC;
            // four independent fp64 chains (a single chain is latency-bound), combined in a fixed order
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
            int r = part;
            for (; r + 3 * PARTS < rows; r += 4 * PARTS) {
                const double v0 = (double)acc_tile[(size_t)r * NC + c], v1 = (double)acc_tile[(size_t)(r + PARTS) * NC + c];
                const double v2 = (double)acc_tile[(size_t)(r + 2 * PARTS) * NC + c], v3 = (double)acc_tile[(size_t)(r + 3 * PARTS) * NC + c];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
                q0 += v0 * v0; q1 += v1 * v1; q2 += v2 * v2; q3 += v3 * v3;
            }
            for (; r < rows; r += PARTS) {
                const double v = (double)acc_tile[(size_t)r * NC + c];
                a0 += v;
                q0 += v * v;
            }
            const double s0 = (a0 + a1) + (a2 + a3), s1 = (q0 + q1) + (q2 + q3);
            double *o = stats + ((size_t)tile * PARTS + part) * 2 * ldy + col0 + c;
            o[0] = s0;
            o[ldy] = s1;
        }
    }
}

}  // namespace sei
