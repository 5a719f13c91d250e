// Sparse convolution forward / dX for gfx950, wave-private form (scn.SubmanifoldConvolution / scn.Convolution:
// resnet.py:30-50,105-110; sparse_building_blocks.py:29-34,110-117).
//
// One WAVEFRONT owns a tile of TR (64/128) dst rows x NC = 16*NT output columns whose fp32 accumulators live in
// its private LDS; there is no barrier, no shared weight buffer and no inter-wave dependency anywhere in the
// kernel.  The wave walks the filter offsets k; the tile's rules of one k are compacted into sub-chunks of 16
// rules (the same tiled rulebook the workgroup-tile kernel uses, built with tile_rows = TR) and processed in
// groups of up to G sub-chunks that share the B operands:
//   * A (features): src rows gathered straight into MFMA A-operand registers, float4 per lane per 16 input planes,
//     one slice ahead of the MFMAs that consume them;
//   * B (weights): read from W in its NATURAL [K][nIn][nOut] layout straight from L2 (W is <= 4 MB and shared by
//     every wave on the chip), one slice ahead; MFMA column tile (4q+j, lane r) is output column 64q + 4r + j, so
//     one global_load_dwordx4 feeds four column tiles -- no weight permutation pass for the forward;
//   * v_mfma_f32_16x16x4_f32 (exact fp32) into zero-initialised accumulators, one chain of nIn/4 steps per offset;
//   * after the offset's last slice the 16 x NC partial sums are added to the tile rows they target by a plain LDS
//     read-modify-write; with the column map above a lane's values of four column tiles are 16 contiguous bytes:
//     ds_read_b128 / ds_write_b128 instead of four b32 pairs.
// Rows of one offset are distinct and a wave's LDS operations complete in order, so the summation order is fixed
// (and the same as the reference's per-offset  out[rows] += in[rows] @ W[k]): run-to-run deterministic, no
// atomics; every dst row is written to HBM exactly once.  Compared with the workgroup-tile kernel the structural
// efficiency on the DUNE batch rises from padding x wave balance = 0.24 / 0.55 / 0.66 / 0.69 (levels 0-3) to the
// padding of 64-row tiles alone, 0.46 / 0.73 / 0.82 / 0.84 (tools/structure_stats.py): waves never wait for each
// other, and the per-phase barrier, zero-init and weight-staging costs are gone.
#include "common.h"
#include "conv_tile.h"
#include "conv_rows.h"

#include <stdlib.h>
#include <string.h>

namespace sei {

// Diagnostic builds only (make EXTRA="-DSEI_STAMPS" / EXTRA="-DSEI_WAVE_ABL=n"; tools/wave_diag.sh): every wave adds
// the s_memtime ticks it spends in each part of the kernel to g_wstamps; the ablation bits remove one cost at a time
// (1: gathers hit the caches, 2: weights from one L1-resident slice, 4: no tile read-modify-write, 8: no MFMAs;
// results are wrong, timing only).  The product build compiles none of it.
#ifndef SEI_WAVE_ABL
#define SEI_WAVE_ABL 0
#endif
#ifdef SEI_STAMPS
__device__ unsigned long long g_wstamps[16];
struct WStamp {
    unsigned long long t, acc[16];
    __device__ __forceinline__ WStamp() {
        for (int i = 0; i < 16; i++) acc[i] = 0;
        t = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void mark(int i) {
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long n = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        acc[i] += n - t;
        t = n;
    }
    __device__ __forceinline__ void add(int i, unsigned long long v) { acc[i] += v; }
    __device__ __forceinline__ void flush() {
        if (threadIdx.x == 0)
            for (int i = 0; i < 16; i++) atomicAdd(&g_wstamps[i], acc[i]);
    }
};
#define WSTAMP(i) st.mark(i)
#define WSTAMP_ADD(i, v) st.add(i, v)
#else
struct WStamp {
    __device__ __forceinline__ void flush() {}
};
#define WSTAMP(i)
#define WSTAMP_ADD(i, v)
#endif

__device__ __forceinline__ f32x4 wmfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int G>
struct WaveIdx {
    int src[G];       // lane r16: src row of rule slot r16 of sub-chunk m (raw load; validity applied by row())
    uint32_t d4[G];   // lane group g: packed tile-local dst rows of slots 4g..4g+3
    int cnt;          // sub-chunks of the group (wave-uniform, 0 past the end)
    __device__ __forceinline__ int row(int m) const { return m < cnt ? src[m] : -1; }
};

// one slice (16 input planes = 4 MFMA steps) for MS sub-chunks sharing the B operands
template <int NT, int G, int MS>
__device__ __forceinline__ void wslice(const float4 (&a)[G], const float (&b)[4][NT], const int (&rows)[G],
                                       f32x4 (&acc)[MS][NT]) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int m = 0; m < MS; m++) {
            float av = e == 0 ? a[m].x : e == 1 ? a[m].y : e == 2 ? a[m].z : a[m].w;
#ifndef SEI_WAVE_NOZERO
            // padding slots feed zeros (applied HERE, a slice after the gather was issued -- a select at the load would
            // wait for it): correctness does not need it, the chip clocks higher on all-zero rows (see conv_fwd.hip)
            av = rows[m] >= 0 ? av : 0.f;
#endif
#pragma unroll
            for (int n = 0; n < NT; n++) {
                if (SEI_WAVE_ABL & 8) {
                    asm volatile("" ::"v"(av), "v"(b[e][n]));  // keeps the loads alive
                } else {
                    acc[m][n] = wmfma16(av, b[e][n], acc[m][n]);
                }
            }
        }
    }
}

// operands of slice j (input planes 16j .. 16j+15) of offset k for the G sub-chunks with src rows `rows`
template <int NT, int G>
__device__ __forceinline__ void wprefetch(const float *__restrict__ X, const float *__restrict__ wl, int ldy, int CIN,
                                          int k, int j, const int (&rows)[G], int cnt, int g, int r16, float4 (&a)[G],
                                          float (&b)[4][NT]) {
    // only the sub-chunks that exist are gathered (wave-uniform branches): every vector-memory instruction costs the
    // CU's address path ~1 KiB of its ~20-30 B/clk, used or not
#pragma unroll
    for (int m = 0; m < G; m++)
        if (m == 0 || m < cnt) a[m] = wload_a<SEI_WAVE_ABL>(X, rows[m], CIN, 16 * j + 4 * g);
    const float *wn = (SEI_WAVE_ABL & 2) ? wl : wl + ((size_t)k * CIN + 16 * j) * ldy;
#pragma unroll
    for (int e = 0; e < 4; e++) wload_b<NT>(wn + (size_t)e * ldy, r16, b[e]);
}

// [r5] B operands of slice j of offset k from the PACKED copy of the 32 x 32 weights (conv.hip: weight_index, col_group -2): two
// global_load_dwordx4 per slice instead of four dwordx2 -- every vector-memory instruction costs the CU's address path the same
// ~1 KiB whatever it carries, and at 32 planes (one sub-chunk per offset and tile) the B loads are two thirds of them
__device__ __forceinline__ void wload_b_packed(const float *__restrict__ Wp, int NS, int k, int j, int lane, float (&b)[4][2]) {
    const float4 *p = reinterpret_cast<const float4 *>(Wp) + ((size_t)(k * NS + j) * 2) * 64 + lane;
    const float4 v0 = p[0], v1 = p[64];
    b[0][0] = v0.x; b[0][1] = v0.y; b[1][0] = v0.z; b[1][1] = v0.w;
    b[2][0] = v1.x; b[2][1] = v1.y; b[3][0] = v1.z; b[3][1] = v1.w;
}

// One group = MS sub-chunks of offset kc: all NS slices, then the read-modify-write of the tile rows.  On entry
// (a0, b0) hold slice 0 of this group; on exit they hold slice 0 of the NEXT group (rows rn, offset kn).  The slices
// ping-pong between two operand sets so that no register copies sit between the MFMAs, and every slice's loads are
// issued one slice (>= 4*MS*NT MFMAs) before they are consumed.  Straight-line per MS: the group size is dispatched
// ONCE outside (a branch inside the slice loop made hipcc copy all accumulators every slice).
template <int NT, int G, int MS>
__device__ __forceinline__ void wgroup(const float *__restrict__ X, const float *__restrict__ wl, int ldy, int CIN, int NS,
                                       int kc, int kn, const int (&rc)[G], const int (&rn)[G], int cn, const uint32_t (&d4)[G],
                                       const int (&src)[G], float *acc_tile, int g, int r16, float4 (&a0)[G],
                                       float (&b0)[4][NT], WStamp &st) {
    constexpr int NC = 16 * NT;
    float4 a1[G];
    float b1[4][NT];
    f32x4 acc[MS][NT];
#pragma unroll
    for (int m = 0; m < MS; m++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 2 <= NS; j += 2) {
        wprefetch<NT, G>(X, wl, ldy, CIN, kc, j + 1, rc, MS, g, r16, a1, b1);
        // the loads above are ISSUED before this slice's MFMAs (hipcc otherwise sinks them behind the MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(1);
        wslice<NT, G, MS>(a0, b0, rc, acc);
        WSTAMP(2);
        const bool same = j + 2 < NS;  // else: slice 0 of the next group
        int rr[G];
#pragma unroll
        for (int m = 0; m < G; m++) rr[m] = same ? rc[m] : rn[m];
        wprefetch<NT, G>(X, wl, ldy, CIN, same ? kc : kn, same ? j + 2 : 0, rr, same ? MS : cn, g, r16, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(1);
        wslice<NT, G, MS>(a1, b1, rc, acc);
        WSTAMP(2);
        WSTAMP_ADD(8, 2 * 4 * MS * NT);
    }
    if (j < NS) {  // odd slice count: the next group's slice 0 lands in the second set and is moved over
        wprefetch<NT, G>(X, wl, ldy, CIN, kn, 0, rn, cn, g, r16, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(1);
        wslice<NT, G, MS>(a0, b0, rc, acc);
        WSTAMP(2);
        WSTAMP_ADD(8, 4 * MS * NT);
#pragma unroll
        for (int m = 0; m < G; m++) a0[m] = a1[m];
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int n = 0; n < NT; n++) b0[e][n] = b1[e][n];
    }
#pragma unroll
    for (int m = 0; m < MS; m++) {
        const int nv = __popcll(__ballot(src[m] >= 0) & 0xFFFFull);
        if (SEI_WAVE_ABL & 4) {
#pragma unroll
            for (int n = 0; n < NT; n++) asm volatile("" ::"v"(acc[m][n]));
        } else {
            wrmw<NT>(acc_tile, NC, d4[m], nv, g, r16, acc[m]);
        }
    }
    WSTAMP(3);
    WSTAMP_ADD(9, 1);
    WSTAMP_ADD(10, MS);
}

// [r5] 32 input planes (NS == 2: level 0 of the DUNE network).  wgroup above requests a slice ONE slice ahead of its use: 16 MFMAs
// (512 pipe cycles) of cover for gathers that come from beyond the L2 -- at ~10 waves per CU the wave then waits ~3000 cycles per
// slice (197 k cycles per 128-row tile against 23 k of MFMA work).  Here BOTH slices of the NEXT group are requested before this
// group's first MFMA: a whole group (32 MFMAs + the tile read-modify-write) of cover, two operand sets (cur / nxt) that the
// caller swaps between groups (no register copies).  Every request is unconditional (a missing sub-chunk re-reads row 0): the
// number of requests in flight is then a compile-time constant and hipcc's own vmcnt waits stay exact.
template <int NT, int G, int MS, bool PACKED = false>
__device__ __forceinline__ void wgroup2(const float *__restrict__ X, const float *__restrict__ wl, int ldy, int CIN, int kn,
                                        const int (&rc)[G], const int (&rn)[G], const uint32_t (&d4)[G], const int (&src)[G],
                                        float *acc_tile, int g, int r16, const float4 (&ac0)[G], const float (&bc0)[4][NT],
                                        const float4 (&ac1)[G], const float (&bc1)[4][NT], float4 (&an0)[G], float (&bn0)[4][NT],
                                        float4 (&an1)[G], float (&bn1)[4][NT], WStamp &st, const float *wpacked = nullptr) {   // PACKED: wpacked = the packed weights
    constexpr int NC = 16 * NT;
    f32x4 acc[MS][NT];
#pragma unroll
    for (int m = 0; m < MS; m++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int m = 0; m < G; m++) (j == 0 ? an0 : an1)[m] = wload_a<SEI_WAVE_ABL>(X, rn[m], CIN, 16 * j + 4 * g);
        if constexpr (PACKED && NT == 2) {   // (a compile-time choice: a branch around the loads would cost hipcc's exact waits)
            wload_b_packed(wpacked, 2, kn, j, 16 * g + r16, j == 0 ? bn0 : bn1);
        } else {
            const float *wn = wl + ((size_t)kn * CIN + 16 * j) * ldy;
#pragma unroll
            for (int e = 0; e < 4; e++) wload_b<NT>(wn + (size_t)e * ldy, r16, (j == 0 ? bn0 : bn1)[e]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // the requests are ISSUED before this group's MFMAs
    WSTAMP(1);
    wslice<NT, G, MS>(ac0, bc0, rc, acc);
    wslice<NT, G, MS>(ac1, bc1, rc, acc);
    WSTAMP(2);
    WSTAMP_ADD(8, 2 * 4 * MS * NT);
#pragma unroll
    for (int m = 0; m < MS; m++) {
        const int nv = __popcll(__ballot(src[m] >= 0) & 0xFFFFull);
        wrmw<NT>(acc_tile, NC, d4[m], nv, g, r16, acc[m]);
    }
    WSTAMP(3);
    WSTAMP_ADD(9, 1);
    WSTAMP_ADD(10, MS);
}

#ifndef SEI_WAVE_GROUP_AHEAD
#define SEI_WAVE_GROUP_AHEAD 1
#endif
constexpr bool GROUP_AHEAD = SEI_WAVE_GROUP_AHEAD != 0;   // experiment builds: -DSEI_WAVE_GROUP_AHEAD=0 = round 2's one-slice-ahead form at 32 planes

constexpr int wave_occ(int nt, int g) {
    // waves per SIMD the register allocator should aim for: accumulators + RMW temporaries + two A and two B
    // slice sets + ~48 registers of addressing / indices (the LDS tile decides the real occupancy)
    // [r5] + the second operand pair of the group-ahead form (32 input planes): LDS holds ~10 waves per CU at 16 KB each, so
    // three waves per SIMD (168 registers) cost no occupancy, while four (128) spilled 52 registers there
    const int est = g * nt * 4 + nt * 4 + 2 * g * 4 + 2 * 4 * nt + 48 + (SEI_WAVE_GROUP_AHEAD && nt <= 2 ? 2 * (g * 4 + 4 * nt) : 0);
    if (SEI_WAVE_GROUP_AHEAD && nt <= 2 && est <= 160) return 3;
    return est <= 120 ? 4 : (est <= 160 ? 3 : (est <= 248 ? 2 : 1));
}

template <int NT, int G, bool PACKED = false>
__global__ __launch_bounds__(64, wave_occ(NT, G)) void conv_wave_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int CIN, int n_tiles,
    const float *__restrict__ init, int ldy, int blocks, int TR, double *__restrict__ stats, ConvEpilogue ep) {
    extern __shared__ float lds[];
    constexpr int NC = 16 * NT;  // columns of this wave = LDS row stride
    // work item = (tile, column block) in the XCD-aware order of xcd_item (conv_tile.h): neighbouring tiles and the column
    // blocks of a tile share an XCD's L2 (speed only)
    const int nwork = n_tiles * blocks;
    const int vid = xcd_item(nwork);
    if (vid >= nwork) return;
    const int tile = vid / blocks, cblk = vid - tile * blocks;
    const int col0 = cblk * NC;
    const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
    float *acc_tile = lds;
    WStamp st;

    // accumulators start from the bias, or from the rows of `init` (Y = init + conv: gradient accumulation)
    {
        constexpr int V = NC >> 2;
        const int64_t row0i = (int64_t)tile * TR;
        for (int idx = lane; idx < TR * V; idx += 64) {
            const int r = idx / V, v = idx - r * V;
            float4 b = bias ? *reinterpret_cast<const float4 *>(bias + col0 + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (init && row0i + r < n_dst)
                b = *reinterpret_cast<const float4 *>(init + (size_t)(row0i + r) * ldy + col0 + 4 * v);
            *reinterpret_cast<float4 *>(acc_tile + (size_t)r * NC + 4 * v) = b;
        }
    }

    // segment table of the tile: lane l holds seg[l] (K + 1 <= 64); nonempty offsets as a bit mask
    const int segv = lane <= K ? tile_seg[(size_t)tile * (K + 1) + lane] : 0;
    const int nsv = __shfl_down(segv, 1) - segv;
    const unsigned long long kmask = __ballot(lane < K && nsv > 0);
    const int64_t sub0 = tile_sub_start[tile];

    struct It { int k, s, e; };
    auto seek = [&](int kfrom) {  // first nonempty offset >= kfrom (K if none)
        It it;
        const unsigned long long rem = kfrom < 64 ? (kmask >> kfrom) << kfrom : 0ull;
        if (rem) {
            it.k = __builtin_amdgcn_readfirstlane(__ffsll((long long)rem) - 1);
            it.s = __builtin_amdgcn_readlane(segv, it.k);
            it.e = __builtin_amdgcn_readlane(segv, it.k + 1);
        } else {
            it.k = K; it.s = 0; it.e = 0;
        }
        return it;
    };
    auto advance = [&](It it) {
        if (it.k >= K) return it;
        it.s += G;
        if (it.s >= it.e) it = seek(it.k + 1);
        return it;
    };
    auto load_idx = [&](const It &it) {
        WaveIdx<G> r;
        r.cnt = __builtin_amdgcn_readfirstlane(it.k < K ? (it.e - it.s < G ? it.e - it.s : G) : 0);
#pragma unroll
        for (int m = 0; m < G; m++) {
            const int s = m < r.cnt ? it.s + m : (it.k < K ? it.s : 0);  // clamped: never branch around a load
            const int64_t sb = (sub0 + s) * SUB_SLOTS;
            r.src[m] = sub_src[sb + r16];
            r.d4[m] = *reinterpret_cast<const uint32_t *>(sub_dst + sb + 4 * g);
        }
        return r;
    };

    It it0 = seek(0);
    if (it0.k < K) {
        const int NS = CIN >> 4;  // slices per offset
        It it1 = advance(it0), it2 = advance(it1);
        WaveIdx<G> ix0 = load_idx(it0), ix1 = load_idx(it1), ix2 = load_idx(it2);
        const float *wl = W + (size_t)(4 * g) * ldy + col0;  // this lane's weight rows: plane 16j + 4g + e
        float4 a0[G];
        float b0[4][NT];
        {
            int r0[G];
#pragma unroll
            for (int m = 0; m < G; m++) r0[m] = ix0.row(m);
            wprefetch<NT, G>(X, wl, ldy, CIN, it0.k, 0, r0, ix0.cnt, g, r16, a0, b0);
        }
        __syncthreads();  // single-wave workgroup: orders the tile initialisation before the first read-modify-write
        WSTAMP(0);

        bool ahead = false;
        if constexpr (GROUP_AHEAD && NT <= 2) ahead = NS == 2;   // the 32-column kernels only: wider ones would pay for the second operand pair in registers
        if (ahead) {
          if constexpr (GROUP_AHEAD && NT <= 2) {
            // two operand sets, swapped between groups: P holds this group's two slices, Q receives the next group's
            float4 pa1[G], qa0[G], qa1[G];
            float pb1[4][NT], qb0[4][NT], qb1[4][NT];
            {
                int r0[G];
#pragma unroll
                for (int m = 0; m < G; m++) r0[m] = ix0.row(m);
#pragma unroll
                for (int m = 0; m < G; m++) pa1[m] = wload_a<SEI_WAVE_ABL>(X, r0[m], CIN, 16 + 4 * g);
                const float *wn = wl + ((size_t)it0.k * CIN + 16) * ldy;
#pragma unroll
                for (int e = 0; e < 4; e++) wload_b<NT>(wn + (size_t)e * ldy, r16, pb1[e]);
#pragma unroll
                for (int m = 0; m < G; m++) a0[m] = wload_a<SEI_WAVE_ABL>(X, r0[m], CIN, 4 * g);   // every sub-chunk slot (wprefetch skipped the missing ones)
                if constexpr (PACKED && NT == 2) {   // (the loads above read the packed copy as if it were natural: overwritten here)
                    wload_b_packed(W, 2, it0.k, 0, lane, b0);
                    wload_b_packed(W, 2, it0.k, 1, lane, pb1);
                }
            }
            bool flip = false;
            while (it0.k < K) {
                const int kn = it1.k < K ? it1.k : it0.k;
                int rc[G], rn[G];
#pragma unroll
                for (int m = 0; m < G; m++) {
                    rc[m] = ix0.row(m);
                    rn[m] = ix1.row(m);
                }
#define WGROUP2(MS_)                                                                                                         \
                do {                                                                                                         \
                    if (!flip) wgroup2<NT, G, (MS_ <= G ? MS_ : 1), PACKED>(X, wl, ldy, CIN, kn, rc, rn, ix0.d4, ix0.src, acc_tile, g, r16, \
                                                                     a0, b0, pa1, pb1, qa0, qb0, qa1, qb1, st, W); \
                    else wgroup2<NT, G, (MS_ <= G ? MS_ : 1), PACKED>(X, wl, ldy, CIN, kn, rc, rn, ix0.d4, ix0.src, acc_tile, g, r16,  \
                                                               qa0, qb0, qa1, qb1, a0, b0, pa1, pb1, st, W); \
                } while (0)
                if (G == 1 || ix0.cnt == 1) WGROUP2(1);
                else if (G == 2 || ix0.cnt == 2) WGROUP2(2);
                else if (G == 3 || ix0.cnt == 3) WGROUP2(3);
                else WGROUP2(4);
#undef WGROUP2
                flip = !flip;
                it0 = it1; it1 = it2; it2 = advance(it2);
                ix0 = ix1; ix1 = ix2; ix2 = load_idx(it2);
                WSTAMP(4);
            }
          }
        } else
        while (it0.k < K) {
            const int kn = it1.k < K ? it1.k : it0.k;  // weights prefetched after the last slice (unused at the very end)
            int rc[G], rn[G];
#pragma unroll
            for (int m = 0; m < G; m++) {
                rc[m] = ix0.row(m);
                rn[m] = ix1.row(m);
            }
#define WGROUP(MS_) wgroup<NT, G, (MS_ <= G ? MS_ : 1)>(X, wl, ldy, CIN, NS, it0.k, kn, rc, rn, ix1.cnt, ix0.d4, ix0.src, acc_tile, g, r16, a0, b0, st)
            if (G == 1 || ix0.cnt == 1) WGROUP(1);
            else if (G == 2 || ix0.cnt == 2) WGROUP(2);
            else if (G == 3 || ix0.cnt == 3) WGROUP(3);
            else WGROUP(4);
#undef WGROUP
            it0 = it1; it1 = it2; it2 = advance(it2);
            ix0 = ix1; ix1 = ix2; ix2 = load_idx(it2);  // two groups ahead: its latency is never waited for
            WSTAMP(4);
        }
    }
    __syncthreads();
    // epilogue: every dst row is written once, as whole float4 row segments
    constexpr int V = NC >> 2;
    const int64_t row0 = (int64_t)tile * TR;
    if (ep.mean) {  // inference: eval-mode BatchNorm (+ residual) + LeakyReLU on the way out (common.h)
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
        // BatchNorm statistics of the output from the LDS tile (see conv_fwd.hip): [tile * PARTS + part][2][ldy] doubles
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
    WSTAMP(5);
    WSTAMP_ADD(7, 1);
    st.flush();
}

#ifdef SEI_EXPERIMENTS
// Kernel-experiment builds only (make EXTRA=-DSEI_EXPERIMENTS; tools/sweep_wave.py): SEI_WAVE_PLAN=
// "nOut:NT,TR,G[;nOut:NT,TR,G...]" overrides the default tiling of the layers with that nOut.
static bool env_wave_plan(int nOut, int *nt, int *tr, int *g) {
    const char *e = getenv("SEI_WAVE_PLAN");
    while (e && *e) {
        int no = 0, a = 0, b = 0, c = 0;
        if (sscanf(e, "%d:%d,%d,%d", &no, &a, &b, &c) == 4 && no == nOut) {
            *nt = a; *tr = b; *g = c;
            return true;
        }
        e = strchr(e, ';');
        if (e) e++;
    }
    return false;
}

static bool wave_instantiated(int nt, int g) { return nt >= 1 && nt <= 8 && (g == 2 || (g == 1 && nt == 2) || (g == 4 && nt <= 6) || (g == 3 && nt <= 8)); }
#endif

// Default use: the 32- and 64-column layers (levels 0-1 of the DUNE network).  There the workgroup-tile kernel's
// phases hold too few sub-chunks to keep four lock-stepped waves busy (structural efficiency 0.24 / 0.55) and the
// wave kernel measured +47 % / +10 % (profiles/r02_conv_wave_sweep1.log).  From 96 columns on, the B operands of
// this kernel (4/MS bytes per lane and MFMA through the vector-memory path, MS ~ 1.7) saturate that path --
// profiles/r02_conv_wave_diag_L2_C96.txt: half of a wave's time goes to ISSUING loads -- and the tile kernel, whose
// weights are shared through LDS, wins.  sei_debug_force_conv_kernel() pins one kernel (bit-identity tests, sweeps).
bool plan_wave(int nIn, int nOut, int K, int64_t n_dst, ConvPlan *p) {
    if (nIn % 16 || nOut % 16 || nIn < 16 || nOut > 256 || K + 1 > 64) return false;
    const int kind = forced_conv_kernel();
    if (kind != CONV_KERNEL_AUTO && kind != CONV_KERNEL_WAVE) return false;
    const bool forced = kind == CONV_KERNEL_WAVE;
    const int tiles = nOut / 16;
    int nt = 0;
    for (int c = 8; c >= 1; c--)
        if (tiles % c == 0) { nt = c; break; }
    if (nt == 1 && tiles > 1) return false;  // 11 / 13 column tiles: the workgroup-tile kernel
    int tr = 64, g = 2;
#ifdef SEI_WAVE32_PLAN_R4   // experiment builds: round 2-4's plan for 32 columns
    if (nOut == 32) { tr = 128; g = 2; }
#else
    // [r5] 32 columns: 64-row tiles, one sub-chunk per group.  Almost every (tile, offset) of level 0 holds ONE sub-chunk; a group of
    // two then spends a third of its vector-memory instructions on the missing one (every request of the group-ahead form is
    // unconditional), and 8 KB tiles at 120 registers put 16 waves on a CU instead of 10-12: 0.703 vs 0.754 ms
    // (profiles/r05_wave_packed_weights_ab.log)
    if (nOut == 32) { tr = 64; g = 1; }
#endif
    if (nOut == 64) { tr = 64; g = 3; }
    bool env = false;
#ifdef SEI_EXPERIMENTS
    int en, et, eg;
    env = env_wave_plan(nOut, &en, &et, &eg) && en >= 1 && tiles % en == 0 &&
          (et == 64 || et == 128 || et == 192 || et == 256) && wave_instantiated(en, eg);
    if (env) { nt = en; tr = et; g = eg; }
#endif
    if (!forced && !env && nOut != 32 && nOut != 64) return false;
    if (!forced && !env && conv_rounds_64(nIn, nOut, K, n_dst)) return false;   // [r5] sixteen-wave multi-offset rounds (conv_fwd.hip)
    if ((size_t)tr * nt * 16 * sizeof(float) > 64 * 1024) return false;
    p->kind = 2; p->tile_rows = tr; p->col_group = 0; p->blocks = tiles / nt; p->nt = nt; p->g = g;
#ifndef SEI_WAVE_NO_PACKED_W
    if (GROUP_AHEAD && nt == 2 && g <= 2 && nIn == 32 && nOut == 32) p->col_group = -2;   // [r5] packed B operands (wload_b_packed)
#endif
    p->lds_bytes = (size_t)tr * nt * 16 * sizeof(float);
    return true;
}

template <int NT, int G>
static int launch_wave(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                       int nIn, int nOut, const float *init, double *stats, hipStream_t st, const ConvEpilogue &ep) {
    const int nwork = t.n_tiles * p.blocks;
    if constexpr (NT == 2 && G <= 2) {
        if (p.col_group == -2) {   // [r5] 32 x 32 planes: B operands from the packed copy (plan_wave)
            conv_wave_mfma_kernel<NT, G, true><<<((nwork + 7) / 8) * 8, 64, p.lds_bytes, st>>>(
                X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init, nOut,
                p.blocks, t.tile_rows, stats, ep);
            SEI_LAUNCH_CHECK();
            return 0;
        }
    }
    if (p.col_group == -2) SEI_FAIL("conv_forward_wave: packed weights without the packed kernel");
    conv_wave_mfma_kernel<NT, G><<<((nwork + 7) / 8) * 8, 64, p.lds_bytes, st>>>(
        X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init, nOut,
        p.blocks, t.tile_rows, stats, ep);
    SEI_LAUNCH_CHECK();
    return 0;
}

#define WAVE_CASE(NT_, G_) \
    if (p.nt == NT_ && p.g == G_) return launch_wave<NT_, G_>(t, p, X, W, bias, Y, nIn, nOut, init, stats, st, ep);

int conv_forward_wave(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                      int nIn, int nOut, const float *init, double *stats, hipStream_t st, const ConvEpilogue &ep) {
    // defaults (32 / 64 columns) and the forced-kernel plans (any column count, groups of 2)
    WAVE_CASE(1, 2) WAVE_CASE(2, 2) WAVE_CASE(3, 2) WAVE_CASE(4, 2) WAVE_CASE(5, 2) WAVE_CASE(6, 2) WAVE_CASE(7, 2) WAVE_CASE(8, 2)
    WAVE_CASE(4, 3) WAVE_CASE(2, 1)
#ifdef SEI_EXPERIMENTS  // alternates reachable through SEI_WAVE_PLAN
    WAVE_CASE(1, 3) WAVE_CASE(2, 3) WAVE_CASE(3, 3) WAVE_CASE(5, 3) WAVE_CASE(6, 3) WAVE_CASE(7, 3) WAVE_CASE(8, 3)
    WAVE_CASE(1, 4) WAVE_CASE(2, 4) WAVE_CASE(3, 4) WAVE_CASE(4, 4) WAVE_CASE(5, 4) WAVE_CASE(6, 4)
#endif
    SEI_FAIL("conv_forward: no wave-kernel instantiation for this plan");
}

}  // namespace sei

#ifdef SEI_STAMPS
extern "C" int sei_debug_wave_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sei::g_wstamps), sizeof(sei::g_wstamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_wstamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
