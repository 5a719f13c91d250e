// Sparse convolution forward / dX for gfx950, workgroup-tile kernel with UNIT dealing (ConvPlan kind 3).
//
// Same output-stationary scheme as conv_fwd.hip -- a 256-thread workgroup owns TR dst rows whose accumulators live in
// LDS, walks the filter offsets, streams W[k] through LDS in slices of 16/32 input planes (LDS-DMA, double buffered,
// one barrier per slice), gathers src rows into MFMA A registers one phase ahead and adds each offset's products into
// the LDS tile by a plain read-modify-write -- but the work of a phase is dealt differently.  conv_fwd.hip deals WHOLE
// sub-chunks (16 rules x all column tiles) round-robin to the four waves; a DUNE tile holds 1-3 sub-chunks per offset,
// so the barrier waits for one wave doing a whole sub-chunk while others idle (per-phase critical path 9.3 column-tile
// units at 96 planes / 128-row tiles where 6.8 would be perfect: tools/group_stats.py).  Here the unit of work is one
// (sub-chunk, 16-column tile): full rounds of four sub-chunks are still dealt whole, the remaining r < 4 sub-chunks are
// cut into r*NT units that are dealt as four contiguous runs, so every wave carries ceil(r*NT/4) units (7.05 on the
// same tiles).  A wave's run touches at most two sub-chunks; with the whole sub-chunks of full rounds that makes S
// "segments" (sub-chunk, first tile, tile count) per wave and offset: S = 2 for 64-row tiles, 3 for 128-row tiles.
//
// B operands: the run's first column tile is only known at run time, so the weight slice is stored as MFMA B fragments
// -- [step][column tile][lane] floats (conv_transform_weights, col_group < 0) -- and every (step, tile) fragment is one
// conflict-free ds_read_b32 at an immediate offset from a per-segment base.
// Summation order per dst value: offsets in order, inside an offset the input planes in order -- the same as
// conv_fwd.hip (bit-identical results), run-to-run deterministic, no atomics.
#include "common.h"
#include "conv_tile.h"

#include <stdlib.h>
#include <string.h>

namespace sei {

// one (k, slice) phase of one segment: CNT column tiles starting at the tile `wl` points to.  B values are read D
// steps ahead of the MFMAs that consume them (short segments have few MFMAs per step to cover an LDS latency).
template <int NT, int SLJ, int CNT>
__device__ __forceinline__ void seg_phase(const float *__restrict__ wl, const float4 (&a)[SLJ], f32x4 (&acc)[NT]) {
    constexpr int STEPS = 4 * SLJ;
    constexpr int D0 = CNT >= 4 ? 1 : (CNT == 3 ? 2 : (CNT == 2 ? 3 : 7));
    constexpr int D = D0 < STEPS - 1 ? D0 : STEPS - 1;
    float b[D + 1][CNT];
#pragma unroll
    for (int p = 0; p < D; p++)
#pragma unroll
        for (int n = 0; n < CNT; n++) b[p][n] = wl[(p * NT + n) * 64];
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
        if (st + D < STEPS) {
#pragma unroll
            for (int n = 0; n < CNT; n++) b[(st + D) % (D + 1)][n] = wl[((st + D) * NT + n) * 64];
        }
        // the LDS reads above are ISSUED before this step's MFMAs (hipcc otherwise sinks them behind the MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        const int jj = st >> 2, e = st & 3;
        const float av = e == 0 ? a[jj].x : e == 1 ? a[jj].y : e == 2 ? a[jj].z : a[jj].w;
#pragma unroll
        for (int n = 0; n < CNT; n++) acc[n] = mfma16(av, b[st % (D + 1)][n], acc[n]);
    }
}

template <int NT, int SLJ>
__device__ __forceinline__ void seg_dispatch(int cnt, const float *__restrict__ wl, const float4 (&a)[SLJ], f32x4 (&acc)[NT]) {
    switch (cnt) {  // wave-uniform
        case 1: seg_phase<NT, SLJ, 1>(wl, a, acc); break;
        case 2: if (NT >= 2) seg_phase<NT, SLJ, (NT >= 2 ? 2 : 1)>(wl, a, acc); break;
        case 3: if (NT >= 3) seg_phase<NT, SLJ, (NT >= 3 ? 3 : 1)>(wl, a, acc); break;
        case 4: if (NT >= 4) seg_phase<NT, SLJ, (NT >= 4 ? 4 : 1)>(wl, a, acc); break;
        case 5: if (NT >= 5) seg_phase<NT, SLJ, (NT >= 5 ? 5 : 1)>(wl, a, acc); break;
        case 6: if (NT >= 6) seg_phase<NT, SLJ, (NT >= 6 ? 6 : 1)>(wl, a, acc); break;
        default: break;
    }
}

template <int S>
struct SegIdx {
    int src[S];       // lane r16: src row of rule slot r16 of the segment's sub-chunk (RAW load: -1 = padding)
    uint32_t d4[S];   // lane group g: packed tile-local dst rows of result rows 4g..4g+3
    int t0[S];        // wave-uniform: first column tile of the segment
    int cnt[S];       // wave-uniform: column tiles of the segment (0 = segment unused for this offset)
};

// NT = 16-column tiles of the workgroup's column block (COUT = 16 NT), SLJ = slice / 16, S = segments per wave:
// 2 -> 64-row tiles (<= 4 sub-chunks per offset), 3 -> 128-row tiles (<= 8).
// LDS: acc[TR][COUT] | wbuf[2][SL][COUT] | seg[K+1]
template <int NT, int SLJ, int S>
__global__ __launch_bounds__(256, (S * NT * 4 + 4 * S * SLJ * 4 + NT * 8 + 70 <= 165 ? 3 : 2)) void conv_tile_units_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int CIN, int n_tiles,
    const float *__restrict__ init, int ldy, int blocks, double *__restrict__ stats) {
    extern __shared__ float lds[];
    constexpr int TR = S == 2 ? 64 : 128;
    constexpr int SL = 16 * SLJ;
    constexpr int COUT = NT * 16;
    constexpr int PIECES = SL * COUT / 256;  // 1-KiB LDS-DMA pieces per weight slice (contiguous in W)
    // every XCD gets one contiguous run of (tile, column block) pairs: neighbouring tiles gather many of the same src
    // rows and the blocks of a tile gather exactly the same ones -- they share that XCD's L2 (speed only)
    const int per_xcd = (n_tiles * blocks + 7) >> 3;
    const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (vid >= n_tiles * blocks) return;
    const int tile = vid / blocks, cblk = vid - tile * blocks;
    W += (size_t)cblk * K * CIN * COUT;
    if (bias) bias += cblk * COUT;
    const int ycol0 = cblk * COUT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    float *acc_tile = lds;
    float *wbuf = lds + (size_t)TR * COUT;
    int *seg = reinterpret_cast<int *>(wbuf + 2 * SL * COUT);

    for (int i = threadIdx.x; i <= K; i += 256) seg[i] = tile_seg[(size_t)tile * (K + 1) + i];
    tile_init<TR, COUT>(acc_tile, bias, init, (int64_t)tile * TR, n_dst, ldy, ycol0);
    __syncthreads();

    const int64_t sub0 = tile_sub_start[tile];
    const int NS = CIN / SL;  // slices per offset

    auto next_k = [&](int k) {  // next offset with rules in this tile (K if none)
        if (k < K) k++;
        while (k < K && seg[k + 1] == seg[k]) k++;
        return __builtin_amdgcn_readfirstlane(k);
    };
    // the wave's segments of offset k and their rule indices (clamped: k == K re-reads the last offset's slots, unused)
    auto load_idx = [&](int k) {
        SegIdx<S> r;
        const int kk = k < K ? k : K - 1;
        const int s0 = __builtin_amdgcn_readfirstlane(seg[kk]);
        int ns = __builtin_amdgcn_readfirstlane(seg[kk + 1]) - s0;
        if (k >= K) ns = 0;
        const int q = ns >> 2, rem = ns & 3;
        // the r*NT units of the remainder sub-chunks as four contiguous runs
        const int T = rem * NT, per = (T + 3) >> 2;
        // the deal rotates with the offset (wave w of every workgroup sits on SIMD w: a fixed deal would always give
        // SIMD 3 the short last run)
        const int rw = (wave + kk) & 3;
        int lo = rw * per;
        lo = lo < T ? lo : T;
        int hi = lo + per;
        hi = hi < T ? hi : T;
        const int len = hi - lo, sA = lo / NT, tA = lo - sA * NT;
        const int cA = (NT - tA) < len ? (NT - tA) : len, cB = len - cA;
        int sc[S];
        if (S == 2) {
            // 64-row tiles: ns <= 4, a full round leaves no remainder
            sc[0] = q ? rw : sA;         r.t0[0] = q ? 0 : tA; r.cnt[0] = q ? NT : cA;
            sc[1] = q ? 0 : sA + 1;      r.t0[1] = 0;          r.cnt[1] = q ? 0 : cB;
        } else {
            // 128-row tiles: ns <= 8; two full rounds leave no remainder
            sc[0] = q ? rw : sA;                              r.t0[0] = q ? 0 : tA;         r.cnt[0] = q ? NT : cA;
            sc[1] = q >= 2 ? rw + 4 : (q ? 4 + sA : sA + 1);   r.t0[1] = q == 1 ? tA : 0;    r.cnt[1] = q >= 2 ? NT : (q ? cA : cB);
            sc[S - 1] = q == 1 ? 4 + sA + 1 : 0;              r.t0[S - 1] = 0;              r.cnt[S - 1] = q == 1 ? cB : 0;
        }
#pragma unroll
        for (int m = 0; m < S; m++) {
            r.t0[m] = __builtin_amdgcn_readfirstlane(r.t0[m]);
            r.cnt[m] = __builtin_amdgcn_readfirstlane(r.cnt[m]);
            // unused segments read sub-chunk 0 of the offset (always mapped); the values are never consumed.
            // The loaded values are NOT touched here: selecting on them would stall this wave for a full memory
            // latency per offset; validity is applied where they are consumed (2 offsets later)
            const int64_t sb = (sub0 + s0 + (r.cnt[m] > 0 ? sc[m] : 0)) * SUB_SLOTS;
            r.src[m] = sub_src[sb + r16];
            r.d4[m] = *reinterpret_cast<const uint32_t *>(sub_dst + sb + 4 * g);
        }
        return r;
    };

    int k = next_k(-1);
    if (k < K) {
        int k1 = next_k(k), k2 = next_k(k1);
        SegIdx<S> ic = load_idx(k), in1 = load_idx(k1), in2 = load_idx(k2);
        stage_slice<PIECES>(W + (size_t)k * CIN * COUT, wbuf, wave, lane);
        float4 a_c[S][SLJ], a_n[S][SLJ];
#pragma unroll
        for (int m = 0; m < S; m++)
#pragma unroll
            for (int jj = 0; jj < SLJ; jj++) {
                a_c[m][jj] = load_a(X, ic.cnt[m] > 0 ? ic.src[m] : -1, CIN, 16 * jj + 4 * g);
                a_n[m][jj] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        int phase = 0;
        while (k < K) {
            const int kw = k1 < K ? k1 : k;  // weights to prefetch after the last slice (unused at the very end)
            f32x4 acc[S][NT];
            for (int s = 0; s < NS; s++) {
                // Barrier: this phase's weight slice (LDS-DMA issued one phase ago) has landed in every wave and the
                // previous phase is fully consumed.  The 2*S index loads of offset k+2 issued at the end of the
                // previous offset are the youngest in flight (vmcnt retires in order): leave exactly them outstanding.
                if (s == 0 && phase > 0)
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * S) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (s == 0) {
#pragma unroll
                    for (int m = 0; m < S; m++)
#pragma unroll
                        for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // prefetch the next phase's weight slice and A fragments while this one computes
                const bool same_k = s + 1 < NS;
                stage_slice<PIECES>(W + ((size_t)(same_k ? k : kw) * CIN + (size_t)(same_k ? s + 1 : 0) * SL) * COUT,
                                    wbuf + (size_t)((phase + 1) & 1) * SL * COUT, wave, lane);
                const int aoff = (same_k ? (s + 1) * SL : 0) + 4 * g;
#pragma unroll
                for (int m = 0; m < S; m++) {
                    const int row = same_k ? (ic.cnt[m] > 0 ? ic.src[m] : -1) : (in1.cnt[m] > 0 ? in1.src[m] : -1);
#pragma unroll
                    for (int jj = 0; jj < SLJ; jj++) a_n[m][jj] = load_a(X, row, CIN, aoff + 16 * jj);
                }
                const float *wb = wbuf + (size_t)(phase & 1) * SL * COUT + lane;
#pragma unroll
                for (int m = 0; m < S; m++)
                    if (ic.cnt[m] > 0) seg_dispatch<NT, SLJ>(ic.cnt[m], wb + ic.t0[m] * 64, a_c[m], acc[m]);
#pragma unroll
                for (int m = 0; m < S; m++)
#pragma unroll
                    for (int jj = 0; jj < SLJ; jj++) a_c[m][jj] = a_n[m][jj];
                phase++;
            }
            // D[row = 4g+i][col = r16] of column tile t0+n belongs to rule slot 4g+i.  Rows of one offset are distinct,
            // two segments of one sub-chunk own disjoint column tiles, and the previous offset's updates are behind
            // the barrier:  tile = tile + partial  is a plain LDS read-modify-write (no atomics, fixed summation order).
#pragma unroll
            for (int m = 0; m < S; m++) {
                if (ic.cnt[m] > 0) {  // wave-uniform
                    const int nv = __popcll(__ballot(ic.src[m] >= 0) & 0xFFFFull);  // padding sits at the end: slots [0, nv) are rules
                    float *prow[4];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        prow[i] = acc_tile + ((ic.d4[m] >> (8 * i)) & 0xFF) * COUT + ic.t0[m] * 16 + r16;
                    f32x4 tv[NT];
#pragma unroll
                    for (int n = 0; n < NT; n++)
                        if (n < ic.cnt[m]) {
#pragma unroll
                            for (int i = 0; i < 4; i++) tv[n][i] = prow[i][16 * n];
                        }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (4 * g + i < nv) {
#pragma unroll
                            for (int n = 0; n < NT; n++)
                                if (n < ic.cnt[m]) prow[i][16 * n] = tv[n][i] + acc[m][n][i];
                        }
                    }
                }
            }
            ic = in1;
            in1 = in2;
            k = k1;
            k1 = k2;
            k2 = next_k(k2);
            in2 = load_idx(k2);  // two offsets ahead: its latency never meets a barrier
        }
    }
    __syncthreads();
    const int64_t row0 = (int64_t)tile * TR;
    tile_store<TR, COUT>(acc_tile, Y, row0, n_dst, ldy, ycol0);
    if (stats) tile_stats<TR, COUT>(acc_tile, stats, tile, row0, n_dst, ldy, ycol0);
}

static size_t units_lds(int tile_rows, int cout, int slj, int K) {
    return ((size_t)tile_rows * cout + 2 * 16 * slj * cout) * sizeof(float) + (K + 1) * sizeof(int);
}

// SEI_UNITS_PLAN="nOut:NT,TR[;...]" overrides the column block width (NT tiles) and the tile height (experiments)
static bool env_units_plan(int nOut, int *nt, int *tr) {
    const char *e = getenv("SEI_UNITS_PLAN");
    while (e && *e) {
        int no = 0;
        if (sscanf(e, "%d:%d,%d", &no, nt, tr) == 3 && no == nOut) return true;
        e = strchr(e, ';');
        if (e) e++;
    }
    return false;
}

// SEI_CONV_KERNEL=units forces this kernel wherever it is instantiated (experiments)
bool plan_units(int nIn, int nOut, int K, int64_t n_dst, ConvPlan *p) {
    if (nIn % 16 || nOut % 16 || nIn < 16 || nOut > 256) return false;
    const char *kind = getenv("SEI_CONV_KERNEL");
    if (kind && (!strcmp(kind, "tile") || !strcmp(kind, "wave"))) return false;
    const bool forced = kind && !strcmp(kind, "units");
    const int tiles = nOut / 16;
    int nt = 0;
    for (int c = 6; c >= 1; c--)
        if (tiles % c == 0) { nt = c; break; }
    if (tiles > 6 && nt > 4) {  // wide layers: column blocks of <= 4 tiles keep three workgroups on a CU
        for (int c = 4; c >= 2; c--)
            if (tiles % c == 0) { nt = c; break; }
    }
    int tr = 128;
    int en, et;
    const bool env = env_units_plan(nOut, &en, &et) && en >= 1 && en <= 6 && tiles % en == 0 && (et == 64 || et == 128);
    if (env) { nt = en; tr = et; }
    if (!forced && !env) return false;
    if (nt == 1 && tiles > 1 && !env) return false;  // 7 / 11 / 13 column tiles: conv_fwd.hip
    p->kind = 3; p->tile_rows = tr; p->blocks = tiles / nt; p->nt = nt; p->cs = 1; p->maxs = tr / 64;
    p->slj = (nIn % 32 == 0) ? 2 : 1;
    p->col_group = -16 * p->slj;  // B-fragment layout, slices of 16*slj planes
    p->lds_bytes = units_lds(tr, nt * 16, p->slj, K);
    return p->lds_bytes <= 160 * 1024;
}

template <int NT, int SLJ, int S>
static int launch_units(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                        int nIn, int nOut, const float *init, double *stats, hipStream_t st) {
    SEI_TRY(ensure_max_lds((const void *)conv_tile_units_kernel<NT, SLJ, S>, 160 * 1024));
    conv_tile_units_kernel<NT, SLJ, S><<<((t.n_tiles * p.blocks + 7) / 8) * 8, 256, p.lds_bytes, st>>>(
        X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init, nOut,
        p.blocks, stats);
    SEI_LAUNCH_CHECK();
    return 0;
}

#define UNITS_CASE(NT_)                                                                                           \
    if (p.nt == NT_) {                                                                                            \
        if (p.slj == 2 && p.tile_rows == 128) return launch_units<NT_, 2, 3>(t, p, X, W, bias, Y, nIn, nOut, init, stats, st); \
        if (p.slj == 1 && p.tile_rows == 128) return launch_units<NT_, 1, 3>(t, p, X, W, bias, Y, nIn, nOut, init, stats, st); \
        if (p.slj == 2 && p.tile_rows == 64) return launch_units<NT_, 2, 2>(t, p, X, W, bias, Y, nIn, nOut, init, stats, st);  \
        if (p.slj == 1 && p.tile_rows == 64) return launch_units<NT_, 1, 2>(t, p, X, W, bias, Y, nIn, nOut, init, stats, st);  \
    }

int conv_forward_units(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                       int nIn, int nOut, const float *init, double *stats, hipStream_t st) {
    UNITS_CASE(1) UNITS_CASE(2) UNITS_CASE(3) UNITS_CASE(4) UNITS_CASE(5) UNITS_CASE(6)
    SEI_FAIL("conv_forward: no unit-dealing instantiation for this plan");
}

}  // namespace sei
