// Sparse convolution forward / dX for gfx950, row-band form (scn.SubmanifoldConvolution / scn.Convolution:
// resnet.py:30-50,105-110; sparse_building_blocks.py:29-34,110-117) -- the 96- to 192-plane layers.
//
// The two older kernels each pay for one thing.  The workgroup-tile kernel (conv_fwd.hip) shares W[k] through LDS but
// deals the sub-chunks of ONE tile to its waves, so every 16-plane slice ends in an s_barrier and a wave spends half
// its time waiting for the slowest one (profiles/r02_conv_tile_diag_L2_C96.txt).  The wave-private kernel
// (conv_wave.hip) has no barrier because each wave owns its dst rows, but every wave pulls its own copy of the B
// operands through the vector-memory path, which saturates from 96 planes on.  This kernel takes one half of each:
//   * ROW BANDS: compute wave w of a workgroup owns 64 consecutive dst rows (a 64-row tile of the rulebook) x NC = 16*NT
//     columns; their fp32 accumulators live in its own part of LDS and its partial sums stay in its registers, so the
//     per-offset read-modify-write never crosses waves and nothing orders the compute waves against each other;
//   * WEIGHT RING: the NCW compute waves walk the same sequence of weight slices (offset k ascending, 16 input planes
//     per slice) -- a ring of D slices in LDS, filled by ONE loader wave with LDS-DMA (global_load_lds_dwordx4, per-lane
//     source address, W in its natural layout).  There is no barrier in the main loop: the loader publishes a landed
//     slice by storing its sequence number to an LDS word, a compute wave polls that word (s_sleep between polls) only
//     when it has caught up, reads its B operands with ds_read_b128 and stores the number of slices it has released
//     to a word of its own; the loader refills a slot when every compute wave has released it.  A slow wave delays the
//     others only when the ring is exhausted; offsets in which a band holds no rule are released without being read.
// Per dst row the summation order is the one of the other two kernels (offsets ascending; inside an offset one MFMA
// chain over the input planes from zero, then tile += partial): bit-identical results, run-to-run deterministic, no
// atomics; every dst row is written to HBM exactly once.
#include "common.h"
#include "conv_tile.h"
#include "conv_rows.h"

#include <limits.h>

namespace sei {

constexpr int BAND_ROWS = 64;            // dst rows of one compute wave = rulebook tile_rows
constexpr int BAND_MAXS = BAND_ROWS / SUB_SLOTS;  // sub-chunks one offset can hold in a band
constexpr int BAND_LAG = 2;              // weight slices the loader keeps in flight before it publishes the older one
constexpr int BAND_SYNC_WORDS = 32;      // [0] slices landed, [8 + w] slices released by compute wave w
constexpr int BAND_SPIN_CAP = 1 << 22;   // every poll loop is bounded: a lost hand-off raises g_band_timeouts, never hangs

__device__ int g_band_timeouts;

#ifdef SEI_STAMPS
__device__ unsigned long long g_bstamps[16];
#define BSTAMP_DECL unsigned long long bs_t = __builtin_amdgcn_s_memtime(), bs_acc[16] = {}
#define BSTAMP(i) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); bs_acc[i] += n_ - bs_t; bs_t = n_; } while (0)
#define BSTAMP_ADD(i, v) bs_acc[i] += (v)
#define BSTAMP_FLUSH do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_bstamps[i_], bs_acc[i_]); } } while (0)
#else
#define BSTAMP_DECL
#define BSTAMP(i)
#define BSTAMP_ADD(i, v)
#define BSTAMP_FLUSH
#endif

struct BandIdx {
    int src[BAND_MAXS];        // lane r16: src row of rule slot r16 of sub-chunk m (raw load; validity applied by row())
    uint32_t d4[BAND_MAXS];    // lane group g: packed band-local dst rows of slots 4g..4g+3
    int ms;                    // sub-chunks of the offset in this band (wave-uniform, 0 past the end)
    __device__ __forceinline__ int row(int m) const { return m < ms ? src[m] : -1; }
};

// One slice (16 input planes = 4 MFMA steps) for MS sub-chunks: B operands from the ring slot `wb` ([16][NC] floats,
// weight rows in their natural layout), read one step ahead of the MFMAs that consume them.
template <int NT, int MS>
__device__ __forceinline__ void band_slice(const float *wb, const float4 (&a)[BAND_MAXS], const int (&rows)[BAND_MAXS],
                                           f32x4 (&acc)[MS][NT], int g, int r16) {
    constexpr int NC = 16 * NT;
    float b[2][NT];
    wload_b<NT>(wb + (4 * g) * NC, r16, b[0]);
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (e + 1 < 4) wload_b<NT>(wb + (4 * g + e + 1) * NC, r16, b[(e + 1) & 1]);
        // the next step's LDS reads are ISSUED before this step's MFMAs (hipcc otherwise sinks them behind the MFMAs)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MS; m++) {
            float av = e == 0 ? a[m].x : e == 1 ? a[m].y : e == 2 ? a[m].z : a[m].w;
            av = rows[m] >= 0 ? av : 0.f;  // padding slots feed zeros (see conv_fwd.hip: the chip clocks higher on them)
#pragma unroll
            for (int n = 0; n < NT; n++) acc[m][n] = mfma16(av, b[e & 1][n], acc[m][n]);
        }
    }
}

struct BandCtx {
    const float *X;
    const float *ring;          // LDS
    volatile int *sync;         // LDS
    float *acc_tile;            // LDS, this wave's band
    int CIN, NS, D, wave, lane, g, r16;
    int known;                  // slices known to have landed (wave-uniform)
};

// waits until slice i of the weight stream has landed in its ring slot
__device__ __forceinline__ void band_wait_filled(BandCtx &c, int i) {
    if (i < c.known) return;  // wave-uniform
    int spins = 0;
    for (;;) {
        c.known = __builtin_amdgcn_readfirstlane(c.sync[0]);
        if (i < c.known) return;
        if (++spins > BAND_SPIN_CAP) {
            if (c.lane == 0) atomicAdd(&g_band_timeouts, 1);
            c.known = INT_MAX;
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// this wave has no further use for slices [0, n): every LDS read of them has been issued before this store, and a
// wave's LDS operations execute in order
__device__ __forceinline__ void band_release(BandCtx &c, int n) {
    asm volatile("" ::: "memory");
    if (c.lane == 0) c.sync[8 + c.wave] = n;
    asm volatile("" ::: "memory");
}

// One offset = MS sub-chunks of this band: NS slices from the ring starting at slice number ib, then the
// read-modify-write of the band rows.  On entry a_c holds the A fragments of slice 0 of this offset; on exit those of
// slice 0 of the next offset (rows rn, msn sub-chunks).
template <int NT, int MS>
__device__ __forceinline__ void band_offset(BandCtx &c, int ib, const int (&rc)[BAND_MAXS], const int (&rn)[BAND_MAXS], int msn,
                                            const uint32_t (&d4)[BAND_MAXS], const int (&src)[BAND_MAXS],
                                            float4 (&a_c)[BAND_MAXS]) {
    constexpr int NC = 16 * NT;
    f32x4 acc[MS][NT];
#pragma unroll
    for (int m = 0; m < MS; m++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int slot = ib % c.D;
    for (int j = 0; j < c.NS; j++) {
        // A fragments one slice ahead (the next offset's first slice after the last one); only the sub-chunks that exist
        // are gathered (wave-uniform branches)
        const bool same = j + 1 < c.NS;
        const int aoff = (same ? 16 * (j + 1) : 0) + 4 * c.g;
        const int cnt = same ? MS : msn;
        float4 a_n[BAND_MAXS];
#pragma unroll
        for (int m = 0; m < BAND_MAXS; m++)
            if (m == 0 || m < cnt) a_n[m] = wload_a<0>(c.X, same ? rc[m] : rn[m], c.CIN, aoff);
        band_wait_filled(c, ib + j);
        BSTAMP(1);
        band_slice<NT, MS>(c.ring + (size_t)slot * 16 * NC, a_c, rc, acc, c.g, c.r16);
        band_release(c, ib + j + 1);
#pragma unroll
        for (int m = 0; m < BAND_MAXS; m++) a_c[m] = a_n[m];
        slot = slot + 1 == c.D ? 0 : slot + 1;
    }
#pragma unroll
    for (int m = 0; m < MS; m++) {
        const int nv = __popcll(__ballot(src[m] >= 0) & 0xFFFFull);
        wrmw<NT>(c.acc_tile, NC, d4[m], nv, c.g, c.r16, acc[m]);
    }
}

template <int NT, int NCW>
__global__ __launch_bounds__(64 * (NCW + 1)) void conv_band_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int CIN, int n_tiles,
    const float *__restrict__ init, int ldy, int blocks, int D, double *__restrict__ stats) {
    extern __shared__ float lds[];
    constexpr int NC = 16 * NT;  // columns of the workgroup = row stride of bands and ring slots
    // work item = (group of NCW consecutive 64-row tiles, column block) in the XCD-aware order of xcd_item (conv_tile.h)
    const int ngroups = (n_tiles + NCW - 1) / NCW;
    const int nwork = ngroups * blocks;
    const int vid = xcd_item(nwork);
    if (vid >= nwork) return;  // whole workgroup
    const int group = vid / blocks, cblk = vid - group * blocks;
    const int col0 = cblk * NC;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    float *ring = lds + (size_t)NCW * BAND_ROWS * NC;
    volatile int *sync = reinterpret_cast<volatile int *>(ring + (size_t)D * 16 * NC);
    const int NS = CIN >> 4;  // slices per offset

    if (threadIdx.x < BAND_SYNC_WORDS) sync[threadIdx.x] = 0;

    // offsets with rules in ANY band of the group = the weight stream of this workgroup; every wave derives it alone
    unsigned long long umask = 0, kmask = 0;
    int segv = 0;
    for (int w2 = 0; w2 < NCW; w2++) {
        const int t2 = group * NCW + w2;
        if (t2 < n_tiles) {
            const int sv = lane <= K ? tile_seg[(size_t)t2 * (K + 1) + lane] : 0;
            const int nsv = __shfl_down(sv, 1) - sv;
            const unsigned long long m = __ballot(lane < K && nsv > 0);
            umask |= m;
            if (w2 == wave) { segv = sv; kmask = m; }
        }
    }
    const int F = __popcll(umask) * NS;  // slices in the stream

    if (wave == NCW) {
        // ------------------------------------------------------------------ loader wave
        __syncthreads();
        // piece p of a slot = floats [256 p, 256 p + 256) of its [16][NC] image; lane l moves 4 floats of one weight row
        int off[NT];
#pragma unroll
        for (int p = 0; p < NT; p++) {
            const int idx = 256 * p + 4 * lane, plane = idx / NC;
            off[p] = plane * ldy + (idx - plane * NC);
        }
        int i = 0, slot = 0, known_free = D;  // slices [0, known_free) may be issued
        for (unsigned long long rem = umask; rem; rem &= rem - 1) {
            const int k = __builtin_amdgcn_readfirstlane(__ffsll((long long)rem) - 1);
            const float *wk = W + (size_t)k * CIN * ldy + col0;
            for (int j = 0; j < NS; j++) {
                if (i >= known_free) {  // the slot's previous slice must have been released by every compute wave
                    int spins = 0;
                    for (;;) {
                        int v = lane < NCW ? sync[8 + lane] : INT_MAX;
#pragma unroll
                        for (int d = 1; d < 8; d <<= 1) v = min(v, __shfl_xor(v, d));
                        const int mn = __builtin_amdgcn_readfirstlane(v);
                        known_free = mn > INT_MAX - D ? INT_MAX : mn + D;
                        if (i < known_free) break;
                        if (++spins > BAND_SPIN_CAP) {
                            if (lane == 0) atomicAdd(&g_band_timeouts, 1);
                            known_free = INT_MAX;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                const float *ws = wk + (size_t)(16 * j) * ldy;
                float *dst = ring + (size_t)slot * 16 * NC;
#pragma unroll
                for (int p = 0; p < NT; p++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ws + off[p]),
                                                     (__attribute__((address_space(3))) void *)(dst + 256 * p), 16, 0, 0);
                if (i >= BAND_LAG) {
                    // slice i - LAG has landed once at most LAG slices' pieces are still outstanding
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT * BAND_LAG) : "memory");
                    if (lane == 0) sync[0] = i - BAND_LAG + 1;
                    asm volatile("" ::: "memory");
                }
                i++;
                slot = slot + 1 == D ? 0 : slot + 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) sync[0] = F;
        return;
    }

    // ---------------------------------------------------------------------- compute wave
    const int tile = group * NCW + wave;
    float *acc_tile = lds + (size_t)wave * BAND_ROWS * NC;
    const bool has_tile = tile < n_tiles;
    const int64_t row0 = (int64_t)tile * BAND_ROWS;
    if (has_tile) band_init<NC>(acc_tile, bias, init, row0, BAND_ROWS, n_dst, ldy, col0, lane);
    __syncthreads();  // the sync words are zeroed; orders this wave's band initialisation before its first read-modify-write
    if (!has_tile) {
        if (lane == 0) sync[8 + wave] = INT_MAX;
        return;
    }
    BSTAMP_DECL;
    BandCtx c{X, ring, sync, acc_tile, CIN, NS, D, wave, lane, g, r16, 0};
    const int64_t sub0 = tile_sub_start[tile];

    auto next_k = [&](int k) {  // next offset with rules in this band (64 if none)
        const unsigned long long rem = k + 1 < 64 ? (kmask >> (k + 1)) << (k + 1) : 0ull;
        return __builtin_amdgcn_readfirstlane(rem ? __ffsll((long long)rem) - 1 : 64);
    };
    auto load_idx = [&](int k) {
        BandIdx r;
        const bool ok = k < 64;
        const int s0 = ok ? __builtin_amdgcn_readlane(segv, k) : 0;
        const int ns = ok ? __builtin_amdgcn_readlane(segv, k + 1) - s0 : 0;
        r.ms = __builtin_amdgcn_readfirstlane(ns < BAND_MAXS ? ns : BAND_MAXS);
#pragma unroll
        for (int m = 0; m < BAND_MAXS; m++) {
            const int s = s0 + (m < r.ms ? m : 0);  // clamped: never branch around a load
            const int64_t sb = (sub0 + s) * SUB_SLOTS;
            r.src[m] = sub_src[sb + r16];
            r.d4[m] = *reinterpret_cast<const uint32_t *>(sub_dst + sb + 4 * g);
        }
        return r;
    };

    int k0 = next_k(-1);
    if (k0 < 64) {
        int k1 = next_k(k0), k2 = next_k(k1);
        BandIdx ix0 = load_idx(k0), ix1 = load_idx(k1), ix2 = load_idx(k2);
        float4 a_c[BAND_MAXS];
#pragma unroll
        for (int m = 0; m < BAND_MAXS; m++)
            if (m == 0 || m < ix0.ms) a_c[m] = wload_a<0>(X, ix0.row(m), CIN, 4 * g);
        int released = 0;
        BSTAMP(0);
        while (k0 < 64) {
            // slice number of this offset's first slice in the workgroup's stream
            const int ib = __popcll(umask & ((1ull << k0) - 1ull)) * NS;
            if (ib > released) band_release(c, ib);  // offsets of other bands only: released unread
            int rc[BAND_MAXS], rn[BAND_MAXS];
#pragma unroll
            for (int m = 0; m < BAND_MAXS; m++) {
                rc[m] = ix0.row(m);
                rn[m] = ix1.row(m);
            }
            switch (ix0.ms) {
                case 1: band_offset<NT, 1>(c, ib, rc, rn, ix1.ms, ix0.d4, ix0.src, a_c); break;
                case 2: band_offset<NT, 2>(c, ib, rc, rn, ix1.ms, ix0.d4, ix0.src, a_c); break;
                case 3: band_offset<NT, 3>(c, ib, rc, rn, ix1.ms, ix0.d4, ix0.src, a_c); break;
                default: band_offset<NT, 4>(c, ib, rc, rn, ix1.ms, ix0.d4, ix0.src, a_c); break;
            }
            released = ib + NS;
            BSTAMP(2);
            BSTAMP_ADD(8, ix0.ms * NS);
            k0 = k1; k1 = k2; k2 = next_k(k2);
            ix0 = ix1; ix1 = ix2; ix2 = load_idx(k2);  // two offsets ahead: its latency is never waited for
        }
    }
    band_release(c, INT_MAX);  // nothing else of the stream is needed by this band
    BSTAMP(3);
    band_store<NC>(acc_tile, Y, stats, tile, row0, BAND_ROWS, n_dst, ldy, col0, lane);
    BSTAMP(4);
    BSTAMP_ADD(7, 1);
    BSTAMP_FLUSH;
}

// LDS of one workgroup: NCW bands + D ring slots + sync words
static size_t band_lds(int nt, int ncw, int d) {
    return ((size_t)ncw * BAND_ROWS * 16 * nt + (size_t)d * 16 * 16 * nt) * sizeof(float) + BAND_SYNC_WORDS * sizeof(int);
}

// Default use: the layers from 96 planes on (levels 2-5 of the DUNE network).  sei_debug_force_conv_kernel(3) pins it
// for every MFMA-sized layer (tests).
bool plan_band(int nIn, int nOut, int K, int64_t n_dst, ConvPlan *p) {
    if (nIn % 16 || nOut % 16 || nIn < 16 || nOut > 256 || K + 1 > 64) return false;
    const int kind = forced_conv_kernel();
    if (kind != CONV_KERNEL_AUTO && kind != CONV_KERNEL_BAND) return false;
    const bool forced = kind == CONV_KERNEL_BAND;
    if (!forced && nOut < 96) return false;
    const int tiles = nOut / 16;
    // column tiles per workgroup: the largest divisor of the tile count up to 6 (one column block for <= 96 columns when
    // forced; two or more above): the band accumulators (64 x NC floats per compute wave) and the ring must fit LDS
    int nt = 0;
    for (int cnd = (nOut >= 96 ? 5 : 6); cnd >= 1; cnd--)
        if (tiles % cnd == 0) { nt = cnd; break; }
    if (nOut == 96) nt = 3;
    if (nt == 1 && tiles > 1) return false;  // 7 / 11 / 13 column tiles: the workgroup-tile kernel
    const int ncw = 4;
    // ring depth: what is left of half a CU's LDS (two workgroups per CU), at least LAG + 2, at most 16 slices
    int d = 16;
    while (d > BAND_LAG + 2 && band_lds(nt, ncw, d) > 80 * 1024 - 512) d--;
    if (band_lds(nt, ncw, d) > 160 * 1024) return false;
    p->kind = 3; p->tile_rows = BAND_ROWS; p->col_group = 0; p->blocks = tiles / nt; p->nt = nt; p->nw = ncw; p->g = d;
    p->lds_bytes = band_lds(nt, ncw, d);
    return true;
}

template <int NT, int NCW>
static int launch_band(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                       int nIn, int nOut, const float *init, double *stats, hipStream_t st) {
    SEI_TRY(ensure_max_lds((const void *)conv_band_mfma_kernel<NT, NCW>, 160 * 1024));
    const int ngroups = (t.n_tiles + NCW - 1) / NCW;
    const int nwork = ngroups * p.blocks;
    conv_band_mfma_kernel<NT, NCW><<<((nwork + 7) / 8) * 8, 64 * (NCW + 1), p.lds_bytes, st>>>(
        X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init, nOut,
        p.blocks, p.g, stats);
    SEI_LAUNCH_CHECK();
    return 0;
}

#define BAND_CASE(NT_, NCW_) \
    if (p.nt == NT_ && p.nw == NCW_) return launch_band<NT_, NCW_>(t, p, X, W, bias, Y, nIn, nOut, init, stats, st);

int conv_forward_band(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                      int nIn, int nOut, const float *init, double *stats, hipStream_t st) {
    if (t.tile_rows != BAND_ROWS) SEI_FAIL("conv_forward_band: needs the 64-row rulebook variant");
    if (p.g < BAND_LAG + 2) SEI_FAIL("conv_forward_band: ring too shallow");
    BAND_CASE(1, 4) BAND_CASE(2, 4) BAND_CASE(3, 4) BAND_CASE(4, 4) BAND_CASE(5, 4) BAND_CASE(6, 4)
    SEI_FAIL("conv_forward: no band-kernel instantiation for this plan");
}

}  // namespace sei

extern "C" int sei_debug_band_timeouts(int32_t *out, int32_t reset) {
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(sei::g_band_timeouts), sizeof(int)) != hipSuccess) {
        sei::set_error("sei_debug_band_timeouts: hipMemcpyFromSymbol failed");
        return 1;
    }
    if (out) *out = v;
    if (reset) {
        int z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_band_timeouts), &z, sizeof(int)) != hipSuccess) return 1;
    }
    return 0;
}

#ifdef SEI_STAMPS
extern "C" int sei_debug_band_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sei::g_bstamps), sizeof(sei::g_bstamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_bstamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
