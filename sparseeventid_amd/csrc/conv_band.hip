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
#include <stdlib.h>
#include <string.h>

// The family measured below the workgroup-tile kernel on every level (profiles/r03_conv_band_*.log) and is no plan's
// choice; its LDS hand-offs are polled with BOUNDED loops, and a poll that ran out would carry on with slots that have not
// landed (counted in g_band_timeouts, results wrong).  It therefore exists only in experiment builds
// (make EXTRA=-DSEI_EXPERIMENTS: tools/sweep_band.py, tests/test_gpu_kernels_agree.py); the shipped library does not compile this file (Makefile) and holds stubs in conv_fwd.hip
// at the end of this file.
#ifdef SEI_EXPERIMENTS

namespace sei {

constexpr int BAND_ROWS = 64;            // dst rows of one compute wave = rulebook tile_rows
constexpr int BAND_MAXS = BAND_ROWS / SUB_SLOTS;  // sub-chunks one offset can hold in a band
#ifndef SEI_BAND_LAG
#define SEI_BAND_LAG 4
#endif
constexpr int BAND_LAG = SEI_BAND_LAG;              // weight slices the loader keeps in flight before it publishes the older one
constexpr int BAND_SYNC_WORDS = 32;      // [0] slices landed, [8 + w] slices released by compute wave w
constexpr int BAND_SPIN_CAP = 1 << 22;   // every poll loop is bounded: a lost hand-off raises g_band_timeouts, never hangs

__device__ int g_band_timeouts;

// Diagnostic builds only (make EXTRA="-DSEI_EXPERIMENTS -DSEI_BAND_ABL=n"; tools/band_abl.sh): the bits remove one cost
// at a time (1: gathers hit the caches, 2: nobody waits for the ring, 4: no band read-modify-write, 8: no MFMAs,
// 16: no B reads from the ring, 32: no gathers at all, 64: shared tile: no write-back ordering; results are wrong, timing only).  The product build compiles none of it.
#ifndef SEI_BAND_ABL
#define SEI_BAND_ABL 0
#endif

// Diagnostic build only (make EXTRA="-DSEI_EXPERIMENTS -DSEI_STAMPS"; tools/band_stamps.py): every wave adds the s_memtime
// ticks it spends in each part of the kernel to g_bstamps.  The product build compiles none of it.
#ifdef SEI_STAMPS
__device__ unsigned long long g_bstamps[16];
struct BandStamps {
    unsigned long long t, acc[16];
    __device__ __forceinline__ void start() {
        for (int i = 0; i < 16; i++) acc[i] = 0;
        t = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void mark(int i) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long n = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        acc[i] += n - t;
        t = n;
    }
    __device__ __forceinline__ void flush(int lane) {
        if (lane == 0)
            for (int i = 0; i < 16; i++)
                if (acc[i]) atomicAdd(&g_bstamps[i], acc[i]);
    }
};
#define BSTAMP(s, i) (s).mark(i)
#define BSTAMP_ADD(s, i, v) (s).acc[i] += (v)
#else
struct BandStamps {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void flush(int) {}
};
#define BSTAMP(s, i)
#define BSTAMP_ADD(s, i, v)
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
    if (SEI_BAND_ABL & 16) {
#pragma unroll
        for (int n = 0; n < NT; n++) b[0][n] = b[1][n] = 1.f + n;
    } else {
        wload_b<NT>(wb + (4 * g) * NC, r16, b[0]);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (e + 1 < 4 && !(SEI_BAND_ABL & 16)) wload_b<NT>(wb + (4 * g + e + 1) * NC, r16, b[(e + 1) & 1]);
        // the next step's LDS reads are ISSUED before this step's MFMAs (hipcc otherwise sinks them behind the MFMAs)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MS; m++) {
            float av = e == 0 ? a[m].x : e == 1 ? a[m].y : e == 2 ? a[m].z : a[m].w;
            av = rows[m] >= 0 ? av : 0.f;  // padding slots feed zeros (see conv_fwd.hip: the chip clocks higher on them)
#pragma unroll
            for (int n = 0; n < NT; n++) {
                if (SEI_BAND_ABL & 8) {
                    asm volatile("" ::"v"(av), "v"(b[e & 1][n]));  // keeps the loads alive
                } else {
                    acc[m][n] = mfma16(av, b[e & 1][n], acc[m][n]);
                }
            }
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
    BandStamps st;
};

// waits until slice i of the weight stream has landed in its ring slot
__device__ __forceinline__ void band_wait_filled(BandCtx &c, int i) {
    if (i < c.known || (SEI_BAND_ABL & 2)) return;  // wave-uniform
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
            if (m == 0 || m < cnt) a_n[m] = wload_a<SEI_BAND_ABL>(c.X, same ? rc[m] : rn[m], c.CIN, aoff);
        BSTAMP(c.st, 1);
        band_wait_filled(c, ib + j);
        BSTAMP(c.st, 2);
        band_slice<NT, MS>(c.ring + (size_t)slot * 16 * NC, a_c, rc, acc, c.g, c.r16);
        BSTAMP(c.st, 3);
        band_release(c, ib + j + 1);
#pragma unroll
        for (int m = 0; m < BAND_MAXS; m++) a_c[m] = a_n[m];
        slot = slot + 1 == c.D ? 0 : slot + 1;
        BSTAMP(c.st, 4);
    }
#pragma unroll
    for (int m = 0; m < MS; m++) {
        const int nv = __popcll(__ballot(src[m] >= 0) & 0xFFFFull);
        if (SEI_BAND_ABL & 4) {
#pragma unroll
            for (int n = 0; n < NT; n++) asm volatile("" ::"v"(acc[m][n]));
        } else {
            wrmw<NT>(c.acc_tile, NC, d4[m], nv, c.g, c.r16, acc[m]);
        }
    }
    BSTAMP(c.st, 5);
    BSTAMP_ADD(c.st, 8, c.NS);
    BSTAMP_ADD(c.st, 9, MS * c.NS);
}

// The loader wave: streams the weight slices of the offsets in `umask` (ascending k, NS slices of 16 input planes each)
// into the ring by LDS-DMA, publishes a slice once it has landed (sync[0] = slices landed) and refills a slot when all
// NCW compute waves have released its previous slice (sync[8 + w] = slices released by compute wave w).
template <int NT>
__device__ __forceinline__ void ring_loader(const float *__restrict__ W, float *ring, volatile int *sync, unsigned long long umask,
                                            int K, int CIN, int ldy, int col0, int NS, int D, int NCW, int F, int lane) {
    constexpr int NC = 16 * NT;
    // piece p of a slot = floats [256 p, 256 p + 256) of its [16][NC] image; lane l moves 4 floats of one weight row
    int off[NT];
#pragma unroll
    for (int p = 0; p < NT; p++) {
        const int idx = 256 * p + 4 * lane, plane = idx / NC;
        off[p] = plane * ldy + (idx - plane * NC);
    }
    BandStamps lst;
    lst.start();
#ifndef SEI_BAND_NOPRIO
    // the loader is the youngest wave of its workgroup and would lose every issue arbitration against the compute waves
    // of its SIMD (oldest first); its few instructions gate all of them
    __builtin_amdgcn_s_setprio(3);
#endif
    int i = 0, slot = 0, known_free = D;  // slices [0, known_free) may be issued
    for (unsigned long long rem = umask; rem; rem &= rem - 1) {
        const int k = __builtin_amdgcn_readfirstlane(__ffsll((long long)rem) - 1);
        const float *wk = W + (size_t)k * CIN * ldy + col0;
        for (int j = 0; j < NS; j++) {
            if (i >= known_free && !(SEI_BAND_ABL & 2)) {  // the slot's previous slice must have been released by every compute wave
                // nothing more can be issued for now: publish everything issued so far as soon as it has landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) sync[0] = i;
                asm volatile("" ::: "memory");
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
            BSTAMP(lst, 10);
            const float *ws = wk + (size_t)(16 * j) * ldy;
            float *dst = ring + (size_t)slot * 16 * NC;
#pragma unroll
            for (int p = 0; p < NT; p++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ws + off[p]),
                                                 (__attribute__((address_space(3))) void *)(dst + 256 * p), 16, 0, 0);
            BSTAMP(lst, 11);
            if (i >= BAND_LAG) {
                // slice i - LAG has landed once at most LAG slices' pieces are still outstanding
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT * BAND_LAG) : "memory");
                if (lane == 0) sync[0] = i - BAND_LAG + 1;
                asm volatile("" ::: "memory");
            }
            BSTAMP(lst, 13);
            BSTAMP_ADD(lst, 14, 1);
            i++;
            slot = slot + 1 == D ? 0 : slot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) sync[0] = F;
    BSTAMP(lst, 13);
    lst.flush(lane);
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
        ring_loader<NT>(W, ring, sync, umask, K, CIN, ldy, col0, NS, D, NCW, F, lane);
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
    BandCtx c{X, ring, sync, acc_tile, CIN, NS, D, wave, lane, g, r16, 0, {}};
    c.st.start();
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
            if (m == 0 || m < ix0.ms) a_c[m] = wload_a<SEI_BAND_ABL>(X, ix0.row(m), CIN, 4 * g);
        int released = 0;
        BSTAMP(c.st, 0);
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
            k0 = k1; k1 = k2; k2 = next_k(k2);
            ix0 = ix1; ix1 = ix2; ix2 = load_idx(k2);  // two offsets ahead: its latency is never waited for
            BSTAMP(c.st, 6);
        }
    }
    band_release(c, INT_MAX);  // nothing else of the stream is needed by this band
    band_store<NC>(acc_tile, Y, stats, tile, row0, BAND_ROWS, n_dst, ldy, col0, lane);
    BSTAMP(c.st, 7);
    BSTAMP_ADD(c.st, 12, 1);
    c.st.flush(lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// Shared-tile form of the same idea ("ring-tile").  Private 64-row bands cost 64 x NC floats of LDS per compute wave
// (8-11 waves per CU) and leave a workgroup waiting for its heaviest band (tools/band_stats.py: balance 0.80-0.85).  Here
// the NCW compute waves share ONE accumulator tile of 16 NCW rows, as in conv_fwd.hip: the sub-chunks of an offset are
// dealt to the waves, one each, ROTATED by the offset's sequence number so that every wave ends up with the same share
// of the tile's sub-chunks -- which only pays because nothing forces the waves through the offsets in lock-step any more:
//   * weights come from the ring above (no per-slice barrier);
//   * the one ordering the shared tile needs -- the read-modify-write of offset q may touch rows that an EARLIER offset's
//     read-modify-write of another wave touches -- is a per-wave word in LDS: done[w] = q means wave w has finished every
//     read-modify-write it owns for offsets before q.  A wave checks min_w done[w] >= q just before its own write-back
//     of offset q, i.e. a whole offset's MFMAs after the others signalled: in the steady state nobody waits.
// Per dst row the summation order is unchanged (offsets ascending): bit-identical to the other kernels.
constexpr int RING_DONE = 16;  // sync[16 + w]

struct RingItem {
    int q, k, sub;  // q = sequence number of the offset among the tile's offsets with rules (>= Q: none), sub-chunk (tile-relative)
};

template <int NT>
__device__ __forceinline__ void ring_slice(const float *wb, const float4 &a, bool valid, f32x4 (&acc)[NT], int g, int r16) {
    constexpr int NC = 16 * NT;
    float b[2][NT];
    wload_b<NT>(wb + (4 * g) * NC, r16, b[0]);
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (e + 1 < 4) wload_b<NT>(wb + (4 * g + e + 1) * NC, r16, b[(e + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);  // the next step's LDS reads are ISSUED before this step's MFMAs
        float av = e == 0 ? a.x : e == 1 ? a.y : e == 2 ? a.z : a.w;
        av = valid ? av : 0.f;  // padding slots feed zeros
#pragma unroll
        for (int n = 0; n < NT; n++) {
            if (SEI_BAND_ABL & 8) {
                asm volatile("" ::"v"(av), "v"(b[e & 1][n]));
            } else {
                acc[n] = mfma16(av, b[e & 1][n], acc[n]);
            }
        }
    }
}

template <int NT, int NCW>
__global__ __launch_bounds__(64 * (NCW + 1)) void conv_ring_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int CIN, int n_tiles,
    const float *__restrict__ init, int ldy, int blocks, int D, double *__restrict__ stats) {
    extern __shared__ float lds[];
    static_assert((NCW & (NCW - 1)) == 0 && NCW <= 8, "the rotated deal needs a power-of-two wave count");
    constexpr int NC = 16 * NT, TR = SUB_SLOTS * NCW;
    const int nwork = n_tiles * blocks;
    const int vid = xcd_item(nwork);
    if (vid >= nwork) return;  // whole workgroup
    const int tile = vid / blocks, cblk = vid - tile * blocks;
    const int col0 = cblk * NC;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    float *acc_tile = lds;
    float *ring = lds + (size_t)TR * NC;
    volatile int *sync = reinterpret_cast<volatile int *>(ring + (size_t)D * 16 * NC);
    const int NS = CIN >> 4;

    if (threadIdx.x < BAND_SYNC_WORDS) sync[threadIdx.x] = 0;
    const int segv = lane <= K ? tile_seg[(size_t)tile * (K + 1) + lane] : 0;
    const int nsv = __shfl_down(segv, 1) - segv;
    const unsigned long long kmask = __ballot(lane < K && nsv > 0);
    const int Q = __popcll(kmask), F = Q * NS;

    if (wave == NCW) {
        __syncthreads();
        ring_loader<NT>(W, ring, sync, kmask, K, CIN, ldy, col0, NS, D, NCW, F, lane);
        return;
    }
    const int64_t row0 = (int64_t)tile * TR;
    tile_init<TR, NC, 64 * NCW>(acc_tile, bias ? bias + col0 : nullptr, init, row0, n_dst, ldy, col0);
    __syncthreads();  // sync words zeroed, tile initialised
    BandCtx c{X, ring, sync, acc_tile, CIN, NS, D, wave, lane, g, r16, 0, {}};
    c.st.start();
    const int64_t sub0 = tile_sub_start[tile];

    // this wave's work items: offset number q gives its sub-chunk s to wave (q + s) mod NCW
    unsigned long long rem = kmask;
    int qn = 0;
    auto next_item = [&]() {
        RingItem it{INT_MAX, 0, 0};
        while (rem) {
            const int k = __builtin_amdgcn_readfirstlane(__ffsll((long long)rem) - 1);
            rem &= rem - 1;
            const int q = qn++;
            const int s0 = __builtin_amdgcn_readlane(segv, k), ns = __builtin_amdgcn_readlane(segv, k + 1) - s0;
            const int s = (wave + 64 - q) & (NCW - 1);
            if (s < ns) {
                it.q = q; it.k = k; it.sub = s0 + s;
                break;
            }
        }
        return it;
    };
    struct Idx { int src; uint32_t d4; };
    auto load_idx = [&](const RingItem &it) {
        Idx r;
        const int64_t sb = (sub0 + (it.q < Q ? it.sub : 0)) * SUB_SLOTS;  // clamped: never branch around a load
        r.src = sub_src[sb + r16];
        r.d4 = *reinterpret_cast<const uint32_t *>(sub_dst + sb + 4 * g);
        return r;
    };
    int known_done = 0;
    auto set_done = [&](int q) {
        asm volatile("" ::: "memory");
        if (lane == 0) sync[RING_DONE + wave] = q;
        asm volatile("" ::: "memory");
    };

    RingItem it0 = next_item(), it1 = next_item(), it2 = next_item();
    set_done(it0.q);  // nothing of this wave lands in the tile before its first item
    if (it0.q < Q) {
        Idx ix0 = load_idx(it0), ix1 = load_idx(it1), ix2 = load_idx(it2);
        float4 a_c = wload_a<SEI_BAND_ABL>(X, ix0.src, CIN, 4 * g);
        int released = 0;
        BSTAMP(c.st, 0);
        while (it0.q < Q) {
            const int ib = it0.q * NS;
            if (ib > released) band_release(c, ib);  // slices of offsets this wave holds no sub-chunk of: released unread
            const int rn = it1.q < Q ? ix1.src : -1;
            f32x4 acc[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            int slot = ib % D;
            for (int j = 0; j < NS; j++) {
                const bool same = j + 1 < NS;
                const float4 a_n = wload_a<SEI_BAND_ABL>(X, same ? ix0.src : rn, CIN, (same ? 16 * (j + 1) : 0) + 4 * g);
                BSTAMP(c.st, 1);
                band_wait_filled(c, ib + j);
                BSTAMP(c.st, 2);
                ring_slice<NT>(ring + (size_t)slot * 16 * NC, a_c, ix0.src >= 0, acc, g, r16);
                BSTAMP(c.st, 3);
                band_release(c, ib + j + 1);
                a_c = a_n;
                slot = slot + 1 == D ? 0 : slot + 1;
                BSTAMP(c.st, 4);
            }
            released = ib + NS;
            // the tile rows of this sub-chunk may still be the target of an EARLIER offset's write-back of another wave
            if (it0.q > known_done && !(SEI_BAND_ABL & 64)) {
                int spins = 0;
                for (;;) {
                    int v = lane < NCW ? sync[RING_DONE + lane] : INT_MAX;
#pragma unroll
                    for (int d = 1; d < 8; d <<= 1) v = min(v, __shfl_xor(v, d));
                    known_done = __builtin_amdgcn_readfirstlane(v);
                    if (it0.q <= known_done) break;
                    if (++spins > BAND_SPIN_CAP) {
                        if (lane == 0) atomicAdd(&g_band_timeouts, 1);
                        known_done = INT_MAX;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            BSTAMP(c.st, 6);
            const int nv = __popcll(__ballot(ix0.src >= 0) & 0xFFFFull);
            if (SEI_BAND_ABL & 4) {
#pragma unroll
                for (int n = 0; n < NT; n++) asm volatile("" ::"v"(acc[n]));
            } else {
                wrmw<NT>(acc_tile, NC, ix0.d4, nv, g, r16, acc);
            }
            set_done(it1.q);  // everything this wave owns before its next item is in the tile
            BSTAMP(c.st, 5);
            BSTAMP_ADD(c.st, 8, NS);
            BSTAMP_ADD(c.st, 9, NS);
            it0 = it1; it1 = it2; it2 = next_item();
            ix0 = ix1; ix1 = ix2; ix2 = load_idx(it2);  // two items ahead: its latency is never waited for
        }
    }
    band_release(c, INT_MAX);
    // every wave's write-backs must be in the tile before it is stored: the done words (all INT_MAX by then), no s_barrier
    // (the loader wave may have exited already)
    if (!(SEI_BAND_ABL & 64)) {
        int spins = 0;
        for (;;) {
            int v = lane < NCW ? sync[RING_DONE + lane] : INT_MAX;
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) v = min(v, __shfl_xor(v, d));
            if (__builtin_amdgcn_readfirstlane(v) >= Q) break;
            if (++spins > BAND_SPIN_CAP) {
                if (lane == 0) atomicAdd(&g_band_timeouts, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    tile_store<TR, NC, 64 * NCW>(acc_tile, Y, row0, n_dst, ldy, col0);
    if (stats) tile_stats<TR, NC, 64 * NCW>(acc_tile, stats, tile, row0, n_dst, ldy, col0);
    BSTAMP(c.st, 7);
    BSTAMP_ADD(c.st, 12, 1);
    c.st.flush(lane);
}

// LDS of one workgroup: accumulator rows (NCW private 64-row bands, or one shared tile of 16 NCW rows) + D ring slots + sync words
static size_t band_lds(int nt, int ncw, int d, bool shared) {
    const size_t rows = shared ? (size_t)SUB_SLOTS * ncw : (size_t)ncw * BAND_ROWS;
    return (rows * 16 * nt + (size_t)d * 16 * 16 * nt) * sizeof(float) + BAND_SYNC_WORDS * sizeof(int);
}

#ifdef SEI_EXPERIMENTS
// Kernel-experiment builds only (make EXTRA=-DSEI_EXPERIMENTS; tools/sweep_band.py): SEI_BAND_PLAN=
// "nOut:NT,NCW,D[,S][;...]" overrides column tiles per workgroup, compute waves per workgroup, ring depth and the form
// (S = 1: shared tile, 0: private bands).
static bool env_band_plan(int nOut, int *nt, int *ncw, int *d, int *shared) {
    const char *e = getenv("SEI_BAND_PLAN");
    while (e && *e) {
        int no = 0, a = 0, b = 0, c = 0, sh = 0;
        const int got = sscanf(e, "%d:%d,%d,%d,%d", &no, &a, &b, &c, &sh);
        if (got >= 4 && no == nOut) {
            *nt = a; *ncw = b; *d = c; *shared = got >= 5 ? sh : 0;
            return true;
        }
        e = strchr(e, ';');
        if (e) e++;
    }
    return false;
}
#endif

// Default use: the layers from 96 planes on (levels 2-5 of the DUNE network).  sei_debug_force_conv_kernel(3) pins it
// for every MFMA-sized layer (tests).
bool plan_band(int nIn, int nOut, int K, int64_t n_dst, ConvPlan *p) {
    if (nIn % 16 || nOut % 16 || nIn < 16 || nOut > 256 || K + 1 > 64) return false;
    const int kind = forced_conv_kernel();
    if (kind != CONV_KERNEL_AUTO && kind != CONV_KERNEL_BAND) return false;
    const bool forced = kind == CONV_KERNEL_BAND;
    // Not a default plan: on the DUNE batch both forms measured below the workgroup-tile kernel they were built to replace
    // (profiles/r03_conv_band_*.log: 58-62 vs 76 TFLOP/s at 96 planes, 67-72 vs 82 at 128) -- with every hand-off wait
    // removed their ceiling is 83-88 (ablation builds), and the waits that remain are structural (weight slices per MFMA
    // through the vector-memory path; the shared tile's write-back order).  Kept as a third, bit-identical family behind
    // sei_debug_force_conv_kernel(3) and the experiment build's SEI_BAND_PLAN.
    if (!forced) return false;
    const int tiles = nOut / 16;
    // column tiles per workgroup: the largest divisor of the tile count up to 5 (two or more column blocks from 96
    // columns on): accumulator rows and ring must leave room for several workgroups per CU
    int nt = 0;
    for (int cnd = (nOut >= 96 ? 5 : 6); cnd >= 1; cnd--)
        if (tiles % cnd == 0) { nt = cnd; break; }
    if (nOut == 96) nt = 3;
    if (nt == 1 && tiles > 1) return false;  // 7 / 11 / 13 column tiles: the workgroup-tile kernel
    int ncw = 8, shared = 1;
    // ring depth: at most 8 slices, fewer where three workgroups per CU would not fit otherwise (never below LAG + 2)
    int d = 8;
    while (d > BAND_LAG + 2 && band_lds(nt, ncw, d, shared) > 53 * 1024) d--;
#ifdef SEI_EXPERIMENTS
    int en, ew, ed, es;
    if (env_band_plan(nOut, &en, &ew, &ed, &es) && en >= 1 && en <= 6 && tiles % en == 0 && ed >= BAND_LAG + 2 &&
        ((es == 0 && (ew == 2 || ew == 3 || ew == 4 || ew == 5 || ew == 6 || ew == 8)) || (es == 1 && (ew == 4 || ew == 8)))) {
        nt = en; ncw = ew; d = ed; shared = es;
    }
#endif
    if (band_lds(nt, ncw, d, shared) > 160 * 1024) return false;
    p->kind = 3; p->cs = shared; p->tile_rows = shared ? SUB_SLOTS * ncw : BAND_ROWS; p->col_group = 0;
    p->blocks = tiles / nt; p->nt = nt; p->nw = ncw; p->g = d;
    p->lds_bytes = band_lds(nt, ncw, d, shared);
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

template <int NT, int NCW>
static int launch_ring(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                       int nIn, int nOut, const float *init, double *stats, hipStream_t st) {
    SEI_TRY(ensure_max_lds((const void *)conv_ring_mfma_kernel<NT, NCW>, 160 * 1024));
    const int nwork = t.n_tiles * p.blocks;
    conv_ring_mfma_kernel<NT, NCW><<<((nwork + 7) / 8) * 8, 64 * (NCW + 1), p.lds_bytes, st>>>(
        X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init, nOut,
        p.blocks, p.g, stats);
    SEI_LAUNCH_CHECK();
    return 0;
}

#define BAND_CASE(NT_, NCW_) \
    if (!p.cs && p.nt == NT_ && p.nw == NCW_) return launch_band<NT_, NCW_>(t, p, X, W, bias, Y, nIn, nOut, init, stats, st);
#define RING_CASE(NT_, NCW_) \
    if (p.cs && p.nt == NT_ && p.nw == NCW_) return launch_ring<NT_, NCW_>(t, p, X, W, bias, Y, nIn, nOut, init, stats, st);

int conv_forward_band(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                      int nIn, int nOut, const float *init, double *stats, hipStream_t st) {
    if (t.tile_rows != p.tile_rows) SEI_FAIL("conv_forward_band: rulebook variant does not match the plan");
    if (p.g < BAND_LAG + 2) SEI_FAIL("conv_forward_band: ring too shallow");
    RING_CASE(1, 8) RING_CASE(2, 8) RING_CASE(3, 8) RING_CASE(4, 8) RING_CASE(5, 8) RING_CASE(6, 8)
#ifdef SEI_EXPERIMENTS  // alternates reachable through SEI_BAND_PLAN
    RING_CASE(2, 4) RING_CASE(3, 4) RING_CASE(4, 4) RING_CASE(5, 4) RING_CASE(6, 4)
    BAND_CASE(1, 4) BAND_CASE(2, 4) BAND_CASE(3, 4) BAND_CASE(4, 4) BAND_CASE(5, 4) BAND_CASE(6, 4)
    BAND_CASE(3, 8) BAND_CASE(4, 8) BAND_CASE(5, 8) BAND_CASE(6, 8)
#endif
    SEI_FAIL("conv_forward: no band-kernel instantiation for this plan");
}

}  // namespace sei

extern "C" int sei_debug_band_timeouts(int32_t *out, int32_t reset) {
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(sei::g_band_timeouts), sizeof(int)) != hipSuccess) {
        sei::set_error("sei_debug_band_timeouts: hipMemcpyFromSymbol failed");
        return 1;
    }
    int w = 0;
    if (sei::dw_ring_timeouts(&w, reset != 0)) {
        sei::set_error("sei_debug_band_timeouts: hipMemcpyFromSymbol failed");
        return 1;
    }
    if (out) *out = v + w;
    if (reset) {
        int z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_band_timeouts), &z, sizeof(int)) != hipSuccess) return 1;
    }
    return 0;
}

#endif  // SEI_EXPERIMENTS

#if defined(SEI_STAMPS) && defined(SEI_EXPERIMENTS)
extern "C" int sei_debug_band_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sei::g_bstamps), sizeof(sei::g_bstamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_bstamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
