// Sparse convolution forward / dX for gfx950, column-owner form (scn.SubmanifoldConvolution: resnet.py:30-50,
// sparse_building_blocks.py:29-34; the C -> C layers of the 96..192-plane levels).
//
// The workgroup-tile kernel (conv_fwd.hip) deals the sub-chunks of one (offset, weight slice) phase round-robin to its
// eight waves; on the DUNE batch a phase of a 128-row tile holds ~4.4 sub-chunks, so half the waves sit at the phase
// barrier (profiles/r04_conv_tile_cu_timeline_*: 39 % of the time no wave of the CU is in an MFMA block).  Here the
// split is by COLUMN instead: wave w of a workgroup of nOut/16 waves owns output columns [16 w, 16 w + 16) of the whole
// 128-row tile and walks EVERY sub-chunk of the tile, so
//   * all waves of a workgroup do exactly the same work per step -- the barrier only absorbs jitter, not imbalance;
//   * the accumulator tile in LDS is never shared between waves (each wave read-modify-writes its own 16 columns):
//     no ordering between waves is needed for the sums, and the per-row order (offsets ascending, sub-chunks in order,
//     one MFMA chain over the input planes per sub-chunk, then tile += partial) is the one of the other kernels --
//     results are bit-identical to theirs;
//   * the A operands (16 gathered src rows, whole rows) are fetched ONCE per workgroup by per-lane-address LDS-DMA
//     (global_load_lds_dwordx4, one 1-KiB piece per wave) into a ring of COLS_D sub-chunks, three sub-chunks ahead, and
//     every wave reads them from LDS (ds_read_b128, conflict-free through a per-row rotation of the 16-byte chunks);
//   * the B operands (the wave's 16 columns of W[k]: nIn/4 registers) stay in REGISTERS for the whole offset and the next
//     offset's are loaded one offset ahead from a packed copy of W ([k][column tile][16-plane group][lane][4]: one
//     global_load_dwordx4 per lane and 16 input planes) -- the same bytes per (tile, offset) as the weight slices the
//     tile kernel stages in LDS, but no LDS capacity: 72 KB per workgroup at 96 planes, two workgroups per CU.
// One barrier per sub-chunk.  The vector-memory queue of a wave holds, in issue order, one A piece per step, the next
// offset's B batch at an offset switch and (wave 0) the index words of a later sub-chunk; s_waitcnt vmcnt(n) immediates
// are chosen from the wave-uniform history of those (every path waits for AT LEAST what it needs).
#include "common.h"
#include "conv_tile.h"

#include <stdlib.h>
#include <string.h>

// Diagnostic builds only (make EXTRA="-DSEI_EXPERIMENTS -DSEI_COLS_ABL=n"; tools/cols_ab.sh; results wrong, timing only):
// 1 the B operands are never reloaded (no batch, no copy, no waits for them), 2 no read-modify-write of the tile,
// 4 no barrier in the loop, 8 no MFMAs, 16 / 32 two / four independent MFMA chains per sub-chunk
#ifndef SEI_COLS_ABL
#define SEI_COLS_ABL 0
#endif


#ifdef SEI_EXPERIMENTS   // the shipped library does not compile this file (Makefile); its stubs are in conv_fwd.hip
namespace sei {

// Diagnostic builds only (make EXTRA="-DSEI_EXPERIMENTS -DSEI_STAMPS"; tools/cols_stamps.py): every wave adds the s_memtime
// ticks it spends in each part of a step to g_cstamps.  The product build compiles none of it.
#ifdef SEI_STAMPS
__device__ unsigned long long g_cstamps[16];
struct CStamp {
    unsigned long long t, acc[16];
    __device__ __forceinline__ CStamp() {
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
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < 16; i++) atomicAdd(&g_cstamps[i], acc[i]);
    }
};
#define CSTAMP(i) cst.mark(i)
#define CSTAMP_ADD(i, v) cst.add(i, v)
#else
struct CStamp {
    __device__ __forceinline__ void flush() {}
};
#define CSTAMP(i)
#define CSTAMP_ADD(i, v)
#endif

constexpr int COLS_TR = 128;        // dst rows per tile
constexpr int COLS_D = 4;           // A ring depth (sub-chunks): step i computes sub-chunk i, reads i + 1, requests i + 4
constexpr int COLS_IDX = 8;         // index ring depth (sub-chunks)
constexpr int COLS_IDX_AHEAD = 7;
constexpr int COLS_IDX_WORDS = 20;  // 16 src rows + 16 dst bytes

// position of 16-byte chunk c of ring row r: r * CH + (c + rot(r)) mod CH; the rotation makes the ds_read_b128 of the
// four lane groups (MI355X_MICROARCH.md, LDS) conflict-free for row strides of 24 / 40 (CH mod 16 == 8) and 32 / 48 chunks
__host__ __device__ constexpr int cols_rot(int r, int ch) { return ch % 16 == 8 ? ((r & 2) | ((r & 8) >> 1)) : (2 * r) % ch; }

template <int N>
__device__ __forceinline__ void cols_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A step issues, in this order: [B batch: NJ loads, first step of an offset only] [wave 0: index words] [A piece].  The
// piece a step waits for is the LAST operation of the step three steps back, so two whole steps came after it:
// at most N = 2 + 2 z + NJ nb operations may still be outstanding (z: wave 0, nb <= 2: batches in those two steps).
template <int NJ>
__device__ __forceinline__ void cols_wait_dma(int z, int nb) {
    if (z == 0) {
        if (nb == 0) cols_vmcnt<2>();
        else if (nb == 1) cols_vmcnt<2 + NJ>();
        else cols_vmcnt<2 + 2 * NJ>();
    } else {
        if (nb == 0) cols_vmcnt<4>();
        else if (nb == 1) cols_vmcnt<4 + NJ>();
        else cols_vmcnt<4 + 2 * NJ>();
    }
}
// the wave's B operands of one offset: NJ x dwordx4 per lane from the packed panel of (k, column tile).  Plain loads: the
// compiler tracks their completion (an inline-asm load would let it copy a register whose data has not arrived); the
// memory barriers pin the batch between the other vector-memory operations of the step, whose ORDER the vmcnt arithmetic
// of cols_wait_dma relies on.
template <int NJ>
__device__ __forceinline__ void cols_bload_all(f32x4 (&b)[NJ], int lane, const float *panel) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < NJ; j++) b[j] = *reinterpret_cast<const f32x4 *>(panel + 256 * j + 4 * lane);
    asm volatile("" ::: "memory");
}

// operands of one sub-chunk as a wave holds them: A (16 rows x nIn as MFMA A fragments) and the index words
template <int NJ>
struct ColsOps {
    f32x4 a[NJ];
    int srcv;       // lane r16: src row of rule slot r16 (validity only)
    uint32_t d4;    // lane group g: packed tile-local dst rows of slots 4g..4g+3
};

template <int NJ>
__device__ __forceinline__ void cols_read(ColsOps<NJ> &o, const float *stg, const int (&aoff)[NJ], const int *ix, int r16, int g) {
#pragma unroll
    for (int j = 0; j < NJ; j++) o.a[j] = *reinterpret_cast<const f32x4 *>(stg + aoff[j]);
    o.srcv = ix[r16];
    o.d4 = (uint32_t)ix[16 + g];
}

// NW = nOut / 16 waves (one column tile each), NJ = nIn / 16 (A pieces per sub-chunk; NJ == NW: one per wave)
template <int NW, int NJ>
__global__ __launch_bounds__(64 * NW, 3) void conv_cols_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ Wp, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int n_tiles,
    const float *__restrict__ init, double *__restrict__ stats, ConvEpilogue ep) {
    static_assert(NJ == NW, "one A piece per wave");
    extern __shared__ float lds[];
    constexpr int TR = COLS_TR, CIN = 16 * NJ, COUT = 16 * NW, CH = CIN / 4, NTH = 64 * NW;
    const int vid = xcd_item(n_tiles);
    if (vid >= n_tiles) return;
    const int tile = vid;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    float *acc_tile = lds;
    float *ring = lds + (size_t)TR * COUT;
    int *idxs = reinterpret_cast<int *>(ring + (size_t)COLS_D * 16 * CIN);

    CStamp cst;
    tile_init<TR, COUT, NTH>(acc_tile, bias, init, (int64_t)tile * TR, n_dst, COUT, 0);

    // segment table of the tile: lane l holds seg[l] (K + 1 <= 64); nonempty offsets as a bit mask
    const int segv = lane <= K ? tile_seg[(size_t)tile * (K + 1) + lane] : 0;
    const int nsv = __shfl_down(segv, 1) - segv;
    const unsigned long long kmask = __ballot(lane < K && nsv > 0);
    const int n = __builtin_amdgcn_readlane(segv, K);   // sub-chunks of the tile
    const int64_t sub0 = tile_sub_start[tile];
    auto next_k = [&](int kfrom, int fallback) {  // first nonempty offset >= kfrom
        const unsigned long long rem = kfrom < 64 ? (kmask >> kfrom) << kfrom : 0ull;
        return rem ? __builtin_amdgcn_readfirstlane(__ffsll((long long)rem) - 1) : fallback;
    };

    if (n > 0) {
        const int z = wave == 0 ? 1 : 0;
        // DMA role of this lane: ring position p = 64 wave + lane holds chunk `chunk` of ring row `drow`
        const int p = 64 * wave + lane;
        const int drow = p / CH, cp = p - drow * CH;
        int chunk = cp - cols_rot(drow, CH);
        if (chunk < 0) chunk += CH;
        // reader role: lane (r16, g), plane group j: chunk 4 j + g of ring row r16 (float offsets into a ring slot)
        int aoff[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            int c = 4 * j + g + cols_rot(r16, CH);   // < 2 CH
            c = c >= CH ? c - CH : c;
            aoff[j] = (r16 * CH + c) * 4;
        }
        const size_t panel_floats = (size_t)NJ * 256;   // one (k, column tile) panel
        auto panel = [&](int k) { return Wp + ((size_t)k * NW + wave) * panel_floats; };

        auto idx_dma = [&](int seq) {  // wave 0: index words of sub-chunk min(seq, n - 1) -> index ring slot seq & 7
            const int j = seq < n ? seq : n - 1;
            const int64_t sb = (sub0 + j) * SUB_SLOTS;
            const int *gp = lane < 16 ? sub_src + sb + lane
                                      : reinterpret_cast<const int *>(sub_dst + sb) + ((lane - 16) & 3);
            if (lane < COLS_IDX_WORDS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                                 (__attribute__((address_space(3))) void *)(idxs + (seq & (COLS_IDX - 1)) * COLS_IDX_WORDS),
                                                 4, 0, 0);
        };
        auto a_dma = [&](int seq) {  // this wave's piece of sub-chunk seq (clamped when its indices were loaded)
            const int src = idxs[(seq & (COLS_IDX - 1)) * COLS_IDX_WORDS + drow];
            const uint32_t row = (uint32_t)(src > 0 ? src : 0);
            const float *gp = X + (uint64_t)row * (uint32_t)CIN + 4 * chunk;
            float *lp = ring + (size_t)(seq & (COLS_D - 1)) * (16 * CIN) + 256 * wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                             (__attribute__((address_space(3))) void *)lp, 16, 0, 0);
        };

        int kc = next_k(0, 0);
        int ke = __builtin_amdgcn_readlane(segv, kc + 1);
        int kn = next_k(kc + 1, kc);
        f32x4 bcur[NJ], bnext[NJ];
        if (wave == 0)
            for (int s = 0; s < COLS_IDX_AHEAD; s++) idx_dma(s);
        cols_bload_all<NJ>(bcur, lane, panel(kc));
        cols_bload_all<NJ>(bnext, lane, panel(kn));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // index words visible; tile initialised
        for (int s = 0; s < COLS_D; s++) a_dma(s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // sub-chunk 0 is read before the loop; step i reads the operands of sub-chunk i + 1 (behind the barrier that says its
        // pieces have landed) BEFORE it runs the MFMAs of sub-chunk i, so that a wave's LDS latency sits under its own MFMAs
        float *acol = acc_tile + 16 * wave + r16;
        ColsOps<NJ> oa, ob;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cols_read<NJ>(oa, ring, aoff, idxs, r16, g);

        CSTAMP(0);
        int sw1 = 0, sw2 = 0, pend = 0;
        auto step = [&](int i, ColsOps<NJ> &oc, ColsOps<NJ> &on) {
            cols_wait_dma<NJ>(z, sw1 + sw2);    // this wave's piece of sub-chunk i + 1 has landed
            CSTAMP(1);
            // ... and everybody's; every wave's reads of sub-chunk i (slot i & 3) have returned: the slot is free
            if (!(SEI_COLS_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            CSTAMP(2);
            const int sw = pend;
            if (pend && !(SEI_COLS_ABL & 1)) {   // first step of an offset: request the operands of the one after it
                cols_bload_all<NJ>(bnext, lane, panel(kn));
                pend = 0;
            }
            if (wave == 0) idx_dma(i + COLS_IDX_AHEAD);
            asm volatile("" ::: "memory");   // issue order is part of the vmcnt arithmetic
            a_dma(i + COLS_D);
            asm volatile("" ::: "memory");
            sw2 = sw1;
            sw1 = sw;
            CSTAMP(3);
            cols_read<NJ>(on, ring + (size_t)((i + 1) & (COLS_D - 1)) * (16 * CIN), aoff,
                          idxs + ((i + 1) & (COLS_IDX - 1)) * COLS_IDX_WORDS, r16, g);
            // tile rows of sub-chunk i: D[row = 4g+q][col = r16] belongs to rule slot 4g+q; slots [0, nv) are rules (padding
            // sits at the end of a sub-chunk; its dst byte is 0 and must not be written)
            float *prow[4];
            float tv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                prow[q] = acol + ((oc.d4 >> (8 * q)) & 0xFF) * COUT;
                tv[q] = *prow[q];
            }
            const int nv = __popcll(__ballot(oc.srcv >= 0) & 0xFFFFull);
            __builtin_amdgcn_sched_barrier(0);   // every LDS read above is issued before the first MFMA
            CSTAMP(4);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (SEI_COLS_ABL & 48) {   // diagnostic: 2 (16) or 4 (32) independent MFMA chains, summed at the end (other rounding)
                constexpr int NCH = (SEI_COLS_ABL & 32) ? 4 : 2;
                f32x4 ac[NCH];
#pragma unroll
                for (int c = 0; c < NCH; c++) ac[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NJ; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) ac[(4 * j + e) % NCH] = mfma16(oc.a[j][e], bcur[j][e], ac[(4 * j + e) % NCH]);
#pragma unroll
                for (int c = 0; c < NCH; c++) acc += ac[c];
            } else
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (SEI_COLS_ABL & 8) asm volatile("" ::"v"(oc.a[j][e]), "v"(bcur[j][e]));
                    else acc = mfma16(oc.a[j][e], bcur[j][e], acc);
                }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (4 * g + q < nv && !(SEI_COLS_ABL & 2)) *prow[q] = tv[q] + acc[q];
            CSTAMP(5);
            CSTAMP_ADD(8, 1);
            if (i + 1 == ke && !(SEI_COLS_ABL & 1)) {
                // last step of an offset: the next one's operands (requested at its first step) move over.  hipcc waits for them
                // here; the request of the following batch sits behind the barrier of the next step, so the copies cannot
                // sink below it (in a flat "copy, then load" block they did: loads into spare registers, then vmcnt(0))
#pragma unroll
                for (int j = 0; j < NJ; j++) bcur[j] = bnext[j];
                kc = kn;
                ke = __builtin_amdgcn_readlane(segv, kc + 1);
                kn = next_k(kc + 1, kc);
                pend = 1;
                CSTAMP(6);
                CSTAMP_ADD(9, 1);
            }
        };
        int i = 0;
        for (; i + 2 <= n; i += 2) {
            step(i, oa, ob);
            step(i + 1, ob, oa);
        }
        if (i < n) step(i, oa, ob);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped look-ahead pieces
    }
    __syncthreads();
    const int64_t row0 = (int64_t)tile * TR;
    tile_store<TR, COUT, NTH>(acc_tile, Y, row0, n_dst, COUT, 0, ep);
    if (stats) tile_stats<TR, COUT, NTH>(acc_tile, stats, tile, row0, n_dst, COUT, 0);
    CSTAMP(7);
    CSTAMP_ADD(10, 1);
    cst.flush();
}

static size_t cols_lds(int nIn, int nOut) {
    return ((size_t)COLS_TR * nOut + (size_t)COLS_D * 16 * nIn) * sizeof(float) + COLS_IDX * COLS_IDX_WORDS * sizeof(int);
}

static bool cols_instantiated(int nIn, int nOut) { return nIn == nOut && (nOut == 96 || nOut == 128 || nOut == 160 || nOut == 192); }

bool plan_cols(int nIn, int nOut, int K, int64_t n_dst, ConvPlan *p) {
    if (!cols_instantiated(nIn, nOut) || K + 1 > 64) return false;
    const int kind = forced_conv_kernel();
    if (kind != CONV_KERNEL_AUTO && kind != CONV_KERNEL_COLS) return false;
    if (kind == CONV_KERNEL_AUTO) return false;   // not a default yet
    p->kind = 4; p->tile_rows = COLS_TR; p->col_group = -1; p->blocks = 1; p->nt = 1; p->nw = nOut / 16;
    p->lds_bytes = cols_lds(nIn, nOut);
    return p->lds_bytes <= 160 * 1024;
}

template <int NW>
static int launch_cols(const TiledRules &t, const ConvPlan &p, const float *X, const float *Wp, const float *bias, float *Y,
                       const float *init, double *stats, hipStream_t st, const ConvEpilogue &ep) {
    SEI_TRY(ensure_max_lds((const void *)conv_cols_mfma_kernel<NW, NW>, 160 * 1024));
    conv_cols_mfma_kernel<NW, NW><<<((t.n_tiles + 7) / 8) * 8, 64 * NW, p.lds_bytes, st>>>(
        X, Wp, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, t.n_tiles, init, stats, ep);
    SEI_LAUNCH_CHECK();
    return 0;
}

int conv_forward_cols(const TiledRules &t, const ConvPlan &p, const float *X, const float *Wp, const float *bias, float *Y,
                      int nIn, int nOut, const float *init, double *stats, hipStream_t st, const ConvEpilogue &ep) {
    if (nIn != nOut) SEI_FAIL("conv_forward: the column-owner kernel serves C -> C layers");
    if (nOut == 96) return launch_cols<6>(t, p, X, Wp, bias, Y, init, stats, st, ep);
    if (nOut == 128) return launch_cols<8>(t, p, X, Wp, bias, Y, init, stats, st, ep);
    if (nOut == 160) return launch_cols<10>(t, p, X, Wp, bias, Y, init, stats, st, ep);
    if (nOut == 192) return launch_cols<12>(t, p, X, Wp, bias, Y, init, stats, st, ep);
    SEI_FAIL("conv_forward: no column-owner instantiation for this plan");
}

}  // namespace sei

#ifdef SEI_STAMPS
extern "C" int sei_debug_cols_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sei::g_cstamps), sizeof(sei::g_cstamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_cstamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
#endif  // SEI_EXPERIMENTS
