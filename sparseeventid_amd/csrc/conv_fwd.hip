// Sparse convolution forward / dX for gfx950 (scn.SubmanifoldConvolution / scn.Convolution:
// resnet.py:30-50,105-110; sparse_building_blocks.py:29-34,110-117).
//
// Output-stationary  gather -> MFMA -> LDS-accumulate.
//   One workgroup of 4 (or 8: NW) waves owns a tile of TR (64/128/256) dst rows whose fp32 accumulators live
//   in LDS.  It walks the filter offsets k; for each k the tile's rules are compacted into
//   sub-chunks of 16 rules that are dealt round-robin to the wavefronts.  W[k] is streamed through
//   LDS by LDS-DMA in slices of 16/32 input planes (double buffered, ONE barrier per slice) and is
//   shared by all waves; src feature rows are gathered straight into MFMA A-operand registers one
//   phase ahead (float4 per lane, 64 B per rule row and step; rule indices two offsets ahead),
//   multiplied with v_mfma_f32_16x16x4_f32 (exact fp32) into zero-initialised accumulators, and after the
//   offset's last slice the 16xN partial sums are added into the LDS tile by a plain read-modify-write.
//   Rows of one k are distinct and the k-phases are barrier-separated, so the summation order is fixed
//   (and the same as the reference's per-offset  out[rows] += in[rows] @ W[k]): run-to-run deterministic,
//   no atomics; every dst row is written to HBM exactly once.  Tiles are sized so that 2-6 workgroups share
//   a CU: the barrier / gather latency of one workgroup is covered by its neighbours' MFMAs.
#include "common.h"
#include <type_traits>
#include "conv_tile.h"

#include <stdlib.h>
#include <string.h>
#include <utility>

#ifdef SEI_ABL_NOBAR  // diagnostic: the per-phase workgroup barrier removed (results are wrong, timing only)
#define SEI_BARRIER_INSN "s_nop 0"
#else
#define SEI_BARRIER_INSN "s_barrier"
#endif

namespace sei {

#ifdef SEI_CLOCK
__device__ unsigned long long g_clk[8][2];   // families 0 (conv_round_mfma_kernel) and 1 (conv_mround_mfma_kernel)
#endif

// Diagnostic build only (make EXTRA=-DSEI_STAMPS; tools/stamps.py): wave 1 of every tile adds the s_memtime
// ticks it spends in each part of the forward kernel to g_stamps.  The product build compiles none of it.
#ifdef SEI_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[16] = {}
#define STAMP(i) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; } while (0)
#define STAMP_ADD(i, v) st_acc[i] += (v)
#define STAMP_FLUSH do { if (threadIdx.x == 64) { for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_stamps[i_], st_acc[i_]); } } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_ADD(i, v)
#define STAMP_FLUSH
#endif

// Diagnostic build only (make EXTRA="-DSEI_EXPERIMENTS -DSEI_TRACE"; tools/trace_tile.py): the TIMELINE of one compute
// unit.  The first workgroup of a launch claims its CU; every workgroup that runs on that CU afterwards takes a slot and
// each of its waves writes, per phase, the s_memtime at: barrier exit, loads issued, MFMA block done, end of phase (after
// the write-back at the end of an offset), with the number of sub-chunks it held in the low bits.  Answers what the
// aggregated stamps cannot: which waves of which workgroups want one SIMD's matrix pipe at the same time, and what the
// last wave at each barrier was doing.  The product build compiles none of it.
#ifdef SEI_TRACE
constexpr int TRACE_SLOTS = 96, TRACE_WAVES = 8, TRACE_EV = 512;
struct TraceBuf {
    int cu_key;                 // -1 until claimed: (xcc_id << 16) | (hw_id & 0xFF00)
    int n_slots;
    int hw[TRACE_SLOTS][TRACE_WAVES];       // raw HW_ID of each wave
    int nev[TRACE_SLOTS][TRACE_WAVES];
    int vid[TRACE_SLOTS];
    unsigned long long t0[TRACE_SLOTS], t1[TRACE_SLOTS];   // workgroup start / end (wave 0)
    unsigned long long ev[TRACE_SLOTS][TRACE_WAVES][TRACE_EV];
};
__device__ TraceBuf g_trace;
#define TRACE_DECL                                                                                      \
    int &tr_slot_s = seg[K + 1];   /* one word behind the segment table (fwd_lds reserves it in this build) */ \
    int tr_n = 0;                                                                                       \
    {                                                                                                   \
        const int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF; \
        const int key = (xcc << 16) | (hw & 0xFF00);                                                    \
        if (threadIdx.x == 0) {                                                                         \
            int prev = atomicCAS(&g_trace.cu_key, -1, key);                                             \
            int sl = -1;                                                                                \
            if (prev == -1 || prev == key) sl = atomicAdd(&g_trace.n_slots, 1);                         \
            if (sl >= TRACE_SLOTS) sl = -1;                                                             \
            tr_slot_s = sl;                                                                             \
            if (sl >= 0) { g_trace.vid[sl] = vid; g_trace.t0[sl] = __builtin_amdgcn_s_memtime(); }      \
        }                                                                                               \
        __syncthreads();                                                                                \
        if (tr_slot_s >= 0 && lane == 0 && wave < TRACE_WAVES) g_trace.hw[tr_slot_s][wave] = hw;        \
    }                                                                                                   \
    const int tr_slot = tr_slot_s
#define TRACE(tag)                                                                                      \
    do {                                                                                                \
        if (tr_slot >= 0 && lane == 0 && wave < TRACE_WAVES && tr_n < TRACE_EV)                         \
            g_trace.ev[tr_slot][wave][tr_n] = (__builtin_amdgcn_s_memtime() << 8) | (unsigned long long)(tag); \
        tr_n++;                                                                                         \
    } while (0)
#define TRACE_END                                                                                       \
    do {                                                                                                \
        if (tr_slot >= 0 && lane == 0 && wave < TRACE_WAVES) g_trace.nev[tr_slot][wave] = tr_n < TRACE_EV ? tr_n : TRACE_EV; \
        if (tr_slot >= 0 && threadIdx.x == 0) g_trace.t1[tr_slot] = __builtin_amdgcn_s_memtime();      \
    } while (0)
#else
#define TRACE_DECL
#define TRACE(tag)
#define TRACE_END
#endif

// B operands of one input plane for the wave's NT column tiles.  The weight columns of every column
// group are stored permuted (conv_permute_col) so that lane r16 finds the values of 4 (2, 1) column
// tiles in ONE ds_read_b128 (b64, b32): position 64b + 4r + j holds column 64b + 16j + r.
template <int NT>
__device__ __forceinline__ void load_b(const float *__restrict__ wrow, int r16, float (&b)[NT]) {
#ifdef SEI_ABL_NOB  // diagnostic: no LDS reads for the B operands
#pragma unroll
    for (int n = 0; n < NT; n++) b[n] = 1.0f + n;
    return;
#endif
    constexpr int B4 = NT / 4, REM = NT % 4;
#pragma unroll
    for (int q = 0; q < B4; q++) {
        const float4 v = *reinterpret_cast<const float4 *>(wrow + 64 * q + 4 * r16);
        b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
    }
    if constexpr (REM >= 2) {
        const float2 v = *reinterpret_cast<const float2 *>(wrow + 64 * B4 + 2 * r16);
        b[4 * B4 + 0] = v.x; b[4 * B4 + 1] = v.y;
    }
    if constexpr (REM == 1) b[4 * B4] = wrow[64 * B4 + r16];
    if constexpr (REM == 3) b[4 * B4 + 2] = wrow[64 * B4 + 32 + r16];
}

// one (k, slice) phase for a wave owning MS sub-chunks: straight-line, so the LDS reads of step
// e+1 are scheduled under the MFMAs of step e.
struct NoSide { __device__ __forceinline__ void operator()(int, int) const {} };
// side(st, tok): work of the wave that does not depend on this phase's products, issued BEHIND the MFMAs of step st, so that it
// runs while those (and the other waves' MFMAs) occupy the matrix pipe instead of in front of / behind the whole block.  `tok`
// is an opaque zero that "comes from" the step's last MFMA: side work must hang its conditions on it, or hipcc hoists a
// conditional block of side work -- loads and all -- in front of the MFMA block ([r5] conv_mround_mfma_kernel, two long phases).
template <int NT, int SLJ, int MS, int MAXS, int COUT, class Side = NoSide>
__device__ __forceinline__ void compute_phase(const float *__restrict__ wb, const float4 (&a)[MAXS][SLJ],
                                              f32x4 (&acc)[MAXS][NT], int g, int r16, Side side = Side()) {
    // B operands are read from LDS one step ahead of the MFMAs that consume them
    constexpr int STEPS = 4 * SLJ;
    float b[2][NT];
    load_b<NT>(wb + (4 * g) * COUT, r16, b[0]);
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
        const int jj = st >> 2, e = st & 3;
        if (st + 1 < STEPS) {
            const int jn = (st + 1) >> 2, en = (st + 1) & 3;
            load_b<NT>(wb + (16 * jn + 4 * g + en) * COUT, r16, b[(st + 1) & 1]);
        }
        // nothing may cross: the next step's LDS reads are ISSUED before this step's MFMAs (hipcc otherwise
        // sinks them behind the MFMAs and waits lgkmcnt(0) with the matrix pipe idle)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MS; m++) {
            const float av = e == 0 ? a[m][jj].x : e == 1 ? a[m][jj].y : e == 2 ? a[m][jj].z : a[m][jj].w;
#pragma unroll
            for (int n = 0; n < NT; n++) {
#ifdef SEI_ABL_NOMFMA  // diagnostic: one VALU fma instead of the MFMA (keeps the operand dependencies)
                acc[m][n][0] = fmaf(av, b[st & 1][n], acc[m][n][0]);
#elif defined(SEI_ABL_MFMA_FRACTION)  // diagnostic: only every SEI_ABL_MFMA_FRACTION-th MFMA step is issued (results wrong, timing
                                      // only): what would a cheaper matrix product (split-bf16 MFMAs) buy with everything else in place?
                if (st % SEI_ABL_MFMA_FRACTION == 0) acc[m][n] = mfma16(av, b[st & 1][n], acc[m][n]);
                else asm volatile("" ::"v"(av), "v"(b[st & 1][n]));
#else
                acc[m][n] = mfma16(av, b[st & 1][n], acc[m][n]);
#endif
            }
        }
        if constexpr (!std::is_same<Side, NoSide>::value) {
            int tok = 0;
            asm volatile("" : "+s"(tok) : "v"(acc[MS - 1][NT - 1]));
            side(st, tok);
        }
    }
}

template <int MAXS>
struct RuleIdx {
    int src[MAXS];        // lane r16: src row of rule slot r16 (A operand row); RAW load, see valid()
    uint32_t d4[MAXS];    // lane group g: packed dst rows (tile-local) of result rows 4g..4g+3
    int ms;               // sub-chunks this wave owns for the offset (wave-uniform); slots m >= ms hold junk
    __device__ __forceinline__ int row(int m) const { return m < ms ? src[m] : -1; }
};

constexpr int occ_for(int maxs, int nt, int slj) {
    // waves per SIMD the register allocator should aim for (LDS decides the rest):
    // accumulators + saved tile values + two A fragment sets + B values + ~80 registers of addressing / indices
    const int est = maxs * nt * 4 + nt * 4 + 2 * maxs * slj * 4 + nt + 80;
    // (NT=4, MAXS=2, SLJ=2 lands on 3 waves and spills 5 VGPRs; at 2 waves without the spill it measured 72.6 vs 82.1
    // TFLOP/s at level 3 of the 64-event batch -- occupancy wins)
    return est <= 132 ? 4 : (est <= 165 ? 3 : (est <= 250 ? 2 : 1));
}

// NT = 16-column tiles per wave, CS = waves sharing one sub-chunk (column split), SLJ = slice / 16,
// MAXS = sub-chunks of one offset a wave can own.  COUT = 16*NT*CS, tile rows TR = 16*MAXS*(4/CS).
// LDS: acc[TR][COUT] | wbuf[2][SL][COUT] | seg[K+1]
constexpr int APAD = 0;  // the shared tile helpers (conv_tile.h) assume an unpadded accumulator row stride

// Eight-wave workgroups: two per CU (4 waves per SIMD, 128 registers) always fit; where the LDS of THREE fits (the 128- and
// 160-column layers in two column blocks: 40 / 50 KB) the third one is worth 15 % (profiles/r04_conv_tile_experiments.log: every
// source variant that pushed the 128-column kernel from 80 to 86+ registers lost exactly that), so the allocator is held to
// 6 waves per SIMD = 80 registers there instead of landing on the step by luck.
constexpr int occ8_for(int nt, int cs, int slj, int maxs) {
    const int cout = 16 * nt * cs, tr = 16 * maxs * (8 / cs);
    return 3 * (tr * cout + 2 * 16 * slj * cout) * 4 + 3 * 128 <= 160 * 1024 ? 6 : 4;
}

template <int NT, int CS, int SLJ, int MAXS, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? occ8_for(NT, CS, SLJ, MAXS) : occ_for(MAXS, NT, SLJ))) void conv_tile_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int CIN, int n_tiles,
    const float *__restrict__ init, int ldy, int blocks, double *__restrict__ stats, ConvEpilogue ep) {
    extern __shared__ float lds[];
    constexpr int U = NW / CS;  // units (groups of CS waves) dealing sub-chunks
    constexpr int TR = 16 * MAXS * U;
    constexpr int SL = 16 * SLJ;
    constexpr int NC = NT * 16;
    constexpr int COUT = NC * CS;
    constexpr int PIECES = SL * COUT / 256;  // 1-KiB LDS-DMA pieces per weight slice (contiguous in W)
    constexpr int ASTRIDE = COUT + APAD;     // accumulator row stride (floats)
    // work item = (tile, column block), XCD-aware order (xcd_item, conv_tile.h): neighbouring tiles -- which gather many
    // of the same src rows -- and the column blocks of a tile -- which gather exactly the same ones -- run on one XCD
    // and share its L2 (speed only)
    const int vid = xcd_item(n_tiles * blocks);
    if (vid >= n_tiles * blocks) return;
    const int tile = vid / blocks, cblk = vid - tile * blocks;
    W += (size_t)cblk * K * CIN * COUT;
    if (bias) bias += cblk * COUT;
    const int ycol0 = cblk * COUT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = wave / CS, cg = wave - unit * CS;
    const int col0 = cg * NC;
    const int r16 = lane & 15, g = lane >> 4;
    float *acc_tile = lds;
    float *wbuf = lds + (size_t)TR * ASTRIDE;
    int *seg = reinterpret_cast<int *>(wbuf + 2 * SL * COUT);

    STAMP_DECL;
    TRACE_DECL;
    for (int i = threadIdx.x; i <= K; i += 64 * NW) seg[i] = tile_seg[(size_t)tile * (K + 1) + i];
    tile_init<TR, COUT, 64 * NW>(acc_tile, bias, init, (int64_t)tile * TR, n_dst, ldy, ycol0);
    __syncthreads();

    const int64_t sub0 = tile_sub_start[tile];
    const int NS = CIN / SL;  // slices per offset

    auto next_k = [&](int k) {  // next offset with rules in this tile (K if none)
        if (k < K) k++;
        while (k < K && seg[k + 1] == seg[k]) k++;
        return __builtin_amdgcn_readfirstlane(k);
    };
    auto load_idx = [&](int k) {  // clamped: k == K re-reads the last offset's (unused) slots
        RuleIdx<MAXS> r;
        const int kk = k < K ? k : K - 1;
        int s0 = __builtin_amdgcn_readfirstlane(seg[kk]), ns = __builtin_amdgcn_readfirstlane(seg[kk + 1]) - s0;
        if (k >= K) ns = 0;
        int ms = 0;
#pragma unroll
        for (int m = 0; m < MAXS; m++) {
            const int s = unit + m * U;
            const bool valid = s < ns;
            const int64_t sb = (sub0 + s0 + (valid ? s : 0)) * SUB_SLOTS;
            // the loaded values are NOT touched here: selecting on them would stall this wave for a full
            // memory latency per offset; validity is applied where they are consumed (2 offsets later)
            r.src[m] = sub_src[sb + r16];
            r.d4[m] = *reinterpret_cast<const uint32_t *>(sub_dst + sb + 4 * g);
            if (valid) ms = m + 1;
        }
        r.ms = __builtin_amdgcn_readfirstlane(ms);
        return r;
    };

    int k = next_k(-1);
    if (k < K) {
        int k1 = next_k(k), k2 = next_k(k1);
        RuleIdx<MAXS> ic = load_idx(k), in1 = load_idx(k1), in2 = load_idx(k2);
        stage_slice<PIECES, NW>(W + (size_t)k * CIN * COUT, wbuf, wave, lane);
        float4 a_c[MAXS][SLJ], a_n[MAXS][SLJ];
#pragma unroll
        for (int m = 0; m < MAXS; m++)
#pragma unroll
            for (int jj = 0; jj < SLJ; jj++) {
                a_c[m][jj] = load_a(X, ic.row(m), CIN, 16 * jj + 4 * g);
                a_n[m][jj] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        int phase = 0;
        STAMP(10);
        while (k < K) {
            const int kw = k1 < K ? k1 : k;  // weights to prefetch after the last slice (unused at the very end)
            f32x4 acc[MAXS][NT];
            float *prow[MAXS][4];
            int nv[MAXS];
#pragma unroll
            for (int m = 0; m < MAXS; m++) {
                // padding sits at the end of a sub-chunk: slots [0, nv) are rules
                nv[m] = __popcll(__ballot(ic.row(m) >= 0) & 0xFFFFull);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    prow[m][i] = acc_tile + ((ic.d4[m] >> (8 * i)) & 0xFF) * ASTRIDE + col0 + r16;
            }
            for (int s = 0; s < NS; s++) {
                // Barrier: this phase's weight slice (LDS-DMA issued one phase ago) has landed in every wave and the
                // previous phase is fully consumed.  __syncthreads() would wait vmcnt(0), i.e. also for the index
                // loads of offset k+2 issued just above -- a full memory latency per offset.  Those 2*MAXS loads are
                // the youngest in flight (vmcnt retires in order), so leave exactly them outstanding.
                if (s == 0 && phase > 0)
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::"n"(2 * MAXS) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::: "memory");
                STAMP(0);
                TRACE(0);
                if (s == 0) {
                    // the offset's products are summed from zero (see the write-back below)
#pragma unroll
                    for (int m = 0; m < MAXS; m++)
#pragma unroll
                        for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                STAMP(1);
                // prefetch the next phase's weight slice and A fragments while this one computes
                const bool same_k = s + 1 < NS;
#ifndef SEI_ABL_NODMA
                stage_slice<PIECES, NW>(W + ((size_t)(same_k ? k : kw) * CIN + (size_t)(same_k ? s + 1 : 0) * SL) * COUT,
                                    wbuf + (size_t)((phase + 1) & 1) * SL * COUT, wave, lane);
#endif
                const int aoff = (same_k ? (s + 1) * SL : 0) + 4 * g;
#ifdef SEI_TILE_SKIP  // experiment: gather only the sub-chunks the wave owns in that phase
                const int msn = same_k ? ic.ms : in1.ms;
#else
                const int msn = MAXS;
#endif
#pragma unroll
                for (int m = 0; m < MAXS; m++)
                    if (m == 0 || m < msn) {
#pragma unroll
                        for (int jj = 0; jj < SLJ; jj++)
                            a_n[m][jj] = load_a(X, same_k ? ic.row(m) : in1.row(m), CIN, aoff + 16 * jj);
                    }
                const float *wb = wbuf + (size_t)(phase & 1) * SL * COUT + col0;
                STAMP(2);
                TRACE(1);
                if (MAXS == 1) {
                    if (ic.ms == 1) compute_phase<NT, SLJ, 1, MAXS, COUT>(wb, a_c, acc, g, r16);
                } else if (MAXS == 2) {
                    if (ic.ms == 1) compute_phase<NT, SLJ, 1, MAXS, COUT>(wb, a_c, acc, g, r16);
                    else if (ic.ms == 2) compute_phase<NT, SLJ, (MAXS >= 2 ? 2 : 1), MAXS, COUT>(wb, a_c, acc, g, r16);
                } else {
                    switch (ic.ms) {
                        case 1: compute_phase<NT, SLJ, 1, MAXS, COUT>(wb, a_c, acc, g, r16); break;
                        case 2: compute_phase<NT, SLJ, (MAXS >= 2 ? 2 : 1), MAXS, COUT>(wb, a_c, acc, g, r16); break;
                        case 3: compute_phase<NT, SLJ, (MAXS >= 3 ? 3 : 1), MAXS, COUT>(wb, a_c, acc, g, r16); break;
                        case 4: compute_phase<NT, SLJ, (MAXS >= 4 ? 4 : 1), MAXS, COUT>(wb, a_c, acc, g, r16); break;
                        default: break;
                    }
                }
                STAMP(3);
                TRACE(2 | (ic.ms << 2));
                STAMP_ADD(8, 1);
                STAMP_ADD(9, ic.ms * NT * 4 * SLJ);
#pragma unroll
                for (int m = 0; m < MAXS; m++)
#pragma unroll
                    for (int jj = 0; jj < SLJ; jj++) a_c[m][jj] = a_n[m][jj];
                phase++;
                STAMP(4);
            }
            // D[row = 4g+i][col = r16] belongs to rule slot 4g+i.  Rows of one offset are distinct and the previous
            // offset's updates are behind the barrier:  tile = tile + partial  is a plain LDS read-modify-write --
            // no atomics (LDS float atomics cost ~150 cycles per wave-instruction on gfx950), fixed summation
            // order, and the fp32 chains stay short (nIn/4 MFMA steps, then one add per offset) like the
            // reference's per-offset  out[rows] += in[rows] @ W[k].  All reads of a sub-chunk are issued before
            // its adds (one LDS latency per sub-chunk and offset, covered by the CU's other waves).
#ifdef SEI_ABL_NORMW  // diagnostic: accumulators are not added into the LDS tile
            if (k == 1000000)
#endif
#pragma unroll
            for (int m = 0; m < MAXS; m++) {
                if (m < ic.ms) {  // wave-uniform: one branch per sub-chunk, loads inside are unconditional
                    f32x4 tv[NT];
#pragma unroll
                    for (int n = 0; n < NT; n++)
#pragma unroll
                        for (int i = 0; i < 4; i++) tv[n][i] = prow[m][i][16 * n];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (4 * g + i < nv[m]) {
#pragma unroll
                            for (int n = 0; n < NT; n++) prow[m][i][16 * n] = tv[n][i] + acc[m][n][i];
                        }
                    }
                }
            }
            STAMP(5);
            TRACE(3);
            ic = in1;
            in1 = in2;
            k = k1;
            k1 = k2;
            k2 = next_k(k2);
            in2 = load_idx(k2);  // two offsets ahead: its latency never meets a barrier
            STAMP(6);
        }
    }
    __syncthreads();
    const int64_t row0 = (int64_t)tile * TR;
    tile_store<TR, COUT, 64 * NW>(acc_tile, Y, row0, n_dst, ldy, ycol0, ep);
    if (stats) tile_stats<TR, COUT, 64 * NW>(acc_tile, stats, tile, row0, n_dst, ldy, ycol0);
    STAMP(11);
    STAMP_ADD(7, 1);
    STAMP_FLUSH;
    TRACE_END;
}

// ------------------------------------------------------------------------------------------ two-offset rounds
// The kernel above deals the sub-chunks of ONE offset to its eight waves per phase; a 128-row tile of the DUNE batch holds
// ~4.4 of them per offset, so on average 3.6 waves idle at every phase barrier (profiles/r04_conv_tile_cu_timeline_*).
// Here the tile's sub-chunks are walked as one stream in ROUNDS of up to NW consecutive sub-chunks that may span TWO
// consecutive (nonempty) offsets a < b: wave w takes sub-chunk q0 + w and reads its B operands from the weight slice of the
// offset that sub-chunk belongs to.  Both offsets' 16-plane slices are staged per phase (2 offsets x 2 buffers x 16 x COUT
// floats: the LDS of one double-buffered 32-plane slice), a round is nIn / 16 phases, and a round that cannot be filled
// from two offsets is cut short (never three: a third slice to stage per phase costs more than its fill returns,
// profiles/r04_conv_round_ab.log).  The sums keep their order: a sub-chunk is still one MFMA chain over the input planes
// added to the tile once, and the sub-chunks of a are added before those of b (which wait for the next round's first
// barrier, see round_begin); rows of one offset are distinct, so results are bit-identical to the kernel above.
template <class F, int... S>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {   // f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct Round {
    int q0, q1;   // tile-relative sub-chunks [q0, q1)
    int a, b;     // their offsets: a for q < ea, b from ea on (b == a: a alone)
    int ea;
};

// NJ == 0: the A fragment of the next phase is gathered per phase, as in the kernel above.  Memory operations retire in order,
// so that gather has ONE phase to land before the next barrier's vmcnt wait can see the weight slice requested behind it --
// and a 16-plane phase is shorter than a gather's latency: every phase waits for one.  NJ > 0 (nIn == 16 NJ): the fragments
// of a sub-chunk live in ONE register set that is refilled a phase after use (see the loop): two phases to land.
// SLJ: a slice is 16 SLJ input planes (the refilled form groups SLJ fragments per phase; the per-phase form takes SLJ == 1)
template <int NT, int NW, int NJ, int SLJ = 1>
__global__ __launch_bounds__(64 * NW, (NJ > 0 ? 4 : occ8_for(NT, 1, 2, 1))) void conv_round_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int CIN, int n_tiles,
    const float *__restrict__ init, int ldy, int blocks, double *__restrict__ stats, ConvEpilogue ep) {
    extern __shared__ float lds[];
    static_assert(SLJ == 1 || (NJ > 0 && NJ % SLJ == 0), "wider slices: refilled form only");
    constexpr int TR = 16 * NW, SL = 16 * SLJ, COUT = NT * 16, PIECES = SL * COUT / 256;
    const int vid = xcd_item(n_tiles * blocks);
    if (vid >= n_tiles * blocks) return;
    const int tile = vid / blocks, cblk = vid - tile * blocks;
    W += (size_t)cblk * K * CIN * COUT;
    if (bias) bias += cblk * COUT;
    const int ycol0 = cblk * COUT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    float *acc_tile = lds;
    float *wbuf = lds + (size_t)TR * COUT;   // [buffer][offset half][SL][COUT]
    int *seg = reinterpret_cast<int *>(wbuf + 4 * SL * COUT);
    CLK_BEGIN;

    TRACE_DECL;
    for (int i = threadIdx.x; i <= K; i += 64 * NW) seg[i] = tile_seg[(size_t)tile * (K + 1) + i];
    tile_init<TR, COUT, 64 * NW>(acc_tile, bias, init, (int64_t)tile * TR, n_dst, ldy, ycol0);
    __syncthreads();

    const int64_t sub0 = tile_sub_start[tile];
    const int NS = CIN / SL;
    const int n = __builtin_amdgcn_readfirstlane(seg[K]);

    auto plan = [&](int q0, int c) {  // the round starting at sub-chunk q0; c = an offset at or below the one holding q0
        Round r;
        r.q0 = q0;
        if (q0 >= n) { r.q1 = q0; r.a = r.b = c < K ? c : K - 1; r.ea = q0; return r; }
        while (__builtin_amdgcn_readfirstlane(seg[c + 1]) <= q0) c++;
        r.a = c;
        r.ea = __builtin_amdgcn_readfirstlane(seg[c + 1]);
        r.b = c;
        if (q0 + NW <= r.ea) { r.q1 = q0 + NW; return r; }
        int b = c + 1;
        while (b < K && __builtin_amdgcn_readfirstlane(seg[b + 1]) == r.ea) b++;   // empty offsets end where a ends
        if (b >= K) { r.q1 = r.ea; return r; }
        r.b = b;
        const int eb = __builtin_amdgcn_readfirstlane(seg[b + 1]);
        r.q1 = q0 + NW < eb ? q0 + NW : eb;
        return r;
    };
    struct Idx { int src; uint32_t d4; };
    auto load_idx = [&](const Round &r) {  // clamped: never branch around a load; validity is applied where it is used
        const int q = r.q0 + wave;
        const int qc = q < r.q1 ? q : (r.q0 < n ? r.q0 : n - 1);
        const int64_t sb = (sub0 + qc) * SUB_SLOTS;
        Idx v;
        v.src = sub_src[sb + r16];
        v.d4 = *reinterpret_cast<const uint32_t *>(sub_dst + sb + 4 * g);
        return v;
    };
    auto stage = [&](const Round &r, int s, int buf) {
#ifdef SEI_ABL_NODMA   // diagnostic: the weight slices are never staged (results wrong, timing only)
        if (r.q0 >= 0) return;
#endif
        float *dst = wbuf + (size_t)buf * 2 * SL * COUT;
        stage_slice<PIECES, NW>(W + ((size_t)r.a * CIN + (size_t)s * SL) * COUT, dst, wave, lane);
        if (r.b != r.a) stage_slice<PIECES, NW>(W + ((size_t)r.b * CIN + (size_t)s * SL) * COUT, dst + SL * COUT, wave, lane);
        asm volatile("" ::: "memory");   // the slices are requested BEFORE the loads that follow (the vmcnt(2) below counts on it)
    };

    if (n > 0) {
        Round r0 = plan(0, 0), r1 = plan(r0.q1, r0.b), r2 = plan(r1.q1, r1.b);
        Idx ic = load_idx(r0), i1 = load_idx(r1), i2 = load_idx(r2);
        stage(r0, 0, 0);
        constexpr int NA = NJ > 0 ? NJ / SLJ : 1;   // phases of a round in the refilled form
        float4 a_c[NA][1][SLJ], a_n[1][1][1];
#pragma unroll
        for (int j = 0; j < NA; j++)
#pragma unroll
            for (int jj = 0; jj < SLJ; jj++)
                a_c[j][0][jj] = load_a(X, r0.q0 + wave < r0.q1 ? ic.src : -1, CIN, 16 * (SLJ * j + jj) + 4 * g);
        int phase = 0;
        // tile rows += partial sums of the sub-chunk a wave held (d4: its packed dst rows, nv: rule slots in it)
        f32x4 acc[1][NT];
        auto rmw = [&](uint32_t d4, int nv) {
#ifdef SEI_ABL_NORMW   // diagnostic: the sums are not added into the LDS tile (results wrong, timing only)
            if (d4 != 0xFFFFFFFFu) {
#pragma unroll
                for (int nn = 0; nn < NT; nn++) asm volatile("" ::"v"(acc[0][nn]));
                return;
            }
#endif
            float *prow[4];
#pragma unroll
            for (int i = 0; i < 4; i++) prow[i] = acc_tile + ((d4 >> (8 * i)) & 0xFF) * COUT + r16;
            f32x4 tv[NT];
#pragma unroll
            for (int nn = 0; nn < NT; nn++)
#pragma unroll
                for (int i = 0; i < 4; i++) tv[nn][i] = prow[i][16 * nn];
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (4 * g + i < nv) {
#pragma unroll
                    for (int nn = 0; nn < NT; nn++) prow[i][16 * nn] = tv[nn][i] + acc[0][nn][i];
                }
        };
        // The sub-chunks of offset b are added after those of a.  Instead of a barrier of its own between the two groups, a wave
        // that held a b sub-chunk keeps its sums and adds them BEHIND THE FIRST BARRIER OF THE NEXT ROUND (which is there anyway),
        // before it clears its accumulators; the a sub-chunks of that round are added six barriers later.
        bool pend_b = false;
        uint32_t pend_d4 = 0;
        int pend_nv = 0;
        auto round_begin = [&]() {   // first thing behind a round's first barrier
            if (pend_b) rmw(pend_d4, pend_nv);
            pend_b = false;
#pragma unroll
            for (int nn = 0; nn < NT; nn++) acc[0][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        while (r0.q0 < n) {
            const bool valid = r0.q0 + wave < r0.q1;           // wave-uniform
            const int in_b = (r0.q0 + wave >= r0.ea) ? 1 : 0;  // this wave's sub-chunk belongs to offset b
            const int rowc = valid ? ic.src : -1, rown = (r1.q0 + wave < r1.q1) ? i1.src : -1;
            Round r3 = r2;
            Idx i3 = i2;
            if constexpr (NJ > 0) {
                constexpr int NP = NJ / SLJ;          // phases of a round
                constexpr int IDXP = NP >= 4 ? 2 : 1; // the phase that requests the index words of round r + 3
                static_assert(NJ == 0 || NP >= 3, "the schedule below uses three phases");
                // ONE set of A fragments: a_c[s] holds the SLJ fragments of phase s of the wave's current sub-chunk until that
                // phase has used them and is refilled in the NEXT phase -- with those of phase s of the next round's sub-chunk
                // (phase NP-1: of the current one, it is used last) -- so a gather has NP-1 phases to be used and, memory
                // operations retiring in order, two phases to land (the barrier of phase p+2 waits for the slice requested in
                // phase p+1, which sits behind it in the queue).
                // Queue of one phase: [slice a][slice b][SLJ gathers]([index loads x 2] in phase IDXP).
                static_for<NP>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if (phase == 0)
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::: "memory");
                    else if (s == (IDXP + 1) % NP)
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::"n"(SLJ + 2) : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::"n"(SLJ) : "memory");
                    TRACE(0);
                    if (s == 0) round_begin();
                    {
                        // the gathers of two phases ago (into a_c[s-3]) are in: tell hipcc here, so that it plants no wait of its own
                        constexpr int jg = (s + 3 * NP - 3) % NP;
#pragma unroll
                        for (int jj = 0; jj < SLJ; jj++)
                            asm volatile("" : "+v"(a_c[jg][0][jj].x), "+v"(a_c[jg][0][jj].y), "+v"(a_c[jg][0][jj].z), "+v"(a_c[jg][0][jj].w));
                    }
                    const bool same = s + 1 < NP;
                    stage(same ? r0 : r1, same ? s + 1 : 0, (phase + 1) & 1);
                    {
                        constexpr int jr = (s + NP - 1) % NP;   // the fragments the previous phase used
#pragma unroll
                        for (int jj = 0; jj < SLJ; jj++)
                            a_c[jr][0][jj] = load_a(X, s == 0 ? rowc : rown, CIN, 16 * (SLJ * jr + jj) + 4 * g);
                        asm volatile("" ::: "memory");
                    }
                    if (s == IDXP) {
                        // the index words of round r + 3 are requested mid-round into registers of their own and move over at
                        // the end of the round, long after they have arrived (requested at the end of the round, hipcc loaded
                        // them into spare registers and waited vmcnt(0) for the copy right there)
                        r3 = plan(r2.q1, r2.b);
                        i3 = load_idx(r3);
                        asm volatile("" ::: "memory");
                    }
                    TRACE(1);
                    if (valid)
                        compute_phase<NT, SLJ, 1, 1, COUT>(wbuf + (size_t)((phase & 1) * 2 + in_b) * SL * COUT, a_c[s], acc, g, r16);
                    TRACE(2 | ((valid ? 1 : 0) << 2));
                    phase++;
                });
            } else
            for (int s = 0; s < NS; s++) {
                // as in the kernel above: the slices staged one phase ago have landed in every wave; only the two index loads of
                // round r + 2 (the youngest operations of the queue) stay outstanding at a round's first phase
                if (s == 0 && phase > 0)
                    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::: "memory");
                TRACE(0);
                if (s == 0) round_begin();
                const bool same = s + 1 < NS;
                stage(same ? r0 : r1, same ? s + 1 : 0, (phase + 1) & 1);
                a_n[0][0][0] = load_a(X, same ? rowc : rown, CIN, (same ? (s + 1) * SL : 0) + 4 * g);
                TRACE(1);
                if (valid)
                    compute_phase<NT, 1, 1, 1, COUT>(wbuf + (size_t)((phase & 1) * 2 + in_b) * SL * COUT, a_c[0], acc, g, r16);
                TRACE(2 | ((valid ? 1 : 0) << 2));
                a_c[0][0][0] = a_n[0][0][0];
                phase++;
            }
            const int nv = __popcll(__ballot(rowc >= 0) & 0xFFFFull);
            if (valid && !in_b) rmw(ic.d4, nv);
            if (valid && in_b) {
                pend_b = true;
                pend_d4 = ic.d4;
                pend_nv = nv;
            }
            TRACE(3);
            r0 = r1;
            r1 = r2;
            ic = i1;
            i1 = i2;
            if constexpr (NJ > 0) {
                r2 = r3;
                i2 = i3;
            } else {
                r2 = plan(r2.q1, r2.b);
                i2 = load_idx(r2);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::: "memory");
        if (pend_b) rmw(pend_d4, pend_nv);
    }
    __syncthreads();
    const int64_t row0 = (int64_t)tile * TR;
    tile_store<TR, COUT, 64 * NW>(acc_tile, Y, row0, n_dst, ldy, ycol0, ep);
    if (stats) tile_stats<TR, COUT, 64 * NW>(acc_tile, stats, tile, row0, n_dst, ldy, ycol0);
    TRACE_END;
    CLK_END(0);
}

// ------------------------------------------------------------------------------------------ multi-offset rounds  [r5]
// The rounds above may span TWO offsets.  A sixteen-wave round on a 256-row tile then fills 0.65 at level 1 of the DUNE batch
// (5.7 sub-chunks per tile and nonempty offset), 0.83 / 0.88 at levels 2 / 3; spanning up to FOUR offsets: 0.87 / 0.95 / 0.96
// (tools/round_stats.py --tile 256).  What kept the rounds at two offsets was the write-back: the sub-chunks of offset b may
// touch rows that those of offset a touch, so the groups of a round are added to the LDS tile in offset order, a barrier
// between any two -- and a barrier per group costs more than the fill returns (profiles/r04_conv_round_ab.log).  Here the
// barriers that order the groups are the PHASE barriers of the next round, which are there anyway:
//   group 0 of round r is added at the end of the round (behind its last MFMA, as before);
//   group g >= 1 moves to a second register set and is added behind barrier g-1 of round r+1, i.e. in that round's phase g-1.
// With MO <= NP (offsets per round <= phases per round) the last group of round r lands in phase MO-2 <= NP-2 of round r+1,
// one barrier before group 0 of round r+1 is added: every pair of groups of different offsets is barrier-separated and in
// offset order, groups of one offset touch distinct rows -- bit-identical to the kernels above.  Per phase the slices of all
// MO offsets are staged (MO x 2 buffers x 16 SLJ x COUT floats), the pieces dealt over all sixteen waves; the segment table
// lives in a lane register (K + 1 <= 64) with a bit mask of the nonempty offsets, so that planning a round costs no LDS
// round trips.  Refilled A fragments and the counted vmcnt schedule are those of conv_round_mfma_kernel<.., NJ > 0>.
// Diagnostic build only (make EXTRA="-DSEI_EXPERIMENTS -DSEI_STAMPS"; tools/mround_stamps.py): EVERY wave adds the s_memtime
// ticks it spends in each part of the kernel below to g_mstamps[wave][slot].  The product build compiles none of it.
#ifdef SEI_STAMPS
__device__ unsigned long long g_mstamps[16][16];
#define MST_DECL unsigned long long mst_t = __builtin_amdgcn_s_memtime(), mst_acc[16] = {}
#define MST(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
                    __builtin_amdgcn_sched_barrier(0); mst_acc[i] += n_ - mst_t; mst_t = n_; } while (0)
#define MST_ADD(i, v) mst_acc[i] += (v)
#define MST_FLUSH do { if (lane == 0) { for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_mstamps[wave & 15][i_], mst_acc[i_]); } } while (0)
#else
#define MST_DECL
#define MST(i)
#define MST_ADD(i, v)
#define MST_FLUSH
#endif

// ---- fp32 products from THREE-WAY bf16 splits on the bf16 matrix pipe (opt-in: sei_set_product_mode(1)) [r5]
// x = x1 + x2 + x3 EXACTLY, each part the bf16 nearest to what is left (the residuals are exact fp32 subtractions, the third one
// is itself a bf16 number; bf16 has fp32's exponent range, so no scaling; |x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|).  a b = sum of nine exact bf16 x bf16 products, of which
// the six largest are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (the three dropped ones are below 2^-23 |a b|: the size of
// the rounding error of one fp32 product).  16 cycles per 16 x 16 x 32 block and part pair: 96 cycles per 32 input planes and
// column tile against 256 on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: 8 x 32).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_bf16(u32x4_t a, u32x4_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// (x, y) -> three words of two packed bf16 each (x in the low half): parts 1, 2, 3.  Each part is the round-to-nearest-even
// bf16 of what is left (v_cvt_pk_bf16_f32); the residuals are exact fp32 subtractions and the third residual IS a bf16 number.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float x, float y) {
    const f32x2_t v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split_pair(float x, float y, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = cvt_pk_bf16(x, y);
    const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = cvt_pk_bf16(rx, ry);
    const float rx2 = rx - __uint_as_float(p2 << 16), ry2 = ry - __uint_as_float(p2 & 0xFFFF0000u);
    p3 = cvt_pk_bf16(rx2, ry2);
}
// One 32-plane phase for one sub-chunk.  a[0], a[1]: the lane's 8 consecutive planes 8 g .. 8 g + 7 of the slice for rule row r16
// (the A layout of the 16 x 16 x 32 block).  wb: this offset's slice in LDS, [part][column tile][lane][8 bf16] (weights_split_kernel):
// one ds_read_b128 per part and column tile.  Small terms first.
template <int NT, int BB>
__device__ __forceinline__ void compute_phase_split(const float *__restrict__ wb, const float4 (&a)[1][2], f32x4 (&acc)[1][NT], int lane) {
    unsigned a1[4], a2[4], a3[4];
    split_pair(a[0][0].x, a[0][0].y, a1[0], a2[0], a3[0]);
    split_pair(a[0][0].z, a[0][0].w, a1[1], a2[1], a3[1]);
    split_pair(a[0][1].x, a[0][1].y, a1[2], a2[2], a3[2]);
    split_pair(a[0][1].z, a[0][1].w, a1[3], a2[3], a3[3]);
    const u32x4_t A1 = {a1[0], a1[1], a1[2], a1[3]}, A2 = {a2[0], a2[1], a2[2], a2[3]}, A3 = {a3[0], a3[1], a3[2], a3[3]};
    const u32x4_t *bp = reinterpret_cast<const u32x4_t *>(wb) + lane;
    // BB == 2: the B parts of the NEXT column tile are read while this one multiplies (24 registers); BB == 1: the reads of a column
    // tile sit behind the previous tile's MFMAs (12 registers) -- at 192 input planes, where the fragments of a whole row leave no
    // room (15 registers spilled with two sets: 110 against 122 fp32-equivalent TFLOP/s at level 5; at 64 / 128 planes two sets
    // are 3-4 % faster: profiles/r05_split_products_ab.log)
    u32x4_t b[BB][3];
#pragma unroll
    for (int q = 0; q < 3; q++) b[0][q] = bp[(q * NT + 0) * 64];
#pragma unroll
    for (int n = 0; n < NT; n++) {
        if (BB == 2 && n + 1 < NT) {
#pragma unroll
            for (int q = 0; q < 3; q++) b[(n + 1) & 1][q] = bp[(q * NT + n + 1) * 64];
        }
        if (BB == 2) __builtin_amdgcn_sched_barrier(0);   // the next column tile's reads are issued before this one's MFMAs
        const int cb = BB == 2 ? (n & 1) : 0;
        f32x4 c = acc[0][n];
        c = mfma_bf16(A3, b[cb][0], c);
        c = mfma_bf16(A1, b[cb][2], c);
        c = mfma_bf16(A2, b[cb][1], c);
        c = mfma_bf16(A2, b[cb][0], c);
        c = mfma_bf16(A1, b[cb][1], c);
        c = mfma_bf16(A1, b[cb][0], c);
        acc[0][n] = c;
        if (BB == 1 && n + 1 < NT) {
#pragma unroll
            for (int q = 0; q < 3; q++) b[0][q] = bp[(q * NT + n + 1) * 64];
        }
    }
}

struct MRound {
    int q0;          // first tile-relative sub-chunk
    int m;           // offsets (groups) in the round, 0 beyond the end of the tile
    uint32_t offs;   // byte j: offset of group j (j >= m: the last one)
    uint32_t ends;   // byte j: end of group j relative to q0 (j >= m: the round's sub-chunk count)
};

// NB: weight buffers (NP >= 3 only).  2: the slices of phase p + 1 are staged in phase p and have ONE phase to land; 3: those of
// phase p + 2, TWO phases (the third buffer is the LDS the 128 / 160 / 192-column plans leave unused; every wave then issues the
// same number of pieces per phase -- surplus ones repeat a neighbour's -- so that the counted vmcnt waits stay exact).
// LATE (NB == 3 only; experiment builds): waves NW/2 .. NW-1 issue their requests of a phase BEHIND its MFMA block instead of in
// front of it.  After a barrier all sixteen waves issue their staging pieces, gathers and index loads at once; the CU's one
// vector-memory path takes them one after the other, and the waves served last start their MFMAs last (in-kernel stamps: the
// upper waves spend 35-40 % of their time in the issue section, the lower ones 8 %).  With every request two phases ahead of
// its use (third weight buffer, refilled fragments) half the waves can compute first.
// PERSIST [r5]: one workgroup per CU walks MANY work items (tile, column block).  With one sixteen-wave workgroup per CU nothing
// runs beside a workgroup's prologue -- segment table -> round plans -> index words -> first gathers, three dependent trips to
// memory, ~5 % of a level-1 tile -- so the NEXT item's ticket (one atomic counter per XCD, `work`: the items keep the XCD-aware
// order of xcd_item), its segment table, its first three round plans and their index words are requested while the current tile
// computes (one step per round, `prefetch_step`); what stays exposed at a tile boundary is the tile's store, the next tile's
// initialisation and ONE trip (the first gathers).  Same rounds, same order of sums: bit-identical.
// SPLIT: the products of a phase as three-way bf16 splits (compute_phase_split; SLJ == 2: a phase is one 32-plane block).  W then points to
// the split weights [block][K][nIn / 32][3][NT][64][8 bf16] and a slice is 3 NT KiB instead of 128 COUT bytes per plane pair.
template <int NT, int NW, int NJ, int SLJ, int MO, int NB = 2, bool LATE = false, bool PERSIST = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW, NW == 12 ? 3 : 4) void conv_mround_mfma_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int CIN, int n_tiles,
    const float *__restrict__ init, int ldy, int blocks, double *__restrict__ stats, ConvEpilogue ep, int *__restrict__ work) {
    extern __shared__ float lds[];
    constexpr int NP = NJ / SLJ;   // phases of a round
#ifdef SEI_NO_SIDE_WORK   // experiment builds: the schedule before the side work
    constexpr bool SIDE = false;
#else
    constexpr bool SIDE = NP == 2 && MO == 2 && !SPLIT && !PERSIST;   // write-back + planning inside the MFMA block
#endif
    static_assert(NJ % SLJ == 0 && NP >= 2, "NP >= 3: the refilled-fragment schedule; NP == 2: two long phases per round");
    static_assert(MO >= 2 && MO <= 4 && MO <= NP, "group g of a round is written back in phase g - 1 of the next one");
    static_assert(NB == 2 || (NB == 3 && NP >= 3), "the third weight buffer belongs to the refilled-fragment schedule");
    static_assert(!LATE || NB == 3, "requests behind the MFMA block need everything two phases ahead");
    static_assert(!SPLIT || (SLJ == 2 && NB == 2 && !LATE), "the split product works on 32-plane phases");
    constexpr int TR = 16 * NW, SL = 16 * SLJ, COUT = NT * 16;
    constexpr int WSLICE = SPLIT ? 3 * NT * 256 : SL * COUT;   // floats of one offset's slice in LDS
    constexpr int PIECES = WSLICE / 256;
    static_assert(WSLICE % 256 == 0, "a slice is a whole number of 1-KiB LDS-DMA pieces");
    constexpr int STAGE_PER = (MO * PIECES + NW - 1) / NW;   // pieces a wave issues per staged phase (NB == 3: exactly)
    const int n_items = n_tiles * blocks;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    float *acc_tile = lds;
    float *wbuf = lds + (size_t)TR * COUT;   // [buffer][group][SL][COUT]
    int *lds_ticket = reinterpret_cast<int *>(wbuf + (size_t)NB * MO * WSLICE);   // PERSIST: the next item's ticket, wave 0 -> all
    MST_DECL;
    CLK_BEGIN;

    // One work item's segment table: lane i holds seg[i] (i <= K < 64); bit i of `nonempty`: offset i has sub-chunks in the tile
    struct Ctx { int segv; unsigned long long nonempty; int n; int sub0; };
    struct CtxReq { int segv, segn, sub0; };
    auto ctx_request = [&](int vid) {   // loads only
        const int tl = vid / blocks;
        const int32_t *segp = tile_seg + (size_t)tl * (K + 1);
        CtxReq q;
        q.segv = segp[lane < K ? lane : K];
        q.segn = segp[lane + 1 < K ? lane + 1 : K];
        q.sub0 = tile_sub_start[tl];
        return q;
    };
    auto ctx_finish = [&](const CtxReq &q) {
        Ctx c;
        c.segv = q.segv;
        c.nonempty = __ballot(q.segn > q.segv);
        c.n = __builtin_amdgcn_readlane(q.segv, __builtin_amdgcn_readfirstlane(K));
        c.sub0 = __builtin_amdgcn_readfirstlane(q.sub0);
        return c;
    };
    auto seg_at = [&](const Ctx &cx, int i) { return __builtin_amdgcn_readlane(cx.segv, __builtin_amdgcn_readfirstlane(i)); };
    auto next_nonempty = [&](const Ctx &cx, int c) {   // smallest nonempty offset > c (K if none); c in [-1, K)
        const unsigned long long m = cx.nonempty >> (c + 1);
        return m ? c + 1 + (int)__builtin_ctzll(m) : K;
    };
    // PERSIST: tickets of this workgroup's XCD -> work items in the order of xcd_item (conv_tile.h); n_items when the XCD's share is used up
    const int xcd = blockIdx.x & 7;
    auto item_of = [&](int i) {
        const int per = (n_items + 7) >> 3;
        if (i >= per) return n_items;
        constexpr int CH = SEI_XCD_CHUNK;
        const int full = per / CH;
        int v;
        if (i < full * CH) v = ((i / CH) * 8 + xcd) * CH + i % CH;
        else v = full * CH * 8 + xcd * (per - full * CH) + (i - full * CH);
        return v < n_items ? v : n_items;
    };
    int vid;
    if constexpr (PERSIST) {
        if (threadIdx.x == 0) *lds_ticket = atomicAdd(&work[xcd], 1);
        __syncthreads();
        vid = item_of(__builtin_amdgcn_readfirstlane(*lds_ticket));
        __syncthreads();   // everybody has read the ticket before prefetch_step may overwrite it
    } else {
        vid = xcd_item(n_items);
    }
    if (vid >= n_items) return;
    Ctx cx = ctx_finish(ctx_request(vid));
    auto cnt_of = [&](const MRound &r) { return (int)((r.ends >> (8 * (MO - 1))) & 0xFF); };
    auto last_of = [&](const MRound &r) { return (int)((r.offs >> (8 * (MO - 1))) & 0xFF); };
    auto plan = [&](const Ctx &cx, int q0, int c) {   // the round starting at sub-chunk q0; c: an offset at or below the one holding q0
        const int n = cx.n;
        MRound r;
        r.q0 = q0;
        if (q0 >= n) { r.m = 0; r.offs = 0x01010101u * (uint32_t)(c < K ? c : K - 1); r.ends = 0; return r; }
        while (seg_at(cx, c + 1) <= q0) c = next_nonempty(cx, c);
        uint32_t offs = 0, ends = 0;
        int m = 0, e = 0;
        bool open = true;
#pragma unroll
        for (int j = 0; j < MO; j++) {
            if (open) {
                const int end = seg_at(cx, c + 1) - q0;
                e = end < NW ? end : NW;
                m = j + 1;
            }
            offs |= (uint32_t)c << (8 * j);
            ends |= (uint32_t)e << (8 * j);
            if (open) {
                if (e == NW) open = false;
                else {
                    const int c2 = next_nonempty(cx, c);
                    if (c2 >= K) open = false; else c = c2;
                }
            }
        }
        r.m = m; r.offs = offs; r.ends = ends;
        return r;
    };
    struct Idx { int src; uint32_t d4; };
    auto load_idx = [&](const Ctx &cx, const MRound &r) {  // clamped: never branch around a load; validity is applied where it is used
        const int q = r.q0 + wave, q1 = r.q0 + cnt_of(r);
        const int qc = q < q1 ? q : (r.q0 < cx.n ? r.q0 : (cx.n > 0 ? cx.n - 1 : 0));
        const int64_t sb = ((int64_t)cx.sub0 + qc) * SUB_SLOTS;
        Idx v;
        v.src = sub_src[sb + r16];
        v.d4 = *reinterpret_cast<const uint32_t *>(sub_dst + sb + 4 * g);
        return v;
    };
    const float *Wt = W;        // this item's column block of the weights / of the bias
    const float *biast = bias;
    int tile = 0, ycol0 = 0;
    auto stage = [&](const MRound &r, int s, int buf) {   // slice s of every offset of round r; pieces dealt over all waves
#ifdef SEI_ABL_NODMA   // diagnostic: the weight slices are never staged (results wrong, timing only)
        if (r.q0 >= 0) return;
#endif
        constexpr int TOT = MO * PIECES, PER = STAGE_PER;
        float *dst = wbuf + (size_t)buf * MO * WSLICE;
#pragma unroll
        for (int t = 0; t < PER; t++) {
            int lin = wave + t * NW;                                                  // wave-uniform
            if (NB == 3 && lin >= TOT) lin -= NW;                                     // surplus piece: the previous one again
            const int j = lin / PIECES, p = lin - j * PIECES;
            if (NB == 3 || (lin < TOT && j < r.m)) {                                  // NB == 3: always (byte j >= m repeats the last offset)
                const int off = (r.offs >> (8 * j)) & 0xFF;
                const float *gp = (SPLIT ? Wt + ((size_t)off * (CIN / 32) + (size_t)s) * WSLICE
                                         : Wt + ((size_t)off * CIN + (size_t)s * SL) * COUT) + p * 256 + lane * 4;
                float *lp = dst + (size_t)j * WSLICE + p * 256;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                                 (__attribute__((address_space(3))) void *)lp, 16, 0, 0);
            }
        }
        asm volatile("" ::: "memory");   // requested BEFORE the gathers that follow (the counted waits rely on it)
    };

    // PERSIST: the next item, requested step by step while this one computes (pf: steps done)
    int pf = 0, nvid = n_items, ticket_v = 0;
    CtxReq nq{};
    Ctx ncx{};
    MRound pr0{}, pr1{}, pr2{};
    Idx pi0{}, pi1{}, pi2{};
    auto prefetch_step = [&]() {   // wave-uniform; steps 1 and 2 are separated by a barrier (they run in different rounds)
        if (pf == 0) {
            if (threadIdx.x == 0) ticket_v = atomicAdd(&work[xcd], 1);
        } else if (pf == 1) {
            if (threadIdx.x == 0) *lds_ticket = ticket_v;
        } else if (pf == 2) {
            nvid = item_of(__builtin_amdgcn_readfirstlane(*lds_ticket));
            if (nvid < n_items) nq = ctx_request(nvid);
        } else if (pf == 3) {
            if (nvid < n_items) {
                ncx = ctx_finish(nq);
                if (ncx.n > 0) {   // (a tile without rules has no sub-chunk to read an index from)
                    pr0 = plan(ncx, 0, 0);
                    pr1 = plan(ncx, cnt_of(pr0), last_of(pr0));
                    pr2 = plan(ncx, pr1.q0 + cnt_of(pr1), last_of(pr1));
                    pi0 = load_idx(ncx, pr0); pi1 = load_idx(ncx, pr1); pi2 = load_idx(ncx, pr2);
                }
            }
        }
        asm volatile("" ::: "memory");
        pf++;
    };
    bool planned = false;   // r0 .. r2 / ic .. i2 of this item came with the prefetch
    MRound r0{}, r1{}, r2{};
    Idx ic{}, i1{}, i2{};
  for (;;) {   // one work item per iteration (PERSIST: until the XCD's tickets are used up; else once)
    tile = vid / blocks;
    {
        const int cblk = vid - tile * blocks;
        Wt = W + (size_t)cblk * K * (SPLIT ? (size_t)(CIN / 32) * WSLICE : (size_t)CIN * COUT);
        biast = bias ? bias + cblk * COUT : nullptr;
        ycol0 = cblk * COUT;
    }
    tile_init<TR, COUT, 64 * NW>(acc_tile, biast, init, (int64_t)tile * TR, n_dst, ldy, ycol0);
    __syncthreads();
    const int n = cx.n;
    pf = 0;
    if (n > 0) {
        if (!planned) {
            r0 = plan(cx, 0, 0); r1 = plan(cx, cnt_of(r0), last_of(r0)); r2 = plan(cx, r1.q0 + cnt_of(r1), last_of(r1));
            ic = load_idx(cx, r0); i1 = load_idx(cx, r1); i2 = load_idx(cx, r2);
        }
        stage(r0, 0, 0);
        if (NB == 3) stage(r0, 1, 1);
        float4 a_c[NP][1][SLJ];
#pragma unroll
        for (int j = 0; j < NP; j++)
#pragma unroll
            for (int jj = 0; jj < SLJ; jj++)
                a_c[j][0][jj] = load_a(X, wave < cnt_of(r0) ? ic.src : -1, CIN, SPLIT ? 32 * j + 8 * g + 4 * jj : 16 * (SLJ * j + jj) + 4 * g);
        int phase = 0;
        int bcur = 0;   // buffer the current phase reads (NB == 3: phase % 3)
        f32x4 acc[1][NT], pacc[NT];
        // tile rows += the partial sums `v` of a sub-chunk (d4: its packed dst rows, nv: rule slots in it)
        // (two column tiles at a time: the deferred groups are added in the middle of the next round, when the fragments, both
        // accumulator sets and the B operands are live -- 4 NT registers of tile values there meant spills from 128 planes on)
        auto rmw = [&](const f32x4 (&v)[NT], uint32_t d4, int nv) {
#ifdef SEI_ABL_NORMW   // diagnostic: the sums are not added into the LDS tile (results wrong, timing only)
            if (d4 != 0xFFFFFFFFu) {
#pragma unroll
                for (int nn = 0; nn < NT; nn++) asm volatile("" ::"v"(v[nn]));
                return;
            }
#endif
            float *prow[4];
#pragma unroll
            for (int i = 0; i < 4; i++) prow[i] = acc_tile + ((d4 >> (8 * i)) & 0xFF) * COUT + r16;
#pragma unroll
            for (int n0 = 0; n0 < NT; n0 += 2) {
                f32x4 tv[2];
#pragma unroll
                for (int nn = n0; nn < n0 + 2 && nn < NT; nn++)
#pragma unroll
                    for (int i = 0; i < 4; i++) tv[nn - n0][i] = prow[i][16 * nn];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (4 * g + i < nv) {
#pragma unroll
                        for (int nn = n0; nn < n0 + 2 && nn < NT; nn++) prow[i][16 * nn] = tv[nn - n0][i] + v[nn][i];
                    }
            }
        };
        int pend_g = 0;        // group (>= 1) of the sub-chunk whose sums wait in pacc; 0: none
        uint32_t pend_d4 = 0;
        int pend_nv = 0;
        MST(5);
        while (r0.q0 < n) {
            const int cnt0 = cnt_of(r0);
            const bool valid = wave < cnt0;                    // wave-uniform
            int gi = 0;                                        // the group (offset of the round) this wave's sub-chunk belongs to
#pragma unroll
            for (int j = 0; j + 1 < MO; j++) gi += wave >= (int)((r0.ends >> (8 * j)) & 0xFF) ? 1 : 0;
            const int rowc = valid ? ic.src : -1, rown = (wave < cnt_of(r1)) ? i1.src : -1;
            MRound r3 = r2;
            Idx i3 = i2;
            if constexpr (NP == 2) {
                // [r5] TWO LONG PHASES per round (64 input planes: 2 x 32).  Ablation builds of the four-phase form at level 1
                // (profiles/r05_mround_ablation_L1_C64.log) showed its phases -- 16 MFMAs per wave, ~2000 cycles of matrix work per
                // SIMD -- to be as long as a weight slice takes to land: with the MFMAs removed the kernel keeps 63 % of its time,
                // MFMA time and the rest ADD.  Here a phase holds 32 MFMAs per wave and every request has one such phase to land:
                // the barrier's wait is a plain vmcnt(0) on loads issued ~4000 cycles earlier.
                //   P0: [slices 32..63 of this round][fragments 32..63 of this round's sub-chunk][index words of round r + 3]
                //   P1: [slices 0..31 of the next round][fragments 0..31 of the next round's sub-chunk]
                // a_c[0] / a_c[1] are each refilled in the phase after their use (one register set).
                static_for<2>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    MST(4);
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::: "memory");
                    MST(0);
                    if constexpr (PERSIST) { if (s == 0 && pf < 4) prefetch_step(); }
                    // everything requested so far has landed: tell hipcc, so that it plants no wait of its own behind the
                    // requests of THIS phase
#pragma unroll
                    for (int jj = 0; jj < SLJ; jj++)
                        asm volatile("" : "+v"(a_c[s][0][jj].x), "+v"(a_c[s][0][jj].y), "+v"(a_c[s][0][jj].z), "+v"(a_c[s][0][jj].w));
                    asm volatile("" : "+v"(i3.src), "+v"(i3.d4));
                    if (s == 0) stage(r0, 1, 1); else stage(r1, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < SLJ; jj++)
                        a_c[1 - s][0][jj] = load_a(X, s == 0 ? rowc : rown, CIN, SPLIT ? 32 * (1 - s) + 8 * g + 4 * jj : 16 * (SLJ * (1 - s) + jj) + 4 * g);
                    asm volatile("" ::: "memory");
                    if constexpr (SIDE) {
                        // [r5] EVERY group of the previous round is added behind a barrier of this one -- group q in phase q, from
                        // `pacc` -- and that write-back, the planning of round r + 3 and its index loads are issued INSIDE the
                        // MFMA block (compute_phase: side work), not in front of it: +2.9 % at level 1
                        // (profiles/r05_side_work_ab.log; the same in the short phases of the NP >= 3 schedule: -2 ... -7 %)
                        if (s == 0) {
                            if (pend_g != 0) {
#pragma unroll
                                for (int nn = 0; nn < NT; nn++) pacc[nn] = acc[0][nn];
                            }
#pragma unroll
                            for (int nn = 0; nn < NT; nn++) acc[0][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        auto side = [&](int st, int tok) {
                            if (s == 0 && st == 0) {
                                r3 = plan(cx, r2.q0 + cnt_of(r2) + tok, last_of(r2));
                                i3 = load_idx(cx, r3);
                                asm volatile("" ::: "memory");
                            }
                            if (st == 1 && pend_g + tok == s + 1) {
                                rmw(pacc, pend_d4, pend_nv);
                                pend_g = 0;
                            }
                        };
                        if (valid) compute_phase<NT, SLJ, 1, 1, COUT>(wbuf + (size_t)(s * MO + gi) * WSLICE, a_c[s], acc, g, r16, side);
                        else { side(0, 0); side(1, 0); }
                    } else {
                    if (s == 0) {
                        r3 = plan(cx, r2.q0 + cnt_of(r2), last_of(r2));
                        i3 = load_idx(cx, r3);
                        asm volatile("" ::: "memory");
                        MST(1);
                        if (pend_g == 1) {   // group 1 of the previous round, behind this barrier (MO == 2)
                            rmw(MO == 2 ? acc[0] : pacc, pend_d4, pend_nv);
                            pend_g = 0;
                        }
#pragma unroll
                        for (int nn = 0; nn < NT; nn++) acc[0][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
                        MST(2);
                    } else {
                        MST(1);
                    }
                    if (valid)
                    {
                        const float *wsl = wbuf + (size_t)(s * MO + gi) * WSLICE;
                        if constexpr (SPLIT) compute_phase_split<NT, (NJ >= 12 ? 1 : 2)>(wsl, a_c[s], acc, lane);
                        else compute_phase<NT, SLJ, 1, 1, COUT>(wsl, a_c[s], acc, g, r16);
                    }
                    }
                    MST(3);
                    MST_ADD(8, valid ? 1 : 0);
                    MST_ADD(9, valid ? 0 : 1);
                });
            } else {
                constexpr int IDXP = NP >= 4 ? 2 : 1;   // the phase that requests the index words of round r + 3
                // Queue of one phase: [slices of the next phase][SLJ gathers]([index loads x 2] in phase IDXP): see the kernel above
                static_for<NP>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    MST(4);
                    // NB == 3: the previous phase's pieces (for the NEXT phase) stay in flight as well
                    constexpr int KEEP = SLJ + (NB == 3 ? STAGE_PER : 0);
                    if (phase == 0)
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::: "memory");
                    else if (s == (IDXP + 1) % NP)
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::"n"(KEEP + 2) : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::"n"(KEEP) : "memory");
                    MST(0);
                    if constexpr (PERSIST) { if (s == 0 && pf < 4) prefetch_step(); }
                    {
                        constexpr int jg = (s + 3 * NP - 3) % NP;   // the gathers of two phases ago are in (no wait of hipcc's own)
    #pragma unroll
                        for (int jj = 0; jj < SLJ; jj++)
                            asm volatile("" : "+v"(a_c[jg][0][jj].x), "+v"(a_c[jg][0][jj].y), "+v"(a_c[jg][0][jj].z), "+v"(a_c[jg][0][jj].w));
                    }
                    auto requests = [&]() {
                        if (NB == 3) {
                            const bool same2 = s + 2 < NP;
                            stage(same2 ? r0 : r1, same2 ? s + 2 : s + 2 - NP, bcur >= 1 ? bcur - 1 : 2);   // (bcur + 2) % 3
                        } else {
                            const bool same = s + 1 < NP;
                            stage(same ? r0 : r1, same ? s + 1 : 0, (phase + 1) & 1);
                        }
                        {
                            constexpr int jr = (s + NP - 1) % NP;   // the fragments the previous phase used
    #pragma unroll
                            for (int jj = 0; jj < SLJ; jj++)
                                a_c[jr][0][jj] = load_a(X, s == 0 ? rowc : rown, CIN, SPLIT ? 32 * jr + 8 * g + 4 * jj : 16 * (SLJ * jr + jj) + 4 * g);
                            asm volatile("" ::: "memory");
                        }
                        if (s == IDXP) {
                            r3 = plan(cx, r2.q0 + cnt_of(r2), last_of(r2));
                            i3 = load_idx(cx, r3);
                            asm volatile("" ::: "memory");
                        }
                    };
                    const bool late = LATE && wave >= NW / 2;   // wave-uniform
                    if (!late) requests();
                    MST(1);
                    if constexpr (s + 1 < MO) {   // group s + 1 of the previous round: every earlier group is behind a barrier
                        if (pend_g == s + 1) {
                            rmw(MO == 2 ? acc[0] : pacc, pend_d4, pend_nv);
                            pend_g = 0;
                        }
                    }
                    if (s == 0) {
    #pragma unroll
                        for (int nn = 0; nn < NT; nn++) acc[0][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    MST(2);
                    if (valid)
                    {
                        const float *wsl = wbuf + (size_t)((NB == 3 ? bcur : (phase & 1)) * MO + gi) * WSLICE;
                        if constexpr (SPLIT) compute_phase_split<NT, (NJ >= 12 ? 1 : 2)>(wsl, a_c[s], acc, lane);
                        else compute_phase<NT, SLJ, 1, 1, COUT>(wsl, a_c[s], acc, g, r16);
                    }
                    if (late) requests();
                    MST(3);
                    MST_ADD(8, valid ? 1 : 0);
                    MST_ADD(9, valid ? 0 : 1);
                    phase++;
                    bcur = bcur == 2 ? 0 : bcur + 1;
                });
            }
            const int nv = __popcll(__ballot(rowc >= 0) & 0xFFFFull);
            if (SIDE && valid) {   // group gi waits for phase gi of the next round
                pend_g = gi + 1;
                pend_d4 = ic.d4;
                pend_nv = nv;
            }
            if (!SIDE && valid && gi == 0) rmw(acc[0], ic.d4, nv);
            if (!SIDE && valid && gi > 0) {
                if (MO > 2) {   // MO == 2: the sums stay in acc, which is cleared only after they have been added (phase 0)
#pragma unroll
                    for (int nn = 0; nn < NT; nn++) pacc[nn] = acc[0][nn];
                }
                pend_g = gi;
                pend_d4 = ic.d4;
                pend_nv = nv;
            }
            r0 = r1; r1 = r2; r2 = r3;
            ic = i1; i1 = i2; i2 = i3;
            MST_ADD(10, 1);
        }
        MST(4);
        // the last round's groups 1 .. MO-1, in order (SIDE: groups 0 .. MO-1, pend_g = group + 1; the sums are still in acc)
#pragma unroll
        for (int gq = 1; gq < MO + (SIDE ? 1 : 0); gq++) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" SEI_BARRIER_INSN ::: "memory");
            if (pend_g == gq) rmw(MO == 2 || SIDE ? acc[0] : pacc, pend_d4, pend_nv);
        }
    }
    __syncthreads();
    MST(6);
    const int64_t row0 = (int64_t)tile * TR;
    tile_store<TR, COUT, 64 * NW>(acc_tile, Y, row0, n_dst, ldy, ycol0, ep);
    if (stats) tile_stats<TR, COUT, 64 * NW>(acc_tile, stats, tile, row0, n_dst, ldy, ycol0);
    MST(7);
    MST_ADD(11, 1);
    if constexpr (!PERSIST) break;
    // a tile of fewer than four rounds has not finished the requests for the next item: the rest runs here, exposed
    while (pf < 4) {
        prefetch_step();
        __syncthreads();
    }
    if (nvid >= n_items) break;
    vid = nvid;
    planned = nvid < n_items && ncx.n > 0;
    cx = ncx;
    r0 = pr0; r1 = pr1; r2 = pr2;
    ic = pi0; i1 = pi1; i2 = pi2;
    __syncthreads();   // the store has read the tile (and every wave the ticket) before the next item initialises / overwrites them
  }
    MST_FLUSH;
    CLK_END(1);
}

// Stem-type convolutions (tiny nIn, e.g. 1 -> 32 with 125 offsets): all of W sits in LDS, lane = output plane so that
// every LDS accumulator access is a contiguous row segment.  HBM / LDS bound, no MFMA (64 flop per rule).
// [r3] no barrier per offset any more: wave w OWNS rows [w TR/4, (w+1) TR/4) of the tile.  A 256-row tile holds ~19 rules
// per offset (125 offsets, 9.5 rules per row): one or two sub-chunks, which used to occupy one or two of the four waves
// while the others waited at the offset's barrier, 125 times per tile.  Now every wave reads the offset's slots (64 per
// pass, one per lane, the gathered x values with them), keeps the ones whose dst row is its own and adds them to its rows
// in slot order; rows of one offset are distinct and a wave's LDS operations execute in order, so the per-row summation
// order (offsets ascending, tile + x*w as before) and the results are unchanged, bit for bit.  Measured: 2.41 -> 2.28 ms
// for the DUNE stem (3.2 M sites, 30 M rules) -- the barrier was not what holds this kernel; it stays ~15x off its HBM
// bound (0.15 ms) on per-rule LDS read-modify-write chains, 0.65 % of the step.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_tile_smallcin_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int TR,
    ConvEpilogue ep) {
    extern __shared__ float lds[];
    float *acc = lds;                       // [TR][COUT]
    float *wl = lds + (size_t)TR * COUT;    // [K][CIN][COUT]
    const int tile = blockIdx.x;
    for (int idx = threadIdx.x; idx < TR * COUT; idx += 256) acc[idx] = bias ? bias[idx % COUT] : 0.f;
    for (int idx = threadIdx.x; idx < K * CIN * COUT; idx += 256) wl[idx] = W[idx];
    __syncthreads();
    const int64_t sub0 = tile_sub_start[tile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int groups = 64 / COUT;       // rules handled per step by one wave (COUT in {16, 32, 64})
    const int co = lane % COUT, grp = lane / COUT;
    const int rpw = TR >> 2;                // rows owned by a wave
    // segment table of the tile in two registers (K + 1 <= 128): lane l holds seg[l] and seg[64 + l]
    const int32_t *segp = tile_seg + (size_t)tile * (K + 1);
    const int sg0 = lane <= K ? segp[lane] : 0, sg1 = lane + 64 <= K ? segp[lane + 64] : 0;
    auto seg_at = [&](int i) { return i < 64 ? __builtin_amdgcn_readlane(sg0, i) : __builtin_amdgcn_readlane(sg1, i - 64); };
    // A PASS = up to 64 slots (4 sub-chunks) of one offset, one slot per lane.  The passes of the tile are walked with
    // their slot indices loaded two passes ahead and the x values one pass ahead: per pass two dependent global loads
    // (index -> x) used to sit in front of a handful of FMAs, 125+ times per tile.
    struct Pass { int k, sb, s1; };
    auto advance = [&](Pass p) {
        p.sb += 4;
        while (p.k < K && p.sb >= p.s1) {
            p.k++;
            if (p.k < K) { p.sb = seg_at(p.k); p.s1 = seg_at(p.k + 1); }
        }
        return p;
    };
    struct Idx { int src, dst; };
    auto load_idx = [&](const Pass &p) {
        Idx r;
        const int sc = p.sb + (lane >> 4);
        const bool in = p.k < K && sc < p.s1;
        const int64_t slot = (sub0 + (in ? sc : 0)) * SUB_SLOTS + (lane & 15);  // clamped: never branch around a load
        const int sv = sub_src[slot];
        r.dst = sub_dst[slot];
        r.src = in ? sv : -1;
        return r;
    };
    Pass p0{-1, 0, 0};
    p0 = advance(p0);
    Pass p1 = advance(p0), p2 = advance(p1);
    Idx i0 = load_idx(p0), i1 = load_idx(p1), i2 = load_idx(p2);
    float x0[CIN], x1[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ci++) x0[ci] = X[(size_t)(i0.src >= 0 ? i0.src : 0) * CIN + ci];
    while (p0.k < K) {
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) x1[ci] = X[(size_t)(i1.src >= 0 ? i1.src : 0) * CIN + ci];
        float wk[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) wk[ci] = wl[(p0.k * CIN + ci) * COUT + co];
        unsigned long long mine = __ballot(i0.src >= 0 && i0.dst / rpw == wave);
        while (mine) {  // `groups` of this wave's rules per step, in slot order
            // (batching four steps' shuffles / LDS reads / LDS writes measured slower: 2.44 vs 2.28 ms)
            int sel = -1;
#pragma unroll
            for (int q = 0; q < groups; q++) {
                const int b = mine ? __ffsll((long long)mine) - 1 : -1;
                if (b >= 0) mine &= mine - 1;
                if (q == grp) sel = b;
            }
            const int bl = sel >= 0 ? sel : 0;
            const int dl = __shfl(i0.dst, bl);
            float v = 0.f;
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) v = fmaf(__shfl(x0[ci], bl), wk[ci], v);
            if (sel >= 0) acc[dl * COUT + co] += v;
        }
        p0 = p1; p1 = p2; p2 = advance(p2);
        i0 = i1; i1 = i2; i2 = load_idx(p2);
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) x0[ci] = x1[ci];
    }
    __syncthreads();
    const int64_t row0 = (int64_t)tile * TR;
    if (ep.mean) {  // inference: eval-mode BatchNorm (+ residual) + LeakyReLU on the way out (common.h)
        constexpr int V = COUT >> 2;
        for (int idx = threadIdx.x; idx < TR * V; idx += 256) {
            const int r = idx / V, v = idx - r * V;
            if (row0 + r < n_dst) {
                const float4 a = *reinterpret_cast<const float4 *>(acc + (size_t)r * COUT + 4 * v);
                *reinterpret_cast<float4 *>(Y + (size_t)(row0 + r) * COUT + 4 * v) =
                    conv_epilogue_apply(ep, a, 4 * v, ep.res ? ep.res + (size_t)(row0 + r) * COUT : nullptr);
            }
        }
        return;
    }
    for (int idx = threadIdx.x; idx < TR * COUT; idx += 256) {
        int r = idx / COUT;
        if (row0 + r < n_dst) Y[(size_t)row0 * COUT + idx] = acc[idx];
    }
}

// Any channel counts (odd test sizes): same tile / sub-chunk walk, scalar FMA.
__global__ __launch_bounds__(256) void conv_tile_generic_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ Y,
    const int32_t *__restrict__ tile_sub_start, const int32_t *__restrict__ tile_seg,
    const int32_t *__restrict__ sub_src, const uint8_t *__restrict__ sub_dst, int64_t n_dst, int K, int CIN, int COUT,
    int TR) {
    extern __shared__ float lds[];
    const int tile = blockIdx.x;
    float *acc = lds;
    for (int idx = threadIdx.x; idx < TR * COUT; idx += 256) acc[idx] = bias ? bias[idx % COUT] : 0.f;
    __syncthreads();
    const int64_t sub0 = tile_sub_start[tile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    for (int k = 0; k < K; k++) {
        const int s0 = tile_seg[(size_t)tile * (K + 1) + k], s1 = tile_seg[(size_t)tile * (K + 1) + k + 1];
        const float *Wk = W + (size_t)k * CIN * COUT;
        for (int s = s0 + wave; s < s1; s += 4) {
            int64_t slot = (sub0 + s) * SUB_SLOTS + r16;
            int src = sub_src[slot], d = sub_dst[slot];
            if (src < 0) continue;
            const float *xr = X + (size_t)src * CIN;
            for (int co = g; co < COUT; co += 4) {
                float sum = 0.f;
                for (int ci = 0; ci < CIN; ci++) sum = fmaf(xr[ci], Wk[(size_t)ci * COUT + co], sum);
                acc[d * COUT + co] += sum;  // rows of one offset are distinct: no atomics needed
            }
        }
        __syncthreads();  // keeps the per-row sum order fixed (k-phases), as in the MFMA kernel
    }
    const int64_t row0 = (int64_t)tile * TR;
    for (int idx = threadIdx.x; idx < TR * COUT; idx += 256) {
        int r = idx / COUT;
        if (row0 + r < n_dst) Y[(size_t)row0 * COUT + idx] = acc[idx];
    }
}

__global__ void fill_bias_kernel(float *Y, int64_t n, int C, const float *bias) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n * C) Y[t] = bias ? bias[t % C] : 0.f;
}

static size_t fwd_lds(int tile_rows, int nOut, int slj, int K) {
    size_t b = ((size_t)tile_rows * (nOut + APAD) + 2 * 16 * slj * nOut) * sizeof(float) + (K + 1) * sizeof(int);
#ifdef SEI_TRACE
    b += 16;   // the trace slot of the workgroup
#endif
    return b;
}

#ifdef SEI_EXPERIMENTS
// Kernel-experiment builds only (make EXTRA=-DSEI_EXPERIMENTS; tools/sweep_tile.py): SEI_FWD_PLAN=
// "maxs,cs,slj,nOut,blocks[,waves][;...]" overrides the default tiling of the layers with that nOut.  The shipped library
// holds the default plans only and reads no environment variable.
static bool env_plan(int nOut, int *maxs, int *cs, int *slj, int *blocks, int *nw) {
    const char *e = getenv("SEI_FWD_PLAN");
    while (e && *e) {
        int no = 0;
        *blocks = 1;
        *nw = 4;
        const int got = sscanf(e, "%d,%d,%d,%d,%d,%d", maxs, cs, slj, &no, blocks, nw);
        if (got >= 4 && no == nOut) {
            if (got == 4) *blocks = 1;
            if (got < 6) *nw = 4;
            return true;
        }
        e = strchr(e, ';');
        if (e) e++;
    }
    return false;
}
#endif

// sixteen-wave instantiations of the round kernels for NT column tiles and nIn input planes: slj = slice width / 16 (0: none),
// mo = offsets a round may span (2: conv_round_mfma_kernel; 3 / 4: conv_mround_mfma_kernel, whose LDS holds mo slices per buffer)
struct Round16 { int slj, mo, fam, nb; };   // fam 2: conv_round_mfma_kernel, 3: conv_mround_mfma_kernel; nb: weight buffers
static Round16 round16_plan(int nt, int nIn) {
    int nb = 2;
    bool long2 = true;
#ifdef SEI_EXPERIMENTS
    int mo_cap = 2;
    bool fam3 = false;   // "SEI_ROUND_FAM=3": conv_mround_mfma_kernel (two offsets, SEI_ROUND_NB buffers) wherever the round kernels run
    if (const char *e = getenv("SEI_ROUND_FAM")) fam3 = atoi(e) == 3;
    if (const char *e = getenv("SEI_ROUND_MO")) mo_cap = atoi(e);
    if (const char *e = getenv("SEI_ROUND_LONG")) long2 = atoi(e) != 0;   // "0": the four-phase forms at 64 input planes
    if (const char *e = getenv("SEI_ROUND_NB")) nb = atoi(e);
#endif
    // [r5] 64 input planes: two long phases of 32 planes per round, two offsets (conv_mround_mfma_kernel<NT, 16, 4, 2, 2>)
    if (long2 && nIn == 64 && (nt == 4 || nt == 6)) return {2, 2, 3, 2};
#ifdef SEI_EXPERIMENTS
    if (mo_cap >= 3 && nIn % 16 == 0) {
        // [r5] multi-offset rounds (measured, not a default: profiles/r05_mround_ab.log): the widest slice and the most offsets
        // whose buffers fit beside the 256-row tile
        Round16 r{0, 0, 3, 2};
        if (nt == 4 && nIn == 64) r = {1, 4, 3, 2};
        if (nt == 4 && nIn == 96) r = {2, 3, 3, 2};
        if (nt == 4 && (nIn == 128 || nIn == 160 || nIn == 192)) r = {2, 4, 3, 2};
        if (nt == 5 && (nIn == 128 || nIn == 160 || nIn == 192)) r = {2, 3, 3, 2};
        if (nt == 6 && (nIn == 64 || nIn == 96 || nIn == 128)) r = {1, 4, 3, 2};
        if (r.mo > mo_cap) r.mo = mo_cap;
        if (r.slj) return r;
    }
#endif
    if (nIn % 32 == 0 && nIn >= 96 && nIn <= 192) {
#ifdef SEI_EXPERIMENTS
        // "SEI_ROUND_L2_SLJ1=1": level 2 with 16-plane slices (120 KB of LDS instead of 144: room for a co-resident dW workgroup)
        if (nt == 6 && nIn == 96 && getenv("SEI_ROUND_L2_SLJ1") && atoi(getenv("SEI_ROUND_L2_SLJ1"))) return {1, 2, 3, 2};
        if (fam3 && ((nt == 6 && nIn <= 128) || nt == 4 || (nt == 5 && nIn >= 128)) && (nb == 2 || nt != 6)) return {2, 2, 3, nb == 3 ? 3 : 2};
#endif
        // [r5] level 5 (192 -> 192 planes in 3 x 64-column blocks, 501 workgroups on 256 CUs): the multi-offset kernel in its
        // two-offset form -- segment table in a lane register (no LDS round trips while a round is planned), staging pieces dealt
        // over all sixteen waves -- runs 90.2 against 83.4 TFLOP/s; at 96 / 128 / 160 planes the two kernels tie within 1 %
        // (88.1 / 92.4 / 89.3 against 86.6 / 92.0 / 88.5) and the round-4 kernel stays.  A THIRD weight buffer (slices staged two
        // phases ahead, SEI_ROUND_NB=3 in experiment builds) changes nothing anywhere: 88.1 / 91.5 / 87.9 / 90.4
        // (profiles/r05_round_third_buffer_ab.log)
        if (nt == 4 && nIn == 192) return {2, 2, 3, nb == 3 ? 3 : 2};
        if (nt == 6 && (nIn == 96 || nIn == 128)) return {2, 2, 2, 2};
        if (nt == 4 && (nIn == 96 || nIn == 128 || nIn == 160 || nIn == 192)) return {2, 2, 2, 2};
        if (nt == 5 && (nIn == 128 || nIn == 160 || nIn == 192)) return {2, 2, 2, 2};
    }
    if (nt == 6 && nIn == 64) return {1, 2, 2, 2};
    return {0, 0, 0, 2};
}

// PERSIST form of conv_mround_mfma_kernel (two offsets, two buffers)
// (experiment builds, SEI_ROUND_PERSIST=1; measured and not adopted: +1.1 % at level 1, -2 ... -6 % at levels 2-4 and -23 % at level 5,
// where the prefetched state spills 43 registers: profiles/r05_round_persistent_ab.log)
static bool round_persist(int fam, int mo, int nb, int nIn) {
    bool on = false;
#ifdef SEI_EXPERIMENTS
    if (const char *e = getenv("SEI_ROUND_PERSIST")) on = atoi(e) != 0;
#endif
    (void)nIn;
    return on && fam == 3 && mo == 2 && nb == 2;
}

static int device_cu_count() {   // compute units of the current device (cached per device)
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    n = (n / 8) * 8 > 0 ? (n / 8) * 8 : 8;
    cus[dev] = n;
    return n;
}

// [r5] the 64 -> 64 plane layers of large levels (level 1 of the DUNE network) leave the wave-private kernel for the
// sixteen-wave multi-offset rounds (conv_mround_mfma_kernel<4, 16, 4, 1, 4>)
bool conv_rounds_64(int nIn, int nOut, int K, int64_t n_dst) {
#ifdef SEI_EXPERIMENTS
    if (const char *e = getenv("SEI_L1_ROUNDS")) if (atoi(e) == 0) return false;   // A/B: the wave kernel keeps level 1
#endif
    return nIn == 64 && nOut == 64 && K + 1 <= 64 && n_dst >= 100000;
}

static ConvPlan plan_tile(int nIn, int nOut, int K, int64_t n_dst) {
    ConvPlan p{};
    bool mfma = (nIn % 16 == 0) && (nOut % 16 == 0) && nIn >= 16 && nOut <= 256;
    if (mfma) {
        // The tiling depends on nOut (and, for the tile height of wide layers, on the row count).
        // nOut <= 96: 128-row tiles, one workgroup owns all columns.  Wider: one workgroup per (tile, column
        // block) of <= 4 column tiles -- the LDS tile and the weight slices shrink with the block, so 128-row
        // tiles fit again and 3-6 workgroups share a CU; the src rows are gathered once per block, and the
        // blocks of a tile run back to back on one XCD and find them in its L2.  Measured at 64 events
        // (profiles/r01_conv_fwd_column_blocks.log): C=128 67 -> 80, C=160 53 -> 72, C=192 51 -> 64 TFLOP/s.
        const int tiles = nOut / 16;
        p.cs = 1; p.maxs = 2; p.blocks = 1;
        if (nOut > 96) {
            int nt = 1;
            for (int c = 2; c <= 4; c++)
                if (tiles % c == 0) nt = c;
            if (nt > 1) {
                p.blocks = tiles / nt;
                if (n_dst < 100000) p.maxs = 1;   // small levels: 64-row tiles fill the chip better
            } else {
                p.maxs = 1;                        // 7, 11, 13 column tiles: undivided, 64-row tiles
            }
        }
        p.slj = (nIn % 32 == 0) ? 2 : 1;
        const int kind_forced = forced_conv_kernel();
        // [r4, late] 192 columns (3 x 64) join from 18 000 rows on: sixteen-wave rounds 83.1 against 69.6 TFLOP/s at level 5 of the
        // 64-event batch (42 666 rows: 501 workgroups), 317.9 -> 316.9 ms per step; 72.1 vs 61.5 at 20 627 rows, 36.4 vs 50.6 at
        // 10 358 (160 columns: 72.9 vs 73.3 at 71 639 rows, 53.8 vs 64.7 at 35 830 -- their threshold stays at 100 000).  64 -> 64 columns do NOT: eight-wave rounds run
        // 64.0 against the wave kernel's 62.2 TFLOP/s at level 1 alone, but the whole step loses 1.6 ms with them (318.5 vs 316.9 ms,
        // builds interleaved twice on one box) -- they stay behind sei_debug_force_conv_kernel(5)
        if (nOut == 96 || ((nOut == 128 || nOut == 160) && (n_dst >= 100000 || kind_forced == CONV_KERNEL_ROUNDS)) ||
            (nOut == 192 && (n_dst >= 18000 || kind_forced == CONV_KERNEL_ROUNDS)) ||
            (nOut == 64 && (kind_forced == CONV_KERNEL_ROUNDS || (kind_forced == CONV_KERNEL_AUTO && conv_rounds_64(nIn, nOut, K, n_dst))))) {
            // 96 / 128 / 160 columns: eight-wave workgroups -- one sub-chunk per wave and offset on 128-row tiles, two
            // column blocks, 16-plane weight slices: 30-50 KB of LDS and 60-90 VGPRs (no spills), three or four
            // workgroups (24-32 waves) per CU.  A phase is one latency chain (barrier -> B from LDS -> MFMA -> LDS
            // add) with little MFMA content; more independent chains per CU is what raises the MFMA rate: 74.4 vs
            // 69.1 TFLOP/s at 96 columns, 82.7 vs 82.1 at 128, 78.0 vs 76.7 at 160 (64-event batch, one device each;
            // profiles/r02_conv_units_experiment.log); 192 columns (level 5, few tiles): slower, not used.
            p.nw = 8; p.maxs = 1; p.slj = 1; p.blocks = nOut == 64 ? 1 : (nOut == 192 ? 3 : 2);
            if (nOut == 96 && nIn % 32 == 0) {
                // [r3] 96 columns in ONE block with 32-plane weight slices: the same speed within noise (75.0 vs 75.3 TFLOP/s
                // at level 2, 359.0 vs 359.3 ms per step on one box) but every src row is gathered once instead of once per
                // column block: raw FETCH_SIZE of the kernel 939 -> 639 MB per launch (profiles/r03_conv_tile_oneblock96.txt)
                p.slj = 2; p.blocks = 1;
            }
        }
        // Two-offset rounds (conv_round_mfma_kernel) for the eight-wave plans: 84.7 / 87.1 / 86.2 against 75.3 / 81.2 / 80.7
        // TFLOP/s at 96 / 128 / 160 planes of the 64-event batch (profiles/r04_conv_round_ab.log), bit-identical.  The A
        // fragments live in one refilled register set where two workgroups per CU are the plan anyway (96 columns in one
        // block: 110 registers) and at 160 planes (2 x 80 columns: 122 registers, two workgroups instead of three, +7 %);
        // the 2 x 64-column blocks of 128 planes keep per-phase gathers and three workgroups (the two forms measured equal).
        if (p.nw == 8 && p.maxs == 1 && nIn >= 64 && kind_forced != CONV_KERNEL_TILE) {   // the schedule uses phases 0..3
            bool on = true;
#ifdef SEI_EXPERIMENTS
            if (const char *e = getenv("SEI_TILE_ROUNDS")) on = atoi(e) != 0;
#endif
            if (on) {
                p.g = 2; p.slj = 2;
                // [r4, late] where instantiated: ONE sixteen-wave workgroup per CU on 256-row tiles (the variant dW walks anyway)
                // with 32-plane slices -- half the phases, half the weight staging per sub-chunk, padding 0.95 instead of 0.90:
                // 88.1 / 92.9 / 89.0 against 84.7 / 86.4 / 86.6 TFLOP/s at 96 / 128 / 160 planes (profiles/r04_conv_round_ab.log)
                // [r5] opt-in (sei_set_product_mode(1)): the products as three-way bf16 splits where the sixteen-wave rounds run;
                // 96 columns then take two blocks of 48 (the split slices of two offsets, 36 KB per buffer at 96 columns, do not
                // fit beside a 96 KB tile)
                const bool split = product_mode() == 1 && K + 1 <= 64 && nIn % 32 == 0 && nIn >= 64 && nIn <= 192 &&
                                   (nOut == 64 || nOut == 96 || nOut == 128 || nOut == 160 || nOut == 192);
                if (split) {
                    p.blocks = nOut == 64 || nOut == 96 ? 1 : (nOut == 192 ? 3 : 2);
                    p.nw = 16; p.mo = 2; p.g = 3; p.nb = 2; p.split = true; p.persist = false;
                    p.slj = 6;   // sizes the LDS: 2 buffers x 2 offsets x 3 nt KiB of split slices
                    // 96 columns in ONE block: 72 KB of split slices leave 72 KB for the tile -- 192 rows, twelve waves (two blocks of 48
                    // columns on 256-row tiles: 116 against NEW fp32-equivalent TFLOP/s at level 2; SEI_SPLIT_L2_BLOCKS=2 in experiment builds)
                    if (nOut == 96) p.nw = 12;
#ifdef SEI_EXPERIMENTS
                    if (nOut == 96 && getenv("SEI_SPLIT_L2_BLOCKS") && atoi(getenv("SEI_SPLIT_L2_BLOCKS")) == 2) { p.blocks = 2; p.nw = 16; }
#endif
                }
                const int cols = nOut / p.blocks;
                const Round16 r16p = (!split && K + 1 <= 64) ? round16_plan(cols / 16, nIn) : Round16{0, 0, 0, 2};
                bool wide = r16p.slj > 0;
#ifdef SEI_EXPERIMENTS
                if (const char *e = getenv("SEI_ROUND_NW")) wide = wide && atoi(e) == 16;
#endif
                if (wide) { p.nw = 16; p.mo = r16p.mo; p.g = r16p.fam; p.nb = r16p.nb; p.slj = (r16p.nb * r16p.mo * r16p.slj + 1) / 2; p.persist = round_persist(r16p.fam, r16p.mo, r16p.nb, nIn); }   // slj sizes the LDS: nb buffers of mo slices
            }
        }
#ifdef SEI_EXPERIMENTS
        int em, ec, es, eb, ew;
        if (env_plan(nOut, &em, &ec, &es, &eb, &ew)) {
            p.g = 0;
            if (tiles % ec == 0 && em >= 1 && em <= 4 && (ec == 1 || ec == 2)) {
                p.maxs = em; p.cs = ec; p.blocks = 1; p.nw = 4;
                p.slj = (nIn % 32 == 0) ? 2 : 1;
                if (ew == 8 && ((ec == 1 && (em == 1 || em == 2)) || (ec == 2 && em == 2))) p.nw = 8;
                if (es == 1 || (es == 2 && nIn % 32 == 0)) p.slj = es;
                if (eb >= 1 && ec == 1 && tiles % eb == 0) p.blocks = eb;
            }
        }
#endif
        p.nt = tiles / p.cs / p.blocks;
        p.tile_rows = 16 * p.maxs * (p.nw / p.cs);
        p.lds_bytes = fwd_lds(p.tile_rows, nOut / p.blocks, p.slj, K);
        if (p.lds_bytes > 160 * 1024) mfma = false;
    }
    p.kind = mfma ? 1 : 0;
    p.col_group = mfma ? p.nt * 16 : 0;
    if (!mfma) {
        p.blocks = 1;
        p.tile_rows = (size_t)256 * nOut * sizeof(float) <= 96 * 1024 ? 256 : 128;
        p.lds_bytes = (size_t)p.tile_rows * nOut * sizeof(float);
    }
    return p;
}

#ifndef SEI_EXPERIMENTS
// The row-band / ring-tile (conv_band.hip) and column-owner (conv_cols.hip, 4.3f) families were measured and are
// never a plan's choice: the shipped library does not compile those files (Makefile: SRCS) and holds these stubs instead.
bool plan_band(int, int, int, int64_t, ConvPlan *) { return false; }
bool plan_cols(int, int, int, int64_t, ConvPlan *) { return false; }
int conv_forward_band(const TiledRules &, const ConvPlan &, const float *, const float *, const float *, float *, int, int,
                      const float *, double *, hipStream_t) {
    SEI_FAIL("conv_forward_band: the row-band kernels exist only in experiment builds (make EXTRA=-DSEI_EXPERIMENTS)");
}
int conv_forward_cols(const TiledRules &, const ConvPlan &, const float *, const float *, const float *, float *, int, int,
                      const float *, double *, hipStream_t, const ConvEpilogue &) {
    SEI_FAIL("conv_forward: the column-owner kernel exists only in experiment builds (make EXTRA=-DSEI_EXPERIMENTS)");
}
#endif

// the plan of one convolution call (forward, or dX with the plane counts swapped)
ConvPlan plan_conv(int nIn, int nOut, int K, int64_t n_dst) {
    ConvPlan p{};
    if (plan_band(nIn, nOut, K, n_dst, &p)) return p;
    if (plan_cols(nIn, nOut, K, n_dst, &p)) return p;
    if (plan_wave(nIn, nOut, K, n_dst, &p)) return p;
    return plan_tile(nIn, nOut, K, n_dst);
}

template <int NT, int CS, int SLJ, int MAXS, int NW = 4>
static int launch_mfma(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias,
                       float *Y, int nIn, const float *init, double *stats, hipStream_t st, const ConvEpilogue &ep) {
    SEI_TRY(ensure_max_lds((const void *)conv_tile_mfma_kernel<NT, CS, SLJ, MAXS, NW>, 160 * 1024));
    conv_tile_mfma_kernel<NT, CS, SLJ, MAXS, NW><<<((t.n_tiles * p.blocks + 7) / 8) * 8, 64 * NW, p.lds_bytes, st>>>(
        X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init,
        NT * CS * 16 * p.blocks, p.blocks, stats, ep);
    SEI_LAUNCH_CHECK();
    return 0;
}

template <int NT>
static int launch_round(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                        int nIn, const float *init, double *stats, hipStream_t st, const ConvEpilogue &ep) {
    // one register set of A fragments refilled a phase after use (NJ = nIn / 16 > 0) where 128 registers hold it; the narrower
    // column blocks keep the per-phase gathers and three workgroups per CU
    if ((p.nw == 16 || p.nw == 12) && p.g == 3 && p.split) {   // [r5] opt-in: products as three-way bf16 splits (W = conv_split_weights' layout)
#define MROUND16S(NW_, NJ_)                                                                                                \
        if (p.nw == NW_ && nIn == 16 * NJ_) {                                                                              \
            SEI_TRY(ensure_max_lds((const void *)conv_mround_mfma_kernel<NT, NW_, NJ_, 2, 2, 2, false, false, true>, 160 * 1024)); \
            conv_mround_mfma_kernel<NT, NW_, NJ_, 2, 2, 2, false, false, true><<<((t.n_tiles * p.blocks + 7) / 8) * 8, 64 * NW_, p.lds_bytes, st>>>( \
                X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init,     \
                NT * 16 * p.blocks, p.blocks, stats, ep, nullptr);                                                         \
            SEI_LAUNCH_CHECK();                                                                                            \
            return 0;                                                                                                      \
        }
        if constexpr (NT == 6) { MROUND16S(12, 4) MROUND16S(12, 6) MROUND16S(12, 8) }
        if constexpr (NT == 4) { MROUND16S(16, 4) MROUND16S(16, 6) MROUND16S(16, 8) MROUND16S(16, 10) MROUND16S(16, 12) }
        if constexpr (NT == 5) { MROUND16S(16, 8) MROUND16S(16, 10) MROUND16S(16, 12) }
#ifdef SEI_EXPERIMENTS   // SEI_SPLIT_L2_BLOCKS=2
        if constexpr (NT == 3) { MROUND16S(16, 4) MROUND16S(16, 6) MROUND16S(16, 8) }
#endif
#undef MROUND16S
        SEI_FAIL("conv_forward: no split-product instantiation for this plan");
    }
    if (p.nw == 16 && p.g == 3) {   // [r5] sixteen-wave rounds spanning up to p.mo offsets / with two long phases (conv_mround_mfma_kernel)
#ifdef SEI_EXPERIMENTS   // SEI_ROUND_LATE=1 (with SEI_ROUND_FAM=3 SEI_ROUND_NB=3): the upper eight waves issue their requests behind the MFMAs
#define MROUND16L(NJ_)                                                                                                     \
        if (nIn == 16 * NJ_ && p.mo == 2 && p.nb == 3 && getenv("SEI_ROUND_LATE") && atoi(getenv("SEI_ROUND_LATE"))) {       \
            SEI_TRY(ensure_max_lds((const void *)conv_mround_mfma_kernel<NT, 16, NJ_, 2, 2, 3, true>, 160 * 1024));          \
            conv_mround_mfma_kernel<NT, 16, NJ_, 2, 2, 3, true><<<((t.n_tiles * p.blocks + 7) / 8) * 8, 1024, p.lds_bytes, st>>>( \
                X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init,     \
                NT * 16 * p.blocks, p.blocks, stats, ep, nullptr);                                                         \
            SEI_LAUNCH_CHECK();                                                                                            \
            return 0;                                                                                                      \
        }
        if constexpr (NT == 4) { MROUND16L(8) MROUND16L(10) MROUND16L(12) }
        if constexpr (NT == 5) { MROUND16L(10) }
#undef MROUND16L
#endif
        // PERSIST (p.persist): one workgroup per CU walks the work items of its XCD by tickets (8 counters, zeroed per launch)
#define MROUND16P(NJ_, SLJ_)                                                                                               \
        if (p.persist && nIn == 16 * NJ_ && p.mo == 2 && p.nb == 2 && p.slj == 2 * SLJ_) {                                 \
            int *work = nullptr;                                                                                           \
            SEI_TRY(pool_alloc_t(&work, 8, st));                                                                           \
            int rc_ = 0;                                                                                                   \
            if (hipMemsetAsync(work, 0, 8 * sizeof(int), st) != hipSuccess) { set_error("conv_forward: hipMemsetAsync failed"); rc_ = 1; } \
            if (!rc_) rc_ = ensure_max_lds((const void *)conv_mround_mfma_kernel<NT, 16, NJ_, SLJ_, 2, 2, false, true>, 160 * 1024); \
            if (!rc_) {                                                                                                    \
                const int items = t.n_tiles * p.blocks, cus = device_cu_count();                                           \
                const int grid = items < cus ? ((items + 7) / 8) * 8 : cus;                                                \
                conv_mround_mfma_kernel<NT, 16, NJ_, SLJ_, 2, 2, false, true><<<grid, 1024, p.lds_bytes, st>>>(            \
                    X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init, \
                    NT * 16 * p.blocks, p.blocks, stats, ep, work);                                                        \
                if (hipGetLastError() != hipSuccess) { set_error("conv_forward: launch failed (persistent rounds)"); rc_ = 1; } \
            }                                                                                                              \
            pool_free(work);                                                                                               \
            return rc_;                                                                                                    \
        }
#ifdef SEI_EXPERIMENTS
        if constexpr (NT == 4) { MROUND16P(4, 2) MROUND16P(12, 2) }
        if constexpr (NT == 6) { MROUND16P(4, 2) }
        if constexpr (NT == 4) { MROUND16P(6, 2) MROUND16P(8, 2) MROUND16P(10, 2) }
        if constexpr (NT == 5) { MROUND16P(8, 2) MROUND16P(10, 2) MROUND16P(12, 2) }
        if constexpr (NT == 6) { MROUND16P(6, 2) MROUND16P(8, 2) }
#endif
#undef MROUND16P
#define MROUND16(NJ_, SLJ_, MO_, NB_)                                                                                      \
        if (nIn == 16 * NJ_ && p.mo == MO_ && p.nb == NB_ && p.slj == (NB_ * MO_ * SLJ_ + 1) / 2) {                         \
            SEI_TRY(ensure_max_lds((const void *)conv_mround_mfma_kernel<NT, 16, NJ_, SLJ_, MO_, NB_>, 160 * 1024));        \
            conv_mround_mfma_kernel<NT, 16, NJ_, SLJ_, MO_, NB_><<<((t.n_tiles * p.blocks + 7) / 8) * 8, 1024, p.lds_bytes, st>>>( \
                X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init,     \
                NT * 16 * p.blocks, p.blocks, stats, ep, nullptr);                                                         \
            SEI_LAUNCH_CHECK();                                                                                            \
            return 0;                                                                                                      \
        }
        if constexpr (NT == 4) { MROUND16(4, 2, 2, 2) MROUND16(12, 2, 2, 2) }
        if constexpr (NT == 6) { MROUND16(4, 2, 2, 2) }
#ifdef SEI_EXPERIMENTS   // SEI_ROUND_FAM=3 with SEI_ROUND_NB=2 / 3
        if constexpr (NT == 4) { MROUND16(6, 2, 2, 2) MROUND16(8, 2, 2, 2) MROUND16(10, 2, 2, 2)
                                 MROUND16(6, 2, 2, 3) MROUND16(8, 2, 2, 3) MROUND16(10, 2, 2, 3) MROUND16(12, 2, 2, 3) }
        if constexpr (NT == 5) { MROUND16(8, 2, 2, 2) MROUND16(10, 2, 2, 2) MROUND16(12, 2, 2, 2)
                                 MROUND16(8, 2, 2, 3) MROUND16(10, 2, 2, 3) MROUND16(12, 2, 2, 3) }
        if constexpr (NT == 6) { MROUND16(6, 2, 2, 2) MROUND16(8, 2, 2, 2) MROUND16(6, 1, 2, 2) }
#endif
#ifdef SEI_EXPERIMENTS   // SEI_ROUND_MO=3 / 4, SEI_ROUND_LONG=0
        if constexpr (NT == 4) { MROUND16(4, 1, 4, 2) MROUND16(6, 2, 3, 2) MROUND16(8, 2, 4, 2) MROUND16(10, 2, 4, 2) MROUND16(12, 2, 4, 2)
                                 MROUND16(4, 1, 3, 2) MROUND16(8, 2, 3, 2) MROUND16(10, 2, 3, 2) MROUND16(12, 2, 3, 2) }
        if constexpr (NT == 5) { MROUND16(8, 2, 3, 2) MROUND16(10, 2, 3, 2) MROUND16(12, 2, 3, 2) }
        if constexpr (NT == 6) { MROUND16(4, 1, 4, 2) MROUND16(6, 1, 4, 2) MROUND16(8, 1, 4, 2) MROUND16(4, 1, 3, 2) MROUND16(6, 1, 3, 2) MROUND16(8, 1, 3, 2) }
#endif
#undef MROUND16
        SEI_FAIL("conv_forward: no multi-offset round instantiation for this plan");
    }
    if (p.nw == 16) {   // one sixteen-wave workgroup per CU, refilled fragments, 32-plane slices (16-plane behind 64 planes)
#define ROUND16(NJ_, SLJ_)                                                                                               \
        if (nIn == 16 * NJ_ && p.slj == 2 * SLJ_) {                                                                      \
            SEI_TRY(ensure_max_lds((const void *)conv_round_mfma_kernel<NT, 16, NJ_, SLJ_>, 160 * 1024));                \
            conv_round_mfma_kernel<NT, 16, NJ_, SLJ_><<<((t.n_tiles * p.blocks + 7) / 8) * 8, 1024, p.lds_bytes, st>>>(  \
                X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init,   \
                NT * 16 * p.blocks, p.blocks, stats, ep);                                                                \
            SEI_LAUNCH_CHECK();                                                                                          \
            return 0;                                                                                                    \
        }
        if constexpr (NT == 6) { ROUND16(6, 2) ROUND16(8, 2) }
        if constexpr (NT == 4) { ROUND16(6, 2) ROUND16(8, 2) ROUND16(10, 2) }
        if constexpr (NT == 5) { ROUND16(8, 2) ROUND16(10, 2) ROUND16(12, 2) }
#ifdef SEI_EXPERIMENTS   // the plans conv_mround_mfma_kernel took over in round 5 (SEI_ROUND_LONG=0, SEI_ROUND_FAM unset ...)
        if constexpr (NT == 6) { ROUND16(4, 1) }
        if constexpr (NT == 4) { ROUND16(12, 2) }
#endif
#undef ROUND16
        SEI_FAIL("conv_forward: no sixteen-wave round instantiation for this plan");
    }
    bool jit = NT == 6 || NT == 5;
#ifdef SEI_EXPERIMENTS
    if (const char *e = getenv("SEI_ROUND_JIT")) jit = atoi(e) != 0;
#endif
#define ROUND_JIT(NJ_)                                                                                              \
    if (jit && nIn == 16 * NJ_) {                                                                                   \
        SEI_TRY(ensure_max_lds((const void *)conv_round_mfma_kernel<NT, 8, NJ_>, 160 * 1024));                      \
        conv_round_mfma_kernel<NT, 8, NJ_><<<((t.n_tiles * p.blocks + 7) / 8) * 8, 512, p.lds_bytes, st>>>(         \
            X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init,  \
            NT * 16 * p.blocks, p.blocks, stats, ep);                                                               \
        SEI_LAUNCH_CHECK();                                                                                         \
        return 0;                                                                                                   \
    }
    if constexpr (NT == 6) { ROUND_JIT(4) ROUND_JIT(6) ROUND_JIT(8) }
    if constexpr (NT == 4) { ROUND_JIT(4) ROUND_JIT(6) ROUND_JIT(8) ROUND_JIT(10) }
    if constexpr (NT == 5) { ROUND_JIT(8) ROUND_JIT(10) ROUND_JIT(12) }
#undef ROUND_JIT
    SEI_TRY(ensure_max_lds((const void *)conv_round_mfma_kernel<NT, 8, 0>, 160 * 1024));
    conv_round_mfma_kernel<NT, 8, 0><<<((t.n_tiles * p.blocks + 7) / 8) * 8, 512, p.lds_bytes, st>>>(
        X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, nIn, t.n_tiles, init,
        NT * 16 * p.blocks, p.blocks, stats, ep);
    SEI_LAUNCH_CHECK();
    return 0;
}

#define FWD_CASE(NT_, CS_, SLJ_, MAXS_)                                            \
    if (p.nt == NT_ && p.cs == CS_ && p.slj == SLJ_ && p.maxs == MAXS_ && p.nw == 4) \
        return launch_mfma<NT_, CS_, SLJ_, MAXS_>(t, p, X, W, bias, Y, nIn, init, stats, st, epv);
#ifdef SEI_EXPERIMENTS
#define FWD_CASE8(NT_)                                                             \
    if (p.nt == NT_ && p.cs == 1 && p.maxs == 1 && p.nw == 8) {                    \
        if (p.slj == 2) return launch_mfma<NT_, 1, 2, 1, 8>(t, p, X, W, bias, Y, nIn, init, stats, st, epv); \
        return launch_mfma<NT_, 1, 1, 1, 8>(t, p, X, W, bias, Y, nIn, init, stats, st, epv);                 \
    }
#else
#define FWD_CASE8(NT_)                                                             \
    if (p.nt == NT_ && p.cs == 1 && p.maxs == 1 && p.nw == 8 && p.slj == 1)        \
        return launch_mfma<NT_, 1, 1, 1, 8>(t, p, X, W, bias, Y, nIn, init, stats, st, epv);
#endif
#define FWD_CASE8S2(NT_)                                                           \
    if (p.nt == NT_ && p.cs == 1 && p.maxs == 1 && p.nw == 8 && p.slj == 2)        \
        return launch_mfma<NT_, 1, 2, 1, 8>(t, p, X, W, bias, Y, nIn, init, stats, st, epv);
#define FWD_CASE8M2(NT_)                                                           \
    if (p.nt == NT_ && p.cs == 1 && p.maxs == 2 && p.nw == 8) {                    \
        if (p.slj == 2) return launch_mfma<NT_, 1, 2, 2, 8>(t, p, X, W, bias, Y, nIn, init, stats, st, epv); \
        return launch_mfma<NT_, 1, 1, 2, 8>(t, p, X, W, bias, Y, nIn, init, stats, st, epv);                 \
    }
#define FWD_CASE8C(NT_)                                                            \
    if (p.nt == NT_ && p.cs == 2 && p.maxs == 2 && p.nw == 8 && p.slj == 2)       \
        return launch_mfma<NT_, 2, 2, 2, 8>(t, p, X, W, bias, Y, nIn, init, stats, st, epv);
#define FWD_BOTH_SLJ(NT_, CS_, MAXS_) FWD_CASE(NT_, CS_, 1, MAXS_) FWD_CASE(NT_, CS_, 2, MAXS_)

int64_t conv_stats_parts(const ConvPlan &p, const TiledRules &t) {
    if (t.n_dst == 0 || t.n_sub == 0) return 0;
    if (p.kind == 1) {
        const int cout = p.nt * p.cs * 16;
        return (int64_t)t.n_tiles * (256 / cout > 0 ? 256 / cout : 1);
    }
    if (p.kind == 4) {  // the workgroup-tile epilogue over all nw * 16 columns
        const int cout = p.nw * 16;
        return (int64_t)t.n_tiles * (256 / cout > 0 ? 256 / cout : 1);
    }
    if (p.kind == 3 && p.cs) {  // shared-tile ring kernel: the workgroup-tile epilogue (conv_tile.h: tile_stats)
        const int cout = p.nt * 16;
        return (int64_t)t.n_tiles * (256 / cout > 0 ? 256 / cout : 1);
    }
    if (p.kind == 2 || p.kind == 3) {
        const int nc = p.nt * 16;
        return (int64_t)t.n_tiles * (64 / nc > 0 ? 64 / nc : 1);
    }
    return 0;
}

// which plans apply an inference epilogue in their store: the MFMA tile and wave kernels and the small-nIn (stem) kernel
bool conv_epilogue_supported(int nIn, int nOut, int K, int64_t n_dst, const TiledRules &t) {
    if (t.n_dst == 0 || t.n_sub == 0 || nOut % 4) return false;
    const ConvPlan p = plan_conv(nIn, nOut, K, n_dst);
    if (p.kind == 1 || p.kind == 2 || p.kind == 4) return true;
    if (p.kind != 0) return false;
    return nIn <= 4 && (nOut == 16 || nOut == 32 || nOut == 64) && t.K + 1 <= 128 && t.tile_rows % 4 == 0 &&
           ((size_t)t.tile_rows * nOut + (size_t)t.K * nIn * nOut) * sizeof(float) <= 160 * 1024;
}

int conv_forward(const TiledRules &t, const float *X, const float *W, const float *bias, float *Y, int nIn, int nOut,
                 hipStream_t st, const float *init, double *stats, const ConvEpilogue *ep) {
    if (t.n_dst == 0) return 0;
    if (init && bias) SEI_FAIL("conv_forward: bias and init are mutually exclusive");
    if (ep && !conv_epilogue_supported(nIn, nOut, t.K, t.n_dst, t)) SEI_FAIL("conv_forward: this plan takes no inference epilogue");
    const ConvEpilogue epv = ep ? *ep : ConvEpilogue{};
    if (t.n_sub == 0 && !init) {
        int64_t tot = t.n_dst * nOut;
        fill_bias_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(Y, t.n_dst, nOut, bias);
        SEI_LAUNCH_CHECK();
        return 0;
    }
    ConvPlan p = plan_conv(nIn, nOut, t.K, t.n_dst);
    if (init && (p.kind == 0 || t.n_sub == 0)) SEI_FAIL("conv_forward: init is only supported by the MFMA kernels");
    if (p.tile_rows != t.tile_rows) SEI_FAIL("conv_forward: rulebook tile size does not match the kernel plan");
    if (stats && (p.kind == 0 || t.n_sub == 0)) SEI_FAIL("conv_forward: statistics are only produced by the MFMA kernels");
    if (p.kind == 2) return conv_forward_wave(t, p, X, W, bias, Y, nIn, nOut, init, stats, st, epv);
    if (p.kind == 3) return conv_forward_band(t, p, X, W, bias, Y, nIn, nOut, init, stats, st);
    if (p.kind == 4) return conv_forward_cols(t, p, X, W, bias, Y, nIn, nOut, init, stats, st, epv);
    if (p.kind == 1 && (p.g == 2 || p.g == 3)) {
        if (nIn % 16 || nIn < 64) SEI_FAIL("conv_forward: the round kernel needs nIn % 16 == 0 and nIn >= 64");
#define ROUND_CASE(NT_) if (p.nt == NT_) return launch_round<NT_>(t, p, X, W, bias, Y, nIn, init, stats, st, epv);
        ROUND_CASE(3) ROUND_CASE(4) ROUND_CASE(5) ROUND_CASE(6)
#undef ROUND_CASE
        SEI_FAIL("conv_forward: no round-kernel instantiation for this plan");
    }
    if (p.kind == 1) {
        // eight-wave workgroups: 96 columns in one block, 128 / 160 (and 96 behind a 16-plane input) in two column blocks
        FWD_CASE8(3) FWD_CASE8(4) FWD_CASE8(5) FWD_CASE8S2(6)
        // default plans
        FWD_BOTH_SLJ(1, 1, 2) FWD_BOTH_SLJ(2, 1, 2) FWD_BOTH_SLJ(3, 1, 2) FWD_BOTH_SLJ(4, 1, 2) FWD_BOTH_SLJ(5, 1, 2)
        FWD_BOTH_SLJ(6, 1, 2)
        FWD_BOTH_SLJ(1, 1, 1) FWD_BOTH_SLJ(2, 1, 1) FWD_BOTH_SLJ(3, 1, 1) FWD_BOTH_SLJ(4, 1, 1) FWD_BOTH_SLJ(5, 1, 1)
        FWD_BOTH_SLJ(6, 1, 1)
        FWD_BOTH_SLJ(7, 1, 1) FWD_BOTH_SLJ(8, 1, 1) FWD_BOTH_SLJ(9, 1, 1) FWD_BOTH_SLJ(10, 1, 1) FWD_BOTH_SLJ(11, 1, 1)
        FWD_BOTH_SLJ(12, 1, 1) FWD_BOTH_SLJ(13, 1, 1) FWD_BOTH_SLJ(14, 1, 1) FWD_BOTH_SLJ(15, 1, 1) FWD_BOTH_SLJ(16, 1, 1)
#ifdef SEI_EXPERIMENTS  // alternates reachable through SEI_FWD_PLAN
        FWD_CASE8(2) FWD_CASE8(6) FWD_CASE8(8) FWD_CASE8(10) FWD_CASE8C(2) FWD_CASE8C(3) FWD_CASE8C(4) FWD_CASE8C(5)
        FWD_CASE8M2(2) FWD_CASE8M2(3) FWD_CASE8M2(4) FWD_CASE8M2(5) FWD_CASE8M2(6)
        FWD_CASE(2, 1, 2, 4) FWD_CASE(4, 1, 2, 4) FWD_CASE(6, 1, 2, 4) FWD_CASE(2, 1, 1, 4) FWD_CASE(4, 1, 1, 4) FWD_CASE(6, 1, 1, 4)
        FWD_CASE(2, 1, 2, 1) FWD_CASE(4, 1, 2, 1) FWD_CASE(6, 1, 2, 1) FWD_CASE(2, 1, 2, 3) FWD_CASE(4, 1, 2, 3)
        FWD_CASE(4, 2, 2, 2) FWD_CASE(5, 2, 2, 2) FWD_CASE(6, 2, 2, 2) FWD_CASE(3, 2, 2, 2)
        FWD_CASE(4, 2, 2, 4) FWD_CASE(5, 2, 2, 4) FWD_CASE(6, 2, 2, 4) FWD_CASE(3, 2, 2, 4) FWD_CASE(2, 2, 2, 4) FWD_CASE(1, 2, 2, 4)
        FWD_CASE(3, 1, 2, 3) FWD_CASE(3, 1, 2, 4) FWD_CASE(5, 1, 2, 1) FWD_CASE(3, 1, 2, 1) FWD_CASE(5, 1, 1, 1) FWD_CASE(3, 1, 1, 1) FWD_CASE(1, 1, 2, 1) FWD_CASE(1, 1, 1, 1)
        FWD_CASE(8, 1, 2, 2) FWD_CASE(4, 2, 1, 2) FWD_CASE(5, 2, 1, 2) FWD_CASE(6, 2, 1, 2)
#endif
        SEI_FAIL("conv_forward: no MFMA instantiation for this plan");
    }
    if (nIn <= 4 && (nOut == 16 || nOut == 32 || nOut == 64) && t.K + 1 <= 128 && t.tile_rows % 4 == 0) {
        size_t lds_small = ((size_t)t.tile_rows * nOut + (size_t)t.K * nIn * nOut) * sizeof(float);
        if (lds_small <= 160 * 1024) {
#define SMF(CIN_, COUT_)                                                                                           \
    do {                                                                                                           \
        SEI_TRY(ensure_max_lds((const void *)conv_tile_smallcin_kernel<CIN_, COUT_>, 160 * 1024));                 \
        conv_tile_smallcin_kernel<CIN_, COUT_><<<t.n_tiles, 256, lds_small, st>>>(                                 \
            X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src, t.sub_dst, t.n_dst, t.K, t.tile_rows, epv);    \
    } while (0)
#define SMFC(CIN_) do { if (nOut == 16) SMF(CIN_, 16); else if (nOut == 32) SMF(CIN_, 32); else SMF(CIN_, 64); } while (0)
            if (nIn == 1) SMFC(1); else if (nIn == 2) SMFC(2); else if (nIn == 3) SMFC(3); else SMFC(4);
#undef SMFC
#undef SMF
            SEI_LAUNCH_CHECK();
            return 0;
        }
    }
    if (p.lds_bytes > 160 * 1024) SEI_FAIL("conv: nOut too large for the generic kernel");
    SEI_TRY(ensure_max_lds((const void *)conv_tile_generic_kernel, 160 * 1024));
    conv_tile_generic_kernel<<<t.n_tiles, 256, p.lds_bytes, st>>>(X, W, bias, Y, t.tile_sub_start, t.tile_seg, t.sub_src,
                                                                 t.sub_dst, t.n_dst, t.K, nIn, nOut, t.tile_rows);
    SEI_LAUNCH_CHECK();
    return 0;
}

}  // namespace sei

#ifndef SEI_EXPERIMENTS
extern "C" int sei_debug_band_timeouts(int32_t *out, int32_t reset) {
    (void)reset;
    if (out) *out = 0;   // no polled hand-off in the shipped library
    return 0;
}
#endif

#ifdef SEI_CLOCK
extern "C" int sei_debug_clock(unsigned long long *out, int reset) {   // [8 families][memtime ticks, memrealtime ticks]
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sei::g_clk), sizeof(sei::g_clk)) != hipSuccess) return 1;
    if (reset) {
        static unsigned long long z[8][2] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_clk), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

#ifdef SEI_TRACE
extern "C" int sei_debug_trace(void *out, long long bytes, int reset) {
    if (out && bytes > 0) {
        if ((size_t)bytes > sizeof(sei::TraceBuf)) bytes = sizeof(sei::TraceBuf);
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sei::g_trace), (size_t)bytes) != hipSuccess) return 1;
    }
    if (reset) {
        static sei::TraceBuf z;   // zero-initialised
        z.cu_key = -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_trace), &z, sizeof(int) * 2) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" long long sei_debug_trace_bytes(void) { return (long long)sizeof(sei::TraceBuf); }
#endif

#ifdef SEI_STAMPS
extern "C" int sei_debug_mround_stamps(unsigned long long *out, int reset) {   // [16 waves][16 slots]
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sei::g_mstamps), sizeof(sei::g_mstamps)) != hipSuccess) return 1;
    if (reset) {
        static unsigned long long z[16][16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_mstamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int sei_debug_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sei::g_stamps), sizeof(sei::g_stamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sei::g_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
