// Internal declarations shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sparseeventid_amd.h"

namespace sei {

// ---------------------------------------------------------------- errors
void set_error(const std::string &msg);
#define SEI_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            sei::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " +       \
                           __FILE__ + ":" + std::to_string(__LINE__));                        \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)
#define SEI_FAIL(msg)            \
    do {                         \
        sei::set_error(msg);     \
        return 1;                \
    } while (0)
#define SEI_TRY(expr)            \
    do {                         \
        int _r = (expr);         \
        if (_r) return _r;       \
    } while (0)
#define SEI_LAUNCH_CHECK() SEI_HIP(hipGetLastError())

// ---------------------------------------------------------------- device memory pool
// Library-owned memory (hash tables, rulebooks, scratch).  Blocks are recycled on free so the steady state of a
// training loop performs no hipMalloc.  Every block remembers the (device, stream) it was allocated for and is
// handed out again ONLY to allocations on that same device and stream: a scratch block freed right after an
// asynchronous launch is therefore reused in stream order (safe), while work on another stream or device gets
// memory of its own instead of aliasing it.
void *pool_alloc(size_t bytes, hipStream_t st);
void pool_free(void *p);
// Blocks that kernels on OTHER streams may still be reading (everything a metadata owns): freed behind a fence = one
// event per stream the library has been handed, recorded now; reusable once all of them have completed.
typedef void *PoolFence;
PoolFence pool_fence_begin();
void pool_free_fenced(void *p, PoolFence fence);
void pool_fence_end(PoolFence fence);
void note_stream(hipStream_t st);  // remembers a stream the library launches work on (entry points that take a stream)
template <typename T>
inline int pool_alloc_t(T **out, size_t count, hipStream_t st) {
    *out = (T *)pool_alloc(count * sizeof(T) + 16, st);
    if (!*out) {
        set_error("device pool: out of memory");
        return 1;
    }
    return 0;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel); thread-safe (autograd calls in from a
// second thread)
int ensure_max_lds(const void *kernel, int bytes);
// every host <-> device synchronisation the library performs goes through here and is counted
// (sei_debug_sync_count: tests assert how many a training step needs)
hipError_t sync_stream(hipStream_t st);

// ---------------------------------------------------------------- hashing (device + host identical)
constexpr uint64_t EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;
__host__ __device__ inline uint64_t pack_key(int x0, int x1, int x2, int b) {
    return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)x0 << 32) | ((uint64_t)(uint16_t)x1 << 16) |
           (uint64_t)(uint16_t)x2;
}
__host__ __device__ inline uint64_t mix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// Locality-preserving slot choice: the 8 keys that differ only in the low 3 bits of x2 (the fastest coordinate)
// share one 8-slot block = one 64-byte line of `keys`; the block is placed by the hash of the other bits.
// Probes of neighbouring sites along x2 -- what consecutive lanes of the rulebook probe kernel issue -- then hit
// the same cache line.  Probe t of a key visits the other 7 slots of its home block first, then walks the following
// blocks the same way: the sequence covers EVERY slot of the table in `cap` steps (a sequence that only stepped by
// whole blocks would confine a key to cap/8 slots and spin forever once more than cap/8 keys share x2 mod 8 --
// planar clouds, spatial size [H,W,1]).  All probe loops are bounded by cap.
__host__ __device__ inline uint64_t hash_slot0(uint64_t key, uint64_t mask) {
    return ((mix64(key >> 3) << 3) | (key & 7ull)) & mask;
}
__host__ __device__ inline uint64_t hash_probe(uint64_t s0, uint64_t t, uint64_t mask) {
    return (((s0 & ~7ull) + ((t >> 3) << 3)) & mask) | ((s0 + t) & 7ull);
}

// bit of `key` in a level's membership filter (struct HashTable below)
__host__ __device__ inline uint64_t filter_bit(uint64_t key, uint64_t fmask) {
    return ((mix64(key >> 5) << 5) | (key & 31ull)) & fmask;
}

#ifdef __HIPCC__
// Insert key; the slot's value becomes the MIN candidate id that hit it (deterministic representative = first
// appearance).  Returns the slot, or 0xFFFFFFFF when the table is full (callers raise the device error flag).
__device__ inline uint32_t hash_insert_min(uint64_t *keys, uint32_t *vals, uint64_t mask, uint64_t key, uint32_t cand) {
    const uint64_t s0 = hash_slot0(key, mask);
    for (uint64_t t = 0; t <= mask; t++) {
        const uint64_t s = hash_probe(s0, t, mask);
        unsigned long long prev = atomicCAS((unsigned long long *)&keys[s], (unsigned long long)EMPTY_KEY,
                                            (unsigned long long)key);
        if (prev == EMPTY_KEY || prev == key) {
            atomicMin(&vals[s], cand);
            return (uint32_t)s;
        }
    }
    return 0xFFFFFFFFu;
}
__device__ inline void filter_set(uint32_t *filter, uint64_t fmask, uint64_t key) {
    const uint64_t b = filter_bit(key, fmask);
    atomicOr(&filter[b >> 5], 1u << (b & 31));
}
__device__ inline bool filter_test(const uint32_t *__restrict__ filter, uint64_t fmask, uint64_t key) {
    const uint64_t b = filter_bit(key, fmask);
    return (filter[b >> 5] >> (b & 31)) & 1u;
}
__device__ inline int hash_find(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t mask,
                                uint64_t key) {
    const uint64_t s0 = hash_slot0(key, mask);
    for (uint64_t t = 0; t <= mask; t++) {
        const uint64_t s = hash_probe(s0, t, mask);
        const uint64_t k = keys[s];
        if (k == key) return (int)vals[s];
        if (k == EMPTY_KEY) return -1;
    }
    return -1;
}
#endif

// Membership filter beside the table: one bit per key at filter_bit(key) in a bitmap of 8 bits per table slot (>= 16 bits
// per key, false-positive rate <= 6 %).  The rulebook probes test it first: 92 % of the stem's 125 probes per site are
// misses, and a miss then costs one word of a bitmap that is 12x smaller than the table (8 MB at level 0: L2 / MALL
// resident) instead of a 64-byte line of the 100 MB table.  The 32 keys that differ only in the low 5 bits of x2 share one
// 32-bit word, so the lanes of a probe wave (neighbouring sites along x2) share words.
struct HashTable {
    uint64_t *keys = nullptr;
    uint32_t *vals = nullptr;
    uint64_t mask = 0;
    uint32_t *filter = nullptr;
    uint64_t fmask = 0;  // bits in the filter - 1
};


// One resolution level: active sites of one spatial size.
struct Level {
    int32_t spatial[3] = {0, 0, 0};
    int64_t n = 0;              // rows (-1 between build_strided_level_begin and _finish: known to the device only)
    const int32_t *n_dev = nullptr;   // while unfinished: the row count on the device, and its upper bound on the host
    int64_t n_ub = 0;
    int32_t *coords = nullptr;  // [n,4] device
    HashTable ht;
    bool failed = false;        // its discovery failed after it was registered (hash table full, lost synchronisation): every
                                // later lookup of the level or of the rulebook that created it reports that instead of using it
};

// ---------------------------------------------------------------- rulebook
constexpr int SUB_SLOTS = 16;  // rule slots per sub-chunk = rows of one v_mfma_f32_16x16x4_f32

// Rules organised for output-stationary execution: dst rows are cut into tiles of `tile_rows`
// (256 or 128) consecutive rows, one workgroup per tile; within a tile the rules are grouped by
// filter offset k, compacted, and padded to sub-chunks of 16 slots (src = -1 marks padding).
struct TiledRules {
    int64_t n_dst = 0, n_src = 0;
    int32_t K = 0;
    int32_t tile_rows = 0;
    int32_t n_tiles = 0;
    int64_t n_sub = 0;                  // sub-chunks (MFMA row blocks executed)
    int64_t n_rules = 0;
    int32_t *tile_sub_start = nullptr;  // [n_tiles+1] first sub-chunk of tile
    int32_t *tile_seg = nullptr;        // [n_tiles][K+1] tile-relative sub-chunk offset where offset k begins
    int32_t *seg_start = nullptr;       // [(K+1)][n_tiles] same, k-major (dW walks one k over many tiles)
    int32_t *sub_src = nullptr;         // [n_sub*16] src row or -1
    uint8_t *sub_dst = nullptr;         // [n_sub*16] dst row local to tile
    bool built = false;
    bool fwd_use = false;               // fetched by a FORWARD convolution (sei_metadata_prebuild_forward_like builds only those)
    bool building = false;              // begun by build_tiled_batch, not finished yet (rb->mu is released in between)
};
void free_tiled(TiledRules &t);

struct ProbeGeom {
    int mode;        // 0 submanifold, 1 strided forward (dst = out sites), 2 strided transposed (dst = in sites)
    int f0, f1, f2;  // filter size
    int t0, t1, t2;  // stride
    int s0, s1, s2;  // spatial size of the SRC level
};

}  // namespace sei

constexpr int SEI_NVARIANT = 4;  // tile_rows 64 / 128 / 192 / 256
struct sei_rulebook {
    sei::TiledRules fwd[SEI_NVARIANT];  // dst = output rows, src = input rows
    sei::TiledRules bwd[SEI_NVARIANT];  // dst = input rows,  src = output rows (strided conv only)
    int32_t *nbr_fwd = nullptr, *nbr_bwd = nullptr;  // k-major neighbour tables [K][n_dst], kept for later tiled variants
    sei_rulebook *donor = nullptr;      // submanifold rulebook of the SAME level with a larger odd filter: its table contains ours
    bool submanifold = true;            // bwd == fwd with mirrored k
    int32_t spatial_in[3], spatial_out[3], size[3], stride[3];
    sei::Level *lin = nullptr, *lout = nullptr;
    std::mutex mu;
    std::condition_variable cv;         // signalled when a variant leaves the `building` state
};

struct sei_metadata {
    std::mutex mu;
    int32_t batch_size = 0;
    int32_t input_mode = 3;
    int64_t n_input = 0;          // raw input points
    int32_t *input_row = nullptr; // [n_input] row of each input point
    int32_t *row_first = nullptr; // [n_rows] first input index of row
    int32_t *row_last = nullptr;  // [n_rows]
    int32_t *row_count = nullptr; // [n_rows]
    int32_t *dup_order = nullptr; // [n_input] point indices sorted by row, input order inside a row (only with duplicate points)
    int32_t *dup_start = nullptr; // [n_rows+1] first entry of each row in dup_order
    std::vector<sei::Level *> levels;
    std::vector<sei_rulebook *> rulebooks;
};

namespace sei {
Level *find_level(sei_metadata *m, const int32_t spatial[3]);

// ---- kernels' host launchers (one per translation unit) ----
// hash.hip
int alloc_table(HashTable *ht, int64_t n, hipStream_t st);
int build_input_level(sei_metadata *m, const int32_t *coords, int64_t n, const int32_t spatial[3], int32_t batch,
                      hipStream_t st, int64_t *n_rows);
// the same in two halves around its count read-back (hash.hip): strided levels begun between the halves read the input
// level's row count from the device, and ONE stream synchronisation serves the InputLayer and the whole chain of levels
struct PendingInput {
    Level *level = nullptr;
    uint32_t *slot_of = nullptr, *new_vals = nullptr;
    int32_t *flag = nullptr, *err_total = nullptr;   // device: [0] first out-of-range point, [1] rows, [2] table-full flag
    unsigned long long *host = nullptr;              // pinned staging block of those words
    const int32_t *coords = nullptr;
    int64_t n = 0;
    bool pending = false;
};
int build_input_level_begin(sei_metadata *m, const int32_t *coords, int64_t n, const int32_t spatial[3], int32_t batch,
                            hipStream_t st, PendingInput *pi);
int build_input_level_finish(sei_metadata *m, PendingInput *pi, hipStream_t st, int64_t *n_rows);
// pinned staging blocks for count read-backs (rulebook.hip), recycled
unsigned long long *stage_get();
void stage_put(unsigned long long *p);
int input_features_fwd(sei_metadata *m, const float *in, int C, float *out, hipStream_t st);
int input_features_bwd(sei_metadata *m, const float *dout, int C, float *din, hipStream_t st);
int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, int32_t *total_dev, hipStream_t st);
int larcv_compact(const float *in, int B, int P, int M, int D, int32_t *coords, float *feats, int64_t *n_out,
                  hipStream_t st);
// rulebook.hip
inline int variant_of(int tile_rows) { return tile_rows / 64 - 1; }
// returns the (lazily built) tiled rules of the rulebook: transposed=false -> fwd, true -> bwd
// forward_use: the caller is a forward convolution (marks the variant for forward-only prebuilds)
int get_tiled(sei_rulebook *rb, bool transposed, int tile_rows, const TiledRules **out, hipStream_t st,
              bool forward_use = false);
// any already-built variant (the smallest tile size first), building the 128-row one if none exists
int get_any_tiled(sei_rulebook *rb, bool transposed, const TiledRules **out, hipStream_t st);
// builds every listed variant that does not exist yet with ONE stream synchronisation for all of them
struct TiledRequest {
    sei_rulebook *rb;
    bool transposed;
    int tile_rows;
    bool forward_use = false;   // carried over from the template's variant
};
int build_tiled_batch(const std::vector<TiledRequest> &want, hipStream_t st);
int build_strided_level(sei_metadata *m, Level *lin, const int32_t size[3], const int32_t stride[3], sei_rulebook *rb,
                        hipStream_t st);
// the same in two halves around the site-count read-back (rulebook.hip): a chain of levels begun back to back needs ONE
// stream synchronisation before they are finished in order
struct PendingLevel {
    Level *level = nullptr;
    int32_t *total = nullptr;            // device: [0] output sites, [1] table-full flag
    unsigned long long *host = nullptr;  // pinned staging block the two words are copied to
    bool pending = false;
};
int build_strided_level_begin(sei_metadata *m, Level *lin, const int32_t size[3], const int32_t stride[3], sei_rulebook *rb,
                              hipStream_t st, PendingLevel *pl);
int build_strided_level_finish(PendingLevel *pl);
// instead of finish when the read-back cannot be trusted (the stream synchronisation failed): releases what finish would
// and marks the level failed
void build_strided_level_abort(PendingLevel *pl);
int export_rules(const TiledRules &t, int64_t *start_host, int32_t *rin, int32_t *rout, hipStream_t st);
// conv_fwd.hip / conv_wave.hip
// How one convolution (forward, or dX with the plane counts swapped) runs: decided ONCE per call from the plane
// counts, the filter volume of the rulebook and the dst row count, and threaded through rulebook build, weight
// transform and launch.
struct ConvPlan {
    int kind = 0;        // 0 scalar tile kernels (tiny / odd plane counts), 1 workgroup-tile MFMA kernel, 2 wave-private MFMA kernel,
                         // 3 row-band MFMA kernel (nw = compute waves per workgroup, g = ring depth in slices),
                         // 4 column-owner MFMA kernel (conv_cols.hip; weights packed per column tile: col_group = -1)
    int tile_rows = 0;   // rulebook variant (dst rows per tile)
    int col_group = 0;   // kind 1: width of the column groups whose columns are stored permuted (0 = natural layout)
    int blocks = 1;      // column blocks (kind 1: weights stored as [block][K][nIn][nOut/blocks])
    int nt = 0, cs = 0, slj = 0, maxs = 0, g = 0;   // kind 1, g == 2: two-offset rounds (conv_round_mfma_kernel), g == 3: with refilled A fragments
    int nw = 4;          // kind 1: waves per workgroup (4, 8 or 16)
    bool split = false;  // kind 1, g == 3: products as three-way bf16 splits (sei_set_product_mode(1)); weights from conv_split_weights
    bool persist = false; // kind 1, g == 3: one workgroup per CU walks the work items (conv_mround_mfma_kernel<.., PERSIST>)
    int nb = 2;          // kind 1, g == 3: weight buffers of the round kernel (3: slices staged two phases ahead)
    int mo = 0;          // kind 1, g == 2 / 3, nw == 16: offsets a round may span (2: conv_round_mfma_kernel, 3 / 4: conv_mround_mfma_kernel)
    size_t lds_bytes = 0;
};
ConvPlan plan_conv(int nIn, int nOut, int K, int64_t n_dst);
// sei_debug_force_conv_kernel(): pins the forward / dX kernel family for every later call of the process
// (bit-identity tests between the kernels, sweeps); AUTO = the plan picks
// TILE: the workgroup-tile kernel with one offset per phase everywhere; ROUNDS: its two-offset-round form wherever it is instantiated
// (AUTO uses that form for the eight-wave plans); BAND / COLS: experiment builds only
enum { CONV_KERNEL_AUTO = 0, CONV_KERNEL_TILE = 1, CONV_KERNEL_WAVE = 2, CONV_KERNEL_BAND = 3, CONV_KERNEL_COLS = 4, CONV_KERNEL_ROUNDS = 5 };
int forced_conv_kernel();
int product_mode();   // sei_set_product_mode(): 0 fp32 MFMA (default), 1 three-way bf16 split (opt-in)
// sei_debug_force_dw_kernel(): dW kernel family (AUTO = the library's choice)
enum { DW_KERNEL_AUTO = 0, DW_KERNEL_REG = 1, DW_KERNEL_RING = 2 };
int forced_dw_kernel();
int dw_ring_timeouts(int *out, bool reset);
bool plan_wave(int nIn, int nOut, int K, int64_t n_dst, ConvPlan *p);
bool conv_rounds_64(int nIn, int nOut, int K, int64_t n_dst);   // conv_fwd.hip: the 64 -> 64 layers that take the multi-offset rounds
bool plan_band(int nIn, int nOut, int K, int64_t n_dst, ConvPlan *p);
bool plan_cols(int nIn, int nOut, int K, int64_t n_dst, ConvPlan *p);
// stats (optional, MFMA kernels only): per-(tile, part) column sums of the OUTPUT rows, [n_parts][2][nOut] doubles
// (sum, sum of squares) accumulated in the kernel epilogue from the LDS tile -- BatchNorm statistics without a second
// read of the conv output.  conv_stats_parts() = how many parts this plan writes (0: the plan cannot).
// Inference epilogue (eval-mode BatchNorm + residual + LeakyReLU applied while the finished tile is written): the rows go
// out as  leaky((y - mean) * (invstd * gamma) + beta (+ res))  -- the expression of bn_apply_kernel, so that the fused
// write equals the two-pass result bit for bit -- and the convolution output itself never reaches memory.
struct ConvEpilogue {
    const float *mean = nullptr, *invstd = nullptr, *gamma = nullptr, *beta = nullptr, *res = nullptr;
    float leak = 1.f;
};
__device__ __forceinline__ float4 conv_epilogue_apply(const ConvEpilogue &ep, float4 v, int c, const float *res_row) {
    const float4 mu = *reinterpret_cast<const float4 *>(ep.mean + c);
    const float4 is = *reinterpret_cast<const float4 *>(ep.invstd + c);
    const float4 ga = ep.gamma ? *reinterpret_cast<const float4 *>(ep.gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 be = ep.beta ? *reinterpret_cast<const float4 *>(ep.beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float z[4] = {(v.x - mu.x) * (is.x * ga.x) + be.x, (v.y - mu.y) * (is.y * ga.y) + be.y,
                  (v.z - mu.z) * (is.z * ga.z) + be.z, (v.w - mu.w) * (is.w * ga.w) + be.w};
    if (res_row) {
        const float4 r = *reinterpret_cast<const float4 *>(res_row + c);
        z[0] += r.x; z[1] += r.y; z[2] += r.z; z[3] += r.w;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) z[i] = z[i] > 0.f ? z[i] : z[i] * ep.leak;
    return make_float4(z[0], z[1], z[2], z[3]);
}
// ep (optional): kinds 1 and 2 and the small-nIn kernel only -- conv_epilogue_supported() says whether a plan takes it
bool conv_epilogue_supported(int nIn, int nOut, int K, int64_t n_dst, const TiledRules &t);
int conv_forward(const TiledRules &t, const float *X, const float *W, const float *bias, float *Y, int nIn, int nOut,
                 hipStream_t st, const float *init = nullptr, double *stats = nullptr, const ConvEpilogue *ep = nullptr);
int conv_forward_wave(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                      int nIn, int nOut, const float *init, double *stats, hipStream_t st,
                      const ConvEpilogue &ep = ConvEpilogue{});
int conv_forward_band(const TiledRules &t, const ConvPlan &p, const float *X, const float *W, const float *bias, float *Y,
                      int nIn, int nOut, const float *init, double *stats, hipStream_t st);
int conv_forward_cols(const TiledRules &t, const ConvPlan &p, const float *X, const float *Wp, const float *bias, float *Y,
                      int nIn, int nOut, const float *init, double *stats, hipStream_t st, const ConvEpilogue &ep);
int64_t conv_stats_parts(const ConvPlan &p, const TiledRules &t);
// Wout[k][r][perm(c)]: transpose=false: rows=nIn, cols=nOut (forward); transpose=true: Wout[k][co][perm(ci)] = W[ks][ci][co]
// with ks = flip ? K-1-k : k (dX).  perm = column permutation inside groups of `col_group` columns (0 = identity).
int conv_transform_weights(const float *W, float *Wout, int K, int nIn, int nOut, bool transpose, bool flip, int col_group, int blocks,
                           hipStream_t st);
// Ws[block][k][s][part][n][lane][8 bf16] <- the three bf16 parts of Wnat[k][32 s + 8 (lane >> 4) + j][16 (block nt + n) + (lane & 15)],
// Wnat = fp32 weights in the natural [K][rows][cols] layout (conv_transform_weights with col_group 0, blocks 1); Ws holds
// K * rows * cols * 3 / 2 floats
int conv_split_weights(const float *Wnat, float *Ws, int K, int rows, int cols, int nt, hipStream_t st);
int conv_backward_weight(const TiledRules &t, const float *X, const float *dY, float *dW, int nIn, int nOut,
                         hipStream_t st);
// elementwise.hip
int colsum(const float *x, int64_t n, int C, float *out, hipStream_t st);
// stats / n_parts: precomputed partial sums of x (conv epilogue) -- training mode then skips its own reduction pass
// eval-mode statistics: mean = running_mean, invstd = 1 / sqrt(running_var + eps) (the values bn_forward's eval path uses)
int bn_eval_stats(const float *rmean, const float *rvar, float eps, int C, float *mean, float *invstd, hipStream_t st);
int bn_forward(const float *x, int64_t n, int C, const float *gamma, const float *beta, float *rmean, float *rvar,
               float eps, float momentum, int training, float leak, const float *residual, float *y, float *save_mean,
               float *save_invstd, hipStream_t st, const double *stats = nullptr, int64_t n_parts = 0);
// y == nullptr (only without a residual; needs beta): the activation mask is recomputed from x -- one read less per pass
int bn_backward(const float *x, const float *y, const float *dy, int64_t n, int C, const float *gamma,
                const float *save_mean, const float *save_invstd, float leak, int training, float *dx, float *dres,
                float *dgamma, float *dbeta, hipStream_t st, float *dx_colsum = nullptr, const float *beta = nullptr);
int leaky_forward(const float *x, float *y, int64_t count, float leak, hipStream_t st);
int leaky_backward(const float *y, const float *dy, float *dx, int64_t count, float leak, hipStream_t st);
int add_forward(const float *a, const float *b, float *out, int64_t count, float leak, bool act, hipStream_t st);
int sparse_to_dense(const Level *lv, int batch, const float *feats, int C, float *dense, bool backward,
                    hipStream_t st);
int pool_tanh(const Level *lv, int batch, const float *feats, int C, float *out, const float *dout, float *dfeats,
              hipStream_t st);

// Diagnostic build only (make EXTRA="-DSEI_EXPERIMENTS -DSEI_CLOCK"; tools/kernel_clock.py): the shader clock a kernel HOLDS while
// it runs (MI355X_MICROARCH.md, DVFS give-back item 6): wave 0 of every workgroup adds the s_memtime ticks (shader cycles) and
// the s_memrealtime ticks (100 MHz) of its lifetime to g_clk[family]; clock = 100 MHz x sum(memtime) / sum(memrealtime).
// The stamps go to a buffer nothing else reads; the product build compiles none of it.
#ifdef SEI_CLOCK
// (one array per translation unit: device symbols do not link across objects without relocatable device code)
#define CLK_BEGIN unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime()
#define CLK_END(fam) do { if (threadIdx.x == 0) { atomicAdd(&g_clk[fam][0], __builtin_amdgcn_s_memtime() - clk_t0); \
                                                  atomicAdd(&g_clk[fam][1], __builtin_amdgcn_s_memrealtime() - clk_r0); } } while (0)
#else
#define CLK_BEGIN
#define CLK_END(fam)
#endif

// ---- profiling (HIP events around launches) ----
// kinds 6 / 7: forward / dX launches of the 32 -> 32 plane layers (level 0 of the DUNE network), kept apart because they
// are priced against the HBM roofline, not the MFMA one
enum ProfKind { PROF_CONV_FWD = 0, PROF_CONV_BWD_IN = 1, PROF_CONV_BWD_W = 2, PROF_RULES = 3, PROF_BN = 4, PROF_OTHER = 5,
                PROF_CONV_FWD_C32 = 6, PROF_CONV_BWD_IN_C32 = 7, PROF_NKIND = 8 };
struct ProfScope {
    int kind;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double flops, bytes;
    ProfScope(int kind, hipStream_t st, double flops, double bytes);
    ~ProfScope();
};
}  // namespace sei
