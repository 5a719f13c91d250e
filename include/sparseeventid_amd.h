/*
 * sparseeventid_amd.h -- C ABI of the MI355X (gfx950) submanifold sparse-convolution engine.
 *
 * This is the drop-in boundary for the ONE hot path of coreyjadams/SparseEventID:
 * the `sparseconvnet` (scn) operators composed by the reference network
 *   /root/reference/src/networks/resnet.py:26-50,105-110,124-125,141-161
 *   /root/reference/src/networks/sparse_building_blocks.py:29-34,39,80,96,110-117,122
 * The reference binds these through the third-party python package `sparseconvnet`
 * (its pybind11 extension `sparseconvnet.SCN`, absent from the reference tree); each
 * entry point below names the scn operator / SCN function family it replaces.
 *
 * Conventions
 *  - plain C: pointers + sizes, no torch types.  All `*_dev` / feature pointers are
 *    DEVICE pointers owned by the caller (PyTorch); the library only borrows them for
 *    the duration of the call.  Metadata / rulebooks are owned by the library.
 *  - every function returns 0 on success, non-zero on error; sei_last_error() returns a
 *    thread-local message (the python binding raises RuntimeError with it).
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it; no hidden
 *    hipDeviceSynchronize.  Functions that must return a count to the host synchronise
 *    that stream only (documented per function).
 *  - handles carry no thread-local state: safe to use from the autograd thread.
 *  - coordinates: int32 [N,4] = (x0, x1, x2, batch), the layout the reference transform
 *    emits (src/io/data_transforms.py:36-46).
 *  - features: fp32 row-major [N, C].  weights: fp32 [K, nIn, nOut], k row-major over the
 *    filter (last dim fastest) -- the scn parameter layout (state_dict contract).
 */
#ifndef SPARSEEVENTID_AMD_H
#define SPARSEEVENTID_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sei_metadata sei_metadata_t; /* scn.Metadata: hash maps + rulebooks, all levels */
typedef struct sei_rulebook sei_rulebook_t; /* one (level, filter, stride) rulebook; owned by metadata */
typedef void *sei_stream_t;                 /* hipStream_t */

const char *sei_last_error(void);
int sei_abi_version(void);

/* ---- metadata (replaces scn.Metadata / SCN `Metadata<3>`) ------------------------------- */
int sei_metadata_create(sei_metadata_t **out);
/* Streams: a metadata may be built on one stream (sei_input_build, sei_metadata_prebuild_like) and its rulebooks read by
 * kernels the caller queues on other streams; the caller orders the BUILD before those kernels (event / wait_stream).
 * Destroying it needs no ordering: its device blocks are released behind a fence of one event per stream the library has
 * been handed, recorded at the time of this call, and are reused only when all of them have completed. */
void sei_metadata_destroy(sei_metadata_t *m);

/* scn.InputLayer(dimension=3, spatial_size, mode) coordinate pass (resnet.py:40-43,143;
 * SCN `setInputSpatialLocations` / `inputLayerRules`).  Builds the level-0 hash map on the
 * device from coords_dev[n,4]; duplicates collapse to one row (first-appearance order).
 * mode: 0/2 keep first, 1 keep last, 3 sum, 4 mean.  Out-of-range coordinates are an error.
 * Synchronises `stream` to return *n_rows. */
int sei_input_build(sei_metadata_t *m, const int32_t *coords_dev, int64_t n, const int32_t spatial[3],
                    int32_t batch_size, int32_t mode, sei_stream_t stream, int64_t *n_rows);
/* larcv batch -> scn input tuple on the device (src/io/data_transforms.py:21-49 3-D, :198-253 2-D):
 * larcv_dev fp32 [B,P,M,D+1] with -999 padding (D=3: (x,y,z,value), P=1; D=2: (x,y,value), P planes).
 * Emits coords int32 [n,4] ((x,y,z,batch) batch-major, or (plane,y,x,batch) plane-major -- the reference's
 * order) and features [n,1]; outputs must hold B*P*M entries.  Synchronises `stream` to return *n_points. */
int sei_larcv_to_scn(const float *larcv_dev, int32_t B, int32_t P, int32_t M, int32_t D, int32_t *coords_out_dev,
                     float *features_out_dev, sei_stream_t stream, int64_t *n_points);

/* InputLayer feature pass: out[n_rows,C] from in[n,C] (SCN `InputLayer_updateOutput`). */
int sei_input_features_fwd(sei_metadata_t *m, const float *in, int32_t C, float *out, sei_stream_t stream);
/* ... and its gradient (SCN `InputLayer_updateGradInput`): din[n,C] from dout[n_rows,C]. */
int sei_input_features_bwd(sei_metadata_t *m, const float *dout, int32_t C, float *din, sei_stream_t stream);

/* level queries (scn.SparseConvNetTensor.get_spatial_locations) */
int sei_level_rows(sei_metadata_t *m, const int32_t spatial[3], int64_t *n_rows);
int sei_level_coords(sei_metadata_t *m, const int32_t spatial[3], int32_t *coords_out_dev, sei_stream_t stream);
int sei_batch_size(sei_metadata_t *m, int32_t *batch_size);

/* Builds in `m` (which must already hold its input level) every rulebook, and every tiled variant of it, that the
 * metadata `like` of an EARLIER batch holds -- same filter geometry, this batch's coordinates.  The reference builds
 * rule books lazily inside the forward pass (SCN Metadata::getSubmanifoldRuleBook / getRuleBook), and so does this
 * library by default, with one count read-back (stream synchronisation) per tiled variant, also in the backward pass.
 * Here all variants are begun back to back and share ONE read-back (plus one per new strided level), so the forward and
 * backward passes that follow synchronise nothing.  May run on the caller's main stream at the top of the forward pass, or
 * on a side stream ahead of it.  Fails like the lazy path would if a level of `like` does not exist for this batch. */
int sei_metadata_prebuild_like(sei_metadata_t *m, sei_metadata_t *like, sei_stream_t stream);
/* The same for a pass that will not differentiate (validation between training steps, inference): only the variants a
 * FORWARD convolution of `like` fetched -- not the transposed rules of the strided layers (dX) nor the weight-gradient
 * kernel's own variants. */
int sei_metadata_prebuild_forward_like(sei_metadata_t *m, sei_metadata_t *like, sei_stream_t stream);
/* sei_input_build for a batch whose network is known from the metadata `like` of an earlier batch: the input level and the
 * whole chain of strided levels `like` holds are discovered back to back with their row counts kept on the device, and ONE
 * stream synchronisation gives the host all of them (sei_input_build followed by sei_metadata_prebuild_like needs one for
 * the InputLayer and one for the levels).  A later sei_metadata_prebuild_like(m, like) then only builds the tiled variants
 * (one more synchronisation): two per batch in all.  `like` == NULL: plain sei_input_build. */
int sei_input_build_like(sei_metadata_t *m, const int32_t *coords_dev, int64_t n_points, const int32_t spatial[3],
                         int32_t batch_size, int32_t mode, sei_metadata_t *like, sei_stream_t stream, int64_t *n_rows);

/* ---- rulebooks (SCN `getSubmanifoldRuleBook` / `getRuleBook`) ---------------------------- */
/* Cached in the metadata per (spatial, filter).  Synchronises `stream` on first build. */
int sei_subm_rulebook(sei_metadata_t *m, const int32_t spatial[3], const int32_t filter[3],
                      sei_stream_t stream, sei_rulebook_t **rb);
/* scn.Convolution: discovers the output active set (new level, spatial_out), builds the
 * forward rulebook and its transpose.  Synchronises `stream` on first build. */
int sei_conv_rulebook(sei_metadata_t *m, const int32_t spatial_in[3], const int32_t size[3],
                      const int32_t stride[3], sei_stream_t stream, sei_rulebook_t **rb,
                      int32_t spatial_out[3]);
/* The tiled rule lists are built lazily, per tile size (256/128/64 dst rows) the conv kernels ask for.
 * info[0]=n_dst rows, [1]=n_src rows, [2]=K; for the first built tile size (zeros if none yet):
 * [3]=rules, [4]=sub-chunks (16 rule slots each = MFMA row blocks executed), [5]=tiles, [6]=sub-chunks,
 * [7]=tile rows. */
int sei_rulebook_info(const sei_rulebook_t *rb, int64_t info[8]);
/* force the build of one tile size (tests / inspection).  Synchronises `stream`. */
int sei_rulebook_build(sei_rulebook_t *rb, int32_t tile_rows, sei_stream_t stream);
/* copy the rules out as k-major (in,out) lists for inspection/tests: start[K+1] (host),
 * rules_in/out device int32 [rules].  Synchronises. */
int sei_rulebook_export(sei_rulebook_t *rb, int64_t *start_host, int32_t *rules_in_dev,
                        int32_t *rules_out_dev, sei_stream_t stream);

/* ---- convolution arithmetic (scn.SubmanifoldConvolution / scn.Convolution;
 *      SCN `SubmanifoldConvolution_updateOutput` / `_backward`, `Convolution_*`) ------------ */
/* Y[n_dst,nOut] = bias + sum_k X[rule.in] * W[k]   (bias may be NULL) */
int sei_conv_forward(sei_rulebook_t *rb, const float *X, const float *W, const float *bias, float *Y,
                     int32_t nIn, int32_t nOut, sei_stream_t stream);
/* dX[n_src,nIn] = sum_k dY[rule.out] * W[k]^T */
int sei_conv_backward_input(sei_rulebook_t *rb, const float *dY, const float *W, float *dX,
                            int32_t nIn, int32_t nOut, sei_stream_t stream);
/* dX = dX_init + sum_k dY[rule.out] * W[k]^T : the residual branch's gradient is folded into the convolution's
 * accumulator initialisation instead of a separate [n,nIn] add pass (plane counts of the MFMA kernel only). */
int sei_conv_backward_input_accumulate(sei_rulebook_t *rb, const float *dY, const float *W, const float *dX_init,
                                       float *dX, int32_t nIn, int32_t nOut, sei_stream_t stream);
/* dW[K,nIn,nOut] = sum_rules X[rule.in]^T dY[rule.out]; dbias[nOut] = colsum(dY) if non-NULL */
int sei_conv_backward_weight(sei_rulebook_t *rb, const float *X, const float *dY, float *dW, float *dbias,
                             int32_t nIn, int32_t nOut, sei_stream_t stream);

/* dX and dW of one convolution in one call (either output may be NULL) -- same results as the two calls above (SCN's
 * SubmanifoldConvolution_backward / Convolution_backward compute both in one call as well).  Back to back on `stream` by
 * default; after sei_set_backward_streams(1) the library runs dW on a side stream of its own (one per device and caller
 * stream) next to dX on `stream` and joins before returning to `stream`.  dX_init (optional):
 * dX = dX_init + conv^T(dY), as sei_conv_backward_input_accumulate.  dbias (optional): column sums of dY. */
int sei_conv_backward(sei_rulebook_t *rb, const float *X, const float *dY, const float *W, const float *dX_init,
                      float *dX, float *dW, float *dbias, int32_t nIn, int32_t nOut, sei_stream_t stream);

/* process-wide switch for sei_conv_backward: 0 = back to back (the default); 1 = dW on the library's side stream, joined
 * before the call returns to `stream`; 2 = dW on the side stream with a DEFERRED join: the call returns with dW still
 * queued, later calls queue behind it, and the caller makes `stream` wait for all of them with sei_backward_join()
 * before anything reads a dW.  In mode 2 the caller keeps X, dY and dW of every call allocated until the join; memory
 * released after the join may be reused by work queued on `stream` behind it. */
int sei_set_backward_streams(int32_t mode);
/* process-wide switch for the matrix products of the forward / dX convolutions (the dense per-rule feature x weight contraction;
 * replaces nothing upstream: SCN multiplies in fp32 by sgemm).  0 = v_mfma_f32_16x16x4_f32, exact fp32 products and sums (the
 * default); 1 = OPT-IN: every fp32 operand as an exact sum of three bf16 numbers, the six largest of the nine part products on
 * the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16) with fp32 accumulation -- per product within 2^-23 of the exact one, i.e. the
 * size of one fp32 rounding; applies where the sixteen-wave round kernels run (64 .. 192 planes on large levels), other plans
 * keep mode 0.  Results of mode 1 are deterministic but not bit-identical to mode 0. */
int sei_set_product_mode(int32_t mode);
/* `stream` waits for everything sei_conv_backward queued on its side stream so far (no-op if nothing was) */
int sei_backward_join(sei_stream_t stream);
/* Test / sweep hook: pins the forward / dX kernel family for every later convolution of the process
 * (0 = the plan picks (default), 1 = workgroup-tile kernel with one filter offset per phase everywhere, 2 = wave-private
 * kernel, 5 = the tile kernel's two-offset-round form wherever it is instantiated (the default uses it for the 96 / 128 /
 * 160-column layers of large levels); 3 = row-band kernel and 4 = column-owner kernel: experiment builds only, the shipped
 * library returns an error for them).  The families produce bit-identical results (same per-row summation order);
 * tests/test_gpu_kernels_agree.py asserts it. */
int sei_debug_force_conv_kernel(int32_t kind);
/* Same for the dW kernel: 0 = the library's choice, 1 = register-operand kernel, 2 = LDS-operand (ring) kernel -- an
 * experiment that measured slower and is compiled only with EXTRA=-DSEI_EXPERIMENTS (the shipped library returns an
 * error for 2).  Both are deterministic; their fp32 summation orders differ (results agree to rounding). */
int sei_debug_force_dw_kernel(int32_t kind);
/* The row-band kernels' (and the LDS-operand dW kernel's) loader / compute waves hand data over through LDS words they poll; every poll loop is
 * bounded, and a loop that ran out (it never should) is counted here instead of hanging the GPU.  Synchronises the device. */
int sei_debug_band_timeouts(int32_t *count, int32_t reset);

/* ---- BatchNormalization (+fused LeakyReLU, +fused residual AddTable)
 *      (scn.BatchNormalization / BatchNormLeakyReLU / BatchNormReLU + scn.AddTable + scn.LeakyReLU;
 *       sparse_building_blocks.py:39,80,96,122; SCN `BatchNormalization_updateOutput/_backward`) */
/* training: batch stats over all n rows (biased var, eps inside rsqrt), running stats updated
 *   r = momentum*r + (1-momentum)*batch (running var unbiased).  eval: uses running stats.
 * y = leaky(gamma*(x-mean)*invstd + beta [+ residual], leak);  leak==1 -> identity.
 * save_mean/save_invstd [C] receive the statistics used. gamma/beta may be NULL (affine off). */
int sei_bn_forward(const float *x, int64_t n, int32_t C, const float *gamma, const float *beta,
                   float *running_mean, float *running_var, float eps, float momentum, int32_t training,
                   float leak, const float *residual, float *y, float *save_mean, float *save_invstd,
                   sei_stream_t stream);
/* Fused scn.SubmanifoldConvolution / scn.Convolution -> scn.BatchNormalization (+LeakyReLU, +AddTable residual)
 * forward (Block.forward / ConvolutionDownsample.forward, sparse_building_blocks.py:47-57,130-139): same results as
 * sei_conv_forward followed by sei_bn_forward, but in training mode the per-channel sums of the convolution output
 * are accumulated in the convolution kernel's epilogue (fixed order), so BatchNorm does not re-read it for its
 * statistics.  Yconv [n_out,nOut] (kept for the backward) and Ybn [n_out,nOut] are both written. */
int sei_conv_bn_forward(sei_rulebook_t *rb, const float *X, const float *W, const float *bias, float *Yconv,
                        int32_t nIn, int32_t nOut, const float *gamma, const float *beta, float *running_mean,
                        float *running_var, float eps, float momentum, int32_t training, float leak,
                        const float *residual, float *Ybn, float *save_mean, float *save_invstd, sei_stream_t stream);

/* Inference (eval-mode BatchNorm, no autograd): Y = LeakyReLU_leak(BN_running(conv(X)) (+ residual)) in ONE kernel -- the
 * normalisation, the residual add and the activation are applied while the convolution kernel writes its finished tile,
 * so the convolution output itself never reaches memory and no statistics are computed (the reference's inference path:
 * bin/exec.py:206-224 -> trainer.validate -> model.eval()).  Same expression as the two-pass eval path of
 * sei_conv_bn_forward(training = 0): bit-identical results.  gamma / beta / bias / residual may be NULL. */
int sei_conv_bn_inference(sei_rulebook_t *rb, const float *X, const float *W, const float *bias, int32_t nIn, int32_t nOut,
                          const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                          float eps, float leak, const float *residual, float *Y, sei_stream_t stream);
/* backward of the above (training statistics): needs x, y (for the activation sign), dy.
 * dx [n,C]; dresidual [n,C] or NULL; dgamma/dbeta [C] or NULL. */
int sei_bn_backward(const float *x, const float *y, const float *dy, int64_t n, int32_t C,
                    const float *gamma, const float *save_mean, const float *save_invstd, float leak,
                    int32_t training, float *dx, float *dresidual, float *dgamma, float *dbeta,
                    sei_stream_t stream);

/* Same, and dx_colsum[C] (non-NULL) receives the column sums of dx computed while dx is written: when x is the output
 * of a convolution with a bias (Block.forward, sparse_building_blocks.py:47-57) that is the bias gradient, so
 * sei_conv_backward_weight can be called with dbias = NULL (no extra read of dx).
 * y may be NULL when the forward had no residual (dresidual must be NULL too): the LeakyReLU mask is then recomputed
 * from x, gamma and beta exactly as the forward computed it -- one tensor read less in each of the two passes. */
int sei_bn_backward_colsum(const float *x, const float *y, const float *dy, int64_t n, int32_t C,
                           const float *gamma, const float *beta, const float *save_mean, const float *save_invstd,
                           float leak, int32_t training, float *dx, float *dresidual, float *dgamma, float *dbeta,
                           float *dx_colsum, sei_stream_t stream);

/* ---- elementwise (scn.LeakyReLU / scn.ReLU / scn.AddTable) -------------------------------- */
int sei_leaky_forward(const float *x, float *y, int64_t count, float leak, sei_stream_t stream);
int sei_leaky_backward(const float *y, const float *dy, float *dx, int64_t count, float leak, sei_stream_t stream);
int sei_add_forward(const float *a, const float *b, float *out, int64_t count, sei_stream_t stream);
/* y = leaky(a + b) fused AddTable + LeakyReLU (sparse_building_blocks.py:96-98) */
int sei_add_leaky_forward(const float *a, const float *b, float *y, int64_t count, float leak, sei_stream_t stream);

/* ---- scn.SparseToDense (resnet.py:124-125,156; SCN `SparseToDense_updateOutput`) ---------- */
/* dense [B,C,D0,D1,D2] (zero-filled by this call), rows scattered by coordinate. */
int sei_sparse_to_dense_forward(sei_metadata_t *m, const int32_t spatial[3], const float *feats, int32_t C,
                                float *dense, sei_stream_t stream);
int sei_sparse_to_dense_backward(sei_metadata_t *m, const int32_t spatial[3], const float *ddense, int32_t C,
                                 float *dfeats, sei_stream_t stream);

/* ---- fused sparse head (opt-in): AvgPool3d_full(tanh(SparseToDense(x))) without the dense
 *      tensor (resnet.py:156-159 + classification_head.py:19-21): pooled[B,C] = sum_active tanh(x) / volume.
 *      Fixed-order two-stage reduction (no float atomics): bit-identical reruns. */
int sei_pool_tanh_forward(sei_metadata_t *m, const int32_t spatial[3], const float *feats, int32_t C, float *pooled,
                          sei_stream_t stream);
int sei_pool_tanh_backward(sei_metadata_t *m, const int32_t spatial[3], const float *feats, const float *dpooled,
                           int32_t C, float *dfeats, sei_stream_t stream);

/* ---- measurement hooks (bench.py roofline leg): HIP events around each conv launch -------- */
/* kinds: 0 conv_fwd, 1 conv_bwd_input, 2 conv_bwd_weight, 3 rulebook_build, 4 batchnorm, 5 other */
int sei_prof_enable(int32_t on);
int sei_prof_reset(void);
/* Synchronises outstanding events.  out = {launches, total_ms, algorithmic flops, algorithmic bytes} */
int sei_prof_query(int32_t kind, double out[4]);

/* number of stream synchronisations (count read-backs) the library has performed since the last reset: one for the
 * InputLayer, one per new strided level, one per lazily built tiled rule variant -- or ONE for all the variants that
 * sei_metadata_prebuild_like builds */
int sei_debug_sync_count(int64_t *count, int32_t reset);

/* library-owned device memory (rulebooks, scratch): bytes currently held by the pool */
int sei_pool_stats(int64_t *bytes_held, int64_t *bytes_in_use);
int sei_pool_trim(void);

#ifdef __cplusplus
}
#endif
#endif
