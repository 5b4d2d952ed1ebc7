/*
 * amc3d.h -- C-ABI of libamc3d_hip.so, the MI355X (gfx950) implementation of the
 * native point operators on AMContrast3D's training hot path.
 *
 * Drop-in boundary.  The reference binds its native layer through two pybind11
 * modules whose entry points take torch tensors and forward raw data pointers:
 *     pointnet2_batch_cuda  openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24
 *     pointops_cuda         openpoints/cpp/pointops/src/pointops_api.cpp:13-25
 * Each function below replaces the launcher behind one of those entry points
 * and keeps its argument order; tensors become plain device pointers and every
 * call gains a trailing `stream` (a hipStream_t passed as void*; NULL = the
 * null stream).  No torch type appears in any signature.  INTEGRATION.md shows
 * the ctypes stub that rebinds the reference's Python wrappers onto this ABI.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - float = IEEE fp32, indices = int32, layouts exactly as in the reference
 *     (dense row-major, batch first);
 *   - return value: 0 on success, otherwise the hipError_t of the failed
 *     launch/argument check (the reference prints and exit(-1)s instead:
 *     ball_query_gpu.cu:68-71); nothing is thrown, nothing is allocated, and
 *     no call synchronises the device -- they are graph-capturable;
 *   - functions that need scratch memory take a caller-owned `workspace`
 *     whose size is reported by the matching *_workspace_bytes() query.
 *   - distances are evaluated exactly as the reference writes them, in fp32,
 *     left to right and without FMA contraction: ((dx*dx + dy*dy) + dz*dz).
 */
#ifndef AMC3D_H
#define AMC3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* library identification: "amc3d-hip gfx950 <abi-version>" */
const char *amc3d_version(void);
/* text of the last error on this thread ("" if none) */
const char *amc3d_last_error(void);

/* A HIP stream (hipStream_t) with a hardware queue of its own, for the long latency-bound furthest-point-sampling
 * launches of a pipelined training loop: ordinary streams share four hardware queues per process, and whatever
 * shares the queue of a running FPS kernel waits milliseconds for it.  (The reference launches everything on the
 * legacy default stream; this is the hook its trainer would use to overlap batches.) */
int amc3d_stream_create_dedicated(void **stream);
/* the same with a CU mask that enables bits [first_cu, first_cu + n_cus) only (n_cus <= 0: every CU) */
int amc3d_stream_create_masked(void **stream, int first_cu, int n_cus);
/* the same with an arbitrary mask (word i, bit j enables CU 32 i + j), and a diagnostic for planning such masks:
 * out[b] = the XCD (0-7) workgroup b of an nblocks-wide launch on `stream` ran on */
int amc3d_stream_create_cu_mask(void **stream, const unsigned int *mask, int words);
int amc3d_probe_xcc_ids(int nblocks, int *out, void *stream);
/* Raise the device's scratch (private segment) high-water mark to bytes_per_lane (256 / 1024 / 4096 / 16384) with one launch, BEFORE
 * graphs are captured: the runtime re-allocates scratch when a kernel asks for more than any before it, and graph nodes
 * instantiated earlier keep the old allocation (a replay then faults).  scratch_out: any device int (not written). */
int amc3d_reserve_scratch(int bytes_per_lane, int *scratch_out, void *stream);
int amc3d_stream_destroy(void *stream);

/* ---- pointnet2_batch surface ------------------------------------------------ */

/* replaces ball_query_wrapper_fast -> ball_query_kernel_launcher_fast
 * (pointnet2_batch/src/ball_query.cpp:29-39, ball_query_gpu.cu:54-73).
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample): the first `nsample`
 * support indices (ascending) with d2 < radius*radius, padded with the first
 * hit; a row with no hit is written as zeros (the reference relies on the
 * caller's zero-fill, group.py:194).
 * With a workspace of amc3d_grid_search_workspace_bytes(b, n, m) the search runs on a uniform grid of cell
 * edge >= radius (27 cells per query instead of all n points); NULL / too small = the all-pairs scan.
 * Both give the reference's result exactly. */
size_t amc3d_grid_search_workspace_bytes(int b, int n_support, int m_queries);
int amc3d_ball_query(int b, int n, int m, float radius, int nsample,
                     const float *new_xyz, const float *xyz, int *idx,
                     void *workspace, size_t workspace_bytes, void *stream);

/* replaces group_points_wrapper_fast (group_points.cpp, group_points_gpu.cu:53-92):
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample) */
int amc3d_group_points(int b, int c, int n, int npoints, int nsample,
                       const float *points, const int *idx, float *out, void *stream);

/* bytes of the optional scratch of the two scatter-add gradients below (one (b,n,c) fp32 image) */
size_t amc3d_scatter_workspace_bytes(int b, int c, int n);

/* replaces group_points_grad_wrapper_fast (group_points_gpu.cu:14-50):
 * grad_points (b,c,n) += scatter(grad_out (b,c,npoints,nsample)); the caller
 * zero-initialises grad_points (group.py:111).  With a workspace of
 * amc3d_scatter_workspace_bytes(b,c,n) the adds go through a point-major image (full-rate atomic
 * shape) and are transposed back; workspace = NULL selects the reference's direct per-element
 * atomics. */
int amc3d_group_points_grad(int b, int c, int n, int npoints, int nsample,
                            const float *grad_out, const int *idx, float *grad_points,
                            void *workspace, size_t workspace_bytes, void *stream);

/* replaces gather_points_wrapper_fast / gather_points_grad_wrapper_fast
 * (sampling_gpu.cu:15-90): points (b,c,n), idx (b,npoints) -> out (b,c,npoints) */
int amc3d_gather_points(int b, int c, int n, int npoints,
                        const float *points, const int *idx, float *out, void *stream);
int amc3d_gather_points_grad(int b, int c, int n, int npoints,
                             const float *grad_out, const int *idx, float *grad_points, void *stream);

/* replaces furthest_point_sampling_wrapper (sampling.cpp, sampling_gpu.cu:100-260):
 * dataset (b,n,3) -> idxs (b,m), idxs[:,0] = 0.  `temp` (b,n) is the reference's
 * running-min-distance buffer (the caller fills it with 1e10, subsample.py:95); it may be
 * NULL for n <= 24576: the running minima then start at 1e10 and stay on chip.  When given it
 * is read as the initial minima and receives the final ones, as in the reference.
 * `workspace`: amc3d_fps_workspace_bytes(b,n) bytes (the cloud's spatial sort order).
 * Ties resolve exactly as the reference's block-strided scan + shared-memory tree does for
 * its block size opt_n_threads(n) (cuda_utils.h:10-14). */
size_t amc3d_fps_workspace_bytes(int b, int n);
int amc3d_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                  float *temp, int *idxs, void *workspace, size_t workspace_bytes,
                                  void *stream);

/* replaces three_nn_wrapper_fast (interpolate_gpu.cu:16-81):
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) squared distances, idx (b,n,3) */
/* workspace: amc3d_grid_search_workspace_bytes(b, m, n) (support = known) selects the grid search; NULL = all pairs */
int amc3d_three_nn(int b, int n, int m, const float *unknown, const float *known,
                   float *dist2, int *idx, void *workspace, size_t workspace_bytes, void *stream);

/* replaces three_interpolate_wrapper_fast (interpolate_gpu.cu:84-124):
 * points (b,c,m), idx/weight (b,n,3) -> out (b,c,n) */
int amc3d_three_interpolate(int b, int c, int m, int n, const float *points,
                            const int *idx, const float *weight, float *out, void *stream);

/* out (b,c,n) = base (b,c,n) + three_interpolate(points (b,c,m), idx, weight): FeaturePropogation's first conv
 * (pointnext_AA.py:210-226) applied before the interpolation, W . [f1 ; up(f2)] = W1 . f1 + up(W2 . f2) */
int amc3d_three_interpolate_add(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                const float *base, float *out, void *stream);
/* replaces three_interpolate_grad_wrapper_fast (interpolate_gpu.cu:127-169):
 * grad_points (b,c,m) += ...; caller zero-initialises (upsampling.py:82).
 * Optional workspace: amc3d_scatter_workspace_bytes(b,c,m), as above. */
int amc3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                 const int *idx, const float *weight, float *grad_points,
                                 void *workspace, size_t workspace_bytes, void *stream);

/* ---- pointops surface --------------------------------------------------------- */

/* replaces knnquery_cuda -> knnquery_cuda_launcher
 * (pointops/src/knnquery/knnquery_cuda.cpp:7-16, knnquery_cuda_kernel.cu:65-108).
 * xyz (n,3) support, new_xyz (m,3) queries, offset / new_offset: cumulative
 * segment ends (nbatch entries each; the reference does not pass nbatch, it
 * walks new_offset until it exceeds the query id -- pass it here).
 * -> idx (m,nsample), dist2 (m,nsample) squared distances, ascending.
 * nsample <= 100 (the reference's per-thread heap is float[100]).
 * Results equal the reference's max-heap + heap-sort including how equal
 * distances are ordered.
 * reuse_grid != 0: the workspace was last used by amc3d_knnquery with the SAME support set (xyz, offset, n,
 * nbatch) and at least as many queries; its cell grid is kept and only the queries run (the loss asks for
 * four neighbour sets of the full-resolution cloud per step).  Results are identical either way. */
size_t amc3d_knnquery_workspace_bytes(int n, int m, int nsample, int nbatch);
/* 1: this problem size goes through the cell grid (so a later call may pass reuse_grid, and this call may reuse an
 * earlier grid); 0: the all-pairs heap kernel answers and the workspace is left untouched */
int amc3d_knnquery_uses_grid(int m, int nsample, int n, int nbatch);
int amc3d_knnquery(int m, int nsample, int n, int nbatch, const float *xyz, const float *new_xyz,
                   const int *offset, const int *new_offset, int *idx, float *dist2,
                   void *workspace, size_t workspace_bytes, int reuse_grid, void *stream);

/* ---- adaptive-margin contrastive loss -------------------------------------------
 * The reference has no native code here: it evaluates this part with torch ops and a Python
 * loop.  These entry points replace, per decoder stage,
 *     openpoints/AMContrast3D/AEF/utils.py:29-41            (amc3d_vote_labels)
 *     openpoints/AMContrast3D/MarginContrast.py:111-115,228-230  (amc3d_posmask)
 *     openpoints/AMContrast3D/AEF/ambiguity.py:11-71 + AEF/function.py:10-39  (amc3d_ambiguity)
 *     openpoints/AMContrast3D/MarginContrast.py:77-79,117-174,250-257  (amc3d_contrast_*)
 * `nbr` is the int32 k-NN index of the stage with `nbr_stride` columns per row; pass the
 * address of the first column to USE (the reference drops column 0, the self match:
 * MarginContrast.py:225-226) and k = number of columns used. */

/* labels[q] = majority class among labels0[nbr_idx[q, 0..kr)] (lowest class id on equal
 * counts): arg-max of the mean one-hot label.  nbr_idx is dense (m,kr), kr <= 128. */
int amc3d_vote_labels(int m, int kr, int num_classes, const int *labels0, const int *nbr_idx, int *labels,
                      void *stream);

/* posmask[i,j] = labels[nbr[i,j]] == labels[i]  -> (m,k) bytes (torch.bool layout) */
int amc3d_posmask(int m, int k, int nbr_stride, const int *labels, const int *nbr, unsigned char *posmask,
                  void *stream);

/* a[i] = ambiguity of point i from its positive mask and neighbour coordinates.
 * mode 1/2/3 = cctype Method1/2/3; beta = ccbeta.  Workspace: amc3d_ambiguity_workspace_bytes(m). */
size_t amc3d_ambiguity_workspace_bytes(int m);
int amc3d_ambiguity(int m, int k, int nbr_stride, int mode, float beta, const float *p,
                    const unsigned char *posmask, const int *nbr, float *a, void *workspace,
                    size_t workspace_bytes, void *stream);

/* The anchors that enter the stage loss, 0 < a <= 1 (MarginContrast.py:250-252: the boolean-mask indexing
 * features[mask], neighbor_feature[mask], ...), as a compact ascending list built once per stage plan:
 * sel[0] = count, sel[1..count] = anchor ids; sel holds amc3d_select_anchors_ints(m) ints (the tail is scratch). */
size_t amc3d_select_anchors_ints(int m);
int amc3d_select_anchors(int m, const float *a, int *sel, size_t sel_ints, void *stream);

/* Stage loss = mean over anchors with 0 < a <= 1 of
 *   -log( sum_+ e^{(s-m_i)/T} / (sum_+ e^{(s-m_i)/T} + sum_- e^{s/T}) + 1e-12 ),  m_i = mu*a_i + nu,
 * s = cosine similarity of the (m,C) embeddings f.  Outputs: norm (m) clamped row norms, sim (m,k),
 * loss_pt (m), mean_cnt[2] = {stage loss, number of anchors}; all kept for the backward.
 * sel: the list of amc3d_select_anchors for this a (sim / loss_pt are then written for the listed anchors
 * only), or NULL to visit and test every anchor.
 * unit (m,C) or NULL: receives the unit rows f_i / norm_i; with it (and C in {16, 32, 64, 128, 256}, 16-byte aligned rows) the
 * row-gather kernels run on those rows -- no division per fetched neighbour element -- and amc3d_contrast_backward_mutual
 * takes the same buffer; NULL: the generic kernel on f and norm. */
int amc3d_contrast_forward(int m, int C, int k, int nbr_stride, const float *f, const int *nbr,
                           const unsigned char *posmask, const float *a, const int *sel, float mu, float nu,
                           float temperature, float *norm, float *unit, float *sim, float *stats, float *loss_pt,
                           float *mean_cnt, void *stream);
/* stats (m,2) or NULL: with the unit-row kernels the pass leaves, per visited anchor, the two sums of exponentials its loss is made
 * of -- amc3d_contrast_backward_mutual builds its per-anchor records from them instead of re-reading sim, and sim may then be NULL
 * (the cosines are not stored: the mutual-edge backward recomputes the ones it needs from the unit rows it fetches anyway). */

/* The same on CHANNEL-major embeddings f_cm (b, C, n) -- the decoder's layout; pointnext_AA.py:518-519 makes the point-major
 * copy with flatten(transpose) -- m = b * n anchors in cloud-major order.  C in {16, 32, 64, 128, 256}; unit (b*n, C) is required
 * and receives the point-major unit rows (what amc3d_contrast_backward_mutual reads); no point-major copy of f is made. */
int amc3d_contrast_forward_cm(int b, int C, int n, int k, int nbr_stride, const float *f_cm, const int *nbr,
                              const unsigned char *posmask, const float *a, const int *sel, float mu, float nu,
                              float temperature, float *norm, float *unit, float *sim, float *stats, float *loss_pt,
                              float *mean_cnt, void *stream);

/* grad_f (m,C) += grad_out[0] * d(stage loss)/d f; the caller zero-initialises grad_f.
 * grad_out is a DEVICE scalar (no host sync).  C <= 512.  sel as in the forward. */
int amc3d_contrast_backward(int m, int C, int k, int nbr_stride, const float *f, const float *norm,
                            const int *nbr, const unsigned char *posmask, const float *a, const int *sel, float mu,
                            float nu, float temperature, const float *sim, const float *mean_cnt,
                            const float *grad_out, float *grad_f, void *stream);

/* The same gradient as a gather over reverse lists instead of float atomics (group_points_grad-style scatter,
 * MarginContrast.py:250-257 through autograd's index_put backward): fixed summation order, every row of grad_f WRITTEN
 * once (no zero-initialisation).  rev = [rev_start (m+1) | rev_edge (m*k)] int32 from amc3d_contrast_csr: the positions
 * i*k + j, ascending, of the selected anchors i whose neighbour j is row n; built with the stage's plan (coordinates and
 * labels only).  gco: m*k floats of scratch (dL/ds per edge).  C in {16, 32, 64, 128, 256}. */
size_t amc3d_contrast_csr_workspace_bytes(int m);
int amc3d_contrast_csr(int m, int k, int nbr_stride, const int *nbr, const int *sel, int *rev, void *workspace,
                       size_t workspace_bytes, void *stream);
int amc3d_contrast_backward_csr_supported(int C);
int amc3d_contrast_backward_csr(int m, int C, int k, int nbr_stride, const float *f, const float *norm, const int *nbr,
                                const unsigned char *posmask, const float *a, const int *sel, const int *rev, float mu,
                                float nu, float temperature, const float *sim, const float *mean_cnt,
                                const float *grad_out, float *gco, float *grad_f, void *stream);

/* The default since round 3: the gradient over the MUTUAL edges.  In the stages' k-NN graphs ~90 % of the edges are mutual
 * (x in N(n) and n in N(x)); cosine and positive mask are symmetric in an edge's two ends, so row n, walking its own list
 * once, computes both directions of every mutual edge from one fetch of f[x] and a 32-byte record of x; only the non-mutual
 * edges of the selected anchors need reverse lists.  Every row of grad_f is WRITTEN once (no zero-initialisation), fixed
 * summation order, no float atomics.
 * amc3d_contrast_mutual: mutual (m*k) bytes, mutual[i*k+s] = how often i is in the list of nbr[i][s] (0 / 1 in a k-NN graph;
 *   counted at the first slot naming that neighbour); rev = [rev_start (m+1) |
 *   rev_edge (m*k)] int32: per row n the positions i*k+s, ascending, of the NON-mutual edges of the anchors with
 *   0 < a[i] <= 1 that point at n (those edges also carry bit 0x80 in mutual[]; the count is in the low 7 bits).  k <= 64.  Coordinates and labels only: part of the stage's plan.  Workspace:
 *   amc3d_contrast_csr_workspace_bytes(m).
 * amc3d_contrast_backward_mutual: workspace amc3d_contrast_backward_mutual_workspace_bytes(m) (the per-anchor records);
 *   unit, norm, sim, mean_cnt as amc3d_contrast_forward wrote them.  C in {16, 32, 64, 128, 256}. */
size_t amc3d_contrast_mutual_workspace_bytes(int m);
int amc3d_contrast_mutual(int m, int k, int nbr_stride, const int *nbr, const float *dist2, const float *a,
                          unsigned char *mutual, int *rev, void *workspace, size_t workspace_bytes, void *stream);
size_t amc3d_contrast_backward_mutual_workspace_bytes(int m);
int amc3d_contrast_backward_mutual(int m, int C, int k, int nbr_stride, const float *unit, const float *norm, const int *nbr,
                                   const unsigned char *posmask, const float *a, const unsigned char *mutual, const int *rev,
                                   float mu, float nu, float temperature, const float *sim, const float *stats /* or NULL */,
                                   const float *mean_cnt, const float *grad_out, void *workspace, size_t workspace_bytes,
                                   float *grad_f, void *stream);

/* ---- grouped 1x1 convolution fused with its gather (fp32 MFMA) ------------------------------------
 * Replaces, for the first layer of a SetAbstraction / LocalAggregation MLP, the chain
 *   grouping_operation(features, idx) -> torch.cat([dp, fj], 1) -> nn.Conv2d 1x1 (bias-free)
 * (openpoints/models/layers/group.py:244-255,323-325; backbone/pointnext_AA.py:57-63,164-166) and its
 * backward (conv backward, slice, group_points_grad) without materialising the (b,cin+3,npoints,nsample)
 * input.  f_pm is the POINT-major (b,n,cin) copy of the features (amc3d_transpose_cn), dp the
 * (b,3,npoints,nsample) relative positions, weight the (cout, cin+3) conv weight with the reference's
 * channel order [dp, features].  Supported: cin <= 64, cout in {32,64,96,128}. */
int amc3d_grouped_conv_supported(int cin, int cout);
int amc3d_transpose_cn(int b, int c, int n, const float *src, float *dst, void *stream);
int amc3d_grouped_conv_forward(int b, int cin, int cout, int n, int npoints, int nsample, const float *f_pm,
                               const float *dp, const int *idx, const float *weight, float *y, void *stream);
size_t amc3d_grouped_conv_workspace_bytes(int b, int cin, int cout, int npoints, int nsample);
/* df_pm (b,n,cin) += scatter(W[:,3:]^T dy) (zero-initialised by the caller; NULL to skip),
 * dweight (cout,cin+3) = dy . X^T (NULL to skip) */
int amc3d_grouped_conv_backward(int b, int cin, int cout, int n, int npoints, int nsample, const float *f_pm,
                                const float *dp, const int *idx, const float *weight, const float *dy,
                                float *df_pm, float *dweight, void *workspace, size_t workspace_bytes, void *stream);

/* Confusion matrix of the training predictions (examples/segmentation/main_AA.py:414-415, utils/metrics.py:50-75:
 * cm.update(logits.argmax(dim=1), target) every iteration): cm (v*v) int64 += histogram of (target, arg-max over the class
 * planes of logits (B,C,N); first maximum as torch.argmax), v = C + has_ignore <= 64 (points whose target == ignore count in
 * the extra row and column); invalid (1) int64 += points whose target lies outside [0, v).  One launch. */
int amc3d_confusion_update(int B, int C, long N, const float *logits, const long long *target, long long ignore,
                           int has_ignore, long long *cm, long long *invalid, void *stream);

/* ---- cross entropy over channel-major logits -----------------------------------------------------
 * nn.CrossEntropyLoss() with its defaults (mean over targets != ignore_index) as CrossEntropyAce applies it
 * (openpoints/loss/build.py:328,338-340), without the (B*N, C) transposed copy of the logits.
 * logits (B,C,N) fp32, target (B,N) int64, lse (B,N) out (saved for backward), mean_cnt[2] = {loss, count}. */
size_t amc3d_cross_entropy_workspace_bytes(int B, long N);
int amc3d_cross_entropy_forward(int B, int C, long N, const float *logits, const long long *target,
                                long long ignore_index, float *lse, float *mean_cnt, void *workspace,
                                size_t workspace_bytes, void *stream);
/* dlogits (B,C,N) = grad_out[0] * (softmax - onehot) / count, zero rows for ignored targets */
int amc3d_cross_entropy_backward(int B, int C, long N, const float *logits, const long long *target,
                                 long long ignore_index, const float *lse, const float *mean_cnt,
                                 const float *grad_out, float *dlogits, void *stream);

/* ---- pointwise (1x1) convolution on fp32 MFMA -----------------------------------------------------
 * Replaces the nn.Conv1d / nn.Conv2d (kernel size 1) layers of the path, which the reference builds in
 * openpoints/models/layers/conv.py:8-21 and runs through cuDNN.  Channel-major tensors: x (b,cin,P),
 * y (b,cout,P), weight (cout,cin), bias (cout) or NULL; P = points (Conv1d) or points*neighbours (Conv2d). */
int amc3d_pointwise_conv_forward(int b, int cin, int cout, long P, const float *x, const float *weight,
                                 const float *bias, float *y, void *stream);
/* forward with scratch: the short deep layers (>= 64 channels on both sides, a few hundred positions per cloud) split their K
 * axis over workgroups and sum the partial products in a fixed order; amc3d_pointwise_conv_forward_workspace_bytes() is 0
 * for every other shape (the call is then amc3d_pointwise_conv_forward) */
size_t amc3d_pointwise_conv_forward_workspace_bytes(int b, int cin, int cout, long P, int has_bias);
int amc3d_pointwise_conv_forward_ws(int b, int cin, int cout, long P, const float *x, const float *weight,
                                    const float *bias, float *y, void *workspace, size_t workspace_bytes, void *stream);
size_t amc3d_pointwise_conv_workspace_bytes(int b, int cin, int cout, long P);
/* dx (b,cin,P) = weight^T . dy (NULL to skip; weight may then be NULL too); dweight (cout,cin) = sum_{b,p} dy x^T (NULL to skip; needs
 * x and the workspace; summed in a fixed order -> deterministic) */
int amc3d_pointwise_conv_backward(int b, int cin, int cout, long P, const float *x, const float *weight,
                                  const float *dy, float *dx, float *dweight, void *workspace,
                                  size_t workspace_bytes, void *stream);

/* a weight matrix cut into two column blocks / two gradient blocks joined (the [W_dp | W_f] weight of a neighbourhood layer,
 * the [W_skip | W_up] weight of a FeaturePropagation conv): w (rows, c1 + c2) <-> a (rows, c1), b (rows, c2), one launch */
int amc3d_split_columns(int rows, int c1, int c2, const float *w, float *a, float *b, void *stream);
int amc3d_join_columns(int rows, int c1, int c2, const float *a, const float *b, float *w, void *stream);
/* dbias (c) = sum_{b,p} dy (b,c,P): bias gradient of a 1x1 convolution with bias (the stem conv and the head's last conv,
 * base_seg.py:236-252, pointnext_AA.py:104-127 with is_head), summed in a fixed order */
int amc3d_bias_grad(int b, int c, long P, const float *dy, float *dbias, void *stream);

/* ---- residual branch of a strided SetAbstraction block (openpoints/models/backbone/pointnext_AA.py:157-168, use_res):
 *     fi = torch.gather(f, -1, idx...); identity = self.skipconv(fi); ...; f = self.act(f + identity)
 * forward:  out (b,cout,m) = relu(y + weight . f[:, :, fps_idx] + bias), f (b,cin,n), fps_idx (b,m) int32 (the FPS picks;
 *           repeats allowed: their gradients sum, as torch.gather's backward does), weight (cout,cin), bias (cout) or NULL, y (b,cout,m) = the pooled main branch;
 *           fi (b,cin,m) or NULL = the gathered columns, kept for the weight gradient.
 * backward: g (b,cout,m) = dout * (out > 0) -- the gradient w.r.t. y AND w.r.t. the skip conv's output;
 *           df (b,cin,n) or NULL = weight^T . g at the sampled columns, zero elsewhere (written whole: no pre-zeroing);
 *           dweight (cout,cin) or NULL (needs fi; fixed-order partial sums); dbias (cout) or NULL (fixed order). */
int amc3d_sa_residual_forward(int b, int cin, int cout, int n, int m, const float *f, const int *fps_idx,
                              const float *weight, const float *bias, const float *y, float *out, float *fi, void *stream);
size_t amc3d_sa_residual_workspace_bytes(int b, int cin, int cout, int m);
int amc3d_sa_residual_backward(int b, int cin, int cout, int n, int m, const float *dout, const float *out,
                               const float *fi, const int *fps_idx, const int *dup_flag, const float *weight, float *g, float *df,
                               float *dweight, float *dbias, void *workspace, size_t workspace_bytes, void *stream);
/* dup_flag: DEVICE int of amc3d_index_duplicates for these picks, or NULL.  torch.gather's backward (pointnext_AA.py:157) SUMS
 * over repeated picks -- FPS re-picks a point when a cloud holds fewer distinct points than picks (crop_pc pads small rooms by
 * repetition, data_util.py:161-167) -- so the scatter of df adds (float atomics) when the flag is 1 or unknown, and stores plainly
 * when the plan found the picks distinct. */
size_t amc3d_index_duplicates_workspace_bytes(int b, int n);
int amc3d_index_duplicates(int b, int n, int m, const int *idx, int *flag, void *workspace, size_t workspace_bytes, void *stream);

/* ---- input pipeline on the device (openpoints/dataset/data_util.py:92-174; dataset/s3dis/s3dis.py:122-144) -----------
 * voxelize: floor(coord / voxel_size) in float64 -> FNV-1a 64-bit hash of the three cell coordinates (fnv_hash_vec) ->
 * stable sort by key -> voxel ids / starts / counts.  coord (n,3) fp32 must be shifted to its min corner (crop_pc does
 * that first).  numpy's argsort is not stable, so the reference leaves the order of the points inside one voxel
 * unspecified; idx_sort here is the stable order.  key (n) = hash per point; idx_sort (n); voxel_idx (n) = voxel id of
 * sorted position i (np.unique's inverse); start (n+1); count (n, zero beyond nvox); nvox (1) -- all device memory. */
size_t amc3d_voxelize_workspace_bytes(int n);
int amc3d_voxelize(int n, const float *coord, double voxel_size, unsigned long long *key, int *idx_sort, int *voxel_idx,
                   int *start, int *count, int *nvox, void *workspace, size_t workspace_bytes, void *stream);
/* train mode (data_util.py:136-140): idx_unique[v] = idx_sort[start[v] + rnd[v] % count[v]], rnd = the caller's
 * np.random.randint(0, count.max(), nvox) draw */
int amc3d_voxel_select(int nvox, const int *start, const int *count, const int *idx_sort, const int *rnd, int *idx_unique,
                       void *stream);
/* crop_pc's crop (data_util.py:157-160): d2 (n) fp32 = squared distances to coord[init_idx] (((dx^2+dy^2)+dz^2), no
 * contraction), crop_idx (keep) = indices of the keep nearest points in ascending distance */
size_t amc3d_crop_nearest_workspace_bytes(int n);
int amc3d_crop_nearest(int n, const float *coord, int init_idx, int keep, float *d2, int *crop_idx, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ---- neighbourhood aggregation with one grouped conv: "convolve first, gather after" ---------------------------------
 * LocalAggregation.forward / single-layer SetAbstraction.forward (openpoints/models/backbone/pointnext_AA.py:57-63,
 * 139-170; layers/group.py:244-255, 323-325): grouping_operation -> cat([dp, fj]) -> Conv2d 1x1 -> BatchNorm2d (batch
 * statistics) [-> ReLU] -> max over the nsample neighbours.  W . [dp ; f[idx]] = (W_f . f)[idx] + W_dp . dp, so the conv
 * runs on the n source points (amc3d_pointwise_conv_forward), BatchNorm's statistics follow from n-sized sums and the
 * geometry moments below, and the only pass over npoints * nsample positions is the gather + max itself (csrc/lagg.hip).
 * Supported: cout in {8,16,32,64,128} or a multiple of 128; nsample <= 64. */
int amc3d_local_aggregation_supported(int cout, int nsample);
/* geometry moments of a neighbourhood query (coordinates only; part of the geometry plan): idx (b,npoints,nsample) into
 * the n support points, dp (b,3,npoints,nsample) -> opaque buffer holding each support point's in-degree and the sum of
 * the dp that reference it (fixed point: independent of the order of the atomics) and the global sums of dp and dp dp^T */
size_t amc3d_group_moments_bytes(int b, int n);
int amc3d_group_moments(int b, int n, int npoints, int nsample, const int *idx, const float *dp, void *moments,
                        size_t moments_bytes, void *stream);
size_t amc3d_local_aggregation_workspace_bytes(int b, int cout, int n, int npoints);
/* g_cm (b,cout,n) = W_f . f from amc3d_pointwise_conv_forward; w_dp (cout,3) = the conv weight's dp columns.
 * Outputs: pooled (b,cout,npoints), arg (b,cout,npoints) bytes = torch.max's indices, ystar = the pre-BatchNorm value at
 * arg, g_pm (b,n,cout) point-major copy of g_cm, gd (cout,3) doubles (all kept for backward), mean / invstd /
 * var_unbiased (cout) + nn.BatchNorm's running update when running_mean != NULL (momentum < 0: left to the caller).
 * training == 0 (eval): mean / invstd are inputs (running statistics); moments, gd, var_unbiased, workspace may be NULL */
int amc3d_local_aggregation_forward(int b, int cout, int n, int npoints, int nsample, int training, int relu, float eps,
                                    float momentum, const float *g_cm, const int *idx, const float *dp, const float *w_dp,
                                    const void *moments, const float *gamma, const float *beta, float *g_pm, float *pooled,
                                    unsigned char *arg, float *ystar, float *mean, float *invstd, float *var_unbiased,
                                    double *gd, float *running_mean, float *running_var, long long *num_batches_tracked,
                                    int phase, double *sums, void *workspace, size_t workspace_bytes, void *stream);
/* Statistics over several ranks (the reference converts every BatchNorm to SyncBatchNorm when world_size > 1,
 * main_AA.py:146-148): phase 0 = one rank, everything in one call.  phase 1 writes this rank's {sum y, sum y^2} per channel
 * and its position count to sums (2*cout + 1 doubles) and returns; the caller all-reduces sums; phase 2 finishes from the
 * reduced sums (same arguments, same workspace).  Backward likewise: phase 1 -> dsums (2*cout doubles: sum dq, sum dq xhat),
 * all-reduce, phase 2 with count = sums + 2*cout of the forward call; parameter gradients stay rank-local. */
/* dg_cm (b,cout,n) = gradient w.r.t. g_cm (feed it to amc3d_pointwise_conv_backward for df and dW_f); dw_dp (cout,3),
 * dgamma, dbeta (cout).  The pooled gradient is scattered with cout * npoints float atomics (not x nsample). */
int amc3d_local_aggregation_backward(int b, int cout, int n, int npoints, int nsample, int relu, const float *dpooled,
                                     const float *ystar, const unsigned char *arg, const float *g_pm, const int *idx,
                                     const float *dp, const float *w_dp, const void *moments, const double *gd,
                                     const float *mean, const float *invstd, const float *gamma, const float *beta,
                                     float *dg_cm, float *dw_dp, float *dgamma, float *dbeta, int phase, double *dsums,
                                     const double *count, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the same three products in bf16 compute / fp32 accumulate (mixed precision: main_AA.py:389-394 wraps model and
 * criterion in autocast; BASELINE config 5).  Tensors stay fp32 in memory; operands are rounded to bf16 as they are
 * staged, multiplied on v_mfma_f32_32x32x16_bf16 and accumulated in fp32 (csrc/gemm_bf16.hip). */
int amc3d_pointwise_conv_forward_bf16(int b, int cin, int cout, long P, const float *x, const float *weight,
                                      const float *bias, float *y, void *stream);
size_t amc3d_pointwise_conv_workspace_bytes_bf16(int b, int cin, int cout, long P);
int amc3d_pointwise_conv_backward_bf16(int b, int cin, int cout, long P, const float *x, const float *weight,
                                       const float *dy, float *dx, float *dweight, void *workspace,
                                       size_t workspace_bytes, void *stream);

/* ---- first layer of a multi-layer SetAbstraction MLP (PointNeXt-S, sa_layers = 2; pointnext_AA.py:104-127, 164-166):
 * the same convolve-before-gather conv + BatchNorm [+ ReLU], with the activation x1 (b,cout,npoints,32) materialised for
 * the layers that follow.  forward: amc3d_pointwise_conv_forward -> g_cm, then this (statistics from the moments, one
 * gather + write pass).  backward: from dx1, ONE pass masks the ReLU, scatters into the source points (row-contiguous float
 * atomics) and accumulates BatchNorm's two sums and the dp weight gradient; dg_cm then goes to
 * amc3d_pointwise_conv_backward.  Supported: nsample == 32, cout in {8,16,32} or a multiple of 64.  Workspace:
 * amc3d_local_aggregation_workspace_bytes. */
int amc3d_grouped_conv_bn_supported(int cout, int nsample);
int amc3d_grouped_conv_bn_forward(int b, int cout, int n, int npoints, int nsample, int training, int relu, float eps,
                                  float momentum, const float *g_cm, const int *idx, const float *dp, const float *w_dp,
                                  const void *moments, const float *gamma, const float *beta, float *g_pm, float *x1,
                                  float *mean, float *invstd, float *var_unbiased, double *gd, float *running_mean,
                                  float *running_var, long long *num_batches_tracked, int phase, double *sums,
                                  void *workspace, size_t workspace_bytes, void *stream);
int amc3d_grouped_conv_bn_backward(int b, int cout, int n, int npoints, int nsample, int relu, const float *dx1,
                                   const float *g_pm, const int *idx, const float *dp, const float *w_dp,
                                   const void *moments, const double *gd, const float *mean, const float *invstd,
                                   const float *gamma, const float *beta, float *dg_cm, float *dw_dp, float *dgamma,
                                   float *dbeta, int phase, double *dsums, const double *count, void *workspace,
                                   size_t workspace_bytes, void *stream);

/* ---- reverse adjacency of a neighbourhood query: gathers instead of float atomics in backward (csrc/csr.hip) ---------
 * The reference's grouping backward scatters with atomicAdd (group_points_gpu.cu:34-51).  The edges depend on coordinates
 * only: amc3d_group_csr sorts them once by target (stable radix sort), after which every source point sums its incoming
 * positions in list order -- deterministic, no atomics.  rev_start (b*n + 1) int32, rev_edge (b*npoints*nsample) int32:
 * rev_edge[rev_start[b*n + j] .. rev_start[b*n + j + 1]) = positions p = m*nsample + k of batch b with idx[b,m,k] == j. */
size_t amc3d_group_csr_workspace_bytes(int b, int npoints, int nsample);
int amc3d_group_csr(int b, int n, int npoints, int nsample, const int *idx, int *rev_start, int *rev_edge, void *workspace,
                    size_t workspace_bytes, void *stream);
/* rev_dp (b*npoints*nsample, 4) fp32, 16-byte aligned: per edge of the lists, in list order, the relative position
 * dp[b, :, p] of its position p and p itself (its int32 bits in the fourth float): what the gather of
 * amc3d_grouped_conv_bn_backward_csr reads as one 16-byte stream instead of an edge id and three scattered floats.
 * Coordinates only: part of the plan. */
int amc3d_group_csr_dp(int b, int npoints, int nsample, const int *rev_edge, const float *dp, float *rev_dp, void *stream);
/* the moments buffer of amc3d_group_moments from the lists (no scattered atomics) */
int amc3d_group_moments_csr(int b, int n, int npoints, int nsample, const int *rev_start, const int *rev_edge,
                            const float *dp, void *moments, size_t moments_bytes, void *stream);
/* amc3d_grouped_conv_bn_backward as a gather over the lists.  dx1 (b,cout,npoints,nsample) is transposed to position-major
 * rows in the workspace first, or, with dx1_position_major, is already (b,npoints,nsample,cout) (amc3d_sa_tail_backward
 * writes it that way) and is read in place. */
size_t amc3d_grouped_conv_bn_csr_workspace_bytes(int b, int cout, int n, int npoints, int nsample);
int amc3d_grouped_conv_bn_backward_csr(int b, int cout, int n, int npoints, int nsample, int relu, const float *dx1,
                                       int dx1_position_major, const float *g_pm, const int *rev_start, const int *rev_edge,
                                       const float *rev_dp /* amc3d_group_csr_dp, or NULL */, const float *dp, const float *w_dp, const void *moments, const double *gd, const float *mean,
                                       const float *invstd, const float *gamma, const float *beta, float *dg_cm, float *dw_dp,
                                       float *dgamma, float *dbeta, int phase, double *dsums, const double *count,
                                       void *workspace, size_t workspace_bytes, void *stream);

/* ---- tail of a two-layer SetAbstraction block, recomputed instead of materialised -----------------------------
 * BN1 -> ReLU -> Conv2d 1x1 (C1 -> C2) -> BN2 [-> ReLU] -> max over the K = 32 neighbours
 * (openpoints/models/backbone/pointnext_AA.py:104-127, 164-166) from the first conv's raw output y1 (B,C1,M,32),
 * without ever writing the (B,C2,M,32) activation: every pass re-creates it tile by tile on the MFMA.
 * Supported: K == 32, even C1 <= 64, C2 <= 128, y1 16-byte aligned.  BN1's statistics (mean1, invstd1) come from
 * amc3d_bn_stats(y1); its backward (from dx1) is amc3d_bn_backward(x = y1, dy = dx1, relu = 1). */
int amc3d_sa_tail_supported(int C1, int C2, int K);
int amc3d_sa_tail_pays(int C1, int C2);  /* 1 where the fused tail is faster than the layer-by-layer kernels (HBM-bound widths) */
size_t amc3d_sa_tail_workspace_bytes(int B, int C1, int C2, int M);
/* pooled (B,C2,M); mean2 / invstd2 / var_unbiased2 (C2) are BN2's batch statistics; running buffers of BN2 are
 * updated when given (momentum2 < 0: left to the caller).  One recomputation pass: it emits the statistics and the
 * raw max / min over the neighbours, from which the pooled output follows because BN's affine and the ReLU are
 * monotone in fp32; the backward finds the arg-max itself (first neighbour attaining the maximum, torch.max's rule) */
int amc3d_sa_tail_forward(int B, int C1, int C2, int M, int K, const float *y1, const float *mean1,
                          const float *invstd1, const float *gamma1, const float *beta1, const float *w2,
                          const float *gamma2, const float *beta2, float eps2, float momentum2, int relu2,
                          float *pooled, float *mean2, float *invstd2, float *var_unbiased2,
                          float *running_mean2, float *running_var2, long long *num_batches_tracked2,
                          float *zext_out, unsigned char *arg_out, void *workspace, size_t workspace_bytes, void *stream);
/* zext_out (B,C2,M) fp32 and arg_out (B,C2,M) bytes -- both or neither: per (b, c2, centroid) the raw extreme of conv2's output
 * over the 32 neighbours that BN2 + max-pool select (the maximum for gamma2 >= 0, the minimum otherwise) and the first
 * neighbour attaining it (torch.max's rule on the raw values).  Handed to amc3d_sa_tail_backward they replace its two
 * recomputation passes by the algebraic form: q is sparse (one neighbour per pooled element) and BatchNorm's backward adds
 * terms affine in z = W2 x1, so  dx1 = W2^T Dq q - (W2^T E W2) x1 - c  and  dW2 = Dq q x1^T - E W2 (x1 x1^T) - t (x1 1)^T
 * (csrc/sa_tail.hip): one pass over x1 with a C1 x C1 product and the Gram matrix on the MFMA and C2 rank-1 terms per centroid. */
/* dx1 (B,C1,M,32) = gradient w.r.t. relu(bn1(y1)) -- written as (B,M,32,C1) rows when dx1_position_major
 * (C1 % 4 == 0), the layout amc3d_grouped_conv_bn_backward_csr gathers from; dw2 (C2,C1) deterministic; dgamma2, dbeta2 (C2);
 * arg_out (B,C2,M) bytes or NULL: the neighbour each pooled gradient was routed to (what torch.max returns as
 * indices, pointnext_AA.py:166) -- the parity tests hold the routing fixed with it */
int amc3d_sa_tail_backward(int B, int C1, int C2, int M, int K, const float *y1, const float *mean1,
                           const float *invstd1, const float *gamma1, const float *beta1, const float *w2,
                           const float *mean2, const float *invstd2, const float *gamma2, const float *beta2,
                           int relu2, const float *dpooled, const float *zext /* of the forward, or NULL */,
                           const unsigned char *arg_ext /* of the forward, or NULL */, float *dx1, int dx1_position_major,
                           float *dw2, float *dgamma2, float *dbeta2, unsigned char *arg_out, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ---- training-mode BatchNorm fused with ReLU / neighbourhood max-pool ---------------------------
 * The reference runs nn.Conv -> nn.BatchNorm -> nn.ReLU(inplace) [-> torch.max over the neighbours]
 * as separate torch layers (openpoints/models/layers/conv.py:24-102, backbone/pointnext_AA.py:166).
 * Tensors are the reference's channel-major (B, C, L); statistics per channel over (B, L). */
size_t amc3d_bn_workspace_bytes(int C);  /* for amc3d_bn_stats; amc3d_bn_backward needs this + 8*C bytes */

/* mean (C), invstd = 1/sqrt(biased var + eps) (C), var_unbiased (C, for the running estimate) */
int amc3d_bn_stats(int B, int C, long L, float eps, const float *x, float *mean, float *invstd,
                   float *var_unbiased, void *workspace, size_t workspace_bytes, void *stream);
/* Training-mode BatchNorm forward in two launches (statistics; normalise [+ReLU] [+max over K neighbours]):
 * K == 0: y (B,C,L);  K > 0: L = M*K, y (B,C,M) and arg (B,C,M) bytes.  Writes mean, invstd, var_unbiased (C)
 * for the backward pass and, when running_mean != NULL, updates nn.BatchNorm's buffers
 * (torch/nn/modules/batchnorm.py; momentum < 0 = momentum None, the cumulative average). */
int amc3d_bn_forward(int B, int C, long L, int K, int relu, float eps, float momentum, const float *x,
                     const float *gamma, const float *beta, float *y, unsigned char *arg, float *mean,
                     float *invstd, float *var_unbiased, float *running_mean, float *running_var,
                     long long *num_batches_tracked, void *workspace, size_t workspace_bytes, void *stream);
/* nn.BatchNorm's training-mode bookkeeping in one launch: num_batches_tracked += 1 and the moving average of
 * mean / unbiased variance (torch/nn/modules/batchnorm.py); momentum < 0 stands for momentum=None (cumulative) */
int amc3d_bn_update_running(int C, float momentum, const float *mean, const float *var_unbiased,
                            float *running_mean, float *running_var, long long *num_batches_tracked, void *stream);

/* y (B,C,L) = [relu](((x - mean) * invstd) * gamma + beta) */
int amc3d_bn_act(int B, int C, long L, int relu, const float *x, const float *mean, const float *invstd,
                 const float *gamma, const float *beta, float *y, void *stream);

/* y (B,C,M) = max over K of [relu](bn(x (B,C,M,K))); arg (B,C,M) = first arg-max (bytes), K <= 255 */
int amc3d_bn_max(int B, int C, int M, int K, int relu, const float *x, const float *mean, const float *invstd,
                 const float *gamma, const float *beta, float *y, unsigned char *arg, void *stream);

/* backward of amc3d_bn_act (arg = NULL, K = 1, dy (B,C,L)) or amc3d_bn_max (arg given, dy (B,C,L/K)) under
 * batch statistics: dx (B,C,L), dgamma (C), dbeta (C) */
int amc3d_bn_backward(int B, int C, long L, int K, int relu, const float *x, const float *dy,
                      const unsigned char *arg, const float *mean, const float *invstd, const float *gamma,
                      const float *beta, float *dx, float *dgamma, float *dbeta, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ---- the same BatchNorm with statistics over all ranks (torch.nn.SyncBatchNorm: the reference converts every BN
 * layer to it when world_size > 1, examples/segmentation/main_AA.py:146-148, 820).  The per-channel sums leave the
 * library between two launches so that the caller can all-reduce them (RCCL):
 *     forward    amc3d_bn_sums -> all-reduce(sums[2C] ++ count) -> amc3d_bn_forward_synced
 *     backward   amc3d_bn_backward_sums -> all-reduce(dsums[2C]) -> amc3d_bn_backward_synced
 * Element counts are read from device memory: ranks may differ in size, and the sequence can be graph-captured. */
/* sums (2C doubles) = {sum x, sum x^2} per channel over this rank's (B, L) */
int amc3d_bn_sums(int B, int C, long L, const float *x, double *sums, void *workspace, size_t workspace_bytes,
                  void *stream);
/* amc3d_bn_forward's second launch from global statistics: sums_count = 2C sums, then the global element count */
int amc3d_bn_forward_synced(int B, int C, long L, int K, int relu, float eps, float momentum, const float *x,
                            const double *sums_count, const float *gamma, const float *beta, float *y,
                            unsigned char *arg, float *mean, float *invstd, float *var_unbiased, float *running_mean,
                            float *running_var, long long *num_batches_tracked, void *stream);
/* dsums (2C doubles) = {sum dq, sum dq*xhat} per channel over this rank; dgamma / dbeta (C) = the same, rank-local
 * (as torch's SyncBatchNorm leaves them; the gradient all-reduce averages parameter gradients) */
int amc3d_bn_backward_sums(int B, int C, long L, int K, int relu, const float *x, const float *dy,
                           const unsigned char *arg, const float *mean, const float *invstd, const float *gamma,
                           const float *beta, double *dsums, float *dgamma, float *dbeta, void *workspace,
                           size_t workspace_bytes, void *stream);
/* dx (B,C,L) from the global dsums and the global element count (*count, device memory) */
int amc3d_bn_backward_synced(int B, int C, long L, int K, int relu, const float *x, const float *dy,
                             const unsigned char *arg, const float *mean, const float *invstd, const float *gamma,
                             const float *beta, const double *dsums, const double *count, float *dx, void *stream);

/* ---- tail of an InvResMLP block (openpoints/models/backbone/pointnext_AA.py:296-307): y = relu(batch_norm(x) + res) with
 * batch statistics -- the last Conv1d -> BatchNorm1d of pwconv, `f += identity`, `self.act(f)` -- in the two launches of a
 * plain BatchNorm layer; backward: dres = dy * (y > 0), dx / dgamma / dbeta = BatchNorm backward of that.  x, res, y, dy,
 * dx, dres (B,C,L) fp32; workspace: amc3d_bn_workspace_bytes(C). */
int amc3d_bn_residual_forward(int B, int C, long L, float eps, float momentum, const float *x, const float *res,
                              const float *gamma, const float *beta, float *y, float *mean, float *invstd,
                              float *var_unbiased, float *running_mean, float *running_var,
                              long long *num_batches_tracked, void *workspace, size_t workspace_bytes, void *stream);
int amc3d_bn_residual_backward(int B, int C, long L, const float *x, const float *y, const float *dy, const float *mean,
                               const float *invstd, const float *gamma, const float *beta, float *dx, float *dres,
                               float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, void *stream);

/* ---- masked refinement of AMContrast3D++, the DualMasks rule with fusion 'MIN' (openpoints/AMContrast3D/MaskedRefine.py:55-131,
 * called per decoder level by models/backbone/pointnext_MM.py:541-560): points whose predicted ambiguity a lies in
 * [threshold, threshold_max] take the feature row of their least ambiguous neighbour,
 *     out = gamma * (f * ~mask + f_rows[best] * mask) + (1 - gamma) * f,
 * with the reference's two reinterpretations kept (f (B,D,n) viewed as (B*n, D) rows for the gather, the mask (B,1,n) broadcast
 * in the (B,D,n) indexing).  f, out, dout, df (B,D,n) fp32; a (B*n) fp32; nbr (B*n, k) int32 with row stride nbr_stride (self
 * match dropped); outputs kept for the backward: best (B*n) int32, mask (B*n) bytes; count (1) int32 = refined points.
 * Backward: df = (gamma * dout) * ~mask + (1 - gamma) * dout, plus gamma * dout of the masked elements added to row best[r]
 * (float atomics, like torch's index_add). */
size_t amc3d_masked_refine_workspace_ints(int m);
int amc3d_masked_refine_forward(int B, int D, int n, int k, int nbr_stride, const float *f, const float *a, const int *nbr,
                                float threshold, float threshold_max, float gamma, float *out, int *best, unsigned char *mask,
                                int *count, int *workspace, void *stream);
int amc3d_masked_refine_backward(int B, int D, int n, float gamma, const float *dout, const int *best, const unsigned char *mask,
                                 float *df, void *stream);

/* y = sigmoid(batch_norm(x)) with batch statistics -- nn.BatchNorm1d -> nn.Sigmoid of the APM towers of AMContrast3D++
 * (openpoints/AMContrast3D/APM/concatenation.py:20-60) -- in the two launches of a plain BatchNorm layer; backward from the saved
 * output: dq = dy * y (1 - y), then BatchNorm backward.  Layout and workspace as amc3d_bn_forward. */
int amc3d_bn_sigmoid_forward(int B, int C, long L, float eps, float momentum, const float *x, const float *gamma,
                             const float *beta, float *y, float *mean, float *invstd, float *var_unbiased,
                             float *running_mean, float *running_var, long long *num_batches_tracked, void *workspace,
                             size_t workspace_bytes, void *stream);
int amc3d_bn_sigmoid_backward(int B, int C, long L, const float *x, const float *y, const float *dy, const float *mean,
                              const float *invstd, const float *gamma, const float *beta, float *dx, float *dgamma, float *dbeta,
                              void *workspace, size_t workspace_bytes, void *stream);

/* ---- gradient-norm clipping + AdamW over all parameter tensors in two launches ---------------------
 * Replaces  torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm, norm_type=2); optimizer.step()  with
 * torch.optim.AdamW (examples/segmentation/main_AA.py:586-592; optimizer built by openpoints/optim/optim_factory.py:160-230).
 * table: ntensors records in DEVICE memory; block_map: nblocks pairs (tensor index, chunk index) in device memory, one per
 * amc3d_adamw_chunk() elements of every tensor; every record's step (a device float of its own: torch counts steps per
 * parameter) is incremented by the call; partial: nblocks doubles of scratch; total_norm: device float or NULL (the 2-norm of all gradients before clipping, what clip_grad_norm_ returns);
 * max_grad_norm <= 0: no clipping.  The clipped gradient is used for the update, the .grad tensors are left as they are. */
typedef struct {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    float *step;
    long long numel;
    float weight_decay;
    float lr;
} amc3d_adamw_tensor;
int amc3d_adamw_chunk(void);
int amc3d_adamw_step(const void *table, const int *block_map, int nblocks, double beta1, double beta2, float eps,
                     float max_grad_norm, double *partial, float *total_norm, void *stream);

/* ---- training-time augmentation of a batch of cropped clouds (the loader's transform chain, cfgs/s3dis/default.yaml:33-43:
 * ChromaticAutoContrast, PointCloudScaling, PointCloudXYZAlign, PointCloudRotation, PointCloudJitter, ChromaticDropGPU,
 * ChromaticNormalize -- openpoints/transforms/point_transform_cpu.py:192-209, point_transformer_gpu.py:70-89,135-164,216-229,
 * 267-311,373-409 -- which the reference applies per cloud in loader workers).  Two launches for the whole batch.  The random
 * draws are the caller's: params (b,24) floats per cloud = {contrast 0/1, blend, scale[3], rot[9] row-major (pos' = pos @ rot^T),
 * drop 0/1, 9 unused}, noise (b,n,3) standard-normal draws.  pos (b,n,3), color (b,n,3) (0..255, or 0..1: colours are divided by
 * 255 when their maximum exceeds 1, as ChromaticNormalize does) -> pos_out (b,n,3), x_out (b,n,3) = (colour - mean) / std,
 * heights (b,n) = the UNtransformed gravity coordinate (dataset/s3dis/s3dis.py:141-142). */
size_t amc3d_augment_workspace_bytes(int b);
int amc3d_augment_clouds(int b, int n, int gravity_dim, float jitter_sigma, float jitter_clip, const float *pos,
                         const float *color, const float *noise, const float *params, const float *color_mean,
                         const float *color_std, float *pos_out, float *x_out, float *heights, void *workspace,
                         size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AMC3D_H */
