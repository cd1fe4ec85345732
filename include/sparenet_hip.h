/*
 * sparenet_hip.h -- C ABI of libsparenet_hip.so, the MI355X (gfx950) native
 * implementation of SpareNet's per-step loss / render hot path.
 *
 * Drop-in boundary: each entry point replaces one pybind function of the
 * reference's torch C++ extensions under /root/reference/cuda/ (cited per
 * function as file:line).  No torch types cross this boundary: every argument
 * is a raw DEVICE pointer (hipMalloc'ed / torch CUDA-tensor storage), an int
 * size or a float parameter, plus the hipStream_t (as void*) to launch on.
 *
 * Conventions
 *   - all tensors are contiguous row-major fp32 / int32, sizes in elements;
 *   - functions are asynchronous: they enqueue kernels on `stream` and return;
 *   - return value: 0 (hipSuccess) on success; a positive hipError_t if a
 *     launch failed; SN_EINVAL (-22) for rejected arguments (the reference
 *     prints and carries on, asserts in Python or exit(-1)s; here every
 *     failure is an error code + sn_last_error() text, and the Python mirror
 *     raises);
 *   - workspace: ops that need scratch take (workspace, workspace_bytes); the
 *     matching sn_*_workspace_bytes() says how much.  Workspace contents need
 *     no initialisation by the caller;
 *   - thread safety: no global mutable state except the thread-local
 *     last-error string; safe to call concurrently from several host threads
 *     on different streams/devices (the reference is driven that way by
 *     nn.DataParallel, runners/sparenet_runner.py:32-34).
 */
#ifndef SPARENET_HIP_H
#define SPARENET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN_EINVAL (-22)
/* a team barrier of an earlier persistent EMD launch on this device gave up (see sn_emd_forward) */
#define SN_ETIMEDOUT (-110)
#define SN_ABI_VERSION 4

int sn_abi_version(void);
/* hash of the HIP sources this library was built from (profiles/ measurements carry it) */
const char *sn_build_id(void);
/* thread-local text of the last failure on the calling thread ("" if none) */
const char *sn_last_error(void);
/* 0, or SN_ETIMEDOUT if a bounded wait inside an earlier multi-workgroup launch (persistent EMD auction, density
 * sampler teams) on the CURRENT device gave up; clears the condition.  No synchronisation (one word of pinned host
 * memory is read): call it where the host already waits for the GPU -- after reading a loss value -- to learn about
 * a time-out in the step that just finished instead of at the next sn_emd_* / sn_mds call. */
int sn_device_status(void);

/* Optional per-kernel timing, off by default (measurement aid, not part of the
 * reference interface): when enabled the heavy kernels ("chamfer_fwd", "emd_auction",
 * "expansion_fwd", "mds", "p2i_max_splat" = the binned gather) are bracketed by hipEventRecord on the
 * stream they are launched on.  sn_prof_read waits for the recorded events and returns
 * the number of launches of `name` since the last reset and their summed duration. */
void sn_prof_enable(int on);
long long sn_prof_read(const char *name, double *total_ms);
void sn_prof_reset(void);
/* With sn_prof_enable(1) the persistent EMD auction also records its own EXECUTION window (first working workgroup's
 * start to the last one's end, in-kernel 100 MHz clock): launches since the last reset on the current device and the
 * sum of the windows.  The HIP-event bracket of "emd_auction" additionally contains the launch's wait for compute
 * units (the grid needs every CU empty), i.e. the schedule around it. */
long long sn_emd_prof_exec(double *total_ms, int reset);

/* ------------------------------------------------------------------ Chamfer
 * replaces cd.forward_cuda  = chamfer_distance_forward_cuda
 *          (cuda/chamfer_distance/chamfer_distance.cpp:26-38,186;
 *           kernel chamfer_distance.cu:7-155)
 * dist1[b,j] = min_k |xyz1[b,j]-xyz2[b,k]|^2, idx1 = lowest k attaining it;
 * symmetric for dist2/idx2.  b>=1, n>=1, m>=1. */
int sn_chamfer_forward(const float *xyz1, const float *xyz2, int b, int n,
                       int m, float *dist1, int *idx1, float *dist2,
                       int *idx2, void *stream);
/* replaces cd.backward_cuda = chamfer_distance_backward_cuda
 *          (chamfer_distance.cpp:40-55,188; kernel chamfer_distance.cu:159-209)
 * gradxyz1/gradxyz2 are fully overwritten (no pre-zeroing needed).  The scatter of the reference is a
 * gather over inverse lists built in `workspace` (integer atomics only): the result is bit-reproducible and
 * bit-equal to the reference's CPU path, which adds the same terms in the same order. */
/* Same result as sn_chamfer_forward (distances bit for bit, indices = lowest k attaining the
 * minimum), computed as a spatially pruned search: both clouds are Morton sorted into the
 * workspace, superblocks of 64 targets are skipped by bounding box, the surviving pairs are
 * filtered on the fp32 matrix cores and only the candidates are evaluated with the reference's
 * expression.  Pays off from a few thousand points per cloud. */
size_t sn_chamfer_workspace_bytes(int b, int n, int m);
int sn_chamfer_forward_sorted(const float *xyz1, const float *xyz2, int b, int n,
                              int m, float *dist1, int *idx1, float *dist2,
                              int *idx2, void *workspace, size_t workspace_bytes,
                              void *stream);
size_t sn_chamfer_backward_workspace_bytes(int b, int n, int m);
int sn_chamfer_backward(const float *xyz1, const float *xyz2,
                        const float *graddist1, const float *graddist2,
                        const int *idx1, const int *idx2, int b, int n, int m,
                        float *gradxyz1, float *gradxyz2, void *workspace,
                        size_t workspace_bytes, void *stream);

/* Host tensors (replaces cd.forward / cd.backward = chamfer_distance_forward / chamfer_distance_backward,
 * chamfer_distance.cpp:91-180,185,187 -- the branch ChamferDistanceFunction takes for CPU tensors,
 * cuda/chamfer_distance/chamfer_distance.py:31-32,53-54; BASELINE config 1).  All pointers are HOST pointers, the
 * calls are synchronous.  Same values as the device entry points and as the reference's CPU code bit for bit: fp32
 * distances (dx*dx + dy*dy) + dz*dz without contraction, lowest index among equal minima, the backward's additions in
 * the reference's order.  threads: worker threads; an explicit value > 0 is taken as given.  0 = the default: one per
 * hardware thread, or SN_HOST_THREADS when set -- the default (either form) is then reduced so that every thread
 * has at least four work items (clouds x query blocks).  Never more than 256, never more than there are items.
 * The library's own host code, not a fallback: device tensors never take this path. */
int sn_chamfer_forward_host(const float *xyz1, const float *xyz2, int b, int n,
                            int m, float *dist1, int *idx1, float *dist2,
                            int *idx2, int threads);
int sn_chamfer_backward_host(const float *xyz1, const float *xyz2,
                             const float *graddist1, const float *graddist2,
                             const int *idx1, const int *idx2, int b, int n, int m,
                             float *gradxyz1, float *gradxyz2, int threads);

/* ---------------------------------------------------------------------- EMD
 * replaces emd.forward = emd_forward -> emd_cuda_forward
 *          (cuda/emd/emd.cpp:13-17,26; emd_cuda.cu:228-282) including the 12
 *          scratch tensors the Python module allocates and initialises
 *          (cuda/emd/emd_module.py:43-54): they live in `workspace` here.
 * Requirements as the reference: n % 1024 == 0, b <= 512 (emd_module.py:36-39).
 * dist[b,n] fp32, assignment[b,n] int32.
 * All iterations run in ONE persistent launch (one workgroup per compute unit; teams of
 * workgroups own a cloud and synchronise through bounded, placement-independent barriers).
 * Environment: SN_EMD_CHECK=1 makes the call synchronise and fail if a barrier timed out.
 * stats (optional device pointer, may be NULL): 2 x int64, zeroed by the caller
 *   stats[0] += sum over iterations and batch of unassigned_count * n
 *               (effective pair evaluations); stats[1] += iterations that had
 *               at least one bidder (one atomic per cloud per iteration).
 * With SN_EMD_DIAG=1|2 in the environment the call also leaves phase timers of the first team
 * in the workspace, 16 + 64*64 int64 words at sn_emd_diag_offset (tools/emd_ab.py).
 *
 * Failure behaviour (the reference returns an error code from emd_cuda_forward, emd_cuda.cu:276-281): the
 * auction is ONE persistent launch whose workgroups wait for each other; every wait is bounded (> 2 s).  If a
 * barrier gives up (a CU withheld from the launch by another tenant or a debugger), every workgroup leaves,
 * the unfinished clouds get dist = NaN and assignment = -1, and the NEXT sn_emd_forward / sn_emd_backward call
 * on that device returns SN_ETIMEDOUT without a host synchronisation (SN_EMD_CHECK=1: the failing call itself
 * synchronises and returns SN_EINVAL).
 * Memory-model note: the launch hands data between workgroups with relaxed coherent accesses and no fences, and
 * keeps the stores of a team that sits on one XCD in that XCD's L2.  This is gfx950 behaviour, verified once per
 * device by a litmus kernel at the first call; on failure, or with SN_EMD_SAFE=1, the launch uses agent-scope
 * release / acquire barriers and agent-scope stores instead (sn_emd_mode() reports which). */
size_t sn_emd_workspace_bytes(int b, int n);
size_t sn_emd_diag_offset(int b, int n);
int sn_emd_mode(void);
/* runs the memory-model litmus of the current device now (what the first sn_emd_forward does lazily, on a private
 * stream, without a device-wide synchronisation) and returns sn_emd_mode()'s answer; -1: undecided (busy device). */
int sn_emd_selftest(void);
int sn_emd_forward(const float *xyz1, const float *xyz2, int b, int n,
                   float eps, int iters, float *dist, int *assignment,
                   void *workspace, size_t workspace_bytes,
                   long long *stats, void *stream);
/* replaces emd.backward = emd_backward -> emd_cuda_backward
 *          (cuda/emd/emd.cpp:19-23,27; emd_cuda.cu:284-316)
 * gradxyz1 is fully overwritten; gradient w.r.t. xyz2 is identically zero. */
int sn_emd_backward(const float *xyz1, const float *xyz2,
                    const float *graddist, const int *assignment, int b, int n,
                    float *gradxyz1, void *stream);

/* -------------------------------------------------------- expansion penalty
 * replaces expansion_penalty.forward = expansion_penalty_forward
 *          (cuda/expansion_penalty/expansion_penalty.cpp:8-12,20;
 *           expansion_penalty_cuda.cu:7-165) without the two [b, n*512]
 *          neighbor/cost scratch tensors
 *          (expansion_penalty_module.py:33-34).
 * primitive_size: power of two, 2..512, n % primitive_size == 0.
 * mean_mst_length[b] = sum over patches (ascending patch order) of the patch's
 * mean MST edge length, UN-normalised exactly like the reference kernel leaves
 * it; expansion_penalty_module.py:40 divides by n/primitive_size in Python and
 * the host mirror does the same. */
size_t sn_expansion_workspace_bytes(int b, int n, int primitive_size);
int sn_expansion_forward(const float *xyz, int b, int n, int primitive_size,
                         float alpha, float *dist, int *assignment,
                         float *mean_mst_length, void *workspace,
                         size_t workspace_bytes, void *stream);
/* replaces expansion_penalty.backward (expansion_penalty.cpp:14-17,21;
 *          expansion_penalty_cuda.cu:167-198); gradxyz fully overwritten. */
int sn_expansion_backward(const float *xyz, const float *graddist,
                          const int *assignment, int b, int n, float *gradxyz,
                          void *stream);

/* ---------------------------------------------------------------------- MDS
 * replaces MDS.minimum_density_sampling (cuda/MDS/MDS.cpp:114-135,140;
 *          kernel MDS_cuda.cu:91-268).  idx[b,m] int32, m <= n.
 * The `temp` tensor MDS.cpp:119-121 allocates lives in registers; only clouds
 * with more than 24576 points need `workspace` (sn_mds_workspace_bytes() > 0).
 * The density kernel is sn_expf (include/sn_expf.h), not libm/OCML expf.
 * Clouds of 2048 .. 20352 points are cluster-sorted and sampled either by one workgroup each or by a TEAM of up to
 * 32 workgroups per cloud that takes several exact picks per exchange (every cloud of >= 8192 points when teams of
 * >= 8 fit the device; index sequences are the reference's either way).  A team's bounded waits can give up when the
 * device is shared with something that keeps compute units from the launch: the row is then -1 and the next sn_mds /
 * sn_emd_* call -- or sn_device_status() -- reports SN_ETIMEDOUT. */
size_t sn_mds_workspace_bytes(int b, int n);
int sn_mds(const float *xyz, int b, int n, int m, const float *mean_mst_length,
           int *idx, void *workspace, size_t workspace_bytes, void *stream);
/* replaces MDS.gather_forward / MDS.gather_backward (MDS.cpp:54-113,138-139;
 *          kernels MDS_cuda.cu:29-79). feat[b,c,n], idx[b,m], out[b,c,m].
 * grad_feat is fully overwritten. */
int sn_gather_forward(const float *feat, const int *idx, int b, int c, int n,
                      int m, float *out, void *stream);
int sn_gather_backward(const float *grad_out, const int *idx, int b, int c,
                       int n, int m, float *grad_feat, void *stream);

/* ---------------------------------------------------------------------- p2i
 * replaces p2i_op.p2i_max_forward_gpu / p2i_max_backward_gpu
 *          (cuda/p2i_op/ext.cpp:8-9; p2i_max.h:145-232; functors :7-143)
 * points[npoints,2] pixel-space (row, col); feat[npoints,channels];
 * batch_inds[npoints]; background[batch,channels,h,w].
 * out/out_ids[batch,channels,h,w]; workspace holds the packed 64-bit image.
 * Ties between equal splat values resolve to the LOWEST point id (the
 * reference's GPU order is a race; its sequential CPU functor gives this). */
size_t sn_p2i_max_workspace_bytes(int batch, int channels, int h, int w);
int sn_p2i_max_forward(const float *points, const float *feat,
                       const int *batch_inds, const float *background,
                       int npoints, int channels, int batch, int h, int w,
                       float radius, float *out, int *out_ids, void *workspace,
                       size_t workspace_bytes, void *stream);
/* Several kernel radii over the same points / features / background in one pass
 * (ComputeDepthMaps calls p2i once per radius with identical inputs,
 * utils/p2i_utils.py:230-251): radii[nradii] is a HOST array, 1 <= nradii <= 4;
 * out / out_ids hold nradii [batch,channels,h,w] tensors, each equal to what
 * sn_p2i_max_forward returns for that radius: one after the other ([R,B,C,H,W],
 * image_major = 0) or interleaved per image ([B,R,C,H,W], image_major = 1: the layout
 * of the [B, len(radius_list), S, S] tensor ComputeDepthMaps returns, so that nothing
 * has to be transposed afterwards; radii <= 16 px).  background may be NULL = all zeros
 * (what ComputeDepthMaps passes; radii <= 16 px): no tensor is allocated, filled or read. */
size_t sn_p2i_max_multi_workspace_bytes(int npoints, int batch, int channels,
                                        int h, int w);
/* Test hook, not part of the reference surface: the renderer's two fp32 series in
 * u = r^2 / R^2 -- weight[i] ~ (cos(pi sqrt(u)) + 1) / 2 and slope[i] ~
 * sin(pi sqrt(u)) / (pi sqrt(u)) -- evaluated on the device for n values of u in
 * [0, 1]; the tests pin the error bounds the forward's decision band relies on. */
int sn_p2i_series(const float *u, int n, float *weight, float *slope, void *stream);
int sn_p2i_max_forward_multi(const float *points, const float *feat,
                             const int *batch_inds, const float *background,
                             int npoints, int channels, int batch, int h, int w,
                             const float *radii, int nradii, int image_major,
                             float *out, int *out_ids, void *workspace,
                             size_t workspace_bytes, void *stream);
/* points_grad[npoints,2], feat_grad[npoints,channels],
 * background_grad[batch,channels,h,w] are fully overwritten.
 * batch_inds: the forward's batch_inds, or NULL (the reference's backward does not
 * receive it; without it every image plane is searched for the point's wins). */
size_t sn_p2i_max_backward_workspace_bytes(int batch, int channels, int h, int w);
int sn_p2i_max_backward(const float *out_grad, const int *out_ids,
                        const float *points, const float *feat,
                        const int *batch_inds, int npoints, int channels,
                        int batch, int h, int w, float radius,
                        float *points_grad, float *feat_grad,
                        float *background_grad, void *workspace,
                        size_t workspace_bytes, void *stream);
/* Backward of nradii splats that share points / features (the gradients of the radii
 * are summed, which is what autograd does with the reference's per-radius calls).
 * out_grad / out_ids: nradii [batch,channels,h,w] tensors in the layout of the forward
 * (image_major as there); background_grad may be NULL (not wanted).  Pixel-centric:
 * every pixel adds its terms to its winner in 64-bit fixed point (integer atomics), so
 * the sums are exact and bit-reproducible. */
size_t sn_p2i_max_backward_multi_workspace_bytes(int npoints, int channels);
int sn_p2i_max_backward_multi(const float *out_grad, const int *out_ids,
                              const float *points, const float *feat,
                              int npoints, int channels, int batch, int h, int w,
                              const float *radii, int nradii, int image_major,
                              float *points_grad, float *feat_grad,
                              float *background_grad,
                              void *workspace, size_t workspace_bytes,
                              void *stream);
/* replaces p2i_op.p2i_sum_forward_gpu / p2i_sum_backward_gpu
 *          (cuda/p2i_op/ext.cpp:6-7; p2i_sum.h:133-214; functors :7-131)
 * out must hold a copy of background on entry (the reference clones it,
 * p2i_sum.h:147); it is accumulated in place. */
int sn_p2i_sum_forward(const float *points, const float *feat,
                       const int *batch_inds, int npoints, int channels,
                       int batch, int h, int w, float radius, float *out,
                       void *stream);
int sn_p2i_sum_backward(const float *out_grad, const float *points,
                        const float *feat, const int *batch_inds, int npoints,
                        int channels, int batch, int h, int w, float radius,
                        float *points_grad, float *feat_grad, void *stream);

/* float64 tensors: the reference dispatches its functors on float and double (p2i_max.h:177,218,
 * p2i_sum.h:162,201) and its own test is a float64 gradcheck (cuda/p2i_op/p2i_test.py:23-35).  Same
 * semantics as the fp32 entry points above (lowest point id on equal values), simple atomics-based
 * kernels -- the rendering path of SpareNet itself is fp32. */
size_t sn_p2i_f64_workspace_bytes(int batch, int channels, int h, int w);
int sn_p2i_max_forward_f64(const double *points, const double *point_features, const int *batch_inds,
                           const double *background, int npoints, int channels, int batch, int h, int w,
                           double kernel_radius, double *out, int *out_point_ids, void *workspace,
                           size_t workspace_bytes, void *stream);
int sn_p2i_max_backward_f64(const double *out_grad, const int *out_point_ids, const double *points,
                            const double *point_features, int npoints, int channels, int batch, int h,
                            int w, double kernel_radius, double *points_grad, double *point_features_grad,
                            double *background_grad, void *stream);
int sn_p2i_sum_forward_f64(const double *points, const double *point_features, const int *batch_inds,
                           int npoints, int channels, int batch, int h, int w, double kernel_radius,
                           double *out, void *stream);
int sn_p2i_sum_backward_f64(const double *out_grad, const double *points, const double *point_features,
                            const int *batch_inds, int npoints, int channels, int batch, int h, int w,
                            double kernel_radius, double *points_grad, double *point_features_grad,
                            void *stream);

/* ---------------------------------------------------------- EdgeConv k-NN graph
 * replaces knn() / get_graph_feature() of models/sparenet_generator.py:852-906 (GPU branch:
 * the un-vendored KNN_CUDA 0.2 wheel).  inner[b,n,n] = x^T x (a plain batched GEMM, computed
 * by the caller with rocBLAS / torch.bmm), xx[b,n] = |x_j|^2.  idx[b,n,k] (int64): the k
 * smallest  |x_j|^2 - 2 x_i.x_j  per row, ascending, equal scores by lower index (the point
 * itself comes first); k in {1, 2, 4, 8, 16, 20, 32}. */
int sn_knn_topk(const float *inner, const float *xx, int b, int n, int k,
                long long *idx, void *stream);
/* The same search as ONE kernel on the fp32 matrix cores, from x[b,c,n] directly: the score tiles
 * 2 x_i.x_j - |x_j|^2 live in the MFMA accumulators and every lane keeps the k best of its queries
 * in registers, so the [b,n,n] matrix is never written (knn_mfma.hip).  idx[b,n,k] as above, the
 * point itself first; 1 <= k <= 20.  workspace: sn_knn_workspace_bytes(b, n) (the squared norms). */
size_t sn_knn_workspace_bytes(int b, int n);
int sn_knn(const float *x, int b, int c, int n, int k, long long *idx,
           void *workspace, size_t workspace_bytes, void *stream);
/* x[b,c,n], idx[b,n,k] -> out[b,2c,n,k]: out[:, ch] = x[idx] - x, out[:, c + ch] = x
 * (cat((feature - x, x), dim=3).permute(0, 3, 1, 2), :899-905); backward: grad_x[b,c,n]. */
int sn_graph_feature_forward(const float *x, const long long *idx, int b, int c,
                             int n, int k, float *out, void *stream);
size_t sn_graph_feature_backward_workspace_bytes(int b, int n, int k);
int sn_graph_feature_backward(const float *grad_out, const long long *idx, int b,
                              int c, int n, int k, float *grad_x, void *workspace,
                              size_t workspace_bytes, void *stream);

/* ------------------------------------------------------- depth-map projection
 * The per-view glue of ComputeDepthMaps.forward (utils/p2i_utils.py:211-228 and the NDC ->
 * pixel rescale of cuda/p2i_op/__init__.py:117-121) fused into two small kernels each way:
 * data[npoints,3] (all batch elements), matrix16 = P@V row major (HOST array), extent =
 * image_size - 1.  Outputs: pixel[npoints,2] (row, col) for sn_p2i_max_forward[_multi],
 * z[npoints] (projected depth), zminmax[2] (device scratch, order-preserving bits of the
 * global min / max of z), feat[npoints] = 1 - (z - zmin) / (zmax - zmin).
 * Backward: g_pixel / g_feat may be NULL; workspace32 = 32 bytes of device scratch;
 * includes the gradient paths through zmin / zmax as torch's autograd does. */
int sn_depth_project_forward(const float *data, long npoints, const float *matrix16,
                             float extent, float *pixel, float *z,
                             unsigned *zminmax, float *feat, void *stream);
int sn_depth_project_backward(const float *data, long npoints, const float *matrix16,
                              float extent, const float *z, const unsigned *zminmax,
                              const float *g_pixel, const float *g_feat,
                              void *workspace32, float *g_data, void *stream);

/* The same for up to 8 views at once (the views of a ComputeDepthMaps sweep become part of the batch:
 * pixel / z / feat are [nviews, npoints], zminmax [nviews, 2], normalisation per view as in the reference;
 * g_data [npoints, 3] receives the sum over the views).  matrices16: nviews x 16 host floats.
 * workspace: 32 bytes per view. */
int sn_depth_project_forward_views(const float *data, long npoints, const float *matrices16, int nviews,
                                   float extent, float *pixel, float *z, unsigned *zminmax, float *feat,
                                   void *stream);
int sn_depth_project_backward_views(const float *data, long npoints, const float *matrices16, int nviews,
                                    float extent, const float *z, const unsigned *zminmax,
                                    const float *g_pixel, const float *g_feat, void *workspace,
                                    float *g_data, void *stream);

/* ----------------------------------------------------------------- gridding
 * replaces gridding.forward / backward (cuda/gridding/gridding_cuda.cpp:44-67,
 *          94-95; gridding.cu:29-211, 213-335).  ptcloud[b,npts,3] already
 *          scaled; bounds are [-scale/2, scale/2-1] per axis as
 *          cuda/gridding/__init__.py:16-19 passes them.
 * grid[b,scale^3] fully overwritten; weights[b,npts,8,3]; indexes[b,npts,8]. */
int sn_gridding_forward(const float *ptcloud, int b, int npts, int scale,
                        float *grid, float *weights, int *indexes,
                        void *stream);
/* The same with the reference module's padding rule done in the kernel: rows whose three
 * (already scaled) coordinates sum to zero are padding and contribute nothing
 * (cuda/gridding/__init__.py:41-47 filters them per sample on the host); their weights
 * are written as 0 and their indexes as -1, so the backward gives them a zero gradient. */
int sn_gridding_forward_padded(const float *ptcloud, int b, int npts, int scale,
                               float *grid, float *weights, int *indexes,
                               void *stream);
int sn_gridding_backward(const float *grad_grid, const float *weights,
                         const int *indexes, int b, int npts, int nverts,
                         float *grad_ptcloud, void *stream);
/* replaces gridding_distance.forward (cuda/gridding_loss/gridding_distance_cuda.cpp:
 *          forward -> gridding_distance.cu:179-212, kernel :29-177): integer bounds
 *          [min, max] per axis, grid[b, nverts, 8] with nverts = len_x len_y len_z (one
 *          accumulator per vertex and corner role), weights[b,npts,8,3], indexes[b,npts,8]
 *          (slot = vertex * 8 + corner).  gridding_distance.backward (:307-329) is
 *          sn_gridding_backward with nverts * 8 slots (grad kernel :214-314). */
int sn_gridding_dist_forward(const float *ptcloud, int b, int npts, int min_x,
                             int max_x, int min_y, int max_y, int min_z, int max_z,
                             float *grid, float *weights, int *indexes,
                             void *stream);
/* replaces gridding.rev_forward / rev_backward (gridding_cuda.cpp:69-91,96-97;
 *          gridding_reverse.cu:30-122, 124-236). grid[b,scale,scale,scale]. */
int sn_gridding_reverse_forward(const float *grid, int b, int scale,
                                float *ptcloud, void *stream);
/* ptcloud = the raw rev_forward output (what GriddingReverseFunction saves,
 * cuda/gridding/__init__.py:55), not the module's rescaled one. */
int sn_gridding_reverse_backward(const float *grad_ptcloud, const float *grid,
                                 const float *ptcloud, int b, int scale,
                                 float *grad_grid, void *stream);

/* --------------------------------------------------- cubic feature sampling
 * replaces cubic_feature_sampling.forward / backward
 *          (cuda/cubic_feature_sampling/cubic_feature_sampling_cuda.cpp:36-61;
 *           cubic_feature_sampling.cu:29-133, 135-205).
 * ptcloud[b,npts,3] in voxel space; feat[b,c,scale^3];
 * out[b,npts,(2ns)^3,c]; idx[b,npts,(2ns)^3]; grad_feat fully overwritten. */
int sn_cubic_forward(const float *ptcloud, const float *feat, int b, int npts,
                     int c, int scale, int ns, float *out, int *idx,
                     void *stream);
int sn_cubic_backward(const float *grad_out, const int *idx, int b, int npts,
                      int c, int scale, int ns, float *grad_feat,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARENET_HIP_H */
