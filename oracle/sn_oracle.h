/*
 * sn_oracle.h -- CPU restatement of the SpareNet loss/render hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * kernels in sparenet_amd/csrc.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product path (sparenet_amd)
 * never calls into it and fails loudly when the HIP library is missing.
 *
 * Every function restates one reference kernel family in plain C (gcc,
 * -ffp-contract=off so that products and sums round separately, exactly like
 * the reference's own g++ build of its CPU Chamfer path), citing the
 * reference file:line it follows.  Pinning status per op is listed in
 * oracle/README.md and DESIGN.md.
 */
#ifndef SN_ORACLE_H
#define SN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Chamfer: cuda/chamfer_distance/chamfer_distance.cpp:57-112 (nnsearch) */
void oracle_chamfer_forward(const float *xyz1, const float *xyz2, int b, int n,
                            int m, float *dist1, int *idx1, float *dist2,
                            int *idx2);
/* multi-threaded (OpenMP over batch x query blocks) variant, same results */
void oracle_chamfer_forward_mt(const float *xyz1, const float *xyz2, int b,
                               int n, int m, float *dist1, int *idx1,
                               float *dist2, int *idx2);
/* cuda/chamfer_distance/chamfer_distance.cpp:114-180 (deterministic order) */
void oracle_chamfer_backward(const float *xyz1, const float *xyz2,
                             const float *graddist1, const float *graddist2,
                             const int *idx1, const int *idx2, int b, int n,
                             int m, float *gradxyz1, float *gradxyz2);

/* ---- EMD auction: cuda/emd/emd_cuda.cu:23-282 + emd_module.py:41-54 ------
 * Sequential restatement.  Tie rules (see DESIGN.md "EMD canonical rules"):
 *   Bid   best_i on exact ties = argmin (chunk(k), k), chunk per emd_cuda.cu:136-139
 *   GetMax several bidders within +-1e-6 of the max -> highest j wins
 *          (sequential ascending-j "last writer wins", emd_cuda.cu:188-191)
 * trace (optional, may be NULL): per-iteration total unassigned count
 * [iters] summed over batch, written before each Bid.
 * Returns effective pair evaluations sum_it sum_b unass_cnt*n. */
long long oracle_emd_forward(const float *xyz1, const float *xyz2, int b,
                             int n, float eps, int iters, float *dist,
                             int *assignment, float *price_out,
                             int *trace_unass);
long long oracle_emd_forward_mt(const float *xyz1, const float *xyz2, int b,
                                int n, float eps, int iters, float *dist,
                                int *assignment, float *price_out,
                                int *trace_unass);
/* cuda/emd/emd_cuda.cu:284-300 */
void oracle_emd_backward(const float *xyz1, const float *xyz2,
                         const float *graddist, const int *assignment, int b,
                         int n, float *gradxyz1);

/* ---- Expansion penalty: cuda/expansion_penalty/expansion_penalty_cuda.cu:7-149
 * mean_mst_length[b] = sum over patches of the patch's mean MST edge length,
 * UN-normalised like the kernel leaves it; the Python module divides by
 * n/primitive_size afterwards (expansion_penalty_module.py:40). */
void oracle_expansion_forward(const float *xyz, int b, int n,
                              int primitive_size, float alpha, float *dist,
                              int *assignment, float *mean_mst_length);
/* expansion_penalty_cuda.cu:167-184 */
void oracle_expansion_backward(const float *xyz, const float *graddist,
                               const int *assignment, int b, int n,
                               float *gradxyz);

/* ---- MDS: cuda/MDS/MDS_cuda.cu:91-211 (intended, race-free semantics) ----
 * exp_mode 0: libm expf (what the reference source says, not bit-portable)
 * exp_mode 1: sn_expf polynomial shared verbatim with the HIP kernel
 * bs_override > 0 replaces the reference's thread count (tie order only) */
void oracle_mds(const float *xyz, int b, int n, int m,
                const float *mean_mst_length, int exp_mode, int bs_override,
                int *idx);
/* MDS_cuda.cu:29-41 / :55-69 */
void oracle_gather_forward(const float *feat, const int *idx, int b, int c,
                           int n, int m, float *out);
void oracle_gather_backward(const float *grad_out, const int *idx, int b,
                            int c, int n, int m, float *grad_feat);

/* ---- p2i: cuda/p2i_op/p2i_max.h:7-143, p2i_sum.h:7-131, utility.h:82-100 --
 * points are pixel-space (row, col) pairs, as handed to the op by
 * cuda/p2i_op/__init__.py:117-121. out must be pre-filled with background,
 * ids with -1. */
void oracle_p2i_max_forward(const float *points, const float *feat,
                            const int *batch_inds, int npoints, int channels,
                            int batch, int h, int w, float radius, float *out,
                            int *out_ids);
void oracle_p2i_max_forward_mt(const float *points, const float *feat,
                            const int *batch_inds, int npoints, int channels,
                            int batch, int h, int w, float radius, float *out,
                            int *out_ids);
void oracle_p2i_max_backward(const float *out_grad, const int *out_ids,
                             const float *points, const float *feat,
                             int npoints, int channels, int batch, int h,
                             int w, float radius, float *points_grad,
                             float *feat_grad, float *background_grad);
void oracle_p2i_max_backward_exact(const float *out_grad, const int *out_ids,
                             const float *points, const float *feat,
                             int npoints, int channels, int batch, int h,
                             int w, float radius, float *points_grad,
                             float *feat_grad);
void oracle_p2i_sum_forward(const float *points, const float *feat,
                            const int *batch_inds, int npoints, int channels,
                            int batch, int h, int w, float radius, float *out);
void oracle_p2i_sum_backward(const float *out_grad, const float *points,
                             const float *feat, const int *batch_inds,
                             int npoints, int channels, int batch, int h,
                             int w, float radius, float *points_grad,
                             float *feat_grad);

/* ---- gridding family: cuda/gridding/gridding.cu:29-177,213-312;
 *      gridding_reverse.cu:30-103,124-214;
 *      cuda/cubic_feature_sampling/cubic_feature_sampling.cu:29-102,135-174 */
void oracle_gridding_forward(const float *ptcloud, int b, int npts, int scale,
                             float *grid, float *weights, int *indexes);
void oracle_gridding_dist_forward(const float *ptcloud, int b, int npts, int min_x, int max_x,
                                  int min_y, int max_y, int min_z, int max_z, float *grid,
                                  float *weights, int *indexes);
void oracle_gridding_backward(const float *grad_grid, const float *weights,
                              const int *indexes, int b, int npts, int nverts,
                              float *grad_ptcloud);
void oracle_gridding_reverse_forward(const float *grid, int b, int scale,
                                     float *ptcloud);
void oracle_gridding_reverse_backward(const float *grad_ptcloud,
                                      const float *grid, const float *ptcloud,
                                      int b, int scale, float *grad_grid);
void oracle_cubic_forward(const float *ptcloud, const float *feat, int b,
                          int npts, int c, int scale, int ns, float *out,
                          int *idx);
void oracle_cubic_backward(const float *grad_out, const int *idx, int b,
                           int npts, int c, int scale, int ns,
                           float *grad_feat);

#ifdef __cplusplus
}
#endif
#endif
