/*
 * nesie_ops.h -- C ABI of libnesie_hip.so (MI355X / gfx950).
 *
 * These are the entry points the reference's pybind11 extension modules expose
 * for the VoteNet/Nesie hot path.  Each one replaces a
 * `*_wrapper` / `forward` / `backward` function of the reference; the cited
 * file:line is the reference interface it stands in for.  Conventions follow
 * the reference's: all buffers are device pointers to contiguous memory owned
 * by the caller, outputs are pre-allocated by the caller and written in place,
 * dimensions are passed redundantly as ints, and the launch is asynchronous on
 * the HIP stream given as the last argument (`hipStream_t` passed as void*;
 * NULL = the null stream).  Differences from the reference, all deliberate:
 *   - no torch types anywhere in the signatures;
 *   - every function returns 0 on success or a non-zero nesie_status, and
 *     nesie_last_error() gives a message, instead of fprintf+exit(-1)
 *     (e.g. reference group_points_cuda.cu:48-53);
 *   - every launch goes to the caller's stream (the reference launches
 *     sort_vertices and points_in_boxes on the legacy default stream).
 */
#ifndef NESIE_OPS_H_
#define NESIE_OPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NESIE_OK = 0,
  NESIE_ERR_INVALID_ARG = 1, /* null pointer / negative or inconsistent size  */
  NESIE_ERR_UNSUPPORTED = 2, /* size outside what the kernels are built for   */
  NESIE_ERR_LAUNCH = 3       /* HIP reported a launch error (see last_error)  */
} nesie_status;

/* Library identity and diagnostics. */
int nesie_abi_version(void);
const char *nesie_last_error(void); /* thread-local, never NULL */

/* Form of the squared distance in furthest_point_sampling / ball_query / three_nn / grid_taps
 * (process-wide; default 0).  The reference writes
 *   (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)
 * (furthest_point_sample_cuda.cu:65-66, ball_query_cuda.cu:41-42, three_nn_cuda.cu:41) and nvcc's
 * default -fmad=true may contract it; no output of that build exists to decide.
 *   0  ((dx*dx) + (dy*dy)) + (dz*dz), no contraction (the tuned kernels)
 *   1  fma(dz, dz, fma(dx, dx, dy*dy))      2  fma(dz, dz, fma(dy, dy, dx*dx))
 * Forms 1 / 2 run on the plain kernels (no bucket pruning, no spatial index; nesie_fps_leaves_index
 * then returns 0): a one-flag check against a CUDA-produced fixture, not a fast path.  The CPU
 * oracle has the same switch (oracle_set_distance_form). */
int nesie_set_distance_form(int form);
int nesie_get_distance_form(void);
/* CUs the persistent grids of the layer / weight-gradient kernels are sized for (default 256 = the
 * chip).  A caller that runs other long-lived work beside them -- the next batch's furthest point
 * sampling holds one CU per scene and XCD for milliseconds (furthest_point_sample_cuda.cu's one block
 * per batch element) -- sizes them for the CUs that are left, so that no workgroup of a one-round
 * grid waits for a CU that will not come free.  n % 8 == 0 (one XCD-balanced grid). */
int nesie_set_cu_count(int n);
int nesie_get_cu_count(void);

/* mmdet3d/ops/furthest_point_sample/src/furthest_point_sample.cpp:32-58
 * furthest_point_sampling_wrapper(b, n, m, points[B,N,3], temp[B,N], idx[B,M]).
 * temp must hold 1e10 on entry (furthest_point_sample.py:30) and holds the
 * final running-min squared distances on return.  idx[:,0] = 0. */
int nesie_furthest_point_sampling_wrapper(int b, int n, int m, const float *xyz,
                                          float *temp, int *idx, void *stream);

/* Same operator with caller-provided scratch (no reference counterpart: the
 * reference kernel needs none).  For 4096 < n <= 65536 the kernel spatially sorts
 * the scene into `workspace` and prunes whole 64-point buckets each round (results
 * bit-identical); nesie_fps_workspace_bytes() says how much it needs (0 = this
 * size runs without scratch).  workspace must be 16-byte aligned.  With a NULL or
 * short workspace this behaves exactly like nesie_furthest_point_sampling_wrapper. */
size_t nesie_fps_workspace_bytes(int b, int n);
int nesie_furthest_point_sampling_ws(int b, int n, int m, const float *xyz, float *temp,
                                     int *idx, void *workspace, size_t workspace_bytes,
                                     void *stream);

/* furthest_point_sample.cpp:59-65  furthest_point_sampling_with_dist_wrapper
 * (b, n, m, dist[B,N,N], temp[B,N], idx[B,M]). */
int nesie_furthest_point_sampling_with_dist_wrapper(int b, int n, int m,
                                                    const float *dist,
                                                    float *temp, int *idx,
                                                    void *stream);

/* mmdet3d/ops/ball_query/src/ball_query.cpp:30-47  ball_query_wrapper
 * (b, n, m, min_radius, max_radius, nsample, new_xyz[B,M,3], xyz[B,N,3],
 *  idx[B,M,nsample]).  idx must be zero on entry (ball_query.py:35). */
int nesie_ball_query_wrapper(int b, int n, int m, float min_radius,
                             float max_radius, int nsample,
                             const float *new_xyz, const float *xyz, int *idx,
                             void *stream);

/* ball_query_wrapper over the spatial index nesie_furthest_point_sampling_ws leaves in its
 * workspace (no reference counterpart; same results as nesie_ball_query_wrapper).  For sizes
 * where nesie_fps_leaves_index(b, n) != 0 the workspace holds, after the call returns, the scene
 * sorted into 64-point buckets with one bounding box each; a ball query of the SAME xyz then
 * visits only the buckets within max_radius.  nsample <= 64; NESIE_ERR_UNSUPPORTED otherwise. */
int nesie_fps_leaves_index(int b, int n);
int nesie_ball_query_indexed(int b, int n, int m, float min_radius, float max_radius,
                             int nsample, const float *new_xyz, const void *fps_workspace,
                             size_t workspace_bytes, int *idx, void *stream);

/* mmdet3d/ops/group_points/src/group_points.cpp:31-45  forward
 * (b, c, n, npoints, nsample, points[B,C,N], idx[B,M,ns], out[B,C,M,ns]). */
int nesie_group_points_forward(int b, int c, int n, int npoints, int nsample,
                               const float *points, const int *idx, float *out,
                               void *stream);

/* group_points.cpp:47-62  backward
 * (b, c, n, npoints, nsample, grad_out[B,C,M,ns], idx, grad_points[B,C,N]).
 * grad_points must be zero on entry (group_points.py:218). */
int nesie_group_points_backward(int b, int c, int n, int npoints, int nsample,
                                const float *grad_out, const int *idx,
                                float *grad_points, void *stream);

/* QueryAndGroup's data movement in one pass (mmdet3d/ops/group_points/group_points.py:100-128
 * transposes the points, groups xyz and features separately, subtracts the centre, divides by
 * the radius and concatenates): out[B, 3+C, M, ns] = cat[(xyz[idx] - centre) / radius,
 * features[idx]]; xyz (B,N,3), centres (B,M,3), features (B,C,N) (c = 0: xyz only),
 * radius <= 0: no division.  backward: grad_features (B,C,N, zeroed) += channels 3.. of
 * grad_out (B, 3+C, M, ns), read in place (no slice copy). */
int nesie_query_and_group_forward(int b, int c, int n, int npoints, int nsample,
                                  const float *xyz, const float *centres,
                                  const float *features, const int *idx, float radius,
                                  float *out, void *stream);
int nesie_query_and_group_backward(int b, int c, int n, int npoints, int nsample,
                                   const float *grad_out, const int *idx, float *grad_features,
                                   void *stream);
/* Same result through an inverted index of idx: order[B, M*ns] = the grouped columns sorted
 * by their source point, sources[B, M*ns] = that point for each (nesie_inverted_index builds
 * both; any n: n <= 2048 by a stable counting sort in one workgroup per scene, larger n in windows
 * of 8192 source points per workgroup).  Lanes own sorted entries, a segmented scan inside each wave sums the runs;
 * every run is summed by exactly ONE wave (the one that holds its first entry follows it through
 * the next chunks) and written by one lane: no float atomics; grad_features is WRITTEN in full
 * (round 5: a point without entries gets its zero here, the caller need not clear it).  Inside a run the columns are in ascending order (scratch[B, M*ns] holds the
 * arrival-order placement that the second pass ranks), so the sums are bitwise reproducible. */
int nesie_inverted_index(int b, int n, long long e_total, const int *idx, int *order,
                         int *sources, int *scratch, void *stream);
/* nesie_group_points_backward / nesie_gather_points_grad_wrapper (nsample = 1) through the same
 * index: grad_points (B,C,N) = scatter of grad_out (B,C,npoints,nsample) (WRITTEN in full: the caller
 * need not clear it), every point's run summed by
 * one wave in ascending column order (replaces the atomicAdd of group_points_cuda.cu:10-31 and
 * gather_points_cuda.cu:51-70 with a defined order). */
int nesie_group_points_backward_csr(int b, int c, int n, int npoints, int nsample,
                                    const float *grad_out, const int *order, const int *sources,
                                    float *grad_points, void *stream);
int nesie_query_and_group_backward_csr(int b, int c, int n, int npoints, int nsample,
                                       const float *grad_out, const int *order,
                                       const int *sources, float *grad_features, void *stream);

/* The tail of VoteModule.forward behind its last convolution (model_utils/vote_module.py:106-147
 * with vote_per_seed = 1, with_res_feat, no vote_xyz_range): raw (B, 3+c, N) = [offset, residual]
 * -> vote_points (B, N, 3) = seed_points + offset^T, vote_feats (B, c, N) = (seed_feats + residual)
 * divided by its channel-wise L2 norm (normalise != 0), inv_norm (B, N) kept for the backward.
 * Backward: g_feats (B, c, N) / g_points (B, N, 3), either may be NULL -> d_raw (B, 3+c, N) (rows
 * 0..2 = g_points^T, rows 3.. = the gradient of seed_feats + residual). */
int nesie_vote_finish_forward(int b, int c, int n, int normalise, const float *raw,
                              const float *seed_points, const float *seed_feats, float *vote_points,
                              float *vote_feats, float *inv_norm, void *stream);
int nesie_vote_finish_backward(int b, int c, int n, int normalise, const float *g_feats,
                               const float *g_points, const float *vote_feats, const float *inv_norm,
                               float *d_raw, void *stream);

/* QueryAndGroup over NETWORK-COMPUTED coordinates (vote aggregation groups the predicted votes
 * around centres sampled from them, nesie_head.py:243 -> point_sa_module.py:122-131,
 * group_points.py:98-110): the sampled centres straight from the (B, N, 3) array, and the
 * coordinate gradient -- what autograd assembles from the backward of transpose / gather_points /
 * group_points (two atomicAdd scatters) / sub / div / cat:
 *   d_xyz[p] = (1/r) sum_{idx[e] = p} g[0:3, e] - (1/r) sum_{sample[m] = p} sum_s g[0:3, m, s]
 *              + sum_{sample[m] = p} d_centres[m]          (r = radius, 0: no division)
 * with grad_out (B, 3+c, npoints, nsample), (order, sources) = nesie_inverted_index(idx),
 * sample (B, npoints) the centres' indices, d_centres (B, npoints, 3) or NULL; d_xyz (B, N, 3) is
 * written (not accumulated), every point by one thread in a fixed order. */
int nesie_gather_rows3(int b, int n, int m, const float *xyz, const int *sample, float *centres,
                       void *stream);
int nesie_query_and_group_backward_xyz(int b, int c, int n, int npoints, int nsample, float radius,
                                       const float *grad_out, const int *order,
                                       const int *sources, const int *sample,
                                       const float *d_centres, float *d_xyz, void *stream);

/* three_interpolate_grad_wrapper through an inverted index of idx[B, n, 3] over the m known
 * points (nesie_inverted_index with e_total = 3n): grad_points[B, C, m] = scatter of
 * weight * grad_out, WRITTEN in full, no float atomics, fixed order; grad_out (B, C, n) with a free
 * batch stride (a channel slice of a wider gradient: no copy). */
int nesie_three_interpolate_grad_csr(int b, int c, int n, int m, const float *grad_out,
                                     long long grad_out_bstride, const float *weight, const int *order,
                                     const int *sources, float *grad_points, void *stream);

/* mmdet3d/ops/gather_points/src/gather_points.cpp:28-42  gather_points_wrapper
 * (b, c, n, npoints, points[B,C,N], idx[B,M], out[B,C,M]). */
int nesie_gather_points_wrapper(int b, int c, int n, int npoints,
                                const float *points, const int *idx, float *out,
                                void *stream);

/* gather_points.cpp:44-59  gather_points_grad_wrapper
 * (b, c, n, npoints, grad_out[B,C,M], idx, grad_points[B,C,N] zeroed). */
int nesie_gather_points_grad_wrapper(int b, int c, int n, int npoints,
                                     const float *grad_out, const int *idx,
                                     float *grad_points, void *stream);

/* mmdet3d/ops/interpolate/src/interpolate.cpp:46-58  three_nn_wrapper
 * (b, n, m, unknown[B,n,3], known[B,m,3], dist2[B,n,3], idx[B,n,3]).
 * Also stands in for mmcv.ops.three_nn (side_pooling_module.py:204). */
int nesie_three_nn_wrapper(int b, int n, int m, const float *unknown,
                           const float *known, float *dist2, int *idx,
                           void *stream);

/* interpolate.cpp:60-75  three_interpolate_wrapper
 * (b, c, m, n, points[B,C,M], idx[B,N,3], weight[B,N,3], out[B,C,N]). */
int nesie_three_interpolate_wrapper(int b, int c, int m, int n,
                                    const float *points, const int *idx,
                                    const float *weight, float *out,
                                    void *stream);

/* Grid points of the side-aware quality head and their 3-NN taps among the seeds, one launch
 * per grid set (dense_heads/side_pooling_module.py:87-157 builds the box-frame grid, selects
 * the faces, rotates and translates it; :204-225 finds the 3 nearest seeds and the
 * inverse-distance weights).  centre/size (B,K,3), heading (B,K), mult/plane (gp,3) = box-frame
 * multipliers of the gp grid points of a proposal and the +-10 % plane factors of the SAQE
 * variant (quelity_estimation_module.py:142-167; zeros otherwise), known (B,m,3) ->
 * idx (B,K*gp,3) i32 (three_nn_wrapper's order), weight (B,K*gp,3), rel (B,K*gp,3) = grid point
 * relative to the proposal centre. */
int nesie_grid_taps(int b, int kprop, int gp, int m, const float *centre, const float *size,
                    const float *heading, const float *mult, const float *plane,
                    const float *known, int *idx, float *weight, float *rel, void *stream);

/* The quality head's grid features in their consumer's layout, optionally folded with the
 * first 1x1 conv of the MiniPointNet that consumes them.
 * Reference: dense_heads/side_pooling_module.py:226-243 builds cat([rel_xyz, interpolated]),
 * :304-313 splits it per face and makes each face contiguous, :346-349 feeds it to
 * Conv2d(3+C, H, 1, bias=False).  The n = K*segs*seg_len queries are ordered (proposal k,
 * face s, grid point g); out is (B, segs, c_total, K*seg_len) and query (k,s,g), channel ch
 * lands at out[b, s, c_offset+ch, k*seg_len+g]:
 *     out = w0*T[j0] + w1*T[j1] + w2*T[j2]  (+ wx[s][ch] . rel[q])
 * with T = table[b, j, s*seg_off + ch], a POINT-major table of row pitch `pitch` floats.
 *   - table = seed features (B, M, C), seg_off = 0, rel = wx = NULL, c_offset = 3:
 *     three_interpolate_wrapper's values (bit-identical) in the per-face layout;
 *   - table = F . W_f^T (seed features times the feature columns of each face's first conv,
 *     (B, M, segs*H), seg_off = H) and wx = the conv's xyz columns (segs, H, 3): the first
 *     conv's OUTPUT, W . cat[rel, blend(F)] = W_xyz . rel + blend(W_f . F) by linearity.
 * backward (second form): dy (B, segs, c, K*seg_len) = the gradient of out; ADDS into
 * d_table (B, M, pitch) columns [s*seg_off, +c) (zeroed by the caller) and WRITES
 * d_wx (B, R, segs, c, 3), R = nesie_blend_conv_runs(n, segs), one partial sum of dy x rel per
 * workgroup for the caller to add up (NULL = skip); c in
 * {64, 128, 192, 256} and K*seg_len % 64 == 0.  Sum order is not fixed (float atomics), like
 * the reference's scatter.  Channels outside [c_offset, c_offset+c) of out are left
 * untouched by the forward. */
int nesie_blend_conv_forward(int b, int c, int m, int n, const float *table, int pitch,
                             int seg_off, const int *idx, const float *weight,
                             const float *rel, const float *wx, float *out, int segs,
                             int seg_len, int c_total, int c_offset,
                             float *stat_partial /* NULL, or [segs*c][B*K*seg_len/64][2]: (sum,
                             sum of squares) per 64-query tile for nesie_bn_relu_forward's
                             pre_partial */, void *stream);
int nesie_blend_conv_runs(int n, int segs);
int nesie_blend_conv_backward(int b, int c, int m, int n, const float *dy, int pitch, int seg_off, const int *idx, const float *weight,
                              const float *rel, float *d_table, float *d_wx, int segs,
                              int seg_len, void *stream);
/* The same backward when the blend's output Z (B, segs, c, n/segs) is followed by a training
 * BatchNorm + ReLU (MiniPointNet.first_conv[1:3], side_pooling_module.py:346-348): da is the
 * gradient of relu(bn(Z)) and the norm backward's apply pass runs on the tile load with
 * bnb [segs*c][8] from nesie_pw_bnb_coef (the partial sums come from the launch that produced da,
 * nesie_pw_dgrad_bn_reduce) -- the pass that read (da, Z) and wrote dZ disappears. */
int nesie_blend_conv_backward_bn(int b, int c, int m, int n, const float *da, const float *z,
                                 const float *bnb, int pitch, int seg_off, const int *idx,
                                 const float *weight, const float *rel, float *d_table,
                                 float *d_wx, int segs, int seg_len, void *stream);

/* The two backward entry points above WITHOUT float atomics (round 4; the default of the python
 * layer since round 5): every 16-query group stores the row it has summed per seed into staging
 * slots of its own (48 per group), a counting sort per chunk of 128 groups lists each seed's slots
 * in ascending order (chunk-major, with the run starts), and one wave per (scene, face, seed) walks
 * the chunks, adds its rows in that order and WRITES the d_table row -- bitwise reproducible,
 * d_table needs no zero fill.  Needs m <= 2047 seeds, pitch and seg_off multiples of 4, d_table
 * 16-byte aligned.  z / bnb both NULL: plain
 * backward; both given: the norm backward on the tile load (as nesie_blend_conv_backward_bn).
 * d_wx as above.  workspace: nesie_blend_conv_backward_workspace_bytes(b, c, n, segs) bytes. */
size_t nesie_blend_conv_backward_workspace_bytes(int b, int c, int n, int segs);
int nesie_blend_conv_backward_staged(int b, int c, int m, int n, const float *dy, const float *z,
                                     const float *bnb, int pitch, int seg_off, const int *idx,
                                     const float *weight, const float *rel, float *d_table,
                                     float *d_wx, int segs, int seg_len, void *workspace,
                                     size_t workspace_bytes, void *stream);
/* out[g * c + ch] = sum over the batch entries n = g (mod ng) and the p positions of x[n][ch][pos];
 * x (nb, c, p) with batch stride x_bstride >= c * p (a channel slice of a wider tensor is fine).
 * The bias gradient of a convolution without a norm behind it (autograd's grad.sum((0, 2)) for the
 * nn.Conv1d outputs of reliable_conv_bbox_module.py:144-177, vote_module.py:75-79,
 * side_pooling_module.py:55-78; ng = the S stacked nets of the quality head); one workgroup per
 * (group, channel), fixed summation order. */
int nesie_channel_sum(int nb, int ng, int c, long long p, const float *x, long long x_bstride,
                      float *out, void *stream);

/* bnb[ch] = (scale, shift, a, mean, d1, e0, -, -) of a BatchNorm + ReLU backward from the partial
 * sums part [(channels) * nslots * 2] = (sum g, sum g zhat), the folded forward coefficients
 * z_coef [channels][4] and gamma; count = elements per channel; dgamma / dbeta written. */
int nesie_pw_bnb_coef(int channels, int nslots, double count, const float *part,
                      const float *z_coef, const float *gamma, float *bnb, float *dgamma,
                      float *dbeta, void *stream);

/* nesie_blend_conv_forward (second form) followed by the training BatchNorm + ReLU of
 * MiniPointNet.first_conv[1:3] (side_pooling_module.py:346-348), fused by recomputation: the
 * conv output and its gradient are never stored (4 tensor passes instead of 11).
 *   forward : out (B, segs, c, K*seg_len) = relu(gamma * (c0 - mean) * invstd + beta) with the
 *             statistics of c0 per stacked channel (segs*c of them; gamma, beta, running_*,
 *             save_* are [segs*c]); fwd_coef [segs*c][4] = (scale, bias, mean, invstd).
 *   backward: from dy (same shape as out): d_table (zeroed by the caller, ADDED into),
 *             d_wx_part (B, R, segs, c, 3) partial sums (R = nesie_blend_conv_runs), dgamma,
 *             dbeta [segs*c].
 * workspace = nesie_blend_conv_bn_workspace_bytes(b, c, n, segs); c in {64,128,192,256},
 * K*seg_len % 64 == 0. */
size_t nesie_blend_conv_bn_workspace_bytes(int b, int c, int n, int segs);
int nesie_blend_conv_bn_forward(int b, int c, int m, int n, const float *table, int pitch,
                                int seg_off, const int *idx, const float *weight,
                                const float *rel, const float *wx, const float *gamma,
                                const float *beta, float *running_mean, float *running_var,
                                float momentum, float eps, float *out, float *save_mean,
                                float *save_invstd, float *fwd_coef, void *workspace,
                                size_t workspace_bytes, int segs, int seg_len, void *stream);
int nesie_blend_conv_bn_backward(int b, int c, int m, int n, const float *dy,
                                 const float *table, int pitch, int seg_off, const int *idx,
                                 const float *weight, const float *rel, const float *wx,
                                 const float *gamma, const float *save_invstd,
                                 const float *fwd_coef, float *d_table, float *d_wx_part,
                                 float *dgamma, float *dbeta, void *workspace,
                                 size_t workspace_bytes, int segs, int seg_len, void *stream);

/* interpolate.cpp:77-93  three_interpolate_grad_wrapper
 * (b, c, n, m, grad_out[B,C,N], idx, weight, grad_points[B,C,M] zeroed). */
int nesie_three_interpolate_grad_wrapper(int b, int c, int n, int m,
                                         const float *grad_out, const int *idx,
                                         const float *weight,
                                         float *grad_points, void *stream);

/* mmdet3d/ops/rotated_iou/cuda_op/sort_vert.cpp:6-30  sort_vertices_forward
 * (vertices[B,N,M,2] f32, mask[B,N,M] bool(1 byte), num_valid[B,N] i32)
 * -> idx[B,N,9] i32.  The reference allocates idx; here the caller does. */
int nesie_sort_vertices_forward(int b, int n, int m, const float *vertices,
                                const uint8_t *mask, const int *num_valid,
                                int *idx, void *stream);

/* mmdet3d/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:46-48 /
 * points_in_boxes_cuda.cu:153-181  points_in_boxes_batch
 * (boxes[B,T,7] LiDAR frame bottom-centre, pts[B,M,3], out[B,M,T] i32 zeroed). */
int nesie_points_in_boxes_batch(int b, int boxes_num, int pts_num,
                                const float *boxes, const float *pts, int *out,
                                void *stream);

/* Max over the neighbourhood axis of a grouped tensor: x[rows, nsample] -> out[rows],
 * argmax[rows] (one byte, smallest index on ties).  No extension entry in the reference:
 * it calls ATen's F.max_pool2d(kernel=[1, nsample]) (point_sa_module.py:149-150) and
 * torch.max(dim=-1) (side_pooling_module.py:361,368); same values, same tie rule.
 * nsample must be a power of two in 4..64; x / grad_x 16-byte aligned. */
int nesie_group_max_pool_forward(long long rows, int nsample, const float *x, float *out,
                                 uint8_t *argmax, void *stream);
int nesie_group_max_pool_backward(long long rows, int nsample, const float *grad_out,
                                  const uint8_t *argmax, float *grad_x, void *stream);
/* grad_x[row, argmax[row]] += grad_out[row] into an existing dense gradient (a tensor that
 * feeds both the max and another consumer: MiniPointNet's f, side_pooling_module.py:359-365). */
int nesie_group_max_pool_backward_add(long long rows, int nsample, const float *grad_out,
                                      const uint8_t *argmax, float *grad_x, void *stream);

/* Same-class LHS-NMS of the teacher's pseudo boxes, on the device: boxes[B,K,8] f32 =
 * axis-aligned (x1,y1,z1,x2,y2,z2,score,class), K <= 64; keep[B,K] u8 = 1 for every box
 * the reference's lhs_3d_faster_samecls returns in `pick`
 * (models/detectors/votenet_nesie.py:733-779; the reference runs it in numpy on the host). */
int nesie_lhs_nms_samecls(int b, int k, const float *boxes, float thr, uint8_t *keep,
                          void *stream);

/* Rotated 3-D IoU of n box pairs (x, y, z, dx, dy, dz, yaw; rotation about z only) with the
 * Jacobian w.r.t. the 7 parameters of box1 (jac may be NULL: value only).  One kernel for
 * the reference's torch chain cal_iou_3d -> cal_iou -> oriented_box_intersection_2d ->
 * sort_v (rotated_iou/oriented_iou_loss.py:86-109, box_intersection_2d.py:13-184), same
 * formulas, masks and vertex order; box2 is treated as a constant. */
int nesie_iou3d_forward(int n, const float *box1, const float *box2, float *iou, float *jac,
                        void *stream);

/* Training-mode BatchNorm with an optional fused ReLU over x[B, C, P] (statistics per
 * channel over B*P; biased variance for normalisation, unbiased for running_var, as
 * torch.nn.BatchNorm{1,2}d).  No extension entry in the reference: it builds
 * conv -> BN -> ReLU from mmcv ConvModule / nn.BatchNorm + nn.ReLU
 * (point_sa_module.py:277-289, side_pooling_module.py:55-78,346-358).
 * forward : y = relu?(gamma * (x - mean) * invstd + beta); writes save_mean/save_invstd
 *           [C] and updates running_mean/var [C] in place (NULL = skip) with `momentum`.
 * backward: dx, dgamma [C], dbeta [C] from dy, x, y (y only read when relu != 0) and the
 *           forward's fwd_coef [C,4] = (scale, bias, mean, invstd).
 * row_bias (NULL = none): the normalised input is x[b,c,i] + row_bias[b,c,i/group] with
 *           row_bias[B, C, P/group] -- the per-proposal half of side_pooling_module.py:359-368's
 *           second_conv input (the max-pooled global feature repeated over the `group` grid
 *           points of a proposal), added on the fly instead of materialising the repeat+concat;
 *           group must be a power of two in 4..256.  backward writes d_row_bias (same shape)
 *           = sum of dx over each group.
 *           group == 0: row_bias[C] is ONE value per channel -- the bias of the convolution in
 *           front (mmcv ConvModule with bias=True and a norm: vote_module.py / base_conv_bbox_head),
 *           added in registers with the rounding of the separate add.  Its gradient is identically
 *           zero (the mean subtraction removes it): d_row_bias must be NULL.
 * workspace: nesie_bn_workspace_bytes(b, c, p) bytes, 16-byte aligned tensors. */
size_t nesie_bn_workspace_bytes(int b, int c, long long p);
int nesie_bn_relu_forward(int b, int c, long long p, const float *x, const float *gamma,
                          const float *beta, float *running_mean, float *running_var,
                          float momentum, float eps, int relu, float *y, float *save_mean,
                          float *save_invstd, float *fwd_coef /* [C,4] out */,
                          const float *row_bias, int group,
                          const float *pre_partial /* NULL, or unshifted (sum, sum^2) partials
                          [C][pre_nslice][2] left by the producer of x: the statistics pass over
                          x is skipped */, int pre_nslice, void *workspace,
                          size_t workspace_bytes, void *stream);
/* y == NULL with relu != 0: the normalised tensor was never stored (nesie_pw_layer_forward applied
 * it on its operand load); the mask is fma(x, scale, bias) > 0 from fwd_coef.  d_row_bias without
 * row_bias: the per-group sums of dx (the producer had added the row term to x itself). */
int nesie_bn_relu_backward(int b, int c, long long p, const float *dy, const float *x,
                           const float *y, const float *gamma, const float *beta,
                           const float *save_mean, const float *save_invstd,
                           const float *fwd_coef /* from the forward */, int relu, float *dx,
                           float *dgamma, float *dbeta, const float *row_bias, int group,
                           float *d_row_bias, void *workspace, size_t workspace_bytes,
                           void *stream);

/* The apply pass of nesie_bn_relu_backward alone (relu, y == NULL form), for a producer that has
 * already left the reduction's partials: partial[(ch * nslice + i) * 2 + {0, 1}] = sum(g),
 * sum(g * xhat) over slice i (nesie_pw_dgrad_bn_reduce).  dgamma / dbeta are written.
 * save_invstd == NULL: column 3 of fwd_coef (also accepted by nesie_bn_relu_maxpool_backward). */
int nesie_bn_relu_backward_apply(int b, int c, long long p, const float *dy, const float *x,
                                 const float *gamma, const float *save_invstd,
                                 const float *fwd_coef, const float *partial, int nslice,
                                 float *dx, float *dgamma, float *dbeta, int group,
                                 float *d_row_bias, void *stream);

/* Training BatchNorm + ReLU + max over the neighbourhood axis for the LAST layer of a
 * set-abstraction MLP: x[B, C, M, ns] -> pooled[B, C, M] (+ argmax[B, C, M], one byte, smallest
 * index on ties) without writing the normalised tensor.  Reference: ConvModule's BN2d + ReLU
 * followed by F.max_pool2d(kernel=[1, ns]) (point_sa_module.py:277-289, 136-158); same
 * values as nesie_bn_relu_forward followed by nesie_group_max_pool_forward.
 * backward: dx[B, C, M, ns], dgamma, dbeta from grad_pooled[B, C, M], argmax, x, pooled and the
 * forward's fwd_coef.  ns a power of two in 4..64; workspace =
 * nesie_bn_workspace_bytes(b, c, m * ns). */
int nesie_bn_relu_maxpool_forward(int b, int c, int m, int ns, const float *x,
                                  const float *gamma, const float *beta, float *running_mean,
                                  float *running_var, float momentum, float eps, float *pooled,
                                  uint8_t *argmax, float *save_mean, float *save_invstd,
                                  float *fwd_coef, void *workspace, size_t workspace_bytes,
                                  void *stream);
int nesie_bn_relu_maxpool_backward(int b, int c, int m, int ns, const float *grad_pooled,
                                   const uint8_t *argmax, const float *x, const float *pooled,
                                   const float *gamma, const float *save_invstd,
                                   const float *fwd_coef, float *dx, float *dgamma,
                                   float *dbeta, void *workspace, size_t workspace_bytes,
                                   void *stream);

/* NesieHead.side2box + Integral + the bbox_probs softmax (dense_heads/nesie_head.py:19-52,
 * 150-209, 255-257), one thread per proposal.  reg (B, 6*bins+2, K) channel-major = the
 * regression branch's output (6 sides x bins logits, then the (sin, cos)-like heading pair);
 * agg (B, K, 3) aggregated points; scale / sign [6].
 *   forward : probs (B, 6, bins, K) = per-side softmax; surface (B, K, 6) = agg +- E[bin] * scale;
 *             bbox (B, K, 7) = ((lo+hi)/2, hi-lo, atan2(h0/|h|, h1/|h|)).
 *   backward: d_reg (B, 6*bins+2, K), d_agg (B, K, 3) from d_surface, d_bbox (either may be NULL).
 * bins <= 33. */
int nesie_side_decode_forward(int b, int k, int bins, const float *reg, const float *agg,
                              const float *scale, const float *sign, float *probs,
                              float *surface, float *bbox, void *stream);
int nesie_side_decode_backward(int b, int k, int bins, const float *reg, const float *probs,
                               const float *scale, const float *sign, const float *d_surface,
                               const float *d_bbox, float *d_reg, float *d_agg, void *stream);

/* Evaluation-mode BatchNorm (+ ReLU) (+ max over the neighbourhood axis): the running statistics
 * are folded by the caller into coef [C,4] = (scale, bias, -, -), scale = gamma / sqrt(var + eps),
 * bias = beta - mean * scale (what torch.nn.BatchNorm{1,2}d.eval() + nn.ReLU compute for the
 * ConvModules of point_sa_module.py:277-289 / side_pooling_module.py:346-358 at test time).
 * row_bias / group as in nesie_bn_relu_forward. */
/* coef [C,4] of an evaluation-mode BatchNorm from its parameters and running statistics (gamma /
 * beta may be NULL); one launch, recomputed per call (torch.nn.BatchNorm.eval() folds the same
 * numbers inside its kernel on every call). */
int nesie_bn_eval_coef(int c, const float *gamma, const float *beta, const float *running_mean,
                       const float *running_var, float eps, float *coef, void *stream);
int nesie_affine_relu_forward(int b, int c, long long p, const float *x, const float *coef,
                              int relu, const float *row_bias, int group, float *y,
                              void *stream);
int nesie_affine_relu_maxpool_forward(int b, int c, int m, int ns, const float *x,
                                      const float *coef, float *pooled, uint8_t *argmax,
                                      void *stream);

/* ---- inference post-processing and evaluation geometry -------------- */

/* aligned_3d_nms (core/post_processing/box3d_nms.py:129-176) for B scenes at once.
 * boxes (B,K,6) f32 axis-aligned (x1,y1,z1,x2,y2,z2), scores (B,K) f32, classes (B,K) i32,
 * valid (B,K) u8 or NULL: only boxes with valid != 0 take part (the reference indexes them
 * out with a boolean mask first, nesie_head.py:752-755).  picks (B,K) i32 = input positions
 * of the kept boxes in pick order (descending score), padded with -1; count (B) i32.
 * fp32 arithmetic in the reference's order; equal scores order by index (the later one is
 * picked first, as a stable ascending argsort read from its end).  K <= 512. */
int nesie_aligned_3d_nms(int b, int k, const float *boxes, const float *scores,
                         const int *classes, const uint8_t *valid, float thr, int *picks,
                         int *count, void *stream);

/* counts[b, t] = number of points of scene b inside box t: the column sums of
 * points_in_boxes_batch (same test, same LiDAR-frame operands) that
 * NesieHead.multiclass_nms_single takes for its non-empty mask
 * (`box_indices.T.sum(1) > 5`, nesie_head.py:744-750), without the (M, T) table. */
int nesie_points_in_boxes_count(int b, int boxes_num, int pts_num, const float *boxes,
                                const float *pts, int *counts, void *stream);

/* iou3d_cuda.boxes_overlap_bev_gpu (ops/iou3d/src/iou3d.cpp:66-90, iou3d_kernel.cu:127-264):
 * overlap AREA of every pair of rotated BEV rectangles (x1, y1, x2, y2, angle);
 * boxes_a (N,5), boxes_b (M,5) -> ans_overlap (N,M).  BaseInstance3DBoxes.overlaps
 * (base_box3d.py:387-438) multiplies it by the height overlap for the 3-D IoU of indoor_eval. */
int nesie_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                            float *ans_overlap, void *stream);

/* ---- input side ---------------------------------------------------------- */

/* One training batch from HBM-resident scenes: the reference's per-sample CPU pipeline
 * IndoorPointSample -> RandomFlip3D -> GlobalRotScaleTrans after GlobalAlignment
 * (datasets/pipelines/transforms_3d.py:821-891, 59-162, 497-648, 410-488) in one gather pass.
 *   pool    (pool_rows, 3) f32  raw xyz of all resident scenes, back to back
 *   height  (pool_rows)    f32  shifted height z - percentile(z, 0.99) (loading.py:424-430)
 *   choices (B, n)         i32  sampled rows of the pool (np.random.choice + scene offset)
 *   xform   (B, 20)        f32  per scene: axis-align R[9] row-major, t[3]; flip_x, flip_y
 *                               (-1 = flipped, +1 = not); cos, sin of the rotation noise;
 *                               scale; trans[3]
 *   out     (B, n, 4)      f32  (x, y, z, height) as DefaultFormatBundle3D hands them on
 * out = ((flip(p R^T + t)) [[c, s, 0], [-s, c, 0], [0, 0, 1]]) * scale + trans; height * scale. */
int nesie_scene_assemble(int b, int n, long long pool_rows, const float *pool,
                         const float *height, const int *choices, const float *xform,
                         float *out, void *stream);

/* One pointwise (1x1) convolution layer of a grouped MLP on the fp32 matrix cores, with the
 * previous layer's BatchNorm + ReLU folded into the operand load and this layer's statistics /
 * pooling tail folded into the epilogue.  No extension entry in the reference: it evaluates mmcv
 * ConvModule(Conv2d 1x1 -> BN2d -> ReLU) and the pooling op by op (point_sa_module.py:277-289,
 * 136-158; side_pooling_module.py:343-370).
 *   y[n] = W[n % ng] . act(x[n]) (+ row_bias) (+ bias)       n < nb, nb % ng == 0
 *   x[n] (k, p) at x + n*x_bstride; y[n] (cout, p) at y + n*y_bstride (y NULL: not stored)
 *   W[g][m][kk] = w[g*w_gstride + m*w_rstride + kk*w_cstride]  (strides: W or its transpose)
 *   act(v) = in_coef ? max(fma(v, in_coef[g*k+kk][0], in_coef[g*k+kk][1]), in_relu ? 0 : -inf) : v
 *   row_bias (nb, cout, p / rb_group) or NULL; bias [ng*cout] or NULL
 *   stat_part (NULL = skip): [ng][nesie_pw_stat_slots(...)][cout][4] = (count, shift,
 *            sum(y - shift), sum((y - shift)^2)) per wave, merged by nesie_pw_stats_finalize
 *   pool_group 0 / 16 / 32: max (and, pool_min != 0, min) over each pool_group consecutive
 *            positions -> pool_*_out (nb, cout, p / pool_group), arg_*_out the position inside it.
 * Built for k <= 260, cout <= 256, p a multiple of the tile (nesie_pw_supported). */
int nesie_pw_supported(int k, int cout, long long p);
int nesie_pw_stat_slots(int nb, int ng, int k, int cout, long long p);
int nesie_pw_layer_forward(int nb, int ng, int k, int cout, long long p, const float *x,
                           long long x_bstride, const float *w, long long w_gstride,
                           int w_rstride, int w_cstride, const float *in_coef, int in_relu,
                           const float *row_bias, int rb_group, const float *bias, float *y,
                           long long y_bstride, float *stat_part, int pool_group, int pool_min,
                           float *pool_max_out, float *pool_min_out, uint8_t *arg_max_out,
                           uint8_t *arg_min_out, void *stream);
/* Input gradient of a layer whose input was relu(bn(Z)): y[n] = W[n % ng] . x[n] (W = the
 * transposed weight view, x = the gradient of the layer's raw output) as nesie_pw_layer_forward
 * computes it, and in the same launch the reduction pass of that BatchNorm's backward
 * (torch.nn.BatchNorm2d backward behind mmcv ConvModule, point_sa_module.py:277-289):
 *   bn_part[(g*cout + m) * nslots + slot][2] = sum(gg), sum(gg * zhat) over the slot's positions,
 *   gg = y [fma(Z, bn_coef[.][0], bn_coef[.][1]) > 0], zhat = (Z - bn_coef[.][2]) * bn_coef[.][3]
 * Z[n] (cout, p) at bn_z + n*bnz_bstride; nslots = nesie_pw_stat_slots(nb, ng, k, cout, p), every
 * slot is written; nesie_bn_relu_backward_apply consumes bn_part with nslice = nslots. */
int nesie_pw_dgrad_bn_reduce(int nb, int ng, int k, int cout, long long p, const float *x,
                             long long x_bstride, const float *w, long long w_gstride,
                             int w_rstride, int w_cstride, float *y, long long y_bstride,
                             const float *bn_z, long long bnz_bstride, const float *bn_coef,
                             float *bn_part, void *stream);
/* stat_part -> coef[ch][4] = (scale, bias, mean, invstd), scale = gamma * invstd, bias = beta -
 * mean * scale (fp64, Chan's merge of the shifted partials), running statistics updated like
 * torch.nn.BatchNorm2d in training mode.  channels = ng * cout (stacked layers).
 * chan_bias [channels] or NULL: the bias of a convolution in FRONT of the norm (mmcv ConvModule
 * with bias=True and a norm, vote_module.py:65-74).  The layer kernel leaves it out -- the mean
 * subtraction cancels it in every normalised value and its gradient is identically zero -- and
 * only the running mean, which tracks the biased output, gets it added here. */
int nesie_pw_stats_finalize(int channels, int cout, int nslots, const float *stat_part,
                            const float *gamma, const float *beta, float *running_mean,
                            float *running_var, float momentum, float eps, float *coef,
                            const float *chan_bias, void *stream);
/* Pooling tail: combine the group / pool_group partial extrema of every group of `group`
 * positions; with coef (this layer's BatchNorm) the value is relu?(scale * ext + bias) of the
 * extremum the sign of the scale selects (= max over the group of the normalised activation).
 * pooled (nb, channels, p / group), argmax the position inside the group (first on ties). */
int nesie_pw_pool_finish(int nb, int ng, int channels, long long p, int group, int pool_group,
                         const float *pmax, const float *pmin, const uint8_t *amax,
                         const uint8_t *amin, const float *coef, int relu, float *pooled,
                         uint8_t *argmax, void *stream);
/* ... also leaving the raw extremum each pooled value came from, zstar (nb, channels, p / group). */
int nesie_pw_pool_finish_z(int nb, int ng, int channels, long long p, int group, int pool_group,
                           const float *pmax, const float *pmin, const uint8_t *amax,
                           const uint8_t *amin, const float *coef, int relu, float *pooled,
                           uint8_t *argmax, float *zstar, void *stream);

/* Backward of a POOLED TAIL -- last 1x1 conv (k -> c, bias-free) + training-mode BatchNorm + ReLU +
 * max over the ns samples of every group (PointSAModule's shared MLP and _pool_features,
 * point_sa_module.py:136-158, 277-289; autograd's max_pool2d / relu / batch_norm / conv2d backward
 * chain in the reference) -- without the dense pre-pool tensor: the forward keeps only pooled,
 * argmax and zstar, and with dZ = ghat + alpha + beta Z (per channel; ghat the pooled gradient at
 * the arg-max, one non-zero per channel and group)
 *   dA = W^T ghat + W^T alpha + (W^T diag(beta) W) A,   dW = sum ghat A^T + alpha s^T + diag(beta) W M,
 *   s = sum A, M = sum A A^T,  A = relu(coef_prev . z_prev) the layer's operand (k, p) per batch element.
 * nesie_pool_tail_supported: 1 when (k, c, p, ns) is built (k = 64, c = 128, ns in {16, 32, 64}).
 * nesie_pool_tail_sizes: out[0] = reduction slots of the input-gradient launch (bn_part [k][out[0]][2]),
 *   out[1] = partials of the weight-gradient pass.
 * nesie_pool_tail_prepare: dgamma, dbeta [c] (written); ab [c][4] = (alpha, beta, gamma invstd, -);
 *   ent [nb][m][c][2] = (masked pooled gradient, arg-max position as int bits);
 *   wcat [k][c + k] = [W^T diag(gamma invstd) | W^T diag(beta) W]; c0 [k] = W^T alpha.
 *   grad_pooled / pooled / zstar / argmax (nb, c, m); coef [c][4] = (scale, bias, mean, invstd) of
 *   this layer's norm; w (c, k) row-major.
 * nesie_pool_tail_dgrad: da (nb, k, p) = gradient of A, and in bn_part the two sums of the
 *   PREVIOUS layer's norm backward (as nesie_pw_dgrad_bn_reduce leaves them).
 * nesie_pool_tail_wgrad: dw (c, k) written; part_m [out[1]][k][k], part_s [out[1]][k],
 *   part_w [out[1]][c][k], ms [k k + k] doubles are scratch. */
int nesie_pool_tail_supported(int k, int c, long long p, int ns);
int nesie_pool_tail_sizes(int nb, int k, int c, long long p, int *out);
int nesie_pool_tail_prepare(int nb, int c, int m, int ns, int k, const float *grad_pooled,
                            const float *pooled, const float *zstar, const uint8_t *argmax,
                            const float *coef, const float *gamma, const float *w, float *dgamma,
                            float *dbeta, float *ab, float *ent, float *wcat, float *c0,
                            void *stream);
int nesie_pool_tail_dgrad(int nb, int k, int c, long long p, int ns, const float *z_prev,
                          long long z_bstride, const float *coef_prev, const float *wcat,
                          const float *c0, const float *ent, float *da, long long da_bstride,
                          float *bn_part, void *stream);
int nesie_pool_tail_wgrad(int nb, int k, int c, long long p, int ns, const float *z_prev,
                          long long z_bstride, const float *coef_prev, const float *ent,
                          const float *ab, const float *w, float *part_m, float *part_s,
                          float *part_w, double *ms, float *dw, void *stream);

/* Weight gradient of the same layers: dw[g][co][ci] = sum over the batches n of group g (n % ng
 * == g) and all positions of dy[n][co][pos] * act(x[n][ci][pos]), act as in
 * nesie_pw_layer_forward (x_coef [ng*ci][4], NULL = identity): the Conv2d weight gradient that
 * autograd computes for ConvModule (point_sa_module.py:277-289) with the normalised activation
 * recomputed on load.  dy[n] (co, p) at dy + n*dy_bstride, x[n] (ci, p) at x + n*x_bstride;
 * dw (ng, co, ci); partials are added in a fixed order (bitwise reproducible).
 * Supported: co <= 256, 9 <= ci <= 1024, p % 32 == 0; a layer wider than one workgroup's
 * accumulators (co <= 128: ci > 320; co > 128: ci > 128) runs as column blocks of dw, one
 * launch each -- or, when the position count is small (nesie_pw_wgrad_tiled: the 1-D chains,
 * 256 x 512 over 8 x 1024 positions), as ONE launch over 64 x 64 blocks of the product, every
 * block with its own runs of positions (split-K with a fixed-order reduction).
 * workspace = nesie_pw_wgrad_workspace_bytes(nb, ng, co, ci, p). */
int nesie_pw_wgrad_supported(int co, int ci, long long p);
int nesie_pw_wgrad_tiled(int nb, int ng, int co, int ci, long long p);
size_t nesie_pw_wgrad_workspace_bytes(int nb, int ng, int co, int ci, long long p);
int nesie_pw_wgrad(int nb, int ng, int co, int ci, long long p, const float *dy,
                   long long dy_bstride, const float *x, long long x_bstride,
                   const float *x_coef, int x_relu, float *dw, void *workspace,
                   size_t workspace_bytes, void *stream);

/* Deferred reductions: nothing in a backward pass reads a weight gradient (its consumers are the
 * optimiser and the gradient all-reduce), so the *_deferred forms run the product only and leave the
 * fixed-order addition of the per-workgroup partials PENDING; nesie_pw_wgrad_flush_deferred(stream)
 * then finishes every pending gradient in ONE launch (descriptor table in the kernel arguments;
 * each element is summed exactly as by the immediate forms: bit-identical).  Until the flush the
 * workspace of a deferred launch must stay untouched and dw holds no valid data.  A launch that
 * runs as several column blocks reduces immediately.  nesie_pw_wgrad_pending(): how many wait;
 * nesie_pw_wgrad_drop_deferred(): forget them (error paths).  Host-side queue, one per process. */
int nesie_pw_wgrad_deferred(int nb, int ng, int co, int ci, long long p, const float *dy,
                            long long dy_bstride, const float *x, long long x_bstride,
                            const float *x_coef, int x_relu, float *dw, void *workspace,
                            size_t workspace_bytes, void *stream);
int nesie_pw_wgrad_flush_deferred(void *stream);
int nesie_pw_wgrad_pending(void);
int nesie_pw_wgrad_drop_deferred(void);

/* The same weight gradient fused with the BatchNorm + ReLU backward that PRODUCES its dY operand
 * (replaces nesie_bn_relu_backward_apply + nesie_pw_wgrad for a conv -> BatchNorm -> ReLU layer of
 * ConvModule, point_sa_module.py:277-289): da (nb, co, p) is the gradient of relu(bn(z)), z the raw
 * conv output (both at n*z_bstride), z_coef [ng*co][4] the layer's folded forward coefficients
 * (scale, shift, mean, invstd), part [(ng*co) * nslots * 2] the (sum g, sum g zhat) partials the
 * input-gradient launch that produced da left (nesie_pw_dgrad_bn_reduce).  Tiles of (da, z) are
 * turned into dz = gamma invstd (g - mean(g) - zhat mean(g zhat)), g = da [fma(z, scale, shift) > 0],
 * on their way into LDS: dz is the MFMA operand and is written once (dz may be da) for the
 * input-gradient launch that follows; dgamma, dbeta [ng*co] are written.  coef_ws: ng*co*8 floats.
 * d_row_bias (optional): the gradient of a per-group row bias that the forward added to z
 * (nesie_pw_layer_forward's row_bias, groups of rb_group = 16 or 64 positions):
 * d_row_bias[n][r][pos / rb_group] = sum of dz over the group; with rb_group 64 the buffer must
 * arrive ZERO-FILLED (two partial sums per group are added atomically); requires z_bstride = co*p.
 * Supported where one launch owns every column of dw (nesie_pw_wgrad_bn_supported). */
int nesie_pw_wgrad_bn_supported(int co, int ci, long long p);
int nesie_pw_wgrad_bn_backward(int nb, int ng, int co, int ci, long long p, const float *da,
                               const float *z, long long z_bstride, const float *z_coef,
                               const float *gamma, const float *part, int nslots, const float *x,
                               long long x_bstride, const float *x_coef, int x_relu, float *dz,
                               float *dw, float *dgamma, float *dbeta, float *coef_ws,
                               float *d_row_bias, int rb_group, void *workspace,
                               size_t workspace_bytes, void *stream);
/* ... with the weight gradient's reduction left pending (dz, dgamma, dbeta complete on return). */
int nesie_pw_wgrad_bn_backward_deferred(int nb, int ng, int co, int ci, long long p, const float *da,
                               const float *z, long long z_bstride, const float *z_coef,
                               const float *gamma, const float *part, int nslots, const float *x,
                               long long x_bstride, const float *x_coef, int x_relu, float *dz,
                               float *dw, float *dgamma, float *dbeta, float *coef_ws,
                               float *d_row_bias, int rb_group, void *workspace,
                               size_t workspace_bytes, void *stream);

/* The layer kernel for skinny HBM-bound first layers (cin <= 64, cout <= 128): W stays in
 * LDS / registers and every wave streams its own 32-position columns straight from global
 * memory into the MFMA operand registers (no LDS tile, no barrier in the main loop).
 *   y[b] = W . act(x[b]),  x[b] (cin, p) at x + b*x_bstride, W (cout, cin), y (B, cout, p);
 *   act(v) = in_coef ? relu?(in_coef[k][0] * v + in_coef[k][1]) : v;
 *   stat_partial (NULL = skip): nesie_mlp_stream_partials(b, p) x cout x 2 floats, the
 *   per-workgroup (sum, sum of squares) of y. */
long long nesie_mlp_stream_partials(int b, long long p);
/* (sum, sum of squares) partials -> coef[c][4] = (scale, bias, mean, invstd) (fp64), running
 * statistics updated like torch.nn.BatchNorm2d in training mode.  stat_partial is [nparts][c][2]
 * (the streaming kernel) or, channel_major != 0, [c][nparts][2] (nesie_blend_conv_forward). */
int nesie_mlp_stat_finalize(int c, long long nparts, double count, const float *stat_partial,
                            const float *gamma, const float *beta, float *running_mean,
                            float *running_var, float momentum, float eps, float *coef,
                            int channel_major, void *stream);
int nesie_mlp_layer_forward_stream(int b, int cin, int cout, long long p, const float *x,
                                   long long x_bstride, const float *w, const float *in_coef,
                                   int in_relu, float *y, float *stat_partial, void *stream);

/* ---- round 5: the first shared-MLP layer of PointSAModule over a 4-channel input WITHOUT its
 * output tensor.  Reference: the first ConvModule (Conv2d(4, 64, 1) + BatchNorm2d + ReLU) of SA1's
 * `self.mlps[0]` (point_sa_module.py:277-289, pointnet2_sa_ssg.py:62-79), its output's three
 * consumers in the forward / backward of the SECOND ConvModule, and autograd's conv2d / batch_norm
 * backward for the first.  Z0 = W0 . X4 (X4 = the grouped coordinates + height, (nb, 4, p), 17 MB at
 * 8 x 131 072 positions; Z0 would be 268 MB) is rebuilt from X4 with one fixed fma chain wherever
 * it is an operand, and the first layer's weight gradient follows from reductions:
 *  - nesie_k4_moments: the first and second moments of X4 (20 sums, per-workgroup partials in
 *    double, nesie_k4_moments_bytes() bytes); nesie_k4_stat_finalize: the first layer's BatchNorm
 *    statistics from them -- mean(Z0[c]) = W0[c] . mean(X4), var(Z0[c]) = W0[c]^T Cov(X4) W0[c] --
 *    as coef [64][4] = (scale, bias, mean, invstd) + the running statistics (no pass over Z0;
 *    nesie_mlp_layer_forward_stream with y == NULL is the direct form of the same statistics);
 *  - nesie_pw_layer_forward_k4: the second layer, y = W . relu(in_coef . Z0) (cout = 64; w0 (64, 4)
 *    row-major; in_coef [64][4] the first layer's folded norm), with statistics like
 *    nesie_pw_layer_forward;
 *  - nesie_pw_wgrad_bn_backward_k4: nesie_pw_wgrad_bn_backward of the second layer (64 x 64) with
 *    its X operand rebuilt (x_coef = in_coef above); defer != 0 leaves the reduction pending;
 *  - nesie_pw_dgrad_bn_reduce_k4: nesie_pw_dgrad_bn_reduce of the second layer whose OUTPUT (the
 *    gradient of the first activation) is not stored: bn_part as there, and
 *    g_part[(m * nslots + slot)][4] = sum over the slot's positions of gg[m] X4[j], j = 0 .. 3
 *    (nslots = nesie_pw_stat_slots(nb, 1, k, 64, p));
 *  - nesie_k4_first_layer_wgrad: dW0[m][j] = a G[m][j] + e0 Sx[j] + d1 (mu Sx[j] - sum_k W0[m][k] M[k][j])
 *    with (.., a, mu, d1, e0, ..) = bnb[m][2..5] of nesie_pw_bnb_coef over that bn_part, G the slot
 *    sums of g_part, Sx / M the first and second moments of X4 (mom_part of nesie_k4_moments).
 * All of it requires no gradient for X4 (the backbone's grouped input coordinates). */
int nesie_pw_layer_forward_k4(int nb, int cout, long long p, const float *x4, long long x4_bstride,
                              const float *w0, const float *w, int w_rstride, int w_cstride,
                              const float *in_coef, float *y, long long y_bstride, float *stat_part,
                              void *stream);
int nesie_pw_dgrad_bn_reduce_k4(int nb, int k, long long p, const float *x, long long x_bstride,
                                const float *w, int w_rstride, int w_cstride, const float *x4,
                                long long x4_bstride, const float *w0, const float *bn_coef,
                                float *bn_part, float *g_part, void *stream);
int nesie_pw_wgrad_bn_backward_k4(int nb, long long p, const float *da, const float *z,
                                  long long z_bstride, const float *z_coef, const float *gamma,
                                  const float *part, int nslots, const float *x4,
                                  long long x4_bstride, const float *w0, const float *x_coef,
                                  float *dz, float *dw, float *dgamma, float *dbeta, float *coef_ws,
                                  void *workspace, size_t workspace_bytes, int defer, void *stream);
/* nesie_pw_wgrad_bn_backward_k4 and nesie_pw_dgrad_bn_reduce_k4 as ONE launch: the dZ tile that the
 * weight gradient forms in LDS is also multiplied by w^T (w (64, 64) row-major, the layer's weight) and
 * only the reductions of that product leave the kernel -- in_part [64][slots][2], in_gpart
 * [64][slots][4], slots = nesie_pw_wgrad_bn_backward_k4_slots(nb, p); dZ itself is written nowhere
 * (da is left untouched). */
int nesie_pw_wgrad_bn_backward_k4_slots(int nb, long long p);
int nesie_pw_wgrad_bn_backward_k4_fused(int nb, long long p, const float *da, const float *z,
                                        long long z_bstride, const float *z_coef, const float *gamma,
                                        const float *part, int nslots, const float *x4,
                                        long long x4_bstride, const float *w0, const float *x_coef,
                                        const float *w, float *dw, float *dgamma, float *dbeta,
                                        float *coef_ws, float *in_part, float *in_gpart,
                                        void *workspace, size_t workspace_bytes, int defer, void *stream);
size_t nesie_k4_moments_bytes(void);
int nesie_k4_moments(int nb, long long p, const float *x4, long long x4_bstride, void *mom_part,
                     void *stream);
int nesie_k4_stat_finalize(double count, const void *mom_part, const float *w0, const float *gamma,
                           const float *beta, float *running_mean, float *running_var, float momentum,
                           float eps, float *coef, void *stream);
int nesie_k4_first_layer_wgrad(const void *mom_part, const float *w0, const float *bnb,
                               const float *g_part, int nslots, float *dw, void *stream);

/* Weight gradient of a 1x1 conv on the matrix cores: dw[cout][cin] = sum over scenes and
 * positions of dy[b][m][p] * act(x[b][k][p]); dy (B, cout, p), x[b] (cin, p) at x + b*x_bstride,
 * act as in nesie_mlp_layer_forward_stream (x_coef NULL = identity).  Reference: the conv2d backward
 * that autograd runs for ConvModule's Conv2d (point_sa_module.py:277-289).  Partials are added in
 * a fixed order (bitwise reproducible).  dy[b] (cout, p) at dy + b*dy_bstride.  cout <= 128 with
 * cin <= 288, or cout <= 256 with cin <= 128;
 * workspace = nesie_conv_wgrad_workspace_bytes(b, cout, cin, p). */
size_t nesie_conv_wgrad_workspace_bytes(int b, int cout, int cin, long long p);
int nesie_conv_wgrad(int b, int cout, int cin, long long p, const float *dy,
                     long long dy_bstride, const float *x, long long x_bstride,
                     const float *x_coef, int x_relu, float *dw, void *workspace,
                     size_t workspace_bytes, void *stream);
/* The same with the BatchNorm + ReLU backward that produces dy applied on the load: da = gradient of
 * relu(bn(z)), bnb [cout][8] from nesie_pw_bnb_coef; dz is never written (for a layer whose input
 * needs no gradient, e.g. the first layer of SA1: nothing else reads it). */
int nesie_conv_wgrad_bn(int b, int cout, int cin, long long p, const float *da, const float *z,
                        long long z_bstride, const float *bnb, const float *x, long long x_bstride,
                        const float *x_coef, int x_relu, float *dw, void *workspace,
                        size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NESIE_OPS_H_ */
