/*
 * pcb_hip.h -- C ABI of libpcb_hip.so, the MI355X (gfx950) implementation of the
 * PointNet++ set-abstraction / feature-propagation operators and the DGCNN kNN / EdgeConv
 * operator of UT-Team-Chun/Pointcloud-bridge.
 *
 * The reference has no FFI for this path: the operators are ATen compositions inside two Python
 * files.  Each entry point below replaces one of those compositions and names it
 * (paths relative to the reference's Highway_bridge/ directory).  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; outputs are pre-allocated; the
 *     library allocates no device memory.  Process-wide state is limited to two things, neither of
 *     which decides which bytes a launch may touch: the concurrency hint (pcb_set_concurrency_hint,
 *     an atomic int read once per call: it sizes grids, never buffers -- slab counts are explicit
 *     arguments) and the roofline timer (pcb_timer_*, a mutex-guarded event list, off unless armed).
 *     Calls on different streams are otherwise independent.
 *   - tensors are dense row-major with the shapes written next to each argument
 *   - indices are int64 at the boundary, as in the reference (torch.long)
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work
 *   - return value: PCB_OK (0) or a negative pcb_status; no exceptions cross the boundary
 */
#ifndef PCB_HIP_H
#define PCB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Storage type of the activation rows of the shared-MLP engine: bf16 (8 columns per 16-byte chunk;
 * BASELINE config 2) or fp32 (4 columns per chunk; the parity mode: logits within 1e-4 of the
 * reference's fp32 Conv/BatchNorm).  Statistics, constants, parameters and their gradients are fp32
 * in both. */
typedef enum pcb_dtype { PCB_DTYPE_BF16 = 0, PCB_DTYPE_F32 = 1 } pcb_dtype;

typedef enum pcb_status {
    PCB_OK = 0,
    PCB_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, k or nsample out of range */
    PCB_ERR_UNSUPPORTED = -2, /* size outside what the kernels are built for (see each call) */
    PCB_ERR_LAUNCH = -3       /* hipGetLastError() reported a failure after the launch */
} pcb_status;

/* Library/ABI version (major*100 + minor). */
int pcb_version(void);

/* Text for a pcb_status value (static storage). */
const char *pcb_status_string(int status);

/*
 * Pairwise squared distances, materialised.  Replaces square_distance,
 * models/pointnet2_utils.py:7-14, for callers that want the matrix itself (none of the operators
 * below do).  src [B,N,3], dst [B,M,3] -> out [B,N,M] fp32, d = ((-2*dot) + |s|^2) + |t|^2.
 */
int pcb_square_distance(const float *src, const float *dst, int B, int N, int M, float *out,
                        void *stream);

/*
 * Farthest point sampling.  Replaces farthest_point_sample, models/pointnet2_utils.py:63-80.
 *   xyz       [B,N,3] fp32
 *   start_idx [B] int64   first sample per scene: the value the reference draws with
 *                         torch.randint on the CPU generator (:69); the caller draws it
 *   out_idx   [B,S] int64
 * Distances are ((dx*dx + dy*dy) + dz*dz) in fp32, the running minimum is updated on strict
 * "<", the next sample is the FIRST index of the maximum -- bit-identical to the reference.
 * Supported: 1 <= N <= 40960 (one workgroup per scene; N <= 16384 keeps the cloud in registers).
 */
int pcb_fps(const float *xyz, int B, int N, int S, const int64_t *start_idx, int64_t *out_idx,
            void *stream);

/*
 * Ball query.  Replaces query_ball_point, models/pointnet2_utils.py:97-112 (and the
 * square_distance call inside it, :7-14, which is never materialised here).
 *   xyz [B,N,3], new_xyz [B,S,3] fp32;  r2 = (float)(radius*radius);  out_idx [B,S,nsample] int64
 * For each centroid: the first `nsample` indices i (ascending) with d(i) <= r2, where
 * d = ((-2*dot) + |c|^2) + |p|^2, dot = fma(cz,pz, fma(cy,py, cx*px)); missing slots repeat the
 * first hit; a centroid with no hit gets N in every slot (as the reference leaves it).
 * Requires 1 <= nsample <= N.
 */
int pcb_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, float r2,
                   int nsample, int64_t *out_idx, void *stream);

/*
 * Two radii in one pass over the cloud (MultiScaleSetAbstraction.forward runs query_ball_point
 * once per radius on the same centroids, models/pointnet2_utils.py:340-341).  Same results as two
 * pcb_ball_query calls.
 */
int pcb_ball_query2(const float *xyz, const float *new_xyz, int B, int N, int S,
                    float r2_a, int nsample_a, int64_t *out_idx_a,
                    float r2_b, int nsample_b, int64_t *out_idx_b, void *stream);

/*
 * Row gather with index clamp.  Replaces index_points, models/pointnet2_utils.py:17-39.
 *   points [B,N,C] fp32, idx [B,M] int64 (clamped to [0,N-1] as :34-36), out [B,M,C]
 */
int pcb_gather_rows(const float *points, const int64_t *idx, int B, int N, int C, int M,
                    float *out, void *stream);

/*
 * Backward of pcb_gather_rows: grad_points[b, clamp(idx[b,m]), :] += grad_out[b,m,:].
 * grad_points [B,N,C] must be zero-filled (or hold the value to accumulate onto) by the caller.
 */
int pcb_gather_rows_bwd(const float *grad_out, const int64_t *idx, int B, int N, int C, int M,
                        float *grad_points, void *stream);

/*
 * Grouping of sample_and_group / MultiScaleSetAbstraction, models/pointnet2_utils.py:51-58 and
 * :342-349: out[b,s,j,:] = cat(xyz[b,idx[b,s,j]] - new_xyz[b,s], feat[b,idx[b,s,j]]).
 *   xyz [B,N,3], new_xyz [B,S,3], feat [B,N,C] or NULL (C = 0), idx [B,S,ns] int64 (clamped),
 *   out [B,S,ns,3+C] fp32
 */
int pcb_group_points(const float *xyz, const float *new_xyz, const float *feat, const int64_t *idx,
                     int B, int N, int S, int ns, int C, float *out, void *stream);

/*
 * Backward of pcb_group_points w.r.t. feat: grad_feat[b,idx,c] += grad_out[b,s,j,3+c].
 * grad_feat [B,N,C] zero-filled by the caller.  (xyz carries no gradient: it is input data.)
 */
int pcb_group_points_bwd(const float *grad_out, const int64_t *idx, int B, int N, int S, int ns,
                         int C, float *grad_feat, void *stream);

/*
 * k nearest centroids for interpolation.  Replaces square_distance + full sort + [:k] in
 * FeaturePropagation.forward (k = 3, models/pointnet2_utils.py:185-188) and
 * EnhancedFeaturePropagation.forward (k = 4, :253-256).
 *   xyz1 [B,N,3] queries, xyz2 [B,S,3] candidates, out_d2 [B,N,k] fp32, out_idx [B,N,k] int64
 * Ascending distance; equal distances keep ascending index order (the reference's CPU sort is
 * stable).  Distances are the unclamped expansion-formula values (may be slightly negative).
 * Requires 1 <= k <= 4 and S >= k.
 */
int pcb_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int k, float *out_d2,
                 int64_t *out_idx, void *stream);

/*
 * Inverse-distance interpolation, models/pointnet2_utils.py:191-196 / :259-267:
 *   w = 1/(d2 + 1e-8); w /= sum_k w; out[b,n,:] = sum_k w_k * feat[b,idx[b,n,k],:]
 *   feat [B,S,C], d2/idx [B,N,k], out [B,N,C], out_w [B,N,k] (optional, NULL to skip)
 */
int pcb_interpolate(const float *feat, const float *d2, const int64_t *idx, int B, int N, int S,
                    int C, int k, float *out, float *out_w, void *stream);

/*
 * Backward of pcb_interpolate w.r.t. feat: grad_feat[b,idx[b,n,q],:] += w[b,n,q]*grad_out[b,n,:].
 * grad_feat [B,S,C] zero-filled by the caller.
 */
int pcb_interpolate_bwd(const float *grad_out, const float *w, const int64_t *idx, int B, int N,
                        int S, int C, int k, float *grad_feat, void *stream);

/*
 * kNN graph.  Replaces DGCNN.knn, models/DGCNN.py:49-70 (the [B,N,N] matrix is never stored).
 *   x [B,N,D] fp32 (the reference transposes its [B,D,N] input to this layout first, :60)
 *   out_idx [B,N,k] int64, nearest first (slot 0 is the point itself on duplicate-free data)
 * pd(i,j) = (|xi|^2 + (-2*<xi,xj>)) + |xj|^2 with <,> an fma chain in channel order and |x|^2 a
 * left-to-right sum of squares; the k smallest pd, ties by lower index.
 * norms [B,N] fp32: caller-owned scratch (receives |x|^2 of every point from a pre-pass).
 * Requires 1 <= k <= 32, k <= N, 1 <= D <= 128.
 */
int pcb_knn(const float *x, int B, int N, int D, int k, float *norms, int64_t *out_idx, void *stream);

/*
 * pcb_knn for feature-space graphs (32 <= D <= 128, k <= 20: DGCNN's second to fourth get_graph_feature calls,
 * models/DGCNN.py:137-150) through a screening pass on the bf16 matrix core (csrc/knn.hip): approximate distances from
 * a two-term bf16 split of the features pick 24 candidates per query, their distances are recomputed exactly and the
 * result is certified against a proven error bound; queries that cannot be certified are recomputed by pcb_knn's own
 * kernel.  SAME output as pcb_knn for every input.  Shapes the pass does not serve (workspace size 0) go to pcb_knn.
 *   workspace: pcb_knn_screen_workspace(B, N, D, k) bytes of caller-owned scratch (NULL: pcb_knn); after the call its
 *   int32 words [B, 2B) hold the number of recomputed queries of every scene.
 */
long pcb_knn_screen_workspace(int B, int N, int D, int k);
int pcb_knn_screened(const float *x, int B, int N, int D, int k, float *norms, void *workspace, int64_t *out_idx,
                     void *stream);

/*
 * kNN on 3-D coordinates (the D = 3 case of pcb_knn: DGCNN's first graph, models/DGCNN.py:134, and
 * torch.cdist + topk of BridgeStructureEncoding, models/attention_modules.py:584-586) through a
 * uniform grid: the cloud is sorted by cell and every query looks at the cells around it in growing
 * cubes until its k-th best distance is provably final.  SAME output as pcb_knn(x = xyz, D = 3):
 * same pd arithmetic, ties by lower index, nearest first.  Scenes whose points crowd into few
 * cells and clouds outside 1024 <= N <= 16384 are computed with all pairs (the pcb_knn kernel).
 *   workspace: pcb_knn_xyz_workspace(B, N) bytes of caller-owned scratch (NULL: all pairs)
 *   norms [B,N] fp32: scratch as in pcb_knn.
 */
long pcb_knn_xyz_workspace(int B, int N);
int pcb_knn_xyz(const float *xyz, int B, int N, int k, float *norms, void *workspace, int64_t *out_idx,
                void *stream);

/*
 * Local-structure descriptor of the k-neighbourhood of every point.  Replaces the neighbour gather
 * and get_structure_features of BridgeStructureEncoding, models/attention_modules.py:595-603 and
 * :620-687 (rel = x_j - x_i; ascending eigenvalues e of rel^T rel/(k-1) by batched eigh; the
 * [B*N,k,k] direction-similarity bmm; std() unbiased).
 *   xyz [B,N,3] fp32, idx [B,N,k] int64 (e.g. from pcb_knn with D = 3; clamped to [0,N-1])
 *   feat [B,N,13] fp32 = ((e0-e1)/(e0+1e-8), (e1-e2)/(e0+1e-8), e2/(e0+1e-8), max/mean/std of
 *        |rel - mean(rel)|, mean_jl(n_j.n_l) with n = rel/(|rel|+1e-8), std(rel_z), range(rel_z),
 *        mean(rel) (3), |std(rel)|) in the reference's order (:674-681)
 *   rel  [B,N,k,3] fp32 or NULL: the offsets themselves (:600), for the encoder's first layer
 * Requires 2 <= k <= 32 (k = 1 makes the reference's std() NaN).
 */
int pcb_structure_features(const float *xyz, const int64_t *idx, int B, int N, int k, float *feat,
                           float *rel, void *stream);

/*
 * Neighbourhood MLP.  Replaces BridgeStructureEncoding.structure_mlp applied to the expanded
 * [B, 6*freq+3+13, N, k] tensor, models/attention_modules.py:548-553 and :606-616
 * (Conv2d 1x1 -> BatchNorm2d -> ReLU -> Conv2d 1x1 -> max over k), with the first convolution
 * split by input block: base [P,C] = bias + W[:, per-point channels] . per-point features (the
 * caller's product, P = B*N points), and per neighbour only  Wr [C,3] . rel[i,j,:]:
 *     y1 = base[i] + Wr.rel[i,j]   z = relu(scale*y1 + shift)   out[i] = max_j (W2.z + b2)
 * All fp32, channels-last; 1 <= C <= 16, 1 <= k <= 255.  Training-mode sequence:
 *   pcb_nbr_mlp_stats    -> sums [parts][2][C] (sum y1, sum y1^2), parts = pcb_nbr_mlp_partials(P)
 *   pcb_bn_finalize(sums, parts, rows = P*k, ...) -> scale, shift, mean, invstd (+ running stats)
 *   pcb_nbr_mlp_forward  -> out [P,C], arg [P,C] uint8 (winning neighbour per channel)
 * Backward (dout [P,C]):
 *   pcb_nbr_mlp_backward_reduce -> sums [parts][2][C] (sum du, sum du*xhat; du = gradient at the
 *        BatchNorm output behind the ReLU mask), dw2 [parts][C][C+1] (dW2 | db2 partial sums)
 *   pcb_bn_bwd_finalize(sums, parts, rows = P*k, ...) -> p, q, dgamma, dbeta
 *   pcb_nbr_mlp_backward_apply  -> dbase [P,C] (sum_j dy1), dwr [parts][C][3] (partials of
 *        sum_ij dy1 (x) rel), with dy1 = scale*du + p*y1 + q
 * The caller adds the `parts` slabs of dw2 / dwr (they are written, not accumulated).
 */
int pcb_nbr_mlp_partials(long points);
int pcb_nbr_mlp_stats(const float *base, const float *rel, long P, int k, int C, const float *wr,
                      float *sums, void *stream);
int pcb_nbr_mlp_forward(const float *base, const float *rel, long P, int k, int C, const float *wr,
                        const float *scale, const float *shift, const float *w2, const float *b2,
                        float *out, unsigned char *arg, void *stream);
int pcb_nbr_mlp_backward_reduce(const float *base, const float *rel, long P, int k, int C,
                                const float *wr, const float *scale, const float *shift,
                                const float *mean, const float *invstd, const float *w2,
                                const float *dout, const unsigned char *arg, float *sums, float *dw2,
                                void *stream);
int pcb_nbr_mlp_backward_apply(const float *base, const float *rel, long P, int k, int C,
                               const float *wr, const float *scale, const float *shift, const float *p,
                               const float *q, const float *w2, const float *dout,
                               const unsigned char *arg, float *dbase, float *dwr, void *stream);

/*
 * Narrow pointwise layers on fp32 rows.  Replace the Conv1d / Conv2d 1x1 layers with 3..40 channels
 * of the bridge encoders, models/attention_modules.py:548-553 (per-point block), :696-716, :759-764
 * (N = 3..16 output columns: no MFMA tile to fill; they are HBM streams of 4*(Ci+Co) bytes per row).
 *   pcb_rows_linear_f32        y [P,Co] = x [P,Ci] . w[Co,Ci]^T + bias[Co] (bias may be NULL)
 *   pcb_rows_linear_dgrad_f32  dx [P,Ci] = dy [P,Co] . w[Co,Ci]
 *   pcb_rows_linear_wgrad_f32  partials [parts][Co*Ci + Co]: per-block sums of dy^T x as a [Co][Ci] block, then
 *                              the Co sums of dy (bias gradient); parts = pcb_rows_linear_wgrad_partials(P); the caller adds
 *                              the slabs.  1 <= Ci, Co <= 64.
 */
int pcb_rows_linear_f32(const float *x, const float *w, const float *bias, long P, int Ci, int Co, float *y,
                        void *stream);
int pcb_rows_linear_dgrad_f32(const float *dy, const float *w, long P, int Ci, int Co, float *dx, void *stream);
int pcb_rows_linear_wgrad_partials(long P);
int pcb_rows_linear_wgrad_f32(const float *dy, const float *x, long P, int Ci, int Co, float *partials,
                              void *stream);

/*
 * EdgeConv edge features.  Replaces the gather/repeat/cat of DGCNN.get_graph_feature,
 * models/DGCNN.py:90-107: out[b,n,j,:] = cat(x[b,idx[b,n,j]] - x[b,n], x[b,n]).
 *   x [B,N,D], idx [B,N,k] int64, out [B,N,k,2D]   (channels-last; the reference's [B,2D,N,k]
 *   is out.permute(0,3,1,2))
 */
int pcb_edge_features(const float *x, const int64_t *idx, int B, int N, int D, int k, float *out,
                      void *stream);

/*
 * Backward of pcb_edge_features: grad_x[b,idx,c] += g[b,n,j,c];
 * grad_x[b,n,c] += sum_j (g[b,n,j,D+c] - g[b,n,j,c]).  grad_x [B,N,D] zero-filled by the caller.
 */
int pcb_edge_features_bwd(const float *grad_out, const int64_t *idx, int B, int N, int D, int k,
                          float *grad_x, void *stream);

/* ------------------------------------------------------------------------------------------
 * bf16 channels-last row kernels around the pointwise-MLP GEMMs (csrc/rowbn.hip).
 * They replace, per layer, the ATen passes behind Conv(1x1) -> BatchNorm -> ReLU/LeakyReLU
 * [-> max over neighbours] in SetAbstraction.forward (models/pointnet2_utils.py:149-154),
 * MultiScaleSetAbstraction.forward (:353-356), FeaturePropagation.forward (:207-209) and the
 * EdgeConv blocks of DGCNN.forward (models/DGCNN.py:134-148).
 * Activations are bf16 rows [rows, C] (C % 8 == 0, C <= 2048); statistics, scale/shift and
 * gradients of the affine parameters are fp32.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.2).
 */

/* Every entry point of this section exists for both row types: *_bf16 (C % 8 == 0, C <= 2048) and
 * *_f32 (C % 4 == 0, C <= 1024) with the same arguments. */

/* sums[0][c] += sum_r y[r][c]; sums[1][c] += sum_r y[r][c]^2.  sums [2,C] fp32, zeroed by the caller (any C that is a multiple of the chunk). */
int pcb_colstats_bf16(const void *y, long rows, int C, float *sums, void *stream);
int pcb_colstats_f32(const void *y, long rows, int C, float *sums, void *stream);
/* The same without atomics (reproducible mode): slabs [nparts][2][C], every one of the launch's nparts (1..2048)
 * workgroups along the rows writes its own; pcb_bn_finalize / pcb_sum_slabs add them in slab order. */
int pcb_colstats_slabs_bf16(const void *y, long rows, int C, float *slabs, int nparts, void *stream);
int pcb_colstats_slabs_f32(const void *y, long rows, int C, float *slabs, int nparts, void *stream);

/* out[i] = sum over k < nparts of slabs[k][i], i < n, in slab order (fp64 accumulation).  The local
 * totals of a [nparts][2][C] statistics buffer (n = 2C) -- what a SyncBatchNorm all-reduce carries. */
int pcb_sum_slabs(const float *slabs, int nparts, int n, float *out, void *stream);

/*
 * BatchNorm bookkeeping of one layer: scale = gamma*invstd, shift = beta - mean*scale.
 * `sums` is [nparts][2][C]: nparts partial (sum, sum of squares) slabs, added here in slab order
 * (nparts = 1 after pcb_colstats_bf16, pcb_gemm_nt_partials(pro,R,N) after pcb_gemm_nt_bf16).
 * training != 0: batch statistics from sums/rows, running_mean/var updated with `momentum`
 * (unbiased variance), `bias` (the conv bias the GEMM leaves out because it cancels inside a
 * train-mode BatchNorm) added to the mean that enters running_mean.  training == 0: running
 * statistics, bias folded into shift.  gamma/beta/bias/running_* may be NULL where unused.
 * mean/invstd [C] are outputs for the backward pass.
 * count: the number of samples the statistics stand for in the unbiased running-variance factor
 * count/(count-1); 0 means rows.  (It exceeds rows when every row is a sample repeated count/rows
 * times -- nearest-neighbour upsampling before the layer, models/model.py:164 -- whose mean and
 * biased variance equal those of the distinct rows.)
 * num_batches_tracked (optional, int64 scalar on the device): incremented by one -- the counter
 * nn.BatchNorm.forward bumps in training mode.
 */
int pcb_bn_finalize(const float *sums, int nparts, long rows, long count, int C, const float *gamma, const float *beta,
                    const float *bias, float *running_mean, float *running_var, float momentum,
                    float eps, int training, float *scale, float *shift, float *mean, float *invstd,
                    long long *num_batches_tracked, void *stream);

/* z = act(y*scale + shift), y/z [rows,C] bf16. */
int pcb_bn_act_bf16(const void *y, const float *scale, const float *shift, long rows, int C, int act,
                    void *z, void *stream);
int pcb_bn_act_f32(const void *y, const float *scale, const float *shift, long rows, int C, int act,
                   void *z, void *stream);

/*
 * Pooled form (torch.max over the neighbour axis, pointnet2_utils.py:154 / :356, DGCNN.py:136):
 * out[g][c] = max_j act(y[g*ns+j][c]*scale+shift), argmax[g][c] = first j attaining it (uint8).
 * y [groups*ns, C] bf16, out [groups, C] bf16, 1 <= ns <= 255.
 */
int pcb_bn_act_max_bf16(const void *y, const float *scale, const float *shift, long groups, int ns,
                        int C, int act, void *out, unsigned char *argmax, void *stream);
int pcb_bn_act_max_f32(const void *y, const float *scale, const float *shift, long groups, int ns,
                       int C, int act, void *out, unsigned char *argmax, void *stream);

/*
 * Backward of act(BatchNorm(y)) for a dense upstream gradient dz [rows,C] bf16:
 * sums (zeroed by the caller) returns (sum du, sum du*xhat) = (dbeta, dgamma);
 * dy = scale*(du - s1/rows - xhat*s2/rows) if use_batch_stats else scale*du.  dy [rows,C] bf16.
 */
int pcb_bn_act_bwd_bf16(const void *dz, const void *y, const float *scale, const float *shift,
                        const float *mean, const float *invstd, long rows, int C, int act,
                        int use_batch_stats, float *sums, void *dy, void *stream);
int pcb_bn_act_bwd_f32(const void *dz, const void *y, const float *scale, const float *shift,
                       const float *mean, const float *invstd, long rows, int C, int act,
                       int use_batch_stats, float *sums, void *dy, void *stream);

/* The apply pass of pcb_bn_act_bwd_* alone, for totals sums [2,C] the caller formed itself (reproducible mode:
 * pcb_bn_act_bwd_reduce_* in slab form + pcb_sum_slabs instead of the atomically accumulated single slab). */
int pcb_bn_act_bwd_apply_bf16(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                              const float *invstd, const float *sums, long rows, int C, int act, int use_batch_stats,
                              void *dy, void *stream);
int pcb_bn_act_bwd_apply_f32(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                             const float *invstd, const float *sums, long rows, int C, int act, int use_batch_stats,
                             void *dy, void *stream);

/* Same for the pooled form: dout [groups,C] fp32 reaches only the arg-max rows; dy [groups*ns,C] bf16. */
int pcb_bn_act_max_bwd_bf16(const float *dout, const unsigned char *argmax, const void *y,
                            const float *scale, const float *shift, const float *mean,
                            const float *invstd, long groups, int ns, int C, int act,
                            int use_batch_stats, float *sums, void *dy, void *stream);
int pcb_bn_act_max_bwd_f32(const float *dout, const unsigned char *argmax, const void *y,
                           const float *scale, const float *shift, const float *mean,
                           const float *invstd, long groups, int ns, int C, int act,
                           int use_batch_stats, float *sums, void *dy, void *stream);

/*
 * Grouping straight into GEMM rows (sample_and_group / MSG grouping, pointnet2_utils.py:51-58,
 * :342-349): out[(b,s,j)] = [feat[b,idx] (C) | xyz[b,idx]-new_xyz[b,s] (3) | 0 ... Kp).
 * Features come FIRST here (16-byte chunks stay aligned); the caller permutes the weight columns.
 * feat [B,N,C] (row type) or NULL, out [B*S*ns, Kp], Kp % 8 == 0 (bf16) / Kp % 4 == 0 (fp32), Kp >= C+3.
 */
int pcb_group_rows_bf16(const float *xyz, const float *new_xyz, const void *feat, const int64_t *idx,
                        int B, int N, int S, int ns, int C, int Kp, void *out, void *stream);
int pcb_group_rows_f32(const float *xyz, const float *new_xyz, const void *feat, const int64_t *idx,
                       int B, int N, int S, int ns, int C, int Kp, void *out, void *stream);

/* grad_feat[b,idx,c] += grad_rows[row][c] (c < C); grad_feat [B,N,C] fp32 zeroed by the caller. */
int pcb_group_rows_bf16_bwd(const void *grad_rows, const int64_t *idx, int B, int N, int S, int ns,
                            int C, int Kp, float *grad_feat, void *stream);
int pcb_group_rows_f32_bwd(const void *grad_rows, const int64_t *idx, int B, int N, int S, int ns,
                           int C, int Kp, float *grad_feat, void *stream);

/* ------------------------------------------------------------------------------------------
 * MFMA row GEMMs with the neighbouring BatchNorm/activation algebra fused into operand loads and
 * epilogues (csrc/gemm.hip: bf16 rows on v_mfma_f32_32x32x16_bf16; csrc/gemm_f32.hip: fp32 rows on
 * the exact fp32 matrix core v_mfma_f32_32x32x2_f32).  Together with pcb_bn_finalize /
 * pcb_bn_act_max_* they are the whole Conv(1x1) -> BatchNorm -> activation stack of
 * models/pointnet2_utils.py:149-154, :207-209, :353-356 and models/DGCNN.py:134-148: a layer stores
 * only y = x W^T.  Every entry point exists as *_bf16 (N, K multiples of 8) and *_f32 (multiples of
 * 4) with the same arguments; in the fp32 mode no library GEMM and no ATen BatchNorm runs.
 *
 * A-operand prologue `pro`:
 *   0  plain rows a0 [R,K]
 *   1  act(a0*scale + shift)                     (previous layer's BatchNorm + activation)
 *   2  dy = scale*dz*act'(y*scale+shift) + p*y + q    with dz = a0, y = a1 (BatchNorm backward)
 *   3  the same with dz[r][c] = (r % ns == argmax[r/ns][c]) ? dout[r/ns][c] : 0  (max-pooled layer)
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.2).  scale/shift/p/q are fp32 [K].
 *
 * Statistics slabs: a launch given `sums` runs EXACTLY `nparts` workgroups along its row axis
 * (1 <= nparts <= 768) and workgroup i stores the column sums / sums of squares of the outputs it
 * produced into slab i of sums = [nparts][2][N] fp32 (no atomics; pcb_bn_finalize adds the slabs in
 * order; a workgroup without rows stores zeros).  The count is the caller's: it sized the buffer for
 * it and passes the same number to the finalize call.  pcb_gemm_nt_partials(pro,R,N) is the count
 * the library would pick itself (resident workgroups for the concurrency hint in force) -- a
 * recommendation, not a contract.
 */

/* out[R,N] = A'[R,K] . w[N,K]^T, fp32 accumulation.  sums/nparts: see above (NULL/0: no statistics). */
int pcb_gemm_nt_bf16(int pro, const void *a0, const void *a1, const float *scale, const float *shift,
                     const float *p, const float *q, const float *dout, const unsigned char *argmax,
                     int ns, int act, const void *w, long R, int N, int K, void *out, float *sums, int nparts,
                     void *stream);
int pcb_gemm_nt_f32(int pro, const void *a0, const void *a1, const float *scale, const float *shift,
                    const float *p, const float *q, const float *dout, const unsigned char *argmax,
                    int ns, int act, const void *w, long R, int N, int K, void *out, float *sums, int nparts,
                    void *stream);

/* out[R,N] fp32 = a[R,K] (bf16) . w[N,K]^T (bf16): the plain GEMM with an UNROUNDED result, for the
 * per-point products of csrc/gatherlin.hip (replaces torch.matmul on R = B*N rows). */
int pcb_gemm_nt_f32out_bf16(const void *a, const void *w, long R, int N, int K, float *out, void *stream);

/*
 * Conv without BatchNorm (the second conv of EnhancedFeaturePropagation.attention / .boundary_aware,
 * models/pointnet2_utils.py:232-236, :279-283, and the classifier conv of the heads, models/model.py:98,
 * models/pointnet2.py:33): out [R,N] = a [R,K] . w [N,K]^T + bias [N] (fp32, may be NULL), the bias
 * added to the fp32 accumulators.
 * pcb_prep_linear_bias_* builds its operands in one launch from the fp32 parameters: w [n,k] ->
 * wp [npad,kp] and (optional) wt [kp,npad] in the row type, zero padded; bias [n] (or NULL) -> bp [npad] fp32.
 * gap = D > 0: the n outputs are laid out like interpolate+concat rows (first D in place, the others
 * from column pad(D), the next multiple of 8 (bf16) / 4 (fp32)), so the result can gate / join such
 * rows without a permutation.
 */
int pcb_gemm_nt_bias_bf16(const void *a, const void *w, const float *bias, long R, int N, int K, void *out,
                          void *stream);
int pcb_gemm_nt_bias_f32(const void *a, const void *w, const float *bias, long R, int N, int K, void *out,
                         void *stream);
/* pcb_gemm_nt_bias_bf16 whose epilogue also adds the bf16 rows res [R,N] to the rounded result: x = trunk + boundary
 * term of EnhancedFeaturePropagation.forward (models/pointnet2_utils.py:296) without an addition pass. */
int pcb_gemm_nt_bias_add_bf16(const void *a, const void *w, const float *bias, const void *res, long R, int N, int K,
                              void *out, void *stream);
/* The GEMM of a Conv+BatchNorm layer whose input is the concatenation [x | level1 | level2] with the coarse levels
 * REPEATED 2^sh times along the rows -- MultiScaleFeatureFusion's nearest upsampling followed by final_fusion's first
 * conv (models/model.py:150-170, :93-99) -- without the concatenated rows: the conv is linear, so the caller forms the
 * coarse levels' share on their own rows (add_l = level_l W_l^T, fp32 [R >> sh_l, N]) and this entry computes
 *   out = bf16(a W^T + add1[r >> sh1] + add2[r >> sh2])      (a [R,K] bf16, w [N,K] prepared, sh >= 2, add2 may be NULL)
 * plus the batch statistics of out as slabs sums [nparts][2][N] (as pcb_gemm_nt_bf16 with sums).
 * pcb_dy_repeat_sums_bf16 is its backward towards the addends: d add_l[i] = sum of dy over the 2^sh_l rows coarse row i
 * stood for, dy rebuilt from (dz, y, BatchNorm-backward constants) as in pcb_dy_rows_bf16 (d2 may be NULL; sh2 >= sh1). */
int pcb_gemm_nt_stats_add_bf16(const void *a, const void *w, long R, int N, int K, void *out, float *sums, int nparts,
                               const float *add1, int sh1, const float *add2, int sh2, const float *centre, void *stream);
/* Centred storage of pre-BatchNorm rows (bf16 mode; reference semantics unchanged: models/pointnet2_utils.py:149-154,
 * :207-209, :353-356 -- BatchNorm is invariant to a per-channel constant in front of it, exactly as it is to the conv
 * bias).  A bf16 value carries an absolute error of 2^-9 |y| and train-mode BatchNorm divides by std(y): stored as is,
 * the rounding error in units of the normalised signal is 2^-9 (|mean|/std + 1) per layer.  So the forward GEMM of a
 * Conv+BatchNorm layer writes
 *   out = bf16(A' W^T - centre[N])        (pro 0: A' = a;  1: A' = act(a*scale + shift); centre may be NULL)
 * with the statistics slabs of THOSE rows (as pcb_gemm_nt_bf16 with sums), and pcb_bn_finalize_centred works on the
 * centred moments: scale, shift, mean, invstd all live in the centred frame, which is the frame every consumer of y
 * (operand prologues, pooling, BatchNorm backward) reads it in -- none of them changes.
 * cmode 0: as pcb_bn_finalize.  Training, cmode 1: running_mean takes mean + centre (+ bias), then centre += mean in the
 * channels where |mean| > std/4 (this batch's mean becomes the next call's centre; a centre that is close enough stays
 * put, which keeps the rounding of y reproducible between two passes over the same batch).  Training, cmode 2 (probe): centre += mean and NOTHING else.  Eval,
 * cmode 1: centre (an output) = running_mean - bias, mean = 0 -- run it BEFORE the GEMM, which subtracts centre.
 * The same `centre` argument on pcb_gemm_nt_stats_add_bf16 and pcb_gather_add_bf16. */
int pcb_gemm_nt_stats_bf16(int pro, const void *a, const float *scale, const float *shift, int act, const void *w, long R,
                           int N, int K, void *out, float *sums, int nparts, const float *centre, void *stream);
int pcb_bn_finalize_centred(const float *sums, int nparts, long rows, long count, int C, const float *gamma,
                            const float *beta, const float *bias, float *running_mean, float *running_var, float momentum,
                            float eps, int training, float *scale, float *shift, float *mean, float *invstd,
                            long long *num_batches_tracked, float *centre, int cmode, void *stream);
int pcb_dy_repeat_sums_bf16(const void *dz, const void *y, const float *scale, const float *shift, const float *p,
                            const float *q, int act, long R, int C, int sh1, float *d1, int sh2, float *d2, void *stream);
int pcb_prep_linear_bias_bf16(const float *w, const float *bias, int n, int k, int npad, int kp, int gap,
                              void *wp, void *wt, float *bp, void *stream);
int pcb_prep_linear_bias_f32(const float *w, const float *bias, int n, int k, int npad, int kp, int gap,
                             void *wp, void *wt, float *bp, void *stream);

/*
 * One layer's backward in ONE pass (bf16 rows): what pcb_gemm_nt_red_bf16 + pcb_gemm_tn_bf16 compute for layer l >= 1 of a
 * stack -- the reference's autograd of Conv(1x1) -> BatchNorm -> ReLU (models/pointnet2_utils.py:149-154, :353-356) -- for
 * the narrow layers that carry most rows (C, K multiples of 8, C <= 256, K <= 128: pcb_bwd_fused_supported):
 *   dx [R,K] bf16 = dy . W                       (= dz of the layer below)
 *   dW [C, out_cols] = dy^T . x'                 (nparts slabs [C*K] in `workspace`, summed in slab order; weight layout
 *                                                 and out_cols / out_perm as pcb_gemm_tn_bf16)
 *   red_sums [nparts][2][K] = the layer below's BatchNorm-backward sums (as pcb_gemm_nt_red_bf16)
 * dy = BatchNorm/activation backward of this layer from (dz, y) (pro 2) or (dout, argmax, y) (pro 3) with scale, shift,
 * p, q; x' = act(x*xscale + xshift) from the layer below's raw rows x [R,K] and its constants; wt [K,C] = the prepared
 * transposed weight.  (dz, y) and x stream from HBM once instead of twice.  The launch runs exactly nparts workgroups.
 */
int pcb_bwd_fused_supported(int C, int K);
int pcb_bwd_fused_bf16(int pro, const void *dz, const void *y, const float *scale, const float *shift, const float *p,
                       const float *q, const float *dout, const unsigned char *argmax, int ns, int act, const void *wt,
                       const void *x, const float *xscale, const float *xshift, const float *xmean, const float *xinvstd,
                       int xact, long R, int C, int K, void *dx, float *red_sums, int nparts, float *workspace, float *dW,
                       int out_cols, int out_perm, void *stream);

/*
 * BatchNorm (+ ReLU / LeakyReLU) on NARROW fp32 rows [R, C], any 1 <= C <= 64, and per-scene column sums: the colour /
 * fusion stacks of the reference's BridgeSeg network (models/attention_modules.py:696-722, :759-764: BatchNorm1d over 3, 6
 * or 16 channels on B*N rows; color_context's AdaptiveAvgPool1d).  Kernels only, no atomics: valid inside a captured step.
 *   pcb_rows_bn_stats_f32           slabs [nparts][2][C] of (sum x, sum x^2), nparts = pcb_rows_bn_partials(R, C) exactly;
 *                                   pcb_bn_finalize(slabs, nparts, R, ...) makes scale / shift / mean / invstd of them
 *   pcb_rows_bn_act_f32             z = act(x*scale + shift)
 *   pcb_rows_bn_act_bwd_reduce_f32  slabs [nparts][2][C] of (sum du, sum du*xhat) (add them with pcb_sum_slabs)
 *   pcb_rows_bn_act_bwd_apply_f32   dx = scale*(du - s1/R - xhat*s2/R) with sums = (s1, s2), or scale*du (use_batch_stats 0)
 *   pcb_scene_sum_f32               slabs [nparts][B][C] of the column sums of each scene's N rows of x [B*N, C],
 *                                   nparts = pcb_scene_sum_partials(N) exactly (pcb_sum_slabs(slabs, nparts, B*C, out))
 */
int pcb_rows_bn_partials(long R, int C);
int pcb_rows_bn_stats_f32(const float *x, long R, int C, float *slabs, int nparts, void *stream);
int pcb_rows_bn_act_f32(const float *x, const float *scale, const float *shift, long R, int C, int act, float *z,
                        void *stream);
int pcb_rows_bn_act_bwd_reduce_f32(const float *dz, const float *x, const float *scale, const float *shift,
                                   const float *mean, const float *invstd, long R, int C, int act, float *slabs,
                                   int nparts, void *stream);
int pcb_rows_bn_act_bwd_apply_f32(const float *dz, const float *x, const float *scale, const float *shift,
                                  const float *mean, const float *invstd, const float *sums, long R, int C, int act,
                                  int use_batch_stats, float *dx, void *stream);
int pcb_scene_sum_partials(int N);
int pcb_scene_sum_f32(const float *x, int B, int N, int C, float *slabs, int nparts, void *stream);

/* nn.Dropout on rows (the Dropout(0.5) in front of the segmentation heads' last conv, models/model.py:97, models/model.py:52):
 * out = x * keep / (1 - p), keep from a stateless 64-bit mix of (*seed, 16-byte vector index) -- 8 bits per element, p rounded
 * to a multiple of 1/256.  The backward pass is the SAME call on the gradient with the same seed (nothing is stored).  seed:
 * an int64 in device memory (a captured step replays the launch; whatever refreshes *seed inside the graph refreshes the
 * mask).  n elements, a multiple of 8 (bf16) / 4 (fp32); x may equal out. */
int pcb_dropout_rows_bf16(const void *x, long n, const long long *seed, float p, void *out, void *stream);
int pcb_dropout_rows_f32(const void *x, long n, const long long *seed, float p, void *out, void *stream);

/* Tell the library that another kernel occupies about `busy_cus` compute units beside the launches
 * that follow (e.g. the next batch's FPS on a side stream during the backward pass): the persistent
 * GEMMs without slabs, the weight-gradient splits and the slab-count recommendation then leave
 * those CUs alone.  0 = the GPU is ours (default).  An atomic hint: it never changes which buffers a
 * launch may touch. */
int pcb_set_concurrency_hint(int busy_cus);

/* Recommended slab count (= workgroups along the row axis) for a gemm_nt launch of these sizes under
 * the current hint; never more than 768. */
int pcb_gemm_nt_partials(int pro, long R, int N);

/* dW[M,N] (fp32, overwritten) = A'[R,M]^T . B'[R,N].
 * A' = dz [R,M] itself (apro 0) or dy (apro 2 or 3, as above, built from dz|dout+argmax and y [R,M]);
 * B' = x [R,N] (bpro 0) or act(x*xscale + xshift) (bpro 1).
 * The rows are split over workgroups; each split stores its partial tile into `workspace`
 * (pcb_gemm_tn_workspace(R,M,N) floats, caller-owned) and a second kernel sums the slabs in a fixed
 * order, so the result is bitwise reproducible (no atomics).
 * Output layout: out_cols <= 0 gives dW [M, N].  out_cols = k > 0 gives dW [M, k] in the layer's
 * REAL weight layout, dropping the padding of the row layout out_perm describes (perm of
 * pcb_prep_weights_*) -- the gradient then needs no unpadding pass. */
int pcb_gemm_tn_bf16(int apro, const void *dz, const void *y, const float *scale, const float *shift,
                     const float *p, const float *q, const float *dout, const unsigned char *argmax,
                     int ns, int act, int bpro, const void *x, const float *xscale, const float *xshift,
                     int xact, long R, int M, int N, float *workspace, float *dW, int out_cols,
                     int out_perm, void *stream);
/* Weight AND bias gradient of a conv without BatchNorm (attention[3] / boundary_aware[3] of EnhancedFeaturePropagation,
 * the classifier convs) in one pass over dy [R,M] and x [R,N]: dW as pcb_gemm_tn_bf16 writes it (apro 0, bpro 0),
 * dbias[M] = column sums of dy -- accumulated by the same workgroups per row split and summed with the slabs, instead of
 * a separate pass over dy (pcb_colstats_bf16). */
int pcb_gemm_tn_bias_bf16(const void *dy, const void *x, long R, int M, int N, float *workspace, float *dW, int out_cols,
                          int out_perm, float *dbias, void *stream);
int pcb_gemm_tn_f32(int apro, const void *dz, const void *y, const float *scale, const float *shift,
                    const float *p, const float *q, const float *dout, const unsigned char *argmax,
                    int ns, int act, int bpro, const void *x, const float *xscale, const float *xshift,
                    int xact, long R, int M, int N, float *workspace, float *dW, int out_cols,
                    int out_perm, void *stream);

/* Number of fp32 elements pcb_gemm_tn_* needs in `workspace` for these sizes (an upper bound for
 * every value of the concurrency hint). */
long pcb_gemm_tn_workspace(long R, int M, int N);

/* p, q of the fused BatchNorm backward from sums = [nparts][2][C] partial slabs of
 * (sum du, sum du*xhat): p = -scale*invstd*s2/rows, q = -scale*s1/rows - p*mean; zeros when
 * use_batch_stats == 0.  The parameter gradients the totals amount to are written to dgamma (= s2),
 * dbeta (= s1) and dbias (0 under batch statistics, scale*s1 otherwise), [C] each, any may be NULL.
 * With nparts == 1 (a slab the reduce kernels accumulated into with atomics) the slab is cleared
 * after use, ready for the next accumulation.
 * global_sums (optional, [2][C]): the same two sums over the rows of ALL ranks (SyncBatchNorm; `rows`
 * then counts all ranks' rows): p and q come from them, the parameter gradients stay local. */
int pcb_bn_bwd_finalize(float *sums, int nparts, long rows, int C, const float *scale,
                        const float *mean, const float *invstd, int use_batch_stats, float *p, float *q,
                        float *dgamma, float *dbeta, float *dbias, const float *global_sums, void *stream);

/* GEMM operands of n <= 8 layers from their fp32 master weights, in one launch
 * (replaces weight.view(Cout,Cin).to(bf16) / F.pad / .t().contiguous() per layer).
 * desc = n x 8 int64 on the HOST: {w fp32 [C,k], wp [C,kp], wt [kp,C] or 0, C, k, kp, perm, ldw}; wp, wt
 * in the row type of the entry point; ldw = floats between the rows of w (0 = k; larger: w is a column slice of a
 * wider parameter, read in place).
 * perm names the column layout of the layer's input rows: 0 = real columns in place, zero padded;
 * C > 0 = pcb_group_rows_* rows (C feature columns, then the 3 centred coordinates);
 * -D < 0 = interpolate+concat rows (first D columns in place, the rest from column pad(D)). */
int pcb_prep_weights_bf16(int n, const long long *desc, void *stream);
int pcb_prep_weights_f32(int n, const long long *desc, void *stream);
/* The same for any number of layers from a table in DEVICE memory: n rows of 8 int64 as above whose last slot
 * holds the row's first workgroup -- a layer takes ceil(C/32)*ceil(kp/32) workgroups, rows in ascending order, `blocks` =
 * their total: the operands of every stack of a network in one launch per optimiser step. */
int pcb_prep_weights_table_bf16(const long long *table, int n, long blocks, void *stream);
int pcb_prep_weights_table_f32(const long long *table, int n, long blocks, void *stream);
/* The same, and `zero` [zero_n] fp32 is cleared by the same launch (a stack's constants buffer). */
int pcb_prep_weights_zero_bf16(int n, const long long *desc, float *zero, long zero_n, void *stream);
int pcb_prep_weights_zero_f32(int n, const long long *desc, float *zero, long zero_n, void *stream);

/* The dy = scale*du + p*y + q of pcb_gemm_nt_bf16's prologue 2 written out as bf16 rows [R,C] (C <= 2048): for
 * layers whose gradient GEMMs would rebuild it once per column tile of a wide partner matrix (used by
 * pcb_mlp_stack_backward where C > 256 and the layer has more than 256 inputs). */
int pcb_dy_rows_bf16(const void *dz, const void *y, const float *scale, const float *shift, const float *p,
                     const float *q, int act, long R, int C, void *dy, void *stream);

/* pcb_gemm_nt_* (pro 2 or 3, N <= 128) whose epilogue also accumulates the BatchNorm-backward
 * sums of the layer BELOW: the produced tile is that layer's dz; with its y (red_y [R,N]) and
 * constants, (sum du, sum du*xhat) go to red_sums = [nparts][2][N] slabs (nparts as above).
 * Saves the separate pcb_bn_act_bwd_reduce_* pass over (dz, y). */
int pcb_gemm_nt_red_bf16(int pro, const void *a0, const void *a1, const float *scale, const float *shift,
                         const float *p, const float *q, const float *dout, const unsigned char *argmax,
                         int ns, int act, const void *w, long R, int N, int K, void *out, const void *red_y,
                         const float *red_scale, const float *red_shift, const float *red_mean,
                         const float *red_invstd, int red_act, float *red_sums, int nparts, void *stream);
int pcb_gemm_nt_red_f32(int pro, const void *a0, const void *a1, const float *scale, const float *shift,
                        const float *p, const float *q, const float *dout, const unsigned char *argmax,
                        int ns, int act, const void *w, long R, int N, int K, void *out, const void *red_y,
                        const float *red_scale, const float *red_shift, const float *red_mean,
                        const float *red_invstd, int red_act, float *red_sums, int nparts, void *stream);

/* Backward sums only (no dy written): (sum du, sum du*xhat) for a dense dz.  sums = [nparts][2][C]:
 * nparts > 1 -- one workgroup per slab, EVERY slab written (idle ones with zeros), no atomics, summed
 * in slab order by pcb_bn_bwd_finalize(sums, nparts, ...) (1 < nparts <= 768);  nparts == 1 -- a
 * single slab that must be zero on entry and is accumulated with fp32 atomics. */
int pcb_bn_act_bwd_reduce_bf16(const void *dz, const void *y, const float *scale, const float *shift,
                               const float *mean, const float *invstd, long rows, int C, int act,
                               float *sums, int nparts, void *stream);
int pcb_bn_act_bwd_reduce_f32(const void *dz, const void *y, const float *scale, const float *shift,
                              const float *mean, const float *invstd, long rows, int C, int act,
                              float *sums, int nparts, void *stream);

/* ... and for a max-pooled layer (dout [groups,C] fp32, argmax [groups,C] uint8). */
int pcb_bn_act_max_bwd_reduce_bf16(const float *dout, const unsigned char *argmax, const void *y,
                                   const float *scale, const float *shift, const float *mean,
                                   const float *invstd, long groups, int ns, int C, int act, float *sums,
                                   int nparts, void *stream);
int pcb_bn_act_max_bwd_reduce_f32(const float *dout, const unsigned char *argmax, const void *y,
                                  const float *scale, const float *shift, const float *mean,
                                  const float *invstd, long groups, int ns, int C, int act, float *sums,
                                  int nparts, void *stream);

/*
 * Interpolation written straight into the row buffer of the following GEMM
 * (FeaturePropagation.forward, models/pointnet2_utils.py:191-203: interpolate + concatenate):
 * out[(b,n)][col0 .. col0+C) = sum_k w_k * feat[b, idx[b,n,k], :], w as in pcb_interpolate.
 * feat [B,S,C], out [B*N, ld] in the row type; C/ld/col0 multiples of 8 (bf16) / 4 (fp32);
 * out_w [B,N,k] fp32 (optional).  The fp32 form rounds like pcb_interpolate (product, then sum).
 */
int pcb_interpolate_bf16(const void *feat, const float *d2, const int64_t *idx, int B, int N, int S, int C,
                         int k, void *out, int ld, int col0, float *out_w, void *stream);
int pcb_interpolate_rows_f32(const void *feat, const float *d2, const int64_t *idx, int B, int N, int S, int C,
                             int k, void *out, int ld, int col0, float *out_w, void *stream);
/* The same with FeaturePropagation's whole torch.cat([points1, interpolated], dim=-1) (:201 / :272) in one launch:
 * out[row][0 .. d1) = skip[row][0 .. d1) (skip rows skip_ld elements apart, the row type), out[row][d1 .. col0) = 0. */
int pcb_interpolate_skip_bf16(const void *feat, const float *d2, const int64_t *idx, int B, int N, int S, int C, int k,
                              void *out, int ld, int col0, float *out_w, const void *skip, int skip_ld, int d1,
                              void *stream);
int pcb_interpolate_rows_skip_f32(const void *feat, const float *d2, const int64_t *idx, int B, int N, int S, int C, int k,
                                  void *out, int ld, int col0, float *out_w, const void *skip, int skip_ld, int d1,
                                  void *stream);

/*
 * Backward of the interpolation without atomics: an inverted index of the (n,q) pairs by target
 * s (counting sort: pcb_interp_csr_count -> prefix sum by the caller -> pcb_interp_csr_fill), then
 * one wave per target sums its contribution rows (pcb_interpolate_bwd_csr_*).
 *   count/cursor [B*S] int32 zeroed by the caller; offsets [B*S+1] int64 = exclusive prefix sum of
 *   count; entries [B*N*k] int32; grad_rows [B*N, ld] (columns col0..col0+C); w [B,N,k] fp32;
 *   grad_feat [B,S,C] (overwritten), both in the row type.
 */
int pcb_interp_csr_count(const int64_t *idx, int B, int N, int S, int k, int *count, void *stream);
int pcb_interp_csr_fill(const int64_t *idx, int B, int N, int S, int k, const long *offsets, int *cursor,
                        int *entries, void *stream);
int pcb_interpolate_bwd_csr_bf16(const void *grad_rows, int ld, int col0, const float *w, const long *offsets,
                                 const int *entries, int B, int N, int S, int C, int k, void *grad_feat,
                                 void *stream);
int pcb_interpolate_bwd_csr_f32(const void *grad_rows, int ld, int col0, const float *w, const long *offsets,
                                const int *entries, int B, int N, int S, int C, int k, void *grad_feat,
                                void *stream);

/* Channel-attention gate of EnhancedFeaturePropagation (models/pointnet2_utils.py:279-280):
 * out = x * sigmoid(a) on n row elements (n % 8 == 0 bf16, n % 4 == 0 fp32), one pass; backward
 * dx = g*sigmoid(a), da = g*x*s*(1-s), one pass. */
int pcb_gate_bf16(const void *x, const void *a, void *out, long n, void *stream);
int pcb_gate_f32(const void *x, const void *a, void *out, long n, void *stream);
int pcb_gate_bwd_bf16(const void *g, const void *x, const void *a, void *dx, void *da, long n, void *stream);
int pcb_gate_bwd_f32(const void *g, const void *x, const void *a, void *dx, void *da, long n, void *stream);

/*
 * Global max-pool over the points of a scene on channels-last rows.  Replaces
 * F.adaptive_max_pool1d(x, 1) in DGCNN.forward, models/DGCNN.py:160 (ATen: a reduction returning int64
 * indices; in backward a zero-fill plus a scatter through them).
 *   rows [B*N, C] (row type), out [B, C] (row type), arg [B, C] int32 = the row n attaining the maximum,
 *   the LOWEST such row on ties; workspace: pcb_scene_max_workspace(B, C) bytes of scratch.
 *   backward: dz[b*N + n, c] = (n == arg[b, c]) ? g[b, c] : 0, written densely (dz [B*N, C], g [B, C], row type).
 * C % 8 == 0 (bf16) / C % 4 == 0 (fp32).
 */
long pcb_scene_max_workspace(int B, int C);
/*
 * The pooled per-scene vector joined back to the per-point rows (models/DGCNN.py:160-164: expand the
 * pooled feature over the points and torch.cat it with the local features):
 *   pcb_scene_concat_*  out[b*N + n, :] = [a[b*N + n, 0:C1] | g[b, 0:C2]]     (a [B*N,C1], g [B,C2], out [B*N,C1+C2])
 *   pcb_scene_colsum_*  backward of the broadcast: out[b, c] = sum_n d[b*N + n, col0 + c] for d [B*N, ld];
 *                       two stages through `workspace` (pcb_scene_colsum_workspace(B,N,C) bytes), fixed order,
 *                       no atomics and no memset.
 * (ATen runs the backward as a two-stage reduction with a semaphore buffer; captured in a hipGraph that
 * reduction returned garbage on every replay but the first -- tools/graph_reduce_repro.py, torch-only.)
 */
long pcb_scene_colsum_workspace(int B, int N, int C);
int pcb_scene_concat_bf16(const void *a, const void *g, int B, int N, int C1, int C2, void *out, void *stream);
int pcb_scene_concat_f32(const void *a, const void *g, int B, int N, int C1, int C2, void *out, void *stream);
int pcb_scene_colsum_bf16(const void *d, int B, int N, int ld, int col0, int C, void *out, float *workspace, void *stream);
int pcb_scene_colsum_f32(const void *d, int B, int N, int ld, int col0, int C, void *out, float *workspace, void *stream);
int pcb_scene_max_bf16(const void *rows, int B, int N, int C, void *out, int *arg, void *workspace, void *stream);
int pcb_scene_max_f32(const void *rows, int B, int N, int C, void *out, int *arg, void *workspace, void *stream);
int pcb_scene_max_bwd_bf16(const void *g, const int *arg, int B, int N, int C, void *dz, void *stream);
int pcb_scene_max_bwd_f32(const void *g, const int *arg, int B, int N, int C, void *dz, void *stream);

/*
 * Feature levels concatenated along channels with nearest repetition of the coarser ones -- replaces
 * MultiScaleFeatureFusion's F.interpolate(level, size=N) + torch.cat (models/model.py:150-170) for levels of
 * S_l rows per scene with r_l = N / S_l a power of two (nearest source of fine row i: coarse row i / r_l):
 *   pcb_repeat_concat_*      out[i, col_l + c] = src_l[i / rep[l], c]          out [rows, sum width], one pass
 *   pcb_repeat_concat_bwd_*  dsrc_l[s, c] = sum_{j < rep[l]} g[s*rep[l] + j, col_l + c]   (fp32 sums, one pass;
 *                            a NULL dsrc_l is skipped)
 * n <= 4 levels, src_l [rows / rep[l], width[l]] rows of the entry point's type, width[l] % 8 == 0 (bf16) /
 * % 4 == 0 (fp32), rows % rep[l] == 0 (scene boundaries need no argument: rep[l] divides the rows per scene).
 */
int pcb_repeat_concat_bf16(int n, const void *const *src, const int *rep, const int *width, long rows, void *out, void *stream);
int pcb_repeat_concat_f32(int n, const void *const *src, const int *rep, const int *width, long rows, void *out, void *stream);
int pcb_repeat_concat_bwd_bf16(int n, const void *g, const int *rep, const int *width, long rows, void *const *dsrc, void *stream);
int pcb_repeat_concat_bwd_f32(int n, const void *g, const int *rep, const int *width, long rows, void *const *dsrc, void *stream);

/*
 * Global scaled-dot-product attention, forward -- row f4: F.scaled_dot_product_attention(q, k, v) of
 * PointAttention.forward (models/PointTransformerV3.py:64-117, the call at :102; inference_ptv3.py:101-105: embed 384,
 * 2 heads, head_dim 192), every point of a scene attending to every point of it.
 *   qkv  [B, N, 3, H, D] bf16: the output of the reference's qkv projection as it stands (:96) -- q, k, v of head h of
 *        token n at offsets 0, H*D, 2*H*D (+ h*D) of the token's 3*H*D values
 *   out  [B, N, H*D] bf16 = softmax(q k^T * scale) v, heads side by side: the layout after the reference's
 *        transpose(1, 2).reshape(B, N, C) (:113)
 * One flash-attention pass (online softmax in fp32, bf16 MFMA products, nothing of size N x N stored).  D in
 * {64, 128, 192, 256}; scale = D^-0.5 in the reference.  Inference configuration: no backward.
 */
int pcb_attention_fwd_bf16(const void *qkv, int B, int N, int H, int D, float scale, void *out, void *stream);

/*
 * Row kernels of the transformer token pipeline around it (models/PointTransformerV3.py:119-148 PointTransformerBlock,
 * :8-21 GEGLU), bf16 rows [R, C], inference:
 *   pcb_add_layernorm_bf16   x' = x + h (h optional: the pending residual; xout, optional, receives x' -- the new residual
 *                            stream, rounded to bf16); out = LayerNorm(x') * gamma + beta (+ pos, optional): `x = x +
 *                            attn(norm1(x) + pos)` / `x = x + mlp(norm2(x))` without separate add / cast / normalise passes.
 *                            fp32 statistics over the row (biased variance, eps as nn.LayerNorm), C a multiple of 8, <= 1024.
 *   pcb_geglu_bf16           out [R, H] = y[:, :H] * gelu(y[:, H:]) for the projection rows y [R, 2H] (erf form: F.gelu's
 *                            default), H a multiple of 8.
 */
int pcb_add_layernorm_bf16(const void *x, const void *h, const void *pos, const float *gamma, const float *beta, float eps,
                           long R, int C, void *xout, void *out, void *stream);
int pcb_geglu_bf16(const void *y, long R, int H, void *out, void *stream);

/*
 * A whole stack of L <= PCB_STACK_MAX_LAYERS shared-MLP layers  x -> act(BN(x W^T + b))  [-> max over
 * each `pool` consecutive rows]  enqueued from ONE call -- the loop the reference writes as
 *   for i, conv in enumerate(self.mlp_convs): new_points = F.relu(self.mlp_bns[i](conv(new_points)))
 *   new_points = torch.max(new_points, 2)[0]
 * (models/pointnet2_utils.py:149-154, :207-209, :353-356; models/DGCNN.py:134-148).  The calls are
 * exactly the sequence of pcb_prep_weights_* / pcb_gemm_nt_* / pcb_bn_finalize /
 * pcb_bn_act(_max)_* (forward) and pcb_bn_act(_max)_bwd_reduce_* / pcb_bn_bwd_finalize /
 * pcb_gemm_tn_* / pcb_gemm_nt(_red)_* (backward) a caller would issue itself; issuing them
 * from native code keeps the host ahead of the GPU (one foreign call per stack and direction).
 * dtype selects the row type (pcb_dtype) of x, y, out, g, dx, dzbuf and wbuf; the sequence is the same.
 *
 * desc: L x PCB_STACK_DESC_SLOTS (18) int64 on the HOST, per layer
 *   [0] w fp32 [C,k]  [1] conv bias [C] or 0  [2] gamma or 0  [3] beta or 0
 *   [4] running_mean or 0  [5] running_var or 0  [6] C (multiple of 8 / 4)  [7] k = real input columns
 *   [8] 1: batch statistics (training), 0: running statistics
 *   [9] y [R,C]: the layer's pre-BatchNorm GEMM output (written by forward, read by backward)
 *   [10] dW fp32 [C,k]  [11] dgamma [C]  [12] dbeta [C]  [13] dbias [C]   (backward outputs, any may be 0)
 *   [14] num_batches_tracked (int64 scalar, forward: += 1) or 0
 *   [15] layer 0 of a plain bf16 stack only, else 0: HOST address of 7 int64 --
 *        {add1, sh1, add2 or 0, sh2, d add1, d add2, row stride of w in floats or 0}: the layer's GEMM is
 *        pcb_gemm_nt_stats_add_bf16 (forward), backward also writes d add1 / d add2 where given
 *        (pcb_dy_repeat_sums_bf16); w may be a column slice of a wider weight (its rows `stride` floats apart)
 *   [16] centre: fp32 [C] the caller keeps ACROSS calls for this BatchNorm layer, or 0.  bf16 rows, forward, training
 *        mode: y is stored as bf16(x W^T - centre) and the finalize kernel moves centre to this batch's mean (see
 *        pcb_gemm_nt_stats_bf16 / pcb_bn_finalize_centred); backward needs nothing (every constant it reads lives in
 *        the centred frame).  Ignored for fp32 rows.
 *   [17] centre flags: bit 0 = the buffer holds no estimate yet (zeros): GEMM + finalize run once to learn the batch
 *        mean, then again centred on it (under `sync` both finalize passes all-reduce: the mean is the global one);  bit 1 = an EVAL-mode layer stores y centred on
 *        running_mean - bias (kept in row 0 of its stz block)
 * fdesc: L x 2 doubles: momentum, eps.  stat_repeat >= 1: every row of x stands for that many
 * identical samples (see pcb_bn_finalize `count`); 1 otherwise.
 * x [R,Kp] rows in the column layout `perm` (see pcb_prep_weights_*); act 0/1/2;
 * pool = 0 (out [R,C_last]) or ns (out [R/ns,C_last] + argmax uint8).
 * Caller-owned scratch shared by forward and backward of the same stack:
 *   wbuf  row type, pcb_mlp_stack_wbuf_elems(L,desc,Kp,need_wt0) elements (prepared weights; need_wt0 = the
 *         input gradient dx will be wanted);   stz fp32 [10 * sum C] (per-layer constants);
 *   parts fp32 [parts_slabs][2][max C]: statistics slabs.  No launch of the call writes more than
 *         parts_slabs slabs, whatever the concurrency hint says (a gathered first layer uses up to 1024,
 *         the GEMMs up to 768; fewer slabs only mean fewer resident workgroups).
 * sync (optional): SyncBatchNorm.  After each training-mode layer's statistics the local totals
 *   [2][C] are handed to sync->allreduce(buf, n, ctx) -- called on the host, between launches; it must
 *   enqueue a SUM all-reduce of the n floats at device pointer buf in stream order (return 0) -- and
 *   the layer is normalised with the statistics of sync->global_rows rows (all ranks' R).  Backward:
 *   the two BatchNorm-backward sums travel the same way; parameter gradients stay local.
 * Forward, need_wt0 bit 2 (value 4): wbuf already holds the operands of the CURRENT weights (prepared by
 * pcb_prep_weights_table_* since the last update): the preparation launch is skipped, nothing else changes.
 * Forward, need_wt0 bit 1 (value 2): every layer is in eval mode and wbuf / stz still hold what an
 * earlier call with the same, unchanged parameters and running statistics left there -- operand
 * preparation and the per-layer BatchNorm finalize are skipped (inference with constant weights).
 * Backward only: g = dz [R,C_last] (pool 0) or dout fp32 [R/ns,C_last]; workspace fp32, >= the
 * SUM of pcb_gemm_tn_workspace(R,C_l,Kp_l) over the layers that have a dW (each keeps its slabs until
 * one launch at the end of the pass sums them all); dzbuf: pcb_mlp_stack_dzbuf_elems(...) elements of the row type (two
 * gradient slots [R][max(8, Kp, inner widths)], wider where a top layer's dy is written out once; 0 = not needed: NULL);
 * dx [R,Kp] or NULL.
 */
#define PCB_STACK_MAX_LAYERS 16
#define PCB_STACK_DESC_SLOTS 18
typedef struct pcb_sync {
    int (*allreduce)(float *buf, int n, void *ctx);
    void *ctx;
    long global_rows;
} pcb_sync;
long pcb_mlp_stack_wbuf_elems(int L, const long long *desc, int Kp, int need_wt0);
long pcb_mlp_stack_dzbuf_elems(int dtype, int L, const long long *desc, long R, int Kp, int pool, int gathered);
int pcb_mlp_stack_forward(int dtype, int L, const long long *desc, const double *fdesc, const void *x, long R,
                          int Kp, int perm, int act, int pool, int need_wt0, int stat_repeat,
                          const long long *gather, void *wbuf, float *stz, float *parts, int parts_slabs,
                          const pcb_sync *sync, void *out, unsigned char *argmax, void *stream);
int pcb_mlp_stack_backward(int dtype, int L, const long long *desc, const void *x, const void *g,
                           const unsigned char *argmax, long R, int Kp, int perm, int act, int pool,
                           int need_wt0, const long long *gather, const void *wbuf, float *stz, float *parts,
                           int parts_slabs, const pcb_sync *sync, float *workspace, void *dzbuf, void *dx,
                           void *stream);

/*
 * First layer of a GROUPED stack evaluated per point and gathered (csrc/gatherlin.hip): a 1x1
 * convolution of [x_j - c_s | f_j] (sample_and_group, models/pointnet2_utils.py:51-58) or
 * [x_j - x_i | x_i] (get_graph_feature, models/DGCNN.py:90-107) equals u[idx(s,j)] + v[s] with the
 * per-point products u (N rows per scene) and v (S rows per scene), both fp32 [.,C].
 *   pcb_gather_add_bf16   y[r,:] = bf16(u[b*N + idx[r],:] + v[r/ns,:] + Wx (xyz[b*N + idx[r]] - ctr[r/ns])),
 *                         r over B*S*ns rows, and the column sums / sums of squares of y into
 *                         [nparts][2][C] slabs for pcb_bn_finalize: the launch runs exactly nparts
 *                         workgroups (1..1024; pcb_gather_add_partials(R,C) is the recommendation).  v may be
 *                         NULL; the coordinate term (the reference's fp32 difference x_j - c_s times
 *                         the first 3 weight columns, wx [C,3] with row stride ldw) is used when wx
 *                         is not NULL (xyz [B,N,3], ctr [B,S,3] fp32).
 *   pcb_scatter_dy_bf16   backward: dy = the layer's BatchNorm/activation backward built from
 *                         (dz, y) or, pooled != 0, from (dout, argmax, y) of a max over the same ns
 *                         rows, with scale/shift/p/q as in pcb_gemm_nt_bf16 pro 2/3;
 *                         du[b*N + idx[r],:] += dy[r,:] (fp32 atomics, du zeroed by the caller),
 *                         dv[r/ns,:] = sum over the group (overwritten; may be NULL),
 *                         det != 0: no atomics -- du is not touched (may be NULL: pcb_scatter_dy_csr_bf16 computes it)
 *                         and dwx has pcb_scatter_dy_slabs(B,S,C,1) slabs;
 *                         dwx (may be NULL): fp32 [33][C][3], all zeroed by the caller; slab 0 receives
 *                         dWx[c,:] = sum_r dy[r,c] (x_j - c_s), slabs 1..32 are scratch (the atomic adds
 *                         are spread over them and summed at the end).
 * The stack calls take these through `gather` (NULL = ordinary stack): 16 int64 on the host,
 *   {u | du, v | dv, idx, B, N, S, ns, xyz, ctr, wx | dwx, ldw, flags (bit 0: det), order, offsets, 0, 0}  (forward reads u, v, wx; backward
 *   writes du, dv, dwx ([33][C][3], result in slab 0) and zeroes du, dwx first);
 * layer 0 of desc then carries no weight (slots [0],[7],[10] unused) and x, Kp, perm, dx are ignored.
 */
int pcb_gather_add_partials(long R, int C);
int pcb_gather_add_bf16(const float *u, const float *v, const int64_t *idx, int B, int N, int S, int ns, int C,
                        const float *xyz, const float *ctr, const float *wx, int ldw, void *y, float *sums,
                        int nparts, const float *centre, void *stream);
int pcb_scatter_dy_bf16(int pooled, const void *dz, const void *y, const float *scale, const float *shift,
                        const float *p, const float *q, const float *dout, const unsigned char *argmax, int act,
                        const int64_t *idx, int B, int N, int S, int ns, int C, const float *xyz,
                        const float *ctr, float *du, float *dv, float *dwx, int det, void *stream);
/* Slabs of the dwx buffer pcb_scatter_dy_bf16 needs ([slabs][C][3] floats, slab 0 = the result): 33, or -- det != 0 -- one
 * per workgroup of the launch + 1. */
long pcb_scatter_dy_slabs(int B, int S, int C, int det);

/*
 * Reproducible backward passes (no float atomics): the scatter-adds of the gathers' gradients -- the reference's
 * index_put_(accumulate=True) behind index_points (models/pointnet2_utils.py:17-39), the grouping (:51-58, :342-349) and
 * DGCNN.get_graph_feature (models/DGCNN.py:90-107) -- as segment sums over an inverted index in a FIXED order.
 *   order [E] int32, offsets [T+1] int64: for target row t the source rows order[offsets[t] .. offsets[t+1]) that read
 *   it, ascending (a stable sort of the targets; the Python side builds it with torch.sort, ops.det_index).
 *   pcb_segment_sum_*       out[t, 0:C] (fp32, rows out_ld floats apart) = (accumulate ? out[t] : 0) + sum over the
 *                           segment of rows[order[e]*ld + col0 + c], added in index order (rows fp32 or bf16)
 *   pcb_scatter_dy_csr_bf16 du[t,:] = sum over the segment of dy[r,:], dy as pcb_scatter_dy_bf16 builds it; with
 *                           pcb_scatter_dy_bf16(det = 1) (dv, dWx through per-workgroup slabs, du untouched / NULL) the
 *                           backward of a gathered first layer without atomics.  The stack calls take det / order / offsets
 *                           through slots [11] (bit 0), [12], [13] of `gather` (16 int64 then).
 */
int pcb_segment_sum_f32(const float *rows, long ld, int col0, int C, const int *order, const long long *offsets,
                        long targets, float *out, long out_ld, int accumulate, void *stream);
int pcb_segment_sum_bf16(const void *rows, long ld, int col0, int C, const int *order, const long long *offsets,
                         long targets, float *out, long out_ld, int accumulate, void *stream);
int pcb_scatter_dy_csr_bf16(int pooled, const void *dz, const void *y, const float *scale, const float *shift,
                            const float *p, const float *q, const float *dout, const unsigned char *argmax, int act,
                            int ns, int C, const int *order, const long long *offsets, long targets, float *du,
                            void *stream);

/* Raw fp32 input columns x [R, k] (rows `ld` floats apart: coordinates, colours) as a zero-padded operand
 * out [R, kp] of the row type (kp a multiple of 8 for bf16, 4 for fp32): cast + pad in one pass. */
int pcb_pad_rows_bf16(const float *x, long ld, long R, int k, int kp, void *out, void *stream);
int pcb_pad_rows_f32(const float *x, long ld, long R, int k, int kp, void *out, void *stream);

/* Many device-to-device copies in one launch: table in DEVICE memory, n rows of 4 int64 {dst, src, bytes (multiple of
 * 4; addresses 4-byte aligned), first workgroup}; a copy takes ceil(bytes / 16384) workgroups, rows in ascending order,
 * `blocks` = their total.  (The captured training step hands the next step's sampling results from its staging to its
 * live buffers with it: StaticSampling.commit.) */
int pcb_copy_table(const long long *table, int n, long blocks, void *stream);
/* The same with the copies given as HOST arrays of n addresses / byte counts (passed to the kernel by value, 32 per
 * launch): for source buffers whose addresses are only known when the call is made -- e.g. results allocated inside a
 * captured step (StaticSampling.compute hands ball-query / three_nn / CSR results to its staging set with it). */
int pcb_copy_list(const long long *dst, const long long *src, const long long *bytes, int n, void *stream);

/*
 * Per-point cross entropy of the segmentation trainers (`criterion = nn.CrossEntropyLoss()` on [B,C,N]
 * logits vs [B,N] labels, train_MulSca_PN2.py:161; on [B*N,C] in train_DGCNN.py:177-197), mean over the
 * points whose label is not ignore_index, as one pass over the logits ROWS: logits = R rows of C <= 64
 * fp32 values, `ld` floats apart (the network's last GEMM output, read in place); labels [R] int64.
 * fwd: partials = 2 * pcb_cross_entropy_partials(R) floats of scratch; loss_count[0] = the loss,
 * loss_count[1] = number of counted points.  bwd: dlogits [R,C] contiguous =
 * (softmax - onehot) * grad_out[0] / count (zero rows for ignored points); grad_out = one float on the device.
 * Kernels only (no atomics, no memset): valid inside a captured hipGraph.
 */
int pcb_cross_entropy_partials(long R);
int pcb_cross_entropy_fwd(const float *logits, long ld, const int64_t *labels, long R, int C, long ignore_index,
                          float *partials, float *loss_count, void *stream);
int pcb_cross_entropy_bwd(const float *logits, long ld, const int64_t *labels, long R, int C, long ignore_index,
                          const float *loss_count, const float *grad_out, float *dlogits, void *stream);

/*
 * The criterion of the reference's BridgeSeg trainer, BridgeStructureLoss (models/model.py:169-260;
 * train_MulSca_BriStruNet_CB.py:151-156, :178), as kernels -- its torch form runs ~40 reductions over [B,N] masks per step.
 *   pcb_bridge_loss_weights   the 5 class weights of the step: per scene, from the PREDICTED labels (arg-max of the logits
 *                             rows [B*N, >=5], `ld` floats apart) and the z coordinates of points [B,N,3], the mean relative
 *                             height of each predicted component class (:189-196) and the order violations between them
 *                             (:218-251), averaged over the scenes, times 1/sqrt(label frequency) * {1,2,1,1,2}
 *                             (:253-256).  stats: scratch [B][22] floats; weights [5] out; base_weights [5] in.  B <= 1024.
 *   pcb_cross_entropy_w_fwd / _bwd   F.cross_entropy(logits, labels, weight=w, label_smoothing=eps) (:258-260): loss =
 *                             sum_i [(1-eps) w[y_i] (-logp_i[y_i]) + eps/C sum_c w[c] (-logp_i[c])] / sum_i w[y_i];
 *                             loss_wsum [2] = (loss, sum of the label weights); partials as pcb_cross_entropy_fwd.
 */
int pcb_bridge_loss_weights(const float *logits, long ld, const int64_t *labels, const float *points, int B, int N,
                            float alpha, float rel_margin, const float *base_weights, float *stats, float *weights,
                            void *stream);
int pcb_cross_entropy_w_fwd(const float *logits, long ld, const int64_t *labels, long R, int C, long ignore_index,
                            const float *weight, float smoothing, float *partials, float *loss_wsum, void *stream);
int pcb_cross_entropy_w_bwd(const float *logits, long ld, const int64_t *labels, long R, int C, long ignore_index,
                            const float *weight, float smoothing, const float *loss_wsum, const float *grad_out,
                            float *dlogits, void *stream);

/*
 * Instruments of bench.py (off unless armed; the only other process-wide state besides the hint):
 *  - HIP-event timing, on the stream they are launched on, of two kernel categories:
 *      0  the gemm_nt family (pcb_gemm_nt_* / pcb_gemm_nt_red_* / _bias / _f32out, also when issued
 *         by the stack calls) -- the roofline kernel;   1  pcb_fps -- the step's latency chain.
 *  - a byte counter: every entry point adds the ALGORITHMIC HBM bytes of the launch it enqueued
 *    (operands read once + outputs written once; for gemm_nt: A operand as read by its prologue +
 *    output [+ y for the RED variant]) -- the whole-step traffic figure.
 * pcb_timer_start arms and clears both; pcb_timer_enable(0/1) pauses/resumes the EVENT sampling (the
 * byte counter keeps counting); pcb_timer_stop disarms, synchronises the recorded events and returns
 * category 0: launches, their summed duration, their bytes.  pcb_timer_read(category, ...) afterwards:
 * the same for category 0 or 1; category -1: every accounted launch and its bytes (duration 0).
 */
int pcb_timer_start(void);
int pcb_timer_enable(int on);
int pcb_timer_stop(long *launches, double *milliseconds, double *bytes);
int pcb_timer_read(int category, long *launches, double *milliseconds, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* PCB_HIP_H */
