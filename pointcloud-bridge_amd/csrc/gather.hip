// Row gathers and their scatter-add backwards for gfx950 (HBM-bound byte movers).
//
// Replace the advanced-indexing compositions of the reference:
//   index_points                      models/pointnet2_utils.py:17-39   (gather_rows)
//   grouping in sample_and_group /
//   MultiScaleSetAbstraction          models/pointnet2_utils.py:51-58, :342-349   (group_points)
//   DGCNN.get_graph_feature           models/DGCNN.py:90-107            (edge_features)
// One lane per output float with the channel index fastest, so a wave reads and writes whole
// contiguous row segments; the index of a row is read once per lane from L1/L2.
// Backwards accumulate with fp32 atomics at memory (MI355X_MICROARCH "Global float atomics").
#include "pcb_common.h"

namespace {

inline int grid_for(size_t total)
{
    size_t blocks = (total + 255) / 256;
    return (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

#define PCB_GRID_STRIDE(e, total)                                                  \
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (total);    \
         e += (size_t)gridDim.x * blockDim.x)

__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ points,
                                                           const int64_t *__restrict__ idx, int N,
                                                           int C, int M, float *__restrict__ out,
                                                           size_t total)
{
    PCB_GRID_STRIDE(e, total)
    {
        const int c = (int)(e % C);
        const size_t row = e / C;  // b*M + m
        const size_t b = row / M;
        const int j = clamp_index(idx[row], N);
        out[e] = points[(b * N + j) * C + c];
    }
}

__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float *__restrict__ g,
                                                               const int64_t *__restrict__ idx,
                                                               int N, int C, int M,
                                                               float *__restrict__ gpoints,
                                                               size_t total)
{
    PCB_GRID_STRIDE(e, total)
    {
        const int c = (int)(e % C);
        const size_t row = e / C;
        const size_t b = row / M;
        const int j = clamp_index(idx[row], N);
        atomicAdd(&gpoints[(b * N + j) * C + c], g[e]);
    }
}

// out[b,s,j,:] = cat(xyz[b,idx] - new_xyz[b,s], feat[b,idx])
__global__ __launch_bounds__(256) void group_points_kernel(const float *__restrict__ xyz,
                                                            const float *__restrict__ new_xyz,
                                                            const float *__restrict__ feat,
                                                            const int64_t *__restrict__ idx, int N,
                                                            int S, int ns, int C,
                                                            float *__restrict__ out, size_t total)
{
    const int W = 3 + C;
    PCB_GRID_STRIDE(e, total)
    {
        const int c = (int)(e % W);
        const size_t row = e / W;      // (b*S + s)*ns + j
        const size_t bs = row / ns;    // b*S + s
        const size_t b = bs / S;
        const int i = clamp_index(idx[row], N);
        float v;
        if (c < 3)
            v = __fsub_rn(xyz[(b * N + i) * 3 + c], new_xyz[bs * 3 + c]);
        else
            v = feat[(b * N + i) * C + (c - 3)];
        out[e] = v;
    }
}

__global__ __launch_bounds__(256) void group_points_bwd_kernel(const float *__restrict__ g,
                                                                const int64_t *__restrict__ idx,
                                                                int N, int S, int ns, int C,
                                                                float *__restrict__ gfeat,
                                                                size_t total)
{
    const int W = 3 + C;
    PCB_GRID_STRIDE(e, total)  // total = rows * C: only the feature columns carry gradient
    {
        const int c = (int)(e % C);
        const size_t row = e / C;
        const size_t b = row / ((size_t)S * ns);
        const int i = clamp_index(idx[row], N);
        atomicAdd(&gfeat[(b * N + i) * C + c], g[row * W + 3 + c]);
    }
}

// out[b,n,j,:] = cat(x[b,idx[b,n,j]] - x[b,n], x[b,n])
__global__ __launch_bounds__(256) void edge_features_kernel(const float *__restrict__ x,
                                                             const int64_t *__restrict__ idx, int N,
                                                             int D, int K, float *__restrict__ out,
                                                             size_t total)
{
    const int W = 2 * D;
    PCB_GRID_STRIDE(e, total)
    {
        const int c = (int)(e % W);
        const size_t row = e / W;    // (b*N + n)*K + j
        const size_t bn = row / K;   // b*N + n
        const size_t b = bn / N;
        float v;
        if (c < D) {
            const int i = clamp_index(idx[row], N);
            v = __fsub_rn(x[(b * N + i) * D + c], x[bn * D + c]);
        } else {
            v = x[bn * D + (c - D)];
        }
        out[e] = v;
    }
}

// One lane per (b, n, c): the K neighbour slots are walked in the lane, so the centre term is a
// single atomic per (n, c) and only the neighbour terms scatter.
__global__ __launch_bounds__(256) void edge_features_bwd_kernel(const float *__restrict__ g,
                                                                 const int64_t *__restrict__ idx,
                                                                 int N, int D, int K,
                                                                 float *__restrict__ gx,
                                                                 size_t total)
{
    const int W = 2 * D;
    PCB_GRID_STRIDE(e, total)  // total = B*N*D
    {
        const int c = (int)(e % D);
        const size_t bn = e / D;
        const size_t b = bn / N;
        float centre = 0.0f;
        for (int j = 0; j < K; ++j) {
            const size_t row = bn * K + j;
            const float gd = g[row * W + c];       // d/d(x_j - x_i)
            const float gc = g[row * W + D + c];   // d/d(x_i)
            centre += gc - gd;
            const int i = clamp_index(idx[row], N);
            atomicAdd(&gx[(b * N + i) * D + c], gd);
        }
        atomicAdd(&gx[e], centre);
    }
}

}  // namespace

extern "C" int pcb_gather_rows(const float *points, const int64_t *idx, int B, int N, int C, int M,
                               float *out, void *stream)
{
    if (!points || !idx || !out || B <= 0 || N <= 0 || C <= 0 || M <= 0) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * M * C;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       points, idx, N, C, M, out, total);
    pcb_account(8.0 * (double)B * M * C + 8.0 * B * M);
    return pcb_check_launch();
}

extern "C" int pcb_gather_rows_bwd(const float *grad_out, const int64_t *idx, int B, int N, int C,
                                   int M, float *grad_points, void *stream)
{
    if (!grad_out || !idx || !grad_points || B <= 0 || N <= 0 || C <= 0 || M <= 0) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * M * C;
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, grad_out, idx, N, C, M, grad_points, total);
    pcb_account(8.0 * (double)B * M * C + 8.0 * B * M);
    return pcb_check_launch();
}

extern "C" int pcb_group_points(const float *xyz, const float *new_xyz, const float *feat,
                                const int64_t *idx, int B, int N, int S, int ns, int C, float *out,
                                void *stream)
{
    if (!xyz || !new_xyz || !idx || !out || B <= 0 || N <= 0 || S <= 0 || ns <= 0 || C < 0) return PCB_ERR_INVALID_ARG;
    if (C > 0 && !feat) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * S * ns * (3 + C);
    hipLaunchKernelGGL(group_points_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       xyz, new_xyz, feat, idx, N, S, ns, C, out, total);
    pcb_account(8.0 * (double)B * S * ns * (3 + C) + 8.0 * B * S * ns);
    return pcb_check_launch();
}

extern "C" int pcb_group_points_bwd(const float *grad_out, const int64_t *idx, int B, int N, int S,
                                    int ns, int C, float *grad_feat, void *stream)
{
    if (!grad_out || !idx || !grad_feat || B <= 0 || N <= 0 || S <= 0 || ns <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * S * ns * C;
    hipLaunchKernelGGL(group_points_bwd_kernel, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, grad_out, idx, N, S, ns, C, grad_feat, total);
    pcb_account(8.0 * (double)B * S * ns * C + 8.0 * B * S * ns);
    return pcb_check_launch();
}

extern "C" int pcb_edge_features(const float *x, const int64_t *idx, int B, int N, int D, int k,
                                 float *out, void *stream)
{
    if (!x || !idx || !out || B <= 0 || N <= 0 || D <= 0 || k <= 0) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * N * k * 2 * D;
    hipLaunchKernelGGL(edge_features_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       x, idx, N, D, k, out, total);
    pcb_account(4.0 * (double)B * N * D + 8.0 * (double)B * N * k + 8.0 * (double)B * N * k * D);
    return pcb_check_launch();
}

extern "C" int pcb_edge_features_bwd(const float *grad_out, const int64_t *idx, int B, int N, int D,
                                     int k, float *grad_x, void *stream)
{
    if (!grad_out || !idx || !grad_x || B <= 0 || N <= 0 || D <= 0 || k <= 0) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * N * D;
    hipLaunchKernelGGL(edge_features_bwd_kernel, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, grad_out, idx, N, D, k, grad_x, total);
    pcb_account(12.0 * (double)B * N * k * D + 8.0 * (double)B * N * k);
    return pcb_check_launch();
}
