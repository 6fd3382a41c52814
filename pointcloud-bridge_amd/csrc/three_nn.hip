// k-nearest centroids (k <= 4) and inverse-distance interpolation for gfx950.
//
// Replaces, in FeaturePropagation.forward / EnhancedFeaturePropagation.forward of the reference
// (models/pointnet2_utils.py:185-196 and :253-267): square_distance -> [B,N,S] fp32, a FULL sort of
// it (values + int64 indices), a slice of the first k, the weight computation and the gather-sum.
// three_nn: one lane per query point, candidates staged once per workgroup through LDS as
// (x, y, z, |p|^2) float4 and read back as wave-wide broadcasts; the k best live in registers.
// Equal distances keep ascending index order (the reference's CPU sort is stable).
#include "pcb_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 1024;  // candidates per LDS tile (16 KiB)

template <int K>
__global__ __launch_bounds__(kThreads) void three_nn_kernel(const float *__restrict__ xyz1,
                                                             const float *__restrict__ xyz2, int N,
                                                             int S, float *__restrict__ out_d,
                                                             int64_t *__restrict__ out_i)
{
    __shared__ float4 tile[kTile];
    const int b = blockIdx.y;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    const bool valid = n < N;
    const float *__restrict__ q = xyz1 + ((size_t)b * N + (valid ? n : N - 1)) * 3;
    const float *__restrict__ c = xyz2 + (size_t)b * S * 3;
    const float sx = q[0], sy = q[1], sz = q[2];
    const float s2 = sq_norm3(sx, sy, sz);

    float bd[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        bd[k] = INFINITY;
        bi[k] = 0;
    }

    for (int base = 0; base < S; base += kTile) {
        const int cnt = min(kTile, S - base);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += kThreads) {
            const float x = c[(base + j) * 3 + 0];
            const float y = c[(base + j) * 3 + 1];
            const float z = c[(base + j) * 3 + 2];
            tile[j] = make_float4(x, y, z, sq_norm3(x, y, z));
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float4 t = tile[j];
            const float d = sqdist_expand(sx, sy, sz, s2, t.x, t.y, t.z, t.w);
            if (d < bd[K - 1]) {
                bd[K - 1] = d;
                bi[K - 1] = base + j;
#pragma unroll
                for (int k = K - 1; k > 0; --k) {
                    // strict "<": a later candidate never overtakes an equal earlier one
                    if (bd[k] < bd[k - 1]) {
                        const float td = bd[k];
                        bd[k] = bd[k - 1];
                        bd[k - 1] = td;
                        const int ti = bi[k];
                        bi[k] = bi[k - 1];
                        bi[k - 1] = ti;
                    }
                }
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            out_d[((size_t)b * N + n) * K + k] = bd[k];
            out_i[((size_t)b * N + n) * K + k] = (int64_t)bi[k];
        }
    }
}

// out[b,n,c] = sum_k w_k * feat[b, idx_k, c]; one lane per (n, c), c fastest (coalesced rows).
template <int K>
__global__ __launch_bounds__(256) void interpolate_kernel(const float *__restrict__ feat,
                                                           const float *__restrict__ d2,
                                                           const int64_t *__restrict__ idx, int N,
                                                           int S, int C, float *__restrict__ out,
                                                           float *__restrict__ out_w, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t row = e / C;  // b*N + n
        const size_t b = row / N;
        float w[K];
        float norm = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            w[k] = __fdiv_rn(1.0f, __fadd_rn(d2[row * K + k], 1e-8f));
            norm = k ? __fadd_rn(norm, w[k]) : w[k];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            w[k] = __fdiv_rn(w[k], norm);
            const int j = clamp_index(idx[row * K + k], S);
            const float v = __fmul_rn(feat[(b * S + j) * C + c], w[k]);
            acc = k ? __fadd_rn(acc, v) : v;
        }
        out[e] = acc;
        if (out_w && c == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) out_w[row * K + k] = w[k];
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void interpolate_bwd_kernel(const float *__restrict__ g,
                                                               const float *__restrict__ w,
                                                               const int64_t *__restrict__ idx,
                                                               int N, int S, int C,
                                                               float *__restrict__ gfeat,
                                                               size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t row = e / C;
        const size_t b = row / N;
        const float go = g[e];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = clamp_index(idx[row * K + k], S);
            atomicAdd(&gfeat[(b * S + j) * C + c], w[row * K + k] * go);
        }
    }
}

inline int grid_for(size_t total)
{
    size_t blocks = (total + 255) / 256;
    return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

}  // namespace

extern "C" int pcb_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int k,
                            float *out_d2, int64_t *out_idx, void *stream)
{
    if (!xyz1 || !xyz2 || !out_d2 || !out_idx || B <= 0 || N <= 0 || S <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 4 || S < k) return PCB_ERR_INVALID_ARG;
    const dim3 grid((N + kThreads - 1) / kThreads, B);
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL((three_nn_kernel<1>), grid, dim3(kThreads), 0, st, xyz1, xyz2, N, S, out_d2, out_idx); break;
        case 2: hipLaunchKernelGGL((three_nn_kernel<2>), grid, dim3(kThreads), 0, st, xyz1, xyz2, N, S, out_d2, out_idx); break;
        case 3: hipLaunchKernelGGL((three_nn_kernel<3>), grid, dim3(kThreads), 0, st, xyz1, xyz2, N, S, out_d2, out_idx); break;
        default: hipLaunchKernelGGL((three_nn_kernel<4>), grid, dim3(kThreads), 0, st, xyz1, xyz2, N, S, out_d2, out_idx); break;
    }
    return pcb_check_launch();
}

extern "C" int pcb_interpolate(const float *feat, const float *d2, const int64_t *idx, int B, int N,
                               int S, int C, int k, float *out, float *out_w, void *stream)
{
    if (!feat || !d2 || !idx || !out || B <= 0 || N <= 0 || S <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 4) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * N * C;
    const dim3 grid(grid_for(total));
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL((interpolate_kernel<1>), grid, dim3(256), 0, st, feat, d2, idx, N, S, C, out, out_w, total); break;
        case 2: hipLaunchKernelGGL((interpolate_kernel<2>), grid, dim3(256), 0, st, feat, d2, idx, N, S, C, out, out_w, total); break;
        case 3: hipLaunchKernelGGL((interpolate_kernel<3>), grid, dim3(256), 0, st, feat, d2, idx, N, S, C, out, out_w, total); break;
        default: hipLaunchKernelGGL((interpolate_kernel<4>), grid, dim3(256), 0, st, feat, d2, idx, N, S, C, out, out_w, total); break;
    }
    return pcb_check_launch();
}

extern "C" int pcb_interpolate_bwd(const float *grad_out, const float *w, const int64_t *idx, int B,
                                   int N, int S, int C, int k, float *grad_feat, void *stream)
{
    if (!grad_out || !w || !idx || !grad_feat || B <= 0 || N <= 0 || S <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 4) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * N * C;
    const dim3 grid(grid_for(total));
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL((interpolate_bwd_kernel<1>), grid, dim3(256), 0, st, grad_out, w, idx, N, S, C, grad_feat, total); break;
        case 2: hipLaunchKernelGGL((interpolate_bwd_kernel<2>), grid, dim3(256), 0, st, grad_out, w, idx, N, S, C, grad_feat, total); break;
        case 3: hipLaunchKernelGGL((interpolate_bwd_kernel<3>), grid, dim3(256), 0, st, grad_out, w, idx, N, S, C, grad_feat, total); break;
        default: hipLaunchKernelGGL((interpolate_bwd_kernel<4>), grid, dim3(256), 0, st, grad_out, w, idx, N, S, C, grad_feat, total); break;
    }
    return pcb_check_launch();
}
