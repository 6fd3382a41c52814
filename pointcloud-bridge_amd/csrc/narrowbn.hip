// BatchNorm (+ ReLU) on NARROW fp32 rows [R, C], any 1 <= C <= 64, and the per-scene column mean -- for the colour /
// fusion stacks of the reference's BridgeSeg network (models/attention_modules.py:696-716 color_mlp / color_attention,
// :718-722 color_context's AdaptiveAvgPool1d, :759-764 fusion_mlp): BatchNorm1d over 3, 6 or 16 channels on B*N = 262144
// rows.  The row engine (rowbn.hip) moves 16-byte vectors and wants C % 4 == 0; through ATen these layers cost four to
// six launches each, and ATen's two-stage reductions (staging buffer + semaphore) do not survive hipGraph replay on this
// stack (tools/graph_reduce_repro.py) -- which kept the whole BridgeSeg step from being captured.  Here:
//   rows_stats        per-workgroup slabs [nparts][2][C] of sum x, sum x^2         (pcb_bn_finalize adds them in order)
//   rows_bn_act       z = act(x*scale + shift)
//   rows_bwd_reduce   slabs of (sum du, sum du*xhat), du = dz*act'(x*scale+shift)
//   rows_bwd_apply    dx = scale*(du - s1/R - xhat*s2/R)  (batch statistics)  |  scale*du  (running statistics)
//   scene_sum         slabs [nparts][B][C] of the column sums of every scene's N rows (pcb_sum_slabs adds them)
// One lane per element, the element index fastest: perfectly coalesced; a lane keeps its channel because the grid stride
// is a multiple of C; lanes of a workgroup that share a channel meet in LDS in a fixed order.  No atomics, kernels only.
#include "pcb_common.h"

namespace {

constexpr int kThreads = 256;

inline float slope_of(int act) { return act == 1 ? 0.0f : (act == 2 ? 0.2f : 1.0f); }
__device__ __forceinline__ float act_fwd(float u, float slope) { return u > 0.0f ? u : fmaf(slope, u, 0.0f); }
__device__ __forceinline__ float act_grad(float u, float slope) { return u > 0.0f ? 1.0f : slope; }

// workgroups of a launch over `total` elements: a multiple of C / gcd(C, 256) (the grid stride is then a multiple of C),
// at most `cap`
inline int grid_for(long total, int C, int cap)
{
    int g = C, t = kThreads;
    while (t) { const int r = g % t; g = t; t = r; }
    const long m = C / g;
    long blocks = (total + kThreads * 4 - 1) / (kThreads * 4);
    blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
    blocks = (blocks + m - 1) / m * m;
    return (int)blocks;
}

// the workgroup's lanes that share this lane's channel add their (a, b) in lane order; lanes t < C with a live first
// element write the result to slab[0][c], slab[1][c]
__device__ __forceinline__ void combine2(float a, float b, int C, long first, long total, float *red, float *slab)
{
    red[threadIdx.x * 2 + 0] = a;
    red[threadIdx.x * 2 + 1] = b;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        const int c = (int)(first % C);
        float s = 0.0f, q = 0.0f;
        for (int t = threadIdx.x; t < kThreads; t += C) {
            s += red[t * 2 + 0];
            q += red[t * 2 + 1];
        }
        slab[c] = s;
        slab[C + c] = q;
    }
    (void)total;
}

__global__ __launch_bounds__(kThreads) void rows_stats_kernel(const float *__restrict__ x, long total, int C,
                                                               float *__restrict__ slabs)
{
    __shared__ float red[kThreads * 2];
    const long first = (long)blockIdx.x * kThreads + threadIdx.x;
    float s = 0.0f, q = 0.0f;
    for (long e = first; e < total; e += (long)gridDim.x * kThreads) {
        const float v = x[e];
        s += v;
        q = fmaf(v, v, q);
    }
    combine2(s, q, C, first, total, red, slabs + (long)blockIdx.x * 2 * C);
}

__global__ __launch_bounds__(kThreads) void rows_bn_act_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                                const float *__restrict__ shift, long total, int C,
                                                                float slope, float *__restrict__ z)
{
    const long first = (long)blockIdx.x * kThreads + threadIdx.x;
    const int c = (int)(first % C);
    const float sc = scale[c], sh = shift[c];
    for (long e = first; e < total; e += (long)gridDim.x * kThreads) z[e] = act_fwd(fmaf(x[e], sc, sh), slope);
}

__global__ __launch_bounds__(kThreads) void rows_bwd_reduce_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                                                                    const float *__restrict__ scale,
                                                                    const float *__restrict__ shift,
                                                                    const float *__restrict__ mean,
                                                                    const float *__restrict__ invstd, long total, int C,
                                                                    float slope, float *__restrict__ slabs)
{
    __shared__ float red[kThreads * 2];
    const long first = (long)blockIdx.x * kThreads + threadIdx.x;
    const int c = (int)(first % C);
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    float s1 = 0.0f, s2 = 0.0f;
    for (long e = first; e < total; e += (long)gridDim.x * kThreads) {
        const float v = x[e];
        const float du = dz[e] * act_grad(fmaf(v, sc, sh), slope);
        s1 += du;
        s2 = fmaf(du, (v - mu) * is, s2);
    }
    combine2(s1, s2, C, first, total, red, slabs + (long)blockIdx.x * 2 * C);
}

__global__ __launch_bounds__(kThreads) void rows_bwd_apply_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                                                                   const float *__restrict__ scale,
                                                                   const float *__restrict__ shift,
                                                                   const float *__restrict__ mean,
                                                                   const float *__restrict__ invstd,
                                                                   const float *__restrict__ sums, long total, int C,
                                                                   long rows, float slope, int use_batch_stats,
                                                                   float *__restrict__ dx)
{
    const long first = (long)blockIdx.x * kThreads + threadIdx.x;
    const int c = (int)(first % C);
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const float inv_r = 1.0f / (float)rows;
    const float a = use_batch_stats ? sums[c] * inv_r : 0.0f, b = use_batch_stats ? sums[C + c] * inv_r : 0.0f;
    for (long e = first; e < total; e += (long)gridDim.x * kThreads) {
        const float v = x[e];
        const float du = dz[e] * act_grad(fmaf(v, sc, sh), slope);
        dx[e] = sc * (du - a - (v - mu) * is * b);
    }
}

// slabs[p][b][c] = sum over the p-th slice of scene b's N rows of x[b][n][c]   (grid = (parts, B))
__global__ __launch_bounds__(kThreads) void scene_sum_kernel(const float *__restrict__ x, int N, int C,
                                                              float *__restrict__ slabs)
{
    __shared__ float red[kThreads * 2];
    const int b = blockIdx.y, B = gridDim.y;
    const long per = (long)N * C;
    // this workgroup's slice of the scene: whole rows, so that a lane's channel is (first element) % C throughout
    const long rows_per = (N + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * rows_per * C, hi = lo + rows_per * C < per ? lo + rows_per * C : per;
    // the lane stride must be a multiple of C: lanes beyond the largest multiple of C <= 256 idle
    const int live = kThreads / C * C;
    float s = 0.0f;
    if ((int)threadIdx.x < live)
        for (long e = lo + threadIdx.x; e < hi; e += live) s += x[(long)b * per + e];
    red[threadIdx.x] = (int)threadIdx.x < live ? s : 0.0f;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float a = 0.0f;
        for (int t = threadIdx.x; t < live; t += C) a += red[t];
        slabs[((long)blockIdx.x * B + b) * C + threadIdx.x] = a;
    }
}

inline bool bad(long R, int C) { return R <= 0 || C < 1 || C > 64; }

}  // namespace

extern "C" {

int pcb_rows_bn_partials(long R, int C) { return bad(R, C) ? 0 : grid_for(R * C, C, 512); }

int pcb_rows_bn_stats_f32(const float *x, long R, int C, float *slabs, int nparts, void *stream)
{
    if (!x || !slabs || bad(R, C) || nparts != grid_for(R * C, C, 512)) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rows_stats_kernel, dim3(nparts), dim3(kThreads), 0, (hipStream_t)stream, x, R * C, C, slabs);
    pcb_account(4.0 * R * C);
    return pcb_check_launch();
}

int pcb_rows_bn_act_f32(const float *x, const float *scale, const float *shift, long R, int C, int act, float *z,
                        void *stream)
{
    if (!x || !scale || !shift || !z || bad(R, C)) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rows_bn_act_kernel, dim3(grid_for(R * C, C, 2048)), dim3(kThreads), 0, (hipStream_t)stream, x, scale,
                       shift, R * C, C, slope_of(act), z);
    pcb_account(8.0 * R * C);
    return pcb_check_launch();
}

int pcb_rows_bn_act_bwd_reduce_f32(const float *dz, const float *x, const float *scale, const float *shift,
                                   const float *mean, const float *invstd, long R, int C, int act, float *slabs,
                                   int nparts, void *stream)
{
    if (!dz || !x || !scale || !shift || !mean || !invstd || !slabs || bad(R, C) || nparts != grid_for(R * C, C, 512))
        return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rows_bwd_reduce_kernel, dim3(nparts), dim3(kThreads), 0, (hipStream_t)stream, dz, x, scale, shift,
                       mean, invstd, R * C, C, slope_of(act), slabs);
    pcb_account(8.0 * R * C);
    return pcb_check_launch();
}

int pcb_rows_bn_act_bwd_apply_f32(const float *dz, const float *x, const float *scale, const float *shift,
                                  const float *mean, const float *invstd, const float *sums, long R, int C, int act,
                                  int use_batch_stats, float *dx, void *stream)
{
    if (!dz || !x || !scale || !shift || !mean || !invstd || !sums || !dx || bad(R, C)) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rows_bwd_apply_kernel, dim3(grid_for(R * C, C, 2048)), dim3(kThreads), 0, (hipStream_t)stream, dz, x,
                       scale, shift, mean, invstd, sums, R * C, C, R, slope_of(act), use_batch_stats, dx);
    pcb_account(12.0 * R * C);
    return pcb_check_launch();
}

int pcb_scene_sum_partials(int N) { return N <= 0 ? 0 : (N >= 64 * 64 ? 64 : (N + 63) / 64); }

int pcb_scene_sum_f32(const float *x, int B, int N, int C, float *slabs, int nparts, void *stream)
{
    if (!x || !slabs || B <= 0 || N <= 0 || C < 1 || C > 64 || nparts != pcb_scene_sum_partials(N)) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(scene_sum_kernel, dim3(nparts, B), dim3(kThreads), 0, (hipStream_t)stream, x, N, C, slabs);
    pcb_account(4.0 * B * N * C);
    return pcb_check_launch();
}

}  // extern "C"
