// Row kernels of the transformer token pipeline (scope row f4: PointTransformerV3's inference pass in bf16 rows,
// models/PointTransformerV3.py:119-148 PointTransformerBlock, :8-21 GEGLU).  Between the library GEMMs and the attention
// kernel a block is elementwise / per-row work on [B*N, C] token rows; through ATen that was ~20 launches per block
// (casts to fp32 and back around every LayerNorm, the positional add, two residual adds, GELU and the gate product as
// separate passes): 2.9 ms of the 7.7 ms cfg5 pass.  Here:
//   add_layernorm   x' = x + h (the pending residual, optional, stored as the new residual stream)
//                   out = LayerNorm(x') * gamma + beta (+ pos)      -- fp32 statistics over the row, one rounding to bf16
//   geglu           out = a * gelu(g) for the two halves [a | g] of a projection's rows (erf form, F.gelu's default)
// One wave per row (C <= 1024 in 16-byte chunks per lane), statistics by two-pass sums in registers (DPP-free shuffles).
#include "rowvec.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunks = 2;   // 16-byte chunks per lane: C <= 1024

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <int NCH>
__global__ __launch_bounds__(kThreads) void add_layernorm_kernel(const pcb_bf16 *__restrict__ x, const pcb_bf16 *__restrict__ h,
                                                                  const pcb_bf16 *__restrict__ pos, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float eps, long R, int C,
                                                                  pcb_bf16 *__restrict__ xout, pcb_bf16 *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (row >= R) return;   // (whole waves leave: no barrier in this kernel)
    float v[NCH][8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            RowVec<pcb_bf16>::unpack(*reinterpret_cast<const uint4 *>(x + row * C + c), v[i]);
            if (h) {
                float hv[8];
                RowVec<pcb_bf16>::unpack(*reinterpret_cast<const uint4 *>(h + row * C + c), hv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = RowVec<pcb_bf16>::stored(v[i][e] + hv[e]);   // the residual stream is bf16
                if (xout) *reinterpret_cast<uint4 *>(xout + row * C + c) = RowVec<pcb_bf16>::pack(v[i]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.0f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                q = fmaf(d, d, q);
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);   // biased variance, as nn.LayerNorm
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            float o[8], pv[8];
            if (pos) RowVec<pcb_bf16>::unpack(*reinterpret_cast<const uint4 *>(pos + row * C + c), pv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = fmaf((v[i][e] - mean) * rstd, gamma[c + e], beta[c + e]);
                if (pos) o[e] += pv[e];
            }
            *reinterpret_cast<uint4 *>(out + row * C + c) = RowVec<pcb_bf16>::pack(o);
        }
    }
}

__global__ __launch_bounds__(kThreads) void geglu_kernel(const pcb_bf16 *__restrict__ y, long R, int H, pcb_bf16 *__restrict__ out)
{
    const long chunks = R * (H / 8);
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < chunks; e += (long)gridDim.x * kThreads) {
        const long row = e / (H / 8);
        const int c = (int)(e % (H / 8)) * 8;
        float a[8], g[8], o[8];
        RowVec<pcb_bf16>::unpack(*reinterpret_cast<const uint4 *>(y + row * 2 * H + c), a);
        RowVec<pcb_bf16>::unpack(*reinterpret_cast<const uint4 *>(y + row * 2 * H + H + c), g);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = a[i] * (0.5f * g[i] * (1.0f + erff(g[i] * 0.70710678118654752f)));
        *reinterpret_cast<uint4 *>(out + row * H + c) = RowVec<pcb_bf16>::pack(o);
    }
}

}  // namespace

extern "C" int pcb_add_layernorm_bf16(const void *x, const void *h, const void *pos, const float *gamma, const float *beta,
                                      float eps, long R, int C, void *xout, void *out, void *stream)
{
    if (!x || !gamma || !beta || !out || R <= 0 || C <= 0 || (C % 8) || C > 64 * 8 * kMaxChunks) return PCB_ERR_INVALID_ARG;
    if (xout && !h) return PCB_ERR_INVALID_ARG;
    const dim3 grid((unsigned)((R + kThreads / 64 - 1) / (kThreads / 64)));
    if (C <= 512)
        hipLaunchKernelGGL((add_layernorm_kernel<1>), grid, dim3(kThreads), 0, (hipStream_t)stream, (const pcb_bf16 *)x,
                           (const pcb_bf16 *)h, (const pcb_bf16 *)pos, gamma, beta, eps, R, C, (pcb_bf16 *)xout, (pcb_bf16 *)out);
    else
        hipLaunchKernelGGL((add_layernorm_kernel<2>), grid, dim3(kThreads), 0, (hipStream_t)stream, (const pcb_bf16 *)x,
                           (const pcb_bf16 *)h, (const pcb_bf16 *)pos, gamma, beta, eps, R, C, (pcb_bf16 *)xout, (pcb_bf16 *)out);
    pcb_account(2.0 * R * C * (2.0 + (h ? 1.0 : 0.0) + (pos ? 1.0 : 0.0) + (xout ? 1.0 : 0.0)));
    return pcb_check_launch();
}

extern "C" int pcb_geglu_bf16(const void *y, long R, int H, void *out, void *stream)
{
    if (!y || !out || R <= 0 || H <= 0 || (H % 8)) return PCB_ERR_INVALID_ARG;
    const long chunks = R * (H / 8);
    long blocks = (chunks + kThreads - 1) / kThreads;
    blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, (const pcb_bf16 *)y, R, H,
                       (pcb_bf16 *)out);
    pcb_account(2.0 * R * H * 3.0);
    return pcb_check_launch();
}
