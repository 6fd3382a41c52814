// Neighbourhood MLP of the bridge structure encoder for gfx950.
//
// Replaces BridgeStructureEncoding.structure_mlp over the [B, 40, N, k] tensor of the reference,
// models/attention_modules.py:548-553 and :606-616 (expand + cat + permute, Conv2d 1x1 ->
// BatchNorm2d -> ReLU -> Conv2d 1x1, max over the k neighbours).  The first convolution is split by
// input block: `base` [P, C] holds bias + the 37 per-point channels times their weight columns
// (one small product per POINT, done by the caller), so that per neighbour only the 3 offset
// channels remain:
//     y1[i,j,:] = base[i,:] + Wr . rel[i,j,:]          z = relu(scale*y1 + shift)   (BatchNorm folded)
//     y2[i,j,:] = W2 . z + b2                           out[i,:] = max_j y2[i,j,:]
// With C <= 16 channels there is nothing for the matrix cores here: 12 bytes per (point, neighbour)
// in, C floats per point out -- HBM-bound on reading rel, one lane per point, everything else in
// registers.  fp32 throughout (these are geometry features).
//   stats pass      : per-block partial (sum y1, sum y1^2)               -> pcb_bn_finalize
//   forward pass    : out + arg-max neighbour per channel
//   backward reduce : partial (sum du, sum du*xhat), dW2, db2             -> pcb_bn_bwd_finalize
//   backward apply  : dy1 = scale*du + p*y1 + q  ->  dbase [P,C], partial dWr
#include "pcb_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / PCB_WAVE;
constexpr int kMaxBlocks = 2048;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // lane 0 holds the total
}

template <int CM>
struct Weights {
    float wr[CM][3], scale[CM], shift[CM], w2[CM][CM], b2[CM];
};

// zero-padded copy of the layer constants in LDS (broadcast reads afterwards)
template <int CM>
__device__ __forceinline__ void load_weights(Weights<CM> &w, int C, const float *wr, const float *scale,
                                             const float *shift, const float *w2, const float *b2)
{
    for (int e = threadIdx.x; e < CM * CM; e += kThreads) {
        const int c = e / CM, d = e - c * CM;
        w.w2[c][d] = (w2 && c < C && d < C) ? w2[c * C + d] : 0.0f;
        if (d < 3) w.wr[c][d] = c < C ? wr[c * 3 + d] : 0.0f;
        if (d == 0) {
            w.scale[c] = (scale && c < C) ? scale[c] : 0.0f;
            w.shift[c] = (shift && c < C) ? shift[c] : 0.0f;
            w.b2[c] = (b2 && c < C) ? b2[c] : 0.0f;
        }
    }
    __syncthreads();
}

// CM = 16: 256 second-layer weights would be hoisted out of the neighbour loop into VGPRs (511 of
// them, with spills); a compiler barrier per iteration keeps them as LDS broadcast reads.  The
// 16-channel encoders run on the coarse levels (a few thousand points), the 3-channel one on all N.
template <int CM>
__device__ __forceinline__ void keep_weights_in_lds()
{
    if constexpr (CM > 4) asm volatile("" ::: "memory");
}

template <int CM>
__device__ __forceinline__ void first_layer(const Weights<CM> &w, const float *bs, float x, float y, float z,
                                            float *y1)
{
#pragma unroll
    for (int c = 0; c < CM; ++c)
        y1[c] = bs[c] + fmaf(w.wr[c][2], z, fmaf(w.wr[c][1], y, w.wr[c][0] * x));
}

// block-level sum of per-thread partials v[0..n) into dst[0..n) (one block's slab)
template <int NV>
__device__ __forceinline__ void block_sums(float (&v)[NV], float *lds /*[kWaves][NV]*/, float *dst, int n_real,
                                           const int *map)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const float s = wave_sum(v[e]);
        if (lane == 0) lds[wave * NV + e] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NV; e += kThreads) {
        float s = 0.0f;
#pragma unroll
        for (int wv = 0; wv < kWaves; ++wv) s += lds[wv * NV + e];
        if (map[e] >= 0) dst[map[e]] = s;
    }
}

template <int CM>
__global__ __launch_bounds__(kThreads) void nbr_stats_kernel(const float *__restrict__ base,
                                                             const float *__restrict__ rel, long P, int k, int C,
                                                             const float *__restrict__ wr,
                                                             float *__restrict__ sums /*[grid][2][C]*/)
{
    __shared__ Weights<CM> w;
    __shared__ float red[kWaves * 2 * CM];
    __shared__ int map[2 * CM];
    load_weights<CM>(w, C, wr, nullptr, nullptr, nullptr, nullptr);
    for (int e = threadIdx.x; e < 2 * CM; e += kThreads) {
        const int which = e / CM, c = e - which * CM;
        map[e] = c < C ? which * C + c : -1;
    }
    float acc[2 * CM];
#pragma unroll
    for (int e = 0; e < 2 * CM; ++e) acc[e] = 0.0f;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < P; i += (long)gridDim.x * kThreads) {
        float bs[CM], y1[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) bs[c] = c < C ? base[i * C + c] : 0.0f;
        const float *__restrict__ r = rel + i * k * 3;
        for (int j = 0; j < k; ++j) {
            first_layer<CM>(w, bs, r[j * 3 + 0], r[j * 3 + 1], r[j * 3 + 2], y1);
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                acc[c] += y1[c];
                acc[CM + c] = fmaf(y1[c], y1[c], acc[CM + c]);
            }
        }
    }
    block_sums<2 * CM>(acc, red, sums + (long)blockIdx.x * 2 * C, 2 * C, map);
}

template <int CM>
__global__ __launch_bounds__(kThreads) void nbr_forward_kernel(
    const float *__restrict__ base, const float *__restrict__ rel, long P, int k, int C,
    const float *__restrict__ wr, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ w2, const float *__restrict__ b2, float *__restrict__ out,
    unsigned char *__restrict__ arg)
{
    __shared__ Weights<CM> w;
    load_weights<CM>(w, C, wr, scale, shift, w2, b2);
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < P; i += (long)gridDim.x * kThreads) {
        float bs[CM], y1[CM], best[CM];
        int barg[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            bs[c] = c < C ? base[i * C + c] : 0.0f;
            best[c] = -INFINITY;
            barg[c] = 0;
        }
        const float *__restrict__ r = rel + i * k * 3;
        for (int j = 0; j < k; ++j) {
            keep_weights_in_lds<CM>();
            first_layer<CM>(w, bs, r[j * 3 + 0], r[j * 3 + 1], r[j * 3 + 2], y1);
#pragma unroll
            for (int c = 0; c < CM; ++c) y1[c] = fmaxf(fmaf(w.scale[c], y1[c], w.shift[c]), 0.0f);
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                float y2 = w.b2[c];
#pragma unroll
                for (int d = 0; d < CM; ++d) y2 = fmaf(w.w2[c][d], y1[d], y2);
                if (y2 > best[c]) {  // first maximum wins, like torch.max(dim)
                    best[c] = y2;
                    barg[c] = j;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) {
                out[i * C + c] = best[c];
                arg[i * C + c] = (unsigned char)barg[c];
            }
    }
}

// du[d] (gradient at the BatchNorm output, after the ReLU mask) of neighbour j, from the pooled
// gradient: dz[d] = sum_{c: arg[c]==j} W2[c][d] * dout[c]
template <int CM>
__device__ __forceinline__ void grad_at_bn(const Weights<CM> &w, const float *go, const int *barg, int j,
                                           const float *zr /*scale*y1+shift*/, float *du)
{
#pragma unroll
    for (int d = 0; d < CM; ++d) du[d] = 0.0f;
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        const float gsel = barg[c] == j ? go[c] : 0.0f;
#pragma unroll
        for (int d = 0; d < CM; ++d) du[d] = fmaf(w.w2[c][d], gsel, du[d]);
    }
#pragma unroll
    for (int d = 0; d < CM; ++d) du[d] = zr[d] > 0.0f ? du[d] : 0.0f;
}

template <int CM>
__global__ __launch_bounds__(kThreads) void nbr_bwd_reduce_kernel(
    const float *__restrict__ base, const float *__restrict__ rel, long P, int k, int C,
    const float *__restrict__ wr, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ w2,
    const float *__restrict__ dout, const unsigned char *__restrict__ arg,
    float *__restrict__ sums /*[grid][2][C]*/, float *__restrict__ dw2 /*[grid][C][C+1]: dW2 | db2*/)
{
    __shared__ Weights<CM> w;
    __shared__ float red[kWaves * 2 * CM];
    __shared__ int map[2 * CM];
    __shared__ float s_mean[CM], s_invstd[CM];
    __shared__ float s_dw2[kWaves][CM][CM + 1];   // one slot per wave: plain adds by its lane 0, summed in wave order (no atomics: reproducible)
    load_weights<CM>(w, C, wr, scale, shift, w2, nullptr);
    for (int e = threadIdx.x; e < 2 * CM; e += kThreads) {
        const int which = e / CM, c = e - which * CM;
        map[e] = c < C ? which * C + c : -1;
        if (which == 0) {
            s_mean[c] = c < C ? mean[c] : 0.0f;
            s_invstd[c] = c < C ? invstd[c] : 0.0f;
        }
    }
    for (int e = threadIdx.x; e < kWaves * CM * (CM + 1); e += kThreads) (&s_dw2[0][0][0])[e] = 0.0f;
    __syncthreads();
    float acc[2 * CM];
#pragma unroll
    for (int e = 0; e < 2 * CM; ++e) acc[e] = 0.0f;
    const int lane = threadIdx.x & 63;
    // wave-uniform trip count: the dW2 sums below are cross-lane reductions, so lanes past the end
    // stay in the loop with a zero gradient instead of leaving it
    for (long i0 = (long)blockIdx.x * kThreads; i0 < P; i0 += (long)gridDim.x * kThreads) {
        const bool valid = i0 + threadIdx.x < P;
        const long i = valid ? i0 + threadIdx.x : P - 1;
        float bs[CM], go[CM], y1[CM], zr[CM], du[CM];
        int barg[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            bs[c] = c < C ? base[i * C + c] : 0.0f;
            go[c] = (valid && c < C) ? dout[i * C + c] : 0.0f;
            barg[c] = c < C ? (int)arg[i * C + c] : 0;
        }
        const float *__restrict__ r = rel + i * k * 3;
        for (int j = 0; j < k; ++j) {
            keep_weights_in_lds<CM>();
            first_layer<CM>(w, bs, r[j * 3 + 0], r[j * 3 + 1], r[j * 3 + 2], y1);
#pragma unroll
            for (int c = 0; c < CM; ++c) zr[c] = fmaf(w.scale[c], y1[c], w.shift[c]);
            grad_at_bn<CM>(w, go, barg, j, zr, du);
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                acc[c] += du[c];
                acc[CM + c] = fmaf(du[c], (y1[c] - s_mean[c]) * s_invstd[c], acc[CM + c]);
            }
        }
        // dW2[c][:] += dout[c] * z[:] at the neighbour that won channel c, db2[c] += dout[c]:
        // recompute z at that neighbour, add up the wave, one LDS atomic per wave and entry
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) {  // wave-uniform
                keep_weights_in_lds<CM>();
                const int jj = barg[c] < k ? barg[c] : 0;
                first_layer<CM>(w, bs, r[jj * 3 + 0], r[jj * 3 + 1], r[jj * 3 + 2], y1);
#pragma unroll
                for (int d = 0; d < CM; ++d)
                    if (d < C) {
                        const float zd = fmaxf(fmaf(w.scale[d], y1[d], w.shift[d]), 0.0f);
                        const float s = wave_sum(go[c] * zd);
                        if (lane == 0) s_dw2[threadIdx.x >> 6][c][d] += s;
                    }
                const float sg = wave_sum(go[c]);
                if (lane == 0) s_dw2[threadIdx.x >> 6][c][CM] += sg;
            }
    }
    block_sums<2 * CM>(acc, red, sums + (long)blockIdx.x * 2 * C, 2 * C, map);
    __syncthreads();
    for (int e = threadIdx.x; e < C * (C + 1); e += kThreads) {
        const int c = e / (C + 1), d = e - c * (C + 1);
        float a = 0.0f;
        for (int wv = 0; wv < kWaves; ++wv) a += s_dw2[wv][c][d < C ? d : CM];
        dw2[(long)blockIdx.x * C * (C + 1) + e] = a;
    }
}

template <int CM>
__global__ __launch_bounds__(kThreads) void nbr_bwd_apply_kernel(
    const float *__restrict__ base, const float *__restrict__ rel, long P, int k, int C,
    const float *__restrict__ wr, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ p, const float *__restrict__ q, const float *__restrict__ w2,
    const float *__restrict__ dout, const unsigned char *__restrict__ arg, float *__restrict__ dbase,
    float *__restrict__ dwr /*[grid][C][3]*/)
{
    __shared__ Weights<CM> w;
    __shared__ float red[kWaves * 3 * CM];
    __shared__ int map[3 * CM];
    __shared__ float s_p[CM], s_q[CM];
    load_weights<CM>(w, C, wr, scale, shift, w2, nullptr);
    for (int e = threadIdx.x; e < 3 * CM; e += kThreads) {
        const int c = e / 3, d = e - c * 3;
        map[e] = c < C ? c * 3 + d : -1;
        if (d == 0) {
            s_p[c] = c < C ? p[c] : 0.0f;
            s_q[c] = c < C ? q[c] : 0.0f;
        }
    }
    __syncthreads();
    float acc[3 * CM];
#pragma unroll
    for (int e = 0; e < 3 * CM; ++e) acc[e] = 0.0f;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < P; i += (long)gridDim.x * kThreads) {
        float bs[CM], go[CM], y1[CM], zr[CM], du[CM], db[CM];
        int barg[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            bs[c] = c < C ? base[i * C + c] : 0.0f;
            go[c] = c < C ? dout[i * C + c] : 0.0f;
            barg[c] = c < C ? (int)arg[i * C + c] : -1;
            db[c] = 0.0f;
        }
        const float *__restrict__ r = rel + i * k * 3;
        for (int j = 0; j < k; ++j) {
            keep_weights_in_lds<CM>();
            const float x = r[j * 3 + 0], y = r[j * 3 + 1], z = r[j * 3 + 2];
            first_layer<CM>(w, bs, x, y, z, y1);
#pragma unroll
            for (int c = 0; c < CM; ++c) zr[c] = fmaf(w.scale[c], y1[c], w.shift[c]);
            grad_at_bn<CM>(w, go, barg, j, zr, du);
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                // BatchNorm backward folded into per-channel constants: dy1 = scale*du + p*y1 + q
                const float dy = fmaf(w.scale[c], du[c], fmaf(s_p[c], y1[c], s_q[c]));
                db[c] += dy;
                acc[c * 3 + 0] = fmaf(dy, x, acc[c * 3 + 0]);
                acc[c * 3 + 1] = fmaf(dy, y, acc[c * 3 + 1]);
                acc[c * 3 + 2] = fmaf(dy, z, acc[c * 3 + 2]);
            }
        }
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) dbase[i * C + c] = db[c];
    }
    block_sums<3 * CM>(acc, red, dwr + (long)blockIdx.x * 3 * C, 3 * C, map);
}

int grid_for_points(long P)
{
    const long b = (P + kThreads - 1) / kThreads;
    return (int)(b < kMaxBlocks ? b : kMaxBlocks);
}

bool bad_shape(long P, int k, int C) { return P <= 0 || k < 1 || k > 255 || C < 1 || C > 16; }

}  // namespace

extern "C" int pcb_nbr_mlp_partials(long points) { return points > 0 ? grid_for_points(points) : 0; }

extern "C" int pcb_nbr_mlp_stats(const float *base, const float *rel, long P, int k, int C, const float *wr,
                                 float *sums, void *stream)
{
    if (!base || !rel || !wr || !sums) return PCB_ERR_INVALID_ARG;
    if (bad_shape(P, k, C)) return PCB_ERR_UNSUPPORTED;
    const dim3 grid(grid_for_points(P));
    if (C <= 4)
        hipLaunchKernelGGL(nbr_stats_kernel<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, base, rel, P, k, C, wr, sums);
    else
        hipLaunchKernelGGL(nbr_stats_kernel<16>, grid, dim3(kThreads), 0, (hipStream_t)stream, base, rel, P, k, C, wr, sums);
    return pcb_check_launch();
}

extern "C" int pcb_nbr_mlp_forward(const float *base, const float *rel, long P, int k, int C, const float *wr,
                                   const float *scale, const float *shift, const float *w2, const float *b2,
                                   float *out, unsigned char *arg, void *stream)
{
    if (!base || !rel || !wr || !scale || !shift || !w2 || !out || !arg) return PCB_ERR_INVALID_ARG;
    if (bad_shape(P, k, C)) return PCB_ERR_UNSUPPORTED;
    const dim3 grid(grid_for_points(P));
    if (C <= 4)
        hipLaunchKernelGGL(nbr_forward_kernel<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, base, rel, P, k, C, wr,
                           scale, shift, w2, b2, out, arg);
    else
        hipLaunchKernelGGL(nbr_forward_kernel<16>, grid, dim3(kThreads), 0, (hipStream_t)stream, base, rel, P, k, C, wr,
                           scale, shift, w2, b2, out, arg);
    return pcb_check_launch();
}

extern "C" int pcb_nbr_mlp_backward_reduce(const float *base, const float *rel, long P, int k, int C,
                                           const float *wr, const float *scale, const float *shift,
                                           const float *mean, const float *invstd, const float *w2,
                                           const float *dout, const unsigned char *arg, float *sums, float *dw2,
                                           void *stream)
{
    if (!base || !rel || !wr || !scale || !shift || !mean || !invstd || !w2 || !dout || !arg || !sums || !dw2)
        return PCB_ERR_INVALID_ARG;
    if (bad_shape(P, k, C)) return PCB_ERR_UNSUPPORTED;
    const dim3 grid(grid_for_points(P));
    if (C <= 4)
        hipLaunchKernelGGL(nbr_bwd_reduce_kernel<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, base, rel, P, k, C,
                           wr, scale, shift, mean, invstd, w2, dout, arg, sums, dw2);
    else
        hipLaunchKernelGGL(nbr_bwd_reduce_kernel<16>, grid, dim3(kThreads), 0, (hipStream_t)stream, base, rel, P, k, C,
                           wr, scale, shift, mean, invstd, w2, dout, arg, sums, dw2);
    return pcb_check_launch();
}

extern "C" int pcb_nbr_mlp_backward_apply(const float *base, const float *rel, long P, int k, int C,
                                          const float *wr, const float *scale, const float *shift, const float *p,
                                          const float *q, const float *w2, const float *dout,
                                          const unsigned char *arg, float *dbase, float *dwr, void *stream)
{
    if (!base || !rel || !wr || !scale || !shift || !p || !q || !w2 || !dout || !arg || !dbase || !dwr)
        return PCB_ERR_INVALID_ARG;
    if (bad_shape(P, k, C)) return PCB_ERR_UNSUPPORTED;
    const dim3 grid(grid_for_points(P));
    if (C <= 4)
        hipLaunchKernelGGL(nbr_bwd_apply_kernel<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, base, rel, P, k, C,
                           wr, scale, shift, p, q, w2, dout, arg, dbase, dwr);
    else
        hipLaunchKernelGGL(nbr_bwd_apply_kernel<16>, grid, dim3(kThreads), 0, (hipStream_t)stream, base, rel, P, k, C,
                           wr, scale, shift, p, q, w2, dout, arg, dbase, dwr);
    return pcb_check_launch();
}
