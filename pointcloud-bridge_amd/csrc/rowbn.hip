// bf16 channels-last row kernels around the pointwise-MLP GEMMs (HBM-bound byte movers).
//
// The reference evaluates every shared MLP as Conv2d/Conv1d(1x1) -> BatchNorm -> ReLU on fp32
// [B,C,S,ns] tensors, with a torch.max over the neighbour axis at the end of a set-abstraction
// level (models/pointnet2_utils.py:149-154, :207-209, :353-356; DGCNN.py:134-148 with LeakyReLU).
// ATen runs that as 5-7 full passes over the activation per layer.  Here activations are bf16 rows
// [rows, C]; the pre-BatchNorm GEMM output y is the only tensor kept per layer, and
//   colstats        one pass:  sum(y), sum(y*y) per channel                (train-mode statistics)
//   bn_finalize     C threads: scale/shift, running-stat update            (BatchNorm bookkeeping)
//   bn_act          one pass:  z = act(y*scale + shift)                    (operand of the next GEMM)
//   bn_act_max      one pass:  max over the ns rows of a group + arg-max   (SA / EdgeConv pooling)
//   *_bwd_reduce    one pass:  s1 = sum(du), s2 = sum(du * xhat)           (BatchNorm backward sums)
//   *_bwd_apply     one pass:  dy = scale * (du - s1/R - xhat * s2/R)      (operand of dgrad/wgrad)
// Every kernel moves 16-byte vectors (8 bf16 channels per lane) and keeps fp32 in registers.
#include "pcb_common.h"

namespace {

typedef unsigned short u16;

__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ u16 f2bf(float f)
{
    return __builtin_bit_cast(u16, (__bf16)f);  // round-to-nearest-even, NaN stays NaN
}
__device__ __forceinline__ void unpack8(const uint4 &v, float *f)
{
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float *f)
{
    uint4 v;
    v.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
    v.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
    v.z = (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16);
    v.w = (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16);
    return v;
}

// activation codes shared with the host: 0 none, 1 ReLU, 2 LeakyReLU(0.2)
// (applied as one select on a per-launch slope: no per-element tests of the activation code)
inline float slope_of(int act) { return act == 1 ? 0.0f : (act == 2 ? 0.2f : 1.0f); }  // host side
__device__ __forceinline__ float act_fwd(float u, float slope) { return u > 0.0f ? u : fmaf(slope, u, 0.0f); }
__device__ __forceinline__ float act_grad(float u, float slope) { return u > 0.0f ? 1.0f : slope; }

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------
// Column statistics: sums[0][c] = sum_r y[r][c], sums[1][c] = sum_r y[r][c]^2   (C % 8 == 0, C <= 2048)
// Lane t owns channel chunk t % CT (8 channels) and walks rows t / CT, + RT, ...; the block
// combines its row-lanes through LDS and issues one fp32 atomic per channel and moment.
__global__ __launch_bounds__(kThreads) void colstats_kernel(const uint4 *__restrict__ y, long rows,
                                                             int C, float *__restrict__ sums)
{
    __shared__ float red[kThreads * 16];
    const int CT = C >> 3;
    const int RT = kThreads / CT;          // row-lanes per block (CT <= 256)
    const int cc = threadIdx.x % CT;
    const int rl = threadIdx.x / CT;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rl < RT) {
        for (long r = (long)blockIdx.x * RT + rl; r < rows; r += (long)gridDim.x * RT) {
            float f[8];
            unpack8(y[r * CT + cc], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s[i] += f[i];
                q[i] = fmaf(f[i], f[i], q[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[threadIdx.x * 16 + i] = s[i];
        red[threadIdx.x * 16 + 8 + i] = q[i];
    }
    __syncthreads();
    // thread t < C*2 reduces one (moment, channel) over the RT row-lanes
    for (int o = threadIdx.x; o < 2 * C; o += kThreads) {
        const int m = o / C, c = o % C;
        float a = 0.0f;
        for (int r = 0; r < RT; ++r) a += red[(r * CT + (c >> 3)) * 16 + m * 8 + (c & 7)];
        atomicAdd(&sums[o], a);
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm bookkeeping for one layer (C threads).
// training: batch statistics from `sums` over `rows` rows of y = x W^T (the conv bias, which the
// GEMM does not add because it cancels inside a train-mode BatchNorm, is added to the mean that
// goes into running_mean); running_var gets the unbiased variance (rows/(rows-1)), as
// torch.nn.BatchNorm does.  eval: running statistics; the bias is folded into the shift.
// Outputs: scale = gamma*invstd, shift = beta - (mean_y)*scale [+ bias*scale in eval], and
// mean_y / invstd for the backward pass.
// Block = 32 channels x 32 slab-lanes: the partial slabs are added by 32 lanes per channel
// (coalesced 128-byte reads across the channels), combined through LDS in a fixed order.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float *__restrict__ sums, int nparts, long rows, long count, int C, const float *__restrict__ gamma,
    const float *__restrict__ beta, const float *__restrict__ bias, float *__restrict__ running_mean,
    float *__restrict__ running_var, float momentum, float eps, int training, float *__restrict__ scale,
    float *__restrict__ shift, float *__restrict__ mean_out, float *__restrict__ invstd_out,
    long long *__restrict__ num_batches_tracked)
{
    __shared__ float red[2][32][32];
    // nn.BatchNorm's step counter (num_batches_tracked += 1 in a train-mode forward), bumped here
    // instead of by one more tiny launch per module
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s1 = 0.0f, s2 = 0.0f;
    if (training && c < C) {
        for (int k = pl; k < nparts; k += 32) {  // sums is [nparts][2][C]
            s1 += sums[((long)k * 2 + 0) * C + c];
            s2 += sums[((long)k * 2 + 1) * C + c];
        }
    }
    red[0][pl][cl] = s1;
    red[1][pl][cl] = s2;
    __syncthreads();
    if (pl != 0 || c >= C) return;
    s1 = 0.0f;
    s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        s1 += red[0][k][cl];
        s2 += red[1][k][cl];
    }
    const float b = bias ? bias[c] : 0.0f;
    float mean_y, invstd;
    if (training) {
        const float n = (float)rows;
        mean_y = s1 / n;
        float var = s2 / n - mean_y * mean_y;
        var = var < 0.0f ? 0.0f : var;
        invstd = rsqrtf(var + eps);
        if (running_mean) {
            const float m = (float)count;
            const float unb = count > 1 ? var * (m / (m - 1.0f)) : var;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (mean_y + b);
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unb;
        }
    } else {
        mean_y = running_mean[c] - b;  // BN(y + b) with running stats == (y - (rm - b)) * invstd
        invstd = rsqrtf(running_var[c] + eps);
    }
    const float g = gamma ? gamma[c] : 1.0f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.0f) - mean_y * sc;
    mean_out[c] = mean_y;
    invstd_out[c] = invstd;
}

// ---------------------------------------------------------------------------------------------
// z = act(y*scale + shift), elementwise over [rows, C] bf16.
__global__ __launch_bounds__(kThreads) void bn_act_kernel(const uint4 *__restrict__ y,
                                                           const float *__restrict__ scale,
                                                           const float *__restrict__ shift, int C,
                                                           float act, uint4 *__restrict__ z, long nvec)
{
    const int CT = C >> 3;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const int c0 = (int)(e % CT) << 3;
        float f[8];
        unpack8(y[e], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = act_fwd(fmaf(f[i], scale[c0 + i], shift[c0 + i]), act);
        z[e] = pack8(f);
    }
}

// out[g][c] = max_j act(y[g*ns + j][c]*scale + shift), arg[g][c] = first j attaining it.
__global__ __launch_bounds__(kThreads) void bn_act_max_kernel(const uint4 *__restrict__ y,
                                                               const float *__restrict__ scale,
                                                               const float *__restrict__ shift,
                                                               int C, int ns, float act,
                                                               uint4 *__restrict__ out,
                                                               unsigned char *__restrict__ arg,
                                                               long nvec /* groups * C/8 */)
{
    const int CT = C >> 3;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const int cc = (int)(e % CT);
        const long g = e / CT;
        float sc[8], sh[8], best[8];
        int bj[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sc[i] = scale[cc * 8 + i];
            sh[i] = shift[cc * 8 + i];
            best[i] = -INFINITY;
            bj[i] = 0;
        }
        for (int j = 0; j < ns; ++j) {
            float f[8];
            unpack8(y[(g * ns + j) * CT + cc], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = act_fwd(fmaf(f[i], sc[i], sh[i]), act);
                if (v > best[i]) {
                    best[i] = v;
                    bj[i] = j;
                }
            }
        }
        out[e] = pack8(best);
        unsigned long long packed = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) packed |= (unsigned long long)(bj[i] & 0xff) << (8 * i);
        *reinterpret_cast<unsigned long long *>(arg + e * 8) = packed;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward sums for a dense upstream gradient dz [rows, C] (bf16):
// du = dz * act'(u), u = y*scale + shift;  s1 += du,  s2 += du * xhat,  xhat = (y - mean)*invstd.
__global__ __launch_bounds__(kThreads) void bn_act_bwd_reduce_kernel(
    const uint4 *__restrict__ dz, const uint4 *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
    long rows, int C, float act, float *__restrict__ sums)
{
    __shared__ float red[kThreads * 16];
    const int CT = C >> 3;
    const int RT = kThreads / CT;
    const int cc = threadIdx.x % CT;
    const int rl = threadIdx.x / CT;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rl < RT) {
        float sc[8], sh[8], mu[8], is[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sc[i] = scale[cc * 8 + i];
            sh[i] = shift[cc * 8 + i];
            mu[i] = mean[cc * 8 + i];
            is[i] = invstd[cc * 8 + i];
        }
        for (long r = (long)blockIdx.x * RT + rl; r < rows; r += (long)gridDim.x * RT) {
            float fy[8], fd[8];
            unpack8(y[r * CT + cc], fy);
            unpack8(dz[r * CT + cc], fd);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float du = fd[i] * act_grad(fmaf(fy[i], sc[i], sh[i]), act);
                s1[i] += du;
                s2[i] = fmaf(du, (fy[i] - mu[i]) * is[i], s2[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[threadIdx.x * 16 + i] = s1[i];
        red[threadIdx.x * 16 + 8 + i] = s2[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * C; o += kThreads) {
        const int m = o / C, c = o % C;
        float a = 0.0f;
        for (int r = 0; r < RT; ++r) a += red[(r * CT + (c >> 3)) * 16 + m * 8 + (c & 7)];
        atomicAdd(&sums[o], a);
    }
}

// dy = scale * (du - s1/R - xhat * s2/R)      (BatchNorm backward, batch statistics)
// eval mode (use_batch_stats == 0): dy = scale * du.
__global__ __launch_bounds__(kThreads) void bn_act_bwd_apply_kernel(
    const uint4 *__restrict__ dz, const uint4 *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ sums, long rows, int C, float act, int use_batch_stats,
    uint4 *__restrict__ dy, long nvec)
{
    const int CT = C >> 3;
    const float invR = 1.0f / (float)rows;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const int c0 = (int)(e % CT) << 3;
        float fy[8], fd[8];
        unpack8(y[e], fy);
        unpack8(dz[e], fd);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + i;
            const float du = fd[i] * act_grad(fmaf(fy[i], scale[c], shift[c]), act);
            const float xh = (fy[i] - mean[c]) * invstd[c];
            const float corr = use_batch_stats ? fmaf(xh, sums[C + c] * invR, sums[c] * invR) : 0.0f;
            fd[i] = scale[c] * (du - corr);
        }
        dy[e] = pack8(fd);
    }
}

// Pooled layers: the upstream gradient dout [groups, C] (fp32) reaches only the arg-max row of each
// (group, channel).  Sums over those rows:
__global__ __launch_bounds__(kThreads) void bn_max_bwd_reduce_kernel(
    const float *__restrict__ dout, const unsigned char *__restrict__ arg, const uint4 *__restrict__ y,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ invstd, long groups, int C, int ns, float act, float *__restrict__ sums)
{
    __shared__ float red[kThreads * 16];
    const int CT = C >> 3;
    const int RT = kThreads / CT;
    const int cc = threadIdx.x % CT;
    const int rl = threadIdx.x / CT;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rl < RT) {
        for (long g = (long)blockIdx.x * RT + rl; g < groups; g += (long)gridDim.x * RT) {
            const unsigned long long a = *reinterpret_cast<const unsigned long long *>(arg + (g * CT + cc) * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cc * 8 + i;
                const int j = (int)((a >> (8 * i)) & 0xff);
                const float yv = bf2f(reinterpret_cast<const u16 *>(y)[(g * ns + j) * (long)C + c]);
                const float du = dout[g * C + c] * act_grad(fmaf(yv, scale[c], shift[c]), act);
                s1[i] += du;
                s2[i] = fmaf(du, (yv - mean[c]) * invstd[c], s2[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[threadIdx.x * 16 + i] = s1[i];
        red[threadIdx.x * 16 + 8 + i] = s2[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * C; o += kThreads) {
        const int m = o / C, c = o % C;
        float t = 0.0f;
        for (int r = 0; r < RT; ++r) t += red[(r * CT + (c >> 3)) * 16 + m * 8 + (c & 7)];
        atomicAdd(&sums[o], t);
    }
}

// dy[g*ns + j][c] = scale * ((j == arg ? dout*act' : 0) - s1/R - xhat*s2/R), dense bf16 [rows, C].
__global__ __launch_bounds__(kThreads) void bn_max_bwd_apply_kernel(
    const float *__restrict__ dout, const unsigned char *__restrict__ arg, const uint4 *__restrict__ y,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ sums, long rows, int C, int ns,
    float act, int use_batch_stats, uint4 *__restrict__ dy, long nvec /* rows * C/8 */)
{
    const int CT = C >> 3;
    const float invR = 1.0f / (float)rows;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const int cc = (int)(e % CT);
        const long r = e / CT;
        const long g = r / ns;
        const int j = (int)(r % ns);
        const unsigned long long a = *reinterpret_cast<const unsigned long long *>(arg + (g * CT + cc) * 8);
        float fy[8], o[8];
        unpack8(y[e], fy);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cc * 8 + i;
            float du = 0.0f;
            if ((int)((a >> (8 * i)) & 0xff) == j)
                du = dout[g * C + c] * act_grad(fmaf(fy[i], scale[c], shift[c]), act);
            const float xh = (fy[i] - mean[c]) * invstd[c];
            const float corr = use_batch_stats ? fmaf(xh, sums[C + c] * invR, sums[c] * invR) : 0.0f;
            o[i] = scale[c] * (du - corr);
        }
        dy[e] = pack8(o);
    }
}

// ---------------------------------------------------------------------------------------------
// Grouping into bf16 GEMM rows: out[(b,s,j)][0:C] = feat[b, idx][0:C], [C:C+3] = xyz[b,idx]-new_xyz[b,s],
// zero up to Kp.  Features FIRST (so that 16-byte chunks of a feature row stay aligned); the host
// permutes the weight columns to match (the reference's order is coordinates first, :56 / :347).
__global__ __launch_bounds__(kThreads) void group_rows_bf16_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const u16 *__restrict__ feat,
    const int64_t *__restrict__ idx, int N, int S, int ns, int C, int Kp, u16 *__restrict__ out,
    long nchunk /* rows * Kp/8 */)
{
    const int KT = Kp >> 3;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nchunk; e += (long)gridDim.x * kThreads) {
        const int k0 = (int)(e % KT) << 3;
        const long row = e / KT;         // (b*S + s)*ns + j
        const long bs = row / ns;
        const long b = bs / S;
        const int i = clamp_index(idx[row], N);
        uint4 v;
        if ((C & 7) == 0 && k0 + 8 <= C) {
            v = *reinterpret_cast<const uint4 *>(feat + (b * N + i) * (long)C + k0);
        } else {
            float f[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = k0 + t;
                float val = 0.0f;
                if (k < C)
                    val = bf2f(feat[(b * N + i) * (long)C + k]);
                else if (k < C + 3)
                    val = __fsub_rn(xyz[(b * N + i) * 3 + (k - C)], new_xyz[bs * 3 + (k - C)]);
                f[t] = val;
            }
            v = pack8(f);
        }
        *reinterpret_cast<uint4 *>(out + row * (long)Kp + k0) = v;
    }
}

// grad_feat[b, idx, c] += g[row][c]   (fp32 accumulation, c < C)
__global__ __launch_bounds__(kThreads) void group_rows_bf16_bwd_kernel(
    const u16 *__restrict__ g, const int64_t *__restrict__ idx, int N, int S, int ns, int C, int Kp,
    float *__restrict__ gfeat, long total /* rows * C */)
{
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
        const int c = (int)(e % C);
        const long row = e / C;
        const long b = row / ((long)S * ns);
        const int i = clamp_index(idx[row], N);
        atomicAdd(&gfeat[(b * N + i) * (long)C + c], bf2f(g[row * (long)Kp + c]));
    }
}

// Channel-attention gate of EnhancedFeaturePropagation (models/pointnet2_utils.py:279-280):
// out = x * sigmoid(a), elementwise on bf16 rows, as ONE pass (the reference and autograd run
// sigmoid and the product separately, and three passes in backward).
__global__ __launch_bounds__(kThreads) void gate_kernel(const uint4 *__restrict__ x, const uint4 *__restrict__ a,
                                                         uint4 *__restrict__ out, long nvec)
{
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        float fx[8], fa[8];
        unpack8(x[e], fx);
        unpack8(a[e], fa);
#pragma unroll
        for (int i = 0; i < 8; ++i) fx[i] = fx[i] / (1.0f + __expf(-fa[i]));
        out[e] = pack8(fx);
    }
}

// dx = g * sigmoid(a),  da = g * x * s * (1 - s)
__global__ __launch_bounds__(kThreads) void gate_bwd_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ x,
                                                             const uint4 *__restrict__ a, uint4 *__restrict__ dx,
                                                             uint4 *__restrict__ da, long nvec)
{
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        float fg[8], fx[8], fa[8], d1[8], d2[8];
        unpack8(g[e], fg);
        unpack8(x[e], fx);
        unpack8(a[e], fa);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sgm = 1.0f / (1.0f + __expf(-fa[i]));
            d1[i] = fg[i] * sgm;
            d2[i] = fg[i] * fx[i] * sgm * (1.0f - sgm);
        }
        dx[e] = pack8(d1);
        da[e] = pack8(d2);
    }
}

inline int grid_for(long work, int per_block = kThreads, int cap = 4096)
{
    long blocks = (work + per_block - 1) / per_block;
    return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

inline bool bad_c(int C) { return C <= 0 || (C & 7) != 0 || C > 2048; }

}  // namespace

extern "C" int pcb_colstats_bf16(const void *y, long rows, int C, float *sums, void *stream)
{
    if (!y || !sums || rows <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_c(C)) return PCB_ERR_UNSUPPORTED;
    const int RT = kThreads / (C >> 3);
    hipLaunchKernelGGL(colstats_kernel, dim3(grid_for(rows, RT * 8, 2048)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const uint4 *)y, rows, C, sums);
    return pcb_check_launch();
}

extern "C" int pcb_bn_finalize(const float *sums, int nparts, long rows, long count, int C, const float *gamma,
                               const float *beta, const float *bias, float *running_mean,
                               float *running_var, float momentum, float eps, int training,
                               float *scale, float *shift, float *mean, float *invstd,
                               long long *num_batches_tracked, void *stream)
{
    if (!scale || !shift || !mean || !invstd || C <= 0 || rows <= 0) return PCB_ERR_INVALID_ARG;
    if (training ? (!sums || nparts < 1) : (!running_mean || !running_var)) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, sums, nparts,
                       rows, count > 0 ? count : rows, C, gamma, beta, bias, running_mean, running_var, momentum, eps, training,
                       scale, shift, mean, invstd, num_batches_tracked);
    return pcb_check_launch();
}

extern "C" int pcb_bn_act_bf16(const void *y, const float *scale, const float *shift, long rows, int C,
                               int act, void *z, void *stream)
{
    if (!y || !scale || !shift || !z || rows <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_c(C)) return PCB_ERR_UNSUPPORTED;
    const long nvec = rows * (C >> 3);
    hipLaunchKernelGGL(bn_act_kernel, dim3(grid_for(nvec)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)y, scale, shift, C, slope_of(act), (uint4 *)z, nvec);
    return pcb_check_launch();
}

extern "C" int pcb_bn_act_max_bf16(const void *y, const float *scale, const float *shift, long groups,
                                   int ns, int C, int act, void *out, unsigned char *argmax,
                                   void *stream)
{
    if (!y || !scale || !shift || !out || !argmax || groups <= 0 || ns <= 0 || ns > 255) return PCB_ERR_INVALID_ARG;
    if (bad_c(C)) return PCB_ERR_UNSUPPORTED;
    const long nvec = groups * (C >> 3);
    hipLaunchKernelGGL(bn_act_max_kernel, dim3(grid_for(nvec, kThreads, 8192)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const uint4 *)y, scale, shift, C, ns, slope_of(act), (uint4 *)out,
                       argmax, nvec);
    return pcb_check_launch();
}

extern "C" int pcb_bn_act_bwd_bf16(const void *dz, const void *y, const float *scale,
                                   const float *shift, const float *mean, const float *invstd,
                                   long rows, int C, int act, int use_batch_stats, float *sums,
                                   void *dy, void *stream)
{
    if (!dz || !y || !scale || !shift || !mean || !invstd || !sums || !dy || rows <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_c(C)) return PCB_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int RT = kThreads / (C >> 3);
    // sums [2,C] must be zero on entry; it returns (dbeta, dgamma) = (s1, s2)
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(grid_for(rows, RT * 8, 2048)), dim3(kThreads), 0, st,
                       (const uint4 *)dz, (const uint4 *)y, scale, shift, mean, invstd, rows, C, slope_of(act), sums);
    const long nvec = rows * (C >> 3);
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(grid_for(nvec)), dim3(kThreads), 0, st,
                       (const uint4 *)dz, (const uint4 *)y, scale, shift, mean, invstd, sums, rows, C, slope_of(act),
                       use_batch_stats, (uint4 *)dy, nvec);
    return pcb_check_launch();
}

extern "C" int pcb_bn_act_max_bwd_bf16(const float *dout, const unsigned char *argmax, const void *y,
                                       const float *scale, const float *shift, const float *mean,
                                       const float *invstd, long groups, int ns, int C, int act,
                                       int use_batch_stats, float *sums, void *dy, void *stream)
{
    if (!dout || !argmax || !y || !scale || !shift || !mean || !invstd || !sums || !dy || groups <= 0 ||
        ns <= 0 || ns > 255)
        return PCB_ERR_INVALID_ARG;
    if (bad_c(C)) return PCB_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int RT = kThreads / (C >> 3);
    hipLaunchKernelGGL(bn_max_bwd_reduce_kernel, dim3(grid_for(groups, RT * 4, 2048)), dim3(kThreads), 0, st,
                       dout, argmax, (const uint4 *)y, scale, shift, mean, invstd, groups, C, ns, slope_of(act), sums);
    const long rows = groups * ns;
    const long nvec = rows * (C >> 3);
    hipLaunchKernelGGL(bn_max_bwd_apply_kernel, dim3(grid_for(nvec)), dim3(kThreads), 0, st, dout, argmax,
                       (const uint4 *)y, scale, shift, mean, invstd, sums, rows, C, ns, slope_of(act),
                       use_batch_stats, (uint4 *)dy, nvec);
    return pcb_check_launch();
}

extern "C" int pcb_group_rows_bf16(const float *xyz, const float *new_xyz, const void *feat,
                                   const int64_t *idx, int B, int N, int S, int ns, int C, int Kp,
                                   void *out, void *stream)
{
    if (!xyz || !new_xyz || !idx || !out || B <= 0 || N <= 0 || S <= 0 || ns <= 0 || C < 0) return PCB_ERR_INVALID_ARG;
    if ((C > 0 && !feat) || Kp < C + 3 || (Kp & 7) != 0) return PCB_ERR_INVALID_ARG;
    const long nchunk = (long)B * S * ns * (Kp >> 3);
    hipLaunchKernelGGL(group_rows_bf16_kernel, dim3(grid_for(nchunk, kThreads, 8192)), dim3(kThreads), 0,
                       (hipStream_t)stream, xyz, new_xyz, (const u16 *)feat, idx, N, S, ns, C, Kp,
                       (u16 *)out, nchunk);
    return pcb_check_launch();
}

extern "C" int pcb_group_rows_bf16_bwd(const void *grad_rows, const int64_t *idx, int B, int N, int S,
                                       int ns, int C, int Kp, float *grad_feat, void *stream)
{
    if (!grad_rows || !idx || !grad_feat || B <= 0 || N <= 0 || S <= 0 || ns <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    const long total = (long)B * S * ns * C;
    hipLaunchKernelGGL(group_rows_bf16_bwd_kernel, dim3(grid_for(total, kThreads, 8192)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const u16 *)grad_rows, idx, N, S, ns, C, Kp, grad_feat, total);
    return pcb_check_launch();
}

extern "C" int pcb_bn_act_bwd_reduce_bf16(const void *dz, const void *y, const float *scale,
                                          const float *shift, const float *mean, const float *invstd,
                                          long rows, int C, int act, float *sums, void *stream)
{
    if (!dz || !y || !scale || !shift || !mean || !invstd || !sums || rows <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_c(C)) return PCB_ERR_UNSUPPORTED;
    const int RT = kThreads / (C >> 3);
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(grid_for(rows, RT * 8, 2048)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const uint4 *)dz, (const uint4 *)y, scale, shift, mean, invstd,
                       rows, C, slope_of(act), sums);
    return pcb_check_launch();
}

extern "C" int pcb_bn_act_max_bwd_reduce_bf16(const float *dout, const unsigned char *argmax, const void *y,
                                              const float *scale, const float *shift, const float *mean,
                                              const float *invstd, long groups, int ns, int C, int act,
                                              float *sums, void *stream)
{
    if (!dout || !argmax || !y || !scale || !shift || !mean || !invstd || !sums || groups <= 0 || ns <= 0 ||
        ns > 255)
        return PCB_ERR_INVALID_ARG;
    if (bad_c(C)) return PCB_ERR_UNSUPPORTED;
    const int RT = kThreads / (C >> 3);
    hipLaunchKernelGGL(bn_max_bwd_reduce_kernel, dim3(grid_for(groups, RT * 4, 2048)), dim3(kThreads), 0,
                       (hipStream_t)stream, dout, argmax, (const uint4 *)y, scale, shift, mean, invstd, groups,
                       C, ns, slope_of(act), sums);
    return pcb_check_launch();
}

extern "C" int pcb_gate_bf16(const void *x, const void *a, void *out, long n, void *stream)
{
    if (!x || !a || !out || n <= 0 || (n & 7)) return PCB_ERR_INVALID_ARG;
    const long nvec = n >> 3;
    hipLaunchKernelGGL(gate_kernel, dim3(grid_for(nvec, kThreads, 8192)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)x, (const uint4 *)a, (uint4 *)out, nvec);
    return pcb_check_launch();
}

extern "C" int pcb_gate_bwd_bf16(const void *g, const void *x, const void *a, void *dx, void *da, long n, void *stream)
{
    if (!g || !x || !a || !dx || !da || n <= 0 || (n & 7)) return PCB_ERR_INVALID_ARG;
    const long nvec = n >> 3;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(nvec, kThreads, 8192)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)g, (const uint4 *)x, (const uint4 *)a, (uint4 *)dx, (uint4 *)da, nvec);
    return pcb_check_launch();
}
