// Channels-last row kernels around the pointwise-MLP GEMMs (HBM-bound byte movers), in both
// arithmetic modes of the engine: bf16 rows (entry points *_bf16) and fp32 rows (*_f32).
//
// The reference evaluates every shared MLP as Conv2d/Conv1d(1x1) -> BatchNorm -> ReLU on fp32
// [B,C,S,ns] tensors, with a torch.max over the neighbour axis at the end of a set-abstraction
// level (models/pointnet2_utils.py:149-154, :207-209, :353-356; DGCNN.py:134-148 with LeakyReLU).
// ATen runs that as 5-7 full passes over the activation per layer.  Here activations are rows
// [rows, C]; the pre-BatchNorm GEMM output y is the only tensor kept per layer, and
//   colstats        one pass:  sum(y), sum(y*y) per channel                (train-mode statistics)
//   bn_finalize     C threads: scale/shift, running-stat update            (BatchNorm bookkeeping)
//   bn_act          one pass:  z = act(y*scale + shift)                    (operand of the next GEMM)
//   bn_act_max      one pass:  max over the ns rows of a group + arg-max   (SA / EdgeConv pooling)
//   *_bwd_reduce    one pass:  s1 = sum(du), s2 = sum(du * xhat)           (BatchNorm backward sums)
//   *_bwd_apply     one pass:  dy = scale * (du - s1/R - xhat * s2/R)      (operand of dgrad/wgrad)
// Every kernel moves 16-byte vectors (8 bf16 or 4 fp32 channels per lane, rowvec.h) and keeps fp32
// in registers.
#include "gemm_shared.h"
#include "rowvec.h"

namespace {

// activation codes shared with the host: 0 none, 1 ReLU, 2 LeakyReLU(0.2)
// (applied as one select on a per-launch slope: no per-element tests of the activation code)
inline float slope_of(int act) { return act == 1 ? 0.0f : (act == 2 ? 0.2f : 1.0f); }  // host side
__device__ __forceinline__ float act_fwd(float u, float slope) { return u > 0.0f ? u : fmaf(slope, u, 0.0f); }
__device__ __forceinline__ float act_grad(float u, float slope) { return u > 0.0f ? 1.0f : slope; }

constexpr int kThreads = 256;

// Block-level combine shared by the three reduction kernels: every thread holds E partial sums of
// two moments for its channel chunk; the block adds its row-lanes through LDS.  The result either
// goes to the block's own slab of `sums` ([gridDim.x][2][C], slabs != 0: no atomics, the finalize
// kernels add the slabs in a fixed order) or is added to a single [2][C] slab with fp32 atomics.
template <int E>
__device__ __forceinline__ void block_moments(const float *s, const float *q, int C, float *red, float *sums, int slabs)
{
    const int CT = C / E;
    const int RT = kThreads / CT;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        red[threadIdx.x * 2 * E + i] = s[i];
        red[threadIdx.x * 2 * E + E + i] = q[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * C; o += kThreads) {
        const int m = o / C, c = o % C;
        float a = 0.0f;
        for (int r = 0; r < RT; ++r) a += red[(r * CT + c / E) * 2 * E + m * E + (c % E)];
        if (slabs)
            sums[(long)blockIdx.x * 2 * C + o] = a;
        else
            atomicAdd(&sums[o], a);
    }
}

// ---------------------------------------------------------------------------------------------
// Column statistics: sums[0][c] = sum_r y[r][c], sums[1][c] = sum_r y[r][c]^2   (C % E == 0, C/E <= 256)
// Lane t owns channel chunk t % CT and walks rows t / CT, + RT, ...
// A launch covers C columns of rows that are `ld` vectors apart (C == the row width, or a column block
// of a wider matrix: blockIdx.y picks the block; the moments of column c go to sums[c], sums[Call + c]).
template <typename T>
__global__ __launch_bounds__(kThreads) void colstats_kernel(const uint4 *__restrict__ y, long rows, int C, int Call,
                                                             float *__restrict__ sums, int slabs)
{
    // slabs != 0: sums is [gridDim.x][2][Call] and every workgroup along x writes its own slab (no atomics: the caller
    // adds the slabs in order -- the reproducible mode); else one [2][Call] slab accumulated with atomics.
    constexpr int E = RowVec<T>::E;
    __shared__ float red[kThreads * 2 * E];
    const int c0 = blockIdx.y * C;             // first column of this block
    const int Cb = Call - c0 < C ? Call - c0 : C;  // columns in this block
    const int CT = Cb / E;
    const int RT = kThreads / CT;          // row-lanes per block (CT <= 256)
    const int cc = threadIdx.x % CT;
    const int rl = threadIdx.x / CT;
    const long ld = Call / E;
    float s[E], q[E];
#pragma unroll
    for (int i = 0; i < E; ++i) s[i] = q[i] = 0.0f;
    if (rl < RT) {
        for (long r = (long)blockIdx.x * RT + rl; r < rows; r += (long)gridDim.x * RT) {
            float f[E];
            RowVec<T>::unpack(y[r * ld + c0 / E + cc], f);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                s[i] += f[i];
                q[i] = fmaf(f[i], f[i], q[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) {
        red[threadIdx.x * 2 * E + i] = s[i];
        red[threadIdx.x * 2 * E + E + i] = q[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * Cb; o += kThreads) {
        const int m = o / Cb, c = o % Cb;
        float a = 0.0f;
        for (int r = 0; r < RT; ++r) a += red[(r * CT + c / E) * 2 * E + m * E + (c % E)];
        if (slabs)
            sums[((long)blockIdx.x * 2 + m) * Call + c0 + c] = a;
        else
            atomicAdd(&sums[(long)m * Call + c0 + c], a);
    }
}

// out[i] = sum_k slabs[k][i], i < n, in a fixed order (SyncBatchNorm: the local totals that travel through the
// all-reduce between a GEMM's statistics epilogue and the finalize kernel; ops.sum_slabs: every per-workgroup partial
// sum of the package that must not go through an ATen reduction).  Block = 32 outputs x 32 slab-lanes: lane p adds
// slabs p, p + 32, ... in fp64 (coalesced 128-byte reads across the outputs), the 32 lanes of an output meet in LDS in
// lane order.  (One thread per output walking all slabs -- the first version -- made a chain of nparts dependent L2
// round trips: 56 us for 2048 slabs of 64 floats.)
__global__ __launch_bounds__(1024) void sum_slabs_kernel(const float *__restrict__ slabs, int nparts, int n,
                                                          float *__restrict__ out)
{
    __shared__ double red[32][33];
    const int il = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + il;
    double a = 0.0;
    if (i < n) {
        for (int k0 = pl; k0 < nparts; k0 += 32 * 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 32 * u;
                const float x = slabs[(long)(k < nparts ? k : nparts - 1) * n + i];
                v[u] = k < nparts ? x : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) a += (double)v[u];
        }
    }
    red[pl][il] = a;
    __syncthreads();
    if (pl == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][il];
        out[i] = (float)t;
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm bookkeeping for one layer.
// training: batch statistics from `sums` over `rows` rows of y = x W^T (the conv bias, which the
// GEMM does not add because it cancels inside a train-mode BatchNorm, is added to the mean that
// goes into running_mean); running_var gets the unbiased variance (count/(count-1)), as
// torch.nn.BatchNorm does.  eval: running statistics; the bias is folded into the shift.
// Outputs: scale = gamma*invstd, shift = beta - (mean_y)*scale [+ bias*scale in eval], and
// mean_y / invstd for the backward pass.
// Block = 32 channels x 32 slab-lanes: the partial slabs are added by 32 lanes per channel
// (coalesced 128-byte reads across the channels), combined through LDS in a fixed order.  The
// slab totals and var = E[y^2] - mean^2 are formed in fp64: the subtraction cancels leading digits
// whenever |mean| >> std, and the fp32 mode promises logits within 1e-4 of the reference.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float *__restrict__ sums, int nparts, long rows, long count, int C, const float *__restrict__ gamma,
    const float *__restrict__ beta, const float *__restrict__ bias, float *__restrict__ running_mean,
    float *__restrict__ running_var, float momentum, float eps, int training, float *__restrict__ scale,
    float *__restrict__ shift, float *__restrict__ mean_out, float *__restrict__ invstd_out,
    long long *__restrict__ num_batches_tracked, float *__restrict__ centre, int cmode)
{
    // centre / cmode: the rows were stored CENTRED, y_c = x W^T - centre (gemm.hip RedArgs::centre), and `sums` are the
    // moments of y_c.  Everything downstream (scale, shift, mean, invstd -> the consumers' prologues, the backward
    // constants) lives in that frame and needs no change; only running_mean wants the true mean back.
    //   training, cmode 1: running_mean from mean_c + centre (+ bias); then, where |mean_c| > std/4, centre += mean_c -- this
    //                      batch's mean is the next step's centre (a persistent per-layer buffer of the caller)
    //   training, cmode 2: PROBE -- centre += mean_c and nothing else (the caller then runs the GEMM again with a centre
    //                      that is this batch's mean: the first step of a fresh layer)
    //   eval,     cmode 1: centre (output, the caller's scratch) = running_mean - bias, the exact centre; the GEMM that
    //                      follows subtracts it, so mean_y = 0 in that frame
    __shared__ double red[2][32][32];
    // nn.BatchNorm's step counter (num_batches_tracked += 1 in a train-mode forward), bumped here
    // instead of by one more tiny launch per module
    if (num_batches_tracked && cmode != 2 && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (training && c < C) {
        // sums is [nparts][2][C]; eight slabs' loads are issued before the first add (the adds keep
        // their order): a dependent load per step made this 10 us kernel a chain of L2 round trips
        for (int k0 = pl; k0 < nparts; k0 += 32 * 8) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // unconditional loads from a clamped slab (a load under a branch is waited for on the spot)
                const int k = k0 + 32 * u, kc = k < nparts ? k : nparts - 1;
                const float va = sums[((long)kc * 2 + 0) * C + c], vb = sums[((long)kc * 2 + 1) * C + c];
                a[u] = k < nparts ? va : 0.0f;
                b[u] = k < nparts ? vb : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s1 += (double)a[u];
                s2 += (double)b[u];
            }
        }
    }
    red[0][pl][cl] = s1;
    red[1][pl][cl] = s2;
    __syncthreads();
    if (pl != 0 || c >= C) return;
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        s1 += red[0][k][cl];
        s2 += red[1][k][cl];
    }
    const float b = bias ? bias[c] : 0.0f;
    const float ce = (centre && cmode && training) ? centre[c] : 0.0f;
    float mean_y, invstd;
    if (training) {
        const double n = (double)rows;
        const double mu = s1 / n;
        double var = s2 / n - mu * mu;
        var = var < 0.0 ? 0.0 : var;
        mean_y = (float)mu;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        // hysteresis: the centre only has to be within a fraction of a standard deviation of the mean, and a centre that
        // does not move from step to step keeps the rounding of y -- and with it everything downstream -- reproducible
        // between two passes over the same batch (an eager step and the replays of its captured twin, a repeated test)
        if (centre && cmode && (cmode == 2 || fabsf(mean_y) * invstd > 0.25f)) centre[c] = ce + mean_y;
        if (cmode == 2) return;
        if (running_mean) {
            const double m = (double)count;
            const float unb = (float)(count > 1 ? var * (m / (m - 1.0)) : var);
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * ((mean_y + ce) + b);
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unb;
        }
    } else {
        mean_y = running_mean[c] - b;  // BN(y + b) with running stats == (y - (rm - b)) * invstd
        invstd = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
        if (centre && cmode) {
            centre[c] = mean_y;
            mean_y = 0.0f;
        }
    }
    const float g = gamma ? gamma[c] : 1.0f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.0f) - mean_y * sc;
    mean_out[c] = mean_y;
    invstd_out[c] = invstd;
}

// ---------------------------------------------------------------------------------------------
// z = act(y*scale + shift), elementwise over [rows, C].
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_act_kernel(const uint4 *__restrict__ y,
                                                           const float *__restrict__ scale,
                                                           const float *__restrict__ shift, int C,
                                                           float act, uint4 *__restrict__ z, long nvec)
{
    constexpr int E = RowVec<T>::E;
    const int CT = C / E;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const int c0 = (int)(e % CT) * E;
        float f[E];
        RowVec<T>::unpack(y[e], f);
#pragma unroll
        for (int i = 0; i < E; ++i) f[i] = act_fwd(fmaf(f[i], scale[c0 + i], shift[c0 + i]), act);
        z[e] = RowVec<T>::pack(f);
    }
}

// out[g][c] = max_j act(y[g*ns + j][c]*scale + shift), arg[g][c] = first j attaining it.
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_act_max_kernel(const uint4 *__restrict__ y,
                                                               const float *__restrict__ scale,
                                                               const float *__restrict__ shift,
                                                               int C, int ns, float act,
                                                               uint4 *__restrict__ out,
                                                               unsigned char *__restrict__ arg,
                                                               long nvec /* groups * C/E */)
{
    constexpr int E = RowVec<T>::E;
    const int CT = C / E;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const int cc = (int)(e % CT);
        const long g = e / CT;
        float sc[E], sh[E], best[E];
        int bj[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            sc[i] = scale[cc * E + i];
            sh[i] = shift[cc * E + i];
            best[i] = -INFINITY;
            bj[i] = 0;
        }
        for (int j = 0; j < ns; ++j) {
            float f[E];
            RowVec<T>::unpack(y[(g * ns + j) * CT + cc], f);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float v = act_fwd(fmaf(f[i], sc[i], sh[i]), act);
                if (v > best[i]) {
                    best[i] = v;
                    bj[i] = j;
                }
            }
        }
        out[e] = RowVec<T>::pack(best);
        unsigned long long packed = 0;
#pragma unroll
        for (int i = 0; i < E; ++i) packed |= (unsigned long long)(bj[i] & 0xff) << (8 * i);
        store_arg_bytes<E>(arg + e * E, packed);
    }
}

// ---------------------------------------------------------------------------------------------
// Backward sums for a dense upstream gradient dz [rows, C]:
// du = dz * act'(u), u = y*scale + shift;  s1 += du,  s2 += du * xhat,  xhat = (y - mean)*invstd.
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_act_bwd_reduce_kernel(
    const uint4 *__restrict__ dz, const uint4 *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
    long rows, int C, float act, float *__restrict__ sums, int slabs)
{
    constexpr int E = RowVec<T>::E;
    __shared__ float red[kThreads * 2 * E];
    const int CT = C / E;
    const int RT = kThreads / CT;
    const int cc = threadIdx.x % CT;
    const int rl = threadIdx.x / CT;
    float s1[E], s2[E];
#pragma unroll
    for (int i = 0; i < E; ++i) s1[i] = s2[i] = 0.0f;
    if (rl < RT) {
        float sc[E], sh[E], mu[E], is[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            sc[i] = scale[cc * E + i];
            sh[i] = shift[cc * E + i];
            mu[i] = mean[cc * E + i];
            is[i] = invstd[cc * E + i];
        }
        // four rows' loads (8 x 16 bytes) in flight per lane and step; rows past the end are read from
        // the last row and masked, so the loads stay unconditional
        const long step = (long)gridDim.x * RT;
        for (long r0 = (long)blockIdx.x * RT + rl; r0 < rows; r0 += 4 * step) {
            uint4 vy[4], vd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long r = r0 + u * step;
                const long rs = r < rows ? r : rows - 1;
                vy[u] = y[rs * CT + cc];
                vd[u] = dz[rs * CT + cc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float fy[E], fd[E];
                RowVec<T>::unpack(vy[u], fy);
                RowVec<T>::unpack(vd[u], fd);
                const float keep = (r0 + u * step < rows) ? 1.0f : 0.0f;
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const float du = keep * fd[i] * act_grad(fmaf(fy[i], sc[i], sh[i]), act);
                    s1[i] += du;
                    s2[i] = fmaf(du, (fy[i] - mu[i]) * is[i], s2[i]);
                }
            }
        }
    }
    block_moments<E>(s1, s2, C, red, sums, slabs);
}

// dy = scale * (du - s1/R - xhat * s2/R)      (BatchNorm backward, batch statistics)
// eval mode (use_batch_stats == 0): dy = scale * du.
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_act_bwd_apply_kernel(
    const uint4 *__restrict__ dz, const uint4 *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ sums, long rows, int C, float act, int use_batch_stats,
    uint4 *__restrict__ dy)
{
    // a lane owns one 16-byte column chunk and walks rows (as the reduction kernels do): the six per-column
    // constants of its chunk live in registers instead of being re-read for every element (the flat
    // element-stride form spent its time on those loads: 0.6 TB/s on [65536, 320] rows)
    constexpr int E = RowVec<T>::E;
    const int CT = C / E;
    const int RT = kThreads / CT;
    const int cc = threadIdx.x % CT;
    const int rl = threadIdx.x / CT;
    if (rl >= RT) return;
    const float invR = 1.0f / (float)rows;
    float sc[E], sh[E], mu[E], is[E], c1[E], c2[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int c = cc * E + i;
        sc[i] = scale[c];
        sh[i] = shift[c];
        mu[i] = mean[c];
        is[i] = invstd[c];
        c1[i] = use_batch_stats ? sums[c] * invR : 0.0f;
        c2[i] = use_batch_stats ? sums[C + c] * invR : 0.0f;
    }
    const long step = (long)gridDim.x * RT;
    for (long r0 = (long)blockIdx.x * RT + rl; r0 < rows; r0 += 2 * step) {
        uint4 vy[2], vd[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long r = r0 + u * step < rows ? r0 + u * step : rows - 1;
            vy[u] = y[r * CT + cc];
            vd[u] = dz[r * CT + cc];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long r = r0 + u * step;
            if (r >= rows) break;
            float fy[E], fd[E];
            RowVec<T>::unpack(vy[u], fy);
            RowVec<T>::unpack(vd[u], fd);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float du = fd[i] * act_grad(fmaf(fy[i], sc[i], sh[i]), act);
                const float xh = (fy[i] - mu[i]) * is[i];
                const float corr = use_batch_stats ? fmaf(xh, c2[i], c1[i]) : 0.0f;
                fd[i] = sc[i] * (du - corr);
            }
            dy[r * CT + cc] = RowVec<T>::pack(fd);
        }
    }
}

// Pooled layers: the upstream gradient dout [groups, C] (fp32) reaches only the arg-max row of each
// (group, channel).  Sums over those rows:
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_max_bwd_reduce_kernel(
    const float *__restrict__ dout, const unsigned char *__restrict__ arg, const T *__restrict__ y,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ invstd, long groups, int C, int ns, float act, float *__restrict__ sums, int slabs)
{
    constexpr int E = RowVec<T>::E;
    __shared__ float red[kThreads * 2 * E];
    const int CT = C / E;
    const int RT = kThreads / CT;
    const int cc = threadIdx.x % CT;
    const int rl = threadIdx.x / CT;
    float s1[E], s2[E];
#pragma unroll
    for (int i = 0; i < E; ++i) s1[i] = s2[i] = 0.0f;
    if (rl < RT) {
        for (long g = (long)blockIdx.x * RT + rl; g < groups; g += (long)gridDim.x * RT) {
            const unsigned long long a = load_arg_bytes<E>(arg + (g * CT + cc) * E);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int c = cc * E + i;
                const int j = (int)((a >> (8 * i)) & 0xff);
                const float yv = RowVec<T>::one(y + (g * ns + j) * (long)C + c);
                const float du = dout[g * C + c] * act_grad(fmaf(yv, scale[c], shift[c]), act);
                s1[i] += du;
                s2[i] = fmaf(du, (yv - mean[c]) * invstd[c], s2[i]);
            }
        }
    }
    block_moments<E>(s1, s2, C, red, sums, slabs);
}

// dy[g*ns + j][c] = scale * ((j == arg ? dout*act' : 0) - s1/R - xhat*s2/R), dense [rows, C].
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_max_bwd_apply_kernel(
    const float *__restrict__ dout, const unsigned char *__restrict__ arg, const uint4 *__restrict__ y,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ sums, long rows, int C, int ns,
    float act, int use_batch_stats, uint4 *__restrict__ dy, long nvec /* rows * C/E */)
{
    constexpr int E = RowVec<T>::E;
    const int CT = C / E;
    const float invR = 1.0f / (float)rows;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const int cc = (int)(e % CT);
        const long r = e / CT;
        const long g = r / ns;
        const int j = (int)(r % ns);
        const unsigned long long a = load_arg_bytes<E>(arg + (g * CT + cc) * E);
        float fy[E], o[E];
        RowVec<T>::unpack(y[e], fy);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int c = cc * E + i;
            float du = 0.0f;
            if ((int)((a >> (8 * i)) & 0xff) == j)
                du = dout[g * C + c] * act_grad(fmaf(fy[i], scale[c], shift[c]), act);
            const float xh = (fy[i] - mean[c]) * invstd[c];
            const float corr = use_batch_stats ? fmaf(xh, sums[C + c] * invR, sums[c] * invR) : 0.0f;
            o[i] = scale[c] * (du - corr);
        }
        dy[e] = RowVec<T>::pack(o);
    }
}

// ---------------------------------------------------------------------------------------------
// Grouping into GEMM rows: out[(b,s,j)][0:C] = feat[b, idx][0:C], [C:C+3] = xyz[b,idx]-new_xyz[b,s],
// zero up to Kp.  Features FIRST (so that 16-byte chunks of a feature row stay aligned); the host
// permutes the weight columns to match (the reference's order is coordinates first, :56 / :347).
template <typename T>
__global__ __launch_bounds__(kThreads) void group_rows_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const T *__restrict__ feat,
    const int64_t *__restrict__ idx, int N, int S, int ns, int C, int Kp, T *__restrict__ out,
    long nchunk /* rows * Kp/E */)
{
    constexpr int E = RowVec<T>::E;
    const int KT = Kp / E;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nchunk; e += (long)gridDim.x * kThreads) {
        const int k0 = (int)(e % KT) * E;
        const long row = e / KT;         // (b*S + s)*ns + j
        const long bs = row / ns;
        const long b = bs / S;
        const int i = clamp_index(idx[row], N);
        uint4 v;
        if ((C % E) == 0 && k0 + E <= C) {
            v = *reinterpret_cast<const uint4 *>(feat + (b * N + i) * (long)C + k0);
        } else {
            float f[E];
#pragma unroll
            for (int t = 0; t < E; ++t) {
                const int k = k0 + t;
                float val = 0.0f;
                if (k < C)
                    val = RowVec<T>::one(feat + (b * N + i) * (long)C + k);
                else if (k < C + 3)
                    val = __fsub_rn(xyz[(b * N + i) * 3 + (k - C)], new_xyz[bs * 3 + (k - C)]);
                f[t] = val;
            }
            v = RowVec<T>::pack(f);
        }
        *reinterpret_cast<uint4 *>(out + row * (long)Kp + k0) = v;
    }
}

// grad_feat[b, idx, c] += g[row][c]   (fp32 accumulation, c < C)
template <typename T>
__global__ __launch_bounds__(kThreads) void group_rows_bwd_kernel(
    const T *__restrict__ g, const int64_t *__restrict__ idx, int N, int S, int ns, int C, int Kp,
    float *__restrict__ gfeat, long total /* rows * C */)
{
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
        const int c = (int)(e % C);
        const long row = e / C;
        const long b = row / ((long)S * ns);
        const int i = clamp_index(idx[row], N);
        atomicAdd(&gfeat[(b * N + i) * (long)C + c], RowVec<T>::one(g + row * (long)Kp + c));
    }
}

// Channel-attention gate of EnhancedFeaturePropagation (models/pointnet2_utils.py:279-280):
// out = x * sigmoid(a), elementwise on rows, as ONE pass (the reference and autograd run
// sigmoid and the product separately, and three passes in backward).
template <typename T>
__global__ __launch_bounds__(kThreads) void gate_kernel(const uint4 *__restrict__ x, const uint4 *__restrict__ a,
                                                         uint4 *__restrict__ out, long nvec)
{
    constexpr int E = RowVec<T>::E;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        float fx[E], fa[E];
        RowVec<T>::unpack(x[e], fx);
        RowVec<T>::unpack(a[e], fa);
#pragma unroll
        for (int i = 0; i < E; ++i) fx[i] = fx[i] / (1.0f + (E == 8 ? __expf(-fa[i]) : expf(-fa[i])));
        out[e] = RowVec<T>::pack(fx);
    }
}

// dx = g * sigmoid(a),  da = g * x * s * (1 - s)
template <typename T>
__global__ __launch_bounds__(kThreads) void gate_bwd_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ x,
                                                             const uint4 *__restrict__ a, uint4 *__restrict__ dx,
                                                             uint4 *__restrict__ da, long nvec)
{
    constexpr int E = RowVec<T>::E;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        float fg[E], fx[E], fa[E], d1[E], d2[E];
        RowVec<T>::unpack(g[e], fg);
        RowVec<T>::unpack(x[e], fx);
        RowVec<T>::unpack(a[e], fa);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const float sgm = 1.0f / (1.0f + (E == 8 ? __expf(-fa[i]) : expf(-fa[i])));
            d1[i] = fg[i] * sgm;
            d2[i] = fg[i] * fx[i] * sgm * (1.0f - sgm);
        }
        dx[e] = RowVec<T>::pack(d1);
        da[e] = RowVec<T>::pack(d2);
    }
}

// Dropout on rows (nn.Dropout(0.5) in front of the segmentation heads' last conv, models/model.py:97, :52): out = x * keep /
// (1 - p) with a STATELESS mask: keep bits come from a 64-bit mix of (seed, vector index), 8 bits per element (p in steps of
// 1/256), so the backward pass applies the very same launch to the gradient with the same seed and nothing is stored.
// ATen: a fused kernel + a bool mask tensor forward (39 us for [262144,128] bf16), masked_scale backward (70 us).
// The seed is read from device memory: a captured step replays the launch and draws fresh masks from a seed tensor that an
// RNG op inside the graph refreshes.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ __launch_bounds__(kThreads) void dropout_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ out, long nvec,
                                                            const long long *__restrict__ seed, unsigned threshold,
                                                            float scale)
{
    constexpr int E = RowVec<T>::E;
    const unsigned long long s = mix64((unsigned long long)seed[0]);
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const unsigned long long h = mix64(s ^ ((unsigned long long)e * 0xd1342543de82ef95ull));
        float f[E];
        RowVec<T>::unpack(x[e], f);
#pragma unroll
        for (int i = 0; i < E; ++i) f[i] = ((unsigned)(h >> (8 * i)) & 0xffu) >= threshold ? f[i] * scale : 0.0f;
        out[e] = RowVec<T>::pack(f);
    }
}

// Raw fp32 input columns (coordinates, colours: [R, k] with k = 3) as a zero-padded GEMM operand [R, kp] of
// the row type: cast + pad in one pass (ATen: a cast, a zero fill and a strided copy).
template <typename T>
__global__ __launch_bounds__(kThreads) void pad_rows_kernel(const float *__restrict__ x, long ld, int k, int kp,
                                                             T *__restrict__ out, long nvec)
{
    constexpr int E = RowVec<T>::E;
    const int KT = kp / E;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const long r = e / KT;
        const int j0 = (int)(e - r * KT) * E;
        float f[E];
#pragma unroll
        for (int i = 0; i < E; ++i) f[i] = j0 + i < k ? x[r * ld + j0 + i] : 0.0f;
        *reinterpret_cast<uint4 *>(out + r * (long)kp + j0) = RowVec<T>::pack(f);
    }
}

inline int grid_for(long work, int per_block = kThreads, int cap = 4096)
{
    long blocks = (work + per_block - 1) / per_block;
    return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

template <typename T>
inline bool bad_c(int C)
{
    constexpr int E = RowVec<T>::E;
    return C <= 0 || (C % E) != 0 || C > 256 * E;
}

// ---- typed launchers (the extern "C" entry points below are their two instantiations) -----------
template <typename T>
int colstats(const void *y, long rows, int C, float *sums, int nparts, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!y || !sums || rows <= 0 || nparts < 0 || nparts > 2048) return PCB_ERR_INVALID_ARG;
    if (C <= 0 || (C % E) != 0) return PCB_ERR_UNSUPPORTED;
    // column blocks of at most 256 vectors (one lane per vector and row-lane)
    const int Cblk = C < 256 * E ? C : 256 * E;
    const int nblk = (C + Cblk - 1) / Cblk;
    const int RT = kThreads / (Cblk / E);
    // nparts > 0: the caller's slab count IS the grid along the rows (slab mode: [nparts][2][C], no atomics)
    hipLaunchKernelGGL(colstats_kernel<T>, dim3(nparts > 0 ? nparts : grid_for(rows, RT * 8, 2048), nblk), dim3(kThreads), 0,
                       (hipStream_t)stream, (const uint4 *)y, rows, Cblk, C, sums, nparts > 0 ? 1 : 0);
    pcb_account((double)sizeof(T) * rows * C);
    return pcb_check_launch();
}

template <typename T>
int bn_act(const void *y, const float *scale, const float *shift, long rows, int C, int act, void *z, void *stream)
{
    if (!y || !scale || !shift || !z || rows <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_c<T>(C)) return PCB_ERR_UNSUPPORTED;
    const long nvec = rows * (C / RowVec<T>::E);
    hipLaunchKernelGGL(bn_act_kernel<T>, dim3(grid_for(nvec)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)y, scale, shift, C, slope_of(act), (uint4 *)z, nvec);
    pcb_account(2.0 * sizeof(T) * rows * C);
    return pcb_check_launch();
}

template <typename T>
int bn_act_max(const void *y, const float *scale, const float *shift, long groups, int ns, int C, int act, void *out,
               unsigned char *argmax, void *stream)
{
    if (!y || !scale || !shift || !out || !argmax || groups <= 0 || ns <= 0 || ns > 255) return PCB_ERR_INVALID_ARG;
    if (bad_c<T>(C)) return PCB_ERR_UNSUPPORTED;
    const long nvec = groups * (C / RowVec<T>::E);
    hipLaunchKernelGGL(bn_act_max_kernel<T>, dim3(grid_for(nvec, kThreads, 8192)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const uint4 *)y, scale, shift, C, ns, slope_of(act), (uint4 *)out,
                       argmax, nvec);
    pcb_account((double)sizeof(T) * groups * ns * C + (sizeof(T) + 1.0) * groups * C);
    return pcb_check_launch();
}

template <typename T>
int bn_act_bwd_reduce(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                      const float *invstd, long rows, int C, int act, float *sums, int nparts, void *stream)
{
    if (!dz || !y || !scale || !shift || !mean || !invstd || !sums || rows <= 0) return PCB_ERR_INVALID_ARG;
    if (nparts < 1 || nparts > PCB_MAX_SLABS) return PCB_ERR_INVALID_ARG;
    if (bad_c<T>(C)) return PCB_ERR_UNSUPPORTED;
    const int RT = kThreads / (C / RowVec<T>::E);
    // nparts > 1: one workgroup per slab, every slab written (no atomics: thousands of workgroups adding
    // into the same 2C words serialise at the memory side -- measured 62 us for a 25 us pass);
    // nparts == 1: a single [2][C] slab, zero on entry, accumulated with atomics by a few workgroups
    const int grid = nparts > 1 ? nparts : grid_for(rows, RT * 32, 256);
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<T>, dim3(grid), dim3(kThreads), 0,
                       (hipStream_t)stream, (const uint4 *)dz, (const uint4 *)y, scale, shift, mean, invstd,
                       rows, C, slope_of(act), sums, nparts > 1 ? 1 : 0);
    pcb_account(2.0 * sizeof(T) * rows * C);
    return pcb_check_launch();
}

template <typename T>
int bn_act_max_bwd_reduce(const float *dout, const unsigned char *argmax, const void *y, const float *scale,
                          const float *shift, const float *mean, const float *invstd, long groups, int ns, int C,
                          int act, float *sums, int nparts, void *stream)
{
    if (!dout || !argmax || !y || !scale || !shift || !mean || !invstd || !sums || groups <= 0 || ns <= 0 ||
        ns > 255)
        return PCB_ERR_INVALID_ARG;
    if (nparts < 1 || nparts > PCB_MAX_SLABS) return PCB_ERR_INVALID_ARG;
    if (bad_c<T>(C)) return PCB_ERR_UNSUPPORTED;
    const int RT = kThreads / (C / RowVec<T>::E);
    const int grid = nparts > 1 ? nparts : grid_for(groups, RT * 16, 256);  // see bn_act_bwd_reduce
    hipLaunchKernelGGL(bn_max_bwd_reduce_kernel<T>, dim3(grid), dim3(kThreads), 0,
                       (hipStream_t)stream, dout, argmax, (const T *)y, scale, shift, mean, invstd, groups,
                       C, ns, slope_of(act), sums, nparts > 1 ? 1 : 0);
    pcb_account((5.0 + sizeof(T)) * groups * C);
    return pcb_check_launch();
}

template <typename T>
int bn_act_bwd(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
               const float *invstd, long rows, int C, int act, int use_batch_stats, float *sums, void *dy, void *stream)
{
    if (!dy) return PCB_ERR_INVALID_ARG;
    // sums [2,C] must be zero on entry; it returns (dbeta, dgamma) = (s1, s2)
    const int st = bn_act_bwd_reduce<T>(dz, y, scale, shift, mean, invstd, rows, C, act, sums, 1, stream);
    if (st != PCB_OK) return st;
    const int RT = kThreads / (C / RowVec<T>::E);
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<T>, dim3(grid_for(rows, RT * 4, 2048)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)dz, (const uint4 *)y, scale, shift, mean, invstd, sums, rows, C, slope_of(act),
                       use_batch_stats, (uint4 *)dy);
    pcb_account(3.0 * sizeof(T) * rows * C);
    return pcb_check_launch();
}

// the apply pass alone: sums [2,C] = (s1, s2) of bn_act_bwd_reduce, totals given by the caller (reproducible mode: the
// slab form of the reduction + pcb_sum_slabs)
template <typename T>
int bn_act_bwd_apply(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                     const float *invstd, const float *sums, long rows, int C, int act, int use_batch_stats, void *dy,
                     void *stream)
{
    if (!dz || !y || !scale || !shift || !mean || !invstd || !sums || !dy || rows <= 0) return PCB_ERR_INVALID_ARG;
    if (bad_c<T>(C)) return PCB_ERR_UNSUPPORTED;
    const int RT = kThreads / (C / RowVec<T>::E);
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<T>, dim3(grid_for(rows, RT * 4, 2048)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)dz, (const uint4 *)y, scale, shift, mean, invstd, sums, rows, C, slope_of(act),
                       use_batch_stats, (uint4 *)dy);
    pcb_account(3.0 * sizeof(T) * rows * C);
    return pcb_check_launch();
}

template <typename T>
int bn_act_max_bwd(const float *dout, const unsigned char *argmax, const void *y, const float *scale,
                   const float *shift, const float *mean, const float *invstd, long groups, int ns, int C, int act,
                   int use_batch_stats, float *sums, void *dy, void *stream)
{
    if (!dy) return PCB_ERR_INVALID_ARG;
    const int st = bn_act_max_bwd_reduce<T>(dout, argmax, y, scale, shift, mean, invstd, groups, ns, C, act, sums, 1, stream);
    if (st != PCB_OK) return st;
    const long rows = groups * ns;
    const long nvec = rows * (C / RowVec<T>::E);
    hipLaunchKernelGGL(bn_max_bwd_apply_kernel<T>, dim3(grid_for(nvec)), dim3(kThreads), 0, (hipStream_t)stream, dout,
                       argmax, (const uint4 *)y, scale, shift, mean, invstd, sums, rows, C, ns, slope_of(act),
                       use_batch_stats, (uint4 *)dy, nvec);
    pcb_account(2.0 * sizeof(T) * rows * C + 5.0 * groups * C);
    return pcb_check_launch();
}

template <typename T>
int group_rows(const float *xyz, const float *new_xyz, const void *feat, const int64_t *idx, int B, int N, int S,
               int ns, int C, int Kp, void *out, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!xyz || !new_xyz || !idx || !out || B <= 0 || N <= 0 || S <= 0 || ns <= 0 || C < 0) return PCB_ERR_INVALID_ARG;
    if ((C > 0 && !feat) || Kp < C + 3 || (Kp % E) != 0) return PCB_ERR_INVALID_ARG;
    const long nchunk = (long)B * S * ns * (Kp / E);
    hipLaunchKernelGGL(group_rows_kernel<T>, dim3(grid_for(nchunk, kThreads, 8192)), dim3(kThreads), 0,
                       (hipStream_t)stream, xyz, new_xyz, (const T *)feat, idx, N, S, ns, C, Kp, (T *)out, nchunk);
    pcb_account(2.0 * sizeof(T) * (double)B * S * ns * Kp + 8.0 * B * S * ns);
    return pcb_check_launch();
}

template <typename T>
int group_rows_bwd(const void *grad_rows, const int64_t *idx, int B, int N, int S, int ns, int C, int Kp,
                   float *grad_feat, void *stream)
{
    if (!grad_rows || !idx || !grad_feat || B <= 0 || N <= 0 || S <= 0 || ns <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    const long total = (long)B * S * ns * C;
    hipLaunchKernelGGL(group_rows_bwd_kernel<T>, dim3(grid_for(total, kThreads, 8192)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const T *)grad_rows, idx, N, S, ns, C, Kp, grad_feat, total);
    pcb_account((sizeof(T) + 4.0) * (double)B * S * ns * C + 8.0 * B * S * ns);
    return pcb_check_launch();
}

template <typename T>
int pad_rows(const float *x, long ld, long R, int k, int kp, void *out, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!x || !out || R <= 0 || k <= 0 || kp < k || (kp % E) || ld < k) return PCB_ERR_INVALID_ARG;
    const long nvec = R * (kp / E);
    hipLaunchKernelGGL(pad_rows_kernel<T>, dim3(grid_for(nvec)), dim3(kThreads), 0, (hipStream_t)stream, x, ld, k, kp,
                       (T *)out, nvec);
    pcb_account(4.0 * R * k + (double)sizeof(T) * R * kp);
    return pcb_check_launch();
}

template <typename T>
int gate(const void *x, const void *a, void *out, long n, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!x || !a || !out || n <= 0 || (n % E)) return PCB_ERR_INVALID_ARG;
    const long nvec = n / E;
    hipLaunchKernelGGL(gate_kernel<T>, dim3(grid_for(nvec, kThreads, 8192)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)x, (const uint4 *)a, (uint4 *)out, nvec);
    pcb_account(3.0 * sizeof(T) * n);
    return pcb_check_launch();
}

template <typename T>
int dropout_rows(const void *x, long n, const long long *seed, float p, void *out, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!x || !out || !seed || n <= 0 || (n % E) || !(p >= 0.0f && p < 1.0f)) return PCB_ERR_INVALID_ARG;
    const unsigned threshold = (unsigned)(p * 256.0f + 0.5f);   // P(drop) = threshold / 256
    if (threshold >= 256) return PCB_ERR_INVALID_ARG;
    const float scale = 256.0f / (float)(256 - threshold);
    const long nvec = n / E;
    hipLaunchKernelGGL(dropout_kernel<T>, dim3(grid_for(nvec, kThreads, 8192)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)x, (uint4 *)out, nvec, seed, threshold, scale);
    pcb_account(2.0 * sizeof(T) * n);
    return pcb_check_launch();
}

template <typename T>
int gate_bwd(const void *g, const void *x, const void *a, void *dx, void *da, long n, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!g || !x || !a || !dx || !da || n <= 0 || (n % E)) return PCB_ERR_INVALID_ARG;
    const long nvec = n / E;
    hipLaunchKernelGGL(gate_bwd_kernel<T>, dim3(grid_for(nvec, kThreads, 8192)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)g, (const uint4 *)x, (const uint4 *)a, (uint4 *)dx, (uint4 *)da, nvec);
    pcb_account(5.0 * sizeof(T) * n);
    return pcb_check_launch();
}

}  // namespace

extern "C" {

int pcb_dropout_rows_bf16(const void *x, long n, const long long *seed, float p, void *out, void *stream)
{
    return dropout_rows<pcb_bf16>(x, n, seed, p, out, stream);
}
int pcb_dropout_rows_f32(const void *x, long n, const long long *seed, float p, void *out, void *stream)
{
    return dropout_rows<float>(x, n, seed, p, out, stream);
}

int pcb_colstats_bf16(const void *y, long rows, int C, float *sums, void *stream) { return colstats<pcb_bf16>(y, rows, C, sums, 0, stream); }
int pcb_colstats_f32(const void *y, long rows, int C, float *sums, void *stream) { return colstats<float>(y, rows, C, sums, 0, stream); }
int pcb_colstats_slabs_bf16(const void *y, long rows, int C, float *slabs, int nparts, void *stream)
{
    return nparts < 1 ? PCB_ERR_INVALID_ARG : colstats<pcb_bf16>(y, rows, C, slabs, nparts, stream);
}
int pcb_colstats_slabs_f32(const void *y, long rows, int C, float *slabs, int nparts, void *stream)
{
    return nparts < 1 ? PCB_ERR_INVALID_ARG : colstats<float>(y, rows, C, slabs, nparts, stream);
}

int pcb_bn_act_bwd_apply_bf16(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                              const float *invstd, const float *sums, long rows, int C, int act, int use_batch_stats,
                              void *dy, void *stream)
{
    return bn_act_bwd_apply<pcb_bf16>(dz, y, scale, shift, mean, invstd, sums, rows, C, act, use_batch_stats, dy, stream);
}
int pcb_bn_act_bwd_apply_f32(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                             const float *invstd, const float *sums, long rows, int C, int act, int use_batch_stats,
                             void *dy, void *stream)
{
    return bn_act_bwd_apply<float>(dz, y, scale, shift, mean, invstd, sums, rows, C, act, use_batch_stats, dy, stream);
}

int pcb_sum_slabs(const float *slabs, int nparts, int n, float *out, void *stream)
{
    if (!slabs || !out || nparts < 1 || n <= 0) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((n + 31) / 32), dim3(1024), 0, (hipStream_t)stream, slabs, nparts, n, out);
    return pcb_check_launch();
}

int pcb_bn_finalize_centred(const float *sums, int nparts, long rows, long count, int C, const float *gamma,
                            const float *beta, const float *bias, float *running_mean, float *running_var, float momentum,
                            float eps, int training, float *scale, float *shift, float *mean, float *invstd,
                            long long *num_batches_tracked, float *centre, int cmode, void *stream)
{
    if (!scale || !shift || !mean || !invstd || C <= 0 || rows <= 0) return PCB_ERR_INVALID_ARG;
    if (training ? (!sums || nparts < 1) : (!running_mean || !running_var)) return PCB_ERR_INVALID_ARG;
    if (cmode < 0 || cmode > 2 || (cmode && !centre) || (cmode == 2 && !training)) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, sums, nparts,
                       rows, count > 0 ? count : rows, C, gamma, beta, bias, running_mean, running_var, momentum, eps, training,
                       scale, shift, mean, invstd, num_batches_tracked, centre, cmode);
    return pcb_check_launch();
}

int pcb_bn_finalize(const float *sums, int nparts, long rows, long count, int C, const float *gamma,
                    const float *beta, const float *bias, float *running_mean, float *running_var, float momentum,
                    float eps, int training, float *scale, float *shift, float *mean, float *invstd,
                    long long *num_batches_tracked, void *stream)
{
    return pcb_bn_finalize_centred(sums, nparts, rows, count, C, gamma, beta, bias, running_mean, running_var, momentum, eps,
                                   training, scale, shift, mean, invstd, num_batches_tracked, nullptr, 0, stream);
}

int pcb_bn_act_bf16(const void *y, const float *scale, const float *shift, long rows, int C, int act, void *z, void *stream)
{
    return bn_act<pcb_bf16>(y, scale, shift, rows, C, act, z, stream);
}
int pcb_bn_act_f32(const void *y, const float *scale, const float *shift, long rows, int C, int act, void *z, void *stream)
{
    return bn_act<float>(y, scale, shift, rows, C, act, z, stream);
}

int pcb_bn_act_max_bf16(const void *y, const float *scale, const float *shift, long groups, int ns, int C, int act,
                        void *out, unsigned char *argmax, void *stream)
{
    return bn_act_max<pcb_bf16>(y, scale, shift, groups, ns, C, act, out, argmax, stream);
}
int pcb_bn_act_max_f32(const void *y, const float *scale, const float *shift, long groups, int ns, int C, int act,
                       void *out, unsigned char *argmax, void *stream)
{
    return bn_act_max<float>(y, scale, shift, groups, ns, C, act, out, argmax, stream);
}

int pcb_bn_act_bwd_bf16(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                        const float *invstd, long rows, int C, int act, int use_batch_stats, float *sums, void *dy,
                        void *stream)
{
    return bn_act_bwd<pcb_bf16>(dz, y, scale, shift, mean, invstd, rows, C, act, use_batch_stats, sums, dy, stream);
}
int pcb_bn_act_bwd_f32(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                       const float *invstd, long rows, int C, int act, int use_batch_stats, float *sums, void *dy,
                       void *stream)
{
    return bn_act_bwd<float>(dz, y, scale, shift, mean, invstd, rows, C, act, use_batch_stats, sums, dy, stream);
}

int pcb_bn_act_max_bwd_bf16(const float *dout, const unsigned char *argmax, const void *y, const float *scale,
                            const float *shift, const float *mean, const float *invstd, long groups, int ns, int C,
                            int act, int use_batch_stats, float *sums, void *dy, void *stream)
{
    return bn_act_max_bwd<pcb_bf16>(dout, argmax, y, scale, shift, mean, invstd, groups, ns, C, act, use_batch_stats, sums, dy, stream);
}
int pcb_bn_act_max_bwd_f32(const float *dout, const unsigned char *argmax, const void *y, const float *scale,
                           const float *shift, const float *mean, const float *invstd, long groups, int ns, int C,
                           int act, int use_batch_stats, float *sums, void *dy, void *stream)
{
    return bn_act_max_bwd<float>(dout, argmax, y, scale, shift, mean, invstd, groups, ns, C, act, use_batch_stats, sums, dy, stream);
}

int pcb_group_rows_bf16(const float *xyz, const float *new_xyz, const void *feat, const int64_t *idx, int B, int N, int S,
                        int ns, int C, int Kp, void *out, void *stream)
{
    return group_rows<pcb_bf16>(xyz, new_xyz, feat, idx, B, N, S, ns, C, Kp, out, stream);
}
int pcb_group_rows_f32(const float *xyz, const float *new_xyz, const void *feat, const int64_t *idx, int B, int N, int S,
                       int ns, int C, int Kp, void *out, void *stream)
{
    return group_rows<float>(xyz, new_xyz, feat, idx, B, N, S, ns, C, Kp, out, stream);
}
int pcb_group_rows_bf16_bwd(const void *grad_rows, const int64_t *idx, int B, int N, int S, int ns, int C, int Kp,
                            float *grad_feat, void *stream)
{
    return group_rows_bwd<pcb_bf16>(grad_rows, idx, B, N, S, ns, C, Kp, grad_feat, stream);
}
int pcb_group_rows_f32_bwd(const void *grad_rows, const int64_t *idx, int B, int N, int S, int ns, int C, int Kp,
                           float *grad_feat, void *stream)
{
    return group_rows_bwd<float>(grad_rows, idx, B, N, S, ns, C, Kp, grad_feat, stream);
}

int pcb_bn_act_bwd_reduce_bf16(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                               const float *invstd, long rows, int C, int act, float *sums, int nparts, void *stream)
{
    return bn_act_bwd_reduce<pcb_bf16>(dz, y, scale, shift, mean, invstd, rows, C, act, sums, nparts, stream);
}
int pcb_bn_act_bwd_reduce_f32(const void *dz, const void *y, const float *scale, const float *shift, const float *mean,
                              const float *invstd, long rows, int C, int act, float *sums, int nparts, void *stream)
{
    return bn_act_bwd_reduce<float>(dz, y, scale, shift, mean, invstd, rows, C, act, sums, nparts, stream);
}

int pcb_bn_act_max_bwd_reduce_bf16(const float *dout, const unsigned char *argmax, const void *y, const float *scale,
                                   const float *shift, const float *mean, const float *invstd, long groups, int ns,
                                   int C, int act, float *sums, int nparts, void *stream)
{
    return bn_act_max_bwd_reduce<pcb_bf16>(dout, argmax, y, scale, shift, mean, invstd, groups, ns, C, act, sums, nparts, stream);
}
int pcb_bn_act_max_bwd_reduce_f32(const float *dout, const unsigned char *argmax, const void *y, const float *scale,
                                  const float *shift, const float *mean, const float *invstd, long groups, int ns,
                                  int C, int act, float *sums, int nparts, void *stream)
{
    return bn_act_max_bwd_reduce<float>(dout, argmax, y, scale, shift, mean, invstd, groups, ns, C, act, sums, nparts, stream);
}

int pcb_pad_rows_bf16(const float *x, long ld, long R, int k, int kp, void *out, void *stream) { return pad_rows<pcb_bf16>(x, ld, R, k, kp, out, stream); }
int pcb_pad_rows_f32(const float *x, long ld, long R, int k, int kp, void *out, void *stream) { return pad_rows<float>(x, ld, R, k, kp, out, stream); }

int pcb_gate_bf16(const void *x, const void *a, void *out, long n, void *stream) { return gate<pcb_bf16>(x, a, out, n, stream); }
int pcb_gate_f32(const void *x, const void *a, void *out, long n, void *stream) { return gate<float>(x, a, out, n, stream); }
int pcb_gate_bwd_bf16(const void *g, const void *x, const void *a, void *dx, void *da, long n, void *stream)
{
    return gate_bwd<pcb_bf16>(g, x, a, dx, da, n, stream);
}
int pcb_gate_bwd_f32(const void *g, const void *x, const void *a, void *dx, void *da, long n, void *stream)
{
    return gate_bwd<float>(g, x, a, dx, da, n, stream);
}

}  // extern "C"
