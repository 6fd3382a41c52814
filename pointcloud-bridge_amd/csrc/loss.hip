// Per-point cross entropy of the segmentation trainers as ONE pass over the logits rows.
//
// Reference: `criterion = nn.CrossEntropyLoss()` on [B,C,N] logits against [B,N] labels
// (train_MulSca_PN2.py:161, train_MulSca_BriStruNet_CB.py; train_DGCNN.py:177-197 on [B*N,C]).  ATen runs
// that as a layout copy, log-softmax, two fills, nll_loss2d and a mean, and as many passes again in
// backward: 14 launches and ~140 us for 1.3 M five-class logits -- the rows of the network's last
// GEMM, 5 MB.  Here a lane owns one point: log-sum-exp over its C logits (C <= 64), loss = lse - x[label],
// per-workgroup partial sums in a fixed order, a one-workgroup finish that forms the mean over the
// points whose label is not ignore_index (torch's default reduction); backward recomputes the softmax
// and writes (softmax - onehot) * g / count as rows.  No atomics, no memset: a captured step replays it.
#include "pcb_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxC = 64;

// logits of row r: `ld` floats apart, C contiguous values.  Two passes over the row's registers/L1 lines
// instead of a local array (C is a runtime value): max, then sum of exp.
__device__ __forceinline__ float row_lse(const float *x, int C, float *mx)
{
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(x[c] - m);
    *mx = m;
    return m + logf(s);
}

__global__ __launch_bounds__(kThreads) void ce_fwd_kernel(const float *__restrict__ logits, long ld,
                                                           const int64_t *__restrict__ labels, long R, int C,
                                                           long ignore_index, float *__restrict__ partials)
{
    __shared__ float red[2][kThreads];
    float loss = 0.0f, cnt = 0.0f;
    for (long r = (long)blockIdx.x * kThreads + threadIdx.x; r < R; r += (long)gridDim.x * kThreads) {
        const int64_t lab = labels[r];
        if (lab == ignore_index || lab < 0 || lab >= C) continue;  // out-of-range labels: skipped like ignored ones
        const float *x = logits + r * ld;
        float m;
        const float lse = row_lse(x, C, &m);
        loss += lse - x[lab];
        cnt += 1.0f;
    }
    red[0][threadIdx.x] = loss;
    red[1][threadIdx.x] = cnt;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = red[0][0];
        partials[gridDim.x + blockIdx.x] = red[1][0];
    }
}

// out[0] = sum(loss) / count (NaN for count == 0, like torch), out[1] = count
__global__ __launch_bounds__(kThreads) void ce_finish_kernel(const float *__restrict__ partials, int nblk,
                                                              float *__restrict__ out)
{
    __shared__ double red[2][kThreads];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kThreads) {
        a += (double)partials[i];
        b += (double)partials[nblk + i];
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(red[0][0] / red[1][0]);
        out[1] = (float)red[1][0];
    }
}

// dlogits[r][c] = (softmax(x_r)[c] - (c == label)) * g / count, 0 for ignored rows.  dlogits rows are C floats
// apart (contiguous [R,C]).
__global__ __launch_bounds__(kThreads) void ce_bwd_kernel(const float *__restrict__ logits, long ld,
                                                           const int64_t *__restrict__ labels, long R, int C,
                                                           long ignore_index, const float *__restrict__ loss_count,
                                                           const float *__restrict__ gout, float *__restrict__ dlogits)
{
    const float scale = gout[0] / loss_count[1];
    for (long r = (long)blockIdx.x * kThreads + threadIdx.x; r < R; r += (long)gridDim.x * kThreads) {
        const int64_t lab = labels[r];
        float *d = dlogits + r * C;
        if (lab == ignore_index || lab < 0 || lab >= C) {
            for (int c = 0; c < C; ++c) d[c] = 0.0f;
            continue;
        }
        const float *x = logits + r * ld;
        float m;
        const float lse = row_lse(x, C, &m);
        for (int c = 0; c < C; ++c) d[c] = (expf(x[c] - lse) - (c == (int)lab ? 1.0f : 0.0f)) * scale;
    }
}

// ---- weighted, label-smoothed cross entropy: F.cross_entropy(x, y, weight=w, label_smoothing=eps) -----------------------
// (the reference's BridgeStructureLoss ends in it, models/model.py:258-260).  Per point, with logp = x - lse:
//   loss_i = (1 - eps) * w[y_i] * (-logp[y_i]) + eps / C * sum_c w[c] * (-logp[c]);   result = sum_i loss_i / sum_i w[y_i]
// partials: [3][nblk] = loss sums | weight sums | point counts.
__global__ __launch_bounds__(kThreads) void cew_fwd_kernel(const float *__restrict__ logits, long ld,
                                                            const int64_t *__restrict__ labels, long R, int C,
                                                            long ignore_index, const float *__restrict__ weight, float eps,
                                                            float *__restrict__ partials)
{
    __shared__ float red[2][kThreads];
    __shared__ float w[kMaxC];
    if ((int)threadIdx.x < C) w[threadIdx.x] = weight[threadIdx.x];
    __syncthreads();
    float wsum = 0.0f;
    for (int c = 0; c < C; ++c) wsum += w[c];
    float loss = 0.0f, wt = 0.0f;
    for (long r = (long)blockIdx.x * kThreads + threadIdx.x; r < R; r += (long)gridDim.x * kThreads) {
        const int64_t lab = labels[r];
        if (lab == ignore_index || lab < 0 || lab >= C) continue;
        const float *x = logits + r * ld;
        float m;
        const float lse = row_lse(x, C, &m);
        float wx = 0.0f;
        for (int c = 0; c < C; ++c) wx = fmaf(w[c], x[c], wx);
        const float smooth = wsum * lse - wx;          // sum_c w[c] (lse - x[c])
        loss += (1.0f - eps) * w[lab] * (lse - x[lab]) + eps / (float)C * smooth;
        wt += w[lab];
    }
    red[0][threadIdx.x] = loss;
    red[1][threadIdx.x] = wt;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = red[0][0];
        partials[gridDim.x + blockIdx.x] = red[1][0];
    }
}

__global__ __launch_bounds__(kThreads) void cew_bwd_kernel(const float *__restrict__ logits, long ld,
                                                            const int64_t *__restrict__ labels, long R, int C,
                                                            long ignore_index, const float *__restrict__ weight, float eps,
                                                            const float *__restrict__ loss_wsum,
                                                            const float *__restrict__ gout, float *__restrict__ dlogits)
{
    __shared__ float w[kMaxC];
    if ((int)threadIdx.x < C) w[threadIdx.x] = weight[threadIdx.x];
    __syncthreads();
    float wsum = 0.0f;
    for (int c = 0; c < C; ++c) wsum += w[c];
    const float scale = gout[0] / loss_wsum[1];
    for (long r = (long)blockIdx.x * kThreads + threadIdx.x; r < R; r += (long)gridDim.x * kThreads) {
        const int64_t lab = labels[r];
        float *d = dlogits + r * C;
        if (lab == ignore_index || lab < 0 || lab >= C) {
            for (int c = 0; c < C; ++c) d[c] = 0.0f;
            continue;
        }
        const float *x = logits + r * ld;
        float m;
        const float lse = row_lse(x, C, &m);
        const float wl = (1.0f - eps) * w[lab];
        for (int c = 0; c < C; ++c) {
            const float p = expf(x[c] - lse);
            d[c] = (wl * (p - (c == (int)lab ? 1.0f : 0.0f)) + eps / (float)C * (p * wsum - w[c])) * scale;
        }
    }
}

// ---- class weights of the reference's BridgeStructureLoss (models/model.py:169-257) ----------------------------------------
// Five classes.  Per scene, from the PREDICTED labels (arg-max of the logits, first maximum) and the z coordinates: for the
// component classes c = 1..4 the mean relative height of the points predicted c -- z of the masked cloud (unselected points
// count as the origin) normalised by that cloud's extent, averaged over the selected points (:189-196) -- and the number
// of points predicted background; per scene the label histogram (:253, and `present`, :208-211).  One workgroup per scene,
// fixed-order LDS trees.  stats[b] = { cnt_pred[5], lo[4], hi[4], zsum[4], label_cnt[5] } (22 floats).
constexpr int kBS = 22;
__global__ __launch_bounds__(1024) void bridge_stats_kernel(const float *__restrict__ logits, long ld,
                                                             const int64_t *__restrict__ labels,
                                                             const float *__restrict__ points, int N,
                                                             float *__restrict__ stats)
{
    __shared__ float red[32][kBS];
    const int b = blockIdx.x;
    float v[kBS];
#pragma unroll
    for (int i = 0; i < kBS; ++i) v[i] = 0.0f;
    // lo / hi start from the first point's contribution: every point contributes z*m (0 when unselected), so the identity
    // elements are +inf / -inf
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        v[5 + c] = INFINITY;
        v[9 + c] = -INFINITY;
    }
    for (int n = threadIdx.x; n < N; n += 1024) {
        const long r = (long)b * N + n;
        const float *x = logits + r * ld;
        int pred = 0;
        float best = x[0];
#pragma unroll
        for (int c = 1; c < 5; ++c)
            if (x[c] > best) {
                best = x[c];
                pred = c;
            }
        const float z = points[r * 3 + 2];
        v[pred] += 1.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float zm = pred == c + 1 ? z : 0.0f;
            v[5 + c] = fminf(v[5 + c], zm);
            v[9 + c] = fmaxf(v[9 + c], zm);
            v[13 + c] += zm;
        }
        const int64_t lab = labels[r];
        if (lab >= 0 && lab < 5) v[17 + lab] += 1.0f;
    }
    // wave-level tree (fixed order), then the 16 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kBS; ++i) {
        float a = v[i];
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(a, off);
            a = (i >= 5 && i < 9) ? fminf(a, o) : ((i >= 9 && i < 13) ? fmaxf(a, o) : a + o);
        }
        if (lane == 0) red[wave][i] = a;
    }
    __syncthreads();
    if (threadIdx.x < kBS) {
        const int i = threadIdx.x;
        float a = red[0][i];
        for (int w = 1; w < 16; ++w) {
            const float o = red[w][i];
            a = (i >= 5 && i < 9) ? fminf(a, o) : ((i >= 9 && i < 13) ? fmaxf(a, o) : a + o);
        }
        stats[(long)b * kBS + i] = a;
    }
}

// w[5] = mean over the scenes of the per-scene weights (base + alpha * order violations, :218-251) times the class weights
// 1/sqrt(label frequency) * {1,2,1,1,2} (:253-256).  One workgroup; thread b owns scene b (B <= 1024).
__global__ __launch_bounds__(1024) void bridge_weights_kernel(const float *__restrict__ stats, int B, int N, float alpha,
                                                               float margin, const float *__restrict__ base,
                                                               float *__restrict__ out)
{
    __shared__ float freq[5];
    __shared__ float cols[32][5];
    if (threadIdx.x < 5) {
        float f = 0.0f;
        for (int b = 0; b < B; ++b) f += stats[(long)b * kBS + 17 + threadIdx.x];
        freq[threadIdx.x] = f;
    }
    __syncthreads();
    float w[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int b = threadIdx.x; b < B; b += 1024) {
        const float *s = stats + (long)b * kBS;
        float h[5], c[5];
        h[0] = 0.0f;
#pragma unroll
        for (int k = 1; k < 5; ++k) {
            const float cnt = s[k], lo = s[5 + k - 1], hi = s[9 + k - 1], zs = s[13 + k - 1];
            h[k] = ((zs - lo * cnt) / (hi - lo + 1e-7f)) / fmaxf(cnt, 1.0f);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) c[k] = base[k];
        // (class, classes that must lie below it, classes that must lie above it): 1 < 2 < 3 < 4 in height
#pragma unroll
        for (int cid = 1; cid < 5; ++cid) {
#pragma unroll
            for (int low = 1; low < 5; ++low) {
                if (low >= cid) continue;
                const float v = fmaxf(margin - (h[cid] - h[low]), 0.0f) * (freq[low] > 0.0f ? 1.0f : 0.0f);
                c[cid] += alpha * v;
                c[low] += alpha * v * 0.5f;
            }
#pragma unroll
            for (int up = 1; up < 5; ++up) {
                if (up <= cid) continue;
                const float v = fmaxf(margin - (h[up] - h[cid]), 0.0f) * (freq[up] > 0.0f ? 1.0f : 0.0f);
                c[cid] += alpha * v;
                c[up] += alpha * v * 0.3f;
            }
        }
        c[0] += alpha * (1.0f - s[0] / (float)N);
#pragma unroll
        for (int k = 0; k < 5; ++k) w[k] += c[k];
    }
    // scenes are few (B <= 1024: one per thread at most): combine in thread order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float a = w[k];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0) cols[wave][k] = a;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        float a = 0.0f;
        for (int wv = 0; wv < 16; ++wv) a += cols[wv][threadIdx.x];
        const float mult[5] = {1.0f, 2.0f, 1.0f, 1.0f, 2.0f};
        out[threadIdx.x] = a / (float)B * (1.0f / sqrtf(fmaxf(freq[threadIdx.x], 1.0f))) * mult[threadIdx.x];
    }
}

inline int blocks_for(long R)
{
    long b = (R + kThreads - 1) / kThreads;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" int pcb_cross_entropy_partials(long R)
{
    return R > 0 ? blocks_for(R) : 0;
}

extern "C" int pcb_cross_entropy_fwd(const float *logits, long ld, const int64_t *labels, long R, int C,
                                     long ignore_index, float *partials, float *loss_count, void *stream)
{
    if (!logits || !labels || !partials || !loss_count || R <= 0 || ld < C) return PCB_ERR_INVALID_ARG;
    if (C < 1 || C > kMaxC) return PCB_ERR_UNSUPPORTED;
    const int nblk = blocks_for(R);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(nblk), dim3(kThreads), 0, st, logits, ld, labels, R, C, ignore_index, partials);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(kThreads), 0, st, (const float *)partials, nblk, loss_count);
    pcb_account(4.0 * R * C + 8.0 * R);
    return pcb_check_launch();
}

extern "C" int pcb_cross_entropy_bwd(const float *logits, long ld, const int64_t *labels, long R, int C,
                                     long ignore_index, const float *loss_count, const float *grad_out,
                                     float *dlogits, void *stream)
{
    if (!logits || !labels || !loss_count || !grad_out || !dlogits || R <= 0 || ld < C) return PCB_ERR_INVALID_ARG;
    if (C < 1 || C > kMaxC) return PCB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(blocks_for(R)), dim3(kThreads), 0, (hipStream_t)stream, logits, ld, labels, R, C,
                       ignore_index, loss_count, grad_out, dlogits);
    pcb_account(8.0 * R * C + 8.0 * R);
    return pcb_check_launch();
}

extern "C" int pcb_cross_entropy_w_fwd(const float *logits, long ld, const int64_t *labels, long R, int C,
                                       long ignore_index, const float *weight, float smoothing, float *partials,
                                       float *loss_wsum, void *stream)
{
    if (!logits || !labels || !weight || !partials || !loss_wsum || R <= 0 || ld < C) return PCB_ERR_INVALID_ARG;
    if (C < 1 || C > kMaxC) return PCB_ERR_UNSUPPORTED;
    const int nblk = blocks_for(R);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cew_fwd_kernel, dim3(nblk), dim3(kThreads), 0, st, logits, ld, labels, R, C, ignore_index, weight,
                       smoothing, partials);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(kThreads), 0, st, (const float *)partials, nblk, loss_wsum);
    pcb_account(4.0 * R * C + 8.0 * R);
    return pcb_check_launch();
}

extern "C" int pcb_cross_entropy_w_bwd(const float *logits, long ld, const int64_t *labels, long R, int C,
                                       long ignore_index, const float *weight, float smoothing, const float *loss_wsum,
                                       const float *grad_out, float *dlogits, void *stream)
{
    if (!logits || !labels || !weight || !loss_wsum || !grad_out || !dlogits || R <= 0 || ld < C) return PCB_ERR_INVALID_ARG;
    if (C < 1 || C > kMaxC) return PCB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cew_bwd_kernel, dim3(blocks_for(R)), dim3(kThreads), 0, (hipStream_t)stream, logits, ld, labels, R, C,
                       ignore_index, weight, smoothing, loss_wsum, grad_out, dlogits);
    pcb_account(8.0 * R * C + 8.0 * R);
    return pcb_check_launch();
}

extern "C" int pcb_bridge_loss_weights(const float *logits, long ld, const int64_t *labels, const float *points, int B,
                                       int N, float alpha, float rel_margin, const float *base_weights, float *stats,
                                       float *weights, void *stream)
{
    if (!logits || !labels || !points || !base_weights || !stats || !weights || B <= 0 || B > 1024 || N <= 0 || ld < 5)
        return PCB_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bridge_stats_kernel, dim3(B), dim3(1024), 0, st, logits, ld, labels, points, N, stats);
    hipLaunchKernelGGL(bridge_weights_kernel, dim3(1), dim3(1024), 0, st, (const float *)stats, B, N, alpha, rel_margin,
                       base_weights, weights);
    pcb_account((4.0 * 5 + 12.0 + 8.0) * B * N);
    return pcb_check_launch();
}
