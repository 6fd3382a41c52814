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
