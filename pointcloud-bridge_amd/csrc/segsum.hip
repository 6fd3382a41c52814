// Deterministic scatter-adds: the backward passes of the row gathers as SEGMENT SUMS over an inverted index.
//
// The reference's gathers (index_points models/pointnet2_utils.py:17-39, the grouping of sample_and_group /
// MultiScaleSetAbstraction :51-58 / :342-349, DGCNN.get_graph_feature models/DGCNN.py:90-107) differentiate into
// index_put_(accumulate=True).  The throughput path adds with fp32 atomics at memory (gather.hip, gatherlin.hip,
// rowbn.hip): as fast as the chip adds floats, but the order of the additions changes from run to run, and one changed
// rounding moves a ReLU mask downstream -- two runs of one training command drift apart (+-2.5 mIoU points after 120
// steps, tools/miou_flake.py).  The reproducible mode (ops.set_deterministic / PCB_DETERMINISTIC=1 /
// torch.use_deterministic_algorithms) takes these kernels instead:
//
//   order[E], offsets[T+1]   the inverted index of a gather: for every target row t the source rows that read it,
//                            in ascending source order (a STABLE sort of the targets, built by the host side with
//                            rocPRIM's radix sort through torch.sort: ops.det_index);
//   segment_sum              out[t,:] (+)= sum over e in [offsets[t], offsets[t+1]) of rows[order[e], col0 : col0+C]
//                            one wave per target, lanes over channels, entries in index order: a fixed summation order;
//   scatter_dy_csr           the same with rows = the BatchNorm/activation backward dy of a gathered first layer, rebuilt
//                            on the fly from (dz | dout + argmax, y) exactly as scatter_dy_kernel does.
// Costs more than the atomics (the index, a second pass over (dz, y) for the gathered layers): not the bench default.
#include "rowvec.h"

namespace {

constexpr int kThreads = 256;

// one wave per target row; lane l owns channels c0 + l of the current 64-channel block
template <typename T>
__global__ __launch_bounds__(kThreads) void segment_sum_kernel(const T *__restrict__ rows, long ld, int col0, int C,
                                                                const int *__restrict__ order,
                                                                const long long *__restrict__ offsets, long targets,
                                                                float *__restrict__ out, long out_ld, int accumulate)
{
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (t >= targets) return;  // wave-uniform
    const long beg = offsets[t], end = offsets[t + 1];
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const int cs = c < C ? c : C - 1;
        float acc = accumulate ? out[t * out_ld + cs] : 0.0f;
        // four entries' index and row loads in flight per step; the additions keep the entry order
        for (long e0 = beg; e0 < end; e0 += 4) {
            int r[4];
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = order[e0 + u < end ? e0 + u : beg];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = RowVec<T>::one(rows + (long)r[u] * ld + col0 + cs);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u < end) acc += v[u];
        }
        if (c < C) out[t * out_ld + c] = acc;
    }
}

template <typename T>
int segment_sum(const void *rows, long ld, int col0, int C, const int *order, const long long *offsets, long targets,
                float *out, long out_ld, int accumulate, void *stream)
{
    if (!rows || !order || !offsets || !out || targets <= 0 || C <= 0 || col0 < 0 || col0 + C > ld || out_ld < C)
        return PCB_ERR_INVALID_ARG;
    const long blocks = (targets + kThreads / 64 - 1) / (kThreads / 64);
    hipLaunchKernelGGL(segment_sum_kernel<T>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, (const T *)rows,
                       ld, col0, C, order, offsets, targets, out, out_ld, accumulate);
    return pcb_check_launch();
}

inline float slope_of(int act) { return act == 1 ? 0.0f : (act == 2 ? 0.2f : 1.0f); }

// du[t,:] = sum over the grouped rows r that gathered source point t of dy[r,:], dy as in gatherlin.hip
// (scale * dz * act'(y*scale + shift) + p*y + q; POOLED: dz = dout[g] where argmax[g] == j).  order[] holds global
// grouped-row numbers r = g * ns + j.
template <int POOLED>
__global__ __launch_bounds__(kThreads) void scatter_dy_csr_kernel(
    const pcb_bf16 *__restrict__ dz, const pcb_bf16 *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ p, const float *__restrict__ q,
    const float *__restrict__ dout, const unsigned char *__restrict__ arg, float slope, int ns, int C,
    const int *__restrict__ order, const long long *__restrict__ offsets, long targets, float *__restrict__ du)
{
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (t >= targets) return;
    const long beg = offsets[t], end = offsets[t + 1];
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const int cs = c < C ? c : C - 1;
        const float sc = scale[cs], sh = shift[cs], pp = p[cs], qq = q[cs];
        float acc = 0.0f;
        for (long e = beg; e < end; ++e) {
            const long r = order[e];
            const float yv = pcb_bf2f(y[r * C + cs]);
            float f;
            if (POOLED) {
                const long g = r / ns;
                const int j = (int)(r - g * ns);
                f = (int)arg[g * C + cs] == j ? dout[g * C + cs] : 0.0f;
            } else {
                f = pcb_bf2f(dz[r * C + cs]);
            }
            const float g1 = f * (fmaf(yv, sc, sh) > 0.0f ? 1.0f : slope);
            acc += fmaf(sc, g1, fmaf(pp, yv, qq));
        }
        if (c < C) du[t * C + c] = acc;
    }
}

}  // namespace

extern "C" int pcb_segment_sum_f32(const float *rows, long ld, int col0, int C, const int *order,
                                   const long long *offsets, long targets, float *out, long out_ld, int accumulate,
                                   void *stream)
{
    return segment_sum<float>(rows, ld, col0, C, order, offsets, targets, out, out_ld, accumulate, stream);
}

extern "C" int pcb_segment_sum_bf16(const void *rows, long ld, int col0, int C, const int *order,
                                    const long long *offsets, long targets, float *out, long out_ld, int accumulate,
                                    void *stream)
{
    return segment_sum<pcb_bf16>(rows, ld, col0, C, order, offsets, targets, out, out_ld, accumulate, stream);
}

extern "C" int pcb_scatter_dy_csr_bf16(int pooled, const void *dz, const void *y, const float *scale, const float *shift,
                                       const float *p, const float *q, const float *dout, const unsigned char *argmax,
                                       int act, int ns, int C, const int *order, const long long *offsets, long targets,
                                       float *du, void *stream)
{
    if (!y || !scale || !shift || !p || !q || !order || !offsets || !du || targets <= 0 || ns <= 0 || C <= 0 || (C & 7))
        return PCB_ERR_INVALID_ARG;
    if (pooled ? (!dout || !argmax || ns > 255) : !dz) return PCB_ERR_INVALID_ARG;
    const long blocks = (targets + kThreads / 64 - 1) / (kThreads / 64);
    hipStream_t st = (hipStream_t)stream;
    if (pooled)
        hipLaunchKernelGGL(scatter_dy_csr_kernel<1>, dim3((unsigned)blocks), dim3(kThreads), 0, st, (const pcb_bf16 *)dz,
                           (const pcb_bf16 *)y, scale, shift, p, q, dout, argmax, slope_of(act), ns, C, order, offsets, targets, du);
    else
        hipLaunchKernelGGL(scatter_dy_csr_kernel<0>, dim3((unsigned)blocks), dim3(kThreads), 0, st, (const pcb_bf16 *)dz,
                           (const pcb_bf16 *)y, scale, shift, p, q, dout, argmax, slope_of(act), ns, C, order, offsets, targets, du);
    return pcb_check_launch();
}
