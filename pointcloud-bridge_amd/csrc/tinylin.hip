// Narrow pointwise (1x1) convolutions on fp32 rows for gfx950: y = x W^T + b with Ci, Co <= 64.
//
// The colour / fusion / structure encoders of the reference's BridgeSeg network,
// models/attention_modules.py:548-553 (the per-point block of structure_mlp[0]), :696-716, :759-764,
// are Conv1d/Conv2d 1x1 layers with 3..40 channels on B*N = 262144 rows.  As GEMMs they have N = 3..16
// columns: far below one MFMA tile, and the library kernels picked for them run at 0.2-0.6 ms each.
// They are plain HBM streams (4*(Ci+Co) bytes per row): a block stages 256 rows in LDS with
// coalesced reads, one lane owns one row, weights are LDS broadcasts, results leave through LDS
// again so that the stores are coalesced too.  The weight gradient is a per-block partial sum
// (slabs added by the caller), no atomics.
#include "pcb_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxC = 64;
constexpr int kMaxBlocks = 2048;
constexpr size_t kDefaultLds = 48 * 1024;  // dynamic LDS a launch may ask for without opting in

__host__ __device__ __forceinline__ int pad_odd(int n) { return n | 1; }  // odd row stride: conflict-free columns

// y[r][o] = b[o] + sum_i x[r][i] * W[o][i]        (TRANS = 0: W is [Co][Ci], the forward)
// y[r][o] =        sum_i x[r][i] * W[i][o]        (TRANS = 1: W is [Ci][Co] -> input gradient dx = dy W)
template <int TRANS>
__global__ __launch_bounds__(kThreads) void rows_linear_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ w,
                                                               const float *__restrict__ b, long P, int Ci, int Co,
                                                               float *__restrict__ y)
{
    extern __shared__ float lds[];
    const int sx = pad_odd(Ci), sy = pad_odd(Co);
    float *xs = lds;                      // [256][sx]
    float *ys = xs + kThreads * sx;       // [256][sy]
    float *ws = ys + kThreads * sy;       // [Co][Ci] as used: ws[o*Ci + i]
    float *bs = ws + Ci * Co;             // [Co]
    for (int e = threadIdx.x; e < Ci * Co; e += kThreads) {
        const int o = e / Ci, i = e - o * Ci;
        ws[e] = TRANS ? w[i * Co + o] : w[e];
    }
    for (int e = threadIdx.x; e < Co; e += kThreads) bs[e] = b ? b[e] : 0.0f;
    for (long r0 = (long)blockIdx.x * kThreads; r0 < P; r0 += (long)gridDim.x * kThreads) {
        const int rows = (int)(P - r0 < kThreads ? P - r0 : kThreads);
        __syncthreads();  // previous tile's ys fully stored; weights visible on the first trip
        for (int e = threadIdx.x; e < rows * Ci; e += kThreads) xs[(e / Ci) * sx + e % Ci] = x[r0 * Ci + e];
        __syncthreads();
        if ((int)threadIdx.x < rows) {
            const float *xr = xs + threadIdx.x * sx;
            for (int o = 0; o < Co; ++o) {
                float acc = bs[o];
                const float *wo = ws + o * Ci;
                for (int i = 0; i < Ci; ++i) acc = fmaf(xr[i], wo[i], acc);
                ys[threadIdx.x * sy + o] = acc;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < rows * Co; e += kThreads) y[r0 * Co + e] = ys[(e / Co) * sy + e % Co];
    }
}

// part[block] = { dW[o][i] = sum over the block's rows of dy[r][o] * x[r][i]  (Co*Ci floats) | db[o] = sum dy[r][o] (Co floats) }.
// Every block writes its whole slab.
__global__ __launch_bounds__(kThreads) void rows_linear_wgrad_kernel(const float *__restrict__ dy,
                                                                     const float *__restrict__ x, long P, int Ci,
                                                                     int Co, float *__restrict__ part)
{
    extern __shared__ float lds[];
    const int sx = pad_odd(Ci + 1), sy = pad_odd(Co);
    float *xs = lds;                 // [256][sx], column Ci = 1.0 (bias gradient rides along)
    float *ds = xs + kThreads * sx;  // [256][sy]
    const int entries = Co * (Ci + 1);
    // entry e = o*(Ci+1) + i is owned by thread e % 256, slot e / 256  (<= 17 slots for 64 x 65)
    float acc[(kMaxC * (kMaxC + 1) + kThreads - 1) / kThreads];
#pragma unroll
    for (int s = 0; s < (int)(sizeof(acc) / sizeof(float)); ++s) acc[s] = 0.0f;
    for (long r0 = (long)blockIdx.x * kThreads; r0 < P; r0 += (long)gridDim.x * kThreads) {
        const int rows = (int)(P - r0 < kThreads ? P - r0 : kThreads);
        __syncthreads();
        for (int e = threadIdx.x; e < rows * Ci; e += kThreads) xs[(e / Ci) * sx + e % Ci] = x[r0 * Ci + e];
        for (int e = threadIdx.x; e < rows; e += kThreads) xs[e * sx + Ci] = 1.0f;
        for (int e = threadIdx.x; e < rows * Co; e += kThreads) ds[(e / Co) * sy + e % Co] = dy[r0 * Co + e];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < (int)(sizeof(acc) / sizeof(float)); ++s) {
            const int e = s * kThreads + threadIdx.x;
            if (e < entries) {
                const int o = e / (Ci + 1), i = e - o * (Ci + 1);
                float a = acc[s];
                for (int r = 0; r < rows; ++r) a = fmaf(ds[r * sy + o], xs[r * sx + i], a);
                acc[s] = a;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < (int)(sizeof(acc) / sizeof(float)); ++s) {
        const int e = s * kThreads + threadIdx.x;
        if (e < entries) {
            // slab layout [Co*Ci weight entries | Co bias entries]: both gradients are contiguous blocks of the summed slab
            const int o = e / (Ci + 1), i = e - o * (Ci + 1);
            part[(long)blockIdx.x * entries + (i < Ci ? o * Ci + i : Co * Ci + o)] = acc[s];
        }
    }
}

int blocks_for(long P)
{
    const long b = (P + kThreads - 1) / kThreads;
    return (int)(b < kMaxBlocks ? b : kMaxBlocks);
}

// the weight-gradient pass walks all 256 rows of a tile per entry: fewer, longer blocks
int wgrad_blocks_for(long P)
{
    const long b = (P + kThreads - 1) / kThreads;
    return (int)(b < 512 ? b : 512);
}

bool bad(long P, int Ci, int Co) { return P <= 0 || Ci < 1 || Co < 1 || Ci > kMaxC || Co > kMaxC; }

size_t fwd_lds(int Ci, int Co)
{
    return sizeof(float) * ((size_t)kThreads * (pad_odd(Ci) + pad_odd(Co)) + (size_t)Ci * Co + Co);
}

}  // namespace

extern "C" int pcb_rows_linear_f32(const float *x, const float *w, const float *bias, long P, int Ci, int Co,
                                   float *y, void *stream)
{
    if (!x || !w || !y) return PCB_ERR_INVALID_ARG;
    if (bad(P, Ci, Co)) return PCB_ERR_UNSUPPORTED;
    const size_t lds = fwd_lds(Ci, Co);
    if (lds > kDefaultLds &&
        hipFuncSetAttribute((const void *)rows_linear_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(rows_linear_kernel<0>, dim3(blocks_for(P)), dim3(kThreads), lds, (hipStream_t)stream, x, w,
                       bias, P, Ci, Co, y);
    return pcb_check_launch();
}

extern "C" int pcb_rows_linear_dgrad_f32(const float *dy, const float *w, long P, int Ci, int Co, float *dx,
                                         void *stream)
{
    if (!dy || !w || !dx) return PCB_ERR_INVALID_ARG;
    if (bad(P, Ci, Co)) return PCB_ERR_UNSUPPORTED;
    // dx [P,Ci] = dy [P,Co] . W [Co,Ci]: the same kernel with the roles of Ci / Co exchanged
    const size_t lds = fwd_lds(Co, Ci);
    if (lds > kDefaultLds &&
        hipFuncSetAttribute((const void *)rows_linear_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(rows_linear_kernel<1>, dim3(blocks_for(P)), dim3(kThreads), lds, (hipStream_t)stream, dy, w,
                       (const float *)nullptr, P, Co, Ci, dx);
    return pcb_check_launch();
}

extern "C" int pcb_rows_linear_wgrad_partials(long P) { return P > 0 ? wgrad_blocks_for(P) : 0; }

extern "C" int pcb_rows_linear_wgrad_f32(const float *dy, const float *x, long P, int Ci, int Co, float *partials,
                                         void *stream)
{
    if (!dy || !x || !partials) return PCB_ERR_INVALID_ARG;
    if (bad(P, Ci, Co)) return PCB_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * (size_t)kThreads * (pad_odd(Ci + 1) + pad_odd(Co));
    if (lds > kDefaultLds &&
        hipFuncSetAttribute((const void *)rows_linear_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PCB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(rows_linear_wgrad_kernel, dim3(wgrad_blocks_for(P)), dim3(kThreads), lds, (hipStream_t)stream,
                       dy, x, P, Ci, Co, partials);
    return pcb_check_launch();
}
