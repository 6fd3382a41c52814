// kNN graph construction for gfx950 (DGCNN dynamic graph).
//
// Replaces DGCNN.knn, models/DGCNN.py:49-70 of the reference: a [B,N,N] fp32 matrix from a batched
// matmul plus two broadcast adds, then torch.topk over it (2.1 GB per layer at B=8, N=8192).
// Here the matrix never exists: one lane owns one query point (its D features in VGPRs), candidate
// rows are staged through LDS in tiles and read back as wave-wide broadcasts, and each lane keeps
// its K best (distance, index) pairs sorted in registers.  The insertion is an unrolled shift that
// a wave only enters when one of its lanes beats its current k-th distance.
//   pd(i,j) = (|xi|^2 + (-2*<xi,xj>)) + |xj|^2      (DGCNN.py:63-65)
// with <,> an fma chain in channel order; smaller pd first, ties by lower index.
#include "pcb_common.h"

namespace {

constexpr int kThreads = 256;

template <int DMAX>
struct TileCfg {
    static constexpr int TJ = DMAX <= 4 ? 512 : DMAX <= 16 ? 256 : DMAX <= 32 ? 128 : DMAX <= 64 ? 64 : 32;
    static constexpr int DP = DMAX <= 4 ? 4 : DMAX + 4;  // row stride in floats (16-byte aligned rows)
};

// D <= DMAX; channels D..DMAX-1 are zero padding, which changes no rounding step:
// fma(0, 0, dot) == dot and s + 0*0 == s.
template <int DMAX, int K>
__global__ __launch_bounds__(kThreads) void knn_kernel(const float *__restrict__ x, int N, int D,
                                                        int k, int64_t *__restrict__ out)
{
    constexpr int TJ = TileCfg<DMAX>::TJ;
    constexpr int DP = TileCfg<DMAX>::DP;
    __shared__ __attribute__((aligned(16))) float tile[TJ * DP];
    __shared__ float nrm[TJ];

    const int b = blockIdx.y;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    const bool valid = n < N;
    const float *__restrict__ xb = x + (size_t)b * N * D;

    float q[DMAX];
    {
        const float *__restrict__ xr = xb + (size_t)(valid ? n : N - 1) * D;
#pragma unroll
        for (int c = 0; c < DMAX; ++c) q[c] = c < D ? xr[c] : 0.0f;
    }
    float qn = __fmul_rn(q[0], q[0]);
#pragma unroll
    for (int c = 1; c < DMAX; ++c) qn = __fadd_rn(qn, __fmul_rn(q[c], q[c]));

    float bd[K];
    int bi[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
        bd[s] = INFINITY;
        bi[s] = 0;
    }

    for (int base = 0; base < N; base += TJ) {
        const int cnt = min(TJ, N - base);
        __syncthreads();
        for (int e = threadIdx.x; e < TJ * DMAX; e += kThreads) {
            const int j = e / DMAX, c = e % DMAX;
            const int row = base + j;
            tile[j * DP + c] = (row < N && c < D) ? xb[(size_t)row * D + c] : 0.0f;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < TJ; j += kThreads) {
            const float *tj = tile + j * DP;
            float s = __fmul_rn(tj[0], tj[0]);
            for (int c = 1; c < DMAX; ++c) s = __fadd_rn(s, __fmul_rn(tj[c], tj[c]));
            nrm[j] = s;
        }
        __syncthreads();

        for (int j = 0; j < cnt; ++j) {
            const float *tj = tile + j * DP;
            float dot = __fmul_rn(q[0], tj[0]);
#pragma unroll
            for (int c = 1; c < DMAX; ++c) dot = __fmaf_rn(q[c], tj[c], dot);
            // (xx + inner) + xx^T with inner = -2*dot (exact), DGCNN.py:63-65
            const float d = __fadd_rn(__fmaf_rn(-2.0f, dot, qn), nrm[j]);
            if (d < bd[K - 1]) {
                const int cand = base + j;
                bool hi = true;  // d < bd[s] for the slot being written
#pragma unroll
                for (int s = K - 1; s > 0; --s) {
                    const bool lo = d < bd[s - 1];
                    bd[s] = lo ? bd[s - 1] : (hi ? d : bd[s]);
                    bi[s] = lo ? bi[s - 1] : (hi ? cand : bi[s]);
                    hi = lo;
                }
                bd[0] = hi ? d : bd[0];
                bi[0] = hi ? cand : bi[0];
            }
        }
    }

    if (valid) {
        int64_t *__restrict__ o = out + ((size_t)b * N + n) * k;
#pragma unroll
        for (int s = 0; s < K; ++s)
            if (s < k) o[s] = (int64_t)bi[s];
    }
}

template <int DMAX>
int launch_knn(const float *x, int B, int N, int D, int k, int64_t *out, hipStream_t st)
{
    const dim3 grid((N + kThreads - 1) / kThreads, B);
    if (k <= 8)
        hipLaunchKernelGGL((knn_kernel<DMAX, 8>), grid, dim3(kThreads), 0, st, x, N, D, k, out);
    else if (k <= 20)
        hipLaunchKernelGGL((knn_kernel<DMAX, 20>), grid, dim3(kThreads), 0, st, x, N, D, k, out);
    else
        hipLaunchKernelGGL((knn_kernel<DMAX, 32>), grid, dim3(kThreads), 0, st, x, N, D, k, out);
    return pcb_check_launch();
}

}  // namespace

extern "C" int pcb_knn(const float *x, int B, int N, int D, int k, int64_t *out_idx, void *stream)
{
    if (!x || !out_idx || B <= 0 || N <= 0 || D <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 32 || k > N) return PCB_ERR_INVALID_ARG;
    if (D > 128) return PCB_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (D <= 4) return launch_knn<4>(x, B, N, D, k, out_idx, st);
    if (D <= 16) return launch_knn<16>(x, B, N, D, k, out_idx, st);
    if (D <= 32) return launch_knn<32>(x, B, N, D, k, out_idx, st);
    if (D <= 64) return launch_knn<64>(x, B, N, D, k, out_idx, st);
    return launch_knn<128>(x, B, N, D, k, out_idx, st);
}
