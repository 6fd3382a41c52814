// kNN graph construction for gfx950 (DGCNN dynamic graph).
//
// Replaces DGCNN.knn, models/DGCNN.py:49-70 of the reference: a [B,N,N] fp32 matrix from a batched
// matmul plus two broadcast adds, then torch.topk over it (2.1 GB per layer at B=8, N=8192).
// Here the matrix never exists.  The inner products run on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32): exact fp32, bit for bit the k-ordered fma chain
//     dot = fma(x_{D-1}, y_{D-1}, ... fma(x_1, y_1, x_0*y_0))
// that the CPU oracle and the reference's K=3 sgemm produce, at the fp32 vector peak rate but with
// both operands in registers -- which leaves the VALU free for the selection.
//
//   workgroup = 4 waves = 128 queries of one scene; wave = 32 queries (MFMA column n = lane & 31)
//   candidates are staged 128 (64 for D > 64) at a time through LDS (shared by the 4 waves), 32 per MFMA tile
//   D[m][n]: lane (n, h) receives 16 of the tile's 32 candidates, m = (i&3) + 8(i>>2) + 4h
//   pd(i,j) = (|xi|^2 + (-2*dot)) + |xj|^2                                   (DGCNN.py:63-65)
//   each lane keeps the K best (distance, index) of ITS half of the candidates sorted in registers.
//   Selection is the expensive part on a SIMD machine: a lane rarely has a new top-K member, but a
//   wave would pay the ~100-instruction sorted insertion whenever ANY of its 64 lanes has one.
//   So lanes only APPEND qualifying candidates (distance below their current k-th best) to a small
//   per-lane queue in LDS -- a predicated store -- and the wave drains all queues together when one
//   runs full: the j-th queued entries of all lanes are inserted in the same pass.
//   lanes n and n+32 merge their lists at the end (ties: lower index first)
//   The candidate stages are double-buffered: the next stage's rows are in flight (registers) while
//   the MFMAs of the current one run, one barrier per stage.  |x_j|^2 comes from a small pre-pass
//   (knn_norms_kernel: one lane per point, in the summation order of the reference's torch.sum(x**2))
//   instead of 64 lanes recomputing it per stage.
#include "pcb_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;

__device__ __forceinline__ int wave_max_i(int v)
{
    v = max(v, dpp_i<PCB_ROW_ROR(8)>(v));
    v = max(v, dpp_i<PCB_ROW_ROR(4)>(v));
    v = max(v, dpp_i<PCB_ROW_ROR(2)>(v));
    v = max(v, dpp_i<PCB_ROW_ROR(1)>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// insertion of (d, cand) into a list sorted ascending by (d, then insertion order)
template <int K>
__device__ __forceinline__ void insert_sorted(float (&bd)[K], int (&bi)[K], float d, int cand)
{
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

// same, ordered by (d, index): used when two lanes' lists are merged
template <int K>
__device__ __forceinline__ void insert_sorted_by_index(float (&bd)[K], int (&bi)[K], float d, int cand)
{
    bool hi = d < bd[K - 1] || (d == bd[K - 1] && cand < bi[K - 1]);
    if (!hi) return;
#pragma unroll
    for (int s = K - 1; s > 0; --s) {
        const bool lo = d < bd[s - 1] || (d == bd[s - 1] && cand < bi[s - 1]);
        bd[s] = lo ? bd[s - 1] : (hi ? d : bd[s]);
        bi[s] = lo ? bi[s - 1] : (hi ? cand : bi[s]);
        hi = lo;
    }
    bd[0] = hi ? d : bd[0];
    bi[0] = hi ? cand : bi[0];
}

// |x|^2 of every point in the order of the reference's torch.sum(x**2, dim=2) (models/DGCNN.py:64), multiply and
// add rounded separately.  ATen's vectorised reduction over a contiguous axis (tools/sgemm_order.py, bit-identical
// for D = 32..256): D % 32 == 0 -> four 8-lane accumulators take the 8-channel chunks in turn, accumulators added
// left to right, then the lanes left to right; other D (D = 3: the coordinates) -> one left-to-right chain.
__global__ __launch_bounds__(kThreads) void knn_norms_kernel(const float *__restrict__ x, long rows, int D,
                                                              float *__restrict__ norms)
{
    const long r = (long)blockIdx.x * kThreads + threadIdx.x;
    if (r >= rows) return;
    const float *__restrict__ p = x + r * D;
    float s = 0.0f;
    if (D % 32 == 0) {
        for (int l = 0; l < 8; ++l) {
            float v = 0.0f;
            for (int u = 0; u < 4; ++u) {
                float a = __fmul_rn(p[u * 8 + l], p[u * 8 + l]);
                for (int c0 = u * 8 + 32; c0 < D; c0 += 32) a = __fadd_rn(a, __fmul_rn(p[c0 + l], p[c0 + l]));
                v = u ? __fadd_rn(v, a) : a;
            }
            s = l ? __fadd_rn(s, v) : v;
        }
    } else {
        for (int c = 0; c < D; ++c) {
            const float v = p[c];
            s = c ? __fadd_rn(s, __fmul_rn(v, v)) : __fmul_rn(v, v);
        }
    }
    norms[r] = s;
}

// DP = D rounded up to a power of two >= 4; channels D..DP-1 are zero padding, which changes no
// rounding step of the chain (fma(0, 0, acc) == acc) nor of the norms (s + 0*0 == s).
template <int DP, int K>
__global__ __launch_bounds__(kThreads, 2) void knn_mfma_kernel(const float *__restrict__ x,
                                                             const float *__restrict__ norms, int N, int D,
                                                             int k, int64_t *__restrict__ out,
                                                             const int *__restrict__ only_if, int only_if_stride)
{
    // optional per-scene switch (csrc/knngrid.hip serves the other scenes): uniform per workgroup
    if (only_if && only_if[(size_t)blockIdx.y * only_if_stride] == 0) return;
    constexpr int S = DP / 2;         // MFMA steps per tile (two channels each)
    constexpr int LD = DP + 4;        // LDS row stride in floats: rows 16 B apart in bank space
    constexpr int kTC = DP <= 32 ? 128 : (DP <= 64 ? 64 : 32);  // candidates per LDS stage (<= 17.4 KB)
    constexpr int QCAP = 20;          // queue slots per lane; drained when a lane has > QCAP - 16
    constexpr int NPT = kTC * DP / kThreads;  // staged elements per thread and stage
    __shared__ __attribute__((aligned(16))) float tiles[2][kTC * LD];  // row m: [h][s] = x[m][2s+h]
    __shared__ __attribute__((aligned(16))) float cnrms[2][kTC];
    __shared__ float qd[QCAP][kThreads];  // [slot][thread]: a lane's slots are a bank-conflict-free column
    __shared__ int qi[QCAP][kThreads];

    const int b = blockIdx.y;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int h = lane >> 5;
    const float *__restrict__ xb = x + (size_t)b * N * D;

    // this lane's query (two lanes per query: h = 0 holds the even channels, h = 1 the odd ones)
    const int qn_idx = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const bool qvalid = qn_idx < N;
    const float *__restrict__ xq = xb + (size_t)(qvalid ? qn_idx : N - 1) * D;
    float q[S];
#pragma unroll
    for (int s = 0; s < S; ++s) q[s] = (2 * s + h) < D ? xq[2 * s + h] : 0.0f;
    const float qnorm = norms[(size_t)b * N + (qvalid ? qn_idx : N - 1)];

    float bd[K];
    int bi[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
        bd[s] = INFINITY;
        bi[s] = 0x7fffffff;
    }
    int qcnt = 0;
    // drain: every lane inserts its j-th queued candidate in the same pass (queue order = index order)
    auto drain = [&]() {
        const int mx = __builtin_amdgcn_readfirstlane(wave_max_i(qcnt));
        for (int j = 0; j < mx; ++j) {
            if (j < qcnt) {
                const float dv = qd[j][t];
                const int ci = qi[j][t];
                if (dv < bd[K - 1]) insert_sorted<K>(bd, bi, dv, ci);
            }
        }
        qcnt = 0;
    };

    // stage = kTC candidate rows, channels de-interleaved into [evens | odds]; a thread's NPT
    // elements are e = t + i*256 (consecutive lanes -> consecutive channels of a row: coalesced)
    float pre[NPT];
    float pre_n = 0.0f;
    auto fetch = [&](int base) {
        if (D == DP && base + kTC <= N) {
            // whole stage inside the cloud, rows DP wide: the stage is one contiguous block and a
            // thread's elements sit at fixed strides from one (wave-uniform + lane) address
            const float *__restrict__ sp = xb + (size_t)base * DP + t;
#pragma unroll
            for (int i = 0; i < NPT; ++i) pre[i] = sp[i * kThreads];
        } else {
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int e = t + i * kThreads;
                const int m = e / DP, c = e % DP;
                const int row = base + m;
                pre[i] = (row < N && c < D) ? xb[(size_t)row * D + c] : 0.0f;
            }
        }
        if (t < kTC) pre_n = base + t < N ? norms[(size_t)b * N + base + t] : 0.0f;
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int e = t + i * kThreads;
            const int m = e / DP, c = e % DP;
            tiles[buf][m * LD + (c & 1) * S + (c >> 1)] = pre[i];
        }
        if (t < kTC) cnrms[buf][t] = pre_n;
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int base = 0; base < N; base += kTC, buf ^= 1) {
        const bool more = base + kTC < N;
        if (more) fetch(base + kTC);  // in flight under this stage's MFMAs
        const float *const tile = tiles[buf];
        const float *const cnrm = cnrms[buf];

#pragma unroll 1
        for (int tt = 0; tt < kTC / 32; ++tt) {
            if (base + tt * 32 >= N) break;  // wave-uniform
            // A operand: lane (m, h) holds x_cand[m][2s + h], s = 0..S-1
            float a[S];
            const float *row = &tile[(tt * 32 + (lane & 31)) * LD + h * S];
#pragma unroll
            for (int s4 = 0; s4 < S / 4; ++s4) {
                const float4 v = *reinterpret_cast<const float4 *>(row + s4 * 4);
                a[s4 * 4 + 0] = v.x;
                a[s4 * 4 + 1] = v.y;
                a[s4 * 4 + 2] = v.z;
                a[s4 * 4 + 3] = v.w;
            }
            if (S < 4) {
#pragma unroll
                for (int s = 0; s < S; ++s) a[s] = row[s];
            }
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], q[s], acc, 0, 0, 0);

            // distances of this lane's 16 candidates; those below the lane's current k-th best are
            // appended to its queue (the threshold is stale until the next drain: a few extra
            // entries, never a missed one)
            const float thr = bd[K - 1];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 cn = *reinterpret_cast<const float4 *>(&cnrm[tt * 32 + 8 * g + 4 * h]);
                const float cnv[4] = {cn.x, cn.y, cn.z, cn.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cand = base + tt * 32 + 8 * g + 4 * h + j;
                    // (xx + inner) + xx^T with inner = -2*dot (exact), DGCNN.py:63-65
                    const float v = __fadd_rn(__fmaf_rn(-2.0f, acc[g * 4 + j], qnorm), cnv[j]);
                    if (cand < N && v < thr) {
                        qd[qcnt][t] = v;
                        qi[qcnt][t] = cand;
                        ++qcnt;
                    }
                }
            }
            if (__any(qcnt > QCAP - 16)) drain();
        }
        if (more) stash(buf ^ 1);  // the other buffer was last read one barrier ago
        __syncthreads();
    }
    drain();

    // lanes n and n+32 saw disjoint halves of the candidates: merge (ties: lower index first)
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const float od = __shfl(bd[s], (lane & 31) + 32);
        const int oi = __shfl(bi[s], (lane & 31) + 32);
        if (h == 0) insert_sorted_by_index<K>(bd, bi, od, oi);
    }
    if (h == 0 && qvalid) {
        int64_t *__restrict__ o = out + ((size_t)b * N + qn_idx) * k;
#pragma unroll
        for (int s = 0; s < K; ++s)
            if (s < k) o[s] = (int64_t)bi[s];
    }
}

template <int DP>
int launch_knn(const float *x, float *norms, int B, int N, int D, int k, int64_t *out, hipStream_t st,
               const int *only_if = nullptr, int stride = 0)
{
    const long rows = (long)B * N;
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((rows + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, x,
                       rows, D, norms);
    const dim3 grid((N + 127) / 128, B);
    if (k <= 8)
        hipLaunchKernelGGL((knn_mfma_kernel<DP, 8>), grid, dim3(kThreads), 0, st, x, norms, N, D, k, out, only_if, stride);
    else if (k <= 20)
        hipLaunchKernelGGL((knn_mfma_kernel<DP, 20>), grid, dim3(kThreads), 0, st, x, norms, N, D, k, out, only_if, stride);
    else
        hipLaunchKernelGGL((knn_mfma_kernel<DP, 32>), grid, dim3(kThreads), 0, st, x, norms, N, D, k, out, only_if, stride);
    if (!only_if) pcb_account(4.0 * (double)D * N * B + 8.0 * (double)N * k * B);  // (the flagged form rides on pcb_knn_xyz's count)
    return pcb_check_launch();
}

}  // namespace

extern "C" int pcb_knn(const float *x, int B, int N, int D, int k, float *norms, int64_t *out_idx, void *stream)
{
    if (!x || !norms || !out_idx || B <= 0 || N <= 0 || D <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 32 || k > N) return PCB_ERR_INVALID_ARG;
    if (D > 128) return PCB_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (D <= 4) return launch_knn<4>(x, norms, B, N, D, k, out_idx, st);
    if (D <= 8) return launch_knn<8>(x, norms, B, N, D, k, out_idx, st);
    if (D <= 16) return launch_knn<16>(x, norms, B, N, D, k, out_idx, st);
    if (D <= 32) return launch_knn<32>(x, norms, B, N, D, k, out_idx, st);
    if (D <= 64) return launch_knn<64>(x, norms, B, N, D, k, out_idx, st);
    return launch_knn<128>(x, norms, B, N, D, k, out_idx, st);
}

// pcb_knn with a per-scene switch: scene b is computed only if only_if[b * stride] != 0 (NULL: all).
int pcb_knn_flagged(const float *x, int B, int N, int D, int k, float *norms, int64_t *out_idx, const int *only_if,
                    int only_if_stride, hipStream_t st)
{
    if (D <= 4) return launch_knn<4>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    if (D <= 8) return launch_knn<8>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    if (D <= 16) return launch_knn<16>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    if (D <= 32) return launch_knn<32>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    if (D <= 64) return launch_knn<64>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
    return launch_knn<128>(x, norms, B, N, D, k, out_idx, st, only_if, only_if_stride);
}
