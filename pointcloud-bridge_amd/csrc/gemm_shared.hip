// Pieces shared by the bf16 and the fp32 row GEMMs of the shared pointwise MLPs (gemm.hip,
// gemm_f32.hip): grid / slab sizing and the concurrency hint, the fixed-order slab sums behind the
// weight-gradient GEMMs, the BatchNorm-backward finalize kernel and the one-launch operand
// preparation from the fp32 master weights of the caller's stock nn.Conv modules
// (models/pointnet2_utils.py:138-142, :166-170, :313-323; models/DGCNN.py:19-33).
#include <stdlib.h>

#include <atomic>

#include "gemm_shared.h"
#include "rowvec.h"

namespace {

// Column layouts of a padded GEMM operand [.., kp] against the layer's real weight [.., k]
// (host side: rowmlp.padded_weight_from).  Returns the real column a padded column j holds, or -1
// for padding.
//   perm = 0      real columns in place, zero padded on the right
//   perm = C > 0  grouped rows (pcb_group_rows_*): the C feature columns first, then the 3
//                 centred coordinates (real order: coordinates first)
//   perm = -D < 0 interpolate+concat rows: the first D columns in place, the rest from pad(D), the
//                 next multiple of `quantum` (8 columns for bf16 rows, 4 for fp32 rows: 16 bytes)
__device__ __forceinline__ int real_column(int j, int k, int perm, int quantum)
{
    if (perm > 0) return j < perm ? 3 + j : (j < perm + 3 ? j - perm : -1);
    if (perm < 0) {
        const int d = -perm, dp = (d + quantum - 1) / quantum * quantum;
        if (j < d) return j;
        const int r = d + (j - dp);
        return (j >= dp && r < k) ? r : -1;
    }
    return j < k ? j : -1;
}

// dW = sum over splits of part[s], in a fixed order (bitwise reproducible weight gradient), written
// in the real weight layout [M, k] (padding columns dropped, see real_column).
// 64 consecutive elements x 16 split-lanes per workgroup: coalesced slab reads, LDS tree at the end.
// Up to kMaxPending sums in ONE launch (blockIdx.y = entry): a stack's backward pass parks the
// reductions of its layers (nothing in that pass reads dW) and runs them together at its end
// instead of one 5-10 us launch behind every weight-gradient GEMM.
constexpr int kMaxPending = 16;
struct PendingReduce {
    const float *part;
    float *dW;
    long elems;
    int splits, N, k, perm, quantum;
    long stride;   // floats from one split's slab to the next (>= elems)
    float *vec;    // optional: `vlen` more sums stored behind each slab's [M,N] part (column sums of dy = a bias gradient)
    int vlen;
};
struct ReduceBatch {
    PendingReduce e[kMaxPending];
};
__global__ __launch_bounds__(1024) void reduce_slabs_multi_kernel(ReduceBatch batch)
{
    const PendingReduce r = batch.e[blockIdx.y];
    const int ex = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const long total = r.elems + (r.vec ? r.vlen : 0);
    if (r.splits <= 16) {
        // few slabs of a large matrix (the wide layers: 6 slabs of 1536 x 1024): one element per lane, all
        // slabs' loads in flight, adds in split order -- no LDS, no idle split-lanes
        for (long e = (long)blockIdx.x * 1024 + threadIdx.x; e < total; e += (long)gridDim.x * 1024) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float x = r.part[(long)(u < r.splits ? u : r.splits - 1) * r.stride + e];
                v[u] = u < r.splits ? x : 0.0f;
            }
            float t = 0.0f;
#pragma unroll
            for (int u = 0; u < 16; ++u) t += v[u];
            if (e >= r.elems) {
                r.vec[e - r.elems] = t;
            } else {
                const long m = e / r.N;
                const int c = real_column((int)(e - m * r.N), r.k, r.perm, r.quantum);
                if (c >= 0) r.dW[m * r.k + c] = t;
            }
        }
        return;
    }
    // many slabs: 256 elements x 16 split-lanes per workgroup and step -- a lane adds four consecutive elements
    // (one 16-byte load per slab; element counts and slab strides are multiples of 4), eight slabs' loads in
    // flight, adds in split order; at most 1024 workgroups (more, shorter-lived ones measured slower)
    __shared__ float4 red4[16][64];
    const long nblk = gridDim.x < 1024 ? gridDim.x : 1024;
    if (blockIdx.x >= nblk) return;
    // (small matrices keep the 64-element chunks: their few chunks should spread over as many workgroups as possible)
    const bool vec_ok = total >= 32768 && ((r.stride | total) & 3) == 0 && (reinterpret_cast<uintptr_t>(r.part) & 15) == 0;
    const long per_iter = vec_ok ? 256 : 64;
    for (long e0 = (long)blockIdx.x * per_iter; e0 < total; e0 += nblk * per_iter) {
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (vec_ok) {
            const long e = e0 + 4L * ex;
            if (e < total) {
                for (int s0 = sy; s0 < r.splits; s0 += 16 * 8) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int sp = s0 + 16 * u;
                        v[u] = *reinterpret_cast<const float4 *>(r.part + (long)(sp < r.splits ? sp : r.splits - 1) * r.stride + e);
                        if (sp >= r.splits) v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
                    }
                }
            }
        } else {
            const long e = e0 + ex;
            if (e < total) {
                for (int s0 = sy; s0 < r.splits; s0 += 16 * 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int sp = s0 + 16 * u;
                        const float x = r.part[(long)(sp < r.splits ? sp : r.splits - 1) * r.stride + e];
                        v[u] = sp < r.splits ? x : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) a.x += v[u];
                }
            }
        }
        red4[sy][ex] = a;
        __syncthreads();
        if (sy == 0) {
            float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float4 q = red4[i][ex];
                t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
            }
            const float tv[4] = {t.x, t.y, t.z, t.w};
            const int cnt = vec_ok ? 4 : 1;
            for (int i = 0; i < cnt; ++i) {
                const long e = vec_ok ? e0 + 4L * ex + i : e0 + ex;
                if (e >= total) break;
                if (e >= r.elems) {
                    r.vec[e - r.elems] = tv[i];
                } else {
                    const long m = e / r.N;
                    const int c = real_column((int)(e - m * r.N), r.k, r.perm, r.quantum);
                    if (c >= 0) r.dW[m * r.k + c] = tv[i];
                }
            }
        }
        __syncthreads();
    }
}
thread_local ReduceBatch g_pending;
thread_local int g_npending = -1;  // < 0: every reduction runs right behind its GEMM (the default)

int launch_reduce_batch(const ReduceBatch &batch, int n, hipStream_t st)
{
    // workgroups an entry can use: few slabs -> 1024 elements per workgroup and step (at most 2048 of them),
    // many slabs -> 256-element chunks walked by at most 1024 workgroups; the grid serves the neediest entry,
    // the others' surplus workgroups leave at once
    long blocks = 1;
    for (int i = 0; i < n; ++i) {
        const long total = batch.e[i].elems + batch.e[i].vlen;
        long b = batch.e[i].splits <= 16 ? (total + 1023) / 1024 : (total >= 32768 ? (total + 255) / 256 : (total + 63) / 64);
        const long cap = batch.e[i].splits <= 16 ? 2048 : 1024;
        b = b > cap ? cap : b;
        blocks = b > blocks ? b : blocks;
    }
    hipLaunchKernelGGL(reduce_slabs_multi_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(1024), 0, st, batch);
    return pcb_check_launch();
}

// p, q of the fused BatchNorm backward of one layer from sums = [nparts][2][C] partial slabs of
// (sum du, sum du*xhat); the parameter gradients the totals amount to go to dgamma, dbeta, dbias
// ([C] each, optional): dbeta = s1, dgamma = s2, dbias = 0 under batch statistics (the mean
// subtraction cancels a bias exactly), scale*s1 otherwise.
// gsums (optional, [2][C]): the same two sums over ALL ranks' rows (SyncBatchNorm: `rows` then
// counts all ranks' rows): p and q come from them, the parameter gradients stay local.
// Block = 32 channels x 32 slab-lanes.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(float *__restrict__ sums, int nparts,
                                                                const float *__restrict__ gsums, long rows, int C,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ invstd,
                                                                int use_batch_stats, float *__restrict__ p,
                                                                float *__restrict__ q, float *__restrict__ dgamma,
                                                                float *__restrict__ dbeta, float *__restrict__ dbias)
{
    __shared__ float red[2][32][32];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s1 = 0.0f, s2 = 0.0f;
    if (c < C) {
        // eight slabs' loads in flight per step, adds in slab order (see bn_finalize_kernel)
        for (int k0 = pl; k0 < nparts; k0 += 32 * 8) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 32 * u, kc = k < nparts ? k : nparts - 1;
                const float va = sums[((long)kc * 2 + 0) * C + c], vb = sums[((long)kc * 2 + 1) * C + c];
                a[u] = k < nparts ? va : 0.0f;
                b[u] = k < nparts ? vb : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s1 += a[u];
                s2 += b[u];
            }
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
    if (nparts == 1) {
        // a single slab was accumulated with atomics: leave it cleared for the next accumulation
        sums[c] = 0.0f;
        sums[C + c] = 0.0f;
    }
    if (dgamma) dgamma[c] = s2;
    if (dbeta) dbeta[c] = s1;
    if (dbias) dbias[c] = use_batch_stats ? 0.0f : scale[c] * s1;
    if (!use_batch_stats) {
        p[c] = 0.0f;
        q[c] = 0.0f;
        return;
    }
    if (gsums) {
        s1 = gsums[c];
        s2 = gsums[C + c];
    }
    const float invR = 1.0f / (float)rows;
    const float a = s1 * invR;       // mean of du
    const float b = s2 * invR;       // mean of du * xhat
    const float sb = scale[c] * b * invstd[c];
    p[c] = -sb;
    q[c] = fmaf(sb, mean[c], -scale[c] * a);
}

// CUs to leave free while another kernel runs beside the backward pass (pcb_set_concurrency_hint).
// Measured with the FPS kernel of the next batch on a side stream (16 workgroups): a 512-workgroup
// persistent gemm_nt slows from 145 to 220 us, one of 448 runs in 163 us either way -- the persistent
// grid assumes it owns every CU, and the workgroups that find their CU taken serialise behind others.
// The only mutable state of the library besides the roofline timer: a hint, read once per entry
// point, that can change grid sizes but never which buffers a launch may touch (slab counts are
// explicit arguments).
std::atomic<int> g_shared_cus{0};

// Tuning knobs for the persistent grids, clamped to [64, PCB_MAX_SLABS].
long grid_knob(const char *name, long def)
{
    const char *e = getenv(name);
    const long v = e ? atol(e) : def;
    return v < 64 ? 64L : (v > PCB_MAX_SLABS ? (long)PCB_MAX_SLABS : v);
}

// ---- weight preparation -------------------------------------------------------------------------
// The fp32 master weights of up to PREP_MAX layers become, in ONE launch, the operands the GEMMs
// read (bf16 or fp32 rows): wp [C, kp] in the padded column layout of the layer's input rows
// (real_column) and, for the input-gradient GEMM, its transpose wt [kp, C].
constexpr int PREP_MAX = 8;
struct PrepLayer {
    const float *w;   // [C, k] fp32, rows ldw floats apart (a column slice of a wider parameter: ldw > k)
    void *wp;         // [C, kp]
    void *wt;         // [kp, C] or NULL
    int C, k, kp, perm, ldw;
};
struct PrepArgs {
    PrepLayer l[PREP_MAX];
    float *zero;   // optional: a float buffer cleared by the same launch (a stack's constants)
    long zero_n;
};

template <typename T>
__device__ __forceinline__ T to_elem(float f);
template <>
__device__ __forceinline__ pcb_bf16 to_elem<pcb_bf16>(float f) { return pcb_f2bf(f); }
template <>
__device__ __forceinline__ float to_elem<float>(float f) { return f; }

template <typename T>
__global__ __launch_bounds__(256) void prep_weights_kernel(PrepArgs args)
{
    if (blockIdx.y == 0)
        for (long e = blockIdx.x * 256 + threadIdx.x; e < args.zero_n; e += gridDim.x * 256) args.zero[e] = 0.0f;
    const PrepLayer L = args.l[blockIdx.y];
    T *const wp = (T *)L.wp;
    T *const wt = (T *)L.wt;
    const int total = L.C * L.kp;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int c = e / L.kp, j = e - c * L.kp;
        const int r = real_column(j, L.k, L.perm, RowVec<T>::E);
        const T h = to_elem<T>(r >= 0 ? L.w[(long)c * L.ldw + r] : 0.0f);
        wp[e] = h;
        if (wt) wt[(long)j * L.C + c] = h;
    }
}

template <typename T>
int prep_weights(int n, const long long *desc, float *zero, long zero_n, void *stream)
{
    if (n < 1 || n > PREP_MAX || !desc || zero_n < 0 || (zero_n && !zero)) return PCB_ERR_INVALID_ARG;
    PrepArgs a;
    a.zero = zero;
    a.zero_n = zero_n;
    int most = 0;
    for (int i = 0; i < n; ++i) {
        const long long *d = desc + 8 * i;
        a.l[i].w = (const float *)d[0];
        a.l[i].wp = (void *)d[1];
        a.l[i].wt = (void *)d[2];
        a.l[i].C = (int)d[3];
        a.l[i].k = (int)d[4];
        a.l[i].kp = (int)d[5];
        a.l[i].perm = (int)d[6];
        a.l[i].ldw = d[7] > 0 ? (int)d[7] : (int)d[4];   // slot [7]: row stride of w in floats (0 = k)
        if (!a.l[i].w || !a.l[i].wp || a.l[i].C <= 0 || a.l[i].k <= 0 || a.l[i].kp < a.l[i].k || a.l[i].ldw < a.l[i].k) return PCB_ERR_INVALID_ARG;
        if (a.l[i].C * a.l[i].kp > most) most = a.l[i].C * a.l[i].kp;
    }
    int gx = (most + 255) / 256;
    if (gx > 2048) gx = 2048;  // (256 made the 1536 x 1024 layers of the bottleneck levels a 21 us launch: 24 scattered 2-byte writes of W^T per lane)
    hipLaunchKernelGGL(prep_weights_kernel<T>, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a);
    return pcb_check_launch();
}

// The same for ANY number of layers, described by a table in device memory (8 int64 per layer as in `desc`:
// w, wp, wt, C, k, kp, perm, 0): the operands of every stack of a network, prepared by one launch per optimiser
// step (rowmlp.prepare_step) instead of one launch at the top of every stack's forward pass.
template <typename T>
__global__ __launch_bounds__(256) void prep_weights_table_kernel(const long long *__restrict__ table, int n)
{
    // slot [7] of a row = the first workgroup of its layer (one 32 x 32 tile per workgroup, rows in ascending
    // order): every workgroup finds its layer by bisection, so the grid holds no idle workgroups however
    // unequal the layers are (a 2-D grid sized for the largest layer spent 20 us dispatching empty ones)
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[8L * mid + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long *d = table + 8L * lo;
    const float *const w = reinterpret_cast<const float *>(static_cast<uintptr_t>(d[0]));
    T *const wp = reinterpret_cast<T *>(static_cast<uintptr_t>(d[1]));
    T *const wt = reinterpret_cast<T *>(static_cast<uintptr_t>(d[2]));
    const int C = (int)d[3], k = (int)d[4], kp = (int)d[5], perm = (int)d[6];
    // a workgroup = one 32 x 32 tile (rows c, columns j) of the layer: W is read and wp written along j, the
    // transpose goes through LDS so that wt is written along c as well (2-byte stores a row of C apart were
    // the whole cost of this kernel)
    __shared__ T tile[32][33];
    const int tiles_j = (kp + 31) / 32;
    const int tid = (int)blockIdx.x - (int)d[7];
    const int c0 = (tid / tiles_j) * 32, j0 = (tid % tiles_j) * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = c0 + ly + 8 * u, j = j0 + lx;
        T h = to_elem<T>(0.0f);
        if (c < C && j < kp) {
            const int r = real_column(j, k, perm, RowVec<T>::E);
            h = to_elem<T>(r >= 0 ? w[(long)c * k + r] : 0.0f);
            wp[(long)c * kp + j] = h;
        }
        tile[ly + 8 * u][lx] = h;
    }
    if (!wt) return;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = j0 + ly + 8 * u, c = c0 + lx;
        if (c < C && j < kp) wt[(long)j * C + c] = tile[lx][ly + 8 * u];
    }
}

template <typename T>
int prep_weights_table(const long long *table, int n, long blocks, void *stream)
{
    if (!table || n < 1 || blocks < 1 || blocks > 0x7fffffffL) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(prep_weights_table_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, n);
    return pcb_check_launch();
}

// Operands of a bias-only conv in one launch: w [n,k] fp32 -> wp [npad,kp] and wt [kp,npad] (zero
// padded), bias [n] -> bp [npad] fp32.  gap = D > 0: the n outputs use the interpolate+concat
// column layout (first D in place, the rest from column pad(D)), see real_column(.., -D).
template <typename T>
__global__ __launch_bounds__(256) void prep_linear_bias_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                                               int n, int k, int npad, int kp, int gap,
                                                               T *__restrict__ wp, T *__restrict__ wt,
                                                               float *__restrict__ bp)
{
    const int total = npad * kp;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int o = e / kp, j = e - o * kp;
        const int r = real_column(o, n, -gap, RowVec<T>::E);   // gap = 0: o < n ? o : -1
        const T h = to_elem<T>((r >= 0 && j < k) ? w[(long)r * k + j] : 0.0f);
        wp[e] = h;
        if (wt) wt[(long)j * npad + o] = h;
        if (j == 0) bp[o] = (r >= 0 && bias) ? bias[r] : 0.0f;
    }
}

template <typename T>
int prep_linear_bias(const float *w, const float *bias, int n, int k, int npad, int kp, int gap, void *wp, void *wt,
                     float *bp, void *stream)
{
    if (!w || !wp || !bp || n <= 0 || k <= 0 || npad < n || kp < k || gap < 0) return PCB_ERR_INVALID_ARG;
    const int blocks = (npad * kp + 255) / 256;
    hipLaunchKernelGGL(prep_linear_bias_kernel<T>, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, (hipStream_t)stream, w,
                       bias, n, k, npad, kp, gap, (T *)wp, (T *)wt, bp);
    return pcb_check_launch();
}

__global__ __launch_bounds__(256) void zero_kernel(uint32_t *__restrict__ p, size_t words)
{
    const size_t vec = words >> 2;
    uint4 *__restrict__ v = reinterpret_cast<uint4 *>(p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vec; i += (size_t)gridDim.x * 256) v[i] = make_uint4(0, 0, 0, 0);
    for (size_t i = (vec << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
__global__ __launch_bounds__(256) void copy_kernel(uint32_t *__restrict__ d, const uint32_t *__restrict__ s, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

}  // namespace

int pcb_zero_async(void *ptr, size_t bytes, hipStream_t st)
{
    if (!ptr || (bytes & 3) || ((uintptr_t)ptr & 15)) return PCB_ERR_INVALID_ARG;
    if (!bytes) return PCB_OK;
    const size_t words = bytes >> 2;
    size_t blocks = ((words >> 2) + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t *)ptr, words);
    return pcb_check_launch();
}

// two buffers cleared by one launch (the du and dWx accumulators of a gathered layer's backward pass)
static __global__ __launch_bounds__(256) void zero2_kernel(uint32_t *__restrict__ a, size_t wa, uint32_t *__restrict__ b, size_t wb)
{
    const size_t va = wa >> 2;   // a: 16-byte stores (the large one), tail and b word by word
    uint4 *__restrict__ a4 = reinterpret_cast<uint4 *>(a);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < va; i += (size_t)gridDim.x * 256) a4[i] = make_uint4(0, 0, 0, 0);
    for (size_t i = (va << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < wa; i += (size_t)gridDim.x * 256) a[i] = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < wb; i += (size_t)gridDim.x * 256) b[i] = 0u;
}
int pcb_zero2_async(void *a, size_t bytes_a, void *b, size_t bytes_b, hipStream_t st)
{
    if (!b || !bytes_b) return pcb_zero_async(a, bytes_a, st);
    if (!a || (bytes_a & 3) || (bytes_b & 3) || ((uintptr_t)a & 15)) return PCB_ERR_INVALID_ARG;
    const size_t wa = bytes_a >> 2, wb = bytes_b >> 2;
    size_t blocks = (((wa >> 2) > wb ? (wa >> 2) : wb) + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(zero2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t *)a, wa, (uint32_t *)b, wb);
    return pcb_check_launch();
}

// Many device-to-device copies in one launch: table (device memory) = n rows of 4 int64 {dst, src, bytes, first
// workgroup}; a workgroup moves 16 KB (bisection over the rows as in prep_weights_table_kernel).  The captured
// step's staging -> live hand-over of the sampling results was 24 copy nodes at the top of every replay.
static __global__ __launch_bounds__(256) void copy_table_kernel(const long long *__restrict__ table, int n)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[4L * mid + 3] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long *d = table + 4L * lo;
    char *const dst = reinterpret_cast<char *>(static_cast<uintptr_t>(d[0]));
    const char *const src = reinterpret_cast<const char *>(static_cast<uintptr_t>(d[1]));
    const long bytes = (long)d[2];
    const long base = ((long)blockIdx.x - (long)d[3]) * 16384;
    const long end = base + 16384 < bytes ? base + 16384 : bytes;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        for (long o = base + 16L * threadIdx.x; o + 16 <= end; o += 16 * 256)
            *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(src + o);
        for (long o = base + ((end - base) & ~15L) + 4L * threadIdx.x; o + 4 <= end; o += 4 * 256)
            *reinterpret_cast<uint32_t *>(dst + o) = *reinterpret_cast<const uint32_t *>(src + o);
    } else {
        for (long o = base + 4L * threadIdx.x; o + 4 <= end; o += 4 * 256)
            *reinterpret_cast<uint32_t *>(dst + o) = *reinterpret_cast<const uint32_t *>(src + o);
    }
}
extern "C" int pcb_copy_table(const long long *table, int n, long blocks, void *stream)
{
    if (!table || n < 1 || blocks < 1 || blocks > 0x7fffffffL) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(copy_table_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, n);
    return pcb_check_launch();
}

// The same with the rows passed BY VALUE in the kernel arguments (host arrays, up to 32 copies per launch): capturable
// where the addresses only exist at capture time -- results a captured step allocates and copies to fixed buffers
// (StaticSampling.compute: ball query / three_nn / CSR results into the staging set).
struct CopyList {
    long long row[32][4];
};
static __global__ __launch_bounds__(256) void copy_list_kernel(const CopyList list, int n)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (list.row[mid][3] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    char *const dst = reinterpret_cast<char *>(static_cast<uintptr_t>(list.row[lo][0]));
    const char *const src = reinterpret_cast<const char *>(static_cast<uintptr_t>(list.row[lo][1]));
    const long bytes = (long)list.row[lo][2];
    const long base = ((long)blockIdx.x - (long)list.row[lo][3]) * 16384;
    const long end = base + 16384 < bytes ? base + 16384 : bytes;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        for (long o = base + 16L * threadIdx.x; o + 16 <= end; o += 16 * 256)
            *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(src + o);
        for (long o = base + ((end - base) & ~15L) + 4L * threadIdx.x; o + 4 <= end; o += 4 * 256)
            *reinterpret_cast<uint32_t *>(dst + o) = *reinterpret_cast<const uint32_t *>(src + o);
    } else {
        for (long o = base + 4L * threadIdx.x; o + 4 <= end; o += 4 * 256)
            *reinterpret_cast<uint32_t *>(dst + o) = *reinterpret_cast<const uint32_t *>(src + o);
    }
}
extern "C" int pcb_copy_list(const long long *dst, const long long *src, const long long *bytes, int n, void *stream)
{
    if (!dst || !src || !bytes || n < 1) return PCB_ERR_INVALID_ARG;
    for (int first = 0; first < n; first += 32) {
        CopyList list;
        const int m = n - first < 32 ? n - first : 32;
        long long blocks = 0;
        for (int i = 0; i < m; ++i) {
            const long long b = bytes[first + i];
            if (b <= 0 || (b & 3) || ((dst[first + i] | src[first + i]) & 3)) return PCB_ERR_INVALID_ARG;
            list.row[i][0] = dst[first + i];
            list.row[i][1] = src[first + i];
            list.row[i][2] = b;
            list.row[i][3] = blocks;
            blocks += (b + 16383) / 16384;
        }
        if (blocks > 0x7fffffffLL) return PCB_ERR_INVALID_ARG;
        hipLaunchKernelGGL(copy_list_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, list, m);
    }
    return pcb_check_launch();
}

int pcb_copy_async(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    if (!dst || !src || (bytes & 3)) return PCB_ERR_INVALID_ARG;
    if (!bytes) return PCB_OK;
    const size_t words = bytes >> 2;
    size_t blocks = (words + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t *)dst, (const uint32_t *)src, words);
    return pcb_check_launch();
}

int pcb_busy_cus() { return g_shared_cus.load(std::memory_order_relaxed); }

long pcb_nt_grid_x(int pro, long R, int N, int busy_cus)
{
    const long tiles = (R + PCB_NT_BM - 1) / PCB_NT_BM;
    const long ny = (N + PCB_NT_BN - 1) / PCB_NT_BN;
    static const long fwd_chip = grid_knob("PCB_NT_FWD_GRID", 512);
    static const long bwd_chip = grid_knob("PCB_NT_BWD_GRID", 512);
    // while another kernel holds CUs the persistent grids leave them alone: 2 workgroups fit a CU
    // (512 threads each in the forward kernels, 256 in the register-heavier backward ones)
    long chip = (pro <= PCB_PRO_BNACT ? fwd_chip : bwd_chip) - 2L * busy_cus;
    if (chip < 64) chip = 64;
    const long resident = chip / ny > 0 ? chip / ny : 1;
    return tiles < resident ? tiles : resident;
}

long pcb_tn_splits(long R, int M, int N, long *rows_per_split, long target)
{
    const long tiles = (long)((M + PCB_TN_BM - 1) / PCB_TN_BM) * ((N + PCB_TN_BN - 1) / PCB_TN_BN);
    if (target < 64) target = 64;
    long splits = (target + tiles - 1) / tiles;
    const long max_splits = (R + 8 * PCB_TN_RS - 1) / (8 * PCB_TN_RS);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long rps = (R + splits - 1) / splits;
    rps = (rps + PCB_TN_RS - 1) / PCB_TN_RS * PCB_TN_RS;
    *rows_per_split = rps;
    return (R + rps - 1) / rps;
}

int pcb_reduce_slabs(const float *part, int splits, long elems, float *dW, int N, int out_cols, int out_perm,
                     int quantum, hipStream_t st)
{
    return pcb_reduce_slabs_vec(part, splits, elems, elems, dW, N, out_cols, out_perm, quantum, nullptr, 0, st);
}

int pcb_reduce_slabs_vec(const float *part, int splits, long stride, long elems, float *dW, int N, int out_cols,
                         int out_perm, int quantum, float *vec, int vlen, hipStream_t st)
{
    const PendingReduce r = {part, dW, elems, splits, N, out_cols, out_perm, quantum, stride, vec, vec ? vlen : 0};
    if (g_npending >= 0 && g_npending < kMaxPending) {
        g_pending.e[g_npending++] = r;
        return PCB_OK;
    }
    ReduceBatch one;
    one.e[0] = r;
    return launch_reduce_batch(one, 1, st);
}

// Deferred slab reductions (see reduce_slabs_multi_kernel): between begin and flush every
// weight-gradient GEMM on this thread needs its OWN workspace region.
void pcb_defer_reduces_begin() { g_npending = 0; }

int pcb_defer_reduces_flush(hipStream_t st)
{
    const int n = g_npending;
    g_npending = -1;
    if (n <= 0) return PCB_OK;
    return launch_reduce_batch(g_pending, n, st);
}

extern "C" {

int pcb_set_concurrency_hint(int busy_cus)
{
    g_shared_cus.store(busy_cus < 0 ? 0 : (busy_cus > 64 ? 64 : busy_cus), std::memory_order_relaxed);
    return PCB_OK;
}

int pcb_gemm_nt_partials(int pro, long R, int N)
{
    if (R <= 0 || N <= 0 || pro < 0 || pro > 3) return 0;
    return (int)pcb_nt_grid_x(pro, R, N, pcb_busy_cus());
}

long pcb_gemm_tn_workspace(long R, int M, int N)
{
    if (R <= 0 || M <= 0 || N <= 0) return 0;
    long rps;
    return pcb_tn_splits(R, M, N, &rps, 512) * ((long)M * N + M);  // + M: the column sums pcb_gemm_tn_bias_* keeps per split
}

int pcb_bn_bwd_finalize(float *sums, int nparts, long rows, int C, const float *scale, const float *mean,
                        const float *invstd, int use_batch_stats, float *p, float *q, float *dgamma, float *dbeta,
                        float *dbias, const float *global_sums, void *stream)
{
    if (!sums || nparts < 1 || !scale || !mean || !invstd || !p || !q || C <= 0 || rows <= 0) return PCB_ERR_INVALID_ARG;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, sums,
                       nparts, global_sums, rows, C, scale, mean, invstd, use_batch_stats, p, q, dgamma, dbeta, dbias);
    return pcb_check_launch();
}

int pcb_prep_weights_bf16(int n, const long long *desc, void *stream)
{
    return prep_weights<pcb_bf16>(n, desc, nullptr, 0, stream);
}
int pcb_prep_weights_zero_bf16(int n, const long long *desc, float *zero, long zero_n, void *stream)
{
    return prep_weights<pcb_bf16>(n, desc, zero, zero_n, stream);
}
int pcb_prep_weights_f32(int n, const long long *desc, void *stream)
{
    return prep_weights<float>(n, desc, nullptr, 0, stream);
}
int pcb_prep_weights_zero_f32(int n, const long long *desc, float *zero, long zero_n, void *stream)
{
    return prep_weights<float>(n, desc, zero, zero_n, stream);
}

int pcb_prep_weights_table_bf16(const long long *table, int n, long blocks, void *stream)
{
    return prep_weights_table<pcb_bf16>(table, n, blocks, stream);
}
int pcb_prep_weights_table_f32(const long long *table, int n, long blocks, void *stream)
{
    return prep_weights_table<float>(table, n, blocks, stream);
}

int pcb_prep_linear_bias_bf16(const float *w, const float *bias, int n, int k, int npad, int kp, int gap, void *wp,
                              void *wt, float *bp, void *stream)
{
    return prep_linear_bias<pcb_bf16>(w, bias, n, k, npad, kp, gap, wp, wt, bp, stream);
}
int pcb_prep_linear_bias_f32(const float *w, const float *bias, int n, int k, int npad, int kp, int gap, void *wp,
                             void *wt, float *bp, void *stream)
{
    return prep_linear_bias<float>(w, bias, n, k, npad, kp, gap, wp, wt, bp, stream);
}

}  // extern "C"
