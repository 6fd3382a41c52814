// k-nearest centroids (k <= 4) and inverse-distance interpolation for gfx950.
//
// Replaces, in FeaturePropagation.forward / EnhancedFeaturePropagation.forward of the reference
// (models/pointnet2_utils.py:185-196 and :253-267): square_distance -> [B,N,S] fp32, a FULL sort of
// it (values + int64 indices), a slice of the first k, the weight computation and the gather-sum.
// three_nn: one lane per query point, candidates staged once per workgroup through LDS as
// (x, y, z, |p|^2) float4 and read back as wave-wide broadcasts; the k best live in registers.
// Equal distances keep ascending index order (the reference's CPU sort is stable).
#include "rowvec.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 1024;  // candidates per LDS tile (16 KiB)

template <int K>
__global__ __launch_bounds__(kThreads) void three_nn_kernel(const float *__restrict__ xyz1,
                                                             const float *__restrict__ xyz2, int N,
                                                             int S, float *__restrict__ out_d,
                                                             int64_t *__restrict__ out_i)
{
    __shared__ float4 tile[kTile];
    const int b = blockIdx.y;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    const bool valid = n < N;
    const float *__restrict__ q = xyz1 + ((size_t)b * N + (valid ? n : N - 1)) * 3;
    const float *__restrict__ c = xyz2 + (size_t)b * S * 3;
    const float sx = q[0], sy = q[1], sz = q[2];
    const float s2 = sq_norm3(sx, sy, sz);

    float bd[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        bd[k] = INFINITY;
        bi[k] = 0;
    }

    for (int base = 0; base < S; base += kTile) {
        const int cnt = min(kTile, S - base);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += kThreads) {
            const float x = c[(base + j) * 3 + 0];
            const float y = c[(base + j) * 3 + 1];
            const float z = c[(base + j) * 3 + 2];
            tile[j] = make_float4(x, y, z, sq_norm3(x, y, z));
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float4 t = tile[j];
            const float d = sqdist_expand(sx, sy, sz, s2, t.x, t.y, t.z, t.w);
            if (d < bd[K - 1]) {
                bd[K - 1] = d;
                bi[K - 1] = base + j;
#pragma unroll
                for (int k = K - 1; k > 0; --k) {
                    // strict "<": a later candidate never overtakes an equal earlier one
                    if (bd[k] < bd[k - 1]) {
                        const float td = bd[k];
                        bd[k] = bd[k - 1];
                        bd[k - 1] = td;
                        const int ti = bi[k];
                        bi[k] = bi[k - 1];
                        bi[k - 1] = ti;
                    }
                }
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            out_d[((size_t)b * N + n) * K + k] = bd[k];
            out_i[((size_t)b * N + n) * K + k] = (int64_t)bi[k];
        }
    }
}

// out[b,n,c] = sum_k w_k * feat[b, idx_k, c]; one lane per (n, c), c fastest (coalesced rows).
template <int K>
__global__ __launch_bounds__(256) void interpolate_kernel(const float *__restrict__ feat,
                                                           const float *__restrict__ d2,
                                                           const int64_t *__restrict__ idx, int N,
                                                           int S, int C, float *__restrict__ out,
                                                           float *__restrict__ out_w, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t row = e / C;  // b*N + n
        const size_t b = row / N;
        float w[K];
        float norm = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            w[k] = __fdiv_rn(1.0f, __fadd_rn(d2[row * K + k], 1e-8f));
            norm = k ? __fadd_rn(norm, w[k]) : w[k];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            w[k] = __fdiv_rn(w[k], norm);
            const int j = clamp_index(idx[row * K + k], S);
            const float v = __fmul_rn(feat[(b * S + j) * C + c], w[k]);
            acc = k ? __fadd_rn(acc, v) : v;
        }
        out[e] = acc;
        if (out_w && c == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) out_w[row * K + k] = w[k];
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void interpolate_bwd_kernel(const float *__restrict__ g,
                                                               const float *__restrict__ w,
                                                               const int64_t *__restrict__ idx,
                                                               int N, int S, int C,
                                                               float *__restrict__ gfeat,
                                                               size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t row = e / C;
        const size_t b = row / N;
        const float go = g[e];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = clamp_index(idx[row * K + k], S);
            atomicAdd(&gfeat[(b * S + j) * C + c], w[row * K + k] * go);
        }
    }
}

inline int grid_for(size_t total)
{
    size_t blocks = (total + 255) / 256;
    return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

}  // namespace

extern "C" int pcb_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int k,
                            float *out_d2, int64_t *out_idx, void *stream)
{
    if (!xyz1 || !xyz2 || !out_d2 || !out_idx || B <= 0 || N <= 0 || S <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 4 || S < k) return PCB_ERR_INVALID_ARG;
    const dim3 grid((N + kThreads - 1) / kThreads, B);
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL((three_nn_kernel<1>), grid, dim3(kThreads), 0, st, xyz1, xyz2, N, S, out_d2, out_idx); break;
        case 2: hipLaunchKernelGGL((three_nn_kernel<2>), grid, dim3(kThreads), 0, st, xyz1, xyz2, N, S, out_d2, out_idx); break;
        case 3: hipLaunchKernelGGL((three_nn_kernel<3>), grid, dim3(kThreads), 0, st, xyz1, xyz2, N, S, out_d2, out_idx); break;
        default: hipLaunchKernelGGL((three_nn_kernel<4>), grid, dim3(kThreads), 0, st, xyz1, xyz2, N, S, out_d2, out_idx); break;
    }
    pcb_account(12.0 * ((double)N + S) * B + 12.0 * (double)k * N * B);
    return pcb_check_launch();
}

extern "C" int pcb_interpolate(const float *feat, const float *d2, const int64_t *idx, int B, int N,
                               int S, int C, int k, float *out, float *out_w, void *stream)
{
    if (!feat || !d2 || !idx || !out || B <= 0 || N <= 0 || S <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 4) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * N * C;
    const dim3 grid(grid_for(total));
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL((interpolate_kernel<1>), grid, dim3(256), 0, st, feat, d2, idx, N, S, C, out, out_w, total); break;
        case 2: hipLaunchKernelGGL((interpolate_kernel<2>), grid, dim3(256), 0, st, feat, d2, idx, N, S, C, out, out_w, total); break;
        case 3: hipLaunchKernelGGL((interpolate_kernel<3>), grid, dim3(256), 0, st, feat, d2, idx, N, S, C, out, out_w, total); break;
        default: hipLaunchKernelGGL((interpolate_kernel<4>), grid, dim3(256), 0, st, feat, d2, idx, N, S, C, out, out_w, total); break;
    }
    pcb_account(4.0 * C * ((double)S + N) * B + 16.0 * (double)k * N * B);
    return pcb_check_launch();
}

extern "C" int pcb_interpolate_bwd(const float *grad_out, const float *w, const int64_t *idx, int B,
                                   int N, int S, int C, int k, float *grad_feat, void *stream)
{
    if (!grad_out || !w || !idx || !grad_feat || B <= 0 || N <= 0 || S <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 4) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * N * C;
    const dim3 grid(grid_for(total));
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL((interpolate_bwd_kernel<1>), grid, dim3(256), 0, st, grad_out, w, idx, N, S, C, grad_feat, total); break;
        case 2: hipLaunchKernelGGL((interpolate_bwd_kernel<2>), grid, dim3(256), 0, st, grad_out, w, idx, N, S, C, grad_feat, total); break;
        case 3: hipLaunchKernelGGL((interpolate_bwd_kernel<3>), grid, dim3(256), 0, st, grad_out, w, idx, N, S, C, grad_feat, total); break;
        default: hipLaunchKernelGGL((interpolate_bwd_kernel<4>), grid, dim3(256), 0, st, grad_out, w, idx, N, S, C, grad_feat, total); break;
    }
    pcb_account(4.0 * C * ((double)N + (double)k * N) * B);
    return pcb_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Interpolation straight into a (padded, concatenated) GEMM input buffer, and its backward
// as a segmented reduction over an inverted index instead of fp32 atomics -- for bf16 rows and for
// fp32 rows (rowvec.h).
//
// FeaturePropagation.forward (models/pointnet2_utils.py:191-203) interpolates the coarse features
// and concatenates them behind the skip features; here the interpolated columns are written at
// their final place in the row buffer of the following GEMM.
// Backward: grad_feat[b,s,:] = sum over (n,q) with idx[b,n,q] == s of w[b,n,q] * g[b,n,:].
// The atomic form moves B*N*C*k fp32 adds to memory (1 GB at B=16, N=16384, C=256: atomic-rate
// bound); the inverted index (counting sort of the (n,q) pairs by target s, ~1 M integers) lets one
// wave sum each target's ~N*k/S contribution rows with plain coalesced 16-byte reads.
namespace {

// one lane per (row, 16-byte channel chunk)
template <typename T, int K>
__global__ __launch_bounds__(256) void interpolate_rows_kernel(const T *__restrict__ feat,
                                                                const float *__restrict__ d2,
                                                                const int64_t *__restrict__ idx, int N, int S,
                                                                int C, T *__restrict__ out, int ld,
                                                                int col0, float *__restrict__ out_w,
                                                                long nchunk, const T *__restrict__ skip = nullptr,
                                                                int skip_ld = 0, int d1 = 0)
{
    constexpr int E = RowVec<T>::E;
    // with skip rows [B*N, d1] (row stride skip_ld): the chunks [0, col0) of every output row are written here too --
    // the skip features, then zeros up to col0 (FeaturePropagation's torch.cat([points1, interpolated], -1), :201 / :272)
    const int CS = skip ? col0 / E : 0;
    const int CT = C / E + CS;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nchunk; e += (long)gridDim.x * 256) {
        int cc = (int)(e % CT);
        const long row = e / CT;  // b*N + n
        if (cc < CS) {
            float f[E];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int c = cc * E + i;
                f[i] = c < d1 ? RowVec<T>::one(skip + row * (long)skip_ld + c) : 0.0f;
            }
            *reinterpret_cast<uint4 *>(out + row * (long)ld + cc * E) = RowVec<T>::pack(f);
            continue;
        }
        cc -= CS;
        const long b = row / N;
        float w[K];
        float norm = 0.0f;
        // fp32 rows: correctly rounded divisions, the reference's `1.0 / (dists + 1e-8)` and `/ norm` (:196-198).
        // bf16 rows: the hardware reciprocal (1 ulp) -- the 2K IEEE divisions, repeated by every lane of a row, were
        // a third of this kernel's instructions, and the products are rounded to bf16 anyway.
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float d = __fadd_rn(d2[row * K + k], 1e-8f);
            w[k] = E == 4 ? __fdiv_rn(1.0f, d) : __builtin_amdgcn_rcpf(d);
            norm = k ? __fadd_rn(norm, w[k]) : w[k];
        }
        const float rnorm = E == 4 ? 0.0f : __builtin_amdgcn_rcpf(norm);
        float acc[E];
#pragma unroll
        for (int i = 0; i < E; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            w[k] = E == 4 ? __fdiv_rn(w[k], norm) : w[k] * rnorm;
            const int j = clamp_index(idx[row * K + k], S);
            float f[E];
            RowVec<T>::unpack(*reinterpret_cast<const uint4 *>(feat + ((b * S + j) * (long)C + cc * E)), f);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                if (E == 4) {
                    // fp32 rows: product and sum rounded separately, in neighbour order -- the
                    // reference's torch.sum(index_points(..) * weight, dim=2) (as interpolate_kernel)
                    const float v = __fmul_rn(f[i], w[k]);
                    acc[i] = k ? __fadd_rn(acc[i], v) : v;
                } else {
                    acc[i] = fmaf(f[i], w[k], acc[i]);
                }
            }
        }
        *reinterpret_cast<uint4 *>(out + row * (long)ld + col0 + cc * E) = RowVec<T>::pack(acc);
        if (out_w && cc == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) out_w[row * K + k] = w[k];
        }
    }
}

// counting sort of the (n, q) pairs by target: pass 1 counts, pass 2 places
__global__ __launch_bounds__(256) void csr_count_kernel(const int64_t *__restrict__ idx, int N, int S, int K,
                                                         int *__restrict__ count, long total)
{
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / ((long)N * K);
        atomicAdd(&count[b * S + clamp_index(idx[e], S)], 1);
    }
}
__global__ __launch_bounds__(256) void csr_fill_kernel(const int64_t *__restrict__ idx, int N, int S, int K,
                                                        const long *__restrict__ offsets,
                                                        int *__restrict__ cursor, int *__restrict__ entries,
                                                        long total)
{
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / ((long)N * K);
        const long seg = b * S + clamp_index(idx[e], S);
        const int pos = atomicAdd(&cursor[seg], 1);
        entries[offsets[seg] + pos] = (int)(e - b * (long)N * K);  // n*K + q inside the scene
    }
}

// one wave per target (b, s).  An entry's row is read by `lpe` lanes (E channels each; lpe = the row's chunks rounded up
// to a power of two, at most 32: wider rows loop over blocks of 32*E channels), so the wave's 64 / lpe lane groups walk
// 64 / lpe entries side by side -- 2 for the wide decoder rows, 8 for a 64-column row.  g is a [B*N, ld] buffer,
// columns col0 .. col0+C.
template <typename T>
__global__ __launch_bounds__(256) void interpolate_bwd_csr_kernel(const T *__restrict__ g, int ld, int col0,
                                                                   const float *__restrict__ w,
                                                                   const long *__restrict__ offsets,
                                                                   const int *__restrict__ entries, int N,
                                                                   int S, int C, int K,
                                                                   T *__restrict__ gfeat, long segments)
{
    constexpr int E = RowVec<T>::E;
    const int lane = threadIdx.x & 63;
    int lpe = 1;
    while (lpe < 32 && lpe * E < C) lpe <<= 1;
    const int groups = 64 / lpe, grp = lane / lpe, gl = lane % lpe;
    const long seg = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= segments) return;  // wave-uniform
    const long b = seg / S;
    const long beg = offsets[seg], end = offsets[seg + 1];
    for (int c0 = 0; c0 < C; c0 += lpe * E) {
        const int c = c0 + gl * E;
        float acc[E];
#pragma unroll
        for (int i = 0; i < E; ++i) acc[i] = 0.0f;
        // four entries per lane group and step: their index, weight and row loads are issued together (one entry per
        // step made every row load wait for its own index load: a chain of two memory latencies per entry), the
        // products are added in entry order as before
        const int cs = c < C ? c : 0;
        for (long e0 = beg + grp; e0 < end; e0 += 4 * groups) {
            int ent[4];
            long row[4];
            float wt[4];
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long e = e0 + groups * u < end ? e0 + groups * u : beg;     // (beg: a valid entry, masked below)
                ent[u] = entries[e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long n = ent[u] / K;
                row[u] = b * N + n;
                wt[u] = w[row[u] * K + (ent[u] - (int)n * K)];
                v[u] = *reinterpret_cast<const uint4 *>(g + row[u] * (long)ld + col0 + cs);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (e0 + groups * u >= end) break;
                if (c < C) {
                    float f[E];
                    RowVec<T>::unpack(v[u], f);
#pragma unroll
                    for (int i = 0; i < E; ++i) acc[i] = fmaf(f[i], wt[u], acc[i]);
                }
            }
        }
        for (int o = lpe; o < 64; o <<= 1)
#pragma unroll
            for (int i = 0; i < E; ++i) acc[i] += __shfl_xor(acc[i], o);
        if (grp == 0 && c < C)
            *reinterpret_cast<uint4 *>(gfeat + seg * (long)C + c) = RowVec<T>::pack(acc);
    }
}

template <typename T>
int interpolate_rows(const void *feat, const float *d2, const int64_t *idx, int B, int N, int S, int C, int k,
                     void *out, int ld, int col0, float *out_w, void *stream, const void *skip = nullptr, int skip_ld = 0,
                     int d1 = 0)
{
    constexpr int E = RowVec<T>::E;
    if (!feat || !d2 || !idx || !out || B <= 0 || N <= 0 || S <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 4 || (C % E) || (ld % E) || (col0 % E) || col0 + C > ld) return PCB_ERR_INVALID_ARG;
    if (skip && (d1 < 1 || d1 > col0 || skip_ld < d1)) return PCB_ERR_INVALID_ARG;
    const long nchunk = (long)B * N * (C / E + (skip ? col0 / E : 0));
    const dim3 grid(grid_for((size_t)nchunk));
    hipStream_t st = (hipStream_t)stream;
    const T *f = (const T *)feat;
    const T *sk = (const T *)skip;
    T *o = (T *)out;
    switch (k) {
        case 1: hipLaunchKernelGGL((interpolate_rows_kernel<T, 1>), grid, dim3(256), 0, st, f, d2, idx, N, S, C, o, ld, col0, out_w, nchunk, sk, skip_ld, d1); break;
        case 2: hipLaunchKernelGGL((interpolate_rows_kernel<T, 2>), grid, dim3(256), 0, st, f, d2, idx, N, S, C, o, ld, col0, out_w, nchunk, sk, skip_ld, d1); break;
        case 3: hipLaunchKernelGGL((interpolate_rows_kernel<T, 3>), grid, dim3(256), 0, st, f, d2, idx, N, S, C, o, ld, col0, out_w, nchunk, sk, skip_ld, d1); break;
        default: hipLaunchKernelGGL((interpolate_rows_kernel<T, 4>), grid, dim3(256), 0, st, f, d2, idx, N, S, C, o, ld, col0, out_w, nchunk, sk, skip_ld, d1); break;
    }
    pcb_account(sizeof(T) * (double)C * ((double)S + N) * B + 12.0 * (double)k * N * B);
    return pcb_check_launch();
}

template <typename T>
int interpolate_bwd_csr(const void *grad_rows, int ld, int col0, const float *w, const long *offsets, const int *entries,
                        int B, int N, int S, int C, int k, void *grad_feat, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!grad_rows || !w || !offsets || !entries || !grad_feat || B <= 0 || N <= 0 || S <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || (C % E) || (ld % E) || (col0 % E) || col0 + C > ld) return PCB_ERR_INVALID_ARG;
    const long segments = (long)B * S;
    hipLaunchKernelGGL(interpolate_bwd_csr_kernel<T>, dim3((unsigned)((segments + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const T *)grad_rows, ld, col0, w, offsets, entries, N, S, C, k,
                       (T *)grad_feat, segments);
    pcb_account(sizeof(T) * (double)C * ((double)k * N + S) * B + 8.0 * (double)k * N * B);
    return pcb_check_launch();
}

}  // namespace

extern "C" int pcb_interpolate_bf16(const void *feat, const float *d2, const int64_t *idx, int B, int N, int S,
                                    int C, int k, void *out, int ld, int col0, float *out_w, void *stream)
{
    return interpolate_rows<pcb_bf16>(feat, d2, idx, B, N, S, C, k, out, ld, col0, out_w, stream);
}

extern "C" int pcb_interpolate_rows_f32(const void *feat, const float *d2, const int64_t *idx, int B, int N, int S,
                                        int C, int k, void *out, int ld, int col0, float *out_w, void *stream)
{
    return interpolate_rows<float>(feat, d2, idx, B, N, S, C, k, out, ld, col0, out_w, stream);
}

// the same with the skip features written by the same launch: out[row][0 .. d1) = skip[row][0 .. d1) (rows skip_ld
// elements apart), out[row][d1 .. col0) = 0 -- the whole torch.cat([points1, interpolated], dim=-1) in one pass
extern "C" int pcb_interpolate_skip_bf16(const void *feat, const float *d2, const int64_t *idx, int B, int N, int S, int C,
                                         int k, void *out, int ld, int col0, float *out_w, const void *skip, int skip_ld,
                                         int d1, void *stream)
{
    return interpolate_rows<pcb_bf16>(feat, d2, idx, B, N, S, C, k, out, ld, col0, out_w, stream, skip, skip_ld, d1);
}

extern "C" int pcb_interpolate_rows_skip_f32(const void *feat, const float *d2, const int64_t *idx, int B, int N, int S,
                                             int C, int k, void *out, int ld, int col0, float *out_w, const void *skip,
                                             int skip_ld, int d1, void *stream)
{
    return interpolate_rows<float>(feat, d2, idx, B, N, S, C, k, out, ld, col0, out_w, stream, skip, skip_ld, d1);
}

extern "C" int pcb_interp_csr_count(const int64_t *idx, int B, int N, int S, int k, int *count, void *stream)
{
    if (!idx || !count || B <= 0 || N <= 0 || S <= 0 || k < 1) return PCB_ERR_INVALID_ARG;
    const long total = (long)B * N * k;
    hipLaunchKernelGGL(csr_count_kernel, dim3(grid_for((size_t)total)), dim3(256), 0, (hipStream_t)stream, idx, N, S, k,
                       count, total);
    return pcb_check_launch();
}

extern "C" int pcb_interp_csr_fill(const int64_t *idx, int B, int N, int S, int k, const long *offsets,
                                   int *cursor, int *entries, void *stream)
{
    if (!idx || !offsets || !cursor || !entries || B <= 0 || N <= 0 || S <= 0 || k < 1) return PCB_ERR_INVALID_ARG;
    const long total = (long)B * N * k;
    hipLaunchKernelGGL(csr_fill_kernel, dim3(grid_for((size_t)total)), dim3(256), 0, (hipStream_t)stream, idx, N, S, k,
                       offsets, cursor, entries, total);
    return pcb_check_launch();
}

extern "C" int pcb_interpolate_bwd_csr_bf16(const void *grad_rows, int ld, int col0, const float *w,
                                            const long *offsets, const int *entries, int B, int N, int S,
                                            int C, int k, void *grad_feat, void *stream)
{
    return interpolate_bwd_csr<pcb_bf16>(grad_rows, ld, col0, w, offsets, entries, B, N, S, C, k, grad_feat, stream);
}

extern "C" int pcb_interpolate_bwd_csr_f32(const void *grad_rows, int ld, int col0, const float *w,
                                           const long *offsets, const int *entries, int B, int N, int S,
                                           int C, int k, void *grad_feat, void *stream)
{
    return interpolate_bwd_csr<float>(grad_rows, ld, col0, w, offsets, entries, B, N, S, C, k, grad_feat, stream);
}
