// Global max-pool over the points of a scene, on channels-last rows: DGCNN's
//     x = F.adaptive_max_pool1d(x, 1)            (models/DGCNN.py:160, after conv5 / bn5 / LeakyReLU)
// i.e. out[b, c] = max_n rows[b*N + n, c], and its backward dz[b*N + n, c] = (n == arg[b, c]) ? g[b, c] : 0.
// The reference (and ATen) run it as a reduction that returns int64 indices plus, in backward, a
// zero-fill and a scatter through those indices.  Here: one pass over the rows that folds (value, row)
// into a sortable 64-bit key per (scene, channel) with a 64-bit atomic max -- ties go to the LOWEST row,
// like torch.max on the CPU -- then a tiny decode; the backward is one dense write (zeros and the
// selected gradients in the same pass), the row indices int32 and range-checked.  Both row types.
#include "rowvec.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ unsigned sortable(float v)  // monotonic float -> unsigned
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unsortable(unsigned u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// grid = (row splits, column blocks, B); block = CTb column chunks x RT row-lanes.
// keys [B, C] u64, zeroed by the caller: hi = sortable(value), lo = ~row.
template <typename T>
__global__ __launch_bounds__(kThreads) void scene_max_kernel(const uint4 *__restrict__ rows, int N, int C, int CTb,
                                                              unsigned long long *__restrict__ keys)
{
    constexpr int E = RowVec<T>::E;
    __shared__ unsigned long long red[kThreads * E];
    const int CT = C / E;
    const int c_chunk0 = blockIdx.y * CTb;
    const int nch = CT - c_chunk0 < CTb ? CT - c_chunk0 : CTb;  // chunks of this block
    const int RT = kThreads / CTb;
    const int cc = threadIdx.x % CTb, rl = threadIdx.x / CTb;
    const long b = blockIdx.z;
    unsigned long long best[E];
#pragma unroll
    for (int i = 0; i < E; ++i) best[i] = 0ull;
    if (cc < nch && rl < RT) {
        const int per = (N + gridDim.x - 1) / gridDim.x;
        const int n0 = blockIdx.x * per, n1 = n0 + per < N ? n0 + per : N;
        for (int n = n0 + rl; n < n1; n += RT) {
            float f[E];
            RowVec<T>::unpack(rows[(b * N + n) * CT + c_chunk0 + cc], f);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                // NaN: torch.max propagates it; sortable() places a positive NaN above +inf
                const unsigned long long key = ((unsigned long long)sortable(f[i]) << 32) | (unsigned)(~n);
                best[i] = key > best[i] ? key : best[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) red[threadIdx.x * E + i] = best[i];
    __syncthreads();
    // row-lanes of the block meet in LDS; one atomic per (scene, channel) and block
    for (int o = threadIdx.x; o < nch * E; o += kThreads) {
        const int ch = o / E, i = o % E;
        unsigned long long m = 0ull;
        for (int r = 0; r < RT; ++r) {
            const unsigned long long v = red[(r * CTb + ch) * E + i];
            m = v > m ? v : m;
        }
        if (m) atomicMax(&keys[b * C + (long)(c_chunk0 + ch) * E + i], m);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void scene_max_decode_kernel(const unsigned long long *__restrict__ keys, long total,
                                                                     T *__restrict__ out, int *__restrict__ arg)
{
    const long e = (long)blockIdx.x * kThreads + threadIdx.x;
    if (e >= total) return;
    const unsigned long long k = keys[e];
    const float v = unsortable((unsigned)(k >> 32));
    if (RowVec<T>::E == 8)   // one element in the row type (exact: the key holds a value that was read from a row)
        reinterpret_cast<pcb_bf16 *>(out)[e] = pcb_f2bf(v);
    else
        reinterpret_cast<float *>(out)[e] = v;
    arg[e] = (int)(~(unsigned)(k & 0xffffffffull));
}

// dz[b*N + n, c] = (n == arg[b, c]) ? g[b, c] : 0, one lane per 16-byte chunk of a row
template <typename T>
__global__ __launch_bounds__(kThreads) void scene_max_bwd_kernel(const T *__restrict__ g, const int *__restrict__ arg,
                                                                  int N, int C, uint4 *__restrict__ dz, long nvec)
{
    constexpr int E = RowVec<T>::E;
    const int CT = C / E;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const int cc = (int)(e % CT);
        const long row = e / CT;
        const long b = row / N;
        const int n = (int)(row - b * N);
        float f[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const long o = b * C + (long)cc * E + i;
            f[i] = arg[o] == n ? RowVec<T>::one(g + o) : 0.0f;
        }
        dz[e] = RowVec<T>::pack(f);
    }
}

// out[b*N + n, :] = [a[b*N + n, 0:C1] | g[b, 0:C2]]: a per-scene vector broadcast to the scene's rows and
// joined to per-point rows -- DGCNN's  torch.cat((x, local_features), dim=1)  with x the max-pooled
// feature expanded over the points (models/DGCNN.py:160-164).  One lane per 16-byte chunk.
template <typename T>
__global__ __launch_bounds__(kThreads) void scene_concat_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ g,
                                                                 int N, int C1, int C2, uint4 *__restrict__ out, long nvec)
{
    constexpr int E = RowVec<T>::E;
    const int T1 = C1 / E, T2 = C2 / E, TT = T1 + T2;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < nvec; e += (long)gridDim.x * kThreads) {
        const int cc = (int)(e % TT);
        const long row = e / TT;
        out[e] = cc < T1 ? a[row * T1 + cc] : g[(row / N) * T2 + (cc - T1)];
    }
}

// Backward of the broadcast: dg[b, c] = sum_n d[b*N + n, col0 + c] over the scene's rows, d [B*N, ld].
// Two stages, no atomics, no memset: grid = (row splits, column blocks, B) writes fp32 partials
// [splits][B][C]; the second kernel adds the splits in order and stores the row type.
template <typename T>
__global__ __launch_bounds__(kThreads) void scene_colsum_kernel(const T *__restrict__ d, int N, int ld, int col0, int C,
                                                                 int CTb, float *__restrict__ part)
{
    constexpr int E = RowVec<T>::E;
    __shared__ float red[kThreads * E];
    const int CT = C / E;
    const int c_chunk0 = blockIdx.y * CTb;
    const int nch = CT - c_chunk0 < CTb ? CT - c_chunk0 : CTb;
    const int RT = kThreads / CTb;
    const int cc = threadIdx.x % CTb, rl = threadIdx.x / CTb;
    const long b = blockIdx.z;
    float s[E];
#pragma unroll
    for (int i = 0; i < E; ++i) s[i] = 0.0f;
    if (cc < nch && rl < RT) {
        const int per = (N + gridDim.x - 1) / gridDim.x;
        const int n0 = blockIdx.x * per, n1 = n0 + per < N ? n0 + per : N;
        for (int n = n0 + rl; n < n1; n += RT) {
            float f[E];
            RowVec<T>::unpack(*reinterpret_cast<const uint4 *>(d + (b * N + n) * (long)ld + col0 + (long)(c_chunk0 + cc) * E), f);
#pragma unroll
            for (int i = 0; i < E; ++i) s[i] += f[i];
        }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) red[threadIdx.x * E + i] = s[i];
    __syncthreads();
    for (int o = threadIdx.x; o < nch * E; o += kThreads) {
        const int ch = o / E, i = o % E;
        float t = 0.0f;
        for (int r = 0; r < RT; ++r) t += red[(r * CTb + ch) * E + i];
        part[((long)blockIdx.x * gridDim.z + b) * C + (long)(c_chunk0 + ch) * E + i] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void scene_colsum_final_kernel(const float *__restrict__ part, int splits, long total,
                                                                       T *__restrict__ out)
{
    const long e = (long)blockIdx.x * kThreads + threadIdx.x;
    if (e >= total) return;
    float t = 0.0f;
    for (int s = 0; s < splits; ++s) t += part[(long)s * total + e];
    if (RowVec<T>::E == 8)
        reinterpret_cast<pcb_bf16 *>(out)[e] = pcb_f2bf(t);
    else
        reinterpret_cast<float *>(out)[e] = t;
}

inline int colsum_splits(int B, int N, int C, int E)
{
    const int CT = C / E;
    const int CTb = CT < 32 ? CT : 32;
    const int RT = kThreads / CTb;
    int splits = (N + 8 * RT - 1) / (8 * RT);
    const int want = 1024 / (((CT + CTb - 1) / CTb) * B);
    if (splits > want) splits = want;
    return splits < 1 ? 1 : splits;
}

template <typename T>
int scene_concat(const void *a, const void *g, int B, int N, int C1, int C2, void *out, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!a || !g || !out || B <= 0 || N <= 0 || C1 <= 0 || C2 <= 0) return PCB_ERR_INVALID_ARG;
    if ((C1 % E) || (C2 % E)) return PCB_ERR_UNSUPPORTED;
    const long nvec = (long)B * N * ((C1 + C2) / E);
    long blocks = (nvec + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scene_concat_kernel<T>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint4 *)a, (const uint4 *)g, N, C1, C2, (uint4 *)out, nvec);
    pcb_account(2.0 * sizeof(T) * (double)B * N * (C1 + C2));
    return pcb_check_launch();
}

template <typename T>
int scene_colsum(const void *d, int B, int N, int ld, int col0, int C, void *out, float *workspace, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!d || !out || !workspace || B <= 0 || N <= 0 || C <= 0 || col0 < 0 || col0 + C > ld) return PCB_ERR_INVALID_ARG;
    if ((C % E) || (ld % E) || (col0 % E)) return PCB_ERR_UNSUPPORTED;
    const int CT = C / E;
    const int CTb = CT < 32 ? CT : 32;
    const int splits = colsum_splits(B, N, C, E);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(scene_colsum_kernel<T>, dim3(splits, (CT + CTb - 1) / CTb, B), dim3(kThreads), 0, st, (const T *)d, N,
                       ld, col0, C, CTb, workspace);
    const long total = (long)B * C;
    hipLaunchKernelGGL(scene_colsum_final_kernel<T>, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                       workspace, splits, total, (T *)out);
    pcb_account((double)sizeof(T) * B * N * C);
    return pcb_check_launch();
}

template <typename T>
int scene_max(const void *rows, int B, int N, int C, void *out, int *arg, void *workspace, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!rows || !out || !arg || !workspace || B <= 0 || N <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    if (C % E) return PCB_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *keys = (unsigned long long *)workspace;
    if (pcb_zero_async(keys, sizeof(unsigned long long) * (size_t)B * C, st) != PCB_OK) return PCB_ERR_LAUNCH;
    const int CT = C / E;
    const int CTb = CT < 32 ? CT : 32;                    // column chunks per block
    const int RT = kThreads / CTb;
    int splits = (N + 8 * RT - 1) / (8 * RT);             // at least 8 rows per row-lane
    const int want = 1024 / (((CT + CTb - 1) / CTb) * B);  // about 1024 blocks in all
    if (splits > want) splits = want;
    if (splits < 1) splits = 1;
    hipLaunchKernelGGL(scene_max_kernel<T>, dim3(splits, (CT + CTb - 1) / CTb, B), dim3(kThreads), 0, st,
                       (const uint4 *)rows, N, C, CTb, keys);
    const long total = (long)B * C;
    hipLaunchKernelGGL(scene_max_decode_kernel<T>, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                       keys, total, (T *)out, arg);
    pcb_account((double)sizeof(T) * B * N * C);
    return pcb_check_launch();
}

template <typename T>
int scene_max_bwd(const void *g, const int *arg, int B, int N, int C, void *dz, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (!g || !arg || !dz || B <= 0 || N <= 0 || C <= 0) return PCB_ERR_INVALID_ARG;
    if (C % E) return PCB_ERR_UNSUPPORTED;
    const long nvec = (long)B * N * (C / E);
    long blocks = (nvec + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scene_max_bwd_kernel<T>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, (const T *)g,
                       arg, N, C, (uint4 *)dz, nvec);
    pcb_account((double)sizeof(T) * B * N * C);
    return pcb_check_launch();
}


// ---- concatenation of feature levels with nearest repetition -----------------------------------
// MultiScaleFeatureFusion (models/model.py:150-170): F.interpolate(level, size=N) of the coarser
// decoder levels followed by torch.cat along channels.  With S_l | N and N / S_l a power of two the
// nearest source of fine row i is coarse row i / r_l (models/containers.py), so
//     out[i, col_l + c] = src_l[i / r_l, c]
// in ONE pass over the output (ATen: one strided copy per level), and backward
//     dsrc_l[s, c] = sum_{j < r_l} g[s * r_l + j, col_l + c]     (fp32 sums, rounded once)
// in one pass over the gradient (ATen: a strided slice made contiguous per level plus one reduction
// kernel per repeated level).
constexpr int kMaxLevels = 4;
struct RepeatLevels {
    const void *src[kMaxLevels];
    void *dst[kMaxLevels];      // backward outputs
    int rep[kMaxLevels];        // r_l >= 1
    int chunk0[kMaxLevels + 1]; // first 16-byte chunk of level l in an output row; [n] = chunks per row
    int n;
};

template <typename T>
__global__ __launch_bounds__(kThreads) void repeat_concat_kernel(RepeatLevels lv, long rows, uint4 *__restrict__ out)
{
    const int CT = lv.chunk0[lv.n];
    const long total = rows * CT;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
        const long i = e / CT;
        const int cc = (int)(e - i * CT);
        int l = 0;
#pragma unroll
        for (int k = 1; k < kMaxLevels; ++k) l += (k < lv.n && cc >= lv.chunk0[k]) ? 1 : 0;
        const int w = lv.chunk0[l + 1] - lv.chunk0[l];
        out[e] = reinterpret_cast<const uint4 *>(lv.src[l])[(i / lv.rep[l]) * w + (cc - lv.chunk0[l])];
    }
}

// one thread per (coarse row, chunk) of one level (blockIdx.y = level)
template <typename T>
__global__ __launch_bounds__(kThreads) void repeat_concat_bwd_kernel(RepeatLevels lv, long rows, const uint4 *__restrict__ g)
{
    constexpr int E = RowVec<T>::E;
    const int l = blockIdx.y;
    const int CT = lv.chunk0[lv.n];
    const int w = lv.chunk0[l + 1] - lv.chunk0[l], r = lv.rep[l];
    const long total = rows / r * w;
    uint4 *const dst = reinterpret_cast<uint4 *>(lv.dst[l]);
    if (!dst) return;
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
        const long s = e / w;
        const int cc = (int)(e - s * w);
        const uint4 *src = g + (s * r) * CT + lv.chunk0[l] + cc;
        if (r == 1) {
            dst[e] = *src;
            continue;
        }
        float acc[E];
#pragma unroll
        for (int i = 0; i < E; ++i) acc[i] = 0.0f;
        for (int j = 0; j < r; ++j) {
            float f[E];
            RowVec<T>::unpack(src[(long)j * CT], f);
#pragma unroll
            for (int i = 0; i < E; ++i) acc[i] += f[i];
        }
        dst[e] = RowVec<T>::pack(acc);
    }
}

template <typename T>
int repeat_concat(int n, const void *const *src, void *const *dst, const int *rep, const int *width, long rows, void *io,
                  bool backward, void *stream)
{
    constexpr int E = RowVec<T>::E;
    if (n < 1 || n > kMaxLevels || !src == !dst || !rep || !width || rows <= 0 || !io) return PCB_ERR_INVALID_ARG;
    RepeatLevels lv;
    lv.n = n;
    int col = 0, most = 0;
    for (int l = 0; l < kMaxLevels; ++l) {
        lv.src[l] = nullptr;
        lv.dst[l] = nullptr;
        lv.rep[l] = 1;
    }
    for (int l = 0; l < n; ++l) {
        if (width[l] <= 0 || width[l] % E || rep[l] < 1 || rows % rep[l]) return PCB_ERR_UNSUPPORTED;
        if (!backward && !src[l]) return PCB_ERR_INVALID_ARG;
        lv.src[l] = backward ? nullptr : src[l];
        lv.dst[l] = backward ? dst[l] : nullptr;
        lv.rep[l] = rep[l];
        lv.chunk0[l] = col;
        col += width[l] / E;
        const long per = rows / rep[l] * (width[l] / E);
        most = per > most ? (int)(per > (1L << 30) ? (1L << 30) : per) : most;
    }
    for (int l = n; l <= kMaxLevels; ++l) lv.chunk0[l] = col;
    hipStream_t st = (hipStream_t)stream;
    if (!backward) {
        const long total = rows * col;
        long blocks = (total + kThreads - 1) / kThreads;
        blocks = blocks > 8192 ? 8192 : blocks;
        hipLaunchKernelGGL(repeat_concat_kernel<T>, dim3((unsigned)blocks), dim3(kThreads), 0, st, lv, rows, (uint4 *)io);
        pcb_account(2.0 * total * 16);
    } else {
        long blocks = ((long)most + kThreads - 1) / kThreads;
        blocks = blocks > 4096 ? 4096 : blocks;
        hipLaunchKernelGGL(repeat_concat_bwd_kernel<T>, dim3((unsigned)blocks, (unsigned)n), dim3(kThreads), 0, st, lv, rows,
                           (const uint4 *)io);
        pcb_account(1.2 * rows * col * 16);
    }
    return pcb_check_launch();
}

}  // namespace

extern "C" {

long pcb_scene_max_workspace(int B, int C) { return (B <= 0 || C <= 0) ? 0 : 8L * B * C; }

long pcb_scene_colsum_workspace(int B, int N, int C)
{
    if (B <= 0 || N <= 0 || C <= 0 || (C & 3)) return 0;
    const int s4 = colsum_splits(B, N, C, 4), s8 = (C & 7) ? 0 : colsum_splits(B, N, C, 8);
    return 4L * (s4 > s8 ? s4 : s8) * B * C;  // enough for either row type
}
int pcb_scene_concat_bf16(const void *a, const void *g, int B, int N, int C1, int C2, void *out, void *stream)
{
    return scene_concat<pcb_bf16>(a, g, B, N, C1, C2, out, stream);
}
int pcb_scene_concat_f32(const void *a, const void *g, int B, int N, int C1, int C2, void *out, void *stream)
{
    return scene_concat<float>(a, g, B, N, C1, C2, out, stream);
}
int pcb_scene_colsum_bf16(const void *d, int B, int N, int ld, int col0, int C, void *out, float *workspace, void *stream)
{
    return scene_colsum<pcb_bf16>(d, B, N, ld, col0, C, out, workspace, stream);
}
int pcb_scene_colsum_f32(const void *d, int B, int N, int ld, int col0, int C, void *out, float *workspace, void *stream)
{
    return scene_colsum<float>(d, B, N, ld, col0, C, out, workspace, stream);
}

int pcb_scene_max_bf16(const void *rows, int B, int N, int C, void *out, int *arg, void *workspace, void *stream)
{
    return scene_max<pcb_bf16>(rows, B, N, C, out, arg, workspace, stream);
}
int pcb_scene_max_f32(const void *rows, int B, int N, int C, void *out, int *arg, void *workspace, void *stream)
{
    return scene_max<float>(rows, B, N, C, out, arg, workspace, stream);
}
int pcb_scene_max_bwd_bf16(const void *g, const int *arg, int B, int N, int C, void *dz, void *stream)
{
    return scene_max_bwd<pcb_bf16>(g, arg, B, N, C, dz, stream);
}
int pcb_scene_max_bwd_f32(const void *g, const int *arg, int B, int N, int C, void *dz, void *stream)
{
    return scene_max_bwd<float>(g, arg, B, N, C, dz, stream);
}

int pcb_repeat_concat_bf16(int n, const void *const *src, const int *rep, const int *width, long rows, void *out, void *stream)
{
    return repeat_concat<pcb_bf16>(n, src, nullptr, rep, width, rows, out, false, stream);
}
int pcb_repeat_concat_f32(int n, const void *const *src, const int *rep, const int *width, long rows, void *out, void *stream)
{
    return repeat_concat<float>(n, src, nullptr, rep, width, rows, out, false, stream);
}
int pcb_repeat_concat_bwd_bf16(int n, const void *g, const int *rep, const int *width, long rows, void *const *dsrc, void *stream)
{
    return repeat_concat<pcb_bf16>(n, nullptr, dsrc, rep, width, rows, (void *)g, true, stream);
}
int pcb_repeat_concat_bwd_f32(int n, const void *g, const int *rep, const int *width, long rows, void *const *dsrc, void *stream)
{
    return repeat_concat<float>(n, nullptr, dsrc, rep, width, rows, (void *)g, true, stream);
}

}  // extern "C"
