// kNN on 3-D coordinates through a uniform grid, for gfx950.
//
// Same result as pcb_knn with D = 3 (csrc/knn.hip; the reference's DGCNN.knn on xyz, models/DGCNN.py:49-70,
// and the torch.cdist + topk of BridgeStructureEncoding, models/attention_modules.py:584-586): the k
// smallest  pd(i,j) = (|xi|^2 + (-2*<xi,xj>)) + |xj|^2  in the same fp32 operation order, ties by
// lower index, nearest first.  The brute-force kernel evaluates all N^2 pairs (4.3 G at B=16,
// N=16384: 4.1 ms); in three dimensions a query only has to look at the cells around it.
//   build  : one workgroup per scene sorts the cloud by cell (counting sort in LDS, as in fps.hip):
//            sorted (x,y,z,|p|^2), original indices, cell offsets, grid parameters
//   query  : one lane per query, in sorted order (neighbouring lanes share cells -> their loads hit
//            the same lines); the k best so far live in LDS as sortable 64-bit keys
//            (distance bits, index), one unsorted column per lane (sorted once at the end); cells are visited in growing cubes
//            around the query's cell, and the search stops when the k-th best distance is below
//            the distance to the boundary of the cube already searched -- with a margin for the
//            rounding of pd and of the cell assignment, so that no pair the brute-force order
//            would select can be missed.  A cube that covers the whole grid ends the search too.
//   scenes whose points crowd into few cells (an outlier stretching the bounding box) are flagged by
//   the build kernel and left to the brute-force kernel, which checks the same flag.
#include <stdlib.h>

#include "pcb_common.h"

namespace {

constexpr int kBuildThreads = 1024;
constexpr int kMaxPerThread = 16;          // build: N <= 16384
constexpr int kMaxCells = 16384;          // capacity of the cell tables (64 KB histogram in the build kernel's LDS)
constexpr int kMaxCodeBits = 12;           // cells actually used: see pcb_knn_xyz
constexpr int kQueryThreads = 256;

struct GridParams {   // 16 words per scene
    float lo[3], inv[3], h[3], eps;
    int bits[3], crowded, ncells, pad;
};
static_assert(sizeof(GridParams) == 64, "16 words");

__device__ __forceinline__ int cell_coord(float v, float lo, float inv, int n)
{
    return min(n - 1, max(0, (int)(__fmul_rn(__fsub_rn(v, lo), inv))));
}

__device__ __forceinline__ float wave_minf(float v) { return -wave_max(-v); }

__global__ __launch_bounds__(kBuildThreads) void knn_grid_build_kernel(const float *__restrict__ xyz, int N, int code_bits,
                                                                       float4 *__restrict__ sorted,
                                                                       int *__restrict__ oidx,
                                                                       int *__restrict__ cell_start,
                                                                       GridParams *__restrict__ params)
{
    constexpr int T = kBuildThreads, NW = T / PCB_WAVE, P = kMaxPerThread;
    __shared__ int s_hist[kMaxCells];
    __shared__ float s_red[8][NW];
    __shared__ int s_scan[NW];
    __shared__ int s_most[NW];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *__restrict__ p = xyz + (size_t)b * N * 3;
    sorted += (size_t)b * N;
    oidx += (size_t)b * N;
    cell_start += (size_t)b * (kMaxCells + 1);

    float px[P], py[P], pz[P];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, big = 0.0f;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int i = q * T + t, ii = i < N ? i : N - 1;
        px[q] = p[ii * 3 + 0];
        py[q] = p[ii * 3 + 1];
        pz[q] = p[ii * 3 + 2];
        lo[0] = fminf(lo[0], px[q]); hi[0] = fmaxf(hi[0], px[q]);
        lo[1] = fminf(lo[1], py[q]); hi[1] = fmaxf(hi[1], py[q]);
        lo[2] = fminf(lo[2], pz[q]); hi[2] = fmaxf(hi[2], pz[q]);
        big = fmaxf(big, sq_norm3(px[q], py[q], pz[q]));
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_minf(lo[a]), h = wave_max(hi[a]);
        if (lane == 0) {
            s_red[a][wave] = l;
            s_red[3 + a][wave] = h;
        }
    }
    {
        const float m = wave_max(big);
        if (lane == 0) s_red[6][wave] = m;
    }
    for (int e = t; e < kMaxCells; e += T) s_hist[e] = 0;
    __syncthreads();
    // cells: code_bits binary splits, each along the axis whose cells are currently longest
    float inv[3], ext[3];
    int bits[3] = {0, 0, 0};
    {
        float cell[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float l = s_red[a][0], h = s_red[3 + a][0];
            for (int w = 1; w < NW; ++w) {
                l = fminf(l, s_red[a][w]);
                h = fmaxf(h, s_red[3 + a][w]);
            }
            lo[a] = l;
            cell[a] = ext[a] = h - l;
        }
        big = s_red[6][0];
        for (int w = 1; w < NW; ++w) big = fmaxf(big, s_red[6][w]);
        for (int step = 0; step < code_bits; ++step) {
            const int a = (cell[0] >= cell[1] && cell[0] >= cell[2]) ? 0 : (cell[1] >= cell[2] ? 1 : 2);
            bits[0] += a == 0; bits[1] += a == 1; bits[2] += a == 2;
            cell[0] *= a == 0 ? 0.5f : 1.0f;
            cell[1] *= a == 1 ? 0.5f : 1.0f;
            cell[2] *= a == 2 ? 0.5f : 1.0f;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) inv[a] = ext[a] > 0.0f ? (float)(1 << bits[a]) / ext[a] : 0.0f;
    }
    const int nx = 1 << bits[0], ny = 1 << bits[1], nz = 1 << bits[2];
    int slot[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int i = q * T + t;
        const int cx = cell_coord(px[q], lo[0], inv[0], nx), cy = cell_coord(py[q], lo[1], inv[1], ny),
                  cz = cell_coord(pz[q], lo[2], inv[2], nz);
        const int code = (cz * ny + cy) * nx + cx;  // row-major: a run of cells along x is a run of points
        slot[q] = i < N ? (code | (atomicAdd(&s_hist[code], 1) << 14)) : -1;
    }
    __syncthreads();
    {   // exclusive scan of the cell counts (16 consecutive cells per thread) + the largest count
        int c[kMaxCells / T], sum = 0, most = 0;
#pragma unroll
        for (int e = 0; e < kMaxCells / T; ++e) {
            c[e] = s_hist[t * (kMaxCells / T) + e];
            sum += c[e];
            most = max(most, c[e]);
        }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(incl, off, 64);
            if (lane >= off) incl += n;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) most = max(most, __shfl_xor(most, off, 64));
        if (lane == 63) s_scan[wave] = incl;
        if (lane == 0) s_most[wave] = most;
        __syncthreads();
        int base = incl - sum;
        for (int w = 0; w < wave; ++w) base += s_scan[w];
#pragma unroll
        for (int e = 0; e < kMaxCells / T; ++e) {
            s_hist[t * (kMaxCells / T) + e] = base;
            cell_start[t * (kMaxCells / T) + e] = base;
            base += c[e];
        }
        if (t == T - 1) cell_start[kMaxCells] = N;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < P; ++q)
        if (slot[q] >= 0) {
            const int dst = s_hist[slot[q] & (kMaxCells - 1)] + (slot[q] >> 14);
            sorted[dst] = make_float4(px[q], py[q], pz[q], sq_norm3(px[q], py[q], pz[q]));
            oidx[dst] = q * T + t;
        }
    if (t == 0) {
        int most = 0;
        for (int w = 0; w < NW; ++w) most = max(most, s_most[w]);
        GridParams g;
        const int ncells = 1 << code_bits;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            g.lo[a] = lo[a];
            g.inv[a] = inv[a];
            g.h[a] = ext[a] / (float)(1 << bits[a]);
            g.bits[a] = bits[a];
        }
        // rounding of pd = (|q|^2 - 2<q,c>) + |c|^2: a handful of ulps of (|q| + |c|)^2 <= 4 max|p|^2
        g.eps = 4.0e-6f * fmaxf(big, 1e-30f);
        // a scene crowded into few cells would make every lane walk thousands of candidates
        g.crowded = most > 16 * ((N + ncells - 1) / ncells) + 64;
        g.ncells = ncells;
        g.pad = 0;
        params[b] = g;
    }
}

__device__ __forceinline__ unsigned sortable(float v)  // monotonic float -> unsigned
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unsortable(unsigned u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

template <int KMAX>
__global__ __launch_bounds__(kQueryThreads) void knn_grid_query_kernel(const float4 *__restrict__ sorted,
                                                                       const int *__restrict__ oidx,
                                                                       const int *__restrict__ cell_start,
                                                                       const GridParams *__restrict__ params, int N,
                                                                       int k, int64_t *__restrict__ out)
{
    __shared__ unsigned long long lst[KMAX * kQueryThreads];  // [slot][lane]: conflict-free columns
    const int b = blockIdx.y, t = threadIdx.x;
    const GridParams g = params[b];
    if (g.crowded) return;  // the brute-force kernel serves this scene
    const int s = blockIdx.x * kQueryThreads + t;
    if (s >= N) return;     // no barrier below: lanes are independent
    sorted += (size_t)b * N;
    oidx += (size_t)b * N;
    cell_start += (size_t)b * (kMaxCells + 1);
    const int nx = 1 << g.bits[0], ny = 1 << g.bits[1], nz = 1 << g.bits[2];
    const float4 q = sorted[s];
    const int cx = cell_coord(q.x, g.lo[0], g.inv[0], nx), cy = cell_coord(q.y, g.lo[1], g.inv[1], ny),
              cz = cell_coord(q.z, g.lo[2], g.inv[2], nz);
    // The k best so far as an UNSORTED set in this lane's LDS column: a qualifying candidate replaces
    // the current largest member and the column is rescanned for the new largest -- k independent
    // reads instead of a dependent shift chain (a wave runs the replacement whenever ANY of its lanes
    // has a new member, i.e. for most candidates; with a sorted insertion chain the kernel took twice as long).
    // Empty slots hold the largest key, so they are replaced first; slots beyond k hold 0 and never are.
#pragma unroll
    for (int e = 0; e < KMAX; ++e) lst[e * kQueryThreads + t] = e < k ? ~0ull : 0ull;
    unsigned long long kth = ~0ull;  // largest member = the key to beat
    int kpos = 0, filled = 0;        // the first k candidates are simply appended (no rescan until the set is full)

    auto scan = [&](int row, int x0, int x1) {  // cells x0..x1 of grid row `row` = one run of sorted points
        const int from = cell_start[row * nx + x0], to = cell_start[row * nx + x1 + 1];
        for (int j = from; j < to; ++j) {
            const float4 c = sorted[j];
            const float dot = __fmaf_rn(q.z, c.z, __fmaf_rn(q.y, c.y, __fmul_rn(q.x, c.x)));
            const float pd = __fadd_rn(__fmaf_rn(-2.0f, dot, q.w), c.w);
            const unsigned long long key = ((unsigned long long)sortable(pd) << 32) | (unsigned)oidx[j];
            if (key >= kth) continue;
            if (filled < k - 1) {
                lst[filled++ * kQueryThreads + t] = key;
                continue;
            }
            lst[(filled < k ? filled++ : kpos) * kQueryThreads + t] = key;
            unsigned long long m = 0;
#pragma unroll
            for (int e = 0; e < KMAX; ++e) {
                const unsigned long long v = lst[e * kQueryThreads + t];
                if (v > m) {
                    m = v;
                    kpos = e;
                }
            }
            kth = m;
        }
    };
    for (int rad = 0;; ++rad) {
        const int z0 = max(cz - rad, 0), z1 = min(cz + rad, nz - 1);
        const int y0 = max(cy - rad, 0), y1 = min(cy + rad, ny - 1);
        const int x0 = max(cx - rad, 0), x1 = min(cx + rad, nx - 1);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const int row = z * ny + y;
                if (rad == 0 || z == cz - rad || z == cz + rad || y == cy - rad || y == cy + rad) {
                    scan(row, x0, x1);  // a face of the cube (or the centre cell): the whole run
                } else {                // an inner row: only the two new end cells
                    if (cx - rad >= 0) scan(row, cx - rad, cx - rad);
                    if (cx + rad <= nx - 1) scan(row, cx + rad, cx + rad);
                }
            }
        // every point outside the cube searched so far is at least `reach` away along an axis on which
        // the cube does not yet span the grid
        const bool whole = x0 == 0 && x1 == nx - 1 && y0 == 0 && y1 == ny - 1 && z0 == 0 && z1 == nz - 1;
        if (whole) break;
        if ((unsigned)(kth >> 32) != 0xffffffffu) {  // k real members
            float reach = INFINITY;
            if (x0 > 0 || x1 < nx - 1) reach = fminf(reach, g.h[0]);
            if (y0 > 0 || y1 < ny - 1) reach = fminf(reach, g.h[1]);
            if (z0 > 0 || z1 < nz - 1) reach = fminf(reach, g.h[2]);
            // a point just inside the query's own cell face is `rad` whole cells from the outside;
            // 1e-3 of a cell covers the rounding of the cell assignment
            reach *= (float)rad - 1e-3f;
            const float far2 = unsortable((unsigned)(kth >> 32));
            if (reach > 0.0f && far2 + 2.0f * g.eps < reach * reach * (1.0f - 1e-6f)) break;
        }
    }
    // nearest first: insertion sort of the k members (all lanes in step, once per query)
    for (int e = 1; e < k; ++e) {
        const unsigned long long key = lst[e * kQueryThreads + t];
        int f = e;
        while (f > 0) {
            const unsigned long long prev = lst[(f - 1) * kQueryThreads + t];
            if (prev <= key) break;
            lst[f * kQueryThreads + t] = prev;
            --f;
        }
        lst[f * kQueryThreads + t] = key;
    }
    int64_t *__restrict__ o = out + ((size_t)b * N + oidx[s]) * k;
    for (int e = 0; e < k; ++e) o[e] = (int64_t)(unsigned)(lst[e * kQueryThreads + t] & 0xffffffffull);
}

}  // namespace

// bytes of caller-owned scratch for pcb_knn_xyz
extern "C" long pcb_knn_xyz_workspace(int B, int N)
{
    if (B <= 0 || N <= 0) return 0;
    return (long)B * ((long)N * (16 + 4) + (long)(kMaxCells + 1) * 4 + (long)sizeof(GridParams)) + 256;
}

// brute-force kernel of knn.hip with a per-scene switch (stride in ints, NULL = all scenes)
int pcb_knn_flagged(const float *x, int B, int N, int D, int k, float *norms, int64_t *out_idx, const int *only_if,
                    int only_if_stride, hipStream_t st);

extern "C" int pcb_knn_xyz(const float *xyz, int B, int N, int k, float *norms, void *workspace, int64_t *out_idx,
                           void *stream)
{
    if (!xyz || !norms || !out_idx || B <= 0 || N <= 0) return PCB_ERR_INVALID_ARG;
    if (k < 1 || k > 32 || k > N) return PCB_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    // small clouds: the N^2 kernel is already cheap; large ones do not fit the build kernel's registers
    if (N < 1024 || N > kBuildThreads * kMaxPerThread || !workspace)
        return pcb_knn_flagged(xyz, B, N, 3, k, norms, out_idx, nullptr, 0, st);
    char *w = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float4 *sorted = (float4 *)w;
    int *oidx = (int *)(w + (size_t)B * N * 16);
    int *cell_start = oidx + (size_t)B * N;
    GridParams *params = (GridParams *)(cell_start + (size_t)B * (kMaxCells + 1));
    // Cells: about 2 points each, at most 2^12 of them (measured at N = 8192 and 16384, k = 16..32: 4096
    // near-cubic cells beat both coarser grids -- more candidates per query -- and finer ones -- more
    // rows to walk, and 13 split bits leave 2:1 cells on a round cloud).  PCB_KNN_PER_CELL overrides.
    int per_cell = 2;
    if (getenv("PCB_KNN_PER_CELL")) per_cell = atoi(getenv("PCB_KNN_PER_CELL"));
    if (per_cell < 1) per_cell = 1;
    int code_bits = 0;
    while ((1 << (code_bits + 1)) * per_cell <= N && code_bits < kMaxCodeBits) ++code_bits;
    hipLaunchKernelGGL(knn_grid_build_kernel, dim3(B), dim3(kBuildThreads), 0, st, xyz, N, code_bits, sorted, oidx,
                       cell_start, params);
    const dim3 grid((N + kQueryThreads - 1) / kQueryThreads, B);
    if (k <= 16)
        hipLaunchKernelGGL(knn_grid_query_kernel<16>, grid, dim3(kQueryThreads), 0, st, sorted, oidx, cell_start, params, N,
                           k, out_idx);
    else if (k <= 24)  // DGCNN's k = 20: 48 KB of LDS per workgroup instead of 64, a shorter rescan
        hipLaunchKernelGGL(knn_grid_query_kernel<24>, grid, dim3(kQueryThreads), 0, st, sorted, oidx, cell_start, params, N,
                           k, out_idx);
    else
        hipLaunchKernelGGL(knn_grid_query_kernel<32>, grid, dim3(kQueryThreads), 0, st, sorted, oidx, cell_start, params, N,
                           k, out_idx);
    if (pcb_check_launch() != PCB_OK) return PCB_ERR_LAUNCH;
    pcb_account(12.0 * (double)N * B + 8.0 * (double)N * k * B);
    // crowded scenes (flag set by the build kernel): all pairs
    return pcb_knn_flagged(xyz, B, N, 3, k, norms, out_idx, &params[0].crowded, (int)(sizeof(GridParams) / sizeof(int)), st);
}
