// Ball query for gfx950.
//
// Replaces query_ball_point, models/pointnet2_utils.py:97-112 of the reference, which materialises
// a [B,S,N] fp32 distance matrix and a [B,S,N] int64 index matrix and fully sorts the latter.
// Here nothing of size S*N exists: one wave owns CPW centroids (coordinates in SGPRs), its 64 lanes
// sweep the cloud 64 points at a time, a radius test becomes one v_cmp whose 64-bit result mask
// (ballot) is ranked with mbcnt, and the first `nsample` hits are stored in ascending index order.
// A wave stops as soon as all of its centroids are full.  Distances use the reference's expansion
// formula and operation order (pcb_common.h: sqdist_expand), so the indices are bit-identical.
#include "pcb_common.h"

namespace {

constexpr int kCPW = 4;       // centroids per wave
constexpr int kWaves = 4;     // waves per workgroup

__device__ __forceinline__ int lane_rank(uint64_t mask)
{
    // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// NR = 1: one radius; NR = 2: two radii tested against the same distance (multi-scale grouping).
template <int NR>
__global__ __launch_bounds__(kWaves * PCB_WAVE) void ball_query_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, int N, int S,
    float r2a, int nsa, int64_t *__restrict__ outa, float r2b, int nsb, int64_t *__restrict__ outb)
{
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s0 = (blockIdx.x * kWaves + wave) * kCPW;
    if (s0 >= S) return;  // wave-uniform; the kernel has no barrier

    const float *__restrict__ p = xyz + (size_t)b * N * 3;
    const float *__restrict__ c = new_xyz + (size_t)b * S * 3;

    float cx[kCPW], cy[kCPW], cz[kCPW], c2[kCPW];
    int cnt[NR][kCPW], first[NR][kCPW];
    const float r2[2] = {r2a, r2b};
    const int ns[2] = {nsa, nsb};
    int64_t *const outp[2] = {outa, outb};
#pragma unroll
    for (int k = 0; k < kCPW; ++k) {
        const int s = min(s0 + k, S - 1);
        cx[k] = c[s * 3 + 0];
        cy[k] = c[s * 3 + 1];
        cz[k] = c[s * 3 + 2];
        c2[k] = sq_norm3(cx[k], cy[k], cz[k]);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            cnt[r][k] = (s0 + k < S) ? 0 : ns[r];  // out-of-range centroid: already "full"
            first[r][k] = N;
        }
    }

    for (int base = 0; base < N; base += PCB_WAVE) {
        const int i = base + lane;
        const bool in = i < N;
        const int ii = in ? i : N - 1;
        const float x = p[ii * 3 + 0];
        const float y = p[ii * 3 + 1];
        const float z = p[ii * 3 + 2];
        const float t2 = sq_norm3(x, y, z);
        bool open = false;
#pragma unroll
        for (int k = 0; k < kCPW; ++k) {
            bool need = false;
#pragma unroll
            for (int r = 0; r < NR; ++r) need = need || (cnt[r][k] < ns[r]);
            if (!need) continue;  // wave-uniform
            const float d = sqdist_expand(cx[k], cy[k], cz[k], c2[k], x, y, z, t2);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (cnt[r][k] < ns[r]) {
                    const bool hit = in && !(d > r2[r]);  // reference drops d > r^2 (:105)
                    const uint64_t mask = __ballot(hit);
                    if (mask) {
                        const int pos = cnt[r][k] + lane_rank(mask);
                        if (hit && pos < ns[r])
                            outp[r][((size_t)b * S + s0 + k) * ns[r] + pos] = (int64_t)i;
                        if (cnt[r][k] == 0) first[r][k] = base + __ffsll((unsigned long long)mask) - 1;
                        cnt[r][k] += __popcll(mask);
                    }
                    open = open || (cnt[r][k] < ns[r]);
                }
            }
        }
        if (!open) break;
    }

    // slots the sweep did not fill repeat the first hit (:108-110); no hit at all leaves N
#pragma unroll
    for (int k = 0; k < kCPW; ++k) {
        if (s0 + k >= S) continue;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int have = min(cnt[r][k], ns[r]);
            for (int q = have + lane; q < ns[r]; q += PCB_WAVE)
                outp[r][((size_t)b * S + s0 + k) * ns[r] + q] = (int64_t)first[r][k];
        }
    }
}

}  // namespace

extern "C" int pcb_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, float r2,
                              int nsample, int64_t *out_idx, void *stream)
{
    if (!xyz || !new_xyz || !out_idx || B <= 0 || N <= 0 || S <= 0) return PCB_ERR_INVALID_ARG;
    if (nsample < 1 || nsample > N) return PCB_ERR_INVALID_ARG;
    const dim3 grid((S + kCPW * kWaves - 1) / (kCPW * kWaves), B);
    hipLaunchKernelGGL((ball_query_kernel<1>), grid, dim3(kWaves * PCB_WAVE), 0, (hipStream_t)stream,
                       xyz, new_xyz, N, S, r2, nsample, out_idx, 0.0f, 0, (int64_t *)nullptr);
    pcb_account(12.0 * ((double)N + S) * B + 8.0 * (double)S * nsample * B);
    return pcb_check_launch();
}

extern "C" int pcb_ball_query2(const float *xyz, const float *new_xyz, int B, int N, int S,
                               float r2_a, int nsample_a, int64_t *out_idx_a,
                               float r2_b, int nsample_b, int64_t *out_idx_b, void *stream)
{
    if (!xyz || !new_xyz || !out_idx_a || !out_idx_b || B <= 0 || N <= 0 || S <= 0) return PCB_ERR_INVALID_ARG;
    if (nsample_a < 1 || nsample_a > N || nsample_b < 1 || nsample_b > N) return PCB_ERR_INVALID_ARG;
    const dim3 grid((S + kCPW * kWaves - 1) / (kCPW * kWaves), B);
    hipLaunchKernelGGL((ball_query_kernel<2>), grid, dim3(kWaves * PCB_WAVE), 0, (hipStream_t)stream,
                       xyz, new_xyz, N, S, r2_a, nsample_a, out_idx_a, r2_b, nsample_b, out_idx_b);
    pcb_account(12.0 * ((double)N + S) * B + 8.0 * (double)S * (nsample_a + nsample_b) * B);
    return pcb_check_launch();
}

// square_distance materialised (models/pointnet2_utils.py:7-14).  The operators above never need
// the matrix; this exists because square_distance is a public name of the reference module.
namespace {
__global__ __launch_bounds__(256) void square_distance_kernel(const float *__restrict__ src,
                                                               const float *__restrict__ dst, int N,
                                                               int M, float *__restrict__ out,
                                                               size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % M);
        const size_t bi = e / M;  // b*N + i
        const size_t b = bi / N;
        const float *s = src + bi * 3;
        const float *t = dst + (b * M + j) * 3;
        const float sx = s[0], sy = s[1], sz = s[2], tx = t[0], ty = t[1], tz = t[2];
        out[e] = sqdist_expand(sx, sy, sz, sq_norm3(sx, sy, sz), tx, ty, tz, sq_norm3(tx, ty, tz));
    }
}
}  // namespace

extern "C" int pcb_square_distance(const float *src, const float *dst, int B, int N, int M,
                                   float *out, void *stream)
{
    if (!src || !dst || !out || B <= 0 || N <= 0 || M <= 0) return PCB_ERR_INVALID_ARG;
    const size_t total = (size_t)B * N * M;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(square_distance_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       src, dst, N, M, out, total);
    pcb_account(12.0 * ((double)N + M) * B + 4.0 * (double)N * M * B);
    return pcb_check_launch();
}
