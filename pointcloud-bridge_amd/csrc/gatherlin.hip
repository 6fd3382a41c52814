// First layer of a grouped shared MLP evaluated per POINT and gathered, instead of per (centroid,
// neighbour) ROW.
//
// The reference builds the grouped tensor and convolves it (models/pointnet2_utils.py:51-58 +
// :149-151, :342-354;  models/DGCNN.py:90-107 + :134-136):
//     SetAbstraction   row (s,j) = [x_j - c_s | f_j],   y = W [x_j - c_s ; f_j]
//     EdgeConv         row (i,j) = [x_j - x_i | x_i],   y = W [x_j - x_i ; x_i]
// A 1x1 convolution is linear, so  y(s,j) = u[idx(s,j)] + v[s]  with per-point products
//     SA:        u = X Wx^T + F Wf^T  (N rows),   v = -C Wx^T        (S rows)
//     EdgeConv:  u = X Wa^T           (N rows),   v = X (Wb - Wa)^T  (N rows)
// i.e. the GEMM shrinks from S*ns rows to N + S rows (ns = 16..32 times fewer) and the grouped
// tensor [S*ns, 3+C] is never written.  u, v are fp32 (the difference of two nearby points must
// not be taken between bf16-rounded products); the caller computes them with ordinary GEMMs.
//
// The coordinate part of a set-abstraction row can also stay exactly as the reference has it,
// Wx (x_j - c_s) with the difference taken in fp32 first: gather_add then takes xyz, the centroids
// and Wx instead of v (u = F Wf^T only), and scatter_dy returns dWx = sum_r dy[r] (x_j - c_s)^T.
//
//   gather_add   y[r,:] = bf16(u[src(r),:] + v[r/ns,:] + Wx (x_j - c_s))  + the BatchNorm
//                statistics of y (slabs); v and the Wx term are optional
//   scatter_dy   backward: dy[r,:] = BatchNorm/activation backward of the layer, built on the fly
//                from (dz | dout+argmax, y) exactly as the GEMM prologues of gemm.hip do;
//                du[src(r),:] += dy[r,:] (fp32 atomics; runs of equal src are combined first: ball
//                query pads short groups by repeating one index), dv[s,:] = sum_j dy (no atomics).
// The gradients of W, X, F, C follow from du, dv through the caller's small GEMMs.
#include "pcb_common.h"

namespace {

typedef unsigned short u16;
constexpr int kThreads = 256;
constexpr int kDwxSlabs = 32;

__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
__device__ __forceinline__ void load8(const float *p, float *f)
{
    const float4 a = *reinterpret_cast<const float4 *>(p);
    const float4 b = *reinterpret_cast<const float4 *>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
    f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
inline float slope_of(int act) { return act == 1 ? 0.0f : (act == 2 ? 0.2f : 1.0f); }

// Lane t owns column chunk t % CT (8 columns) and walks rows t / CT, + RT, ... (CT = C/8 <= 256).
__global__ __launch_bounds__(kThreads) void gather_add_kernel(const float *__restrict__ u,
                                                               const float *__restrict__ v,
                                                               const int64_t *__restrict__ idx, int N, int S,
                                                               int ns, int C, long R, u16 *__restrict__ y,
                                                               float *__restrict__ sums,
                                                               const float *__restrict__ xyz,
                                                               const float *__restrict__ ctr,
                                                               const float *__restrict__ wx, int ldw,
                                                               const float *__restrict__ centre)
{
    __shared__ float red[kThreads * 16];
    const int CT = C >> 3;
    const int RT = kThreads / CT;
    const int cc = threadIdx.x % CT;
    const int rl = threadIdx.x / CT;
    const long per_scene = (long)S * ns;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float w0[8], w1[8], w2[8];  // Wx rows of this lane's 8 columns
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool on = wx != nullptr && rl < RT;
        w0[i] = on ? wx[(long)(cc * 8 + i) * ldw + 0] : 0.0f;
        w1[i] = on ? wx[(long)(cc * 8 + i) * ldw + 1] : 0.0f;
        w2[i] = on ? wx[(long)(cc * 8 + i) * ldw + 2] : 0.0f;
    }
    float ce[8];   // rows stored centred (see gemm.hip RedArgs::centre); NULL: no centring
#pragma unroll
    for (int i = 0; i < 8; ++i) ce[i] = (centre != nullptr && rl < RT) ? centre[cc * 8 + i] : 0.0f;
    if (rl < RT) {
        for (long r = (long)blockIdx.x * RT + rl; r < R; r += (long)gridDim.x * RT) {
            const long b = r / per_scene;
            const long src = b * N + clamp_index(idx[r], N);
            float a[8], c[8];
            load8(u + src * C + cc * 8, a);
            if (v) {
                load8(v + (r / ns) * C + cc * 8, c);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = 0.0f;
            }
            if (wx) {
                const float *pj = xyz + src * 3, *pc = ctr + (r / ns) * 3;
                const float d0 = pj[0] - pc[0], d1 = pj[1] - pc[1], d2 = pj[2] - pc[2];  // as reference :56
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] += fmaf(w2[i], d2, fmaf(w1[i], d1, w0[i] * d0));
            }
            uint4 o;
            uint32_t w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u16 lo = f2bf((a[2 * i] + c[2 * i]) - ce[2 * i]), hi = f2bf((a[2 * i + 1] + c[2 * i + 1]) - ce[2 * i + 1]);
                w[i] = (uint32_t)lo | ((uint32_t)hi << 16);
                const float f0 = bf2f(lo), f1 = bf2f(hi);  // statistics of the values the next kernels read
                s[2 * i] += f0;
                s[2 * i + 1] += f1;
                q[2 * i] = fmaf(f0, f0, q[2 * i]);
                q[2 * i + 1] = fmaf(f1, f1, q[2 * i + 1]);
            }
            o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
            *reinterpret_cast<uint4 *>(y + r * C + cc * 8) = o;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[threadIdx.x * 16 + i] = s[i];
        red[threadIdx.x * 16 + 8 + i] = q[i];
    }
    __syncthreads();
    // the workgroup's totals go to ITS slab (no atomics; pcb_bn_finalize adds the slabs in order)
    for (int o = threadIdx.x; o < 2 * C; o += kThreads) {
        const int m = o / C, c = o % C;
        float a = 0.0f;
        for (int r = 0; r < RT; ++r) a += red[(r * CT + (c >> 3)) * 16 + m * 8 + (c & 7)];
        sums[((long)blockIdx.x * 2 + m) * C + c] = a;
    }
}

// One lane per (centroid g, column c), column fastest: a wave's loads cover whole row segments and
// -- what matters -- its atomics hit consecutive addresses (the same kernel with 8 columns per lane,
// i.e. lanes 32 bytes apart, ran ten times slower: float atomics are resolved per cache line).
//   POOLED = 0: dz rows (bf16) given;  POOLED = 1: dout [G,C] fp32 + argmax [G,C] uint8 of a layer
//   max-pooled over the same ns rows.
template <int POOLED>
__global__ __launch_bounds__(kThreads) void scatter_dy_kernel(
    const u16 *__restrict__ dz, const u16 *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ p, const float *__restrict__ q,
    const float *__restrict__ dout, const unsigned char *__restrict__ arg, float slope,
    const int64_t *__restrict__ idx, int N, int S, int ns, int C, long G, float *__restrict__ du,
    float *__restrict__ dv, const float *__restrict__ xyz, const float *__restrict__ ctr,
    float *__restrict__ dwx, int det)
{
    // det != 0 (reproducible mode, see segsum.hip): no atomics at all -- du is left to pcb_scatter_dy_csr_bf16 (du may be
    // NULL), and every workgroup adds its dWx partials into a slab of its OWN (dwx = [1 + gridDim.x][C][3], zeroed by the
    // caller), summed in slab order afterwards.
    // the grid stride is a multiple of C (see the launcher): a lane keeps its column for good
    __shared__ float red[kThreads * 3];
    const long total = G * C;
    const long first = (long)blockIdx.x * kThreads + threadIdx.x;
    const int c = (int)(first % C);
    const float sc = scale[c], sh = shift[c], pp = p[c], qq = q[c];
    float wacc0 = 0.0f, wacc1 = 0.0f, wacc2 = 0.0f;  // dWx[c][0..2]
    for (long e = first; e < total; e += (long)gridDim.x * kThreads) {
        const long g = e / C;
        const long b = g / S;
        float d = 0.0f;
        int am = -1;
        if (POOLED) {
            d = dout[e];
            am = arg[e];
        }
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        if (dwx) {
            c0 = ctr[g * 3 + 0];
            c1 = ctr[g * 3 + 1];
            c2 = ctr[g * 3 + 2];
        }
        float acc = 0.0f;   // dv: the whole group
        float run = 0.0f;   // du: the current run of equal source indices
        long run_src = -1;
        const long r0 = g * ns;
        for (int j = 0; j < ns; ++j) {
            const long r = r0 + j;
            const long src = b * N + clamp_index(idx[r], N);
            const float yv = bf2f(y[r * C + c]);
            const float f = POOLED ? (j == am ? d : 0.0f) : bf2f(dz[r * C + c]);
            if (src != run_src) {  // uniform over the lanes of a group
                if (run_src >= 0 && !det) atomicAdd(&du[run_src * C + c], run);
                run = 0.0f;
                run_src = src;
            }
            const float g1 = f * (fmaf(yv, sc, sh) > 0.0f ? 1.0f : slope);
            const float dy = fmaf(sc, g1, fmaf(pp, yv, qq));
            run += dy;
            acc += dy;
            if (dwx) {
                wacc0 = fmaf(dy, xyz[src * 3 + 0] - c0, wacc0);
                wacc1 = fmaf(dy, xyz[src * 3 + 1] - c1, wacc1);
                wacc2 = fmaf(dy, xyz[src * 3 + 2] - c2, wacc2);
            }
        }
        if (run_src >= 0 && !det) atomicAdd(&du[run_src * C + c], run);
        if (dv) dv[e] = acc;
    }
    if (dwx) {
        // lanes of the block holding the same column meet in LDS; one atomic per column and block
        red[threadIdx.x * 3 + 0] = wacc0;
        red[threadIdx.x * 3 + 1] = wacc1;
        red[threadIdx.x * 3 + 2] = wacc2;
        __syncthreads();
        if (threadIdx.x < C && first < total) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
            for (int t = threadIdx.x; t < kThreads; t += C) {
                if ((long)blockIdx.x * kThreads + t < total) {
                    a0 += red[t * 3 + 0];
                    a1 += red[t * 3 + 1];
                    a2 += red[t * 3 + 2];
                }
            }
            // thousands of workgroups adding into 3C addresses would queue up on them: the adds are
            // spread over kDwxSlabs copies (slabs 1..), summed into slab 0 by sum_slabs_kernel
            if (det) {
                // (one thread per column and workgroup reaches this point: plain stores into the workgroup's own slab)
                float *own = dwx + (long)(1 + blockIdx.x) * C * 3;
                own[c * 3 + 0] = a0;
                own[c * 3 + 1] = a1;
                own[c * 3 + 2] = a2;
            } else {
                float *slab = dwx + (long)(1 + blockIdx.x % kDwxSlabs) * C * 3;
                atomicAdd(&slab[c * 3 + 0], a0);
                atomicAdd(&slab[c * 3 + 1], a1);
                atomicAdd(&slab[c * 3 + 2], a2);
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void sum_slabs_kernel(float *__restrict__ dwx, int elems, int nslabs)
{
    const int e = blockIdx.x * kThreads + threadIdx.x;
    if (e >= elems) return;
    float a = 0.0f;
    for (int k = 1; k <= nslabs; ++k) a += dwx[(long)k * elems + e];
    dwx[e] = a;
}

// grid of scatter_dy_kernel: grid * 256 must be a multiple of C (a lane keeps its column): multiples of C / gcd(C, 256)
inline long scatter_dy_grid(long G, int C)
{
    int gcd = C, t = kThreads;
    while (t) { const int r = gcd % t; gcd = t; t = r; }
    const long m = C / gcd;
    long blocks = (G * C + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    return (blocks + m - 1) / m * m;
}

inline long gather_add_grid(long R, int C)
{
    const int RT = kThreads / (C >> 3);
    long blocks = (R + (long)RT * 8 - 1) / ((long)RT * 8);
    return blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
}

inline bool bad_c(int C) { return C <= 0 || (C & 7) != 0 || C > 2048; }

}  // namespace

extern "C" int pcb_gather_add_partials(long R, int C)
{
    if (R <= 0 || bad_c(C)) return 0;
    return (int)gather_add_grid(R, C);
}

extern "C" long pcb_scatter_dy_slabs(int B, int S, int C, int det)
{
    if (B <= 0 || S <= 0 || bad_c(C)) return 0;
    return 1 + (det ? scatter_dy_grid((long)B * S, C) : kDwxSlabs);
}

extern "C" int pcb_gather_add_bf16(const float *u, const float *v, const int64_t *idx, int B, int N, int S,
                                   int ns, int C, const float *xyz, const float *ctr, const float *wx, int ldw,
                                   void *y, float *sums, int nparts, const float *centre, void *stream)
{
    if (!u || !idx || !y || !sums || B <= 0 || N <= 0 || S <= 0 || ns <= 0) return PCB_ERR_INVALID_ARG;
    if (nparts < 1 || nparts > 1024) return PCB_ERR_INVALID_ARG;  // the caller's slab count IS the grid
    if (wx && (!xyz || !ctr || ldw < 3)) return PCB_ERR_INVALID_ARG;
    if (bad_c(C)) return PCB_ERR_UNSUPPORTED;
    const long R = (long)B * S * ns;
    hipLaunchKernelGGL(gather_add_kernel, dim3((unsigned)nparts), dim3(kThreads), 0,
                       (hipStream_t)stream, u, v, idx, N, S, ns, C, R, (u16 *)y, sums, xyz, ctr, wx, ldw, centre);
    pcb_account(6.0 * R * C + 8.0 * R + (v ? 4.0 * B * S * C : 0.0));
    return pcb_check_launch();
}

extern "C" int pcb_scatter_dy_bf16(int pooled, const void *dz, const void *y, const float *scale,
                                   const float *shift, const float *p, const float *q, const float *dout,
                                   const unsigned char *argmax, int act, const int64_t *idx, int B, int N,
                                   int S, int ns, int C, const float *xyz, const float *ctr, float *du, float *dv,
                                   float *dwx, int det, void *stream)
{
    if (!y || !scale || !shift || !p || !q || !idx || (!du && !det) || B <= 0 || N <= 0 || S <= 0 || ns <= 0)
        return PCB_ERR_INVALID_ARG;
    if (pooled ? (!dout || !argmax || ns > 255) : !dz) return PCB_ERR_INVALID_ARG;
    if (dwx && (!xyz || !ctr)) return PCB_ERR_INVALID_ARG;
    if (bad_c(C)) return PCB_ERR_UNSUPPORTED;
    const long G = (long)B * S;
    const long blocks = scatter_dy_grid(G, C);
    hipStream_t st = (hipStream_t)stream;
    if (pooled)
        hipLaunchKernelGGL(scatter_dy_kernel<1>, dim3((unsigned)blocks), dim3(kThreads), 0, st, (const u16 *)dz,
                           (const u16 *)y, scale, shift, p, q, dout, argmax, slope_of(act), idx, N, S, ns, C, G, du, dv, xyz, ctr, dwx, det);
    else
        hipLaunchKernelGGL(scatter_dy_kernel<0>, dim3((unsigned)blocks), dim3(kThreads), 0, st, (const u16 *)dz,
                           (const u16 *)y, scale, shift, p, q, dout, argmax, slope_of(act), idx, N, S, ns, C, G, du, dv, xyz, ctr, dwx, det);
    if (dwx)
        hipLaunchKernelGGL(sum_slabs_kernel, dim3((3 * C + kThreads - 1) / kThreads), dim3(kThreads), 0, st, dwx, 3 * C,
                           det ? (int)blocks : kDwxSlabs);
    pcb_account(6.0 * (double)G * ns * C + (pooled ? 5.0 * G * C : 2.0 * (double)G * ns * C) + 8.0 * G * ns);
    return pcb_check_launch();
}
