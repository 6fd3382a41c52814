// Farthest point sampling for gfx950.
//
// Replaces farthest_point_sample, models/pointnet2_utils.py:63-80 of the reference (a Python loop
// of S iterations, ~10 tiny ATen kernels each).  FPS is a latency chain: S dependent arg-max steps
// per scene, no HBM traffic after the first read.  One workgroup owns one scene and keeps the whole
// cloud plus the running minimum distance in VGPRs (T threads x P points); one iteration is
//   scan P points per lane -> DPP reduction in the wave -> one LDS slot per wave -> ONE barrier
//   -> every wave reduces the <=16 slots redundantly (no second barrier; slots are double-buffered).
// Arithmetic is the reference's: d = (dx*dx + dy*dy) + dz*dz, strict "<" min-update, first index
// of the maximum.
#include "pcb_common.h"

namespace {

constexpr int kPad = 0x7fffffff;

// T threads, P points per lane held in registers (N <= T*P).
template <int T, int P>
__global__ __launch_bounds__(T) void fps_regs_kernel(const float *__restrict__ xyz, int N, int S,
                                                      const int64_t *__restrict__ start,
                                                      int64_t *__restrict__ out)
{
    constexpr int NW = T / PCB_WAVE;
    __shared__ float s_val[2][NW > 1 ? NW : 1];
    __shared__ int s_idx[2][NW > 1 ? NW : 1];

    const int b = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const float *__restrict__ p = xyz + (size_t)b * N * 3;
    int64_t *__restrict__ o = out + (size_t)b * S;

    float px[P], py[P], pz[P], run[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int i = q * T + t;
        const int ii = i < N ? i : N - 1;  // unconditional loads so the P reads pipeline
        px[q] = p[ii * 3 + 0];
        py[q] = p[ii * 3 + 1];
        pz[q] = p[ii * 3 + 2];
        // 1e10: pointnet2_utils.py:68.  Padding slot: below every real distance, never selected.
        run[q] = i < N ? 1e10f : -1.0f;
    }

    int far = (int)start[b];
    far = far < 0 ? 0 : (far >= N ? N - 1 : far);
    for (int s = 0; s < S; ++s) {
        if (t == 0) o[s] = (int64_t)far;
        far = __builtin_amdgcn_readfirstlane(far);
        const float cx = p[far * 3 + 0];
        const float cy = p[far * 3 + 1];
        const float cz = p[far * 3 + 2];

        float best = -2.0f;
        // two points per instruction: packed fp32 subtract / multiply / add round each component
        // exactly like the scalar operations (and -ffp-contract=off keeps them unfused)
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
#pragma unroll
        for (int q = 0; q + 1 < P; q += 2) {
            const f2 dx = f2{px[q], px[q + 1]} - cx2;
            const f2 dy = f2{py[q], py[q + 1]} - cy2;
            const f2 dz = f2{pz[q], pz[q + 1]} - cz2;
            const f2 d = (dx * dx + dy * dy) + dz * dz;
            // d >= 0 and never NaN here, so the hardware minimum IS the reference's `d < run ? d : run`
            run[q] = __builtin_fminf(d.x, run[q]);
            run[q + 1] = __builtin_fminf(d.y, run[q + 1]);
            best = __builtin_fmaxf(best, __builtin_fmaxf(run[q], run[q + 1]));
        }
        if (P & 1) {
            const int q = P - 1;
            const float dx = __fsub_rn(px[q], cx);
            const float dy = __fsub_rn(py[q], cy);
            const float dz = __fsub_rn(pz[q], cz);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            run[q] = __builtin_fminf(d, run[q]);
            best = __builtin_fmaxf(best, run[q]);
        }
        // wave: maximum value first, THEN the lowest index that attains it -- the per-point
        // (compare, select value, select index) of a running arg-max becomes one max per point plus
        // one compare/select per point against the wave's maximum
        const float wmax = wave_max(best);
        int besti = kPad;
#pragma unroll
        for (int q = P - 1; q >= 0; --q) besti = run[q] == wmax ? q * T + t : besti;  // lowest q wins
        const int wcand = wave_min(besti);
        if (NW == 1) {
            far = wcand;
        } else {
            const int buf = s & 1;
            if (lane == 0) {
                s_val[buf][wave] = wmax;
                s_idx[buf][wave] = wcand;
            }
            __syncthreads();
            // NW <= 16 slots: lane l reads slot l % NW, one row of 16 lanes covers them all
            const float v = s_val[buf][lane % NW];
            const int vi = s_idx[buf][lane % NW];
            const float m = row16_max(v);
            far = row16_min(v == m ? vi : kPad);
        }
    }
}

// Larger clouds: running distances live in LDS (40960 floats = 160 KiB), coordinates are re-read
// from L2 every iteration.  Slower per iteration; exists so that N up to 40960 is served.
constexpr int kLdsThreads = 1024;
constexpr int kLdsMaxN = 40960 - 64;

__global__ __launch_bounds__(kLdsThreads) void fps_lds_kernel(const float *__restrict__ xyz, int N,
                                                               int S,
                                                               const int64_t *__restrict__ start,
                                                               int64_t *__restrict__ out)
{
    extern __shared__ float lds[];
    float *run = lds;                          // [N]
    float *s_val = lds + N;                    // [2][16]
    int *s_idx = (int *)(lds + N + 32);        // [2][16]
    const int b = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const float *__restrict__ p = xyz + (size_t)b * N * 3;
    int64_t *__restrict__ o = out + (size_t)b * S;
    for (int i = t; i < N; i += kLdsThreads) run[i] = 1e10f;
    int far = (int)start[b];
    far = far < 0 ? 0 : (far >= N ? N - 1 : far);
    for (int s = 0; s < S; ++s) {
        if (t == 0) o[s] = (int64_t)far;
        far = __builtin_amdgcn_readfirstlane(far);
        const float cx = p[far * 3 + 0], cy = p[far * 3 + 1], cz = p[far * 3 + 2];
        float best = -2.0f;
        int besti = kPad;
        for (int i = t; i < N; i += kLdsThreads) {
            const float dx = __fsub_rn(p[i * 3 + 0], cx);
            const float dy = __fsub_rn(p[i * 3 + 1], cy);
            const float dz = __fsub_rn(p[i * 3 + 2], cz);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            float r = run[i];
            r = d < r ? d : r;
            run[i] = r;
            if (r > best) {
                best = r;
                besti = i;
            }
        }
        const float wmax = wave_max(best);
        const int wcand = wave_min(best == wmax ? besti : kPad);
        const int buf = s & 1;
        if (lane == 0) {
            s_val[buf * 16 + wave] = wmax;
            s_idx[buf * 16 + wave] = wcand;
        }
        __syncthreads();
        const float v = s_val[buf * 16 + (lane & 15)];
        const int vi = s_idx[buf * 16 + (lane & 15)];
        const float m = row16_max(v);
        far = row16_min(v == m ? vi : kPad);
    }
}

template <int T, int P>
void launch_regs(const float *xyz, int B, int N, int S, const int64_t *start, int64_t *out, hipStream_t st)
{
    hipLaunchKernelGGL((fps_regs_kernel<T, P>), dim3(B), dim3(T), 0, st, xyz, N, S, start, out);
}

}  // namespace

extern "C" int pcb_fps(const float *xyz, int B, int N, int S, const int64_t *start_idx,
                       int64_t *out_idx, void *stream)
{
    if (!xyz || !start_idx || !out_idx || B <= 0 || N <= 0 || S <= 0) return PCB_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    // Few waves for small clouds (cheaper cross-wave step), all 16 waves of a CU for large ones.
    if (N <= 64) launch_regs<64, 1>(xyz, B, N, S, start_idx, out_idx, st);
    else if (N <= 128) launch_regs<64, 2>(xyz, B, N, S, start_idx, out_idx, st);
    else if (N <= 256) launch_regs<64, 4>(xyz, B, N, S, start_idx, out_idx, st);
    else if (N <= 512) launch_regs<128, 4>(xyz, B, N, S, start_idx, out_idx, st);
    else if (N <= 1024) launch_regs<256, 4>(xyz, B, N, S, start_idx, out_idx, st);
    else if (N <= 2048) launch_regs<256, 8>(xyz, B, N, S, start_idx, out_idx, st);
    else if (N <= 4096) launch_regs<512, 8>(xyz, B, N, S, start_idx, out_idx, st);
    else if (N <= 8192) launch_regs<1024, 8>(xyz, B, N, S, start_idx, out_idx, st);
    else if (N <= 16384) launch_regs<1024, 16>(xyz, B, N, S, start_idx, out_idx, st);
    else if (N <= kLdsMaxN) {
        const size_t lds = sizeof(float) * (size_t)(N + 64);
        if (hipFuncSetAttribute((const void *)fps_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PCB_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(fps_lds_kernel, dim3(B), dim3(kLdsThreads), lds, st, xyz, N, S, start_idx, out_idx);
    } else {
        return PCB_ERR_UNSUPPORTED;
    }
    return pcb_check_launch();
}
