// Shared device helpers for the gfx950 kernels of libpcb_hip.so.
// Built with -ffp-contract=off: every fused multiply-add is written out (__fmaf_rn), because the
// sample / neighbour indices must be bit-identical to the reference's fp32 arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcb_hip.h"

#define PCB_WAVE 64

// ---- DPP cross-lane moves (full-rate VALU, no LDS traffic) -------------------------------
// row_ror:n rotates within each row of 16 lanes; four of them (8,4,2,1) leave a commutative
// reduction of the row in every lane of that row.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

#define PCB_ROW_ROR(n) (0x120 + (n))

__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, dpp_f<PCB_ROW_ROR(8)>(v));
    v = fmaxf(v, dpp_f<PCB_ROW_ROR(4)>(v));
    v = fmaxf(v, dpp_f<PCB_ROW_ROR(2)>(v));
    v = fmaxf(v, dpp_f<PCB_ROW_ROR(1)>(v));
    return v;
}
__device__ __forceinline__ int row16_min(int v)
{
    v = min(v, dpp_i<PCB_ROW_ROR(8)>(v));
    v = min(v, dpp_i<PCB_ROW_ROR(4)>(v));
    v = min(v, dpp_i<PCB_ROW_ROR(2)>(v));
    v = min(v, dpp_i<PCB_ROW_ROR(1)>(v));
    return v;
}
// Whole-wave (64 lanes) reductions; the result is wave-uniform.
__device__ __forceinline__ float wave_max(float v)
{
    v = row16_max(v);
    float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ int wave_min(int v)
{
    v = row16_min(v);
    int a = __builtin_amdgcn_readlane(v, 0);
    int b = __builtin_amdgcn_readlane(v, 16);
    int c = __builtin_amdgcn_readlane(v, 32);
    int d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

// ---- the reference's fp32 distance arithmetic --------------------------------------------
// |p|^2 = (x*x + y*y) + z*z, every step rounded (torch.sum(p ** 2, -1))
__device__ __forceinline__ float sq_norm3(float x, float y, float z)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}
// square_distance, models/pointnet2_utils.py:11-13, for one (src, dst) pair:
// dot = fma(s2,t2, fma(s1,t1, s0*t0)); d = ((-2*dot) + |s|^2) + |t|^2.
// -2*dot is exact, so fma(-2, dot, |s|^2) rounds exactly like the reference's two steps.
__device__ __forceinline__ float sqdist_expand(float sx, float sy, float sz, float s2, float tx,
                                               float ty, float tz, float t2)
{
    float dot = __fmaf_rn(sz, tz, __fmaf_rn(sy, ty, __fmul_rn(sx, tx)));
    return __fadd_rn(__fmaf_rn(-2.0f, dot, s2), t2);
}

__device__ __forceinline__ int clamp_index(int64_t j, int n)
{
    return (int)(j < 0 ? 0 : (j > (int64_t)(n - 1) ? (int64_t)(n - 1) : j));
}

// roofline timer hooks (stack.hip): no-ops unless pcb_timer_start armed the timer
void pcb_timer_begin(hipStream_t st, hipEvent_t *stop);                 // category 0: the gemm_nt family
void pcb_timer_begin_cat(hipStream_t st, hipEvent_t *stop, int cat);    // 1: farthest point sampling
void pcb_timer_end(hipStream_t st, hipEvent_t stop, double bytes, int pro, long R, int N, int K);  // also accounts `bytes`
// algorithmic HBM bytes of a launch that is not event-timed (the whole-step figure of bench.py)
void pcb_account(double bytes);

// weight-gradient slab reductions of a whole stack in one launch (gemm.hip)
void pcb_defer_reduces_begin();
int pcb_defer_reduces_flush(hipStream_t st);

// Zero-fill / device copy as KERNELS (gemm_shared.hip), never hipMemsetAsync / hipMemcpyAsync: a step
// of this library must stay correct when it is captured into a hipGraph and replayed, and memset nodes
// do not survive replay on this stack (captured steps that contained one -- this library's or ATen's
// semaphore memset inside its two-stage reductions -- returned garbage on every replay but the first
// once anything ran between replays; tools/graph_reduce_repro.py).  bytes must be a multiple of 4.
int pcb_zero_async(void *ptr, size_t bytes, hipStream_t st);
int pcb_copy_async(void *dst, const void *src, size_t bytes, hipStream_t st);
int pcb_zero2_async(void *a, size_t bytes_a, void *b, size_t bytes_b, hipStream_t st);  // b may be NULL

static inline int pcb_check_launch()
{
    return hipGetLastError() == hipSuccess ? PCB_OK : PCB_ERR_LAUNCH;
}
