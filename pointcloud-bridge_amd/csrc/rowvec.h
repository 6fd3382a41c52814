// 16-byte row vectors of the channels-last activation rows, for the kernels that exist in both
// arithmetic modes of the shared-MLP engine: bf16 rows (8 channels per vector; BASELINE config 2)
// and fp32 rows (4 channels per vector; the parity mode whose logits stay within 1e-4 of the
// reference's fp32 Conv/BatchNorm, models/pointnet2_utils.py:149-154).  All math is fp32 in
// registers either way; only the storage type of the rows differs.
#pragma once
#include "pcb_common.h"

typedef unsigned short pcb_bf16;  // raw bits

__device__ __forceinline__ float pcb_bf2f(pcb_bf16 h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ pcb_bf16 pcb_f2bf(float f)
{
    return __builtin_bit_cast(pcb_bf16, (__bf16)f);  // round-to-nearest-even, NaN stays NaN
}

template <typename T>
struct RowVec;

template <>
struct RowVec<pcb_bf16> {
    static constexpr int E = 8;  // elements per 16-byte vector
    static __device__ __forceinline__ void unpack(const uint4 &v, float *f)
    {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint4 pack(const float *f)
    {
        uint4 v;
        v.x = (uint32_t)pcb_f2bf(f[0]) | ((uint32_t)pcb_f2bf(f[1]) << 16);
        v.y = (uint32_t)pcb_f2bf(f[2]) | ((uint32_t)pcb_f2bf(f[3]) << 16);
        v.z = (uint32_t)pcb_f2bf(f[4]) | ((uint32_t)pcb_f2bf(f[5]) << 16);
        v.w = (uint32_t)pcb_f2bf(f[6]) | ((uint32_t)pcb_f2bf(f[7]) << 16);
        return v;
    }
    // the value the next kernel will read back after a store of f
    static __device__ __forceinline__ float stored(float f) { return pcb_bf2f(pcb_f2bf(f)); }
    static __device__ __forceinline__ float one(const pcb_bf16 *p) { return pcb_bf2f(*p); }
};

template <>
struct RowVec<float> {
    static constexpr int E = 4;
    static __device__ __forceinline__ void unpack(const uint4 &v, float *f)
    {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const float *f)
    {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
    static __device__ __forceinline__ float stored(float f) { return f; }
    static __device__ __forceinline__ float one(const float *p) { return *p; }
};

// arg-max bytes of the E channels of one vector ([groups, C] uint8, C % E == 0)
template <int E>
__device__ __forceinline__ unsigned long long load_arg_bytes(const unsigned char *p)
{
    if (E == 8) return *reinterpret_cast<const unsigned long long *>(p);
    return (unsigned long long)*reinterpret_cast<const uint32_t *>(p);
}
template <int E>
__device__ __forceinline__ void store_arg_bytes(unsigned char *p, unsigned long long v)
{
    if (E == 8)
        *reinterpret_cast<unsigned long long *>(p) = v;
    else
        *reinterpret_cast<uint32_t *>(p) = (uint32_t)v;
}
