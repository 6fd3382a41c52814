// Read-bandwidth microbenchmark for gfx950: how many bytes must a CU keep in flight to stream from
// HBM at a given rate?  Persistent workgroups of 256 threads (G per CU), each thread issues U
// independent 16-byte loads (one "stage": 256*U*16 bytes per workgroup, contiguous), consumes them
// and goes on -- the access pattern of the row GEMMs' operand staging.  Optionally every stage
// also writes W*stage bytes (the GEMM's output stream).
//   hipcc --offload-arch=gfx950 -O3 -o membw tools/micro/membw.hip && ./membw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int U, int PIPE>
__global__ __launch_bounds__(256) void stream_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long stages,
                                                    int wmul, unsigned *sink)
{
    // stage s = 256*U consecutive uint4; PIPE = 1: loads of stage s+1 issued before stage s is consumed
    uint4 cur[U], nxt[U];
    unsigned acc = 0;
    long s = blockIdx.x;
    auto load = [&](uint4 (&r)[U], long st) {
#pragma unroll
        for (int i = 0; i < U; ++i) r[i] = src[(st * U + i) * 256 + threadIdx.x];
    };
    if (s < stages) load(cur, s);
    for (; s < stages; s += gridDim.x) {
        const long n = s + gridDim.x;
        if (PIPE && n < stages) load(nxt, n);
#pragma unroll
        for (int i = 0; i < U; ++i) acc ^= cur[i].x ^ cur[i].y ^ cur[i].z ^ cur[i].w;
        __syncthreads();
        for (int w = 0; w < wmul; ++w)
#pragma unroll
            for (int i = 0; i < U; ++i) dst[((s * wmul + w) * U + i) * 256 + threadIdx.x] = make_uint4(acc, i, w, 0);
        if (PIPE) {
#pragma unroll
            for (int i = 0; i < U; ++i) cur[i] = nxt[i];
        } else if (n < stages) {
            load(cur, n);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int U>
void run(const uint4 *src, uint4 *dst, size_t bytes, int per_cu, int wmul, unsigned *sink)
{
    const long stages = bytes / (256L * U * 16);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = per_cu * 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((stream_kernel<U, 1>), dim3(grid), dim3(256), 0, 0, src, dst, stages, wmul, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double tb = 4.0 * bytes * (1 + wmul) / (ms * 1e-3) / 1e12;
    printf("U=%2d G=%d wmul=%d: in flight/CU %4d KB  %6.2f TB/s (read+write)  read %5.2f TB/s\n", U, per_cu, wmul,
           per_cu * U * 4, tb, tb / (1 + wmul));
}

int main()
{
    const size_t bytes = 1UL << 30;
    uint4 *src, *dst;
    unsigned *sink;
    hipMalloc(&src, bytes);
    hipMalloc(&dst, 2 * bytes);
    hipMalloc(&sink, 64);
    hipMemset(src, 1, bytes);
    for (int wmul = 0; wmul <= 2; ++wmul)
        for (int g = 1; g <= 4; ++g) {
            run<1>(src, dst, bytes, g, wmul, sink);
            run<2>(src, dst, bytes, g, wmul, sink);
            run<4>(src, dst, bytes, g, wmul, sink);
            run<8>(src, dst, bytes, g, wmul, sink);
            run<16>(src, dst, bytes, g, wmul, sink);
        }
    return 0;
}
