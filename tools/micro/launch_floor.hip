// Cost of a dependent tiny kernel on a busy stream (gfx950): N back-to-back launches of (a) an empty kernel,
// (b) a 4-block kernel that sums 512 x 256 floats of slabs, timed with events around the whole train.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void empty_kernel(float *p) { if (p && threadIdx.x == 9999) p[0] = 1.0f; }
__global__ __launch_bounds__(1024) void slab_kernel(const float *s, float *o, int nparts, int C)
{
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
    float a = 0.f;
    for (int k = pl; k < nparts; k += 32) a += s[(long)k * C + c];
    __shared__ float red[32][32];
    red[pl][threadIdx.x & 31] = a;
    __syncthreads();
    if (pl == 0) { float t = 0.f; for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x & 31]; o[c] = t; }
}
__global__ void big_kernel(float4 *p, long n) { for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) { float4 v = p[i]; v.x += 1.f; p[i] = v; } }
int main()
{
    float *s, *o; float4 *big;
    hipMalloc(&s, 512 * 256 * 4); hipMalloc(&o, 1024); hipMalloc(&big, 256 << 20);
    hipMemset(s, 0, 512 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, o);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("empty kernel: %.2f us per launch\n", ms * 1e3 / N);
        hipEventRecord(a);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(slab_kernel, dim3(4), dim3(1024), 0, 0, s, o, 512, 128);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("slab kernel (4 blocks, 512 slabs): %.2f us per launch\n", ms * 1e3 / N);
        // alternating big (64 MB rw) and tiny kernels: the extra time a tiny kernel adds
        hipEventRecord(a);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(big_kernel, dim3(2048), dim3(256), 0, 0, big, (long)(64 << 20) / 16);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        const float only_big = ms * 1e3 / 200;
        hipEventRecord(a);
        for (int i = 0; i < 200; ++i) {
            hipLaunchKernelGGL(big_kernel, dim3(2048), dim3(256), 0, 0, big, (long)(64 << 20) / 16);
            hipLaunchKernelGGL(slab_kernel, dim3(4), dim3(1024), 0, 0, s, o, 512, 128);
        }
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("big kernel alone %.2f us; big + slab kernel %.2f us -> a tiny kernel between big ones adds %.2f us\n", only_big,
               ms * 1e3 / 200, ms * 1e3 / 200 - only_big);
    }
    return 0;
}
