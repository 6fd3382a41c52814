// What would one iteration of a multi-workgroup FPS cost?  Every iteration needs the partial arg-max of ALL
// workgroups of a scene before any of them can go on: an all-to-all exchange through global memory.
// W workgroups (one per CU) run `iters` rounds of: publish (value, round) -> wait until all W slots show the
// round -> next.  Reported: microseconds per round.  (W = 1 has no partner: the cost of the atomics alone.)
//   hipcc --offload-arch=gfx950 -O3 -o pingpong.bin tools/micro/pingpong.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void rounds_kernel(unsigned long long *slots, int W, int iters, unsigned long long *out)
{
    const int w = blockIdx.x;
    unsigned long long best = 0;
    for (int r = 1; r <= iters; ++r) {
        if (threadIdx.x == 0) {
            // publish this workgroup's candidate for round r (value in the low bits, round in the high ones)
            __hip_atomic_store(&slots[w * 16], ((unsigned long long)r << 32) | (unsigned)(w * 7 + r), __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x < W) {
            unsigned long long v;
            do {
                v = __hip_atomic_load(&slots[threadIdx.x * 16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            } while ((v >> 32) < (unsigned long long)r);
            best = v > best ? v : best;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[w] = best;
}
int main()
{
    unsigned long long *slots, *out;
    hipMalloc(&slots, 64 * 16 * 8);
    hipMalloc(&out, 64 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 1024;
    for (int W : {1, 2, 4, 8, 16}) {
        float best_ms = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(slots, 0, 64 * 16 * 8);
            hipEventRecord(a);
            hipLaunchKernelGGL(rounds_kernel, dim3(W), dim3(256), 0, 0, slots, W, iters, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best_ms = ms < best_ms ? ms : best_ms;
        }
        printf("W=%2d workgroups: %.2f us per round (%d rounds: %.2f ms)\n", W, best_ms * 1e3 / iters, iters, best_ms);
    }
    return 0;
}
