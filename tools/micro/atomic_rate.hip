// How many returning atomic adds per second one 32-bit word takes from the persistent workgroups of a stage launch,
// and what one costs the workgroup that waits for it.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void grab(uint32_t* ctr, uint32_t limit, int words, int stride, uint32_t* sink)
{
    __shared__ uint32_t s;
    uint32_t acc = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            uint32_t v = atomicAdd(ctr, 1u);
            for (int k = 1; k < words; k++) v += atomicAdd(ctr + k * stride, 1u) & 1u;     // dependent on nothing: issued back to back
            s = v;
        }
        __syncthreads();
        uint32_t v = s;
        __syncthreads();
        if (v >= limit) break;
        acc += v;
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main()
{
    uint32_t *ctr, *sink;
    hipMalloc(&ctr, 1 << 16); hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t limit = 200000;
    for (int blocks : {256, 1024, 2048, 4096}) for (int words : {1, 2, 3}) for (int stride : {1, 64}) {
        if (words == 1 && stride != 1) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(ctr, 0, 1 << 16);
            hipEventRecord(a);
            grab<<<blocks, 256>>>(ctr, limit, words, stride, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("blocks %5d words %d stride %3d words apart: %8.3f ms for %u grabs = %6.1f ns per grab\n", blocks, words, stride, best, limit, best * 1e6 / limit);
    }
    return 0;
}
