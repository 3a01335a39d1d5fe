// Microbenchmark: how fast can a CU gather randomly placed node-sized records?
// Each lane chases a pseudo-random chain of record indices (dependent loads, like BVH traversal).
// Variants: bytes per record fetched by one lane as N x 16-byte loads, or 4 lanes sharing one record.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NLOAD, int STRIDE16>
__global__ __launch_bounds__(256) void chase(const float4* __restrict__ buf, uint32_t nrec, int iters, uint32_t* out)
{
    uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u % nrec;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        const float4* p = buf + (size_t)idx * STRIDE16;
        float4 v[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; k++) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < NLOAD; k++) acc += v[k].y + v[k].z;
        idx = (__float_as_uint(v[0].x) + (uint32_t)(acc != 12345.f)) % nrec;      // next index stored in the record
    }
    out[blockIdx.x * 256 + threadIdx.x] = idx + (uint32_t)acc;
}

// same chase with an LDS footprint that limits occupancy and ALU work between the dependent loads
template <int LDS_BYTES, int ALU>
__global__ __launch_bounds__(256) void chase_model(const float4* __restrict__ buf, uint32_t nrec, int iters, uint32_t* out)
{
    __shared__ float lds[LDS_BYTES / 4];
    uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u % nrec;
    float acc = 0.f;
    lds[threadIdx.x] = 0.f;
    for (int i = 0; i < iters; i++) {
        const float4* p = buf + (size_t)idx * 4;
        float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        float a = v0.y + v1.y + v2.y + v3.y;
#pragma unroll
        for (int k = 0; k < ALU; k++) a = a * 1.0001f + acc;
        acc += a;
        lds[threadIdx.x + 256 * (i & 7)] = acc;
        idx = (__float_as_uint(v0.x) + (uint32_t)(acc == 12345.f) + (uint32_t)(lds[threadIdx.x + 256 * ((i + 3) & 7)] == 7.f)) % nrec;
    }
    out[blockIdx.x * 256 + threadIdx.x] = idx + (uint32_t)acc;
}

// 4 lanes share one 64-byte record: each loads 16 bytes (coalesced 64 B), values exchanged by shuffles
__global__ __launch_bounds__(256) void chase_quad(const float4* __restrict__ buf, uint32_t nrec, int iters, uint32_t* out)
{
    int lane = threadIdx.x & 63;
    uint32_t ray = (blockIdx.x * 256 + threadIdx.x) >> 2;
    uint32_t idx = ray * 2654435761u % nrec;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        float4 v = buf[(size_t)idx * 4 + (lane & 3)];
        float s = v.y + v.z;
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
        acc += s;
        uint32_t nx = __float_as_uint(__shfl(v.x, lane & ~3, 64));
        idx = (nx + (uint32_t)(acc != 12345.f)) % nrec;
    }
    out[blockIdx.x * 256 + threadIdx.x] = idx + (uint32_t)acc;
}

int main(int argc, char** argv)
{
    size_t mb = argc > 1 ? atoi(argv[1]) : 16;
    uint32_t nrec = (uint32_t)(mb * 1024 * 1024 / 64);
    std::vector<float> h((size_t)nrec * 16);
    uint32_t s = 12345;
    for (uint32_t r = 0; r < nrec; r++) {
        for (int k = 0; k < 16; k++) h[(size_t)r * 16 + k] = 1.0f;
        s = s * 1664525u + 1013904223u;
        uint32_t nx = (s >> 4) % nrec;
        for (int q = 0; q < 4; q++) memcpy(&h[(size_t)r * 16 + q * 4], &nx, 4);
    }
    float4* d; uint32_t* out;
    int blocks = 256 * 8, iters = 400;
    CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double lanes_per_rec) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double recs = (double)blocks * 256 * iters / lanes_per_rec;
        printf("%-28s %8.3f ms  %8.2f Grec/s  %.2f clk/rec/CU @2.4GHz\n", name, ms, recs / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / recs);
    };
    printf("buffer %zu MB, %u records, %d blocks x 256, %d dependent steps per lane\n", mb, nrec, blocks, iters);
    run("1 x 16B of a 64B record", [&] { hipLaunchKernelGGL((chase<1, 4>), dim3(blocks), dim3(256), 0, 0, d, nrec, iters, out); }, 1);
    run("2 x 16B of a 64B record", [&] { hipLaunchKernelGGL((chase<2, 4>), dim3(blocks), dim3(256), 0, 0, d, nrec, iters, out); }, 1);
    run("4 x 16B of a 64B record", [&] { hipLaunchKernelGGL((chase<4, 4>), dim3(blocks), dim3(256), 0, 0, d, nrec, iters, out); }, 1);
    run("2 x 16B of a 32B record", [&] { hipLaunchKernelGGL((chase<2, 2>), dim3(blocks), dim3(256), 0, 0, d, nrec * 2, iters, out); }, 1);
    // other record sizes: n16 = number of 16-byte units in the buffer
    const uint32_t n16 = nrec * 4;
    run("3 x 16B of a 48B record", [&] { hipLaunchKernelGGL((chase<3, 3>), dim3(blocks), dim3(256), 0, 0, d, n16 / 3, iters, out); }, 1);
    run("5 x 16B of an 80B record", [&] { hipLaunchKernelGGL((chase<5, 5>), dim3(blocks), dim3(256), 0, 0, d, n16 / 5, iters, out); }, 1);
    run("5 x 16B in a 128B slot", [&] { hipLaunchKernelGGL((chase<5, 8>), dim3(blocks), dim3(256), 0, 0, d, n16 / 8, iters, out); }, 1);
    run("6 x 16B of a 96B record", [&] { hipLaunchKernelGGL((chase<6, 6>), dim3(blocks), dim3(256), 0, 0, d, n16 / 6, iters, out); }, 1);
    run("8 x 16B of a 128B record", [&] { hipLaunchKernelGGL((chase<8, 8>), dim3(blocks), dim3(256), 0, 0, d, n16 / 8, iters, out); }, 1);
    run("4 x 16B in a 128B slot", [&] { hipLaunchKernelGGL((chase<4, 8>), dim3(blocks), dim3(256), 0, 0, d, n16 / 8, iters, out); }, 1);
    run("4x16B, 32 waves/CU, 64 ALU", [&] { hipLaunchKernelGGL((chase_model<16384, 64>), dim3(blocks), dim3(256), 0, 0, d, nrec, iters, out); }, 1);
    run("4x16B, 20 waves/CU, 64 ALU", [&] { hipLaunchKernelGGL((chase_model<32768, 64>), dim3(blocks), dim3(256), 0, 0, d, nrec, iters, out); }, 1);
    run("4x16B, 20 waves/CU, 160 ALU", [&] { hipLaunchKernelGGL((chase_model<32768, 160>), dim3(blocks), dim3(256), 0, 0, d, nrec, iters, out); }, 1);
    run("4x16B, 12 waves/CU, 160 ALU", [&] { hipLaunchKernelGGL((chase_model<49152, 160>), dim3(blocks), dim3(256), 0, 0, d, nrec, iters, out); }, 1);
    run("4x16B, 32 waves/CU, 160 ALU", [&] { hipLaunchKernelGGL((chase_model<16384, 160>), dim3(blocks), dim3(256), 0, 0, d, nrec, iters, out); }, 1);
    run("4 lanes share a 64B record", [&] { hipLaunchKernelGGL(chase_quad, dim3(blocks), dim3(256), 0, 0, d, nrec, iters, out); }, 4);
    return 0;
}
