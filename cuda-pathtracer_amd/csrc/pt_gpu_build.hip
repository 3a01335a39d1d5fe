// BVH build on the GPU: a linear BVH (Morton order + the radix-tree construction of Karras 2012) written in
// the reference's node layout (structs.cuh:130-137), so that it enters the library through the same upload
// path as a tree from BVH_builder.cpp. The reference has no counterpart: its builder is a host loop of
// O(297 * n * depth). Closest hits do not depend on the tree (every triangle is tested against the same
// rule), so a render with this tree gives the same image; only the traversal cost and the test counters differ.
//
// Layout produced: node 0 is the root; the children of radix-tree node j sit at 2j+1 and 2j+2 (the reference
// keeps children as a consecutive pair, leftFirst and leftFirst+1); every leaf holds one triangle,
// leftFirst = position in tri_idx, which lists triangles in Morton order. Boxes carry the reference's
// padding of 0.1 per vertex (BVH_builder.cpp:19-24): the box test rejects boxes of zero thickness.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdint>
#include "pt_internal.h"
#include "pt_launch.h"

namespace ptamd {
namespace {

constexpr float kPad = 0.1f;

#define GB_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); rc = PT_ERR_HIP; goto done; } } while (0)

// floats as order-preserving unsigned integers, for atomic min/max
__device__ inline uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

struct Bounds { uint32_t lo[3], hi[3]; };

__global__ __launch_bounds__(256) void gb_scene_bounds(const pt_triangle* __restrict__ tris, int n, Bounds* __restrict__ b)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    float c[3] = {0.f, 0.f, 0.f};
    bool ok = i < n;
    if (ok) {
        const pt_triangle& t = tris[i];
        c[0] = (t.a.x + t.b.x + t.c.x) * 0.3333f; c[1] = (t.a.y + t.b.y + t.c.y) * 0.3333f; c[2] = (t.a.z + t.b.z + t.c.z) * 0.3333f;
        ok = isfinite(c[0]) && isfinite(c[1]) && isfinite(c[2]);
    }
    for (int a = 0; a < 3; a++) {
        float lo = ok ? c[a] : 3.0e38f, hi = ok ? c[a] : -3.0e38f;
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_down(lo, o, 64)); hi = fmaxf(hi, __shfl_down(hi, o, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&b->lo[a], f2ord(lo)); atomicMax(&b->hi[a], f2ord(hi)); }
    }
}

__device__ inline uint64_t spread21(uint32_t v)       // 21 bits -> every third bit of 63
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void gb_morton(const pt_triangle* __restrict__ tris, int n, const Bounds* __restrict__ b,
                                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ ids)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const pt_triangle& t = tris[i];
    float c[3] = {(t.a.x + t.b.x + t.c.x) * 0.3333f, (t.a.y + t.b.y + t.c.y) * 0.3333f, (t.a.z + t.b.z + t.c.z) * 0.3333f};
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        float lo = ord2f(b->lo[a]), hi = ord2f(b->hi[a]);
        float ext = hi - lo;
        float u = ext > 0.f ? (c[a] - lo) / ext : 0.f;
        if (!(u >= 0.f)) u = 0.f;         // also non-finite centroids
        if (u > 1.f) u = 1.f;
        q[a] = (uint32_t)(u * 2097151.0f);
    }
    keys[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
    ids[i] = (uint32_t)i;
}

// length of the common prefix of the keys at sorted positions i and j; equal keys are told apart by position
__device__ inline int prefix(const uint64_t* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// radix-tree node j (0 <= j < n-1): its children (>= 0: internal node, < 0: ~leaf) and the parent links
__global__ __launch_bounds__(256) void gb_radix_tree(const uint64_t* __restrict__ keys, int n, int* __restrict__ left, int* __restrict__ right,
                                                     int* __restrict__ parent_of_internal, int* __restrict__ parent_of_leaf)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int dmin = prefix(keys, n, i, i - d);
    int lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = prefix(keys, n, i, j);
    int s = 0;
    for (int div = 2, t = (l + div - 1) / div; ; div <<= 1, t = (l + div - 1) / div) {
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    int gamma = i + s * d + (d < 0 ? -1 : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    int cl = lo == gamma ? ~gamma : gamma;
    int cr = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    left[i] = cl; right[i] = cr;
    // a child learns its parent and on which side it hangs: bit 0 = right side
    if (cl >= 0) parent_of_internal[cl] = i << 1; else parent_of_leaf[~cl] = i << 1;
    if (cr >= 0) parent_of_internal[cr] = i << 1 | 1; else parent_of_leaf[~cr] = i << 1 | 1;
}

// Leaves write their node and walk up; the second child to arrive at a node computes its box and goes on.
__global__ __launch_bounds__(256) void gb_refit(const pt_triangle* __restrict__ tris, const uint32_t* __restrict__ ids, int n,
                                                const int* __restrict__ left, const int* __restrict__ right,
                                                const int* __restrict__ parent_of_internal, const int* __restrict__ parent_of_leaf,
                                                uint32_t* __restrict__ visits, pt_bvh_node* __restrict__ nodes, int* __restrict__ tri_idx)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const pt_triangle& t = tris[ids[i]];
    pt_bvh_node leaf;
    leaf.aabbMin = pt_float3{fminf(fminf(t.a.x, t.b.x), t.c.x) - kPad, fminf(fminf(t.a.y, t.b.y), t.c.y) - kPad, fminf(fminf(t.a.z, t.b.z), t.c.z) - kPad};
    leaf.aabbMax = pt_float3{fmaxf(fmaxf(t.a.x, t.b.x), t.c.x) + kPad, fmaxf(fmaxf(t.a.y, t.b.y), t.c.y) + kPad, fmaxf(fmaxf(t.a.z, t.b.z), t.c.z) + kPad};
    leaf.leftFirst = i; leaf.triCount = 1;
    tri_idx[i] = (int)ids[i];
    if (n == 1) { nodes[0] = leaf; return; }
    int link = parent_of_leaf[i];
    nodes[2 * (link >> 1) + 1 + (link & 1)] = leaf;
    __threadfence();
    int cur = link >> 1;
    for (;;) {
        if (atomicAdd(&visits[cur], 1u) == 0u) return;        // the sibling subtree is not done yet
        __threadfence();
        const pt_bvh_node a = nodes[2 * cur + 1], b = nodes[2 * cur + 2];
        pt_bvh_node me;
        me.aabbMin = pt_float3{fminf(a.aabbMin.x, b.aabbMin.x), fminf(a.aabbMin.y, b.aabbMin.y), fminf(a.aabbMin.z, b.aabbMin.z)};
        me.aabbMax = pt_float3{fmaxf(a.aabbMax.x, b.aabbMax.x), fmaxf(a.aabbMax.y, b.aabbMax.y), fmaxf(a.aabbMax.z, b.aabbMax.z)};
        me.leftFirst = 2 * cur + 1; me.triCount = 0;
        if (cur == 0) { nodes[0] = me; return; }
        int up = parent_of_internal[cur];
        nodes[2 * (up >> 1) + 1 + (up & 1)] = me;
        __threadfence();
        cur = up >> 1;
    }
}

} // namespace

// tris_dev: n triangles on the device. nodes_dev: 2n-1 nodes, tri_idx_dev: n indices, both written here.
int gpu_build_lbvh(const pt_triangle* tris_dev, int n, pt_bvh_node* nodes_dev, int* tri_idx_dev, hipStream_t stream)
{
    if (!tris_dev || n <= 0 || !nodes_dev || !tri_idx_dev) { set_error("gpu_build_lbvh: bad argument"); return PT_ERR_ARG; }
    int rc = PT_OK;
    Bounds* d_bounds = nullptr;
    uint64_t *d_keys = nullptr, *d_keys_sorted = nullptr;
    uint32_t *d_ids = nullptr, *d_ids_sorted = nullptr, *d_visits = nullptr;
    int* d_links = nullptr;          // left, right, parent_of_internal, parent_of_leaf: n entries each
    void* d_temp = nullptr;
    size_t temp_bytes = 0;
    const dim3 grid((n + 255) / 256), block(256);
    {
        GB_TRY(hipMalloc((void**)&d_bounds, sizeof(Bounds)));
        GB_TRY(hipMalloc((void**)&d_keys, (size_t)n * sizeof(uint64_t)));
        GB_TRY(hipMalloc((void**)&d_keys_sorted, (size_t)n * sizeof(uint64_t)));
        GB_TRY(hipMalloc((void**)&d_ids, (size_t)n * sizeof(uint32_t)));
        GB_TRY(hipMalloc((void**)&d_ids_sorted, (size_t)n * sizeof(uint32_t)));
        GB_TRY(hipMalloc((void**)&d_visits, (size_t)n * sizeof(uint32_t)));
        GB_TRY(hipMalloc((void**)&d_links, (size_t)n * 4 * sizeof(int)));
        Bounds init;
        for (int a = 0; a < 3; a++) { init.lo[a] = 0xffffffffu; init.hi[a] = 0u; }
        GB_TRY(hipMemcpyAsync(d_bounds, &init, sizeof(init), hipMemcpyHostToDevice, stream));
        GB_TRY(hipMemsetAsync(d_visits, 0, (size_t)n * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(gb_scene_bounds, grid, block, 0, stream, tris_dev, n, d_bounds);
        hipLaunchKernelGGL(gb_morton, grid, block, 0, stream, tris_dev, n, d_bounds, d_keys, d_ids);
        GB_TRY(hipGetLastError());
        GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, n, 0, 63, stream));
        GB_TRY(hipMalloc(&d_temp, temp_bytes));
        GB_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, n, 0, 63, stream));
        int* left = d_links; int* right = d_links + n; int* pin = d_links + 2 * (size_t)n; int* pleaf = d_links + 3 * (size_t)n;
        if (n > 1) hipLaunchKernelGGL(gb_radix_tree, grid, block, 0, stream, d_keys_sorted, n, left, right, pin, pleaf);
        hipLaunchKernelGGL(gb_refit, grid, block, 0, stream, tris_dev, d_ids_sorted, n, left, right, pin, pleaf, d_visits, nodes_dev, tri_idx_dev);
        GB_TRY(hipGetLastError());
        GB_TRY(hipStreamSynchronize(stream));
    }
done:
    if (d_temp) (void)hipFree(d_temp);
    if (d_links) (void)hipFree(d_links);
    if (d_visits) (void)hipFree(d_visits);
    if (d_ids_sorted) (void)hipFree(d_ids_sorted);
    if (d_ids) (void)hipFree(d_ids);
    if (d_keys_sorted) (void)hipFree(d_keys_sorted);
    if (d_keys) (void)hipFree(d_keys);
    if (d_bounds) (void)hipFree(d_bounds);
    return rc;
}

} // namespace ptamd
