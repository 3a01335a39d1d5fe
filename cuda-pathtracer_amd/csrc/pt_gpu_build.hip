// BVH build on the GPU: a linear BVH (Morton order + the radix-tree construction of Karras 2012). Two outputs:
// gpu_build_scene leaves the traversal layouts themselves in device memory (binary nodes in breadth-first order,
// their quantized twins, leaf-ordered triangle records, 4-wide nodes; nothing but a few counters returns to the
// host), and gpu_build_lbvh writes the reference's node layout (structs.cuh:130-137), so that the same tree can enter
// the library through the upload path of a tree from BVH_builder.cpp (kept as the cross-check of the first). The reference has no counterpart: its builder is a host loop of
// O(297 * n * depth). Closest hits do not depend on the tree (every triangle is tested against the same
// rule), so a render with this tree gives the same image; only the traversal cost and the test counters differ.
//
// Layout produced: node 0 is the root; the children of radix-tree node j sit at 2j+1 and 2j+2 (the reference
// keeps children as a consecutive pair, leftFirst and leftFirst+1); every leaf holds one triangle,
// leftFirst = position in tri_idx, which lists triangles in Morton order. Boxes carry the reference's
// padding of 0.1 per vertex (BVH_builder.cpp:19-24): the box test rejects boxes of zero thickness.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <vector>
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
                                                     int* __restrict__ parent_of_internal, int* __restrict__ parent_of_leaf,
                                                     int* __restrict__ range_first, int* __restrict__ range_count)
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
    if (range_first) { range_first[i] = lo; range_count[i] = hi - lo + 1; }      // the leaves below: sorted positions [lo, hi]
    if (!parent_of_internal) return;
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

// ---------------------------------------------------------------- traversal layouts, built on the device
// The radix tree is renumbered breadth first (node 0 = root, the interior children of consecutive nodes are
// consecutive): the bottom-up pass then runs level by level over contiguous nodes with plain stores, no atomics.

struct TightBox { float lo[3], hi[3]; };
struct NodeAux { int first, count, height, mergeable; };     // leaves below (sorted positions), levels below, "may become one leaf"

// interior children of the nodes of one level: 0, 1 or 2 each
__global__ __launch_bounds__(256) void gb_bfs_count(const int* __restrict__ order, int lo, int cnt, const int* __restrict__ left, const int* __restrict__ right, int* __restrict__ kids)
{
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= cnt) return;
    int r = order[lo + q];
    kids[q] = (left[r] >= 0 ? 1 : 0) + (right[r] >= 0 ? 1 : 0);
}

// the next level: children in the order of their parents, left before right; off = exclusive running sum of kids
__global__ __launch_bounds__(256) void gb_bfs_emit(int* __restrict__ order, int lo, int cnt, const int* __restrict__ left, const int* __restrict__ right,
                                                   const int* __restrict__ off, int* __restrict__ newid)
{
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= cnt) return;
    int r = order[lo + q];
    int at = lo + cnt + off[q];
    int l = left[r], rr = right[r];
    if (l >= 0) { order[at] = l; newid[l] = at; at++; }
    if (rr >= 0) { order[at] = rr; newid[rr] = at; }
}

__device__ inline TightBox tri_bounds(const pt_triangle& t)
{
    TightBox b;
    b.lo[0] = fminf(fminf(t.a.x, t.b.x), t.c.x); b.lo[1] = fminf(fminf(t.a.y, t.b.y), t.c.y); b.lo[2] = fminf(fminf(t.a.z, t.b.z), t.c.z);
    b.hi[0] = fmaxf(fmaxf(t.a.x, t.b.x), t.c.x); b.hi[1] = fmaxf(fmaxf(t.a.y, t.b.y), t.c.y); b.hi[2] = fmaxf(fmaxf(t.a.z, t.b.z), t.c.z);
    return b;
}
__device__ inline TightBox join(const TightBox& x, const TightBox& y)
{
    TightBox b;
    for (int a = 0; a < 3; a++) { b.lo[a] = fminf(x.lo[a], y.lo[a]); b.hi[a] = fmaxf(x.hi[a], y.hi[a]); }
    return b;
}
__device__ inline double half_area(const float* lo, const float* hi)
{
    double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
    return x * y + y * z + z * x;
}

// One level of the bottom-up pass (deepest level first). Node at breadth-first position p: the tight bounds of
// the triangles below each child, the binary node with the reference's padded boxes (0.1 per side,
// BVH_builder.cpp:19-24; padding commutes with the union), the range of leaves, the height, and whether the
// subtree may be tested as one leaf (its union is no larger in surface than its halves together, recursively).
__global__ __launch_bounds__(256) void gb_fit_level(const pt_triangle* __restrict__ tris, const uint32_t* __restrict__ ids, const int* __restrict__ order,
                                                    const int* __restrict__ newid, const int* __restrict__ left, const int* __restrict__ right,
                                                    int lo, int cnt, DNode* __restrict__ nodes, TightBox* __restrict__ tight_l, TightBox* __restrict__ tight_r,
                                                    NodeAux* __restrict__ aux)
{
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= cnt) return;
    const int p = lo + q;
    const int r = order[p];
    const int L = left[r], R = right[r];
    TightBox bl, br;
    int fl, cl, hl, ml, fr, cr, hr, mr, refl, refr;
    if (L < 0) { fl = ~L; cl = 1; hl = 0; ml = 1; bl = tri_bounds(tris[ids[fl]]); refl = L; }
    else { const int c = newid[L]; const NodeAux a = aux[c]; fl = a.first; cl = a.count; hl = a.height; ml = a.mergeable; bl = join(tight_l[c], tight_r[c]); refl = c; }
    if (R < 0) { fr = ~R; cr = 1; hr = 0; mr = 1; br = tri_bounds(tris[ids[fr]]); refr = R; }
    else { const int c = newid[R]; const NodeAux a = aux[c]; fr = a.first; cr = a.count; hr = a.height; mr = a.mergeable; br = join(tight_l[c], tight_r[c]); refr = c; }
    tight_l[p] = bl; tight_r[p] = br;
    const TightBox u = join(bl, br);
    NodeAux me;
    me.first = fl < fr ? fl : fr; me.count = cl + cr; me.height = 1 + (hl > hr ? hl : hr);
    me.mergeable = (ml && mr && half_area(u.lo, u.hi) <= half_area(bl.lo, bl.hi) + half_area(br.lo, br.hi)) ? 1 : 0;
    aux[p] = me;
    DNode d;
    for (int a = 0; a < 3; a++) { d.lmin[a] = bl.lo[a] - kPad; d.lmax[a] = bl.hi[a] + kPad; d.rmin[a] = br.lo[a] - kPad; d.rmax[a] = br.hi[a] + kPad; }
    d.left = refl; d.right = refr; d.pad0 = 0; d.pad1 = 0;
    nodes[p] = d;
}

// leaf-ordered triangle records (one triangle per leaf) and the index array in the same order
__global__ __launch_bounds__(256) void gb_emit_tris(const pt_triangle* __restrict__ tris, const uint32_t* __restrict__ ids, int n, DTri* __restrict__ out,
                                                    int* __restrict__ tri_idx, uint32_t* __restrict__ max_coord_bits)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    float m = 0.f;
    if (i < n) {
        const uint32_t id = ids[i];
        const pt_triangle& T = tris[id];
        DTri r;
        r.q0 = make_float4(T.a.x, T.a.y, T.a.z, __uint_as_float(id << 1));
        r.q1 = make_float4(T.b.x - T.a.x, T.b.y - T.a.y, T.b.z - T.a.z, __int_as_float(1));
        r.q2 = make_float4(T.c.x - T.a.x, T.c.y - T.a.y, T.c.z - T.a.z, 0.f);
        out[i] = r;
        tri_idx[i] = (int)id;
        const float v[9] = {T.a.x, T.a.y, T.a.z, T.b.x, T.b.y, T.b.z, T.c.x, T.c.y, T.c.z};
        for (int k = 0; k < 9; k++) { float a = fabsf(v[k]); if (a > m && a < 1e30f) m = a; }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(max_coord_bits, __float_as_uint(m));      // non-negative floats order as integers
}

// ---------------------------------------------------------------- agglomerative tree (PLOC, Meister & Bittner 2018)
// Clusters start as the triangles in Morton order. Every round each cluster looks at its kPlocRadius neighbours on
// either side for the one with which it would form the smallest box; two clusters that choose each other merge.
// The trees come out close to a full SAH sweep's in traversal cost, at a few more passes over the data than the
// radix tree. Everything is decided by running sums, not by atomics: the same input gives the same tree.
constexpr int kPlocRadius = 16;
constexpr int kTopNodes = 8191;         // nodes of the first levels that the host rebuilds with a surface-area sweep

__global__ __launch_bounds__(256) void gb_ploc_init(const pt_triangle* __restrict__ tris, const uint32_t* __restrict__ ids, int n, int* __restrict__ cl_node, TightBox* __restrict__ cl_box)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    cl_node[i] = ~i;
    cl_box[i] = tri_bounds(tris[ids[i]]);
}

__global__ __launch_bounds__(256) void gb_ploc_nearest(const TightBox* __restrict__ cl_box, int m, int radius, int* __restrict__ nn)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const TightBox me = cl_box[i];
    double best = INFINITY; int pick = -1;
    auto look = [&](int j) {
        if (j < 0 || j >= m || j == i) return;
        const TightBox u = join(me, cl_box[j]);
        const double a = half_area(u.lo, u.hi);
        if (a < best || pick < 0) { best = a; pick = j; }
    };
    // equal boxes: the partner of the pair (2k, 2k+1) is looked at first and wins ties, so that a run of identical
    // clusters halves every round instead of merging one pair
    look(i ^ 1);
    for (int d = 1; d <= radius; d++) { if ((i - d) != (i ^ 1)) look(i - d); if ((i + d) != (i ^ 1)) look(i + d); }
    nn[i] = pick;
}

// keep[i]: cluster i survives the round (alone, or as the merged pair); fresh[i]: it becomes a new node
__global__ __launch_bounds__(256) void gb_ploc_flags(const int* __restrict__ nn, int m, int* __restrict__ keep, int* __restrict__ fresh)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int j = nn[i];
    const bool pair = j >= 0 && nn[j] == i;
    keep[i] = (!pair || i < j) ? 1 : 0;
    fresh[i] = (pair && i < j) ? 1 : 0;
}

__global__ __launch_bounds__(256) void gb_ploc_apply(const int* __restrict__ nn, int m, const int* __restrict__ keep, const int* __restrict__ keep_off,
                                                     const int* __restrict__ fresh, const int* __restrict__ fresh_off, int node_base,
                                                     const int* __restrict__ cl_node, const TightBox* __restrict__ cl_box,
                                                     int* __restrict__ nx_node, TightBox* __restrict__ nx_box, int* __restrict__ left, int* __restrict__ right)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m || !keep[i]) return;
    const int at = keep_off[i];
    if (fresh[i]) {
        const int j = nn[i];
        const int id = node_base + fresh_off[i];
        left[id] = cl_node[i]; right[id] = cl_node[j];
        nx_node[at] = id;
        nx_box[at] = join(cl_box[i], cl_box[j]);
    } else {
        nx_node[at] = cl_node[i];
        nx_box[at] = cl_box[i];
    }
}

// leaves below every node of a level (bottom-up), then the first leaf position of every child (top-down): a subtree's
// triangles become contiguous, left subtree first
__global__ __launch_bounds__(256) void gb_count_level(const int* __restrict__ order, const int* __restrict__ newid, const int* __restrict__ left, const int* __restrict__ right,
                                                      int lo, int cnt, int* __restrict__ count)
{
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= cnt) return;
    const int r = order[lo + q];
    const int L = left[r], R = right[r];
    count[lo + q] = (L < 0 ? 1 : count[newid[L]]) + (R < 0 ? 1 : count[newid[R]]);
}

__global__ __launch_bounds__(256) void gb_first_level(const int* __restrict__ order, const int* __restrict__ newid, const int* __restrict__ left, const int* __restrict__ right,
                                                      int lo, int cnt, const int* __restrict__ count, int* __restrict__ first, int* __restrict__ leaf_pos)
{
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= cnt) return;
    const int r = order[lo + q];
    const int L = left[r], R = right[r];
    const int f = first[lo + q];
    int nl;
    if (L < 0) { leaf_pos[~L] = f; nl = 1; } else { const int c = newid[L]; first[c] = f; nl = count[c]; }
    if (R < 0) leaf_pos[~R] = f + nl; else first[newid[R]] = f + nl;
}

__global__ __launch_bounds__(256) void gb_remap_leaves(int n, const int* __restrict__ leaf_pos, const uint32_t* __restrict__ ids_in, uint32_t* __restrict__ ids_out,
                                                       int* __restrict__ left, int* __restrict__ right)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) ids_out[leaf_pos[i]] = ids_in[i];
    if (i < n - 1) {
        const int L = left[i], R = right[i];
        if (L < 0) left[i] = ~leaf_pos[~L];
        if (R < 0) right[i] = ~leaf_pos[~R];
    }
}

struct QuantGrid { float origin[3], cell[3]; };

// the binary nodes on the global 16-bit grid, two cells of padding (the fallback walk of the production kernel)
__global__ __launch_bounds__(256) void gb_quantize(const DNode* __restrict__ nodes, int n_nodes, QuantGrid g, DNodeQ* __restrict__ out)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    const DNode s = nodes[i];
    DNodeQ t;
    for (int a = 0; a < 3; a++) {
        const double o = (double)g.origin[a], c = (double)g.cell[a];
        auto qlo = [&](float v) { double q = floor(((double)v - o) / c) - 2.0; return (uint16_t)(q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q)); };
        auto qhi = [&](float v) { double q = ceil(((double)v - o) / c) + 2.0; return (uint16_t)(q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q)); };
        t.q[a] = qlo(s.lmin[a]); t.q[3 + a] = qhi(s.lmax[a]); t.q[6 + a] = qlo(s.rmin[a]); t.q[9 + a] = qhi(s.rmax[a]);
    }
    t.left = s.left; t.right = s.right;
    out[i] = t;
}

// ---- 4-wide nodes, level by level from the root. An item = (binary node to collapse, wide node to write, stack
// entries a walk may hold when it arrives there).
__device__ inline int wide_leaf_ref(int first, int count) { return ~(int)((uint32_t)first | ((uint32_t)(count - 1) << 26)); }

// binary reference -> wide reference: a leaf, a subtree that becomes one leaf, or the binary node itself (>= 0)
__device__ inline int wide_ref_of(int ref, const NodeAux* __restrict__ aux, int leaf_merge)
{
    if (ref < 0) return wide_leaf_ref(~ref, 1);
    const NodeAux a = aux[ref];
    if (a.mergeable && a.count <= leaf_merge && a.count <= 16) return wide_leaf_ref(a.first, a.count);
    return ref;
}

// Writes the wide node of every item with its interior children still as binary node numbers (>= 0) and counts them.
__global__ __launch_bounds__(256) void gb_wide_make(const int* __restrict__ it_src, const int* __restrict__ it_dst, const int* __restrict__ it_need, int cnt,
                                                    const DNode* __restrict__ nodes, const TightBox* __restrict__ tight_l, const TightBox* __restrict__ tight_r,
                                                    const NodeAux* __restrict__ aux, int leaf_merge, double margin, DNode4* __restrict__ n4,
                                                    int* __restrict__ kids, int* __restrict__ node_need, int* __restrict__ flags /* [0] max need, [1] bad box */)
{
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= cnt) return;
    const int src = it_src[q];
    int ref[4]; TightBox box[4]; double area[4];
    int nc = 2;
    ref[0] = wide_ref_of(nodes[src].left, aux, leaf_merge); box[0] = tight_l[src];
    ref[1] = wide_ref_of(nodes[src].right, aux, leaf_merge); box[1] = tight_r[src];
    area[0] = half_area(box[0].lo, box[0].hi); area[1] = half_area(box[1].lo, box[1].hi);
    while (nc < 4) {
        int pick = -1; double best = -1.0;
        for (int k = 0; k < nc; k++) if (ref[k] >= 0 && area[k] > best) { best = area[k]; pick = k; }
        if (pick < 0) break;
        const int s2 = ref[pick];
        ref[pick] = wide_ref_of(nodes[s2].left, aux, leaf_merge); box[pick] = tight_l[s2]; area[pick] = half_area(box[pick].lo, box[pick].hi);
        ref[nc] = wide_ref_of(nodes[s2].right, aux, leaf_merge); box[nc] = tight_r[s2]; area[nc] = half_area(box[nc].lo, box[nc].hi);
        nc++;
    }
    DNode4 d;
    for (int k = 0; k < 12; k++) d.w[k] = 0u;
    const int need = it_need[q] + nc - 1;
    // the node's grid: origin at the lowest corner of the widened boxes (rounded down to single precision),
    // cell = the smallest power of two with which the boxes span at most 255 cells
    uint32_t exps = 0;
    bool bad = false;
    for (int a = 0; a < 3; a++) {
        double lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < nc; k++) { lo = fmin(lo, (double)box[k].lo[a] - margin); hi = fmax(hi, (double)box[k].hi[a] + margin); }
        if (!(lo > -1e30 && hi < 1e30)) { bad = true; lo = 0.0; hi = 1.0; }
        float org = (float)lo;
        if ((double)org > lo) org = nextafterf(org, -INFINITY);
        int e = 0;
        (void)frexp((hi - (double)org) / 255.0, &e);
        if (e > -126 && hi - (double)org <= 255.0 * ldexp(1.0, e - 1)) e--;
        e = e < -126 ? -126 : (e > 127 ? 127 : e);
        const double cell = ldexp(1.0, e);
        d.w[a] = __float_as_uint(org);
        exps |= (uint32_t)(uint8_t)(int8_t)e << (8 * a);
        for (int k = 0; k < nc; k++) {
            double ql = floor(((double)box[k].lo[a] - margin - (double)org) / cell), qh = ceil(((double)box[k].hi[a] + margin - (double)org) / cell);
            ql = ql < 0.0 ? 0.0 : (ql > 255.0 ? 255.0 : ql); qh = qh < 0.0 ? 0.0 : (qh > 255.0 ? 255.0 : qh);
            d.w[4 + a] |= (uint32_t)ql << (8 * k);
            d.w[7 + a] |= (uint32_t)qh << (8 * k);
        }
        for (int k = nc; k < 4; k++) d.w[4 + a] |= 255u << (8 * k);      // a child that does not exist: inverted slab, never entered
    }
    d.w[3] = exps;
    int nk = 0;
    for (int k = 0; k < 4; k++) {
        if (k >= nc) { d.ref[k] = (int)0x80000000; continue; }
        d.ref[k] = ref[k];
        if (ref[k] >= 0) nk++;
    }
    n4[it_dst[q]] = d;
    kids[q] = nk;
    node_need[q] = need;
    atomicMax(&flags[0], need);
    if (bad) flags[1] = 1;
}

// Numbers the interior children (consecutive per node, nodes in item order) and emits the next level's items.
__global__ __launch_bounds__(256) void gb_wide_link(const int* __restrict__ it_dst, const int* __restrict__ node_need, int cnt, const int* __restrict__ off, int base,
                                                    DNode4* __restrict__ n4, int* __restrict__ nx_src, int* __restrict__ nx_dst, int* __restrict__ nx_need)
{
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= cnt) return;
    DNode4& d = n4[it_dst[q]];
    int j = off[q];
    for (int k = 0; k < 4; k++) {
        const int r = d.ref[k];
        if (r < 0) continue;
        nx_src[j] = r; nx_dst[j] = base + j; nx_need[j] = node_need[q];
        d.ref[k] = base + j;
        j++;
    }
}

// ---------------------------------------------------------------- SAH over the top of the tree (host)
// Morton-order builders decide the large-scale structure of the tree from a space-filling curve; a sweep over all
// split planes does that part better. The first levels (a few thousand nodes in breadth-first order) are therefore
// rebuilt on the host: their lowest children -- subtrees and single leaves -- become the items of a top-down build with
// the full surface-area sweep on three axes, weighted by triangle counts. The subtrees below stay as the device made
// them, and so does the order of the triangle records: no top node is ever tested as one leaf.
struct TopItem { int ref; TightBox box; int count, height; };      // ref: as in DNode (>= 0: node position, < 0: leaf)

static double item_area(const TightBox& b) { double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2]; return x * y + y * z + z * x; }
static TightBox item_join(const TightBox& a, const TightBox& b)
{
    TightBox u;
    for (int k = 0; k < 3; k++) { u.lo[k] = std::fmin(a.lo[k], b.lo[k]); u.hi[k] = std::fmax(a.hi[k], b.hi[k]); }
    return u;
}

// Twice the centre of a box along an axis, as a sort key. A box with vertices at both infinities has no centre
// (-inf + inf): it sorts as 0, so that the comparison stays a strict weak order whatever the vertices are.
static float centre_key(const TightBox& b, int axis) { const float c = b.lo[axis] + b.hi[axis]; return c == c ? c : 0.f; }

struct TopBuilder {
    std::vector<TopItem>& items;
    std::vector<DNode>& nodes; std::vector<TightBox>& tl; std::vector<TightBox>& tr; std::vector<NodeAux>& aux;
    int next_node = 0;
    std::vector<double> suffix;
    // builds the subtree over items[first, last) (at least two) into node `at`; returns its box, count and height
    void build(int at, int first, int last, TightBox* box_out, int* count_out, int* height_out, int depth = 0)
    {
        const int n = last - first;
        int best_axis = 0, best_split = first + n / 2;
        double best_cost = INFINITY;
        // (below 48 levels of lopsided splits the halves are taken by the median: the recursion stays shallow)
        for (int axis = 0; axis < 3 && n > 2; axis++) {
            std::sort(items.begin() + first, items.begin() + last, [axis](const TopItem& a, const TopItem& b) {
                const float ca = centre_key(a.box, axis), cb = centre_key(b.box, axis);
                return ca < cb || (ca == cb && a.ref < b.ref); });
            suffix.assign((size_t)n + 1, 0.0);
            TightBox acc = items[(size_t)last - 1].box; long long cnt = 0;
            for (int i = last - 1; i > first; i--) { acc = item_join(acc, items[(size_t)i].box); cnt += items[(size_t)i].count; suffix[(size_t)(i - first)] = item_area(acc) * (double)cnt; }
            acc = items[(size_t)first].box; cnt = 0;
            for (int i = first; i < last - 1; i++) {
                acc = item_join(acc, items[(size_t)i].box); cnt += items[(size_t)i].count;
                const double cost = item_area(acc) * (double)cnt + suffix[(size_t)(i + 1 - first)];
                if (cost < best_cost && (depth < 48 || axis == 0)) { best_cost = cost; best_axis = axis; best_split = depth < 48 ? i + 1 : first + n / 2; }
            }
        }
        if (n > 2) {
            const int axis = best_axis;
            std::sort(items.begin() + first, items.begin() + last, [axis](const TopItem& a, const TopItem& b) {
                const float ca = centre_key(a.box, axis), cb = centre_key(b.box, axis);
                return ca < cb || (ca == cb && a.ref < b.ref); });
        }
        TightBox cb[2]; int cc[2], ch[2], cref[2];
        const int lo[2] = {first, best_split}, hi[2] = {best_split, last};
        for (int side = 0; side < 2; side++) {
            if (hi[side] - lo[side] == 1) {
                const TopItem& it = items[(size_t)lo[side]];
                cref[side] = it.ref; cb[side] = it.box; cc[side] = it.count; ch[side] = it.height;
            } else {
                const int child = next_node++;
                cref[side] = child;
                build(child, lo[side], hi[side], &cb[side], &cc[side], &ch[side], depth + 1);
            }
        }
        DNode d{};
        for (int a = 0; a < 3; a++) { d.lmin[a] = cb[0].lo[a] - kPad; d.lmax[a] = cb[0].hi[a] + kPad; d.rmin[a] = cb[1].lo[a] - kPad; d.rmax[a] = cb[1].hi[a] + kPad; }
        d.left = cref[0]; d.right = cref[1];
        nodes[(size_t)at] = d; tl[(size_t)at] = cb[0]; tr[(size_t)at] = cb[1];
        NodeAux me;
        me.first = 0; me.count = cc[0] + cc[1]; me.height = 1 + (ch[0] > ch[1] ? ch[0] : ch[1]); me.mergeable = 0;
        aux[(size_t)at] = me;
        *box_out = item_join(cb[0], cb[1]); *count_out = me.count; *height_out = me.height;
    }
};

// Rebuilds nodes [0, n_top) (whole breadth-first levels) in place. d_nodes, d_tl, d_tr, d_aux are indexed by node position.
static int rebuild_top(DNode* d_nodes, TightBox* d_tl, TightBox* d_tr, NodeAux* d_aux, int n_top, int n_known, int min_count, hipStream_t stream)
{
#define GB_TRYR(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return PT_ERR_HIP; } } while (0)
    std::vector<DNode> nodes((size_t)n_top);
    std::vector<TightBox> tl((size_t)n_top), tr((size_t)n_top);
    std::vector<NodeAux> aux((size_t)n_known);
    GB_TRYR(hipMemcpyAsync(nodes.data(), d_nodes, nodes.size() * sizeof(DNode), hipMemcpyDeviceToHost, stream));
    GB_TRYR(hipMemcpyAsync(tl.data(), d_tl, tl.size() * sizeof(TightBox), hipMemcpyDeviceToHost, stream));
    GB_TRYR(hipMemcpyAsync(tr.data(), d_tr, tr.size() * sizeof(TightBox), hipMemcpyDeviceToHost, stream));
    GB_TRYR(hipMemcpyAsync(aux.data(), d_aux, aux.size() * sizeof(NodeAux), hipMemcpyDeviceToHost, stream));
    GB_TRYR(hipStreamSynchronize(stream));
    std::vector<TopItem> items;
    for (int p = 0; p < n_top; p++) {
        const int ref[2] = {nodes[(size_t)p].left, nodes[(size_t)p].right};
        const TightBox* box[2] = {&tl[(size_t)p], &tr[(size_t)p]};
        for (int side = 0; side < 2; side++) {
            if (ref[side] >= 0 && ref[side] < n_top) continue;         // a top node itself
            if (ref[side] >= n_known) { set_error("rebuild_top: child %d of node %d lies beyond the known levels", ref[side], p); return PT_ERR_HIP; }
            TopItem it;
            it.ref = ref[side]; it.box = *box[side];
            it.count = ref[side] < 0 ? 1 : aux[(size_t)ref[side]].count;
            it.height = ref[side] < 0 ? 0 : aux[(size_t)ref[side]].height;
            items.push_back(it);
        }
    }
    if ((int)items.size() != n_top + 1) { set_error("rebuild_top: %d items under %d nodes", (int)items.size(), n_top); return PT_ERR_HIP; }
    (void)min_count;
    std::vector<NodeAux> top_aux((size_t)n_top);
    TopBuilder tb{items, nodes, tl, tr, top_aux};
    tb.next_node = 1;
    TightBox rb; int rc_count = 0, rh = 0;
    tb.build(0, 0, (int)items.size(), &rb, &rc_count, &rh);
    if (tb.next_node != n_top) { set_error("rebuild_top: %d of %d nodes rebuilt", tb.next_node, n_top); return PT_ERR_HIP; }
    GB_TRYR(hipMemcpyAsync(d_nodes, nodes.data(), nodes.size() * sizeof(DNode), hipMemcpyHostToDevice, stream));
    GB_TRYR(hipMemcpyAsync(d_tl, tl.data(), tl.size() * sizeof(TightBox), hipMemcpyHostToDevice, stream));
    GB_TRYR(hipMemcpyAsync(d_tr, tr.data(), tr.size() * sizeof(TightBox), hipMemcpyHostToDevice, stream));
    GB_TRYR(hipMemcpyAsync(d_aux, top_aux.data(), top_aux.size() * sizeof(NodeAux), hipMemcpyHostToDevice, stream));
    GB_TRYR(hipStreamSynchronize(stream));
    return PT_OK;
#undef GB_TRYR
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
        if (n > 1) hipLaunchKernelGGL(gb_radix_tree, grid, block, 0, stream, d_keys_sorted, n, left, right, pin, pleaf, (int*)nullptr, (int*)nullptr);
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

// tris_dev: n >= 2 triangles on the device. Builds the tree (the radix tree over the Morton order, or the agglomerative
// tree over the same order) and every traversal layout on the device; tri_idx_host receives the triangle order of the
// leaves (n entries).
int gpu_build_scene(const pt_triangle* tris_dev, int n, int leaf_merge, bool agglomerative, bool top_sah, GpuScene* out, int* tri_idx_host, hipStream_t stream)
{
    if (!tris_dev || n < 2 || !out || !tri_idx_host) { set_error("gpu_build_scene: bad argument"); return PT_ERR_ARG; }
    int rc = PT_OK;
    const int ni = n - 1;          // interior nodes
    Bounds* d_bounds = nullptr;
    uint64_t *d_keys = nullptr, *d_keys_sorted = nullptr;
    uint32_t *d_ids = nullptr, *d_ids_sorted = nullptr, *d_misc = nullptr;
    int *d_links = nullptr, *d_order = nullptr, *d_newid = nullptr, *d_kids = nullptr, *d_off = nullptr, *d_items = nullptr, *d_tri_idx = nullptr, *d_need = nullptr;
    TightBox *d_tl = nullptr, *d_tr = nullptr, *d_clbox = nullptr;
    NodeAux* d_aux = nullptr;
    int *d_ploc = nullptr;          // cluster nodes (two generations), nearest neighbours, keep/fresh flags and their running sums
    uint32_t* d_ids_dfs = nullptr;
    int root = 0;
    DNode4* d_n4_big = nullptr;
    void *d_temp = nullptr, *d_scan_temp = nullptr;
    size_t temp_bytes = 0, scan_bytes = 0;
    GpuScene g;
    const dim3 grid((n + 255) / 256), block(256);
    auto blocks = [](int cnt) { return dim3((unsigned)((cnt + 255) / 256)); };
    std::vector<int> level_start;
    {
        GB_TRY(hipMalloc((void**)&d_bounds, sizeof(Bounds)));
        GB_TRY(hipMalloc((void**)&d_keys, (size_t)n * sizeof(uint64_t)));
        GB_TRY(hipMalloc((void**)&d_keys_sorted, (size_t)n * sizeof(uint64_t)));
        GB_TRY(hipMalloc((void**)&d_ids, (size_t)n * sizeof(uint32_t)));
        GB_TRY(hipMalloc((void**)&d_ids_sorted, (size_t)n * sizeof(uint32_t)));
        GB_TRY(hipMalloc((void**)&d_links, (size_t)n * 2 * sizeof(int)));
        GB_TRY(hipMalloc((void**)&d_misc, 4 * sizeof(uint32_t)));         // [0] largest coordinate (float bits), [1] largest stack need, [2] bad box
        Bounds init;
        for (int a = 0; a < 3; a++) { init.lo[a] = 0xffffffffu; init.hi[a] = 0u; }
        GB_TRY(hipMemcpyAsync(d_bounds, &init, sizeof(init), hipMemcpyHostToDevice, stream));
        GB_TRY(hipMemsetAsync(d_misc, 0, 4 * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(gb_scene_bounds, grid, block, 0, stream, tris_dev, n, d_bounds);
        hipLaunchKernelGGL(gb_morton, grid, block, 0, stream, tris_dev, n, d_bounds, d_keys, d_ids);
        GB_TRY(hipGetLastError());
        GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, n, 0, 63, stream));
        GB_TRY(hipMalloc(&d_temp, temp_bytes));
        GB_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, n, 0, 63, stream));
        int* left = d_links; int* right = d_links + n;
        if (!agglomerative) {
            hipLaunchKernelGGL(gb_radix_tree, grid, block, 0, stream, d_keys_sorted, n, left, right, (int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr);
            GB_TRY(hipGetLastError());
        }
        // the sort's buffers are free again: release them before the big arrays come
        GB_TRY(hipStreamSynchronize(stream));
        (void)hipFree(d_temp); d_temp = nullptr;
        (void)hipFree(d_keys); d_keys = nullptr;
        (void)hipFree(d_keys_sorted); d_keys_sorted = nullptr;
        (void)hipFree(d_ids); d_ids = nullptr;
        GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, n, stream));
        GB_TRY(hipMalloc(&d_scan_temp, scan_bytes));
        if (agglomerative) {
            GB_TRY(hipMalloc((void**)&d_ploc, (size_t)n * 7 * sizeof(int)));
            GB_TRY(hipMalloc((void**)&d_clbox, (size_t)n * 2 * sizeof(TightBox)));
            int* cl_node[2] = {d_ploc, d_ploc + n};
            int *nn = d_ploc + 2 * (size_t)n, *keep = d_ploc + 3 * (size_t)n, *keep_off = d_ploc + 4 * (size_t)n, *fresh = d_ploc + 5 * (size_t)n, *fresh_off = d_ploc + 6 * (size_t)n;
            TightBox* cl_box[2] = {d_clbox, d_clbox + n};
            hipLaunchKernelGGL(gb_ploc_init, grid, block, 0, stream, tris_dev, d_ids_sorted, n, cl_node[0], cl_box[0]);
            int m = n, made = 0, cur = 0;
            const int radius = kPlocRadius;
            for (int round = 0; m > 1; round++) {
                if (round > 4096) { set_error("gpu_build_scene: the agglomerative build did not converge"); rc = PT_ERR_HIP; goto done; }
                hipLaunchKernelGGL(gb_ploc_nearest, blocks(m), block, 0, stream, cl_box[cur], m, radius, nn);
                hipLaunchKernelGGL(gb_ploc_flags, blocks(m), block, 0, stream, nn, m, keep, fresh);
                GB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp, scan_bytes, keep, keep_off, m, stream));
                GB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp, scan_bytes, fresh, fresh_off, m, stream));
                hipLaunchKernelGGL(gb_ploc_apply, blocks(m), block, 0, stream, nn, m, keep, keep_off, fresh, fresh_off, made, cl_node[cur], cl_box[cur],
                                   cl_node[cur ^ 1], cl_box[cur ^ 1], left, right);
                GB_TRY(hipGetLastError());
                int last[4];
                GB_TRY(hipMemcpyAsync(&last[0], keep_off + (m - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
                GB_TRY(hipMemcpyAsync(&last[1], keep + (m - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
                GB_TRY(hipMemcpyAsync(&last[2], fresh_off + (m - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
                GB_TRY(hipMemcpyAsync(&last[3], fresh + (m - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
                GB_TRY(hipStreamSynchronize(stream));
                const int merged = last[2] + last[3];
                if (merged <= 0 || last[0] + last[1] != m - merged) { set_error("gpu_build_scene: agglomerative round %d merged %d of %d clusters", round, merged, m); rc = PT_ERR_HIP; goto done; }
                m -= merged; made += merged; cur ^= 1;
            }
            if (made != ni) { set_error("gpu_build_scene: %d of %d nodes made", made, ni); rc = PT_ERR_HIP; goto done; }
            GB_TRY(hipMemcpy(&root, cl_node[cur], sizeof(int), hipMemcpyDeviceToHost));
            if (root < 0 || root >= ni) { set_error("gpu_build_scene: bad root"); rc = PT_ERR_HIP; goto done; }
            (void)hipFree(d_clbox); d_clbox = nullptr;
        }

        // ---- breadth-first numbering
        GB_TRY(hipMalloc((void**)&d_order, (size_t)ni * sizeof(int)));
        GB_TRY(hipMalloc((void**)&d_newid, (size_t)ni * sizeof(int)));
        GB_TRY(hipMalloc((void**)&d_kids, (size_t)ni * sizeof(int)));
        GB_TRY(hipMalloc((void**)&d_off, (size_t)ni * sizeof(int)));
        GB_TRY(hipMalloc((void**)&d_need, (size_t)ni * sizeof(int)));
        GB_TRY(hipMemcpyAsync(d_order, &root, sizeof(int), hipMemcpyHostToDevice, stream));      // (the root of the radix tree is its node 0)
        GB_TRY(hipMemsetAsync(d_newid + root, 0, sizeof(int), stream));
        level_start.push_back(0);
        for (int lo = 0, cnt = 1; cnt > 0;) {
            if ((int)level_start.size() > 4096) { set_error("gpu_build_scene: tree deeper than 4096 levels"); rc = PT_ERR_ARG; goto done; }
            hipLaunchKernelGGL(gb_bfs_count, blocks(cnt), block, 0, stream, d_order, lo, cnt, left, right, d_kids);
            GB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp, scan_bytes, d_kids, d_off, cnt, stream));
            hipLaunchKernelGGL(gb_bfs_emit, blocks(cnt), block, 0, stream, d_order, lo, cnt, left, right, d_off, d_newid);
            GB_TRY(hipGetLastError());
            int last[2];
            GB_TRY(hipMemcpyAsync(&last[0], d_off + (cnt - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            GB_TRY(hipMemcpyAsync(&last[1], d_kids + (cnt - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            GB_TRY(hipStreamSynchronize(stream));
            lo += cnt; cnt = last[0] + last[1];
            level_start.push_back(lo);
            if (lo + cnt > ni) { set_error("gpu_build_scene: the radix tree is not a tree"); rc = PT_ERR_HIP; goto done; }
        }
        if (level_start.back() != ni) { set_error("gpu_build_scene: %d of %d nodes reached from the root", level_start.back(), ni); rc = PT_ERR_HIP; goto done; }

        if (agglomerative) {
            // leaf order: every subtree's triangles contiguous, left subtree first (the radix tree has that by construction)
            int *count = d_kids, *first = d_off, *leaf_pos = d_ploc;
            for (int l = (int)level_start.size() - 2; l >= 0; l--) {
                const int lo = level_start[(size_t)l], cnt = level_start[(size_t)l + 1] - lo;
                hipLaunchKernelGGL(gb_count_level, blocks(cnt), block, 0, stream, d_order, d_newid, left, right, lo, cnt, count);
            }
            GB_TRY(hipMemsetAsync(first, 0, sizeof(int), stream));
            for (int l = 0; l + 1 < (int)level_start.size(); l++) {
                const int lo = level_start[(size_t)l], cnt = level_start[(size_t)l + 1] - lo;
                hipLaunchKernelGGL(gb_first_level, blocks(cnt), block, 0, stream, d_order, d_newid, left, right, lo, cnt, count, first, leaf_pos);
            }
            GB_TRY(hipMalloc((void**)&d_ids_dfs, (size_t)n * sizeof(uint32_t)));
            hipLaunchKernelGGL(gb_remap_leaves, grid, block, 0, stream, n, leaf_pos, d_ids_sorted, d_ids_dfs, left, right);
            GB_TRY(hipGetLastError());
            GB_TRY(hipStreamSynchronize(stream));
            (void)hipFree(d_ids_sorted); d_ids_sorted = d_ids_dfs; d_ids_dfs = nullptr;
        }
        // ---- triangle records, bottom-up pass
        GB_TRY(hipMalloc((void**)&g.tris, ((size_t)n + 1) * sizeof(DTri)));
        GB_TRY(hipMemsetAsync(g.tris + n, 0, sizeof(DTri), stream));      // one record of padding: leaves are read two records at a time
        GB_TRY(hipMalloc((void**)&d_tri_idx, (size_t)n * sizeof(int)));
        hipLaunchKernelGGL(gb_emit_tris, grid, block, 0, stream, tris_dev, d_ids_sorted, n, g.tris, d_tri_idx, d_misc);
        GB_TRY(hipMalloc((void**)&g.nodes, (size_t)ni * sizeof(DNode)));
        GB_TRY(hipMalloc((void**)&d_tl, (size_t)ni * sizeof(TightBox)));
        GB_TRY(hipMalloc((void**)&d_tr, (size_t)ni * sizeof(TightBox)));
        GB_TRY(hipMalloc((void**)&d_aux, (size_t)ni * sizeof(NodeAux)));
        for (int l = (int)level_start.size() - 2; l >= 0; l--) {
            const int lo = level_start[(size_t)l], cnt = level_start[(size_t)l + 1] - lo;
            hipLaunchKernelGGL(gb_fit_level, blocks(cnt), block, 0, stream, tris_dev, d_ids_sorted, d_order, d_newid, left, right, lo, cnt, g.nodes, d_tl, d_tr, d_aux);
        }
        GB_TRY(hipGetLastError());
        {
            // the top of the tree: whole levels, as many as hold at most kTopNodes nodes, rebuilt with the surface-area sweep
            const int top_limit = kTopNodes;
            int levels = 0;
            while (levels + 2 < (int)level_start.size() && level_start[(size_t)levels + 1] <= top_limit) levels++;
            if (top_sah && levels >= 2) {
                const int n_top = level_start[(size_t)levels], n_known = level_start[(size_t)levels + 1];
                if ((rc = rebuild_top(g.nodes, d_tl, d_tr, d_aux, n_top, n_known, 0, stream)) != PT_OK) goto done;
            }
        }
        TightBox root_lr[2]; NodeAux root_aux; uint32_t misc[4];
        GB_TRY(hipMemcpyAsync(&root_lr[0], d_tl, sizeof(TightBox), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipMemcpyAsync(&root_lr[1], d_tr, sizeof(TightBox), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipMemcpyAsync(&root_aux, d_aux, sizeof(NodeAux), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipMemcpyAsync(misc, d_misc, sizeof(misc), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipMemcpyAsync(tri_idx_host, d_tri_idx, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipStreamSynchronize(stream));
        g.n_nodes = (size_t)ni;
        g.stack_need_binary = root_aux.height;          // one pending sibling per level of the binary walk
        QuantGrid qg;
        for (int a = 0; a < 3; a++) {
            g.root_min[a] = fminf(root_lr[0].lo[a], root_lr[1].lo[a]) - kPad;
            g.root_max[a] = fmaxf(root_lr[0].hi[a], root_lr[1].hi[a]) + kPad;
            double ext = (double)g.root_max[a] - g.root_min[a];
            if (!(ext > 0.0)) ext = 1.0;
            const double cell = ext / 65531.0;            // cells 2..65533 span the box, two cells of slack on each side
            g.q_origin[a] = (float)((double)g.root_min[a] - 2.0 * cell);
            g.q_cell[a] = (float)cell;
            g.q_inv_cell[a] = (float)(1.0 / cell);
            qg.origin[a] = g.q_origin[a]; qg.cell[a] = g.q_cell[a];
        }
        GB_TRY(hipMalloc((void**)&g.qnodes, (size_t)ni * sizeof(DNodeQ)));
        hipLaunchKernelGGL(gb_quantize, blocks(ni), block, 0, stream, g.nodes, ni, qg, g.qnodes);
        GB_TRY(hipGetLastError());

        // ---- 4-wide nodes (the root is interior: n >= 2 and a subtree becomes a leaf only below it... unless the whole tree may)
        float max_coord = 0.f;
        memcpy(&max_coord, &misc[0], sizeof(float));
        const double margin = ldexp(max_coord > 0.f ? (double)max_coord : 1.0, -19);
        if (root_aux.mergeable && root_aux.count <= leaf_merge && root_aux.count <= 16) { set_error("gpu_build_scene: the whole scene is one leaf"); rc = PT_ERR_ARG; goto done; }
        GB_TRY(hipMalloc((void**)&d_n4_big, (size_t)ni * sizeof(DNode4)));
        GB_TRY(hipMalloc((void**)&d_items, (size_t)ni * 6 * sizeof(int)));
        int* it[2][3] = {{d_items, d_items + ni, d_items + 2 * (size_t)ni}, {d_items + 3 * (size_t)ni, d_items + 4 * (size_t)ni, d_items + 5 * (size_t)ni}};
        GB_TRY(hipMemsetAsync(d_items, 0, (size_t)ni * 6 * sizeof(int), stream));        // first item: binary node 0 -> wide node 0, need 0
        int made = 1, cur = 0;
        for (int cnt = 1, level = 0; cnt > 0; level++) {
            if (level > 4096) { set_error("gpu_build_scene: wide tree deeper than 4096 levels"); rc = PT_ERR_ARG; goto done; }
            hipLaunchKernelGGL(gb_wide_make, blocks(cnt), block, 0, stream, it[cur][0], it[cur][1], it[cur][2], cnt, g.nodes, d_tl, d_tr, d_aux, leaf_merge, margin,
                               d_n4_big, d_kids, d_need, reinterpret_cast<int*>(d_misc) + 1);
            GB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp, scan_bytes, d_kids, d_off, cnt, stream));
            hipLaunchKernelGGL(gb_wide_link, blocks(cnt), block, 0, stream, it[cur][1], d_need, cnt, d_off, made, d_n4_big, it[cur ^ 1][0], it[cur ^ 1][1], it[cur ^ 1][2]);
            GB_TRY(hipGetLastError());
            int last[2];
            GB_TRY(hipMemcpyAsync(&last[0], d_off + (cnt - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            GB_TRY(hipMemcpyAsync(&last[1], d_kids + (cnt - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            GB_TRY(hipStreamSynchronize(stream));
            cnt = last[0] + last[1];
            if ((long long)made + cnt > ni) { set_error("gpu_build_scene: more wide nodes than binary nodes"); rc = PT_ERR_HIP; goto done; }
            made += cnt; cur ^= 1;
        }
        GB_TRY(hipMemcpyAsync(misc, d_misc, sizeof(misc), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipMalloc((void**)&g.nodes4, (size_t)made * sizeof(DNode4)));
        GB_TRY(hipMemcpyAsync(g.nodes4, d_n4_big, (size_t)made * sizeof(DNode4), hipMemcpyDeviceToDevice, stream));
        GB_TRY(hipStreamSynchronize(stream));
        g.n_nodes4 = (size_t)made;
        g.stack_need_wide = (int)misc[1];
        g.bad_box = misc[2] != 0;
    }
done:
    if (d_items) (void)hipFree(d_items);
    if (d_ploc) (void)hipFree(d_ploc);
    if (d_clbox) (void)hipFree(d_clbox);
    if (d_ids_dfs) (void)hipFree(d_ids_dfs);
    if (d_n4_big) (void)hipFree(d_n4_big);
    if (d_aux) (void)hipFree(d_aux);
    if (d_tr) (void)hipFree(d_tr);
    if (d_tl) (void)hipFree(d_tl);
    if (d_tri_idx) (void)hipFree(d_tri_idx);
    if (d_scan_temp) (void)hipFree(d_scan_temp);
    if (d_need) (void)hipFree(d_need);
    if (d_off) (void)hipFree(d_off);
    if (d_kids) (void)hipFree(d_kids);
    if (d_newid) (void)hipFree(d_newid);
    if (d_order) (void)hipFree(d_order);
    if (d_temp) (void)hipFree(d_temp);
    if (d_misc) (void)hipFree(d_misc);
    if (d_links) (void)hipFree(d_links);
    if (d_ids_sorted) (void)hipFree(d_ids_sorted);
    if (d_ids) (void)hipFree(d_ids);
    if (d_keys_sorted) (void)hipFree(d_keys_sorted);
    if (d_keys) (void)hipFree(d_keys);
    if (d_bounds) (void)hipFree(d_bounds);
    if (rc != PT_OK) {
        if (g.nodes) (void)hipFree(g.nodes);
        if (g.qnodes) (void)hipFree(g.qnodes);
        if (g.nodes4) (void)hipFree(g.nodes4);
        if (g.tris) (void)hipFree(g.tris);
        return rc;
    }
    *out = g;
    return PT_OK;
}

} // namespace ptamd
