// Host BVH construction for the render path.
//
// PT_BVH_REFERENCE produces, node for node and index for index, the tree that the reference's
// builder produces ("Cuda Pathtracer/core/BVH/BVH_builder.cpp:12-128"): centroid = (a+b+c)*0.3333f,
// node boxes padded by 0.1 per vertex, 99 uniform candidate planes per axis scored with
// count * half-area of the unpadded child boxes, split while the best score beats
// count * half-area of the padded parent, in-place two-pointer partition of the index array,
// children allocated as a consecutive pair in depth-first order.
//
// The reference scores every plane with a pass over all triangles of the node (297 passes per
// node).  Here each triangle is dropped once into the bin of the first plane it lies left of;
// prefix/suffix sweeps over the 100 bins then give every plane's counts and boxes.  Plane
// positions are monotone in the plane number, min/max are exact, and the scores are evaluated
// with the same float expression, so the chosen split is identical while the work per node drops
// from 297*n to 3*n.  Large nodes are binned by all threads; small subtrees are built as
// independent tasks and renumbered into the reference's allocation order afterwards.
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <utility>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../../include/pt_capi.h"
#include "pt_internal.h"

namespace ptamd {

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { lo[0] = lo[1] = lo[2] = 1e30f; hi[0] = hi[1] = hi[2] = -1e30f; }
    void grow(const pt_float3& p)
    {
        lo[0] = lo[0] < p.x ? lo[0] : p.x; lo[1] = lo[1] < p.y ? lo[1] : p.y; lo[2] = lo[2] < p.z ? lo[2] : p.z;
        hi[0] = hi[0] > p.x ? hi[0] : p.x; hi[1] = hi[1] > p.y ? hi[1] : p.y; hi[2] = hi[2] > p.z ? hi[2] : p.z;
    }
    void merge(const Box& b)
    {
        for (int k = 0; k < 3; k++) { lo[k] = lo[k] < b.lo[k] ? lo[k] : b.lo[k]; hi[k] = hi[k] > b.hi[k] ? hi[k] : b.hi[k]; }
    }
    float half_area() const
    {
        float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        return ex * ey + ey * ez + ez * ex;
    }
};

inline float axis_of(const pt_float3& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }

constexpr int kPlanes = 99;          // candidate planes 1..99 of BVH_builder.cpp:57
constexpr int kBins = kPlanes + 1;   // bin k holds triangles whose first "left of" plane is k+1; the last bin is never left

struct Bins {
    Box box[3][kBins];
    int cnt[3][kBins];
    void reset() { for (int a = 0; a < 3; a++) for (int k = 0; k < kBins; k++) { box[a][k].reset(); cnt[a][k] = 0; } }
    void merge(const Bins& o)
    {
        for (int a = 0; a < 3; a++) for (int k = 0; k < kBins; k++) { box[a][k].merge(o.box[a][k]); cnt[a][k] += o.cnt[a][k]; }
    }
};

// temporary node: children by pointer-like index into the pool, any allocation order
struct TNode {
    pt_float3 bmin, bmax;
    int first, count;     // index range while a leaf
    int left;             // pool index of the left child (right = left + 1), -1 for a leaf
};

struct Build {
    pt_triangle* tris;
    int* idx;
    std::vector<TNode> pool;
    int pool_used;        // guarded by omp atomic capture

    void node_bounds(TNode& n) const
    {
        Box b; b.reset();
        for (int i = 0; i < n.count; i++) {
            const pt_triangle& t = tris[idx[n.first + i]];
            b.grow(pt_float3{t.a.x - 0.1f, t.a.y - 0.1f, t.a.z - 0.1f});
            b.grow(pt_float3{t.b.x - 0.1f, t.b.y - 0.1f, t.b.z - 0.1f});
            b.grow(pt_float3{t.c.x - 0.1f, t.c.y - 0.1f, t.c.z - 0.1f});
            b.grow(pt_float3{t.a.x + 0.1f, t.a.y + 0.1f, t.a.z + 0.1f});
            b.grow(pt_float3{t.b.x + 0.1f, t.b.y + 0.1f, t.b.z + 0.1f});
            b.grow(pt_float3{t.c.x + 0.1f, t.c.y + 0.1f, t.c.z + 0.1f});
        }
        // the reference takes min over (v-0.1) and max over (v+0.1) separately; growing one box with
        // both gives the same extremes because v-0.1 <= v+0.1
        n.bmin = pt_float3{b.lo[0], b.lo[1], b.lo[2]};
        n.bmax = pt_float3{b.hi[0], b.hi[1], b.hi[2]};
    }

    static void planes_for(float bmin, float bmax, float planes[kPlanes])
    {
        float scale = (float)((double)(bmax - bmin) / 100.);
        for (int i = 1; i <= kPlanes; i++) planes[i - 1] = bmin + (float)i * scale;
    }

    // number of leading planes p with !(c < p): the triangle is "left" for every plane after those
    static int bin_of(const float planes[kPlanes], float c)
    {
        int lo = 0, hi = kPlanes;             // first index with c < planes[index]
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (c < planes[mid]) hi = mid; else lo = mid + 1;
        }
        return lo;                            // 0..99
    }

    void bin_range(const TNode& n, const float planes[3][kPlanes], const bool live[3], int begin, int end, Bins& out) const
    {
        for (int i = begin; i < end; i++) {
            const pt_triangle& t = tris[idx[n.first + i]];
            for (int a = 0; a < 3; a++) {
                if (!live[a]) continue;
                int k = bin_of(planes[a], axis_of(t.origin, a));
                Box& b = out.box[a][k];
                b.grow(t.a); b.grow(t.b); b.grow(t.c);
                out.cnt[a][k]++;
            }
        }
    }

    // Small nodes: score each plane with a pass over the node's triangles, as the reference does
    // (BVH_builder.cpp:28-65); cheaper than clearing and sweeping 300 bins when count is tiny.
    static constexpr int kDirectCount = 6;
    bool choose_split_direct(const TNode& n, const float planes[3][kPlanes], const bool live[3], int& axis, float& pos) const
    {
        float best = 1e30f;
        for (int a = 0; a < 3; a++) {
            if (!live[a]) continue;
            for (int i = 0; i < kPlanes; i++) {
                float cp = planes[a][i];
                Box lb, rb; lb.reset(); rb.reset();
                int lc = 0, rc = 0;
                for (int k = 0; k < n.count; k++) {
                    const pt_triangle& t = tris[idx[n.first + k]];
                    if (axis_of(t.origin, a) < cp) { lc++; lb.grow(t.a); lb.grow(t.b); lb.grow(t.c); }
                    else { rc++; rb.grow(t.a); rb.grow(t.b); rb.grow(t.c); }
                }
                float cost = lc * lb.half_area() + rc * rb.half_area();
                cost = cost > 0 ? cost : 1e30f;
                if (cost < best) { pos = cp; axis = a; best = cost; }
            }
        }
        float ex = n.bmax.x - n.bmin.x, ey = n.bmax.y - n.bmin.y, ez = n.bmax.z - n.bmin.z;
        float parentArea = ex * ey + ey * ez + ez * ex;
        float parentCost = n.count * parentArea;
        return !(best >= parentCost);
    }

    // returns true and fills axis/pos when the node should be split (BVH_builder.cpp:49-86)
    bool choose_split(const TNode& n, int& axis, float& pos, bool threaded) const
    {
        float planes[3][kPlanes];
        bool live[3];
        for (int a = 0; a < 3; a++) {
            float bmin = axis_of(n.bmin, a), bmax = axis_of(n.bmax, a);
            live[a] = !(bmin == bmax);
            if (live[a]) planes_for(bmin, bmax, planes[a]);
        }
        if (n.count <= kDirectCount) return choose_split_direct(n, planes, live, axis, pos);
        Bins bins; bins.reset();
#ifdef _OPENMP
        if (threaded && n.count > 65536) {
#pragma omp parallel
            {
                Bins local; local.reset();
                int nt = omp_get_num_threads(), id = omp_get_thread_num();
                long long b = (long long)n.count * id / nt, e = (long long)n.count * (id + 1) / nt;
                bin_range(n, planes, live, (int)b, (int)e, local);
#pragma omp critical
                bins.merge(local);
            }
        } else
#endif
        {
            bin_range(n, planes, live, 0, n.count, bins);
        }
        float best = 1e30f;
        for (int a = 0; a < 3; a++) {
            if (!live[a]) continue;
            // suffix boxes: right side of plane i (1-based) is bins i..99 (0-based bins hold k = first left plane index)
            Box rbox[kBins + 1]; int rcnt[kBins + 1];
            rbox[kBins].reset(); rcnt[kBins] = 0;
            for (int k = kBins - 1; k >= 0; k--) { rbox[k] = rbox[k + 1]; rbox[k].merge(bins.box[a][k]); rcnt[k] = rcnt[k + 1] + bins.cnt[a][k]; }
            Box lbox; lbox.reset(); int lcnt = 0;
            for (int i = 1; i <= kPlanes; i++) {
                // a triangle is left of plane i iff its bin index k (first plane index, 0-based) is <= i-1
                lbox.merge(bins.box[a][i - 1]); lcnt += bins.cnt[a][i - 1];
                int rc = rcnt[i];
                float cost = lcnt * lbox.half_area() + rc * rbox[i].half_area();
                cost = cost > 0 ? cost : 1e30f;
                if (cost < best) { pos = planes[a][i - 1]; axis = a; best = cost; }
            }
        }
        float ex = n.bmax.x - n.bmin.x, ey = n.bmax.y - n.bmin.y, ez = n.bmax.z - n.bmin.z;
        float parentArea = ex * ey + ey * ez + ez * ex;
        float parentCost = n.count * parentArea;
        return !(best >= parentCost);
    }

    // BVH_builder.cpp:88-109. Returns false when the node stays a leaf.
    bool split(int ni, bool threaded)
    {
        TNode n = pool[ni];
        int axis = 0; float pos = 0.f;
        if (!choose_split(n, axis, pos, threaded)) return false;
        int i = n.first, j = i + n.count - 1;
        while (i <= j) {
            if (axis_of(tris[idx[i]].origin, axis) < pos) i++;
            else std::swap(idx[i], idx[j--]);
        }
        int leftCount = i - n.first;
        if (leftCount == 0 || leftCount == n.count) return false;
        int l;
#pragma omp atomic capture
        { l = pool_used; pool_used += 2; }
        TNode& L = pool[l]; TNode& R = pool[l + 1];
        L.first = n.first; L.count = leftCount; L.left = -1;
        R.first = i; R.count = n.count - leftCount; R.left = -1;
        node_bounds(L); node_bounds(R);
        pool[ni].left = l;
        return true;
    }

    void build_subtree(int root)
    {
        std::vector<int> todo;
        todo.push_back(root);
        while (!todo.empty()) {
            int ni = todo.back(); todo.pop_back();
            if (split(ni, false)) { todo.push_back(pool[ni].left + 1); todo.push_back(pool[ni].left); }
        }
    }
};

} // namespace

int build_bvh_reference(pt_triangle* tris, int* idx, int n, pt_bvh_node* nodes_out, int* nodes_used_out)
{
    if (n <= 0) return PT_ERR_ARG;
    for (int i = 0; i < n; i++) {
        pt_triangle& t = tris[i];
        t.origin = pt_float3{(t.a.x + t.b.x + t.c.x) * 0.3333f, (t.a.y + t.b.y + t.c.y) * 0.3333f, (t.a.z + t.b.z + t.c.z) * 0.3333f};
    }
    Build B;
    B.tris = tris; B.idx = idx;
    B.pool.resize((size_t)2 * n - 1 + 2);
    B.pool_used = 1;
    B.pool[0].first = 0; B.pool[0].count = n; B.pool[0].left = -1;
    B.node_bounds(B.pool[0]);

    // phase 1: large nodes one at a time, binned by all threads
    std::vector<int> big, small;
    big.push_back(0);
    const int kTaskSize = 32768;
    while (!big.empty()) {
        int ni = big.back(); big.pop_back();
        if (B.pool[ni].count <= kTaskSize) { small.push_back(ni); continue; }
        if (B.split(ni, true)) { big.push_back(B.pool[ni].left + 1); big.push_back(B.pool[ni].left); }
    }
    // phase 2: independent subtrees
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = 0; k < (int)small.size(); k++) B.build_subtree(small[k]);

    // phase 3: renumber into the reference's allocation order (children pair allocated when the
    // parent splits, left subtree completed before the right one: BVH_builder.cpp:98-112)
    int used = 1;
    std::vector<std::pair<int, int>> walk;   // (pool node, output index)
    walk.push_back({0, 0});
    while (!walk.empty()) {
        auto [tn, on] = walk.back(); walk.pop_back();
        const TNode& t = B.pool[tn];
        pt_bvh_node& o = nodes_out[on];
        o.aabbMin = t.bmin; o.aabbMax = t.bmax;
        if (t.left < 0) { o.leftFirst = t.first; o.triCount = t.count; continue; }
        int l = used; used += 2;
        o.leftFirst = l; o.triCount = 0;
        walk.push_back({t.left + 1, l + 1});
        walk.push_back({t.left, l});
    }
    for (int i = used; i < 2 * n - 1; i++) { nodes_out[i] = pt_bvh_node{{0, 0, 0}, {0, 0, 0}, 0, 0}; }
    *nodes_used_out = used;
    return PT_OK;
}

} // namespace ptamd
