/*
 * oracle.cpp — CPU restatement of the reference render path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the shipped library includes, links or calls this file.  Only the
 * test-suite, the smoke check and the benchmark's cpu_baseline leg load it.
 *
 * It restates, function by function, what the files under "Cuda Pathtracer/core/CUDA" and
 * "core/BVH/BVH_builder.cpp" of the reference compute, in scalar host C++ with the
 * reference's operand types (float where the reference is float, double where a
 * double literal promotes a sub-expression) and its operation order.  Each function
 * cites the reference lines it follows.  The reference itself cannot be compiled in
 * this environment (it needs the CUDA toolkit, cuRAND and SFML headers), so this
 * restatement is pinned by: the Random123 Philox4x32-10 known-answer vectors, the
 * bunny BVH statistics recorded from a host compile of the reference (583 nodes,
 * depth 11), brute-force closest-hit cross checks, and hand-derived unit values.
 * The reference ships no tests or golden images; whatever those pins do not cover is
 * "parity unpinned".
 *
 * Deliberate deviations from the reference source, each needed to make the code
 * defined behaviour on a CPU:
 *   D1  compound-assignment operators return their result (float3_header.cuh:17-28 lack a return)
 *   D2  traversal stack of 64 entries instead of 10 (pathtracer.cu:194 overflows on deep trees)
 *   D3  bounce cap is a parameter (pathtracer.cu:405 hard-codes 512)
 *   D4  the kernel launch is a host loop over pixels (pathtracer.cu:602-606)
 *   D5  the two jitter draws are sequenced x first, then y (pathtracer.cu:562 leaves it unspecified)
 *   D6  closest hits carry the triangle index; exact-t ties go to the lowest index
 *       (pathtracer.cu:139 lets the last tested triangle win)
 *   D7  the camera basis is computed once per frame instead of once per pixel (pathtracer.cu:568-571)
 *   D8  a sky-mode switch selects the sun/sky gradient (pathtracer.cu:346-363, dead code in the
 *       reference) instead of the environment map; in that mode no light sampling is done
 *   D9  transcendental functions are the pinned algorithms of include/pt_detmath.h, rsqrtf is 1/sqrtf
 *   D10 texture lookups use a software bilinear sampler (wrap, normalized coordinates, texel
 *       centres at +0.5, fp32 weights) in place of the CUDA texture unit (image_loader.h:37-45)
 *   D11 rays with a non-finite origin or direction miss everything
 */
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <utility>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/pt_capi.h"
#include "../include/pt_detmath.h"

namespace {

// ---------------------------------------------------------------- vector helpers
// float3_header.cuh:5-123, float2_header.cuh:5-64 (component-wise fp32)
struct V3 { float x, y, z; };
struct V2 { float x, y; };
inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
inline V3 from(pt_float3 p) { return V3{p.x, p.y, p.z}; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
inline V3 operator+(V3 a, float b) { return mk(a.x + b, a.y + b, a.z + b); }
inline V3 operator-(V3 a, float b) { return mk(a.x - b, a.y - b, a.z - b); }
inline V3 operator*(V3 a, float b) { return mk(a.x * b, a.y * b, a.z * b); }
inline V3 operator/(V3 a, float b) { return mk(a.x / b, a.y / b, a.z / b); }
inline V3 operator+(float a, V3 b) { return mk(a + b.x, a + b.y, a + b.z); }
inline V3 operator-(float a, V3 b) { return mk(a - b.x, a - b.y, a - b.z); }
inline V3 operator*(float a, V3 b) { return mk(a * b.x, a * b.y, a * b.z); }
inline V3 operator/(float a, V3 b) { return mk(a / b.x, a / b.y, a / b.z); }
inline float dot3(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
inline V3 cross3(V3 p, V3 q) { return mk(p.y * q.z - q.y * p.z, p.z * q.x - q.z * p.x, p.x * q.y - q.x * p.y); }
inline float root(float x) { return __builtin_sqrtf(x); }
inline float inv_root(float x) { return 1.0f / __builtin_sqrtf(x); }          // D9
inline V3 unit(V3 a) { float s = inv_root(a.x * a.x + a.y * a.y + a.z * a.z); return mk(a.x * s, a.y * s, a.z * s); }
inline float len3(V3 a) { return root(a.x * a.x + a.y * a.y + a.z * a.z); }
inline float fabs_(float x) { return __builtin_fabsf(x); }
inline float fmin_(float a, float b) { return pt_fminf(a, b); }
inline float fmax_(float a, float b) { return pt_fmaxf(a, b); }
inline V3 lo3(V3 a, V3 b) { return mk(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); }   // float3_header.cuh:105-110
inline V3 hi3(V3 a, V3 b) { return mk(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }   // float3_header.cuh:111-116
inline float comp(V3 v, int n) { return n == 0 ? v.x : (n == 1 ? v.y : v.z); }          // cuda_utility.cuh:42-52
inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
inline V2 operator*(V2 a, float b) { return V2{a.x * b, a.y * b}; }

// constants.cuh:4-8 (literals kept, including the pi/4 that is named INV_4_PI)
const float kPI = 3.1415926535f;
const float kINV_PI = 0.318309f;
const float kTWO_PI = 6.283184f;
const float kINV_TWO_PI = 0.159154976203148f;
const float kINV_4_PI = 0.78539816339744830961566084581988f;

// cuda_utility.cuh:7-32
inline float clampf(float a, float l, float h) { return fmin_(h, fmax_(l, a)); }
inline float signf(float x) { if (x == 0) return 0; return x > 0. ? 1. : -1.; }
inline float smooth(float e0, float e1, float x) { float t = clampf((x - e0) / (e1 - e0), 0., 1.); return t * t * (3.0 - 2.0 * t); }
inline float lerp1(float a, float b, float m) { return a * (1. - m) + b * m; }
inline V3 lerp3(V3 a, V3 b, float m) { return mk(a.x * (1. - m) + b.x * m, a.y * (1. - m) + b.y * m, a.z * (1. - m) + b.z * m); }
// cuda_utility.cuh:53-78
inline void frame_of(V3 N, V3& T, V3& B)
{
    V3 up = fabs_(N.z) < 0.999 ? mk(0, 0, 1) : mk(1, 0, 0);
    T = unit(cross3(up, N));
    B = cross3(N, T);
}
inline V3 to_world(V3 X, V3 Y, V3 Z, V3 v) { return v.x * X + v.y * Y + v.z * Z; }
inline V3 to_local(V3 X, V3 Y, V3 Z, V3 v) { return mk(dot3(v, X), dot3(v, Y), dot3(v, Z)); }
inline V3 flip_toward(V3 v, V3 r) { float k = dot3(v, r); return k > 0.0f ? v : v - 2.0f * r * k; }
inline float power_heur(float a, float b) { float t = a * a; return t / (b * b + t); }
// float3_header.cuh:84-93
inline V3 mirror(V3 I, V3 N) { return I - N * 2.0f * dot3(N, I); }
inline V3 bend(V3 I, V3 N, float ior)
{
    float k = 1.0 - ior * ior * (1. - dot3(N, I) * dot3(N, I));
    if (k < 0.) return mk(0., 0., 0.);
    return I * ior - N * (ior * dot3(N, I) + root(k));
}

// ---------------------------------------------------------------- Philox4x32-10
// cuda_utility.cuh:3-4 + pathtracer.cu:560: curand_init(seed = pixel index, subsequence 0,
// offset 4096*frame) and curand_uniform.  Philox4x32-10 as published (Salmon et al., SC'11):
// key = (seed lo, seed hi); the 128-bit counter advances by one per four outputs, so an offset
// of 4096 draws starts at counter 1024*frame; outputs are consumed x, y, z, w; a draw maps to
// (0,1] as u*2^-32 + 2^-33.
inline void philox_round(uint32_t c[4], const uint32_t k[2])
{
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
inline void philox10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
struct Rng {
    uint32_t key[2];
    uint64_t block;      // low 64 bits of the counter
    uint32_t out[4];
    int pos;
    uint64_t draws;
    void seed(uint32_t pixel, int frame)
    {
        key[0] = pixel; key[1] = 0;
        block = (uint64_t)1024 * (uint64_t)(uint32_t)frame;
        pos = 0; draws = 0;
        refill();
    }
    void refill()
    {
        uint32_t c[4] = {(uint32_t)block, (uint32_t)(block >> 32), 0, 0};
        philox10(c, key, out);
    }
    float next()
    {
        uint32_t v = out[pos++];
        if (pos == 4) { block++; pos = 0; refill(); }
        draws++;
        return (float)v * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
    }
};

// ---------------------------------------------------------------- scene view
struct OrcTexture { const float* texels; int w, h, comps; };
struct Stats {
    uint64_t rays = 0, samples = 0, box_tests = 0, tri_tests = 0, shade_vertices = 0, env_samples = 0, tex_taps = 0, tex_taps_r = 0;
    void add(const Stats& o)
    {
        rays += o.rays; samples += o.samples; box_tests += o.box_tests; tri_tests += o.tri_tests;
        shade_vertices += o.shade_vertices; env_samples += o.env_samples; tex_taps += o.tex_taps; tex_taps_r += o.tex_taps_r;
    }
};
struct Scene {
    const pt_triangle* tris; int n_tris;
    const int* tri_idx;
    const pt_bvh_node* nodes; int n_nodes;
    const pt_material* mats; int n_mats;
    const OrcTexture* tex; int n_tex;
    const float* env_rgba; const float* env_cdf; int env_w, env_h; float env_sum;
    int sky_mode;
    V3 sun;
};

// material with texture handles stripped (structs.cuh:45-95)
struct Mat {
    V3 base, emis;
    float rough, aniso, metal, subsurf, sheen, coat, coat_rough, strans, alpha, ior;
    float G; V3 sigA, sigS;
};
struct HitRec { float t; uint32_t tri; float u, v; bool hit; };
struct Surf { float t; V3 n; Mat m; bool hit; bool inside; };
const uint32_t kNoTri = 0xFFFFFFFFu;

inline bool finite3(V3 a)
{
    return (a.x - a.x == 0.0f) && (a.y - a.y == 0.0f) && (a.z - a.z == 0.0f);
}

// ---------------------------------------------------------------- bilinear sampler (D10)
// Stands in for tex2D on a wrap / linear / normalized-coordinate texture object
// (image_loader.h:37-45, envmap.cpp:49-57,74-82).
inline int wrap_idx(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
inline void tex_fetch(const OrcTexture& T, float u, float v, float out[4], Stats* st)
{
    if (!(fabs_(u) < 65536.0f)) u = 0.0f;
    if (!(fabs_(v) < 65536.0f)) v = 0.0f;
    float x = u * (float)T.w - 0.5f;
    float y = v * (float)T.h - 0.5f;
    float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    float ax = x - fx, ay = y - fy;
    int ix = (int)fx, iy = (int)fy;
    int x0 = wrap_idx(ix, T.w), x1 = wrap_idx(ix + 1, T.w);
    int y0 = wrap_idx(iy, T.h), y1 = wrap_idx(iy + 1, T.h);
    const float* p00 = T.texels + ((size_t)y0 * T.w + x0) * T.comps;
    const float* p10 = T.texels + ((size_t)y0 * T.w + x1) * T.comps;
    const float* p01 = T.texels + ((size_t)y1 * T.w + x0) * T.comps;
    const float* p11 = T.texels + ((size_t)y1 * T.w + x1) * T.comps;
    float bx = 1.0f - ax, by = 1.0f - ay;
    for (int c = 0; c < T.comps; c++)
        out[c] = (p00[c] * bx + p10[c] * ax) * by + (p01[c] * bx + p11[c] * ax) * ay;
    if (st) { if (T.comps == 4) st->tex_taps++; else st->tex_taps_r++; }
}

// ---------------------------------------------------------------- intersection
// intersectors.cuh:59-71 (boxIntersectF)
inline float box_entry(V3 o, V3 inv, V3 bmin, V3 bmax)
{
    V3 p0 = -1.0 * (o - bmin) * inv;
    V3 p1 = -1.0 * (o - bmax) * inv;
    V3 nr = mk(fmin_(p0.x, p1.x), fmin_(p0.y, p1.y), fmin_(p0.z, p1.z));
    V3 fr = mk(fmax_(p0.x, p1.x), fmax_(p0.y, p1.y), fmax_(p0.z, p1.z));
    float tn = fmax_(fmax_(nr.x, nr.y), nr.z);
    float tf = fmin_(fmin_(fr.x, fr.y), fr.z);
    if (tf > tn && tf > 0) return tn;
    return 1e30;
}

// pathtracer.cu:124-139 (the geometric part of triangleIntersect). limit plays hit.t.
inline bool tri_geom(const pt_triangle& T, V3 o, V3 d, float limit, float& t, float& u, float& v)
{
    V3 e1 = from(T.b) - from(T.a);
    V3 e2 = from(T.c) - from(T.a);
    V3 h = cross3(d, e2);
    float a = dot3(e1, h);
    if (a > -0.0001f && a < 0.0001f) return false;
    float f = 1. / a;
    V3 s = o - from(T.a);
    u = f * dot3(s, h);
    if (u < 0. || u > 1) return false;
    V3 q = cross3(s, e1);
    v = f * dot3(d, q);
    if (v < 0 || u + v > 1) return false;
    t = f * dot3(e2, q);
    if (t < 0. || t > limit) return false;
    return true;
}

// Contract closest hit by exhaustive search (D6): minimum t over all triangles that pass the
// reference's tests, lowest index on ties.
HitRec closest_brute(const Scene& S, V3 o, V3 d)
{
    HitRec best{1e10f, kNoTri, 0.f, 0.f, false};
    if (!finite3(o) || !finite3(d)) return best;                       // D11
    for (int i = 0; i < S.n_tris; i++) {
        float t, u, v;
        if (!tri_geom(S.tris[i], o, d, best.t, t, u, v)) continue;
        if (best.hit && t == best.t) continue;
        best = HitRec{t, (uint32_t)i, u, v, true};
    }
    return best;
}

unsigned long long* g_depth_hist = nullptr;     // optional: histogram of stack depth at each push (diagnostics)

// pathtracer.cu:185-234 (BVHIntersect) with D2 and D6: same visiting order (nearer child first,
// farther child deferred, a deferred node dropped when its entry distance no longer beats the
// current hit), triangle index carried, ties to the lowest index.
HitRec closest_bvh(const Scene& S, V3 o, V3 d, Stats* st)
{
    HitRec best{1e10f, kNoTri, 0.f, 0.f, false};
    if (st) st->rays++;
    if (!finite3(o) || !finite3(d)) return best;                       // D11
    V3 inv = 1. / d;
    struct Entry { int node; float tn; };
    Entry stack[64];
    int sp = 0;
    const pt_bvh_node* N = S.nodes;
    uint64_t nb = 1, nt = 0;
    float troot = box_entry(o, inv, from(N[0].aabbMin), from(N[0].aabbMax));
    int cur = (troot < best.t) ? 0 : -1;
    while (cur >= 0) {
        const pt_bvh_node& nd = N[cur];
        cur = -1;
        if (nd.triCount > 0) {
            for (int i = 0; i < nd.triCount; i++) {
                int id = S.tri_idx[nd.leftFirst + i];
                float t, u, v;
                nt++;
                if (!tri_geom(S.tris[id], o, d, best.t, t, u, v)) continue;
                if (best.hit && t == best.t && (uint32_t)id > best.tri) continue;
                best = HitRec{t, (uint32_t)id, u, v, true};
            }
        } else {
            int l = nd.leftFirst, r = nd.leftFirst + 1;
            float dl = box_entry(o, inv, from(N[l].aabbMin), from(N[l].aabbMax));
            float dr = box_entry(o, inv, from(N[r].aabbMin), from(N[r].aabbMax));
            nb += 2;
            int nearN, farN; float dn, df;
            if (dl > dr) { nearN = r; dn = dr; farN = l; df = dl; }
            else         { nearN = l; dn = dl; farN = r; df = dr; }
            if (df < best.t) {
                if (sp >= 64) { fprintf(stderr, "oracle: traversal stack overflow\n"); abort(); }
                stack[sp++] = Entry{farN, df};
                if (g_depth_hist) __atomic_fetch_add(&g_depth_hist[sp], 1ull, __ATOMIC_RELAXED);
            }
            if (dn < best.t) cur = nearN;
        }
        while (cur < 0 && sp > 0) {
            Entry e = stack[--sp];
            if (e.tn < best.t) cur = e.node;
        }
    }
    if (st) { st->box_tests += nb; st->tri_tests += nt; }
    return best;
}

// pathtracer.cu:185-234 restated literally (only D2 applied): index stack, box re-test on pop,
// last accepted triangle wins. Used to measure how far D6 moves results from the reference.
HitRec closest_reference_order(const Scene& S, V3 o, V3 d)
{
    HitRec h{1e10f, kNoTri, 0.f, 0.f, false};
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    V3 inv = 1. / d;
    while (sp > 0) {
        const pt_bvh_node& nd = S.nodes[stack[--sp]];
        if (box_entry(o, inv, from(nd.aabbMin), from(nd.aabbMax)) < h.t) {
            if (nd.triCount > 0) {
                for (int i = 0; i < nd.triCount; i++) {
                    int id = S.tri_idx[i + nd.leftFirst];
                    float t, u, v;
                    if (tri_geom(S.tris[id], o, d, h.t, t, u, v)) h = HitRec{t, (uint32_t)id, u, v, true};
                }
            } else {
                const pt_bvh_node& cl = S.nodes[nd.leftFirst];
                const pt_bvh_node& cr = S.nodes[nd.leftFirst + 1];
                float dl = box_entry(o, inv, from(cl.aabbMin), from(cl.aabbMax));
                float dr = box_entry(o, inv, from(cr.aabbMin), from(cr.aabbMax));
                int l = nd.leftFirst, r = nd.leftFirst + 1;
                if (dl > dr) {
                    if (dl < h.t && sp < 64) stack[sp++] = l;
                    if (dr < h.t && sp < 64) stack[sp++] = r;
                } else {
                    if (dr < h.t && sp < 64) stack[sp++] = r;
                    if (dl < h.t && sp < 64) stack[sp++] = l;
                }
            }
        }
    }
    return h;
}

// pathtracer.cu:140-181: everything triangleIntersect does after accepting a hit. Only the
// final hit of a query influences the result, so it is evaluated once per query.
Surf resolve(const Scene& S, const HitRec& h, V3 o, V3 d, Stats* st)
{
    Surf s;
    s.t = h.t; s.hit = h.hit; s.inside = false;
    s.n = mk(0, 0, 0);
    // structs.cuh:75-94 default material (map() starts from Material())
    s.m.base = mk(0, 0, 0); s.m.emis = mk(0, 0, 0);
    s.m.rough = 0.01f; s.m.aniso = 0; s.m.metal = 0; s.m.subsurf = 0; s.m.sheen = 0; s.m.coat = 0; s.m.coat_rough = 0;
    s.m.strans = 0; s.m.alpha = 1.; s.m.ior = 1.5f; s.m.G = 0; s.m.sigA = mk(0, 0, 0); s.m.sigS = mk(0, 0, 0);
    if (!h.hit) return s;
    const pt_triangle& T = S.tris[h.tri];
    const pt_material& M = S.mats[T.matIndex];
    s.m.base = from(M.baseColor); s.m.emis = from(M.emissive);
    s.m.rough = M.roughness; s.m.aniso = M.anisotropic; s.m.metal = M.metallic; s.m.subsurf = M.subsurface;
    s.m.sheen = M.sheen; s.m.coat = M.clearcoat; s.m.coat_rough = M.clearcoatRoughness; s.m.strans = M.specTrans;
    s.m.alpha = M.alpha; s.m.ior = M.ior; s.m.G = M.medium.G; s.m.sigA = from(M.medium.sigmaA); s.m.sigS = from(M.medium.sigmaS);

    float u = h.u, v = h.v;
    float w = 1.f - u - v;
    V3 e1 = from(T.b) - from(T.a);
    V3 e2 = from(T.c) - from(T.a);
    V3 ng = unit(cross3(e1, e2));
    float mult = -signf(dot3(d, ng));
    V3 nf = unit(from(T.nA) * w + from(T.nB) * u + from(T.nC) * v) * mult;
    s.n = nf;
    V2 uv = V2{T.tA.x, T.tA.y} * w + V2{T.tB.x, T.tB.y} * u + V2{T.tC.x, T.tC.y} * v;
    float rd[4];
    if (M.useTexture) {
        tex_fetch(S.tex[M.diffuseTexture - 1], uv.x, uv.y, rd, st);
        s.m.base = mk(rd[0], rd[1], rd[2]) * rd[3];
        s.m.alpha = rd[3];
    }
    if (M.use_mapPr) { tex_fetch(S.tex[M.roughnessTexture - 1], uv.x, uv.y, rd, st); s.m.rough = rd[0]; }
    if (M.use_mapPm) { tex_fetch(S.tex[M.metallicTexture - 1], uv.x, uv.y, rd, st); s.m.metal = rd[0]; }
    if (M.use_mapKe) { tex_fetch(S.tex[M.emissiveTexture - 1], uv.x, uv.y, rd, st); s.m.emis = mk(rd[0], rd[0], rd[0]); }
    if (M.use_mapNor) {
        V3 Tn, Bn;
        frame_of(s.n, Tn, Bn);
        tex_fetch(S.tex[M.normalTexture - 1], uv.x, uv.y, rd, st);
        s.n = to_world(Tn, Bn, s.n, mk(rd[0] * 2.f - 1.f, rd[1] * 2.f - 1.f, rd[2] * 2.f - 1.f));
    }
    s.n = flip_toward(s.n, -1.0f * d);
    return s;
}

// pathtracer.cu:236-266 (map)
inline Surf trace(const Scene& S, V3 o, V3 d, Stats* st)
{
    HitRec h = closest_bvh(S, o, d, st);
    return resolve(S, h, o, d, st);
}

// ---------------------------------------------------------------- Disney BSDF, disney.cuh:29-460
namespace bsdf {
inline float lum(V3 c) { return 0.212671 * c.x + 0.715160 * c.y + 0.072169 * c.z; }                 // :29-32
inline float schlick_w(float u) { float m = clampf(1.0f - u, 0.0f, 1.0f); float m2 = m * m; return m2 * m2 * m; }  // :86-91
inline V3 ggx_vndf(V3 V, float ax, float ay, float r1, float r2)                                     // :41-59
{
    V3 Vh = unit(mk(ax * V.x, ay * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    V3 T1 = lensq > 0.0 ? mk(-Vh.y, Vh.x, 0.0) * inv_root(lensq) : mk(1, 0, 0);
    V3 T2 = cross3(Vh, T1);
    float r = root(r1);
    float phi = 2.0 * kPI * r2;
    float t1 = r * pt_cosf(phi);
    float t2 = r * pt_sinf(phi);
    float s = 0.5 * (1.0 + Vh.z);
    t2 = (1.0 - s) * root(1.0 - t1 * t1) + s * t2;
    double rem = 1.0 - t1 * t1 - t2 * t2;
    V3 Nh = t1 * T1 + t2 * T2 + root(rem > 0.0 ? rem : 0.0) * Vh;
    return unit(mk(ax * Nh.x, ay * Nh.y, fmax_(0.0, Nh.z)));
}
inline float fresnel_diel(float ci, float eta)                                                       // :61-75
{
    float st2 = eta * eta * (1.0f - ci * ci);
    if (st2 > 1.0f) return 1.0f;
    double rem = 1.0 - st2;
    float ct = root(rem > 0.0 ? rem : 0.0);
    float rs = (eta * ct - ci) / (eta * ct + ci);
    float rp = (eta * ci - ct) / (eta * ci + ct);
    return 0.5f * (rs * rs + rp * rp);
}
inline V3 cos_hemi(float r1, float r2)                                                               // :76-85
{
    V3 dir;
    float r = root(r1);
    float phi = kTWO_PI * r2;
    dir.x = r * pt_cosf(phi);
    dir.y = r * pt_sinf(phi);
    dir.z = root(fmax_(0.0, 1.0 - dir.x * dir.x - dir.y * dir.y));
    return dir;
}
inline void tints(const Mat& m, float eta, float& F0, V3& Csheen, V3& Cspec0)                        // :92-102
{
    F0 = (1.0 - eta) / (1.0 + eta);
    F0 *= F0;
    Cspec0 = F0 * mk(1.f, 1.f, 1.f);
    Csheen = mk(1.f, 1.f, 1.f);
}
inline float gtr2_aniso(float nh, float hx, float hy, float ax, float ay)                            // :103-109
{
    float a = hx / ax, b = hy / ay;
    float c = a * a + b * b + nh * nh;
    return 1.0 / (kPI * ax * ay * c * c);
}
inline float smith_g(float nv, float ag)                                                             // :110-115
{
    float a = ag * ag, b = nv * nv;
    return (2.0 * nv) / (nv + root(a + b - a * b));
}
inline float smith_g_aniso(float nv, float vx, float vy, float ax, float ay)                         // :116-122
{
    float a = vx * ax, b = vy * ay, c = nv;
    return (2.0 * nv) / (nv + root(a * a + b * b + c * c));
}
inline float gtr1(float nh, float a)                                                                 // :123-130
{
    if (a >= 1.0) return kINV_PI;
    float a2 = a * a;
    float t = 1.0 + (a2 - 1.0) * nh * nh;
    return (a2 - 1.0) / (kPI * pt_logf(a2) * t);
}
inline V3 sample_gtr1(float rgh, float r1, float r2)                                                 // :132-145
{
    float a = fmax_(0.001, rgh);
    float a2 = a * a;
    float phi = r1 * kTWO_PI;
    float ct = root((1.0 - pt_dm_pow(a2, 1.0 - r2)) / (1.0 - a2));
    float st = clampf(pt_dm_sqrt(1.0 - (ct * ct)), 0.0, 1.0);
    float sp = pt_sinf(phi), cp = pt_cosf(phi);
    return mk(st * cp, st * sp, ct);
}
inline V3 eval_diffuse(const Mat& m, V3 Csheen, V3 V, V3 L, V3 H, float& pdf)                        // :154-181
{
    pdf = 0.0;
    if (L.z <= 0.0) return mk(0, 0, 0);
    float lh = dot3(L, H);
    float Rr = 2.0f * m.rough * lh * lh;
    float FL = schlick_w(L.z), FV = schlick_w(V.z);
    float Fretro = Rr * (FL + FV + FL * FV * (Rr - 1.0f));
    float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
    float Fss90 = 0.5 * Rr;
    float Fss = lerp1(1.0f, Fss90, FL) * lerp1(1.0f, Fss90, FV);
    float ss = 1.25f * (Fss * (1.0f / (L.z + V.z) - 0.5f) + 0.5f);
    float FH = schlick_w(lh);
    V3 Fsheen = FH * m.sheen * Csheen;
    pdf = L.z * kINV_PI;
    return kINV_PI * m.base * lerp1(Fd + Fretro, ss, m.subsurf) + Fsheen;
}
inline V3 eval_reflect(V3 V, V3 L, V3 H, V3 F, float ax, float ay, float& pdf)                       // :183-195
{
    pdf = 0.0;
    if (L.z <= 0.0) return mk(0, 0, 0);
    float D = gtr2_aniso(H.z, H.x, H.y, ax, ay);
    float G1 = smith_g_aniso(fabs_(V.z), V.x, V.y, ax, ay);
    float G2 = G1 * smith_g_aniso(fabs_(L.z), L.x, L.y, ax, ay);
    pdf = G1 * D / (4.0 * V.z);
    return F * D * G2 / (float)(4.0 * L.z * V.z);
}
inline V3 pow3(V3 a, float p) { return mk(pt_powf(a.x, p), pt_powf(a.y, p), pt_powf(a.z, p)); }     // float3_header.cuh:98-100
inline V3 eval_refract(const Mat& m, float eta, V3 V, V3 L, V3 H, V3 F, float ax, float ay, float& pdf) // :197-216
{
    pdf = 0.0;
    if (L.z >= 0.0) return mk(0, 0, 0);
    float lh = dot3(L, H), vh = dot3(V, H);
    float D = gtr2_aniso(H.z, H.x, H.y, ax, ay);
    float G1 = smith_g_aniso(fabs_(V.z), V.x, V.y, ax, ay);
    float G2 = G1 * smith_g_aniso(fabs_(L.z), L.x, L.y, ax, ay);
    float denom = lh + vh * eta;
    denom *= denom;
    float eta2 = eta * eta;
    float jac = fabs_(lh) / denom;
    pdf = G1 * fmax_(0.0, vh) * D * jac / V.z;
    return pow3(m.base, 0.5f) * (1.0f - F) * D * G2 * fabs_(vh) * jac * eta2 / fabs_(L.z * V.z);
}
inline V3 eval_coat(const Mat& m, V3 V, V3 L, V3 H, float& pdf)                                      // :218-233
{
    pdf = 0.0;
    if (L.z <= 0.0) return mk(0, 0, 0);
    float vh = dot3(V, H);
    float F = lerp1(0.04, 1.0, schlick_w(vh));
    float D = gtr1(H.z, m.coat_rough);
    float G = smith_g(L.z, 0.25) * smith_g(V.z, 0.25f);
    float jac = 1.0 / (4.0 * vh);
    pdf = D * H.z * jac;
    return mk(F, F, F) * D * G;
}
V3 eval(const Surf& s, V3 V, V3 N, V3 L, float& pdf)                                                 // :234-347 (DisneyEval_2)
{
    const Mat& m = s.m;
    pdf = 0.0;
    float eta = s.inside ? m.ior : 1.f / m.ior;
    V3 f = mk(0, 0, 0);
    float aspect = inv_root(1.f - 0.9f * m.aniso);
    float ax = m.rough * m.rough / aspect;
    float ay = m.rough * m.rough * aspect;
    V3 T, B;
    frame_of(N, T, B);
    V = to_local(T, B, N, V);
    L = to_local(T, B, N, L);
    V3 H;
    if (L.z > 0.0) H = unit(L + V);
    else H = unit(L + V * eta);
    if (H.z < 0.0) H = -1.f * H;
    V3 Csheen, Cspec0; float F0;
    tints(m, eta, F0, Csheen, Cspec0);
    float w_diel = (1.0f - m.metal) * (1.0f - m.strans);
    float w_metal = m.metal;
    float w_glass = (1.0f - m.metal) * m.strans;
    float sw = schlick_w(V.z);
    float p_diff = w_diel * lum(m.base);
    float p_diel = w_diel * lum(lerp3(Cspec0, mk(1, 1, 1), sw));
    float p_metal = w_metal * lum(lerp3(m.base, mk(1, 1, 1), sw));
    float p_glass = w_glass;
    float p_coat = 0.25f * m.coat;
    float inv_tot = 1.0f / (p_diff + p_diel + p_metal + p_glass + p_coat);
    p_diff *= inv_tot; p_diel *= inv_tot; p_metal *= inv_tot; p_glass *= inv_tot; p_coat *= inv_tot;
    bool refl = L.z * V.z > 0;
    float tp = 0.0;
    float vh = fabs_(dot3(V, H));
    if (p_diff > 0.0 && refl) {
        f = f + eval_diffuse(m, Csheen, V, L, H, tp) * w_diel;
        pdf += tp * p_diff;
    }
    if (p_diel > 0.0 && refl) {
        float F = (fresnel_diel(vh, 1.0 / m.ior) - F0) / (1.0 - F0);
        f = f + eval_reflect(V, L, H, lerp3(Cspec0, mk(1, 1, 1), F), ax, ay, tp) * w_diel;
        pdf += tp * p_diel;
    }
    if (p_metal > 0.0 && refl) {
        V3 F = lerp3(m.base, mk(1, 1, 1), schlick_w(vh));
        f = f + eval_reflect(V, L, H, F, ax, ay, tp) * w_metal;
        pdf += tp * p_metal;
    }
    if (p_glass > 0.0 && fabs_(m.ior - 1.f) > 0.01f) {
        float F = fresnel_diel(vh, eta);
        if (refl) {
            f = f + eval_reflect(V, L, H, mk(F, F, F), ax, ay, tp) * w_glass;
            pdf += tp * p_glass * F;
        } else {
            f = f + eval_refract(m, eta, V, L, H, mk(F, F, F), ax, ay, tp) * w_glass;
            pdf += tp * p_glass * (1.0f - F);
        }
    }
    if (p_coat > 0.0 && refl) {
        f = f + eval_coat(m, V, L, H, tp) * 0.25f * m.coat;
        pdf += tp * p_coat;
    }
    return f * fabs_(L.z);
}
V3 sample(const Surf& s, V3 V, V3 N, V3& L, float& pdf, Rng& rng, bool& inside)                      // :348-460 (DisneySample_2)
{
    const Mat& m = s.m;
    pdf = 0.0;
    float eta = s.inside ? m.ior : 1.f / m.ior;
    float aspect = inv_root(1.f - 0.9f * m.aniso);
    float ax = m.rough * m.rough / aspect;
    float ay = m.rough * m.rough * aspect;
    float r1 = rng.next();
    float r2 = rng.next();
    V3 T, B;
    frame_of(N, T, B);
    V = to_local(T, B, N, V);
    V3 Csheen, Cspec0; float F0;
    tints(m, eta, F0, Csheen, Cspec0);
    float w_diel = (1.0 - m.metal) * (1.0 - m.strans);
    float w_metal = m.metal;
    float w_glass = (1.0 - m.metal) * m.strans;
    float sw = schlick_w(V.z);
    float p_diff = w_diel * lum(m.base);
    float p_diel = w_diel * lum(lerp3(Cspec0, mk(1, 1, 1), sw));
    float p_metal = w_metal * lum(lerp3(m.base, mk(1, 1, 1), sw));
    float p_glass = w_glass;
    float p_coat = 0.25 * m.coat;
    float inv_tot = 1.0 / (p_diff + p_diel + p_metal + p_glass + p_coat);
    p_diff *= inv_tot; p_diel *= inv_tot; p_metal *= inv_tot; p_glass *= inv_tot; p_coat *= inv_tot;
    float cdf[5];
    cdf[0] = p_diff;
    cdf[1] = cdf[0] + p_diel;
    cdf[2] = cdf[1] + p_metal;
    cdf[3] = cdf[2] + p_glass;
    cdf[4] = cdf[3] + p_coat;
    float r3 = rng.next();
    if (r3 < cdf[0]) {
        L = cos_hemi(r1, r2);
    } else if (r3 < cdf[2]) {
        V3 H = ggx_vndf(V, ax, ay, r1, r2);
        if (H.z < 0.0) H = -1.f * H;
        L = unit(mirror(-1.f * V, H));
    } else if (r3 < cdf[3]) {
        if (fabs_(m.ior - 1.f) <= 0.01f) {
            L = to_world(T, B, N, -1.f * V);
            pdf = 1.f;
            inside = !inside;
            return mk(1.f, 1.f, 1.f);
        }
        V3 H = ggx_vndf(V, ax, ay, r1, r2);
        float F = fresnel_diel(fabs_(dot3(V, H)), eta);
        if (H.z < 0.0) H = -1.f * H;
        r3 = (r3 - cdf[2]) / (cdf[3] - cdf[2]);
        if (r3 < F) {
            L = unit(mirror(-1.f * V, H));
        } else {
            L = unit(bend(-1.f * V, H, eta));
            inside = !inside;
        }
    } else {
        V3 H = sample_gtr1(m.coat_rough, r1, r2);
        if (H.z < 0.0) H = -1.f * H;
        L = unit(mirror(-1.f * V, H));
    }
    L = to_world(T, B, N, L);
    V = to_world(T, B, N, V);
    return eval(s, V, N, L, pdf);
}
} // namespace bsdf

// ---------------------------------------------------------------- phase function, phase_function.cu:5-20
inline float hg_eval(float mu, float g)
{
    return kINV_4_PI * (1. - g * g) / (pt_powf(1. + g * g - 2.0f * g * mu, 1.5f));
}
inline V3 hg_sample(V3 d, float g, Rng& rng)
{
    float xi = rng.next();
    float t = (1.f - g * g) / (1.f - g + 2.0f * g * xi);
    float mu = (0.5f / g) * ((1.f + g * g) - t * t);
    float phi = kTWO_PI * rng.next();
    float st = root(fmax_(0.f, 1.f - mu * mu));
    V3 T, B;
    frame_of(d, T, B);
    return unit(st * pt_cosf(phi) * T + st * pt_sinf(phi) * B + mu * d);
}

// ---------------------------------------------------------------- environment, envmap_cuda.cu:12-74
struct Env4 { V3 c; float pdf; };
inline Env4 env_eval(const Scene& S, V3 dir, Stats* st)                                              // :12-22
{
    float theta = pt_acosf(clampf(dir.y, -1.0f, 1.0f));
    float u = (kPI + pt_atan2f(dir.z, dir.x)) * kINV_TWO_PI;
    float v = theta * kINV_PI;
    OrcTexture T{S.env_rgba, S.env_w, S.env_h, 4};
    float rd[4];
    tex_fetch(T, u, v, rd, st);
    V3 col = mk(rd[0], rd[1], rd[2]);
    float pdf = bsdf::lum(col) / S.env_sum;
    return Env4{col, (pdf * S.env_w * S.env_h) / (kTWO_PI * kPI * pt_sinf(theta))};
}
inline V2 env_search(const Scene& S, float xi, Stats* st)                                            // :28-57
{
    OrcTexture C{S.env_cdf, S.env_w, S.env_h, 1};
    float rd[4];
    int lower = 0, upper = S.env_h - 1;
    while (lower < upper) {
        int mid = (lower + upper) >> 1;
        tex_fetch(C, 1., (float)mid / (float)S.env_h, rd, st);
        if (xi < rd[0]) upper = mid; else lower = mid + 1;
    }
    int y = (int)clampf(lower, 0, S.env_h - 1);
    lower = 0; upper = S.env_w - 1;
    while (lower < upper) {
        int mid = (lower + upper) >> 1;
        tex_fetch(C, (float)mid / S.env_w, (float)y / (float)S.env_h, rd, st);
        if (xi < rd[0]) upper = mid; else lower = mid + 1;
    }
    int x = (int)clampf(lower, 0, S.env_w - 1);
    return V2{(float)x / (float)S.env_w, (float)y / (float)S.env_h};
}
inline Env4 env_sample(const Scene& S, V3& col, Rng& rng, Stats* st)                                 // :59-74; returns direction in .c
{
    V2 uv = env_search(S, rng.next() * S.env_sum, st);
    OrcTexture T{S.env_rgba, S.env_w, S.env_h, 4};
    float rd[4];
    tex_fetch(T, uv.x, uv.y, rd, st);
    col = mk(rd[0], rd[1], rd[2]);
    float pdf = bsdf::lum(col) / S.env_sum;
    float phi = uv.x * kTWO_PI;
    float theta = uv.y * kPI;
    float sn = pt_sinf(theta);
    if (sn == 0.f) pdf = 0.f;
    if (st) st->env_samples++;
    return Env4{mk(-sn * pt_cosf(phi), pt_cosf(theta), -sn * pt_sinf(phi)), (pdf * S.env_w * S.env_h) / (kTWO_PI * kPI * sn)};
}

// pathtracer.cu:340-363 (HG, rayleigh, skyGradient), selected by D8
inline float sky_hg(float g, float c) { float gg = g * g; return (1. - gg) / pt_dm_pow(1. + gg - 2. * g * c, 1.5); }
inline float sky_rayleigh(float u) { return 0.75 * (1. + clampf(u * u, 0., 1.)); }
V3 sky_gradient(V3 rd, V3 sun)
{
    V3 sunColor = lerp3(mk(1., 0.9, 0.8), mk(1., 0.2, 0.), pt_dm_exp(-fabs_(sun.y) * 5.));
    float c0 = dot3(rd, sun);
    float blend = smooth(-1., 0.6, c0);
    V3 day = lerp3(mk(0.3, 0.6, 1.) * 0.7 * sky_rayleigh(c0), mk(0.9, 0.95, 1.), pt_dm_exp(-fabs_(rd.y) * 5.));
    V3 even = lerp3(mk(0.3, 0.6, 1.) * 0.7, lerp3(mk(0.2, 0.3, 0.4), mk(1., 0.2, 0.), blend), pt_dm_exp(-fabs_(rd.y) * 5.));
    V3 sunDisk = (float)(0.02 / len3(rd - sun)) * sunColor;
    V3 night = mk(0.05, 0.1, 0.2) + clampf(0.02 / len3(rd + sun), 0., 1.) * mk(0.7, 0.8, 1.) * 0.6;
    V3 dayF = lerp3(day, even, pt_dm_exp(-fabs_(sun.y) * 5.)) + sunDisk;
    float earth = smooth(-0.02, 0.02, rd.y) * 0.5 + 0.5;
    float mieM = 1. - fabs_(dot3(sun, mk(0., 1., 0.))) + 0.3;
    return lerp3(night, dayF + sky_hg(0.6, c0) * sunColor * 0.3 * mieM, smooth(-0.2, 0.0, sun.y)) * earth * 0.75;
}

// ---------------------------------------------------------------- integrator
// pathtracer.cu:267-297 (visibility)
V3 shadow_walk(const Scene& S, V3 o, V3 d, Rng& rng, bool inside, Stats* st)
{
    V3 light = mk(1.f, 1.f, 1.f);
    for (int i = 0; i < 32; i++) {
        Surf info = trace(S, o, d, st);
        if (!info.hit) return light;
        bool refractive = (1.f - info.m.metal) * info.m.strans > 0.f && fabs_(info.m.ior - 1.f) <= 0.01f;
        bool alpha = rng.next() > info.m.alpha;
        if (inside) {
            V3 a = -info.t * (info.m.sigA + info.m.sigS);
            light = light * mk(pt_expf(a.x), pt_expf(a.y), pt_expf(a.z));
        }
        if (info.hit && !(refractive || alpha)) return mk(0.f, 0.f, 0.f);
        else if (refractive) inside = !inside;
        o = o + d * (info.t + 0.1f);
    }
    return light;
}

// pathtracer.cu:298-339 (sampleSkyboxOnBounce)
V3 light_sample(const Scene& S, const Surf& info, V3 o, V3 d, Rng& rng, bool isSurface, Stats* st)
{
    if (S.sky_mode == PT_SKY_SUNSKY) return mk(0, 0, 0);                                             // D8
    V3 color;
    Env4 e = env_sample(S, color, rng, st);
    V3 L = e.c;
    float L_pdf = e.pdf;
    V3 trans = shadow_walk(S, o, L, rng, !isSurface, st);
    if (dot3(trans, trans) <= 0.05f) return mk(0, 0, 0);
    float bp;
    V3 bf;
    if (isSurface) {
        bf = bsdf::eval(info, -1.f * d, info.n, L, bp);
    } else {
        bp = hg_eval(dot3(L, d), info.m.G);
        bf = bp * mk(1, 1, 1);
    }
    if (bp > 0.f) {
        float mis = power_heur(L_pdf, bp);
        if (mis > 0.f) return mis * color * bf * trans / L_pdf;
    }
    return mk(0, 0, 0);
}

// pathtracer.cu:381-493 (pathtrace): one colour channel per sample
void integrate(float& result, const Scene& S, V3 o, V3 d, Rng& rng, int ch, int max_bounces, Stats* st)
{
    float thr = 1.0;
    bool surfaceScatter = false, inside = false, mediumScatter = false;
    float pdf = 0.f;
    for (int i = 0; i < max_bounces; i++) {
        Surf info = trace(S, o, d, st);
        info.inside = inside;
        if (!info.hit) {
            V3 envc; float envp;
            if (S.sky_mode == PT_SKY_SUNSKY) { envc = sky_gradient(d, S.sun); envp = 0.f; }          // D8
            else { Env4 e = env_eval(S, d, st); envc = e.c; envp = e.pdf; }
            float mis = 1.f;
            if (i > 0) mis = power_heur(pdf, envp);
            if (!surfaceScatter) mis = 1.f;
            if (S.sky_mode == PT_SKY_SUNSKY) mis = 1.f;
            if (mis > 0.f) result += comp(mis * envc * thr, ch);
            return;
        }
        if (st) st->shade_vertices++;
        info.m.rough = fmax_(info.m.rough, 0.0001f);
        mediumScatter = false;
        surfaceScatter = false;
        if (info.inside) {
            V3 sigT = info.m.sigA + info.m.sigS;
            float zeta = rng.next();
            if (zeta < comp(info.m.sigA / sigT, ch)) {
                thr *= pt_expf(-comp(sigT * info.t, ch));
            } else {
                float sd = -pt_logf(1.f - rng.next()) / comp(sigT, ch);
                if (sd < info.t) {
                    mediumScatter = true;
                    o = o + d * sd;
                    result += comp(light_sample(S, info, o, d, rng, false, st), ch) * thr;
                    V3 mem = d;
                    d = hg_sample(d, info.m.G, rng);
                    pdf = hg_eval(dot3(mem, d), info.m.G);
                }
            }
        }
        if (!mediumScatter) {
            o = o + d * info.t;
            float rC = rng.next();
            if (rC < info.m.alpha) {
                surfaceScatter = true;
                result += comp(light_sample(S, info, o + info.n * 0.01f, d, rng, true, st), ch) * thr;
                V3 L;
                V3 f = bsdf::sample(info, d * -1.f, info.n, L, pdf, rng, inside);
                d = L;
                result += comp(info.m.emis, ch) * thr;
                if (pdf > 0.f) thr *= comp(f / pdf, ch);
                else return;
            }
            o = o + d * 0.01f;
        }
        {
            float p = thr;
            if (rng.next() > p) return;
            thr *= 1.0f / p;
        }
    }
}

struct Camera { V3 origin, ww, uu, vv; };

// pathtracer.cu:567-571 (D7)
Camera make_camera(const pt_params& P)
{
    Camera c;
    c.origin = from(P.rayOrigin);
    float ax = P.cameraAngle.x, ay = P.cameraAngle.y;
    V3 target = c.origin + mk(pt_sinf(ax) * pt_cosf(ay), pt_sinf(ay), -pt_cosf(ax) * pt_cosf(ay));
    c.ww = unit(target - c.origin);
    c.uu = unit(cross3(c.ww, mk(0.f, 1.f, 0.f)));
    c.vv = unit(cross3(c.uu, c.ww));
    return c;
}

// cuda_utility.cuh:33-40
inline V3 sphere_sample(Rng& rng)
{
    float z = rng.next() * 2.0f - 1.0f;
    float a = rng.next() * kPI * 2.;
    float r = root(1.0f - z * z);
    return mk(r * pt_cosf(a), r * pt_sinf(a), z);
}

// pathtracer.cu:544-580 + 494-543: one sample of one pixel, before accumulation
V3 pixel_sample(const Scene& S, const pt_params& P, const Camera& cam, int x, int y, int frame, int max_bounces, Stats* st)
{
    int W = P.windowSize.x, H = P.windowSize.y;
    int idx = x + y * W;
    Rng rng;
    rng.seed((uint32_t)idx, frame);
    float jx = rng.next();                                                                            // D5
    float jy = rng.next();
    float uvx = ((float)x + jx * 2.0f - 1.f) / W;
    float uvy = 1.f - ((float)y + jy * 2.0f - 1.f) / H;
    float cx = uvx - 0.5f, cy = uvy - 0.5f;
    cx *= (float)W / H;
    V3 origin = cam.origin;
    V3 rd = unit(cam.uu * cx + cam.vv * cy + cam.ww * P.fov);
    if (P.DOF_strenght > 0.f) {
        float RdoT = dot3(cam.ww, rd);
        V3 target = origin + rd * P.focalDistance / RdoT;
        origin = origin + sphere_sample(rng) * P.DOF_strenght;
        rd = unit(target - origin);
    }
    if (st) st->samples++;
    V3 color = mk(0., 0., 0.);
    if (!P.isRendering) {                                                                             // :504-525 preview
        Surf info = trace(S, origin, rd, st);
        if (info.hit) {
            float NoV = fmax_(dot3(rd * -1.0, info.n), 0.f);
            color = NoV * info.m.base;
        } else if (S.sky_mode == PT_SKY_SUNSKY) {
            color = sky_gradient(rd, S.sun);
        } else {
            color = env_eval(S, rd, st).c;
        }
    } else {                                                                                          // :527-540
        int ch = int(rng.next() * 3.);
        switch (ch) {
        case 0: integrate(color.x, S, origin, rd, rng, 0, max_bounces, st); break;
        case 1: integrate(color.y, S, origin, rd, rng, 1, max_bounces, st); break;
        case 2: integrate(color.z, S, origin, rd, rng, 2, max_bounces, st); break;
        }
        color = color * 3.f;
    }
    return hi3(color, mk(0.f, 0.f, 0.f));                                                             // :580
}

// ---------------------------------------------------------------- tonemap, tonemapping.cuh:17-63
inline V3 agx_contrast(V3 x)
{
    V3 x2 = x * x;
    V3 x4 = x2 * x2;
    return 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 + 0.1191f * x - 0.00232f;
}
inline V3 agx_tonemap(V3 x)
{
    const float min_ev = -12.47393f, max_ev = 4.026069f;
    x = mk(0.842479062253094f * x.x + 0.0423282422610123f * x.y + 0.0423756549057051f * x.z,
           0.0784335999999992f * x.x + 0.878468636469772f * x.y + 0.0784336f * x.z,
           0.0792237451477643f * x.x + 0.0791661274605434f * x.y + 0.879142973793104f * x.z);
    x = mk(clampf(pt_log2f(x.x), min_ev, max_ev), clampf(pt_log2f(x.y), min_ev, max_ev), clampf(pt_log2f(x.z), min_ev, max_ev));
    x = (x - min_ev) / (max_ev - min_ev);
    x = agx_contrast(x);
    // AgX_look
    const V3 lw = mk(0.2126f, 0.7152f, 0.0722f);
    float luma = dot3(lw, x);
    V3 offset = mk(0.f, 0.f, 0.f), slope = mk(1.f, 1.f, 1.f);
    float power = 1.35f, sat = 1.4f;
    x = bsdf::pow3(x * slope + offset, power);
    x = luma + sat * (x - luma);
    // AgX_Eotf
    return mk(1.19687900512017f * x.x - 0.0528968517574562f * x.y - 0.0529716355144438f * x.z,
              -0.0980208811401368f * x.x + 1.15190312990417f * x.y - 0.0980434501171241f * x.z,
              -0.0990297440797205f * x.x - 0.0989611768448433f * x.y + 1.15107367264116f * x.z);
}
// pathtracer.cu:588-599. float->int of NaN or out-of-range values is pinned to what both x86
// cvttss2si-then-clamp and gfx950 v_cvt_i32_f32-then-clamp agree on: NaN and negatives give 0.
inline uint8_t to_byte(float c)
{
    float s = c * 255;
    if (!(s == s)) return 0;
    if (s <= -2147483648.0f) return 0;
    if (s >= 2147483648.0f) return 255;
    int v = (int)s;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// ---------------------------------------------------------------- BVH build, BVH_builder.cpp:12-128
struct Builder {
    pt_triangle* tris; int* idx; pt_bvh_node* nodes; int used;
    void bounds(int ni)                                                                              // :12-26
    {
        pt_bvh_node& n = nodes[ni];
        V3 lo = mk(1e30, 1e30, 1e30), hi = mk(-1e30, -1e30, -1e30);
        for (int first = n.leftFirst, i = 0; i < n.triCount; i++) {
            const pt_triangle& t = tris[idx[first + i]];
            lo = lo3(lo, from(t.a) - 0.1); lo = lo3(lo, from(t.b) - 0.1); lo = lo3(lo, from(t.c) - 0.1);
            hi = hi3(hi, from(t.a) + 0.1); hi = hi3(hi, from(t.b) + 0.1); hi = hi3(hi, from(t.c) + 0.1);
        }
        n.aabbMin = pt_float3{lo.x, lo.y, lo.z};
        n.aabbMax = pt_float3{hi.x, hi.y, hi.z};
    }
    static float half_area(V3 lo, V3 hi) { V3 e = hi - lo; return e.x * e.y + e.y * e.z + e.z * e.x; }   // structs.cuh:143-146
    float sah(const pt_bvh_node& n, int axis, float pos)                                             // :28-48
    {
        V3 llo = mk(1e30, 1e30, 1e30), lhi = mk(-1e30, -1e30, -1e30), rlo = llo, rhi = lhi;
        int lc = 0, rc = 0;
        for (unsigned i = 0; i < (unsigned)n.triCount; i++) {
            const pt_triangle& t = tris[idx[n.leftFirst + i]];
            if (comp(from(t.origin), axis) < pos) {
                lc++;
                llo = lo3(llo, from(t.a)); lhi = hi3(lhi, from(t.a));
                llo = lo3(llo, from(t.b)); lhi = hi3(lhi, from(t.b));
                llo = lo3(llo, from(t.c)); lhi = hi3(lhi, from(t.c));
            } else {
                rc++;
                rlo = lo3(rlo, from(t.a)); rhi = hi3(rhi, from(t.a));
                rlo = lo3(rlo, from(t.b)); rhi = hi3(rhi, from(t.b));
                rlo = lo3(rlo, from(t.c)); rhi = hi3(rhi, from(t.c));
            }
        }
        float cost = lc * half_area(llo, lhi) + rc * half_area(rlo, rhi);
        return cost > 0 ? cost : 1e30;
    }
    float best_plane(const pt_bvh_node& n, int& axis, float& split)                                  // :49-65
    {
        float best = 1e30;
        for (int a = 0; a < 3; a++) {
            float bmin = comp(from(n.aabbMin), a), bmax = comp(from(n.aabbMax), a);
            if (bmin == bmax) continue;
            float scale = (bmax - bmin) / 100.;
            for (int i = 1; i < 100; i++) {
                float cp = bmin + i * scale;
                float cost = sah(n, a, cp);
                if (cost < best) { split = cp; axis = a; best = cost; }
            }
        }
        return best;
    }
    void subdivide(int ni)                                                                           // :66-113
    {
        V3 e = from(nodes[ni].aabbMax) - from(nodes[ni].aabbMin);
        float parentArea = e.x * e.y + e.y * e.z + e.z * e.x;
        float parentCost = nodes[ni].triCount * parentArea;
        int axis = 0; float pos = 0;
        float cost = best_plane(nodes[ni], axis, pos);
        if (cost >= parentCost) return;
        int i = nodes[ni].leftFirst;
        int j = i + nodes[ni].triCount - 1;
        while (i <= j) {
            if (comp(from(tris[idx[i]].origin), axis) < pos) i++;
            else std::swap(idx[i], idx[j--]);
        }
        int leftCount = i - nodes[ni].leftFirst;
        if (leftCount == 0 || leftCount == nodes[ni].triCount) return;
        int l = used++, r = used++;
        nodes[l].leftFirst = nodes[ni].leftFirst; nodes[l].triCount = leftCount;
        nodes[r].leftFirst = i; nodes[r].triCount = nodes[ni].triCount - leftCount;
        nodes[ni].leftFirst = l; nodes[ni].triCount = 0;
        bounds(l); bounds(r);
        subdivide(l); subdivide(r);
    }
};

} // namespace

// ================================================================ C interface for the tests
extern "C" {

struct orc_texture { const float* texels; int w, h, comps; };
struct orc_scene {
    const pt_triangle* tris; int n_tris;
    const int* tri_idx;
    const pt_bvh_node* nodes; int n_nodes;
    const pt_material* mats; int n_mats;
    const orc_texture* textures; int n_textures;
    const float* env_rgba; const float* env_cdf; int env_w, env_h; float env_sum;
    int sky_mode;
    float sun[3];
};
struct orc_stats { uint64_t rays, samples, box_tests, tri_tests, shade_vertices, env_samples, tex_taps, tex_taps_r; };
struct orc_hit { float t; uint32_t tri; float u, v; };

} // extern "C"

namespace {
Scene view(const orc_scene* s)
{
    Scene S;
    S.tris = s->tris; S.n_tris = s->n_tris; S.tri_idx = s->tri_idx; S.nodes = s->nodes; S.n_nodes = s->n_nodes;
    S.mats = s->mats; S.n_mats = s->n_mats; S.tex = reinterpret_cast<const OrcTexture*>(s->textures); S.n_tex = s->n_textures;
    S.env_rgba = s->env_rgba; S.env_cdf = s->env_cdf; S.env_w = s->env_w; S.env_h = s->env_h; S.env_sum = s->env_sum;
    S.sky_mode = s->sky_mode; S.sun = mk(s->sun[0], s->sun[1], s->sun[2]);
    return S;
}
void put(orc_stats* o, const Stats& s)
{
    if (!o) return;
    o->rays = s.rays; o->samples = s.samples; o->box_tests = s.box_tests; o->tri_tests = s.tri_tests;
    o->shade_vertices = s.shade_vertices; o->env_samples = s.env_samples; o->tex_taps = s.tex_taps; o->tex_taps_r = s.tex_taps_r;
}
}

extern "C" {

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox10(ctr, key, out); }

void orc_uniform_stream(uint32_t pixel, int frame, int n, float* out)
{
    Rng r; r.seed(pixel, frame);
    for (int i = 0; i < n; i++) out[i] = r.next();
}

// buildBVH, BVH_builder.cpp:115-128. nodes must hold 2n-1 entries.
int orc_build_bvh(pt_triangle* tris, int* idx, int n, pt_bvh_node* nodes)
{
    for (int i = 0; i < 2 * n - 1; i++) { nodes[i] = pt_bvh_node{{0, 0, 0}, {0, 0, 0}, 0, 0}; }
    for (int i = 0; i < n; i++) {
        V3 c = (from(tris[i].a) + from(tris[i].b) + from(tris[i].c)) * 0.3333f;
        tris[i].origin = pt_float3{c.x, c.y, c.z};
    }
    Builder b{tris, idx, nodes, 1};
    nodes[0].leftFirst = 0; nodes[0].triCount = n;
    b.bounds(0);
    b.subdivide(0);
    return b.used;
}

// envMap::buildCDF, envmap.cpp:11-28
void orc_build_cdf(const float* rgba, int w, int h, float* cdf, float* sum)
{
    float run = 0.f;
    for (int i = 0; i < w * h; i++) {
        float wgt = 0.212671 * rgba[4 * i] + 0.715160 * rgba[4 * i + 1] + 0.072169 * rgba[4 * i + 2];
        run = (i == 0) ? wgt : run + wgt;
        cdf[i] = run;
    }
    *sum = cdf[w * h - 1];
}

void orc_closest_brute(const orc_scene* s, const pt_ray* rays, int n, orc_hit* out)
{
    Scene S = view(s);
    // one ray costs a pass over every triangle: small units, so a few hundred rays still use every core
#pragma omp parallel for schedule(dynamic, 2)
    for (int i = 0; i < n; i++) {
        HitRec h = closest_brute(S, from(rays[i].o), from(rays[i].d));
        out[i] = orc_hit{h.t, h.tri, h.u, h.v};
    }
}

void orc_closest_bvh(const orc_scene* s, const pt_ray* rays, int n, orc_hit* out, uint32_t* box_tests, uint32_t* tri_tests)
{
    Scene S = view(s);
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        Stats st;
        HitRec h = closest_bvh(S, from(rays[i].o), from(rays[i].d), &st);
        out[i] = orc_hit{h.t, h.tri, h.u, h.v};
        if (box_tests) box_tests[i] = (uint32_t)st.box_tests;
        if (tri_tests) tri_tests[i] = (uint32_t)st.tri_tests;
    }
}

void orc_closest_reference_order(const orc_scene* s, const pt_ray* rays, int n, orc_hit* out)
{
    Scene S = view(s);
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        HitRec h = closest_reference_order(S, from(rays[i].o), from(rays[i].d));
        out[i] = orc_hit{h.t, h.tri, h.u, h.v};
    }
}

void orc_camera_basis(const pt_params* P, float ww[3], float uu[3], float vv[3])
{
    Camera c = make_camera(*P);
    ww[0] = c.ww.x; ww[1] = c.ww.y; ww[2] = c.ww.z;
    uu[0] = c.uu.x; uu[1] = c.uu.y; uu[2] = c.uu.z;
    vv[0] = c.vv.x; vv[1] = c.vv.y; vv[2] = c.vv.z;
}

// Frames [frame_begin, frame_end) added into accum (3*W*H floats, caller-zeroed for a fresh render:
// 0 + c equals the overwrite of pathtracer.cu:583-584 bit for bit because c >= +0).
int orc_render(const orc_scene* s, const pt_params* P, int frame_begin, int frame_end, int max_bounces,
               float* accum, orc_stats* stats, int nthreads)
{
    Scene S = view(s);
    Camera cam = make_camera(*P);
    int W = P->windowSize.x, H = P->windowSize.y;
    Stats total;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        Stats local;
#pragma omp for schedule(dynamic, 1)
        for (int y = 0; y < H; y++) {
            for (int x = 0; x < W; x++) {
                size_t p = (size_t)(x + y * W) * 3;
                for (int f = frame_begin; f < frame_end; f++) {
                    V3 c = pixel_sample(S, *P, cam, x, y, f, max_bounces, &local);
                    accum[p] += c.x; accum[p + 1] += c.y; accum[p + 2] += c.z;
                }
            }
        }
#pragma omp critical
        total.add(local);
    }
    put(stats, total);
    return 0;
}

// one sample of one pixel (debug hook for the tests)
void orc_pixel_sample(const orc_scene* s, const pt_params* P, int x, int y, int frame, int max_bounces, float rgb[3], orc_stats* stats)
{
    Scene S = view(s);
    Camera cam = make_camera(*P);
    Stats st;
    V3 c = pixel_sample(S, *P, cam, x, y, frame, max_bounces, &st);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    put(stats, st);
}

// pathtracer.cu:588-599: accum / n_frames -> AgX -> RGBA8
void orc_resolve_display(const float* accum, int w, int h, int n_frames, uint8_t* rgba)
{
    float div = (float)n_frames;
    for (int i = 0; i < w * h; i++) {
        V3 c = mk(accum[3 * i], accum[3 * i + 1], accum[3 * i + 2]) / div;
        c = agx_tonemap(c);
        rgba[4 * i] = to_byte(c.x); rgba[4 * i + 1] = to_byte(c.y); rgba[4 * i + 2] = to_byte(c.z); rgba[4 * i + 3] = 255;
    }
}

// post_process.glsl:19-23 applied to an RGBA8 image: exposure, multiplier, saturation about the luma, gamma;
// texel = byte / 255, output byte = round(clamp(x) * 255) (the fixed-function conversions around the shader)
void orc_post_process(const uint8_t* in, int w, int h, const pt_params* P, uint8_t* out)
{
    auto unorm = [](float v) -> uint8_t { if (!(v > 0.f)) return 0; if (v >= 1.f) return 255; return (uint8_t)(int)(v * 255.0f + 0.5f); };
    for (int i = 0; i < w * h; i++) {
        V3 col = mk((float)in[4 * i] / 255.0f, (float)in[4 * i + 1] / 255.0f, (float)in[4 * i + 2] / 255.0f);
        col = col * P->exposure;
        col = col * from(P->mult);
        float lw = 0.2126f * col.x + 0.7152f * col.y + 0.0722f * col.z;
        float s = P->saturation;
        col = mk(lw * (1.0f - s) + col.x * s, lw * (1.0f - s) + col.y * s, lw * (1.0f - s) + col.z * s);      // mix(vec3(lw), col, s)
        col = mk(pt_powf(col.x, P->gamma), pt_powf(col.y, P->gamma), pt_powf(col.z, P->gamma));
        out[4 * i] = unorm(col.x); out[4 * i + 1] = unorm(col.y); out[4 * i + 2] = unorm(col.z); out[4 * i + 3] = 255;
    }
}

void orc_agx(const float in[3], float out[3])
{
    V3 c = agx_tonemap(mk(in[0], in[1], in[2]));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

void orc_tex2d(const orc_texture* t, float u, float v, float out[4])
{
    OrcTexture T{t->texels, t->w, t->h, t->comps};
    out[0] = out[1] = out[2] = out[3] = 0;
    tex_fetch(T, u, v, out, nullptr);
}

// BSDF unit hooks: surface given by a material and a normal; returns f (3) and pdf
void orc_bsdf_eval(const pt_material* M, int inside, const float V[3], const float N[3], const float L[3], float f_out[3], float* pdf_out)
{
    Surf s; s.hit = true; s.inside = inside != 0; s.t = 0; s.n = mk(N[0], N[1], N[2]);
    s.m.base = from(M->baseColor); s.m.emis = from(M->emissive); s.m.rough = fmax_(M->roughness, 0.0001f); s.m.aniso = M->anisotropic;
    s.m.metal = M->metallic; s.m.subsurf = M->subsurface; s.m.sheen = M->sheen; s.m.coat = M->clearcoat; s.m.coat_rough = M->clearcoatRoughness;
    s.m.strans = M->specTrans; s.m.alpha = M->alpha; s.m.ior = M->ior; s.m.G = M->medium.G; s.m.sigA = from(M->medium.sigmaA); s.m.sigS = from(M->medium.sigmaS);
    float pdf;
    V3 f = bsdf::eval(s, mk(V[0], V[1], V[2]), s.n, mk(L[0], L[1], L[2]), pdf);
    f_out[0] = f.x; f_out[1] = f.y; f_out[2] = f.z; *pdf_out = pdf;
}
void orc_bsdf_sample(const pt_material* M, int inside, const float V[3], const float N[3], uint32_t pixel, int frame,
                     float L_out[3], float f_out[3], float* pdf_out, int* inside_out)
{
    Surf s; s.hit = true; s.inside = inside != 0; s.t = 0; s.n = mk(N[0], N[1], N[2]);
    s.m.base = from(M->baseColor); s.m.emis = from(M->emissive); s.m.rough = fmax_(M->roughness, 0.0001f); s.m.aniso = M->anisotropic;
    s.m.metal = M->metallic; s.m.subsurf = M->subsurface; s.m.sheen = M->sheen; s.m.coat = M->clearcoat; s.m.coat_rough = M->clearcoatRoughness;
    s.m.strans = M->specTrans; s.m.alpha = M->alpha; s.m.ior = M->ior; s.m.G = M->medium.G; s.m.sigA = from(M->medium.sigmaA); s.m.sigS = from(M->medium.sigmaS);
    Rng rng; rng.seed(pixel, frame);
    bool in2 = inside != 0;
    float pdf; V3 L;
    V3 f = bsdf::sample(s, mk(V[0], V[1], V[2]), s.n, L, pdf, rng, in2);
    L_out[0] = L.x; L_out[1] = L.y; L_out[2] = L.z; f_out[0] = f.x; f_out[1] = f.y; f_out[2] = f.z; *pdf_out = pdf; *inside_out = in2;
}

// the same two hooks over arrays (n directions, or the random streams of pixels pixel0 .. pixel0 + n - 1)
void orc_bsdf_eval_batch(const pt_material* M, int inside, const float V[3], const float N[3], const float* L, int n, float* f_out, float* pdf_out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) orc_bsdf_eval(M, inside, V, N, L + 3 * (size_t)i, f_out + 3 * (size_t)i, pdf_out + i);
}
void orc_bsdf_sample_batch(const pt_material* M, int inside, const float V[3], const float N[3], uint32_t pixel0, int n, int frame,
                           float* L_out, float* f_out, float* pdf_out, int* inside_out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) orc_bsdf_sample(M, inside, V, N, pixel0 + (uint32_t)i, frame, L_out + 3 * (size_t)i, f_out + 3 * (size_t)i, pdf_out + i, inside_out + i);
}

void orc_env_eval(const orc_scene* s, const float d[3], float out[4])
{
    Scene S = view(s);
    Env4 e = env_eval(S, mk(d[0], d[1], d[2]), nullptr);
    out[0] = e.c.x; out[1] = e.c.y; out[2] = e.c.z; out[3] = e.pdf;
}
void orc_env_sample(const orc_scene* s, uint32_t pixel, int frame, float dir_pdf[4], float color[3])
{
    Scene S = view(s);
    Rng rng; rng.seed(pixel, frame);
    V3 col;
    Env4 e = env_sample(S, col, rng, nullptr);
    dir_pdf[0] = e.c.x; dir_pdf[1] = e.c.y; dir_pdf[2] = e.c.z; dir_pdf[3] = e.pdf;
    color[0] = col.x; color[1] = col.y; color[2] = col.z;
}
void orc_sky_gradient(const float d[3], const float sun[3], float out[3])
{
    V3 c = sky_gradient(mk(d[0], d[1], d[2]), mk(sun[0], sun[1], sun[2]));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

// host evaluation of the pinned transcendentals, same 12 outputs per input as the device probe
void orc_detmath_probe(const float* x, const float* y, int n, float* out)
{
    for (int i = 0; i < n; i++) {
        float a = x[i], b = y[i];
        float* o = out + (size_t)i * 12;
        o[0] = pt_sinf(a); o[1] = pt_cosf(a); o[2] = pt_expf(a); o[3] = pt_logf(a); o[4] = pt_log2f(a);
        o[5] = pt_powf(a, b); o[6] = pt_acosf(a); o[7] = pt_atan2f(a, b);
        o[8] = 1.0f / a; o[9] = __builtin_sqrtf(a); o[10] = a / b; o[11] = (float)pt_dm_pow((double)a, (double)b);
    }
}

void orc_set_depth_histogram(unsigned long long* hist65) { g_depth_hist = hist65; }

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

} // extern "C"
