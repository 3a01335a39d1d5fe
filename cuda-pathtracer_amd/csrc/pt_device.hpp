// Device-side building blocks of the render path for gfx950: vector algebra, Philox stream,
// software texture sampler, BVH traversal with an LDS-staged stack, material resolve, Disney BSDF,
// phase function, environment lighting and the per-sample state machine.
//
// Arithmetic contract: every expression below is evaluated in the operand types and the order the
// reference uses ("Cuda Pathtracer/core/CUDA/*"; the function comments give the lines), with FMA
// contraction disabled for the whole translation unit, IEEE division and square root, and the
// transcendentals of include/pt_detmath.h.  That makes results reproducible bit for bit on any
// IEEE machine, which is what the parity tests check.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pt_capi.h"
#include "../../include/pt_detmath.h"
#include "pt_layout.h"

#pragma clang fp contract(off)

namespace ptamd {

#define PT_DEV __device__ __forceinline__

// ---------------------------------------------------------------- vectors (float3_header.cuh:5-123)
struct F3 { float x, y, z; };
struct F2 { float x, y; };
PT_DEV F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV F3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }
PT_DEV F3 ld3(const pt_float3& p) { return f3(p.x, p.y, p.z); }
PT_DEV F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV F3 operator/(F3 a, F3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_DEV F3 operator+(F3 a, float b) { return f3(a.x + b, a.y + b, a.z + b); }
PT_DEV F3 operator-(F3 a, float b) { return f3(a.x - b, a.y - b, a.z - b); }
PT_DEV F3 operator*(F3 a, float b) { return f3(a.x * b, a.y * b, a.z * b); }
PT_DEV F3 operator/(F3 a, float b) { return f3(a.x / b, a.y / b, a.z / b); }
PT_DEV F3 operator+(float a, F3 b) { return f3(a + b.x, a + b.y, a + b.z); }
PT_DEV F3 operator-(float a, F3 b) { return f3(a - b.x, a - b.y, a - b.z); }
PT_DEV F3 operator*(float a, F3 b) { return f3(a * b.x, a * b.y, a * b.z); }
PT_DEV F3 operator/(float a, F3 b) { return f3(a / b.x, a / b.y, a / b.z); }
PT_DEV float dot(F3 u, F3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
PT_DEV F3 cross(F3 p, F3 q) { return f3(p.y * q.z - q.y * p.z, p.z * q.x - q.z * p.x, p.x * q.y - q.x * p.y); }
PT_DEV float sqrt_(float x) { return __builtin_sqrtf(x); }
PT_DEV float rsqrt_(float x) { return 1.0f / __builtin_sqrtf(x); }
PT_DEV F3 normalize(F3 a) { float s = rsqrt_(a.x * a.x + a.y * a.y + a.z * a.z); return f3(a.x * s, a.y * s, a.z * s); }
PT_DEV float length(F3 a) { return sqrt_(a.x * a.x + a.y * a.y + a.z * a.z); }
PT_DEV float abs_(float x) { return __builtin_fabsf(x); }
PT_DEV F3 max3(F3 a, F3 b) { return f3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }
PT_DEV float getN(F3 v, int n) { return n == 0 ? v.x : (n == 1 ? v.y : v.z); }
PT_DEV bool finite3(F3 a) { return (a.x - a.x == 0.0f) && (a.y - a.y == 0.0f) && (a.z - a.z == 0.0f); }

// constants.cuh:4-8
#define PT_PI 3.1415926535f
#define PT_INV_PI 0.318309f
#define PT_TWO_PI 6.283184f
#define PT_INV_TWO_PI 0.159154976203148f
#define PT_INV_4_PI 0.78539816339744830961566084581988f

// cuda_utility.cuh:7-78
PT_DEV float clamp_(float a, float l, float h) { return pt_fminf(h, pt_fmaxf(l, a)); }
PT_DEV float sign_(float x) { if (x == 0) return 0; return x > 0. ? 1. : -1.; }
PT_DEV float smoothstep_(float e0, float e1, float x) { float t = clamp_((x - e0) / (e1 - e0), 0., 1.); return t * t * (3.0 - 2.0 * t); }
PT_DEV float mix_(float a, float b, float m) { return a * (1. - m) + b * m; }
PT_DEV F3 mix_(F3 a, F3 b, float m) { return f3(a.x * (1. - m) + b.x * m, a.y * (1. - m) + b.y * m, a.z * (1. - m) + b.z * m); }
PT_DEV void onb(F3 N, F3& T, F3& B)
{
    F3 up = abs_(N.z) < 0.999 ? f3(0, 0, 1) : f3(1, 0, 0);
    T = normalize(cross(up, N));
    B = cross(N, T);
}
PT_DEV F3 to_world(F3 X, F3 Y, F3 Z, F3 v) { return v.x * X + v.y * Y + v.z * Z; }
PT_DEV F3 to_local(F3 X, F3 Y, F3 Z, F3 v) { return f3(dot(v, X), dot(v, Y), dot(v, Z)); }
PT_DEV F3 ref_if_neg(F3 v, F3 r) { float k = dot(v, r); return k > 0.0f ? v : v - 2.0f * r * k; }
PT_DEV float power_heuristic(float a, float b) { float t = a * a; return t / (b * b + t); }
PT_DEV F3 reflect_(F3 I, F3 N) { return I - N * 2.0f * dot(N, I); }
PT_DEV F3 refract_(F3 I, F3 N, float ior)
{
    float k = 1.0 - ior * ior * (1. - dot(N, I) * dot(N, I));
    if (k < 0.) return f3(0., 0., 0.);
    return I * ior - N * (ior * dot(N, I) + sqrt_(k));
}
PT_DEV F3 pow3(F3 a, float p) { return f3(pt_powf(a.x, p), pt_powf(a.y, p), pt_powf(a.z, p)); }

// ---------------------------------------------------------------- Philox4x32-10 stream
// curand_init(pixel, 0, 4096*frame) + curand_uniform (cuda_utility.cuh:3-4, pathtracer.cu:560):
// key (pixel, 0), counter 1024*frame + draw/4, outputs consumed in order, u*2^-32 + 2^-33.
struct Rng {
    uint32_t key;
    uint32_t base;     // counter of draw 0
    uint32_t draws;    // draws consumed so far
    uint32_t o0, o1, o2, o3;
};
PT_DEV void philox_block(uint32_t key0, uint32_t c0, uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3)
{
    uint32_t x0 = c0, x1 = 0, x2 = 0, x3 = 0, k0 = key0, k1 = 0;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of separate high and low halves
        unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ x1 ^ k0, n2 = hi0 ^ x3 ^ k1;
        x0 = n0; x1 = lo1; x2 = n2; x3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r0 = x0; r1 = x1; r2 = x2; r3 = x3;
}
PT_DEV void rng_seed(Rng& g, uint32_t pixel, uint32_t frame, uint32_t draws)
{
    g.key = pixel; g.base = 1024u * frame; g.draws = draws;
    philox_block(g.key, g.base + (draws >> 2), g.o0, g.o1, g.o2, g.o3);
}
PT_DEV float rng_next(Rng& g)
{
    uint32_t p = g.draws & 3u;
    uint32_t v = p == 0 ? g.o0 : (p == 1 ? g.o1 : (p == 2 ? g.o2 : g.o3));
    g.draws++;
    if (p == 3) philox_block(g.key, g.base + (g.draws >> 2), g.o0, g.o1, g.o2, g.o3);
    return (float)v * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}

// ---------------------------------------------------------------- software texture unit
// Wrap addressing, normalized coordinates, bilinear filter with texel centres at +0.5 and fp32
// weights; stands in for the CUDA texture objects of image_loader.h:37-45 / envmap.cpp:49-57.
PT_DEV int wrap_(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
struct Taps { int i00, i10, i01, i11; float ax, ay, bx, by; };
PT_DEV Taps tex_taps(int w, int h, float u, float v)
{
    if (!(abs_(u) < 65536.0f)) u = 0.0f;
    if (!(abs_(v) < 65536.0f)) v = 0.0f;
    float x = u * (float)w - 0.5f;
    float y = v * (float)h - 0.5f;
    float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    Taps t;
    t.ax = x - fx; t.ay = y - fy;
    t.bx = 1.0f - t.ax; t.by = 1.0f - t.ay;
    int ix = (int)fx, iy = (int)fy;
    int x0, x1, y0, y1;
    if (((w & (w - 1)) | (h & (h - 1))) == 0) {
        // both sizes are powers of two: the non-negative remainder is a mask (same values, no integer divisions)
        x0 = ix & (w - 1); x1 = (ix + 1) & (w - 1);
        y0 = iy & (h - 1); y1 = (iy + 1) & (h - 1);
    } else {
        x0 = wrap_(ix, w); x1 = wrap_(ix + 1, w);
        y0 = wrap_(iy, h); y1 = wrap_(iy + 1, h);
    }
    t.i00 = y0 * w + x0; t.i10 = y0 * w + x1; t.i01 = y1 * w + x0; t.i11 = y1 * w + x1;
    return t;
}
PT_DEV float tex_mix(const Taps& t, float c00, float c10, float c01, float c11)
{
    return (c00 * t.bx + c10 * t.ax) * t.by + (c01 * t.bx + c11 * t.ax) * t.ay;
}
// (texel pointers come out of a descriptor in memory, so the compiler cannot know their address space: said here, the loads
// are global loads instead of flat ones)
#define PT_GLOBAL __attribute__((address_space(1)))
PT_DEV float4 tex4(const PT_GLOBAL float* texels, int w, int h, float u, float v)
{
    Taps t = tex_taps(w, h, u, v);
    typedef float __attribute__((ext_vector_type(4))) vec4;        // (the float4 class has no constructor from another address space)
    const PT_GLOBAL vec4* p = (const PT_GLOBAL vec4*)texels;
    const vec4 a = p[t.i00], b = p[t.i10], c = p[t.i01], d = p[t.i11];
    return make_float4(tex_mix(t, a.x, b.x, c.x, d.x), tex_mix(t, a.y, b.y, c.y, d.y),
                       tex_mix(t, a.z, b.z, c.z, d.z), tex_mix(t, a.w, b.w, c.w, d.w));
}
PT_DEV float tex1(const PT_GLOBAL float* texels, int w, int h, float u, float v)
{
    Taps t = tex_taps(w, h, u, v);
    return tex_mix(t, texels[t.i00], texels[t.i10], texels[t.i01], texels[t.i11]);
}
// k / 255 correctly rounded, for k = 0..255, without a division: one Newton correction of k * (1/255)
// (checked against the division for all 256 values by the tests)
PT_DEV float unorm8(uint32_t k)
{
    const float r = 0.003921568859368563f;
    float kf = (float)k;
    float q = kf * r;
    float e = __builtin_fmaf(-q, 255.0f, kf);
    return __builtin_fmaf(e, r, q);
}
PT_DEV float4 tex4_u8(const PT_GLOBAL uint32_t* texels, int w, int h, float u, float v)
{
    Taps t = tex_taps(w, h, u, v);
    uint32_t a = texels[t.i00], b = texels[t.i10], c = texels[t.i01], d = texels[t.i11];
    return make_float4(tex_mix(t, unorm8(a & 255u), unorm8(b & 255u), unorm8(c & 255u), unorm8(d & 255u)),
                       tex_mix(t, unorm8((a >> 8) & 255u), unorm8((b >> 8) & 255u), unorm8((c >> 8) & 255u), unorm8((d >> 8) & 255u)),
                       tex_mix(t, unorm8((a >> 16) & 255u), unorm8((b >> 16) & 255u), unorm8((c >> 16) & 255u), unorm8((d >> 16) & 255u)),
                       tex_mix(t, unorm8(a >> 24), unorm8(b >> 24), unorm8(c >> 24), unorm8(d >> 24)));
}
PT_DEV float tex1_u8(const PT_GLOBAL uint8_t* texels, int w, int h, float u, float v)
{
    Taps t = tex_taps(w, h, u, v);
    return tex_mix(t, unorm8(texels[t.i00]), unorm8(texels[t.i10]), unorm8(texels[t.i01]), unorm8(texels[t.i11]));
}
PT_DEV float4 tex4_any(const DTex& tx, float u, float v)
{
    return tx.format ? tex4_u8((const PT_GLOBAL uint32_t*)tx.texels, tx.w, tx.h, u, v) : tex4((const PT_GLOBAL float*)tx.texels, tx.w, tx.h, u, v);
}
PT_DEV float tex1_any(const DTex& tx, float u, float v)
{
    return tx.format ? tex1_u8((const PT_GLOBAL uint8_t*)tx.texels, tx.w, tx.h, u, v) : tex1((const PT_GLOBAL float*)tx.texels, tx.w, tx.h, u, v);
}

// ---------------------------------------------------------------- intersection
// intersectors.cuh:59-71 (boxIntersectF)
PT_DEV float box_entry(F3 o, F3 inv, const float* bmin, const float* bmax)
{
    F3 p0 = -1.0f * (o - ld3(bmin)) * inv;
    F3 p1 = -1.0f * (o - ld3(bmax)) * inv;
    float nx = pt_fminf(p0.x, p1.x), ny = pt_fminf(p0.y, p1.y), nz = pt_fminf(p0.z, p1.z);
    float fx = pt_fmaxf(p0.x, p1.x), fy = pt_fmaxf(p0.y, p1.y), fz = pt_fmaxf(p0.z, p1.z);
    float tn = pt_fmaxf(pt_fmaxf(nx, ny), nz);
    float tf = pt_fminf(pt_fminf(fx, fy), fz);
    if (tf > tn && tf > 0) return tn;
    return 1e30f;
}

struct Hit { float t; uint32_t tri; float u, v; };
#define PT_NO_TRI 0xFFFFFFFFu

// pathtracer.cu:124-139: Moeller-Trumbore with the reference's rejection tests; limit plays hit.t
PT_DEV bool tri_test(F3 a, F3 e1, F3 e2, F3 o, F3 d, float limit, float& t, float& u, float& v)
{
    F3 h = cross(d, e2);
    float det = dot(e1, h);
    if (det > -0.0001f && det < 0.0001f) return false;
    float f = 1.0f / det;
    F3 s = o - a;
    u = f * dot(s, h);
    if (u < 0.f || u > 1.f) return false;
    F3 q = cross(s, e1);
    v = f * dot(d, q);
    if (v < 0.f || u + v > 1.f) return false;
    t = f * dot(e2, q);
    if (t < 0.f || t > limit) return false;
    return true;
}

// Closest hit (pathtracer.cu:185-266: BVHIntersect + map). Same visiting order as the reference:
// the nearer child is entered first, the farther one is deferred together with its entry distance
// and dropped on pop if that distance no longer beats the current hit. Exact-t ties go to the
// lowest original triangle index. Rays with a non-finite component miss.
//
// Per-lane traversal stack: the first LDS_DEPTH entries live in LDS, laid out [entry][lane] so that a
// wave's push or pop is one contiguous 512-byte row (no bank conflicts); deeper entries spill to
// private memory. The loop is "descend until a leaf, then test the leaf": lanes of a wave spend
// their iterations in the same kind of work.
constexpr int kLdsDepth = 16;
constexpr int kSpillDepth = 48;
#define PT_TRAV_EMPTY ((int)0x80000000)

template <int BLOCK, int LDS_DEPTH, bool COUNT>
PT_DEV Hit closest_hit(const DScene& S, F3 o, F3 d, int tid, uint32_t& nbox, uint32_t& ntri)
{
    __shared__ uint2 s_stack[LDS_DEPTH * BLOCK];
    uint2 spill[kSpillDepth];
    int sp = 0;
    Hit best; best.t = 1e10f; best.tri = PT_NO_TRI; best.u = 0.f; best.v = 0.f;
    if (!finite3(o) || !finite3(d)) return best;
    F3 inv = 1.0f / d;
    if (COUNT) nbox += 1;
    float troot = box_entry(o, inv, S.root_min, S.root_max);
    if (!(troot < best.t)) return best;
    int cur = S.root_ref;

    auto push = [&](int ref, float tn) {
        uint2 e = make_uint2((uint32_t)ref, __float_as_uint(tn));
        if (sp < LDS_DEPTH) s_stack[sp * BLOCK + tid] = e;
        else if (sp - LDS_DEPTH < kSpillDepth) spill[sp - LDS_DEPTH] = e;
        sp++;
    };
    auto pop_valid = [&]() -> int {
        while (sp > 0) {
            sp--;
            uint2 e;
            if (sp < LDS_DEPTH) e = s_stack[sp * BLOCK + tid];
            else {
                e = spill[(sp - LDS_DEPTH) < kSpillDepth ? (sp - LDS_DEPTH) : (kSpillDepth - 1)];
                // keeps the two loads in separate blocks: merged, they become one generic-pointer load
                asm volatile("" : "+v"(e.x), "+v"(e.y));
            }
            if (__uint_as_float(e.y) < best.t) return (int)e.x;
        }
        return PT_TRAV_EMPTY;
    };

    for (;;) {
        while (cur >= 0) {
            const float4* np = reinterpret_cast<const float4*>(S.nodes + cur);
            float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            float lmin[3] = {n0.x, n0.y, n0.z}, lmax[3] = {n0.w, n1.x, n1.y};
            float rmin[3] = {n1.z, n1.w, n2.x}, rmax[3] = {n2.y, n2.z, n2.w};
            int left = __float_as_int(n3.x), right = __float_as_int(n3.y);
            float dl = box_entry(o, inv, lmin, lmax);
            float dr = box_entry(o, inv, rmin, rmax);
            if (COUNT) nbox += 2;
            int nearR, farR; float dn, df;
            if (dl > dr) { nearR = right; dn = dr; farR = left; df = dl; }
            else         { nearR = left;  dn = dl; farR = right; df = dr; }
            if (df < best.t) push(farR, df);
            if (dn < best.t) cur = nearR;
            else cur = pop_valid();
        }
        if (cur == PT_TRAV_EMPTY) break;
        {
            const DTri* tp = S.tris + (~cur);
            int count = 1;
            for (int i = 0; i < count; i++) {
                float4 q0 = tp[i].q0, q1 = tp[i].q1, q2 = tp[i].q2;
                if (i == 0) count = __float_as_int(q1.w);
                uint32_t id = __float_as_uint(q0.w) >> 1;       // (the id word carries a mark below the index, see DTri)
                float t, u, v;
                if (COUNT) ntri += 1;
                if (!tri_test(f3(q0.x, q0.y, q0.z), f3(q1.x, q1.y, q1.z), f3(q2.x, q2.y, q2.z), o, d, best.t, t, u, v)) continue;
                if (best.tri != PT_NO_TRI && t == best.t && id > best.tri) continue;
                best.t = t; best.tri = id; best.u = u; best.v = v;
            }
        }
        cur = pop_valid();
        if (cur == PT_TRAV_EMPTY) break;
    }
    return best;
}

// The exit distance of a wide box is taken 2^-19 wider (relative) before it is compared with the entry distance. The
// boxes' own margin is absolute (a fraction of the scene's largest coordinate), while the rounding of the plane
// distances grows with the distance: the hardware reciprocals of two axes differ by up to 2^-22 of t, so for a ray that
// starts many scene extents away a thin box (a flat floor seen from far above) would compute entry >= exit and be lost.
// Near the surface the factor changes nothing (2^-19 of a small t), so thin boxes stay thin for rays that leave them.
#ifndef PT_WIDE_EXIT_SLACK
#define PT_WIDE_EXIT_SLACK 1.0000019073486328f
#endif
constexpr float kWideExitSlack = PT_WIDE_EXIT_SLACK;

// One step through a 4-wide node (DNode4: a grid origin, a power-of-two cell size per axis, four child boxes as
// 8-bit grid coordinates, four child references): the entry distances of the four children, culled against the
// bound `cull` (the best hit so far, taken a little wide), sorted nearest first (1e30 = not to be visited). A plane
// at q cells from the origin org lies at distance q * (inv * cell) + (org - o) * inv along the ray: one conversion
// and one fused multiply-add per plane after three subtractions, multiplications and scalings per node. The
// caller keeps inv finite (an axis the ray is parallel to would give inf - inf = NaN on both planes). Boxes on
// the grid are conservative, so a hit is never lost and the closest hit is exact.
PT_DEV void wide_step(const float4& n0, const float4& n1, const float4& n2, const float4& n3, F3 o, F3 inv, float cull,
                      int& r0, int& r1, int& r2, int& r3, float& d0, float& d1, float& d2, float& d3)
{
    const uint32_t ex = __float_as_uint(n0.w);
    const float sx = __builtin_ldexpf(inv.x, (int)(int8_t)(ex & 255u)), sy = __builtin_ldexpf(inv.y, (int)(int8_t)((ex >> 8) & 255u)),
                sz = __builtin_ldexpf(inv.z, (int)(int8_t)((ex >> 16) & 255u));
    const float bx = (n0.x - o.x) * inv.x, by = (n0.y - o.y) * inv.y, bz = (n0.z - o.z) * inv.z;
    const uint32_t lx = __float_as_uint(n1.x), ly = __float_as_uint(n1.y), lz = __float_as_uint(n1.z);
    const uint32_t hx = __float_as_uint(n1.w), hy = __float_as_uint(n2.x), hz = __float_as_uint(n2.y);
    r0 = __float_as_int(n3.x); r1 = __float_as_int(n3.y); r2 = __float_as_int(n3.z); r3 = __float_as_int(n3.w);
    // the near plane of a slab is its low one for a direction component >= 0, its high one otherwise (the plane
    // distance is monotonic in the cell coordinate): choosing the words once per node replaces a min and a max per
    // child and axis, with bit-identical results
    const bool ngx = inv.x < 0.f, ngy = inv.y < 0.f, ngz = inv.z < 0.f;
    const uint32_t ax = ngx ? hx : lx, fx = ngx ? lx : hx, ay = ngy ? hy : ly, fy = ngy ? ly : hy, az = ngz ? hz : lz, fz = ngz ? lz : hz;
    float dk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float tnx = __builtin_fmaf((float)((ax >> (8 * k)) & 255u), sx, bx), tfx = __builtin_fmaf((float)((fx >> (8 * k)) & 255u), sx, bx);
        const float tny = __builtin_fmaf((float)((ay >> (8 * k)) & 255u), sy, by), tfy = __builtin_fmaf((float)((fy >> (8 * k)) & 255u), sy, by);
        const float tnz = __builtin_fmaf((float)((az >> (8 * k)) & 255u), sz, bz), tfz = __builtin_fmaf((float)((fz >> (8 * k)) & 255u), sz, bz);
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
        dk[k] = (tf * kWideExitSlack > tn && tf > 0.f && tn < cull) ? tn : 1e30f;
    }
    d0 = dk[0]; d1 = dk[1]; d2 = dk[2]; d3 = dk[3];       // (a missing child's slabs are inverted: never entered)
    auto cswap = [](float& da, int& ra, float& db, int& rb) {
        bool sw = db < da;
        float td = sw ? db : da; int tr = sw ? rb : ra;
        db = sw ? da : db; rb = sw ? ra : rb;
        da = td; ra = tr;
    };
    cswap(d0, r0, d1, r1); cswap(d2, r2, d3, r3); cswap(d0, r0, d2, r2); cswap(d1, r1, d3, r3); cswap(d1, r1, d2, r2);
}

// The bound a box's entry distance is compared with: the best hit so far plus 2^-14 of it. The wide boxes carry a
// margin of only a few rounding errors, so that they stay thinner than the offset rays leave a surface with; the
// comparison is therefore taken wide instead: a triangle whose computed distance ties or undercuts the best hit by a
// rounding error is still reached. (No hit yet: 1e10 stays far above every distance.)
PT_DEV float wide_cull(float best_t) { return best_t * 1.00006103515625f; }

// The reciprocal direction for wide_step. The boxes' margin covers the hardware reciprocal's last-bit error. A
// huge finite slope stands in for the infinite one of an axis the ray is parallel to: both planes of a slab the
// origin is outside of stay on the same side, and no product is inf * 0.
PT_DEV F3 wide_inv(F3 d)
{
    F3 inv = f3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    return f3(__builtin_fminf(__builtin_fmaxf(inv.x, -1e30f), 1e30f), __builtin_fminf(__builtin_fmaxf(inv.y, -1e30f), 1e30f),
              __builtin_fminf(__builtin_fmaxf(inv.z, -1e30f), 1e30f));
}

// closest_hit over the 4-wide quantized nodes, one ray per lane from start to end (the persistent traversal
// kernel of the stage pipeline shares wide_step but regroups its lanes between passes). The visiting order is
// not the reference's; the closest hit and its tie rule do not depend on it.
template <int BLOCK, int LDS_DEPTH, bool COUNT = false>
PT_DEV Hit closest_hit_wide(const DScene& S, F3 o, F3 d, int tid, uint32_t& nnode, uint32_t& ntri)
{
    __shared__ uint2 s_stack[LDS_DEPTH * BLOCK];
    uint2 spill[kSpillDepth];
    int sp = 0;
    Hit best; best.t = 1e10f; best.tri = PT_NO_TRI; best.u = 0.f; best.v = 0.f;
    if (!finite3(o) || !finite3(d)) return best;
    const F3 inv = wide_inv(d);
    int cur = S.root_ref4;
    auto push = [&](int ref, float tn) {
        uint2 e = make_uint2((uint32_t)ref, __float_as_uint(tn));
        if (sp < LDS_DEPTH) s_stack[sp * BLOCK + tid] = e;
        else if (sp - LDS_DEPTH < kSpillDepth) spill[sp - LDS_DEPTH] = e;
        sp++;
    };
    auto pop_valid = [&]() -> int {
        while (sp > 0) {
            sp--;
            uint2 e;
            if (sp < LDS_DEPTH) e = s_stack[sp * BLOCK + tid];
            else {
                e = spill[(sp - LDS_DEPTH) < kSpillDepth ? (sp - LDS_DEPTH) : (kSpillDepth - 1)];
                asm volatile("" : "+v"(e.x), "+v"(e.y));
            }
            if (__uint_as_float(e.y) < wide_cull(best.t)) return (int)e.x;
        }
        return PT_TRAV_EMPTY;
    };
    for (;;) {
        while (cur >= 0) {
            const float4* np = reinterpret_cast<const float4*>(S.nodes4 + cur);
            float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            int r0, r1, r2, r3; float d0, d1, d2, d3;
            if (COUNT) nnode += 1;
            wide_step(n0, n1, n2, n3, o, inv, wide_cull(best.t), r0, r1, r2, r3, d0, d1, d2, d3);
            if (d3 < 1e30f) push(r3, d3);
            if (d2 < 1e30f) push(r2, d2);
            if (d1 < 1e30f) push(r1, d1);
            cur = d0 < 1e30f ? r0 : pop_valid();
        }
        if (cur == PT_TRAV_EMPTY) break;
        {
            const uint32_t enc = ~(uint32_t)cur;       // wide leaf: first record and count - 1
            const DTri* tp = S.tris + (enc & 0x3FFFFFFu);
            const int count = (int)(enc >> 26) + 1;
            for (int i = 0; i < count; i++) {
                float4 q0 = tp[i].q0, q1 = tp[i].q1, q2 = tp[i].q2;
                uint32_t id = __float_as_uint(q0.w) >> 1;       // (the id word carries a mark below the index, see DTri)
                float t, u, v;
                if (COUNT) ntri += 1;
                if (!tri_test(f3(q0.x, q0.y, q0.z), f3(q1.x, q1.y, q1.z), f3(q2.x, q2.y, q2.z), o, d, best.t, t, u, v)) continue;
                if (best.tri != PT_NO_TRI && t == best.t && id > best.tri) continue;
                best.t = t; best.tri = id; best.u = u; best.v = v;
            }
        }
        cur = pop_valid();
        if (cur == PT_TRAV_EMPTY) break;
    }
    return best;
}

// ---------------------------------------------------------------- surface resolve
struct Mat {
    F3 base, emis;
    float rough, aniso, metal, subsurf, sheen, coat, coat_rough, strans, alpha, ior;
    float G; F3 sigA, sigS;
};
struct Surf { float t; F3 n; Mat m; bool hit; bool inside; bool emis_map; };      // emis_map: the emission came out of a texture map

PT_DEV void default_mat(Mat& m)          // structs.cuh:75-94
{
    m.base = f3(0, 0, 0); m.emis = f3(0, 0, 0);
    m.rough = 0.01f; m.aniso = 0; m.metal = 0; m.subsurf = 0; m.sheen = 0; m.coat = 0; m.coat_rough = 0;
    m.strans = 0; m.alpha = 1.f; m.ior = 1.5f; m.G = 0; m.sigA = f3(0, 0, 0); m.sigS = f3(0, 0, 0);
}

// pathtracer.cu:140-181: what triangleIntersect does after accepting a hit, evaluated once for the
// final hit of a query (intermediate accepted hits never reach the caller).
// diagnostic builds (-DPT_STAMPS_PATH=1): section clocks, every pending memory operation drained at each mark
#ifndef PT_STAMPS_PATH
#define PT_STAMPS_PATH 0
#endif
#if PT_STAMPS_PATH
#define PT_MARK(st, k) do { if (st) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); (st)[k] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define PT_MARK(st, k) do { } while (0)
#endif

// LOCAL: the material and texture-descriptor tables are read from the caller's copies (a kernel's LDS) instead of S.mats
// and S.tex.
template <bool COUNT, bool LOCAL = false>
PT_DEV void resolve_hit(const DScene& S, const Hit& h, F3 d, Surf& s, uint32_t& taps, uint32_t& taps_r, bool visibility_only = false, unsigned long long* marks = nullptr,
                        const float4* local_mats = nullptr, const DTex* local_tex = nullptr)
{
    // visibility_only: the caller is a segment of the visibility walk, which looks at alpha, metallic, specTrans, ior and
    // the medium only (pathtracer.cu:279-295): the roughness, emission and normal maps are not fetched (the counting
    // build fetches them, as the reference does, so its tap counts stay the reference's)
    const bool all = COUNT || !visibility_only;
    s.t = h.t; s.hit = h.tri != PT_NO_TRI; s.inside = false; s.n = f3(0, 0, 0); s.emis_map = false;
    default_mat(s.m);
    if (!s.hit) return;
    const pt_triangle* T = S.tri_ref + h.tri;
    const float4* tq = reinterpret_cast<const float4*>(T);      // 112 bytes = 7 x 16
    uint32_t info = 0;
    if (S.tri_info) info = S.tri_info[h.tri];       // (issued with the triangle's loads: the material does not wait for them)
    float4 t0 = tq[0], t1 = tq[1], t2 = tq[2], t3 = tq[3], t4 = tq[4], t5 = tq[5], t6 = tq[6];
    F3 a = f3(t0.x, t0.y, t0.z), b = f3(t0.w, t1.x, t1.y), c = f3(t1.z, t1.w, t2.x);
    F3 nA = f3(t3.x, t3.y, t3.z), nB = f3(t3.w, t4.x, t4.y), nC = f3(t4.z, t4.w, t5.x);
    F2 tA = {t5.y, t5.z}, tB = {t5.w, t6.x}, tC = {t6.y, t6.z};
    int mi = S.tri_info ? (int)(info & 0x7FFFFFFFu) : __float_as_int(t6.w);
    const float4* mq = LOCAL ? local_mats + 8 * mi : reinterpret_cast<const float4*>(S.mats + mi);
    const DTex* texs = LOCAL ? local_tex : S.tex;
    float4 m0 = mq[0], m1 = mq[1], m2 = mq[2], m3 = mq[3], m4 = mq[4], m5 = mq[5], m6 = mq[6];
    s.m.base = f3(m0.x, m0.y, m0.z); s.m.emis = f3(m0.w, m1.x, m1.y);
    s.m.rough = m1.z; s.m.aniso = m1.w; s.m.metal = m2.x; s.m.subsurf = m2.y; s.m.sheen = m2.z; s.m.coat = m2.w;
    s.m.coat_rough = m3.x; s.m.strans = m3.y; s.m.alpha = m3.z; s.m.ior = m3.w;
    s.m.G = m4.x; s.m.sigA = f3(m4.y, m4.z, m4.w); s.m.sigS = f3(m5.x, m5.y, m5.z);
    int tx_d = __float_as_int(m5.w), tx_r = __float_as_int(m6.x), tx_m = __float_as_int(m6.y);
    int tx_e = __float_as_int(m6.z), tx_n = __float_as_int(m6.w);
    PT_MARK(marks, 2);

    float u = h.u, v = h.v;
    float w = 1.f - u - v;
    F3 e1 = b - a, e2 = c - a;
    F3 ng = normalize(cross(e1, e2));
    float mult = -sign_(dot(d, ng));
    s.n = normalize(nA * w + nB * u + nC * v) * mult;
    float px = tA.x * w + tB.x * u + tC.x * v;
    float py = tA.y * w + tB.y * u + tC.y * v;
    if (tx_d) {
        const DTex tx = texs[tx_d - 1];
        float4 r = tex4_any(tx, px, py);
        s.m.base = f3(r.x, r.y, r.z) * r.w;
        s.m.alpha = r.w;
        if (COUNT) taps++;
    }
    PT_MARK(marks, 3);
    if (tx_r && all) { const DTex tx = texs[tx_r - 1]; s.m.rough = tex1_any(tx, px, py); if (COUNT) taps_r++; }
    if (tx_m) { const DTex tx = texs[tx_m - 1]; s.m.metal = tex1_any(tx, px, py); if (COUNT) taps_r++; }
    if (tx_e && all) { const DTex tx = texs[tx_e - 1]; float r = tex1_any(tx, px, py); s.m.emis = f3(r, r, r); s.emis_map = true; if (COUNT) taps_r++; }
    if (tx_n && all) {
        F3 Tn, Bn;
        onb(s.n, Tn, Bn);
        const DTex tx = texs[tx_n - 1];
        float4 r = tex4_any(tx, px, py);
        s.n = to_world(Tn, Bn, s.n, f3(r.x * 2.f - 1.f, r.y * 2.f - 1.f, r.z * 2.f - 1.f));
        if (COUNT) taps++;
    }
    PT_MARK(marks, 4);
    s.n = ref_if_neg(s.n, -1.0f * d);
}

// ---------------------------------------------------------------- Disney BSDF (disney.cuh:29-460)
namespace disney {
PT_DEV float luminance(F3 c) { return 0.212671 * c.x + 0.715160 * c.y + 0.072169 * c.z; }
PT_DEV float schlick_weight(float u) { float m = clamp_(1.0f - u, 0.0f, 1.0f); float m2 = m * m; return m2 * m2 * m; }
PT_DEV F3 sample_ggx_vndf(F3 V, float ax, float ay, float r1, float r2)          // :41-59
{
    F3 Vh = normalize(f3(ax * V.x, ay * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    F3 T1 = lensq > 0.0 ? f3(-Vh.y, Vh.x, 0.0) * rsqrt_(lensq) : f3(1, 0, 0);
    F3 T2 = cross(Vh, T1);
    float r = sqrt_(r1);
    float phi = 2.0 * PT_PI * r2;
    float t1 = r * pt_cosf(phi);
    float t2 = r * pt_sinf(phi);
    float s = 0.5 * (1.0 + Vh.z);
    t2 = (1.0 - s) * sqrt_(1.0 - t1 * t1) + s * t2;
    double rem = 1.0 - t1 * t1 - t2 * t2;
    F3 Nh = t1 * T1 + t2 * T2 + sqrt_(rem > 0.0 ? rem : 0.0) * Vh;
    return normalize(f3(ax * Nh.x, ay * Nh.y, pt_fmaxf(0.0, Nh.z)));
}
PT_DEV float dielectric_fresnel(float ci, float eta)                             // :61-75
{
    float st2 = eta * eta * (1.0f - ci * ci);
    if (st2 > 1.0f) return 1.0f;
    double rem = 1.0 - st2;
    float ct = sqrt_(rem > 0.0 ? rem : 0.0);
    float rs = (eta * ct - ci) / (eta * ct + ci);
    float rp = (eta * ci - ct) / (eta * ci + ct);
    return 0.5f * (rs * rs + rp * rp);
}
PT_DEV F3 cosine_hemisphere(float r1, float r2)                                  // :76-85
{
    F3 dir;
    float r = sqrt_(r1);
    float phi = PT_TWO_PI * r2;
    dir.x = r * pt_cosf(phi);
    dir.y = r * pt_sinf(phi);
    dir.z = sqrt_(pt_fmaxf(0.0, 1.0 - dir.x * dir.x - dir.y * dir.y));
    return dir;
}
PT_DEV float gtr2_aniso(float nh, float hx, float hy, float ax, float ay)        // :103-109
{
    float a = hx / ax, b = hy / ay;
    float c = a * a + b * b + nh * nh;
    return 1.0 / (PT_PI * ax * ay * c * c);
}
PT_DEV float smith_g(float nv, float ag)                                         // :110-115
{
    float a = ag * ag, b = nv * nv;
    return (2.0 * nv) / (nv + sqrt_(a + b - a * b));
}
PT_DEV float smith_g_aniso(float nv, float vx, float vy, float ax, float ay)     // :116-122
{
    float a = vx * ax, b = vy * ay, c = nv;
    return (2.0 * nv) / (nv + sqrt_(a * a + b * b + c * c));
}
PT_DEV float gtr1(float nh, float a)                                             // :123-130
{
    if (a >= 1.0) return PT_INV_PI;
    float a2 = a * a;
    float t = 1.0 + (a2 - 1.0) * nh * nh;
    return (a2 - 1.0) / (PT_PI * pt_logf(a2) * t);
}
PT_DEV F3 sample_gtr1(float rgh, float r1, float r2)                             // :132-145
{
    float a = pt_fmaxf(0.001, rgh);
    float a2 = a * a;
    float phi = r1 * PT_TWO_PI;
    float ct = sqrt_((1.0 - pt_dm_pow(a2, 1.0 - r2)) / (1.0 - a2));
    float st = clamp_(pt_dm_sqrt(1.0 - (ct * ct)), 0.0, 1.0);
    float sp = pt_sinf(phi), cp = pt_cosf(phi);
    return f3(st * cp, st * sp, ct);
}
PT_DEV F3 eval_diffuse(const Mat& m, F3 Csheen, F3 V, F3 L, F3 H, float& pdf)    // :154-181
{
    pdf = 0.0;
    if (L.z <= 0.0) return f3(0, 0, 0);
    float lh = dot(L, H);
    float Rr = 2.0f * m.rough * lh * lh;
    float FL = schlick_weight(L.z), FV = schlick_weight(V.z);
    float Fretro = Rr * (FL + FV + FL * FV * (Rr - 1.0f));
    float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
    float Fss90 = 0.5 * Rr;
    float Fss = mix_(1.0f, Fss90, FL) * mix_(1.0f, Fss90, FV);
    float ss = 1.25f * (Fss * (1.0f / (L.z + V.z) - 0.5f) + 0.5f);
    float FH = schlick_weight(lh);
    F3 Fsheen = FH * m.sheen * Csheen;
    pdf = L.z * PT_INV_PI;
    return PT_INV_PI * m.base * mix_(Fd + Fretro, ss, m.subsurf) + Fsheen;
}
PT_DEV F3 eval_reflection(F3 V, F3 L, F3 H, F3 F, float ax, float ay, float& pdf) // :183-195
{
    pdf = 0.0;
    if (L.z <= 0.0) return f3(0, 0, 0);
    float D = gtr2_aniso(H.z, H.x, H.y, ax, ay);
    float G1 = smith_g_aniso(abs_(V.z), V.x, V.y, ax, ay);
    float G2 = G1 * smith_g_aniso(abs_(L.z), L.x, L.y, ax, ay);
    pdf = G1 * D / (4.0 * V.z);
    return F * D * G2 / (float)(4.0 * L.z * V.z);
}
PT_DEV F3 eval_refraction(const Mat& m, float eta, F3 V, F3 L, F3 H, F3 F, float ax, float ay, float& pdf) // :197-216
{
    pdf = 0.0;
    if (L.z >= 0.0) return f3(0, 0, 0);
    float lh = dot(L, H), vh = dot(V, H);
    float D = gtr2_aniso(H.z, H.x, H.y, ax, ay);
    float G1 = smith_g_aniso(abs_(V.z), V.x, V.y, ax, ay);
    float G2 = G1 * smith_g_aniso(abs_(L.z), L.x, L.y, ax, ay);
    float denom = lh + vh * eta;
    denom *= denom;
    float eta2 = eta * eta;
    float jac = abs_(lh) / denom;
    pdf = G1 * pt_fmaxf(0.0, vh) * D * jac / V.z;
    return pow3(m.base, 0.5f) * (1.0f - F) * D * G2 * abs_(vh) * jac * eta2 / abs_(L.z * V.z);
}
PT_DEV F3 eval_clearcoat(const Mat& m, F3 V, F3 L, F3 H, float& pdf)             // :218-233
{
    pdf = 0.0;
    if (L.z <= 0.0) return f3(0, 0, 0);
    float vh = dot(V, H);
    float F = mix_(0.04, 1.0, schlick_weight(vh));
    float D = gtr1(H.z, m.coat_rough);
    float G = smith_g(L.z, 0.25) * smith_g(V.z, 0.25f);
    float jac = 1.0 / (4.0 * vh);
    pdf = D * H.z * jac;
    return f3(F, F, F) * D * G;
}
// :234-347 (DisneyEval_2)
// inlined on purpose: as a call, the material record travels through scratch memory
__device__ inline F3 eval(const Mat& m, bool is_inside, F3 V, F3 N, F3 L, float& pdf)
{
    pdf = 0.0;
    float eta = is_inside ? m.ior : 1.f / m.ior;
    F3 f = f3(0, 0, 0);
    float aspect = rsqrt_(1.f - 0.9f * m.aniso);
    float ax = m.rough * m.rough / aspect;
    float ay = m.rough * m.rough * aspect;
    F3 T, B;
    onb(N, T, B);
    V = to_local(T, B, N, V);
    L = to_local(T, B, N, L);
    F3 H;
    if (L.z > 0.0) H = normalize(L + V);
    else H = normalize(L + V * eta);
    if (H.z < 0.0) H = -1.f * H;
    float F0 = (1.0 - eta) / (1.0 + eta);                                         // TintColors :92-102
    F0 *= F0;
    F3 Cspec0 = F0 * f3(1.f, 1.f, 1.f);
    F3 Csheen = f3(1.f, 1.f, 1.f);
    float w_diel = (1.0f - m.metal) * (1.0f - m.strans);
    float w_metal = m.metal;
    float w_glass = (1.0f - m.metal) * m.strans;
    float sw = schlick_weight(V.z);
    float p_diff = w_diel * luminance(m.base);
    float p_diel = w_diel * luminance(mix_(Cspec0, f3(1, 1, 1), sw));
    float p_metal = w_metal * luminance(mix_(m.base, f3(1, 1, 1), sw));
    float p_glass = w_glass;
    float p_coat = 0.25f * m.coat;
    float inv_tot = 1.0f / (p_diff + p_diel + p_metal + p_glass + p_coat);
    p_diff *= inv_tot; p_diel *= inv_tot; p_metal *= inv_tot; p_glass *= inv_tot; p_coat *= inv_tot;
    bool refl = L.z * V.z > 0;
    float tp = 0.0;
    float vh = abs_(dot(V, H));
    if (p_diff > 0.0 && refl) {
        f = f + eval_diffuse(m, Csheen, V, L, H, tp) * w_diel;
        pdf += tp * p_diff;
    }
    if (p_diel > 0.0 && refl) {
        float F = (dielectric_fresnel(vh, 1.0 / m.ior) - F0) / (1.0 - F0);
        f = f + eval_reflection(V, L, H, mix_(Cspec0, f3(1, 1, 1), F), ax, ay, tp) * w_diel;
        pdf += tp * p_diel;
    }
    if (p_metal > 0.0 && refl) {
        F3 F = mix_(m.base, f3(1, 1, 1), schlick_weight(vh));
        f = f + eval_reflection(V, L, H, F, ax, ay, tp) * w_metal;
        pdf += tp * p_metal;
    }
    if (p_glass > 0.0 && abs_(m.ior - 1.f) > 0.01f) {
        float F = dielectric_fresnel(vh, eta);
        if (refl) {
            f = f + eval_reflection(V, L, H, f3(F, F, F), ax, ay, tp) * w_glass;
            pdf += tp * p_glass * F;
        } else {
            f = f + eval_refraction(m, eta, V, L, H, f3(F, F, F), ax, ay, tp) * w_glass;
            pdf += tp * p_glass * (1.0f - F);
        }
    }
    if (p_coat > 0.0 && refl) {
        f = f + eval_clearcoat(m, V, L, H, tp) * 0.25f * m.coat;
        pdf += tp * p_coat;
    }
    return f * abs_(L.z);
}
// :348-457 (DisneySample_2 up to the final evaluation). Returns true when the sampled event is the
// index-matched pass-through (f = 1, pdf = 1); otherwise the caller evaluates the BSDF for L.
PT_DEV bool sample_dir(const Mat& m, bool is_inside, F3 V, F3 N, F3& L, Rng& rng, bool& inside)
{
    float eta = is_inside ? m.ior : 1.f / m.ior;
    float aspect = rsqrt_(1.f - 0.9f * m.aniso);
    float ax = m.rough * m.rough / aspect;
    float ay = m.rough * m.rough * aspect;
    float r1 = rng_next(rng);
    float r2 = rng_next(rng);
    F3 T, B;
    onb(N, T, B);
    V = to_local(T, B, N, V);
    float F0 = (1.0 - eta) / (1.0 + eta);
    F0 *= F0;
    F3 Cspec0 = F0 * f3(1.f, 1.f, 1.f);
    float w_diel = (1.0 - m.metal) * (1.0 - m.strans);
    float w_metal = m.metal;
    float w_glass = (1.0 - m.metal) * m.strans;
    float sw = schlick_weight(V.z);
    float p_diff = w_diel * luminance(m.base);
    float p_diel = w_diel * luminance(mix_(Cspec0, f3(1, 1, 1), sw));
    float p_metal = w_metal * luminance(mix_(m.base, f3(1, 1, 1), sw));
    float p_glass = w_glass;
    float p_coat = 0.25 * m.coat;
    float inv_tot = 1.0 / (p_diff + p_diel + p_metal + p_glass + p_coat);
    p_diff *= inv_tot; p_diel *= inv_tot; p_metal *= inv_tot; p_glass *= inv_tot; p_coat *= inv_tot;
    float c0 = p_diff;
    float c1 = c0 + p_diel;
    float c2 = c1 + p_metal;
    float c3 = c2 + p_glass;
    float r3 = rng_next(rng);
    if (r3 < c0) {
        L = cosine_hemisphere(r1, r2);
    } else if (r3 < c2) {
        F3 H = sample_ggx_vndf(V, ax, ay, r1, r2);
        if (H.z < 0.0) H = -1.f * H;
        L = normalize(reflect_(-1.f * V, H));
    } else if (r3 < c3) {
        if (abs_(m.ior - 1.f) <= 0.01f) {
            L = to_world(T, B, N, -1.f * V);
            inside = !inside;
            return true;
        }
        F3 H = sample_ggx_vndf(V, ax, ay, r1, r2);
        float F = dielectric_fresnel(abs_(dot(V, H)), eta);
        if (H.z < 0.0) H = -1.f * H;
        r3 = (r3 - c2) / (c3 - c2);
        if (r3 < F) {
            L = normalize(reflect_(-1.f * V, H));
        } else {
            L = normalize(refract_(-1.f * V, H, eta));
            inside = !inside;
        }
    } else {
        F3 H = sample_gtr1(m.coat_rough, r1, r2);
        if (H.z < 0.0) H = -1.f * H;
        L = normalize(reflect_(-1.f * V, H));
    }
    L = to_world(T, B, N, L);
    return false;
}
} // namespace disney

// ---------------------------------------------------------------- phase function (phase_function.cu:5-20)
PT_DEV float eval_hg(float mu, float g)
{
    return PT_INV_4_PI * (1. - g * g) / (pt_powf(1. + g * g - 2.0f * g * mu, 1.5f));
}
PT_DEV F3 sample_hg(F3 d, float g, Rng& rng)
{
    float xi = rng_next(rng);
    float t = (1.f - g * g) / (1.f - g + 2.0f * g * xi);
    float mu = (0.5f / g) * ((1.f + g * g) - t * t);
    float phi = PT_TWO_PI * rng_next(rng);
    float st = sqrt_(pt_fmaxf(0.f, 1.f - mu * mu));
    F3 T, B;
    onb(d, T, B);
    return normalize(st * pt_cosf(phi) * T + st * pt_sinf(phi) * B + mu * d);
}

// ---------------------------------------------------------------- sky (envmap_cuda.cu:12-74, pathtracer.cu:340-363)
PT_DEV float4 eval_envmap(const DScene& S, F3 dir)
{
    float theta = pt_acosf(clamp_(dir.y, -1.0f, 1.0f));
    float u = (PT_PI + pt_atan2f(dir.z, dir.x)) * PT_INV_TWO_PI;
    float v = theta * PT_INV_PI;
    float4 r = tex4((const PT_GLOBAL float*)S.env_rgba, S.env_w, S.env_h, u, v);
    float pdf = disney::luminance(f3(r.x, r.y, r.z)) / S.env_sum;
    return make_float4(r.x, r.y, r.z, (pdf * S.env_w * S.env_h) / (PT_TWO_PI * PT_PI * pt_sinf(theta)));
}
// binarySearch of envmap_cuda.cu:28-57 on one table of n entries: the index the reference's bisection ends at.
template <bool COUNT>
PT_DEV int env_bisect(const float* __restrict__ table, int n, float xi, uint32_t& taps_r)
{
    int lower = 0, upper = n - 1;
    while (lower < upper) {
        int mid = (lower + upper) >> 1;
        float c = table[mid];
        if (COUNT) taps_r++;
        if (xi < c) upper = mid; else lower = mid + 1;
    }
    return lower;
}
// The same index with fewer dependent loads. Entries 1.. of a table are non-decreasing (a running sum, filtered),
// so the bisection ends at the first index >= 1 whose entry exceeds xi wherever it starts; the guide narrows the
// start to the bucket of xi. Entry 0 is the odd one (the filter wraps it around to the end of the table) and the
// bisection only consults it when xi is below entry 1. Whether this holds for a given map is not assumed: every
// breakpoint is compared with env_bisect when the map is uploaded.
PT_DEV int env_guided(const float* __restrict__ table, int n, float xi, uint32_t guide)
{
    int lower = (int)(guide & 0xffffu), upper = (int)(guide >> 16);
    while (lower < upper) {
        int mid = (lower + upper) >> 1;
        if (xi < table[mid]) upper = mid; else lower = mid + 1;
    }
    if (lower == 1 && n > 1 && xi < table[0]) lower = 0;
    return lower;
}
PT_DEV uint32_t env_bucket(float xi, float scale, int buckets)
{
    float f = xi * scale;
    int k = f >= 0.f ? (int)f : 0;          // NaN and negatives go to bucket 0
    return (uint32_t)(k < buckets ? k : buckets - 1);
}
// Row and column sampleEnvMap's two searches end at (envmap_cuda.cu:28-57), clamped to the map as it does.
template <bool COUNT>
PT_DEV void env_search_texel(const DScene& S, float xi, uint32_t& taps_r, int& x, int& y)
{
    int row, col;
    if (!COUNT && S.env_fast) {
        if (S.env_fast2) {
            // one load names the row and the column range when xi's bucket lies inside a row (nearly always:
            // there are about as many buckets as texels, and bright texels, where samples land, span several)
            uint2 g = S.env_texel_guide[env_bucket(xi, S.env_texel_scale, kEnvTexelBuckets)];
            if (g.x != 0xFFFFFFFFu) {
                y = (int)g.x;
                col = env_guided(S.env_col_table + (size_t)y * S.env_w, S.env_w, xi, g.y);
                x = (int)clamp_((float)col, 0.f, (float)(S.env_w - 1));
                return;
            }
        }
        row = env_guided(S.env_row_table, S.env_h, xi, S.env_row_guide[env_bucket(xi, S.env_row_scale, kEnvRowBuckets)]);
        y = (int)clamp_((float)row, 0.f, (float)(S.env_h - 1));
        col = env_guided(S.env_col_table + (size_t)y * S.env_w, S.env_w, xi, S.env_col_guide[(size_t)y * kEnvColBuckets + env_bucket(xi, S.env_col_scale, kEnvColBuckets)]);
        x = (int)clamp_((float)col, 0.f, (float)(S.env_w - 1));
        return;
    }
    // the value tex1(cdf, 1, mid/H) the reference looks up in the row search and tex1(cdf, mid/W, y/H) in the
    // column search, tabulated when the map was uploaded
    row = env_bisect<COUNT>(S.env_row_table, S.env_h, xi, taps_r);
    y = (int)clamp_((float)row, 0.f, (float)(S.env_h - 1));
    col = env_bisect<COUNT>(S.env_col_table + (size_t)y * S.env_w, S.env_w, xi, taps_r);
    x = (int)clamp_((float)col, 0.f, (float)(S.env_w - 1));
}
// What sampleEnvMap (envmap_cuda.cu:59-74) returns for the texel its searches ended at: direction + pdf; color
// receives the radiance. Every value depends on the texel alone.
PT_DEV float4 sample_envmap_texel(const DScene& S, int x, int y, F3& color)
{
    F2 uv = {(float)x / (float)S.env_w, (float)y / (float)S.env_h};
    float4 r = tex4((const PT_GLOBAL float*)S.env_rgba, S.env_w, S.env_h, uv.x, uv.y);
    color = f3(r.x, r.y, r.z);
    float pdf = disney::luminance(color) / S.env_sum;
    float phi = uv.x * PT_TWO_PI;
    float theta = uv.y * PT_PI;
    float sn = pt_sinf(theta);
    if (sn == 0.f) pdf = 0.f;
    return make_float4(-sn * pt_cosf(phi), pt_cosf(theta), -sn * pt_sinf(phi), (pdf * S.env_w * S.env_h) / (PT_TWO_PI * PT_PI * sn));
}
// One light sample: a draw, the two searches, then the texel's values, read from the per-texel table made when the
// map was uploaded (the same function evaluated once per texel: two loads instead of four filtered lookups and
// four trigonometric functions per sample) or computed on the spot.
template <bool COUNT>
PT_DEV float4 sample_envmap(const DScene& S, F3& color, Rng& rng, uint32_t& taps, uint32_t& taps_r)
{
    int x, y;
    env_search_texel<COUNT>(S, rng_next(rng) * S.env_sum, taps_r, x, y);
    if (COUNT) taps++;
    if (!COUNT && S.env_sample_table) {
        const float4* e = S.env_sample_table + 2 * ((size_t)y * S.env_w + x);
        const float4 dir = e[0], c = e[1];
        color = f3(c.x, c.y, c.z);
        return dir;
    }
    return sample_envmap_texel(S, x, y, color);
}
PT_DEV float sky_hg(float g, float c) { float gg = g * g; return (1. - gg) / pt_dm_pow(1. + gg - 2. * g * c, 1.5); }
PT_DEV float sky_rayleigh(float u) { return 0.75 * (1. + clamp_(u * u, 0., 1.)); }
__device__ __noinline__ F3 sky_gradient(F3 rd, F3 sun)
{
    F3 sunColor = mix_(f3(1., 0.9, 0.8), f3(1., 0.2, 0.), pt_dm_exp(-abs_(sun.y) * 5.));
    float c0 = dot(rd, sun);
    float blend = smoothstep_(-1., 0.6, c0);
    F3 day = mix_(f3(0.3, 0.6, 1.) * 0.7 * sky_rayleigh(c0), f3(0.9, 0.95, 1.), pt_dm_exp(-abs_(rd.y) * 5.));
    F3 even = mix_(f3(0.3, 0.6, 1.) * 0.7, mix_(f3(0.2, 0.3, 0.4), f3(1., 0.2, 0.), blend), pt_dm_exp(-abs_(rd.y) * 5.));
    F3 sunDisk = (float)(0.02 / length(rd - sun)) * sunColor;
    F3 night = f3(0.05, 0.1, 0.2) + clamp_(0.02 / length(rd + sun), 0., 1.) * f3(0.7, 0.8, 1.) * 0.6;
    F3 dayF = mix_(day, even, pt_dm_exp(-abs_(sun.y) * 5.)) + sunDisk;
    float earth = smoothstep_(-0.02, 0.02, rd.y) * 0.5 + 0.5;
    float mieM = 1. - abs_(dot(sun, f3(0., 1., 0.))) + 0.3;
    return mix_(night, dayF + sky_hg(0.6, c0) * sunColor * 0.3 * mieM, smoothstep_(-0.2, 0.0, sun.y)) * earth * 0.75;
}

// ---------------------------------------------------------------- tonemap (tonemapping.cuh:17-63, pathtracer.cu:588-599)
PT_DEV F3 agx_tonemap(F3 x)
{
    const float min_ev = -12.47393f, max_ev = 4.026069f;
    x = f3(0.842479062253094f * x.x + 0.0423282422610123f * x.y + 0.0423756549057051f * x.z,
           0.0784335999999992f * x.x + 0.878468636469772f * x.y + 0.0784336f * x.z,
           0.0792237451477643f * x.x + 0.0791661274605434f * x.y + 0.879142973793104f * x.z);
    x = f3(clamp_(pt_log2f(x.x), min_ev, max_ev), clamp_(pt_log2f(x.y), min_ev, max_ev), clamp_(pt_log2f(x.z), min_ev, max_ev));
    x = (x - min_ev) / (max_ev - min_ev);
    F3 x2 = x * x;
    F3 x4 = x2 * x2;
    x = 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 + 0.1191f * x - 0.00232f;
    const F3 lw = f3(0.2126f, 0.7152f, 0.0722f);
    float luma = dot(lw, x);
    F3 offset = f3(0.f, 0.f, 0.f), slope = f3(1.f, 1.f, 1.f);
    float power = 1.35f, sat = 1.4f;
    x = pow3(x * slope + offset, power);
    x = luma + sat * (x - luma);
    return f3(1.19687900512017f * x.x - 0.0528968517574562f * x.y - 0.0529716355144438f * x.z,
              -0.0980208811401368f * x.x + 1.15190312990417f * x.y - 0.0980434501171241f * x.z,
              -0.0990297440797205f * x.x - 0.0989611768448433f * x.y + 1.15107367264116f * x.z);
}
PT_DEV uint8_t to_byte(float c)
{
    float s = c * 255;
    if (!(s == s)) return 0;
    if (s <= -2147483648.0f) return 0;
    if (s >= 2147483648.0f) return 255;
    int v = (int)s;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

} // namespace ptamd
