// Per-sample state machine of the integrator.
//
// The reference integrator ("Cuda Pathtracer/core/CUDA/pathtracer.cu:267-543") nests its closest-hit
// queries: pathtrace -> sampleSkyboxOnBounce -> visibility -> map, each consuming random draws in
// sequence.  Here the same computation is cut at every closest-hit query, so that a sample is a
// small state record that alternates between "needs a ray traced" and "consumes a hit":
//
//     begin_sample -> [trace] -> on_path_hit -+-> (light sample) [trace] -> on_shadow_hit ... -+
//                        ^                    |                                              |
//                        +---- after_light <--+----------------------------------------------+
//
// state ST_PATH / ST_SHADOW: the sample waits for the closest hit of its path / shadow ray;
// ST_AFTER: the light sample of the vertex is finished (transmittance in `light`), the vertex still
// has to pick its next direction; ST_DONE: the sample's radiance is final.
//
// One traversal call site serves camera, bounce and shadow-walk rays alike, and the same steps can
// run inside one kernel (a lane loops over them) or as separate stage kernels over queues.
// The order of random draws and of floating-point operations per sample is that of the reference.
#pragma once
#include "pt_device.hpp"

namespace ptamd {

enum : int { ST_PATH = 0, ST_SHADOW = 1, ST_AFTER = 2, ST_DONE = 3 };

struct Path {
    Rng rng;
    F3 o, d;                 // current path ray; after a hit, o is the vertex position
    float thr, pdf, result;  // scalar throughput of the chosen channel, last sampling pdf, radiance
    int ch, bounce, state;
    bool inside, surface_scatter;
    // shadow walk (pathtracer.cu:267-297)
    F3 so, sd, light;
    int seg;
    bool sh_inside;
    // light sample context (pathtracer.cu:298-339)
    F3 env_color;
    float env_pdf;
    bool nee_surface, nee_skip;
    // vertex context kept across the shadow walk
    Mat vm;
    F3 vn;
    bool v_inside;
    int lobes;               // which lobes of the vertex material carry weight (lobe_class), for queue ordering only
};

// Which branches DisneyEval_2 / DisneySample_2 take for a material (disney.cuh:267-344): bit 0 diffuse + dielectric
// reflection, bit 1 metal, bit 2 glass, bit 3 clearcoat. Used to group vertices with like materials in one wave; no
// result depends on it.
PT_DEV int lobe_class(const Mat& m)
{
    return ((1.0f - m.metal) * (1.0f - m.strans) > 0.f ? 1 : 0) | (m.metal > 0.f ? 2 : 0) | ((1.0f - m.metal) * m.strans > 0.f ? 4 : 0) | (m.coat > 0.f ? 8 : 0);
}

struct Tally { uint32_t rays, box, tri, verts, envs, taps, taps_r, path_rays; };

// pathtracer.cu:559-579 + 527: pixel seed, jitter, camera ray, optional thin lens, channel pick.
PT_DEV void begin_sample(Path& P, const DCamera& cam, int x, int y, int frame)
{
    int idx = x + y * cam.W;
    rng_seed(P.rng, (uint32_t)idx, (uint32_t)frame, 0);
    float jx = rng_next(P.rng);
    float jy = rng_next(P.rng);
    float uvx = ((float)x + jx * 2.0f - 1.f) / cam.W;
    float uvy = 1.f - ((float)y + jy * 2.0f - 1.f) / cam.H;
    float cx = uvx - 0.5f, cy = uvy - 0.5f;
    cx *= (float)cam.W / cam.H;
    F3 origin = ld3(cam.origin);
    F3 ww = ld3(cam.ww), uu = ld3(cam.uu), vv = ld3(cam.vv);
    F3 rd = normalize(uu * cx + vv * cy + ww * cam.fov);
    if (cam.dof > 0.f) {
        float RdoT = dot(ww, rd);
        F3 target = origin + rd * cam.focal / RdoT;
        float z = rng_next(P.rng) * 2.0f - 1.0f;                 // cuda_utility.cuh:33-40
        float a = rng_next(P.rng) * PT_PI * 2.;
        float r = sqrt_(1.0f - z * z);
        origin = origin + f3(r * pt_cosf(a), r * pt_sinf(a), z) * cam.dof;
        rd = normalize(target - origin);
    }
    P.o = origin; P.d = rd;
    P.thr = 1.0f; P.pdf = 0.f; P.result = 0.f;
    P.bounce = 0; P.inside = false; P.surface_scatter = false; P.lobes = 0;
    P.state = ST_PATH;
    P.ch = 0;
    if (cam.is_rendering) {
        P.ch = int(rng_next(P.rng) * 3.);
        if (P.ch > 2) P.state = ST_DONE;        // a draw of exactly 1.0 selects no channel (pathtracer.cu:527-539)
    }
}

// pathtracer.cu:484-490 and the loop bound of :405
PT_DEV void roulette_and_advance(Path& P, const DCamera& cam)
{
    float p = P.thr;
    if (rng_next(P.rng) > p) { P.state = ST_DONE; return; }
    P.thr *= 1.0f / p;
    P.bounce++;
    P.state = P.bounce < cam.max_bounces ? ST_PATH : ST_DONE;
}

// Everything after the light sample of a vertex: pathtracer.cu:318-338 (contribution), :449-451 or
// :471-480 (next direction), then roulette.
template <bool COUNT>
PT_DEV void after_light(const DScene& S, const DCamera& cam, Path& P, Tally& tl)
{
    F3 trans = P.light;
    F3 c = f3(0.f, 0.f, 0.f);
    if (!P.nee_skip && !(dot(trans, trans) <= 0.05f)) {
        float bp;
        F3 bf;
        F3 L = P.sd;
        if (P.nee_surface) {
            bf = disney::eval(P.vm, P.v_inside, -1.f * P.d, P.vn, L, bp);
        } else {
            bp = eval_hg(dot(L, P.d), P.vm.G);
            bf = bp * f3(1.f, 1.f, 1.f);
        }
        if (bp > 0.f) {
            float mis = power_heuristic(P.env_pdf, bp);
            if (mis > 0.f) c = mis * P.env_color * bf * trans / P.env_pdf;
        }
    }
    P.result += getN(c, P.ch) * P.thr;
    if (P.nee_surface) {
        F3 L;
        F3 V = P.d * -1.f;
        F3 f;
        // DisneySample_2 (disney.cuh:348-460): direction first, then the evaluation for it
        float eta_unused = 0.f; (void)eta_unused;
        F3 T, B;
        bool pass = disney::sample_dir(P.vm, P.v_inside, V, P.vn, L, P.rng, P.inside);
        if (pass) { f = f3(1.f, 1.f, 1.f); P.pdf = 1.f; }
        else {
            // the reference evaluates with V taken to the local frame and back (disney.cuh:365,457)
            onb(P.vn, T, B);
            F3 Vrt = to_world(T, B, P.vn, to_local(T, B, P.vn, V));
            f = disney::eval(P.vm, P.v_inside, Vrt, P.vn, L, P.pdf);
        }
        P.d = L;
        P.result += getN(P.vm.emis, P.ch) * P.thr;
        if (P.pdf > 0.f) P.thr *= getN(f / P.pdf, P.ch);
        else { P.state = ST_DONE; return; }
        P.o = P.o + P.d * 0.01f;
    } else {
        F3 mem = P.d;
        P.d = sample_hg(P.d, P.vm.G, P.rng);
        P.pdf = eval_hg(dot(mem, P.d), P.vm.G);
    }
    roulette_and_advance(P, cam);
}

// Start of sampleSkyboxOnBounce (pathtracer.cu:298-317): draw the light direction and open the
// shadow walk.  In sun/sky mode there is no light sampling and no draw.
template <bool COUNT>
PT_DEV void start_light(const DScene& S, const DCamera& cam, Path& P, F3 origin, bool is_surface, Tally& tl)
{
    P.nee_surface = is_surface;
    if (S.sky_mode == PT_SKY_SUNSKY) {
        P.nee_skip = true;
        P.light = f3(0.f, 0.f, 0.f);
        P.state = ST_AFTER;
        return;
    }
    P.nee_skip = false;
    float4 dat = sample_envmap<COUNT>(S, P.env_color, P.rng, tl.taps, tl.taps_r);
    if (COUNT) tl.envs++;
    P.sd = f3(dat.x, dat.y, dat.z);
    P.env_pdf = dat.w;
    P.so = origin;
    P.light = f3(1.f, 1.f, 1.f);
    P.seg = 0;
    P.sh_inside = !is_surface;
    P.state = ST_SHADOW;
}

// The path ray left the scene (pathtracer.cu:409-425): the environment's radiance, weighted against the light
// sampling of the last vertex, ends the sample.
template <bool COUNT>
PT_DEV void on_path_miss(const DScene& S, Path& P, Tally& tl)
{
    F3 envc; float envp;
    if (S.sky_mode == PT_SKY_SUNSKY) { envc = sky_gradient(P.d, ld3(S.sun)); envp = 0.f; }
    else { float4 e = eval_envmap(S, P.d); envc = f3(e.x, e.y, e.z); envp = e.w; if (COUNT) tl.taps++; }
    float mis = 1.f;
    if (P.bounce > 0) mis = power_heuristic(P.pdf, envp);
    if (!P.surface_scatter) mis = 1.f;
    if (S.sky_mode == PT_SKY_SUNSKY) mis = 1.f;
    if (mis > 0.f) P.result += getN(mis * envc * P.thr, P.ch);
    P.state = ST_DONE;
}

// One vertex of pathtrace (pathtracer.cu:426-481), given the resolved surface, up to the point where the sample needs
// its next ray.
template <bool COUNT>
PT_DEV void on_path_vertex(const DScene& S, const DCamera& cam, Path& P, Surf& info, Tally& tl)
{
    info.inside = P.inside;
    if (COUNT) tl.verts++;
    info.m.rough = pt_fmaxf(info.m.rough, 0.0001f);
    P.surface_scatter = false;
    P.vm = info.m; P.vn = info.n; P.v_inside = info.inside;
    P.lobes = lobe_class(info.m);
    if (info.inside) {
        F3 sigT = info.m.sigA + info.m.sigS;
        float zeta = rng_next(P.rng);
        if (zeta < getN(info.m.sigA / sigT, P.ch)) {
            P.thr *= pt_expf(-getN(sigT * info.t, P.ch));
        } else {
            float sd = -pt_logf(1.f - rng_next(P.rng)) / getN(sigT, P.ch);
            if (sd < info.t) {
                P.o = P.o + P.d * sd;
                start_light<COUNT>(S, cam, P, P.o, false, tl);
                return;
            }
        }
    }
    P.o = P.o + P.d * info.t;
    float rC = rng_next(P.rng);
    if (rC < info.m.alpha) {
        P.surface_scatter = true;
        start_light<COUNT>(S, cam, P, P.o + info.n * 0.01f, true, tl);
        return;
    }
    P.o = P.o + P.d * 0.01f;
    roulette_and_advance(P, cam);
}

// pathtracer.cu:406-481: what a path ray's closest hit leads to
template <bool COUNT>
PT_DEV void on_path_hit(const DScene& S, const DCamera& cam, Path& P, Surf& info, Tally& tl)
{
    if (!info.hit) { on_path_miss<COUNT>(S, P, tl); return; }
    on_path_vertex<COUNT>(S, cam, P, info, tl);
}

// One segment of visibility (pathtracer.cu:279-295)
template <bool COUNT>
PT_DEV void on_shadow_hit(const DScene& S, const DCamera& cam, Path& P, const Surf& info, Tally& tl)
{
    if (!info.hit) { P.state = ST_AFTER; return; }
    bool refractive = (1.f - info.m.metal) * info.m.strans > 0.f && abs_(info.m.ior - 1.f) <= 0.01f;
    bool alpha = rng_next(P.rng) > info.m.alpha;
    if (P.sh_inside) {
        F3 a = -info.t * (info.m.sigA + info.m.sigS);
        P.light = P.light * f3(pt_expf(a.x), pt_expf(a.y), pt_expf(a.z));
    }
    if (!(refractive || alpha)) { P.light = f3(0.f, 0.f, 0.f); P.state = ST_AFTER; return; }
    else if (refractive) P.sh_inside = !P.sh_inside;
    P.so = P.so + P.sd * (info.t + 0.1f);
    P.seg++;
    if (P.seg >= 32) P.state = ST_AFTER;
}

} // namespace ptamd
