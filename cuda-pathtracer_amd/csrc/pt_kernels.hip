// HIP kernels of the render path (gfx950) and their launchers.
#include <hip/hip_runtime.h>
#include "pt_path.hpp"
#include "pt_launch.h"

#pragma clang fp contract(off)

namespace ptamd {

// ---------------------------------------------------------------- closest-hit batch
// One ray per lane, 256 lanes per workgroup; the traversal stack of each lane sits in LDS.
template <bool COUNT>
__global__ __launch_bounds__(256) void trace_closest_kernel(DScene S, const pt_ray* __restrict__ rays, int n,
                                                            pt_hit* __restrict__ hits, uint32_t* __restrict__ nbox_out,
                                                            uint32_t* __restrict__ ntri_out)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* rp = reinterpret_cast<const float4*>(rays + i);
    float4 r0 = rp[0], r1 = rp[1];
    uint32_t nb = 0, nt = 0;
    Hit h = closest_hit<256, kLdsDepth, COUNT>(S, f3(r0.x, r0.y, r0.z), f3(r1.x, r1.y, r1.z), threadIdx.x, nb, nt);
    pt_hit out; out.t = h.t; out.tri = h.tri;
    hits[i] = out;
    if (COUNT) { nbox_out[i] = nb; ntri_out[i] = nt; }
}

void launch_trace_closest(const DScene& S, const pt_ray* rays_dev, int n, pt_hit* hits_dev, uint32_t* nbox_dev, uint32_t* ntri_dev, hipStream_t stream)
{
    if (n <= 0) return;
    dim3 grid((n + 255) / 256), block(256);
    if (nbox_dev && ntri_dev)
        hipLaunchKernelGGL(trace_closest_kernel<true>, grid, block, 0, stream, S, rays_dev, n, hits_dev, nbox_dev, ntri_dev);
    else
        hipLaunchKernelGGL(trace_closest_kernel<false>, grid, block, 0, stream, S, rays_dev, n, hits_dev, nbox_dev, ntri_dev);
}

// ---------------------------------------------------------------- whole-sample kernel
// One pixel per lane, an 8x8 pixel tile per wave (pathtracer.cu:604 uses the same tile), frames
// [frame_begin, frame_end) in order so the accumulator sees the same additions as a frame loop.
// Each lane runs the sample state machine; all its rays (camera, bounce, shadow segments) go
// through one traversal call site.
template <bool COUNT>
__global__ __launch_bounds__(64) void render_samples_kernel(DScene S, DCamera cam, int frame_begin, int frame_end,
                                                            float* __restrict__ accum, DCounters* __restrict__ ctr)
{
    int tiles_x = (cam.W + 7) >> 3;
    int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    int x = tx * 8 + (threadIdx.x & 7), y = ty * 8 + (threadIdx.x >> 3);
    if (x >= cam.W || y >= cam.H) return;
    size_t pix = (size_t)(x + y * cam.W) * 3;
    F3 acc = f3(accum[pix], accum[pix + 1], accum[pix + 2]);
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t nsamples = 0;
    for (int frame = frame_begin; frame < frame_end; frame++) {
        Path P;
        begin_sample(P, cam, x, y, frame);
        nsamples++;
        F3 color = f3(0.f, 0.f, 0.f);
        if (!cam.is_rendering) {
            // preview: pathtracer.cu:504-525
            Hit h = closest_hit<64, kLdsDepth, COUNT>(S, P.o, P.d, threadIdx.x, tl.box, tl.tri);
            if (COUNT) { tl.rays++; tl.path_rays++; }
            Surf info;
            resolve_hit<COUNT>(S, h, P.d, info, tl.taps, tl.taps_r);
            if (info.hit) {
                float NoV = pt_fmaxf(dot(P.d * -1.0f, info.n), 0.f);
                color = NoV * info.m.base;
            } else if (S.sky_mode == PT_SKY_SUNSKY) {
                color = sky_gradient(P.d, ld3(S.sun));
            } else {
                float4 e = eval_envmap(S, P.d);
                if (COUNT) tl.taps++;
                color = f3(e.x, e.y, e.z);
            }
        } else {
            while (P.state != ST_DONE) {
                if (P.state == ST_AFTER) { after_light<COUNT>(S, cam, P, tl); continue; }
                bool shadow = P.state == ST_SHADOW;
                F3 ro = shadow ? P.so : P.o;
                F3 rd = shadow ? P.sd : P.d;
                Hit h = closest_hit<64, kLdsDepth, COUNT>(S, ro, rd, threadIdx.x, tl.box, tl.tri);
                if (COUNT) { tl.rays++; if (!shadow) tl.path_rays++; }
                Surf info;
                resolve_hit<COUNT>(S, h, rd, info, tl.taps, tl.taps_r);
                if (shadow) on_shadow_hit<COUNT>(S, cam, P, info, tl);
                else on_path_hit<COUNT>(S, cam, P, info, tl);
            }
            F3 c = f3(0.f, 0.f, 0.f);
            if (P.ch == 0) c.x = P.result; else if (P.ch == 1) c.y = P.result; else if (P.ch == 2) c.z = P.result;
            color = c * 3.f;
        }
        F3 c0 = max3(color, f3(0.f, 0.f, 0.f));
        acc = acc + c0;
    }
    accum[pix] = acc.x; accum[pix + 1] = acc.y; accum[pix + 2] = acc.z;
    if (COUNT) {
        atomicAdd(&ctr->rays, (unsigned long long)tl.rays);
        atomicAdd(&ctr->path_rays, (unsigned long long)tl.path_rays);
        atomicAdd(&ctr->samples, (unsigned long long)nsamples);
        atomicAdd(&ctr->box_tests, (unsigned long long)tl.box);
        atomicAdd(&ctr->tri_tests, (unsigned long long)tl.tri);
        atomicAdd(&ctr->shade_vertices, (unsigned long long)tl.verts);
        atomicAdd(&ctr->env_samples, (unsigned long long)tl.envs);
        atomicAdd(&ctr->tex_taps, (unsigned long long)tl.taps);
        atomicAdd(&ctr->tex_taps_r, (unsigned long long)tl.taps_r);
    }
}

void launch_render_samples(const DScene& S, const DCamera& cam, int frame_begin, int frame_end, float* accum_dev, DCounters* ctr_dev, hipStream_t stream)
{
    int tiles = ((cam.W + 7) / 8) * ((cam.H + 7) / 8);
    dim3 grid(tiles), block(64);
    if (ctr_dev)
        hipLaunchKernelGGL(render_samples_kernel<true>, grid, block, 0, stream, S, cam, frame_begin, frame_end, accum_dev, ctr_dev);
    else
        hipLaunchKernelGGL(render_samples_kernel<false>, grid, block, 0, stream, S, cam, frame_begin, frame_end, accum_dev, ctr_dev);
}

// ---------------------------------------------------------------- opaque marks
// Sets the mark in the id word of every leaf-order triangle record (DTri::q0.w) from its triangle's entry of DScene::tri_info.
__global__ __launch_bounds__(256) void mark_opaque_triangles_kernel(DTri* __restrict__ tris, int n, const uint32_t* __restrict__ tri_info)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = __float_as_uint(tris[i].q0.w) >> 1;
    tris[i].q0.w = __uint_as_float((id << 1) | (tri_info[id] >> 31));
}

void launch_mark_opaque_triangles(DTri* tris_dev, int n, const uint32_t* tri_info_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(mark_opaque_triangles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tris_dev, n, tri_info_dev);
}

// ---------------------------------------------------------------- display resolve
// pathtracer.cu:588-599: accum / frames -> AgX -> RGBA8
__global__ __launch_bounds__(256) void resolve_display_kernel(const float* __restrict__ accum, int n_pixels, float frames, uint8_t* __restrict__ rgba)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    F3 c = f3(accum[3 * i], accum[3 * i + 1], accum[3 * i + 2]) / frames;
    c = agx_tonemap(c);
    uchar4 o = make_uchar4(to_byte(c.x), to_byte(c.y), to_byte(c.z), 255);
    reinterpret_cast<uchar4*>(rgba)[i] = o;
}

void launch_resolve_display(const float* accum_dev, int n_pixels, int n_frames, uint8_t* rgba_dev, hipStream_t stream)
{
    dim3 grid((n_pixels + 255) / 256), block(256);
    hipLaunchKernelGGL(resolve_display_kernel, grid, block, 0, stream, accum_dev, n_pixels, (float)n_frames, rgba_dev);
}

// ---------------------------------------------------------------- display post-process
// post_process.glsl:19-23 on the RGBA8 image the window would show: exposure, colour multiplier,
// saturation about the Rec.709 luma, gamma; bytes are read as x/255 (texture fetch of an 8-bit
// texture) and written as round(clamp(x) * 255) (framebuffer conversion).
__global__ __launch_bounds__(256) void post_process_kernel(const uint8_t* __restrict__ in, int n_pixels, float exposure, float mx, float my, float mz, float saturation, float gamma,
                                                           uint8_t* __restrict__ out)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    uchar4 p = reinterpret_cast<const uchar4*>(in)[i];
    F3 col = f3((float)p.x / 255.0f, (float)p.y / 255.0f, (float)p.z / 255.0f);
    col = col * exposure;
    col = col * f3(mx, my, mz);
    float lw = 0.2126f * col.x + 0.7152f * col.y + 0.0722f * col.z;
    col = f3(lw * (1.0f - saturation) + col.x * saturation, lw * (1.0f - saturation) + col.y * saturation, lw * (1.0f - saturation) + col.z * saturation);
    col = f3(pt_powf(col.x, gamma), pt_powf(col.y, gamma), pt_powf(col.z, gamma));
    auto unorm = [](float v) -> uint8_t { if (!(v > 0.f)) return 0; if (v >= 1.f) return 255; return (uint8_t)(int)(v * 255.0f + 0.5f); };
    reinterpret_cast<uchar4*>(out)[i] = make_uchar4(unorm(col.x), unorm(col.y), unorm(col.z), 255);
}
void launch_post_process(const uint8_t* in_dev, int n_pixels, float exposure, const float* mult3, float saturation, float gamma, uint8_t* out_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(post_process_kernel, dim3((n_pixels + 255) / 256), dim3(256), 0, stream, in_dev, n_pixels, exposure, mult3[0], mult3[1], mult3[2], saturation, gamma, out_dev);
}

// ---------------------------------------------------------------- environment search tables
// sampleEnvMap's binary searches (envmap_cuda.cu:28-57) compare against bilinearly filtered reads of
// the CDF texture at fixed texel coordinates. Those reads depend only on the map, so they are
// evaluated once here with the same sampler arithmetic; a search step then costs one load, not four.
__global__ __launch_bounds__(256) void env_tables_kernel(const float* __restrict__ cdf, int w, int h, float* __restrict__ row_table, float* __restrict__ col_table)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    int y = i / w, mid = i - y * w;
    col_table[i] = tex1((const PT_GLOBAL float*)cdf, w, h, (float)mid / (float)w, (float)y / (float)h);
    if (mid == 0) row_table[y] = tex1((const PT_GLOBAL float*)cdf, w, h, 1.f, (float)y / (float)h);
}
void launch_env_tables(const float* cdf_dev, int w, int h, float* row_table_dev, float* col_table_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(env_tables_kernel, dim3((w * h + 255) / 256), dim3(256), 0, stream, cdf_dev, w, h, row_table_dev, col_table_dev);
}

// The light sample of every texel (sample_envmap_texel), evaluated once: (direction, pdf) and (radiance, 0).
__global__ __launch_bounds__(256) void env_sample_table_kernel(DScene S, float4* __restrict__ table)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)S.env_w * S.env_h) return;
    int y = (int)(i / (size_t)S.env_w), x = (int)(i - (size_t)y * S.env_w);
    F3 color;
    float4 dir = sample_envmap_texel(S, x, y, color);
    table[2 * i] = dir;
    table[2 * i + 1] = make_float4(color.x, color.y, color.z, 0.f);
}
void launch_env_sample_table(const DScene& S, float4* table_dev, hipStream_t stream)
{
    size_t n = (size_t)S.env_w * S.env_h;
    hipLaunchKernelGGL(env_sample_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, table_dev);
}

// ---------------------------------------------------------------- environment search guides
// One workgroup per table (blockIdx 0: the row table, blockIdx 1 + y: row y of the column table).
// Bucket k covers the xi with (int)(xi * scale) == k; its range runs from the index the search ends at for
// the smallest such xi to the index for the largest, both taken a little wide.
PT_DEV int env_first_above(const float* __restrict__ table, int n, float xi)       // first index in [1, n-1) with xi < entry, else n - 1
{
    int lower = 1 < n - 1 ? 1 : n - 1, upper = n - 1;
    while (lower < upper) {
        int mid = (lower + upper) >> 1;
        if (xi < table[mid]) upper = mid; else lower = mid + 1;
    }
    return lower;
}
__global__ __launch_bounds__(256) void env_guides_kernel(const float* __restrict__ row_table, const float* __restrict__ col_table, int w, int h,
                                                         float row_scale, float col_scale, uint32_t* __restrict__ row_guide, uint32_t* __restrict__ col_guide)
{
    const bool rows = blockIdx.x == 0;
    const float* table = rows ? row_table : col_table + (size_t)(blockIdx.x - 1) * w;
    const int n = rows ? h : w, buckets = rows ? kEnvRowBuckets : kEnvColBuckets;
    const float scale = rows ? row_scale : col_scale;
    uint32_t* guide = rows ? row_guide : col_guide + (size_t)(blockIdx.x - 1) * kEnvColBuckets;
    for (int k = threadIdx.x; k < buckets; k += 256) {
        float lo_xi = ((float)k - 0.01f) / scale, hi_xi = ((float)k + 1.01f) / scale;
        int lo = env_first_above(table, n, lo_xi);
        int hi = k == buckets - 1 ? n - 1 : env_first_above(table, n, hi_xi);
        guide[k] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
}
// Compares the guided search with the reference's bisection at every value where either can change its answer:
// each table entry and its two neighbours in float order, and the floats around every bucket edge.
__global__ __launch_bounds__(256) void env_verify_kernel(DScene S, uint32_t* __restrict__ mismatches)
{
    const bool rows = blockIdx.x == 0;
    const float* table = rows ? S.env_row_table : S.env_col_table + (size_t)(blockIdx.x - 1) * S.env_w;
    const int n = rows ? S.env_h : S.env_w, buckets = rows ? kEnvRowBuckets : kEnvColBuckets;
    const float scale = rows ? S.env_row_scale : S.env_col_scale;
    const uint32_t* guide = rows ? S.env_row_guide : S.env_col_guide + (size_t)(blockIdx.x - 1) * kEnvColBuckets;
    uint32_t bad = 0, dummy = 0;
    auto check = [&](float xi) {
        if (!(xi >= 0.f)) return;
        int want = env_bisect<false>(table, n, xi, dummy);
        int got = env_guided(table, n, xi, guide[env_bucket(xi, scale, buckets)]);
        if (want != got) bad++;
    };
    auto around = [&](float v, int reach) {
        uint32_t b = __float_as_uint(v);
        for (int d = -reach; d <= reach; d++) {
            long long nb = (long long)b + d;
            if (nb >= 0 && nb < 0x7f800000ll) check(__uint_as_float((uint32_t)nb));
        }
    };
    for (int i = threadIdx.x; i < n; i += 256) around(table[i], 1);
    for (int k = threadIdx.x; k <= buckets; k += 256) around((float)k / scale, 8);
    if (threadIdx.x == 0) { check(0.f); check(S.env_sum); check(S.env_sum * 2.f); }
    if (bad) atomicAdd(mismatches, bad);
}
// The table over the whole range of xi. Bucket k serves xi with k <= xi * scale < k + 1; the products and quotients
// round (by up to 0.03 of a bucket at the top of the range), so each entry covers half a bucket more on either side.
__global__ __launch_bounds__(256) void env_texel_guide_kernel(DScene S, uint2* __restrict__ guide)
{
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k >= kEnvTexelBuckets) return;
    uint32_t dummy = 0;
    float lo_xi = ((float)k - 0.5f) / S.env_texel_scale, hi_xi = ((float)k + 1.5f) / S.env_texel_scale;
    lo_xi = lo_xi > 0.f ? lo_xi : 0.f;
    const bool last = k == kEnvTexelBuckets - 1;       // also takes every xi beyond the table
    int row_lo = env_bisect<false>(S.env_row_table, S.env_h, lo_xi, dummy);
    int row_hi = last ? S.env_h - 1 : env_bisect<false>(S.env_row_table, S.env_h, hi_xi, dummy);
    uint2 g = make_uint2(0xFFFFFFFFu, 0u);
    if (row_lo == row_hi && row_lo >= 2 && !last) {        // rows 0 and 1 involve the wrapped first entry: left to the general search
        const float* table = S.env_col_table + (size_t)row_lo * S.env_w;
        int lo = env_first_above(table, S.env_w, lo_xi), hi = env_first_above(table, S.env_w, hi_xi);
        g = make_uint2((uint32_t)row_lo, (uint32_t)lo | ((uint32_t)hi << 16));
    }
    guide[k] = g;
}
// The complete search (texel guide, else row and column guides) against the two bisections, at every value where
// either can change its answer: around each entry of both tables and around every bucket edge.
__global__ __launch_bounds__(256) void env_verify_texel_kernel(DScene S, uint32_t* __restrict__ mismatches)
{
    const size_t n_col = (size_t)S.env_w * S.env_h, n_row = (size_t)S.env_h, n_edge = (size_t)kEnvTexelBuckets + 1;
    uint32_t bad = 0, dummy = 0;
    auto check = [&](float xi) {
        if (!(xi >= 0.f)) return;
        int row = env_bisect<false>(S.env_row_table, S.env_h, xi, dummy);
        int y = (int)clamp_((float)row, 0.f, (float)(S.env_h - 1));
        int col = env_bisect<false>(S.env_col_table + (size_t)y * S.env_w, S.env_w, xi, dummy);
        int x = (int)clamp_((float)col, 0.f, (float)(S.env_w - 1));
        int gx, gy;
        env_search_texel<false>(S, xi, dummy, gx, gy);
        if (gx != x || gy != y) bad++;
    };
    auto around = [&](float v, int reach) {
        uint32_t b = __float_as_uint(v);
        for (int d = -reach; d <= reach; d++) {
            long long nb = (long long)b + d;
            if (nb >= 0 && nb < 0x7f800000ll) check(__uint_as_float((uint32_t)nb));
        }
    };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_col + n_row + n_edge; i += (size_t)gridDim.x * 256) {
        if (i < n_col) around(S.env_col_table[i], 1);
        else if (i < n_col + n_row) around(S.env_row_table[i - n_col], 1);
        else around((float)(i - n_col - n_row) / S.env_texel_scale, 8);
    }
    if (bad) atomicAdd(mismatches, bad);
}
// S must have env_fast and env_fast2 set and env_texel_guide pointing at guide_dev
void launch_env_texel_guide(const DScene& S, uint2* guide_dev, uint32_t* mismatches_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(env_texel_guide_kernel, dim3(kEnvTexelBuckets / 256), dim3(256), 0, stream, S, guide_dev);
    hipLaunchKernelGGL(env_verify_texel_kernel, dim3(2048), dim3(256), 0, stream, S, mismatches_dev);
}
void launch_env_guides(const DScene& S, uint32_t* row_guide_dev, uint32_t* col_guide_dev, uint32_t* mismatches_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(env_guides_kernel, dim3(1 + S.env_h), dim3(256), 0, stream, S.env_row_table, S.env_col_table, S.env_w, S.env_h,
                       S.env_row_scale, S.env_col_scale, row_guide_dev, col_guide_dev);
    hipLaunchKernelGGL(env_verify_kernel, dim3(1 + S.env_h), dim3(256), 0, stream, S, mismatches_dev);
}

// ---------------------------------------------------------------- numerics probe
// Evaluates the pinned transcendentals on the device so tests can compare bits with the host.
__global__ void detmath_probe_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y[i];
    float* o = out + (size_t)i * 12;
    o[0] = pt_sinf(a); o[1] = pt_cosf(a); o[2] = pt_expf(a); o[3] = pt_logf(a); o[4] = pt_log2f(a);
    o[5] = pt_powf(a, b); o[6] = pt_acosf(a); o[7] = pt_atan2f(a, b);
    o[8] = 1.0f / a; o[9] = __builtin_sqrtf(a); o[10] = a / b; o[11] = (float)pt_dm_pow((double)a, (double)b);
}
// Device evaluation of the BSDF alone, for lobe-by-lobe comparison with the CPU restatement: item i evaluates
// DisneyEval_2 (disney.cuh:234-347) for (V, N, L) and DisneySample_2 (disney.cuh:348-460) for (V, N) with the
// random stream of (pixel[i], frame), on material record i. 12 floats per item: eval f.xyz, eval pdf, sampled
// L.xyz, its f.xyz, its pdf, inside flag after the sample.
__global__ __launch_bounds__(64) void bsdf_probe_kernel(const pt_material* __restrict__ mats, const int* __restrict__ inside, const float* __restrict__ V3,
                                                         const float* __restrict__ N3, const float* __restrict__ L3, const uint32_t* __restrict__ pixel,
                                                         int frame, int n, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pt_material& M = mats[i];
    Mat m;
    m.base = ld3(M.baseColor); m.emis = ld3(M.emissive);
    m.rough = pt_fmaxf(M.roughness, 0.0001f);            // pathtracer.cu:427
    m.aniso = M.anisotropic; m.metal = M.metallic; m.subsurf = M.subsurface; m.sheen = M.sheen; m.coat = M.clearcoat;
    m.coat_rough = M.clearcoatRoughness; m.strans = M.specTrans; m.alpha = M.alpha; m.ior = M.ior;
    m.G = M.medium.G; m.sigA = ld3(M.medium.sigmaA); m.sigS = ld3(M.medium.sigmaS);
    const bool in = inside[i] != 0;
    const F3 V = ld3(V3 + 3 * (size_t)i), N = ld3(N3 + 3 * (size_t)i), L = ld3(L3 + 3 * (size_t)i);
    float* o = out + (size_t)i * 12;
    float pdf;
    F3 f = disney::eval(m, in, V, N, L, pdf);
    o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = pdf;
    Rng rng;
    rng_seed(rng, pixel[i], (uint32_t)frame, 0);
    bool in2 = in;
    F3 Ls;
    if (disney::sample_dir(m, in, V, N, Ls, rng, in2)) { f = f3(1.f, 1.f, 1.f); pdf = 1.f; }
    else {
        F3 T, B;
        onb(N, T, B);
        F3 Vrt = to_world(T, B, N, to_local(T, B, N, V));       // the local-frame round trip of disney.cuh:365,457
        f = disney::eval(m, in, Vrt, N, Ls, pdf);
    }
    o[4] = Ls.x; o[5] = Ls.y; o[6] = Ls.z; o[7] = f.x; o[8] = f.y; o[9] = f.z; o[10] = pdf; o[11] = in2 ? 1.f : 0.f;
}
void launch_bsdf_probe(const pt_material* mats_dev, const int* inside_dev, const float* V_dev, const float* N_dev, const float* L_dev, const uint32_t* pixel_dev,
                       int frame, int n, float* out_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(bsdf_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, mats_dev, inside_dev, V_dev, N_dev, L_dev, pixel_dev, frame, n, out_dev);
}
__global__ void unorm8_probe_kernel(float* __restrict__ out) { out[threadIdx.x] = unorm8(threadIdx.x); }
void launch_unorm8_probe(float* out_dev, hipStream_t stream) { hipLaunchKernelGGL(unorm8_probe_kernel, dim3(1), dim3(256), 0, stream, out_dev); }
void launch_detmath_probe(const float* x_dev, const float* y_dev, int n, float* out_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(detmath_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x_dev, y_dev, n, out_dev);
}

} // namespace ptamd
