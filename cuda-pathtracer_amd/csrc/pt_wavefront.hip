// Wavefront form of the render path: the per-sample state machine of pt_path.hpp cut into stage
// kernels over two dense queues of sample records that live in HBM.
//
//   generate : the first samples: camera rays appended to the path queue
//   plan     : one small workgroup turns the tails of the queues the stages appended to into the sizes of the
//              current ones and resets the per-iteration counters
//   extend   : closest hit for the pending ray of every record of both queues (camera, bounce and shadow-walk
//              rays alike): persistent waves over 4-wide quantized nodes, traversal stack in LDS
//   resolve  : consumes the hits of the path queue: shading normal and material values at the hit (the chain of
//              dependent gathers hit -> triangle -> material -> textures), written into a fresh record of the next
//              shadow queue; rays that left the scene are listed for the miss stage
//   vertex   : the rest of the vertex for those records (medium / surface event, alpha test, light-sample set-up);
//              a vertex that drew a light sample completes its record, a ray that passed through moves to the next
//              path queue
//   miss     : environment radiance for the listed rays; their samples are finished and the lanes start new ones
//   shadow   : consumes the hits of the shadow queue, one segment of the visibility walk each (the light-weight
//              kernel: alpha, metallic, transmission, ior and the medium of the surface it hit); a walk that goes
//              on moves to the next shadow queue, a finished one is listed for the bsdf stage
//   bsdf     : Disney evaluation for the light sample, direction sampling, roulette (the fat, divergent code) over
//              the listed records; the continuing path moves to the next path queue
//   drain    : once every sample has been started and few paths remain, each runs to its end in one
//              kernel instead of one launch sequence per remaining path vertex
//   accumulate: adds the per-sample results to the accumulator in frame order per pixel
//
// Every stage kernel therefore runs waves whose 64 lanes do the same kind of work on consecutive records: the
// reference's megakernel interleaves these kinds per thread (pathtracer.cu:405-481 with :267-339 nested in it), and a
// pool of slots that stay in place mixes them per wave.
//
// A finished sample writes its value to a staging array and the lane starts the next unstarted sample (from a
// 256-id chunk its workgroup holds), so the number of samples in flight stays at the pool size until the
// work runs out.  Which queue position renders which sample does not matter: a sample's value
// depends only on (pixel, frame) through its counter-based random stream, and the ordered
// accumulate pass makes the sum independent of scheduling.  Results are therefore bit-identical to
// the frame loop of the reference (pathtracer.cu:583-586) whatever the pool size or lane count.
#include <hip/hip_runtime.h>
#include <mutex>
#include "pt_path.hpp"
#include "pt_launch.h"

#pragma clang fp contract(off)

namespace ptamd {

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu;


// flags word: state[1:0] ch[3:2] inside[4] surface_scatter[5] sh_inside[6] nee_surface[7] nee_skip[8] v_inside[9] seg[15:10] lobes[19:16] bounce[31:20]
PT_DEV uint32_t pack_flags(const Path& P)
{
    return (uint32_t)P.state | ((uint32_t)(P.ch & 3) << 2) | ((uint32_t)P.inside << 4) | ((uint32_t)P.surface_scatter << 5) |
           ((uint32_t)P.sh_inside << 6) | ((uint32_t)P.nee_surface << 7) | ((uint32_t)P.nee_skip << 8) | ((uint32_t)P.v_inside << 9) |
           ((uint32_t)(P.seg & 63) << 10) | ((uint32_t)(P.lobes & 15) << 16) | ((uint32_t)P.bounce << 20);
}
PT_DEV void unpack_flags(Path& P, uint32_t f)
{
    P.state = (int)(f & 3u); P.ch = (int)((f >> 2) & 3u); P.inside = (f >> 4) & 1u; P.surface_scatter = (f >> 5) & 1u;
    P.sh_inside = (f >> 6) & 1u; P.nee_surface = (f >> 7) & 1u; P.nee_skip = (f >> 8) & 1u; P.v_inside = (f >> 9) & 1u;
    P.seg = (int)((f >> 10) & 63u); P.lobes = (int)((f >> 16) & 15u); P.bounce = (int)(f >> 20);
}

struct SampleId { int x, y, frame; uint32_t pixel; uint32_t stage_index; };

// Position of a lane-local pixel number in the image. Pixels are walked in 8x8 tiles when the image allows it,
// so that the 64 samples a wave takes together form one tile (coherent camera rays), else row by row; blocks of
// 64 consecutive pixels of that order are dealt to the lanes in turn.
PT_DEV uint32_t fast_div(uint32_t n, const FastDiv& f)
{
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
PT_DEV void lane_pixel(const WfParams& w, uint32_t q, int& x, int& y)
{
    uint32_t p = ((q >> 6) * (uint32_t)w.px_lanes + (uint32_t)w.px_lane) * 64u + (q & 63u);
    if (w.tiled) {
        uint32_t tile = p >> 6, l = p & 63u;
        uint32_t ty = fast_div(tile, w.div_tiles_x), tx = tile - ty * w.div_tiles_x.d;
        x = (int)(tx * 8 + (l & 7u));
        y = (int)(ty * 8 + (l >> 3));
    } else {
        uint32_t py = fast_div(p, w.div_W);
        x = (int)(p - py * w.div_W.d);
        y = (int)py;
    }
}
// sample id -> (pixel, frame): the lane's pixels frame after frame
PT_DEV SampleId decode_sample(const WfParams& w, uint32_t s)
{
    SampleId id;
    uint32_t fr = fast_div(s, w.div_pixels), q = s - fr * w.n_pixels;
    lane_pixel(w, q, id.x, id.y);
    id.frame = w.frame_begin + (int)fr;
    id.pixel = (uint32_t)(id.x + id.y * w.W);
    id.stage_index = s;
    return id;
}

// The triangle of a hit as the traversal kernel reports it: the id word of the record it tested (DTri::q0.w)
PT_DEV uint32_t hit_triangle(uint32_t word) { return word == PT_NO_TRI ? PT_NO_TRI : word >> 1; }

// ---- records in HBM: structures of 16-byte arrays indexed by queue position, so a wave's access is contiguous
PT_DEV void store_core(float4* c0, float4* c1, float4* c2, uint32_t i, const Path& P, uint32_t sample)
{
    c0[i] = make_float4(P.o.x, P.o.y, P.o.z, P.thr);
    c1[i] = make_float4(P.d.x, P.d.y, P.d.z, P.pdf);
    c2[i] = make_float4(P.result, __uint_as_float(pack_flags(P)), __uint_as_float(P.rng.draws), __uint_as_float(sample));
}
PT_DEV uint32_t unpack_core(const float4& a, const float4& b, const float4& c, const WfParams& w, Path& P, SampleId& id)
{
    P.o = f3(a.x, a.y, a.z); P.thr = a.w;
    P.d = f3(b.x, b.y, b.z); P.pdf = b.w;
    P.result = c.x;
    unpack_flags(P, __float_as_uint(c.y));
    uint32_t sample = __float_as_uint(c.w);
    id = decode_sample(w, sample);
    rng_seed(P.rng, id.pixel, (uint32_t)id.frame, __float_as_uint(c.z));
    return sample;
}
PT_DEV uint32_t load_core(const float4* c0, const float4* c1, const float4* c2, uint32_t i, const WfParams& w, Path& P, SampleId& id)
{
    float4 a = c0[i], b = c1[i], c = c2[i];
    return unpack_core(a, b, c, w, P, id);
}
PT_DEV void store_path(const WfPathQueue& q, uint32_t i, const Path& P, uint32_t sample) { store_core(q.c0, q.c1, q.c2, i, P, sample); }
// a record of the path queue; the fields only a light sample sets start out cleared
PT_DEV uint32_t load_path(const WfPathQueue& q, uint32_t i, const WfParams& w, Path& P, SampleId& id)
{
    P.so = P.sd = P.light = P.env_color = f3(0.f, 0.f, 0.f); P.env_pdf = 0.f;
    return load_core(q.c0, q.c1, q.c2, i, w, P, id);
}
// A whole record of the shadow queue. The record carries the values of the vertex that differ from hit to hit (what
// the texture maps and the shading normal give: e0, e4, the metallic value, and e1 where the material has an emission
// map) and the index of its material (so.w); the material's other parameters, which no map changes (resolve_hit), are
// read from its record, which stays in the caches. Returns whether the material has an emission map.
PT_DEV bool material_constants(const DScene& S, int mat_index, Mat& m)
{
    const float4* mq = reinterpret_cast<const float4*>(S.mats + mat_index);
    const float4 m0 = mq[0], m1 = mq[1], m2 = mq[2], m3 = mq[3], m4 = mq[4], m6 = mq[6];
    m.emis = f3(m0.w, m1.x, m1.y);
    m.aniso = m1.w; m.subsurf = m2.y; m.sheen = m2.z; m.coat = m2.w;
    m.coat_rough = m3.x; m.strans = m3.y; m.ior = m3.w; m.G = m4.x;
    return __float_as_int(m6.z) != 0;
}
// (everything but the path state, which the caller has loaded: P.pdf holds the word that travels in its place)
PT_DEV void load_vertex(const WfShadowQueue& q, uint32_t i, const DScene& S, Path& P)
{
    float4 so = q.so[i], sd = q.sd[i], l1 = q.d1[i];
    float4 a = q.e0[i], e = q.e4[i];
    P.so = f3(so.x, so.y, so.z); P.sd = f3(sd.x, sd.y, sd.z);
    // the transmittance of the walk so far is (1, 1, 1) unless a segment crossed a medium (marked in sd.w)
    P.light = f3(1.f, 1.f, 1.f);
    if (__float_as_int(sd.w) == 2) { const float4 l0 = q.d0[i]; P.light = f3(l0.x, l0.y, l0.z); }
    P.env_color = f3(l1.x, l1.y, l1.z); P.env_pdf = l1.w;
    Mat& m = P.vm;
    m.base = f3(a.x, a.y, a.z); m.rough = a.w;
    m.metal = P.pdf;            // (a vertex record keeps the metallic value where the last bounce's pdf was: that pdf is dead here, after_light sets the next)
    if (material_constants(S, __float_as_int(so.w), m)) { const float4 b = q.e1[i]; m.emis = f3(b.x, b.y, b.z); }
    P.vn = f3(e.x, e.y, e.z); m.alpha = e.w;
    m.sigA = f3(0.f, 0.f, 0.f); m.sigS = f3(0.f, 0.f, 0.f);     // not used after the vertex
}
PT_DEV uint32_t load_shadow(const WfShadowQueue& q, uint32_t i, const WfParams& w, const DScene& S, Path& P, SampleId& id)
{
    uint32_t sample = load_core(q.c0, q.c1, q.c2, i, w, P, id);
    load_vertex(q, i, S, P);
    return sample;
}

// Sample ids are handed out per workgroup from a chunk of kSampleChunk consecutive ids that the
// workgroup owns (four 8x8 tiles of one frame when the image is tiled). The stage kernels are launched with a
// fixed number of persistent workgroups that take 256-record segments of their queue in turn; workgroup b of any
// stage launch owns chunk b, whose cursor lives in global memory between launches. A new chunk costs one atomic on
// the global counter, so a launch over millions of records issues a few thousand atomics instead of one per
// wave (a single counter word serves only about 90 returning atomics per microsecond chip-wide). Ids left in
// chunks when the global counter has run out are started by wf_flush_leftover_kernel.
constexpr uint32_t kSampleChunk = 256;

// A finished sample: publish its value (pathtracer.cu:540,580: channel value * 3, clamped below at 0)
// and start the next unstarted sample in this lane, if any. Must be called by every thread of the
// workgroup (it synchronises).
template <bool COUNT>
PT_DEV void finish_and_regenerate(const WfParams& w, const DCamera& cam, Path& P, uint32_t& sample, SampleId& id, bool finished, uint32_t& nsamples)
{
    __shared__ uint32_t s_need[kBlock / 64];
    __shared__ uint32_t s_plan[3];      // old cursor, ids available at the old cursor, base of the new chunk
    __shared__ uint32_t s_chunk[2];     // cursor and end of the workgroup's chunk
    const int wid = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    uint32_t* cw = w.wg_chunk + 2 * (size_t)blockIdx.x;
    if (threadIdx.x == 0) { s_chunk[0] = cw[0]; s_chunk[1] = cw[1]; }
    // loops only while some fresh sample is born finished (its channel draw was exactly 1.0: 3e-8)
    for (;;) {
        if (finished) {
            float v = P.result * 3.f;
            w.staging[id.stage_index] = v > 0.f ? v : 0.f;
        }
        unsigned long long mask = __ballot(finished);
        if (lane == 0) s_need[wid] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t total = 0, woff = 0;
#pragma unroll
        for (int j = 0; j < kBlock / 64; j++) { uint32_t nj = s_need[j]; if (j < wid) woff += nj; total += nj; }
        if (total == 0) break;          // uniform over the workgroup
        if (threadIdx.x == 0) {
            uint32_t next = s_chunk[0], end = s_chunk[1];
            uint32_t avail = end - next;
            s_plan[0] = next;
            if (total <= avail) {
                s_plan[1] = total; s_plan[2] = 0;
                next += total;
            } else if (next >= w.n_samples) {
                // every sample id has been handed out (the cursor was parked at n_samples): the lanes retire
                s_plan[1] = 0; s_plan[2] = w.n_samples;
            } else {
                uint32_t c = atomicAdd(w.counters + WF_CTR_NEXT_SAMPLE, kSampleChunk);
                s_plan[1] = avail; s_plan[2] = c;
                next = c + (total - avail); end = c + kSampleChunk;
                if (c >= w.n_samples) { next = w.n_samples; end = w.n_samples; }
            }
            s_chunk[0] = next; s_chunk[1] = end;
        }
        __syncthreads();
        if (finished) {
            uint32_t r = woff + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            uint32_t next = r < s_plan[1] ? s_plan[0] + r : s_plan[2] + (r - s_plan[1]);
            if (next < w.n_samples) {
                sample = next;
                id = decode_sample(w, sample);
                begin_sample(P, cam, id.x, id.y, id.frame);
                if (COUNT) nsamples++;
                finished = P.state == ST_DONE;
            } else {
                P.state = ST_DONE;      // nothing left to start: the lane emits no record
                sample = kNone;
                finished = false;
            }
        }
        __syncthreads();                // s_need and s_plan are reused by the next round
    }
    if (threadIdx.x == 0) { cw[0] = s_chunk[0]; cw[1] = s_chunk[1]; }
}

// Entry l of stripe r of a striped queue or list (see kStripes).
PT_DEV uint32_t stripe_pos(uint32_t r, uint32_t l) { return ((((l >> 8) << kStripeShift) | r) << 8) | (l & 255u); }
// Entries in use of block B of a striped array whose stripes hold counts[0..kStripes) entries.
PT_DEV uint32_t block_fill(uint32_t B, const uint32_t* counts)
{
    const uint32_t n = counts[B & (uint32_t)(kStripes - 1)], base = (B >> kStripeShift) << 8;
    return n > base ? (n - base < 256u ? n - base : 256u) : 0u;
}

// Queue positions for the lanes of a workgroup that append to a queue: the lanes for which `want` holds get
// consecutive entries of stripe r, in thread order, from one atomic on the stripe's tail word (tail[k] names the
// family). Every thread of the workgroup must call it (it synchronises); the others get kNone. Up to four queues
// at once share the barriers.
struct Appends { bool want[4]; uint32_t* tail[4]; uint32_t pos[4]; };
template <int N>
PT_DEV void reserve(Appends& a, uint32_t r)
{
    __shared__ uint32_t s_cnt[4][kBlock / 64];
    __shared__ uint32_t s_base[4];
    const int wid = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    unsigned long long m[4];
#pragma unroll
    for (int k = 0; k < N; k++) {
        m[k] = __ballot(a.want[k]);
        if (lane == 0) s_cnt[k][wid] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        if (threadIdx.x == (unsigned)(k * 64)) {        // one lane of a different wave per queue: the atomics go out together
            uint32_t total = 0;
#pragma unroll
            for (int j = 0; j < kBlock / 64; j++) total += s_cnt[k][j];
            s_base[k] = total ? atomicAdd(a.tail[k] + r * (uint32_t)kStripeWords, total) : 0u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        uint32_t woff = 0;
#pragma unroll
        for (int j = 0; j < kBlock / 64; j++) if (j < wid) woff += s_cnt[k][j];
        a.pos[k] = a.want[k] ? stripe_pos(r, s_base[k] + woff + (uint32_t)__popcll(m[k] & ((1ull << lane) - 1ull))) : kNone;
    }
    __syncthreads();                    // the counts are reused by the caller's next block
}

// How a workgroup of a stage launch takes its input. The stripes' block cursors (the family `cursors`) are handed
// out by atomics; a workgroup takes the blocks of one stripe, starting with the stripe of its own number, so the
// workgroups of a launch spread their atomics evenly over the stripes' words, and moves to the next stripe that has
// blocks left when its own runs out (it then looks at all the cursors at once: plain loads, one latency).
//   first_block: every thread calls it once. Copies the stripes' record counts (count_words[r * count_stride]) to
//     counts[] in LDS and picks the first stripe. A workgroup that starts when everything is taken (the later
//     workgroups of a launch, which wait for a place on the chip until the earlier ones leave) finds that out here
//     with one round of loads and no atomic.
//   next_block: every thread calls it per block; returns the block or kNone. The atomic for the block after this one
//     goes out at once and is waited for only at the next call, behind the work on this block.
struct BlockCursor { uint32_t r, k; };      // (thread 0's) stripe in hand, and what its cursor returned for the next block
PT_DEV uint32_t blocks_of(uint32_t records) { return (records + 255u) >> 8; }
PT_DEV uint32_t stripe_with_blocks_after(uint32_t r, const uint32_t* cursors, const uint32_t* counts, bool including_r)
{
    uint32_t cur[kStripes];
#pragma unroll
    for (int j = 0; j < kStripes; j++) cur[j] = __hip_atomic_load(cursors + j * kStripeWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t next = kNone;
#pragma unroll
    for (int j = kStripes - 1; j >= 0; j--) {           // the nearest one wins
        if (j == 0 && !including_r) break;
        const uint32_t q = (r + (uint32_t)j) & (uint32_t)(kStripes - 1);
        uint32_t cq = 0;
#pragma unroll
        for (int t = 0; t < kStripes; t++) if ((uint32_t)t == q) cq = cur[t];
        if (cq < blocks_of(counts[q])) next = q;
    }
    return next;
}
PT_DEV BlockCursor first_block(uint32_t* cursors, const uint32_t* count_words, uint32_t count_stride, uint32_t* counts)
{
    if (threadIdx.x < kStripes) counts[threadIdx.x] = count_words[threadIdx.x * count_stride];
    __syncthreads();
    BlockCursor c; c.r = kNone; c.k = 0;
    if (threadIdx.x == 0) {
        c.r = stripe_with_blocks_after(blockIdx.x & (uint32_t)(kStripes - 1), cursors, counts, true);
        if (c.r != kNone) c.k = atomicAdd(cursors + c.r * (uint32_t)kStripeWords, 1u);
    }
    return c;
}
PT_DEV uint32_t next_block(uint32_t* cursors, const uint32_t* counts, BlockCursor& c)
{
    __shared__ uint32_t s_blk;
    if (threadIdx.x == 0) {
        uint32_t blk = kNone;
        while (c.r != kNone) {
            if (c.k < blocks_of(counts[c.r])) {
                blk = (c.k << kStripeShift) | c.r;
                c.k = atomicAdd(cursors + c.r * (uint32_t)kStripeWords, 1u);
                break;
            }
            c.r = stripe_with_blocks_after(c.r, cursors, counts, false);
            if (c.r != kNone) c.k = atomicAdd(cursors + c.r * (uint32_t)kStripeWords, 1u);
        }
        s_blk = blk;
    }
    __syncthreads();
    const uint32_t blk = s_blk;
    __syncthreads();
    return blk;
}

template <bool COUNT>
PT_DEV void flush_tally(DCounters* ctr, const Tally& tl, uint32_t nsamples)
{
    if (!COUNT) return;
    // one atomic per counter per wave
    auto wave_sum = [](uint32_t v) { for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_down((int)v, o, 64); return v; };
    uint32_t r = wave_sum(tl.rays), b = wave_sum(tl.box), t = wave_sum(tl.tri), v = wave_sum(tl.verts), e = wave_sum(tl.envs);
    uint32_t x = wave_sum(tl.taps), y = wave_sum(tl.taps_r), s = wave_sum(nsamples), p = wave_sum(tl.path_rays);
    if ((threadIdx.x & 63u) == 0) {
        if (r) atomicAdd(&ctr->rays, (unsigned long long)r);
        if (p) atomicAdd(&ctr->path_rays, (unsigned long long)p);
        if (b) atomicAdd(&ctr->box_tests, (unsigned long long)b);
        if (t) atomicAdd(&ctr->tri_tests, (unsigned long long)t);
        if (v) atomicAdd(&ctr->shade_vertices, (unsigned long long)v);
        if (e) atomicAdd(&ctr->env_samples, (unsigned long long)e);
        if (x) atomicAdd(&ctr->tex_taps, (unsigned long long)x);
        if (y) atomicAdd(&ctr->tex_taps_r, (unsigned long long)y);
        if (s) atomicAdd(&ctr->samples, (unsigned long long)s);
    }
}

} // namespace

// ---------------------------------------------------------------- generate
// The first samples of a chunk of frames: ids [0, n_start) as camera rays at the front of the path queue the
// first iteration reads (passed here as the "next" one).
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void wf_generate_kernel(WfParams w, DCamera cam, uint32_t n_start, DCounters* ctr)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    Path P;
    P.state = ST_DONE;
    SampleId id = decode_sample(w, 0);
    uint32_t nsamples = 0;
    bool have = s < n_start;
    if (have) {
        id = decode_sample(w, s);
        P.so = P.sd = P.light = P.env_color = f3(0.f, 0.f, 0.f); P.env_pdf = 0.f;
        P.sh_inside = P.nee_surface = P.nee_skip = P.v_inside = false; P.seg = 0;
        begin_sample(P, cam, id.x, id.y, id.frame);
        if (COUNT) nsamples++;
        if (P.state == ST_DONE) { w.staging[id.stage_index] = 0.f; have = false; }      // born finished: its channel draw was exactly 1.0
    }
    Appends a;
    a.want[0] = have; a.tail[0] = w.counters + WF_CTR_TAIL_P;
    reserve<1>(a, blockIdx.x & (uint32_t)(kStripes - 1));     // 256 ids to a stripe, in turn
    if (have) store_path(w.Pn, a.pos[0], P, s);
    if (blockIdx.x == 0 && threadIdx.x == 0) w.counters[WF_CTR_NEXT_SAMPLE] = n_start;
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
    flush_tally<COUNT>(ctr, tl, nsamples);
}

// ---------------------------------------------------------------- plan
// Runs before every extend launch (one small workgroup): what the stages of the last iteration appended becomes
// the current queues, and the per-iteration counters start over.
// no_walks: the scene draws no light samples (sun/sky mode), so every record of the shadow queue went through the bsdf
// stage in the iteration that wrote it and none waits for a ray: the queue is dropped instead of being scanned.
__global__ __launch_bounds__(kBlock) void wf_plan_kernel(uint32_t* __restrict__ counters, const uint32_t* __restrict__ wg_chunk, int stage_blocks, uint32_t n_samples, int no_walks)
{
    __shared__ uint32_t s_left[kBlock / 64];
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t lsum = 0;          // sample ids still waiting in the workgroups' chunks
    for (int b = tid; b < stage_blocks; b += kBlock) {
        uint32_t cur = wg_chunk[2 * b], end = wg_chunk[2 * b + 1];
        cur = cur < n_samples ? cur : n_samples; end = end < n_samples ? end : n_samples;
        lsum += end > cur ? end - cur : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) lsum += (uint32_t)__shfl_down((int)lsum, o, 64);
    if (lane == 0) s_left[wid] = lsum;
    __syncthreads();
    // per stripe: what was appended becomes the current count, the tails and block cursors start over
    uint32_t bP = 0, bS = 0;
    if (tid < kStripes) {
        uint32_t* f = counters + tid * kStripeWords;
        const uint32_t nP = f[WF_CTR_TAIL_P], nS = no_walks ? 0u : f[WF_CTR_TAIL_S];
        counters[WF_CTR_N_P + tid] = nP; counters[WF_CTR_N_S + tid] = nS;
        f[WF_CTR_TAIL_P] = 0; f[WF_CTR_TAIL_S] = 0; f[WF_CTR_TAIL_B] = 0; f[WF_CTR_TAIL_M] = 0;
        f[WF_CTR_SEG_PATH] = 0; f[WF_CTR_SEG_SHADOW] = 0; f[WF_CTR_SEG_BSDF] = 0; f[WF_CTR_SEG_MISS] = 0; f[WF_CTR_SEG_VERTEX] = 0;
        bP = (nP + 255u) >> 8; bS = (nS + 255u) >> 8;
    }
    // the queues extend as far as their longest stripe (the first wave holds every stripe)
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t oP = (uint32_t)__shfl_down((int)bP, o, 64), oS = (uint32_t)__shfl_down((int)bS, o, 64);
        bP = oP > bP ? oP : bP; bS = oS > bS ? oS : bS;
    }
    if (tid == 0) {
        uint32_t leftover = 0;
        for (int j = 0; j < kBlock / 64; j++) leftover += s_left[j];
        counters[WF_CTR_RAY_NEXT] = 0;
        counters[WF_CTR_RAY_P_CHUNKS] = bP << kStripeShift;
        counters[WF_CTR_RAY_CHUNKS] = (bP + bS) << kStripeShift;
        counters[WF_CTR_LEFTOVER] = leftover;
    }
}

// ---------------------------------------------------------------- extend
// Persistent waves over the rays of both queues. Traversal lengths vary by more than 10x between rays, so a
// wave that kept its 64 rays until the last one finished would run mostly empty (measured: about 6
// of 64 lanes busy). Here a lane whose ray is finished takes the next untraced ray (from a block of
// 256 queue positions the wave took with one atomic: path-queue blocks first, then shadow-queue blocks) as soon as
// w.refill_lanes lanes are idle, and leaf work is
// postponed until w.leaf_vote lanes wait at a leaf (or nobody can descend), so each pass of the loop
// does one kind of work for most lanes.
//   COUNT: walks the 64-byte fp32 binary nodes in the reference's order and tallies box and triangle
//          tests, which then equal the oracle's counts.
//   WIDE:  walks the 4-wide quantized nodes (closest hits do not depend on the visiting order).
//   both:  the 4-wide walk with its own work tallied: wide nodes visited (in the box counter) and triangle tests.
//   neither: the binary quantized nodes, for trees too deep for the 4-wide stack.
#ifndef PT_SKIP_ROOT_TEST
#define PT_SKIP_ROOT_TEST 1
#endif
#ifndef PT_TAIL_FACTOR
#define PT_TAIL_FACTOR 2u
#endif
#ifndef PT_STAMPS
#define PT_STAMPS 0      // diagnostic build: pass statistics of the extend kernel
#endif
#ifndef PT_TWO_RAYS
#define PT_TWO_RAYS 0    // experiment: the 4-wide walk carries two rays per lane (see the loop top of wf_extend_kernel). Same hits; fuller passes
                         // (55 instead of 44 descending lanes) but each 16 % longer: no gain unless the SIMDs are short of waves
#endif
#ifndef PT_EXTEND_LDS2
#define PT_EXTEND_LDS2 8
#endif
constexpr int kRayChunk = 256;
#ifndef PT_EXTEND_LDS
#define PT_EXTEND_LDS 14
#endif
// Stack entries per lane in LDS: 14 x 256 lanes x 8 B = 28 KB per workgroup. With 16 the workgroup's other shared words
// made it 32 KB + 128 B: four workgroups per CU instead of the five its registers allow, and no room for a workgroup
// of the other lane's traversal launch beside them (16 -> 15, 12 or 8: +3 % on the bench, the same for all three).
constexpr int kExtendLds = PT_EXTEND_LDS;
constexpr int kExtendLds2 = PT_EXTEND_LDS2;     // ... per ray when a lane carries two

template <bool COUNT, bool WIDE>
__global__ __launch_bounds__(kBlock) void wf_extend_kernel(WfParams w, DScene S, DCounters* ctr)
{
    constexpr bool TWO = PT_TWO_RAYS && WIDE;
    constexpr int LDSE = TWO ? kExtendLds2 : kExtendLds;        // stack entries per ray in LDS
    __shared__ uint2 s_stack[(TWO ? 2 : 1) * LDSE * kBlock];
    uint2 spill[(TWO ? 2 : 1) * kSpillDepth];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    int cur = PT_TRAV_EMPTY, sp = 0;
    // TWO: the lane's second ray, set aside while the first one is worked on (pcur == PT_TRAV_EMPTY: none), and the
    // half of the lane's stack space the ray in hand uses
    int pcur = PT_TRAV_EMPTY, psp = 0, bank = 0;
    uint32_t pslot = 0;
    F3 po = f3(0.f, 0.f, 0.f), pd = po, pinv = po;
    Hit pbest; pbest.t = 1e10f; pbest.tri = PT_NO_TRI; pbest.u = 0.f; pbest.v = 0.f;
    uint32_t slot = 0;
    F3 o = f3(0.f, 0.f, 0.f), d = o, inv = o;
    // ray in grid coordinates for the binary compact nodes
    F3 oq = o, invq = o, noq = o;
    constexpr bool QUANT = WIDE || !COUNT;
    Hit best; best.t = 1e10f; best.tri = PT_NO_TRI; best.u = 0.f; best.v = 0.f;
    bool has_ray = false;
    bool more = true;
    uint32_t chunk_next = 0, chunk_end = 0;      // wave-uniform
    bool chunk_shadow = false;                   // wave-uniform: the block in hand belongs to the shadow queue
    const uint32_t n_chunks = w.counters[WF_CTR_RAY_CHUNKS], n_pchunks = w.counters[WF_CTR_RAY_P_CHUNKS];
    __shared__ uint32_t s_count[2 * kStripes];      // records per stripe of the path queue, then of the shadow queue
    if (tid < 2 * kStripes) s_count[tid] = w.counters[WF_CTR_N_P + tid];
    __syncthreads();
    const uint32_t n_small = n_chunks < PT_TAIL_FACTOR * gridDim.x ? n_chunks : PT_TAIL_FACTOR * gridDim.x, n_big = n_chunks - n_small;
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
#if PT_STAMPS
    unsigned long long st_iters = 0, st_lanes = 0, st_leafphases = 0, st_refills = 0, st_leaflanes = 0, st_idle = 0, st_tail = 0, st_tail_idle = 0, st_tail_desc = 0, st_tail_t0 = 0, st_t0 = __builtin_amdgcn_s_memtime();
    uint32_t st_age = 0; unsigned long long st_h64 = 0, st_h128 = 0, st_h256 = 0, st_h512 = 0, st_max = 0;
#endif

    auto push = [&](int ref, float tn) {
        uint2 e = make_uint2((uint32_t)ref, __float_as_uint(tn));
        if (sp < LDSE) s_stack[(bank * LDSE + sp) * kBlock + tid] = e;
        else if (sp - LDSE < kSpillDepth) spill[bank * kSpillDepth + sp - LDSE] = e;
        sp++;
    };
    auto pop_valid = [&]() -> int {
        while (sp > 0) {
            sp--;
            uint2 e;
            if (sp < LDSE) e = s_stack[(bank * LDSE + sp) * kBlock + tid];
            else {
                e = spill[bank * kSpillDepth + ((sp - LDSE) < kSpillDepth ? (sp - LDSE) : (kSpillDepth - 1))];
                asm volatile("" : "+v"(e.x), "+v"(e.y));    // keep the LDS and the private load apart
            }
            if (__uint_as_float(e.y) < (WIDE ? wide_cull(best.t) : best.t)) return (int)e.x;
        }
        return PT_TRAV_EMPTY;
    };
    auto retire = [&]() {      // the ray of this lane is finished: publish its hit
        ((slot >> 31) ? w.hitS : w.hitP)[slot & 0x7FFFFFFFu] = make_float4(best.t, __uint_as_float(best.tri), best.u, best.v);
        has_ray = false;
#if PT_STAMPS
        st_h64 += st_age >= 64; st_h128 += st_age >= 128; st_h256 += st_age >= 256; st_h512 += st_age >= 512; st_max = st_age > st_max ? st_age : st_max;
#if PT_STAMPS > 1
        if (st_age >= 600) { unsigned long long* dbg = reinterpret_cast<unsigned long long*>(w.counters + WF_CTR_WORDS); dbg[4] = __float_as_uint(o.x); dbg[5] = __float_as_uint(o.y); dbg[6] = __float_as_uint(o.z); dbg[7] = __float_as_uint(d.x); dbg[8] = __float_as_uint(d.y); dbg[9] = __float_as_uint(d.z); dbg[10] = st_age; dbg[11] = __float_as_uint(best.t); }
#endif
#endif
    };

    for (;;) {
        // ---- decide what this pass does. A ray is descending (cur is an interior node) or waiting at a leaf.
        // TWO: a lane carries up to two rays and works on the one that fits the pass: while one waits for the wave's
        // next leaf pass the other descends, so fewer lanes sit out a pass (a ray's hit does not depend on when its
        // steps are taken). The ray set aside keeps its registers; the two change places only when the lane has
        // nothing to do for the one in hand.
        bool leaf_on = false;
        unsigned long long none_mask = 0ull;
        if constexpr (TWO) {
            const bool p_none = pcur == PT_TRAV_EMPTY, p_node = pcur >= 0, p_leaf = !p_none && pcur < 0;
            const bool a_node = has_ray && cur >= 0, a_leaf = has_ray && cur < 0;
            const bool any_node = a_node || p_node, any_leaf = a_leaf || p_leaf;
            none_mask = __ballot(!has_ray && p_none);
            const int n_act = 64 - __popcll(none_mask);
            const int need = (n_act >> 1) < w.leaf_vote ? ((n_act >> 1) > 1 ? (n_act >> 1) : 1) : w.leaf_vote;
            // a leaf pass when enough lanes have only leaf work left, or nobody can descend
            leaf_on = __popcll(__ballot(any_leaf && !any_node)) >= need || __ballot(any_node) == 0ull;
            const bool change = leaf_on ? (!a_leaf && p_leaf) : (!a_node && p_node);
            if (change) {
                // (one exchange instruction per register: written out as three moves each, the compiler also copied the
                // whole state at both ends of the loop)
#define PT_SWAP(a, b) asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b))
                PT_SWAP(cur, pcur); PT_SWAP(sp, psp); PT_SWAP(slot, pslot);
                PT_SWAP(o.x, po.x); PT_SWAP(o.y, po.y); PT_SWAP(o.z, po.z);
                PT_SWAP(d.x, pd.x); PT_SWAP(d.y, pd.y); PT_SWAP(d.z, pd.z);
                PT_SWAP(inv.x, pinv.x); PT_SWAP(inv.y, pinv.y); PT_SWAP(inv.z, pinv.z);
                PT_SWAP(best.t, pbest.t); PT_SWAP(best.tri, pbest.tri); PT_SWAP(best.u, pbest.u); PT_SWAP(best.v, pbest.v);
#undef PT_SWAP
                bank ^= 1;
                has_ray = cur != PT_TRAV_EMPTY;
            }
        }
        const bool idle = TWO ? (!has_ray || pcur == PT_TRAV_EMPTY) : !has_ray;      // the lane can take a new ray
        const bool descending = has_ray && cur >= 0;
        const bool at_leaf = has_ray && cur < 0;
        const unsigned long long idle_mask = __ballot(idle);
        const unsigned long long desc_mask = __ballot(descending);
        const unsigned long long leaf_mask = __ballot(at_leaf);
        const int n_idle = __popcll(idle_mask);
        if (!more && (TWO ? none_mask == ~0ull : n_idle == 64)) break;
        const bool do_refill = more && n_idle >= w.refill_lanes;
        // the vote is taken among the lanes that carry a ray: a wave that is running out of rays must not wait for
        // sixteen of them at leaves
        const int n_active = 64 - n_idle;
        const int vote_need = (n_active >> 1) < w.leaf_vote ? ((n_active >> 1) > 1 ? (n_active >> 1) : 1) : w.leaf_vote;
        const bool leaf_now = TWO ? (at_leaf && leaf_on) : (at_leaf && (__popcll(leaf_mask) >= vote_need || desc_mask == 0ull));
#if PT_STAMPS
        st_age++; st_iters++; st_lanes += __popcll(desc_mask); st_leafphases += __any(leaf_now) ? 1 : 0; st_refills += do_refill ? 1 : 0;
        st_leaflanes += __popcll(__ballot(leaf_now)); st_idle += __popcll(__ballot(!descending && !leaf_now));
        if (!more) { if (!st_tail) st_tail_t0 = __builtin_amdgcn_s_memtime(); st_tail++; st_tail_idle += n_idle; st_tail_desc += __popcll(desc_mask); }
#endif
        // ---- issue every load of the pass before anything waits: a pass then costs one memory latency,
        // not one per kind of work
        bool got = false;
        uint32_t idx = 0;
        float4 ro4 = make_float4(0.f, 0.f, 0.f, 0.f), rd4 = ro4;
        if (do_refill) {
            // rays are handed out in blocks of kRayChunk consecutive queue positions, taken with one atomic each
            // (a static deal of blocks to waves was measured slower)
            if (chunk_next >= chunk_end) {
                uint32_t li = 0;
                if (lane == 0) li = atomicAdd(w.counters + WF_CTR_RAY_NEXT, 1u);
                li = (uint32_t)__builtin_amdgcn_readfirstlane((int)li);
                if (li >= n_big + 4u * n_small) { more = false; chunk_end = chunk_next; }
                else {
                    // the last blocks go out in quarters, so that no wave ends the launch alone with 256 rays
                    uint32_t blk = li, off = 0, len = kRayChunk;
                    if (li >= n_big) { const uint32_t j = li - n_big; blk = n_big + (j >> 2); off = (j & 3u) * (uint32_t)(kRayChunk / 4); len = kRayChunk / 4; }
                    chunk_shadow = blk >= n_pchunks;
                    const uint32_t B = chunk_shadow ? blk - n_pchunks : blk;
                    // (the block may be partly or wholly empty: its stripe is shorter than the longest one)
                    const uint32_t fill = block_fill(B, s_count + (chunk_shadow ? kStripes : 0));
                    chunk_next = B * (uint32_t)kRayChunk + off;
                    chunk_end = chunk_next + (fill > off ? (fill - off < len ? fill - off : len) : 0u);
                }
            }
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t rank = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
            got = idle && rank < avail;
            idx = chunk_next + rank;
            chunk_next += (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
            // a path record's ray is its (o, d); a shadow record's the pending segment of its visibility walk
            if (got) { rd4 = (chunk_shadow ? w.Sc.sd : w.Pc.c1)[idx]; ro4 = (chunk_shadow ? w.Sc.so : w.Pc.c0)[idx]; }
        }
        float4 n0, n1, n2, n3;          // (read only by the lanes that load them: no defined value elsewhere, so no copies)
        if (descending) {
            if (WIDE) {
                const float4* np = reinterpret_cast<const float4*>(S.nodes4 + cur);
                n0 = np[0]; n1 = np[1]; n2 = np[2]; n3 = np[3];
            } else if (QUANT) {
                const float4* np = reinterpret_cast<const float4*>(S.qnodes + cur);
                n0 = np[0]; n1 = np[1];
            } else {
                const float4* np = reinterpret_cast<const float4*>(S.nodes + cur);
                n0 = np[0]; n1 = np[1]; n2 = np[2]; n3 = np[3];
            }
        }
        float4 q0, q1, q2, p0, p1, p2;
        const DTri* tp = S.tris;
        int leaf_count = 0;
        if (leaf_now) {
            if (WIDE) { uint32_t enc = ~(uint32_t)cur; tp = S.tris + (enc & 0x3FFFFFFu); leaf_count = (int)(enc >> 26) + 1; }
            else tp = S.tris + (~cur);
            q0 = tp[0].q0; q1 = tp[0].q1; q2 = tp[0].q2;
            p0 = tp[1].q0; p1 = tp[1].q1; p2 = tp[1].q2;
        }
        asm volatile("" ::: "memory");      // loads stay above, their uses below

        // ---- one interior step for every descending lane
        if (descending && WIDE) {
            // four children: entry distances on the node's grid, culled against the best hit, then sorted
            // (the same arithmetic as wide_step, written out here: the call, though inlined, measured 7 % slower)
            const float cull = wide_cull(best.t);
            const uint32_t ex = __float_as_uint(n0.w);
            const float sx = __builtin_ldexpf(inv.x, (int)(int8_t)(ex & 255u)), sy = __builtin_ldexpf(inv.y, (int)(int8_t)((ex >> 8) & 255u)),
                        sz = __builtin_ldexpf(inv.z, (int)(int8_t)((ex >> 16) & 255u));
            const float bx = (n0.x - o.x) * inv.x, by = (n0.y - o.y) * inv.y, bz = (n0.z - o.z) * inv.z;
            const uint32_t lx = __float_as_uint(n1.x), ly = __float_as_uint(n1.y), lz = __float_as_uint(n1.z);
            const uint32_t hx = __float_as_uint(n1.w), hy = __float_as_uint(n2.x), hz = __float_as_uint(n2.y);
            int r0 = __float_as_int(n3.x), r1 = __float_as_int(n3.y), r2 = __float_as_int(n3.z), r3 = __float_as_int(n3.w);
            if (COUNT) tl.box += 1;
            // the near plane of a slab is its low one for a direction component >= 0, its high one otherwise (the plane
            // distance is monotonic in the cell coordinate): choosing the words once per node replaces a min and a max
            // per child and axis, with bit-identical results
            const bool ngx = inv.x < 0.f, ngy = inv.y < 0.f, ngz = inv.z < 0.f;
            const uint32_t ax = ngx ? hx : lx, fx = ngx ? lx : hx, ay = ngy ? hy : ly, fy = ngy ? ly : hy, az = ngz ? hz : lz, fz = ngz ? lz : hz;
            // (plain fused multiply-adds: the packed two-lane form issues at half their rate on this chip)
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
            float d0 = dk[0], d1 = dk[1], d2 = dk[2], d3 = dk[3];       // (a missing child's slabs are inverted: never entered)
            auto cswap = [](float& da, int& ra, float& db, int& rb) {
                bool sw = db < da;
                float td = sw ? db : da; int tr = sw ? rb : ra;
                db = sw ? da : db; rb = sw ? ra : rb;
                da = td; ra = tr;
            };
            cswap(d0, r0, d1, r1); cswap(d2, r2, d3, r3); cswap(d0, r0, d2, r2); cswap(d1, r1, d3, r3); cswap(d1, r1, d2, r2);      // (a partial sort, nearest only: 5 % more nodes visited)
            if (d3 < 1e30f) push(r3, d3);
            if (d2 < 1e30f) push(r2, d2);
            if (d1 < 1e30f) push(r1, d1);
            if (d0 < 1e30f) cur = r0;
            else { cur = pop_valid(); if (cur == PT_TRAV_EMPTY) retire(); }
        }
        if (descending && !WIDE) {
            int left, right;
            float dl, dr;
            if (QUANT) {
                uint32_t w0 = __float_as_uint(n0.x), w1 = __float_as_uint(n0.y), w2 = __float_as_uint(n0.z), w3 = __float_as_uint(n0.w);
                uint32_t w4 = __float_as_uint(n1.x), w5 = __float_as_uint(n1.y);
                float lmin[3] = {(float)(w0 & 0xffffu), (float)(w0 >> 16), (float)(w1 & 0xffffu)};
                float lmax[3] = {(float)(w1 >> 16), (float)(w2 & 0xffffu), (float)(w2 >> 16)};
                float rmin[3] = {(float)(w3 & 0xffffu), (float)(w3 >> 16), (float)(w4 & 0xffffu)};
                float rmax[3] = {(float)(w4 >> 16), (float)(w5 & 0xffffu), (float)(w5 >> 16)};
                left = __float_as_int(n1.z); right = __float_as_int(n1.w);
                dl = box_entry(oq, invq, lmin, lmax);
                dr = box_entry(oq, invq, rmin, rmax);
            } else {
                float lmin[3] = {n0.x, n0.y, n0.z}, lmax[3] = {n0.w, n1.x, n1.y};
                float rmin[3] = {n1.z, n1.w, n2.x}, rmax[3] = {n2.y, n2.z, n2.w};
                left = __float_as_int(n3.x); right = __float_as_int(n3.y);
                dl = box_entry(o, inv, lmin, lmax);
                dr = box_entry(o, inv, rmin, rmax);
            }
            if (COUNT) tl.box += 2;
            int nearR, farR; float dn, df;
            if (dl > dr) { nearR = right; dn = dr; farR = left; df = dl; }
            else         { nearR = left;  dn = dl; farR = right; df = dr; }
            if (df < best.t) push(farR, df);
            if (dn < best.t) cur = nearR;
            else { cur = pop_valid(); if (cur == PT_TRAV_EMPTY) retire(); }
        }
        // ---- leaf work for the lanes that waited at a leaf
        if (leaf_now) {
            const int count = WIDE ? leaf_count : __float_as_int(q1.w);
            auto test = [&](const float4& a0, const float4& a1, const float4& a2) {
                uint32_t id = __float_as_uint(a0.w);         // (twice the triangle's index, plus its opaque mark: see DTri. Hits carry it as it is)
                float t, u, v;
                if (COUNT) tl.tri += 1;
                if (tri_test(f3(a0.x, a0.y, a0.z), f3(a1.x, a1.y, a1.z), f3(a2.x, a2.y, a2.z), o, d, best.t, t, u, v) &&
                    !(best.tri != PT_NO_TRI && t == best.t && id > best.tri)) {
                    best.t = t; best.tri = id; best.u = u; best.v = v;
                }
            };
            // records come two at a time (the second of the first pair was fetched with the first; the
            // array has one record of padding, so reading one past a leaf is safe)
            test(q0, q1, q2);
            if (count > 1) {
                test(p0, p1, p2);
                for (int i = 2; i < count; i += 2) {
                    q0 = tp[i].q0; q1 = tp[i].q1; q2 = tp[i].q2;
                    p0 = tp[i + 1].q0; p1 = tp[i + 1].q1; p2 = tp[i + 1].q2;
                    test(q0, q1, q2);
                    if (i + 1 < count) test(p0, p1, p2);
                }
            }
            cur = pop_valid();
            if (cur == PT_TRAV_EMPTY) retire();
        }
        // ---- new rays for the lanes that were idle
        if (got && (!chunk_shadow || __float_as_int(rd4.w) != 0)) {
            const uint32_t nslot = idx | (chunk_shadow ? 0x80000000u : 0u);
            const F3 no = f3(ro4.x, ro4.y, ro4.z), nd = f3(rd4.x, rd4.y, rd4.z);
            F3 ninv = f3(0.f, 0.f, 0.f);
            if (COUNT) { tl.rays++; if (!chunk_shadow) tl.path_rays++; }
            int ncur = PT_TRAV_EMPTY;
            if (finite3(no) && finite3(nd)) {
                // the quantized boxes carry two cells of padding (the root box 0.1 units): the hardware reciprocal's
                // last-bit error cannot lose a box, and saves three IEEE divisions per ray
                if (WIDE) ninv = wide_inv(nd);
                else if (QUANT) ninv = f3(__builtin_amdgcn_rcpf(nd.x), __builtin_amdgcn_rcpf(nd.y), __builtin_amdgcn_rcpf(nd.z));
                else ninv = 1.0f / nd;
                if (COUNT && !WIDE) tl.box += 1;
#if PT_SKIP_ROOT_TEST
                // the 4-wide walk starts at the root node without testing the scene box: the root's child boxes cull
                // a ray that misses the scene in one step, and nearly every ray of a render starts inside it
                if (WIDE && S.root_ref4 >= 0) ncur = S.root_ref4;
                else
#endif
                {
                    float troot = box_entry(no, ninv, S.root_min, S.root_max);
                    if (troot < 1e10f) ncur = WIDE ? S.root_ref4 : S.root_ref;
                }
            }
            if (ncur == PT_TRAV_EMPTY) {
                // nothing to walk: the ray's record says so at once
                ((nslot >> 31) ? w.hitS : w.hitP)[nslot & 0x7FFFFFFFu] = make_float4(1e10f, __uint_as_float(PT_NO_TRI), 0.f, 0.f);
            } else if (TWO && has_ray) {
                pslot = nslot; po = no; pd = nd; pinv = ninv;
                pbest.t = 1e10f; pbest.tri = PT_NO_TRI; pbest.u = 0.f; pbest.v = 0.f;
                psp = 0; pcur = ncur;
            } else {
                slot = nslot; o = no; d = nd; inv = ninv;
                best.t = 1e10f; best.tri = PT_NO_TRI; best.u = 0.f; best.v = 0.f;
                sp = 0; cur = ncur;
#if PT_STAMPS
                st_age = 0;
#endif
                has_ray = true;
                if (QUANT && !WIDE) {
                    oq = (o - ld3(S.q_origin)) * ld3(S.q_inv_cell);
                    invq = inv * ld3(S.q_cell);
                    // an axis the ray is parallel to: a huge finite slope keeps q * inv - o * inv free of inf - inf and
                    // still puts both planes of a slab the origin is outside of on the same side (boxes carry two
                    // cells of padding, far more than the rounding of the products)
                    invq = f3(__builtin_fminf(__builtin_fmaxf(invq.x, -1e30f), 1e30f), __builtin_fminf(__builtin_fmaxf(invq.y, -1e30f), 1e30f),
                              __builtin_fminf(__builtin_fmaxf(invq.z, -1e30f), 1e30f));
                    noq = f3(-(oq.x * invq.x), -(oq.y * invq.y), -(oq.z * invq.z));
                }
            }
        }
    }
#if PT_STAMPS
    if (n_chunks * 10u >= (uint32_t)(w.n_slots / kRayChunk) * 9u) { unsigned long long* dbg = reinterpret_cast<unsigned long long*>(w.counters + WF_CTR_WORDS); if (st_h64) atomicAdd(dbg + 13, st_h64); if (st_h128) atomicAdd(dbg + 14, st_h128); if (st_h256) atomicAdd(dbg + 15, st_h256); atomicMax(dbg + 2, st_max); }
    if (PT_STAMPS == 1 && lane == 0 && n_chunks * 10u >= (uint32_t)(w.n_slots / kRayChunk) * 9u) {      // full launches only
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(w.counters + WF_CTR_WORDS);
        atomicAdd(dbg + 0, __builtin_amdgcn_s_memtime() - st_t0); atomicAdd(dbg + 1, st_refills); atomicAdd(dbg + 3, st_iters);
        atomicAdd(dbg + 4, st_lanes); atomicAdd(dbg + 5, st_leafphases); atomicAdd(dbg + 6, 1ull); atomicAdd(dbg + 7, st_leaflanes); atomicAdd(dbg + 8, st_idle); atomicAdd(dbg + 9, st_tail); atomicAdd(dbg + 10, st_tail_idle); atomicAdd(dbg + 11, st_tail_desc); atomicAdd(dbg + 12, st_tail ? __builtin_amdgcn_s_memtime() - st_tail_t0 : 0ull);
    }
#endif
    flush_tally<COUNT>(ctr, tl, 0);
}

// ---------------------------------------------------------------- path hit
// A path vertex (pathtracer.cu:140-181 for the hit the traversal kept, then :426-481) for every record of the path
// queue whose ray hit a surface: shading normal and material values at the hit with up to five filtered texture
// lookups (a chain of dependent gathers: hit -> triangle -> material -> texture -> texels), medium event, alpha test,
// light-sample set-up. A vertex that drew a light sample takes a record of the next shadow queue; a ray that passed
// through the surface moves on to the next path queue; rays that left the scene are listed for the miss stage. The
// path record is fetched beside the hit, not after it.
// The two halves were kernels of their own at first (the gathers at five waves per SIMD, the rest at four, the
// resolved surface handed over through the record): one kernel on the state in registers is 2.1 % faster on the bench
// at four waves per SIMD (128 registers; the same as the two kernels at three, 112 spilled registers at five).
// DIRECT: for scenes that draw no light samples (sun/sky mode). The vertex has no visibility walk to wait for, so
// its bsdf stage (after_light) runs here as well and no shadow record is written: three waves per SIMD (168
// registers, 29 spilled; 152 spilled at four: -4 %), 10 % faster on the sun/sky scene than resolve and vertex + bsdf
// as two kernels.
// TABLES: the material records and texture descriptors are copied to LDS first (scenes with at most kLocalMats
// materials and kLocalTex textures): a lane's seven 16-byte reads of its material are then LDS reads instead of
// seven gathers the wave sends through the texture addresser, which this kernel keeps busy for 0.7 of its time (+0.8 %).
#ifndef PT_PATHHIT_WAVES
#define PT_PATHHIT_WAVES 4
#endif
#ifndef PT_PATHHIT_DIRECT_WAVES
#define PT_PATHHIT_DIRECT_WAVES 3
#endif
#ifndef PT_PATHHIT_TABLES
#define PT_PATHHIT_TABLES 1
#endif
constexpr int kLocalMats = 64, kLocalTex = 64;
template <bool COUNT, bool TABLES, bool DIRECT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(DIRECT ? PT_PATHHIT_DIRECT_WAVES : PT_PATHHIT_WAVES))) void wf_pathhit_kernel(WfParams w, DScene S, DCamera cam, DCounters* ctr)
{
    __shared__ uint32_t counts[kStripes];
    __shared__ float4 s_mats[TABLES ? kLocalMats * 8 : 1];
    __shared__ DTex s_tex[TABLES ? kLocalTex : 1];
    if (TABLES) {
        for (int k = (int)threadIdx.x; k < S.n_mats * 8; k += kBlock) s_mats[k] = reinterpret_cast<const float4*>(S.mats)[k];
        for (int k = (int)threadIdx.x; k < S.n_tex; k += kBlock) s_tex[k] = S.tex[k];
        __syncthreads();
    }
    BlockCursor bc = first_block(w.counters + WF_CTR_SEG_PATH, w.counters + WF_CTR_N_P, 1, counts);
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t nsamples = 0;
    for (;;) {
        const uint32_t blk = next_block(w.counters + WF_CTR_SEG_PATH, counts, bc);
        if (blk == kNone) break;                    // uniform over the workgroup
        const uint32_t stripe = blk & (uint32_t)(kStripes - 1);
        const uint32_t i = blk * kBlock + threadIdx.x;
        const bool have = threadIdx.x < block_fill(blk, counts);
        Hit h; h.t = 0.f; h.tri = PT_NO_TRI; h.u = 0.f; h.v = 0.f;
        if (have) { float4 h4 = w.hitP[i]; h.t = h4.x; h.tri = hit_triangle(__float_as_uint(h4.y)); h.u = h4.z; h.v = h4.w; }
        const bool hit = have && h.tri != PT_NO_TRI;
        Path P;
        SampleId id = {0, 0, 0, 0u, 0u};
        uint32_t sample = kNone;
        P.state = ST_DONE;
        bool finished = false;
        float mat_bits = 0.f;
        bool emis_map = false;
        float4 pc0, pc1, pc2;           // (fetched beside the hit: one dependent load less; the one ray in six that misses pays 48 bytes for nothing)
        if (have) { pc0 = w.Pc.c0[i]; pc1 = w.Pc.c1[i]; pc2 = w.Pc.c2[i]; }
        if (hit) {
            P.so = P.sd = P.light = P.env_color = f3(0.f, 0.f, 0.f); P.env_pdf = 0.f;
            sample = unpack_core(pc0, pc1, pc2, w, P, id);
            Surf info;
            resolve_hit<COUNT, TABLES>(S, h, P.d, info, tl.taps, tl.taps_r, false, nullptr, s_mats, s_tex);
            mat_bits = __int_as_float(S.tri_info ? (int)(S.tri_info[h.tri] & 0x7FFFFFFFu) : S.tri_ref[h.tri].matIndex);
            emis_map = info.emis_map;
            on_path_vertex<COUNT>(S, cam, P, info, tl);
            if (DIRECT && P.state == ST_AFTER) after_light<COUNT>(S, cam, P, tl);      // (no light sample to wait for: the bsdf stage runs here)
            finished = P.state == ST_DONE;
        }
        finish_and_regenerate<COUNT>(w, cam, P, sample, id, finished, nsamples);
        const bool alive = hit && P.state != ST_DONE;
        Appends a;
        a.want[0] = alive && P.state == ST_PATH; a.tail[0] = w.counters + WF_CTR_TAIL_P;
        a.want[1] = alive && P.state == ST_SHADOW; a.tail[1] = w.counters + WF_CTR_TAIL_S;
        a.want[2] = have && !hit; a.tail[2] = w.counters + WF_CTR_TAIL_M;
        reserve<3>(a, stripe);
        if (a.want[0]) store_path(w.Pn, a.pos[0], P, sample);
        if (a.want[1]) {
            const uint32_t j = a.pos[1];
            const Mat& m = P.vm;
            // 128 bytes: the walk's first segment (its transmittance starts at (1, 1, 1): sd.w = 1 says that d0 holds nothing yet),
            // the path state with the vertex's metallic value in the place of the last bounce's pdf (dead at a vertex), the light
            // sample, and what the maps gave; the emission only where it came out of a map
            const float metal = m.metal;
            w.Sn.so[j] = make_float4(P.so.x, P.so.y, P.so.z, mat_bits);
            w.Sn.sd[j] = make_float4(P.sd.x, P.sd.y, P.sd.z, __int_as_float(1));
            P.pdf = metal;
            store_core(w.Sn.c0, w.Sn.c1, w.Sn.c2, j, P, sample);
            w.Sn.d1[j] = make_float4(P.env_color.x, P.env_color.y, P.env_color.z, P.env_pdf);
            w.Sn.e0[j] = make_float4(m.base.x, m.base.y, m.base.z, m.rough);
            if (emis_map) w.Sn.e1[j] = make_float4(m.emis.x, m.emis.y, m.emis.z, 0.f);
            w.Sn.e4[j] = make_float4(P.vn.x, P.vn.y, P.vn.z, m.alpha);
        }
        if (a.want[2]) w.miss_list[a.pos[2]] = i;
    }
    flush_tally<COUNT>(ctr, tl, nsamples);
}

// ---------------------------------------------------------------- miss
// The path rays that left the scene (pathtracer.cu:409-425): environment radiance, then every lane's sample is
// finished and the lane starts a new one. Dense in both respects, which the same work was not as a branch of the
// path stage: one ray in six misses there, and only those lanes start new samples.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void wf_miss_kernel(WfParams w, DScene S, DCamera cam, DCounters* ctr)
{
    __shared__ uint32_t s_counts[kStripes];
    BlockCursor bc = first_block(w.counters + WF_CTR_SEG_MISS, w.counters + WF_CTR_TAIL_M, kStripeWords, s_counts);
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t nsamples = 0;
    for (;;) {
        const uint32_t blk = next_block(w.counters + WF_CTR_SEG_MISS, s_counts, bc);
        if (blk == kNone) break;
        const uint32_t stripe = blk & (uint32_t)(kStripes - 1);
        const uint32_t k = blk * kBlock + threadIdx.x;
        const bool have = threadIdx.x < block_fill(blk, s_counts);
        Path P;
        SampleId id = {0, 0, 0, 0u, 0u};
        uint32_t sample = kNone;
        P.state = ST_DONE;
        if (have) {
            sample = load_path(w.Pc, w.miss_list[k], w, P, id);
            on_path_miss<COUNT>(S, P, tl);
        }
        finish_and_regenerate<COUNT>(w, cam, P, sample, id, have, nsamples);
        Appends a;
        a.want[0] = have && P.state == ST_PATH; a.tail[0] = w.counters + WF_CTR_TAIL_P;
        reserve<1>(a, stripe);
        if (a.want[0]) store_path(w.Pn, a.pos[0], P, sample);
    }
    flush_tally<COUNT>(ctr, tl, nsamples);
}

// ---------------------------------------------------------------- shadow
// One segment of visibility (pathtracer.cu:279-295) for every record of the shadow queue. The segment looks at
// alpha, metallic, transmission, ior and the medium of the surface it hit, and draws one number: it needs the
// walk's part of the record only (64 of 192 bytes) and few registers, so many waves fit a SIMD and hide the
// dependent loads (hit -> triangle -> material -> texture -> texels). A finished walk stays where it is and its
// position goes onto the bsdf list, ordered within the workgroup by the kind of work the bsdf stage will do
// for it; a walk that goes on (it passed an alpha cut-out or an index-matched surface) moves to the next shadow queue.
#ifndef PT_SHADOW_WAVES
#define PT_SHADOW_WAVES 8       // 71 registers: no spill at 8 (6: -1 % on the bench)
#endif
// What one segment makes of a record of the shadow queue.
struct Walk {
    bool have, to_bsdf, goes_on;        // the record waits for a hit; its walk has ended; it goes on with another segment
    bool light_loaded, light_changed;   // the transmittance was in the record (d0); this segment changed it
    uint32_t kind;                      // of work the bsdf stage will do for it
    uint32_t flags, draws;              // the sample's state word and draw count after the segment
    F3 light, step;                     // transmittance after the segment; what a walk that goes on adds to its origin
    float4 sd4, c2;                     // the record's segment and state words as read
};
template <bool COUNT>
PT_DEV void walk_segment(const WfParams& w, const DScene& S, const DCamera& cam, uint32_t i, bool in_block, Tally& tl, Walk& k)
{
    // (the hit and the state word are fetched with the segment, not after it: nearly every record of a launch that
    // has any still waits for its hit, and the chain of dependent loads is one shorter)
    float4 sd4 = make_float4(0.f, 0.f, 0.f, 0.f), h4 = sd4, c2_in = sd4;
    if (in_block) { sd4 = w.Sc.sd[i]; h4 = w.hitS[i]; c2_in = w.Sc.c2[i]; }
    // (every record of the queue carries a pending ray, marked in sd.w; the check guards the material index behind it)
    const bool have = in_block && __float_as_int(sd4.w) != 0;
    Path P;
    P.state = ST_DONE; P.lobes = 0; P.nee_surface = false; P.nee_skip = false;
    P.light = f3(1.f, 1.f, 1.f); P.env_pdf = 0.f;
    P.so = f3(0.f, 0.f, 0.f); P.rng.draws = 0u;
    float4 c2 = sd4;
    // The transmittance (d0) exists only for the few walks of which a segment crossed a medium (sd.w == 2). For every
    // other walk it is (1, 1, 1).
    bool light_loaded = false, light_changed = false;
    if (have) {
        c2 = c2_in;
        unpack_flags(P, __float_as_uint(c2.y));
        if (__float_as_int(sd4.w) == 2) {
            float4 l0 = w.Sc.d0[i];
            P.light = f3(l0.x, l0.y, l0.z);
            light_loaded = true;
        }
        P.so = f3(0.f, 0.f, 0.f); P.sd = f3(sd4.x, sd4.y, sd4.z);
        Hit h; h.t = h4.x; h.tri = hit_triangle(__float_as_uint(h4.y)); h.u = h4.z; h.v = h4.w;
        const bool in_medium = P.sh_inside;
        // A material marked opaque at upload ends the walk whatever the hit point: the alpha test draws its
        // number and fails, the transmittance is zero (pathtracer.cu:279-295). Most hits are of this kind and
        // skip the texture coordinates, the material record, the texels and the random-number block. The mark comes with
        // the hit (from the triangle record the traversal kernel tested), so nothing is looked up behind it.
        bool opaque = false;
        if (!COUNT && h.tri != PT_NO_TRI) opaque = S.tri_info ? (__float_as_uint(h4.y) & 1u) != 0 : S.mats[S.tri_ref[h.tri].matIndex].opaque != 0;
        if (opaque) {
            P.rng.draws = __float_as_uint(c2.z) + 1u;
            P.light = f3(0.f, 0.f, 0.f);
            P.state = ST_AFTER;
        } else {
            const SampleId id = decode_sample(w, __float_as_uint(c2.w));
            rng_seed(P.rng, id.pixel, (uint32_t)id.frame, __float_as_uint(c2.z));
            Surf info;
            resolve_hit<COUNT>(S, h, P.sd, info, tl.taps, tl.taps_r, true);
            on_shadow_hit<COUNT>(S, cam, P, info, tl);
            light_changed = in_medium && info.hit;
        }
        if (P.state == ST_AFTER && dot(P.light, P.light) <= 0.05f) {
            // Blocked (or all but): the test of pathtracer.cu:318 is taken here and the record says so through
            // the flag that skips the light's term, so the transmittance is not passed on.
            P.nee_skip = true;
            light_changed = false;
        }
    }
    k.have = have;
    k.to_bsdf = have && P.state == ST_AFTER;
    k.goes_on = have && P.state == ST_SHADOW;
    // Order of a workgroup's entries for the bsdf stage: by which lobes the material carries (the branches of
    // DisneyEval_2, disney.cuh:294-344), surface or medium vertex, and whether the light sample arrived (its
    // evaluation is skipped otherwise, pathtracer.cu:318).
    k.kind = 0;
    if (k.to_bsdf) {
        const bool lit = !P.nee_skip && !(dot(P.light, P.light) <= 0.05f);
        k.kind = (uint32_t)(P.lobes & 15) | (P.nee_surface ? 0u : 16u) | (lit ? 32u : 0u);
    }
    k.light_loaded = light_loaded; k.light_changed = light_changed;
    k.flags = have ? pack_flags(P) : 0u; k.draws = P.rng.draws;
    k.light = P.light; k.step = P.so;
    k.sd4 = sd4; k.c2 = c2;
}
// Counting sort of a workgroup's finished walks by kind, in LDS (s_kind[64], cleared before the walks' segments and a
// barrier): the waves of the bsdf stage then run mostly one kind each instead of every branch for a few lanes. Every
// thread calls it; returns the number of finished walks of the block and, for each, its place among them.
PT_DEV uint32_t sort_by_kind(uint32_t* s_kind, bool to_bsdf, uint32_t kind, uint32_t& place)
{
    const int lane = (int)(threadIdx.x & 63u);
    uint32_t rank = 0;
    // one LDS atomic per kind a wave holds (a wave's lanes are mostly of one or two kinds: per-lane atomics on
    // one word would be served one after the other)
    for (unsigned long long todo = __ballot(to_bsdf); todo != 0ull;) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k = (uint32_t)__shfl((int)kind, leader, 64);
        const bool mine = to_bsdf && kind == k;
        const unsigned long long same = __ballot(mine);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&s_kind[k], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (mine) rank = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    __syncthreads();
    // exclusive prefix over the kinds, computed by every wave for itself (lane l holds kind l)
    const uint32_t held = s_kind[lane];
    uint32_t incl = held;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)incl, dlt, 64);
        if (lane >= dlt) incl += t;
    }
    place = (uint32_t)__shfl((int)(incl - held), (int)kind, 64) + rank;
    return (uint32_t)__shfl((int)incl, 63, 64);
}
// a walk that goes on: its record, with the next segment, moves to position j of the next shadow queue
PT_DEV void store_next_segment(const WfParams& w, uint32_t i, uint32_t j, const Walk& k)
{
    // (k.step holds the step along the ray from the record's origin)
    const float4 so4 = w.Sc.so[i];
    const F3 o = f3(so4.x, so4.y, so4.z) + k.step;
    w.Sn.so[j] = make_float4(o.x, o.y, o.z, so4.w);
    const bool crossed = k.light_loaded || k.light_changed;
    w.Sn.sd[j] = make_float4(k.sd4.x, k.sd4.y, k.sd4.z, __int_as_float(crossed ? 2 : 1));
    w.Sn.c0[j] = w.Sc.c0[i]; w.Sn.c1[j] = w.Sc.c1[i];
    w.Sn.c2[j] = make_float4(k.c2.x, __uint_as_float(k.flags), __uint_as_float(k.draws), k.c2.w);
    if (crossed) w.Sn.d0[j] = make_float4(k.light.x, k.light.y, k.light.z, 0.f);
    w.Sn.d1[j] = w.Sc.d1[i];
    w.Sn.e0[j] = w.Sc.e0[i]; w.Sn.e1[j] = w.Sc.e1[i]; w.Sn.e4[j] = w.Sc.e4[i];
}
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_SHADOW_WAVES))) void wf_shadow_kernel(WfParams w, DScene S, DCamera cam, DCounters* ctr)
{
    constexpr int kKinds = 64;
    __shared__ uint32_t s_kind[kKinds];
    __shared__ uint32_t s_list_base;
    __shared__ uint32_t counts[kStripes];
    BlockCursor bc = first_block(w.counters + WF_CTR_SEG_SHADOW, w.counters + WF_CTR_N_S, 1, counts);
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
    for (;;) {
        if (threadIdx.x < kKinds) s_kind[threadIdx.x] = 0u;
        const uint32_t blk = next_block(w.counters + WF_CTR_SEG_SHADOW, counts, bc);      // (its barriers also publish the cleared counts)
        if (blk == kNone) break;
        const uint32_t stripe = blk & (uint32_t)(kStripes - 1);
        const uint32_t i = blk * kBlock + threadIdx.x;
        Walk k;
        walk_segment<COUNT>(w, S, cam, i, threadIdx.x < block_fill(blk, counts), tl, k);
        uint32_t place = 0;
        const uint32_t listed = sort_by_kind(s_kind, k.to_bsdf, k.kind, place);
        if (threadIdx.x == 0) s_list_base = listed ? atomicAdd(w.counters + WF_CTR_TAIL_B + stripe * (uint32_t)kStripeWords, listed) : 0u;
        Appends a;
        a.want[0] = k.goes_on; a.tail[0] = w.counters + WF_CTR_TAIL_S;
        reserve<1>(a, stripe);                      // (its barriers also publish s_list_base and protect s_kind)
        if (k.to_bsdf) {
            // the walk's result stays in the record: transmittance (where a medium changed it), state and the draws
            // it consumed
            if (k.light_changed) {
                w.Sc.d0[i] = make_float4(k.light.x, k.light.y, k.light.z, 0.f);
                if (!k.light_loaded) w.Sc.sd[i] = make_float4(k.sd4.x, k.sd4.y, k.sd4.z, __int_as_float(2));
            }
            w.Sc.c2[i] = make_float4(k.c2.x, __uint_as_float(k.flags), __uint_as_float(k.draws), k.c2.w);
            w.bsdf_list[stripe_pos(stripe, s_list_base + place)] = i;
        }
        if (k.goes_on) store_next_segment(w, i, a.pos[0], k);
    }
    flush_tally<COUNT>(ctr, tl, 0);
}

// ---------------------------------------------------------------- bsdf
// Everything after the light sample of a vertex (pathtracer.cu:318-338, :449-451 or :471-480, :484-490) for the
// records on the bsdf list.
#ifndef PT_BSDF_WAVES
#define PT_BSDF_WAVES 3         // (4 spills 38 registers: -0.5 %; 5: -6 %)
#endif
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_BSDF_WAVES))) void wf_bsdf_kernel(WfParams w, DScene S, DCamera cam, DCounters* ctr)
{
    __shared__ uint32_t s_counts[kStripes];
    BlockCursor bc = first_block(w.counters + WF_CTR_SEG_BSDF, w.counters + WF_CTR_TAIL_B, kStripeWords, s_counts);
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t nsamples = 0;
    for (;;) {
        const uint32_t blk = next_block(w.counters + WF_CTR_SEG_BSDF, s_counts, bc);
        if (blk == kNone) break;
        const uint32_t stripe = blk & (uint32_t)(kStripes - 1);
        const uint32_t k = blk * kBlock + threadIdx.x;
        const bool have = threadIdx.x < block_fill(blk, s_counts);
        Path P;
        SampleId id = {0, 0, 0, 0u, 0u};
        uint32_t sample = kNone;
        P.state = ST_DONE;
        bool finished = false;
        if (have) {
            sample = load_shadow(w.Sc, w.bsdf_list[k], w, S, P, id);
            after_light<COUNT>(S, cam, P, tl);
            finished = P.state == ST_DONE;
        }
        finish_and_regenerate<COUNT>(w, cam, P, sample, id, finished, nsamples);
        Appends a;
        a.want[0] = have && P.state == ST_PATH; a.tail[0] = w.counters + WF_CTR_TAIL_P;
        reserve<1>(a, stripe);
        if (a.want[0]) store_path(w.Pn, a.pos[0], P, sample);
    }
    flush_tally<COUNT>(ctr, tl, nsamples);
}

// ---------------------------------------------------------------- shadow and bsdf stages in one kernel
// The two stages above on one block of the shadow queue at a time: the block's segments, the sort of its finished walks
// by kind, then the bsdf stage for them in sorted order (thread t takes the t-th finished walk). What a walk leaves for
// the bsdf stage (state word, draw count, transmittance) goes through LDS instead of the record, and there is no list.
// Same images. Measured on the bench against the two kernels: -0.5 % at three waves per SIMD (the segments' two dependent
// loads are hidden by eight waves in a kernel of their own, not by three), +0.35 % at four (35 spilled registers): not
// enabled (-DPT_FUSED_LIGHT=1 -DPT_BSDF_WAVES=4).
#ifndef PT_FUSED_LIGHT
#define PT_FUSED_LIGHT 0
#endif
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_BSDF_WAVES))) void wf_light_kernel(WfParams w, DScene S, DCamera cam, DCounters* ctr)
{
    constexpr int kKinds = 64;
    __shared__ uint32_t s_kind[kKinds];
    __shared__ uint32_t counts[kStripes];
    __shared__ uint32_t s_from[kBlock];         // which record of the block the t-th finished walk is
    __shared__ uint2 s_state[kBlock];           // per record of the block: state word and draw count after the walk
    __shared__ float4 s_light[kBlock];          // transmittance after the walk (w != 0: this segment changed it)
    BlockCursor bc = first_block(w.counters + WF_CTR_SEG_SHADOW, w.counters + WF_CTR_N_S, 1, counts);
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t nsamples = 0;
    for (;;) {
        if (threadIdx.x < kKinds) s_kind[threadIdx.x] = 0u;
        const uint32_t blk = next_block(w.counters + WF_CTR_SEG_SHADOW, counts, bc);      // (its barriers also publish the cleared counts)
        if (blk == kNone) break;
        const uint32_t stripe = blk & (uint32_t)(kStripes - 1);
        const uint32_t i = blk * kBlock + threadIdx.x;
        uint32_t listed;
        {
            Walk k;
            walk_segment<COUNT>(w, S, cam, i, threadIdx.x < block_fill(blk, counts), tl, k);
            uint32_t place = 0;
            listed = sort_by_kind(s_kind, k.to_bsdf, k.kind, place);
            if (k.to_bsdf) {
                s_from[place] = threadIdx.x;
                s_state[threadIdx.x] = make_uint2(k.flags, k.draws);
                s_light[threadIdx.x] = make_float4(k.light.x, k.light.y, k.light.z, k.light_changed ? 1.f : 0.f);
            }
            Appends a;
            a.want[0] = k.goes_on; a.tail[0] = w.counters + WF_CTR_TAIL_S;
            reserve<1>(a, stripe);                  // (its barriers also publish the three arrays and protect s_kind)
            if (k.goes_on) store_next_segment(w, i, a.pos[0], k);
        }
        const bool have = threadIdx.x < listed;
        Path P;
        SampleId id = {0, 0, 0, 0u, 0u};
        uint32_t sample = kNone;
        P.state = ST_DONE;
        bool finished = false;
        if (have) {
            const uint32_t from = s_from[threadIdx.x];
            const uint32_t r = blk * kBlock + from;
            float4 c0 = w.Sc.c0[r], c1 = w.Sc.c1[r], c2 = w.Sc.c2[r];
            const uint2 st = s_state[from];
            c2.y = __uint_as_float(st.x); c2.z = __uint_as_float(st.y);
            sample = unpack_core(c0, c1, c2, w, P, id);
            load_vertex(w.Sc, r, S, P);
            const float4 l = s_light[from];
            if (l.w != 0.f) P.light = f3(l.x, l.y, l.z);
            after_light<COUNT>(S, cam, P, tl);
            finished = P.state == ST_DONE;
        }
        finish_and_regenerate<COUNT>(w, cam, P, sample, id, finished, nsamples);
        Appends ap;
        ap.want[0] = have && P.state == ST_PATH; ap.tail[0] = w.counters + WF_CTR_TAIL_P;
        reserve<1>(ap, stripe);
        if (ap.want[0]) store_path(w.Pn, ap.pos[0], P, sample);
    }
    flush_tally<COUNT>(ctr, tl, nsamples);
}

// ---------------------------------------------------------------- leftover sample ids
// Once the global hand-out has run dry, the ids that workgroups still hold in their chunks would wait for those
// workgroups to finish samples of their own, which the thinning queues give them ever more rarely. This kernel
// (workgroup b empties chunk b) starts them all at once, as camera rays at the tail of the path queue the next
// iteration reads. From then on no workgroup holds an id, and a lane whose sample ends emits nothing.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void wf_flush_leftover_kernel(WfParams w, DCamera cam, DCounters* ctr)
{
    uint32_t* cw = w.wg_chunk + 2 * (size_t)blockIdx.x;
    const uint32_t cur = cw[0], end = cw[1] < w.n_samples ? cw[1] : w.n_samples;
    if (cur >= end) return;                         // uniform over the workgroup
    __syncthreads();                                // every thread has read the chunk before it is emptied
    if (threadIdx.x == 0) { cw[0] = w.n_samples; cw[1] = w.n_samples; }
    const uint32_t s = cur + threadIdx.x;           // a chunk holds at most kSampleChunk = kBlock ids
    Path P;
    P.state = ST_DONE;
    uint32_t nsamples = 0;
    bool have = s < end;
    if (have) {
        const SampleId id = decode_sample(w, s);
        P.so = P.sd = P.light = P.env_color = f3(0.f, 0.f, 0.f); P.env_pdf = 0.f;
        P.sh_inside = P.nee_surface = P.nee_skip = P.v_inside = false; P.seg = 0;
        begin_sample(P, cam, id.x, id.y, id.frame);
        if (COUNT) nsamples++;
        if (P.state == ST_DONE) { w.staging[id.stage_index] = 0.f; have = false; }
    }
    Appends a;
    a.want[0] = have; a.tail[0] = w.counters + WF_CTR_TAIL_P;
    reserve<1>(a, blockIdx.x & (uint32_t)(kStripes - 1));     // (the queue arrays have room for one chunk per workgroup dealt this way)
    if (have) store_path(w.Pn, a.pos[0], P, s);
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
    flush_tally<COUNT>(ctr, tl, nsamples);
}

// ---------------------------------------------------------------- drain
// End of a render: every sample id has been started and few records are left. Another hundred stage launches
// for the longest paths would cost their launch latencies in sequence; instead each remaining record runs its
// sample to the end in one kernel (the same state machine, one path per lane, traversal with the LDS stack).
// Workgroup b takes block b of the path queue, or block b - (blocks of the path queue) of the shadow queue.
template <bool COUNT, bool WIDE>
__global__ __launch_bounds__(kBlock) void wf_drain_kernel(WfParams w, DScene S, DCamera cam, DCounters* ctr)
{
    const uint32_t n_pblocks = w.counters[WF_CTR_RAY_P_CHUNKS];
    if (blockIdx.x >= w.counters[WF_CTR_RAY_CHUNKS]) return;
    const bool from_shadow = blockIdx.x >= n_pblocks;
    const uint32_t blk = from_shadow ? blockIdx.x - n_pblocks : blockIdx.x;
    const uint32_t q = blk * (uint32_t)kBlock + threadIdx.x;
    bool have = threadIdx.x < block_fill(blk, w.counters + (from_shadow ? WF_CTR_N_S : WF_CTR_N_P));    // lanes past the end stay for the wave-wide tally
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0};
    Path P;
    SampleId id = {0, 0, 0, 0u, 0u};
    P.state = ST_DONE;
    if (have) {
        if (!from_shadow) load_path(w.Pc, q, w, P, id);
        else {
            // (a record without a pending ray, sd.w == 0, holds no material index to follow: none is written at present)
            if (__float_as_int(w.Sc.sd[q].w) == 0) have = false;
            else load_shadow(w.Sc, q, w, S, P, id);
        }
    }
    if (have) {
        while (P.state != ST_DONE) {
            if (P.state == ST_AFTER) { after_light<COUNT>(S, cam, P, tl); continue; }
            bool shadow = P.state == ST_SHADOW;
            F3 ro = shadow ? P.so : P.o;
            F3 rd = shadow ? P.sd : P.d;
            Hit h;
            if (WIDE) h = closest_hit_wide<kBlock, kLdsDepth, COUNT>(S, ro, rd, threadIdx.x, tl.box, tl.tri);
            else h = closest_hit<kBlock, kLdsDepth, COUNT>(S, ro, rd, threadIdx.x, tl.box, tl.tri);
            if (COUNT) { tl.rays++; if (!shadow) tl.path_rays++; }
            Surf info;
            resolve_hit<COUNT>(S, h, rd, info, tl.taps, tl.taps_r, shadow);
            if (shadow) on_shadow_hit<COUNT>(S, cam, P, info, tl);
            else on_path_hit<COUNT>(S, cam, P, info, tl);
        }
        float v = P.result * 3.f;
        w.staging[id.stage_index] = v > 0.f ? v : 0.f;
    }
    flush_tally<COUNT>(ctr, tl, 0);
}

// ---------------------------------------------------------------- accumulate
// pathtracer.cu:583-586 over the chunk: per pixel, frames in order; the channel a sample went to is
// re-derived from its random stream (draw 2, or 4 with the thin lens: pathtracer.cu:562,577,527).
__global__ __launch_bounds__(kBlock) void wf_accumulate_kernel(WfParams w, DCamera cam, float* __restrict__ accum, int first, int count)
{
    uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= w.n_pixels) return;
    int x, y;
    lane_pixel(w, q, x, y);
    uint32_t p = (uint32_t)(x + y * w.W);
    F3 acc = f3(accum[3 * (size_t)p], accum[3 * (size_t)p + 1], accum[3 * (size_t)p + 2]);
    uint32_t ch_draw = cam.dof > 0.f ? 4u : 2u;
    for (int f = first; f < first + count; f++) {
        float v = w.staging[(size_t)f * w.n_pixels + q];
        Rng g;
        rng_seed(g, p, (uint32_t)(w.frame_begin + f), ch_draw);
        int ch = int(rng_next(g) * 3.);
        F3 c0 = f3(ch == 0 ? v : 0.f, ch == 1 ? v : 0.f, ch == 2 ? v : 0.f);
        acc = acc + c0;
    }
    accum[3 * (size_t)p] = acc.x; accum[3 * (size_t)p + 1] = acc.y; accum[3 * (size_t)p + 2] = acc.z;
}

// ---------------------------------------------------------------- closest-hit batch through the production traversal
// The batch is laid out as a shadow queue of n records (origins, directions with the pending flag set) and an
// empty path queue, which is all the extend kernel looks at.
__global__ __launch_bounds__(kBlock) void wf_pack_rays_kernel(const pt_ray* __restrict__ rays, int n, float4* __restrict__ ray_o, float4* __restrict__ ray_d,
                                                              uint32_t* __restrict__ counters)
{
    int i = blockIdx.x * kBlock + threadIdx.x;      // the grid covers n rounded up to whole blocks
    if (i < kStripes) {
        // positions 0 .. n-1 in a row, seen as stripes: stripe r holds the blocks r, r + kStripes, ... below n / 256,
        // and the stripe of block n / 256 the remainder
        const uint32_t full = (uint32_t)n >> 8, r = (uint32_t)i;
        const uint32_t mine = full > r ? ((full - r - 1u) >> kStripeShift) + 1u : 0u;
        counters[WF_CTR_N_P + i] = 0;
        counters[WF_CTR_N_S + i] = (mine << 8) + ((full & (uint32_t)(kStripes - 1)) == r ? (uint32_t)n & 255u : 0u);
    }
    if (i == 0) {
        counters[WF_CTR_RAY_NEXT] = 0;
        counters[WF_CTR_RAY_P_CHUNKS] = 0; counters[WF_CTR_RAY_CHUNKS] = (uint32_t)((n + kBlock - 1) / kBlock);
    }
    if (i >= n) return;
    const float4* r = reinterpret_cast<const float4*>(rays + i);
    float4 o = r[0], d = r[1];
    ray_o[i] = make_float4(o.x, o.y, o.z, 0.f);
    ray_d[i] = make_float4(d.x, d.y, d.z, __int_as_float(1));
}
__global__ __launch_bounds__(kBlock) void wf_unpack_hits_kernel(const float4* __restrict__ hit, int n, pt_hit* __restrict__ out)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 h = hit[i];
    pt_hit o; o.t = h.x; o.tri = hit_triangle(__float_as_uint(h.y));
    out[i] = o;
}
// w: Sc.so, Sc.sd and hitS with room for n rounded up to whole blocks, counters, and the extend tuning
void launch_wf_trace(const WfParams& w, const DScene& S, const pt_ray* rays_dev, int n, pt_hit* hits_dev, hipStream_t s)
{
    dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
    hipLaunchKernelGGL(wf_pack_rays_kernel, grid, block, 0, s, rays_dev, n, w.Sc.so, w.Sc.sd, w.counters);
    int ext_blocks = (int)grid.x < w.extend_blocks ? (int)grid.x : w.extend_blocks;
    if (S.wide) hipLaunchKernelGGL((wf_extend_kernel<false, true>), dim3(ext_blocks), block, 0, s, w, S, (DCounters*)nullptr);
    else hipLaunchKernelGGL((wf_extend_kernel<false, false>), dim3(ext_blocks), block, 0, s, w, S, (DCounters*)nullptr);
    hipLaunchKernelGGL(wf_unpack_hits_kernel, grid, block, 0, s, w.hitS, n, hits_dev);
}

// ---------------------------------------------------------------- closest-hit batch without allocations
// The batch above borrows queue arrays that the library allocates on demand. Callers that must not allocate
// (code under graph capture, serving loops) get the same kernels over a fixed block of device memory that
// is part of the library image: ray origins, ray directions and hits of up to kStaticMaxRays rays, and the
// counter words.
constexpr size_t kStaticCounterBytes = WF_CTR_WORDS * sizeof(uint32_t) + 16 * sizeof(unsigned long long);
constexpr size_t kStaticBytes = 3 * (size_t)kStaticMaxRays * sizeof(float4) + kStaticCounterBytes;
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

// Every GPU holds its own copy of the block at its own address; the address is looked up once per device,
// with that device current, and kept. Host code never uses the array's name for anything else.
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[kStaticMaxDevices];
static bool g_scratch_known[kStaticMaxDevices];

hipError_t wf_static_scratch(int device, void** ptr_out)
{
    *ptr_out = nullptr;
    if (device < 0 || device >= kStaticMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_known[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = p;
        g_scratch_known[device] = true;
    }
    *ptr_out = g_scratch_ptr[device];
    return hipSuccess;
}

// n <= kStaticMaxRays rays through the production traversal kernel, every intermediate in the block at
// `base` (what wf_static_scratch returned for the current device). The block keeps the previous call's
// contents: the pack kernel rewrites the ray arrays of every position the launch covers and the counter words
// it hands to the traversal kernel, and the counter words are cleared first, so nothing of an earlier call is read.
hipError_t launch_wf_trace_static(void* base, const DScene& S, const pt_ray* rays_dev, int n, pt_hit* hits_dev, int refill_lanes, int leaf_vote,
                                  int extend_blocks, hipStream_t s)
{
    if (!base || n <= 0 || n > kStaticMaxRays) return hipErrorInvalidValue;
    WfParams w{};
    float4* f = static_cast<float4*>(base);
    w.Sc.so = f; w.Sc.sd = f + kStaticMaxRays; w.hitS = f + 2 * (size_t)kStaticMaxRays;
    w.counters = reinterpret_cast<uint32_t*>(f + 3 * (size_t)kStaticMaxRays);
    w.n_slots = (n + kBlock - 1) / kBlock * kBlock;
    w.refill_lanes = refill_lanes; w.leaf_vote = leaf_vote; w.extend_blocks = extend_blocks;
    hipError_t e = hipMemsetAsync(w.counters, 0, kStaticCounterBytes, s);
    if (e != hipSuccess) return e;
    launch_wf_trace(w, S, rays_dev, n, hits_dev, s);
    return hipGetLastError();
}

// ---------------------------------------------------------------- launchers
// w.Pn: the path queue the first iteration will read
void launch_wf_generate(const WfParams& w, const DCamera& cam, DCounters* ctr, hipStream_t s)
{
    const uint32_t n_start = (uint32_t)w.n_slots < w.n_samples ? (uint32_t)w.n_slots : w.n_samples;
    dim3 grid((n_start + kBlock - 1) / kBlock), block(kBlock);
    if (ctr) hipLaunchKernelGGL(wf_generate_kernel<true>, grid, block, 0, s, w, cam, n_start, ctr);
    else hipLaunchKernelGGL(wf_generate_kernel<false>, grid, block, 0, s, w, cam, n_start, ctr);
}

// One iteration: the queues the last iteration appended to (w.Pc, w.Sc) are traced and consumed, w.Pn and w.Sn filled.
// ev: null, or kStageEvents events recorded before the extend launch and after each of the six stage kernels
void launch_wf_iteration(const WfParams& w, const DScene& S, const DCamera& cam, DCounters* ctr, hipStream_t s, hipEvent_t* ev)
{
    dim3 block(kBlock), stage(w.stage_blocks);
    hipLaunchKernelGGL(wf_plan_kernel, dim3(1), block, 0, s, w.counters, (const uint32_t*)w.wg_chunk, w.stage_blocks, w.n_samples, S.sky_mode == PT_SKY_SUNSKY ? 1 : 0);
    if (ev) (void)hipEventRecord(ev[0], s);
    // persistent waves: enough workgroups to fill every CU, each pulls rays until both queues are drained
    int ext_blocks = (w.n_slots + kBlock - 1) / kBlock;
    if (ext_blocks > w.extend_blocks) ext_blocks = w.extend_blocks;
    if (ctr && w.count_wide && S.wide) hipLaunchKernelGGL((wf_extend_kernel<true, true>), dim3(ext_blocks), block, 0, s, w, S, ctr);
    else if (ctr) hipLaunchKernelGGL((wf_extend_kernel<true, false>), dim3(ext_blocks), block, 0, s, w, S, ctr);
    else if (S.wide) hipLaunchKernelGGL((wf_extend_kernel<false, true>), dim3(ext_blocks), block, 0, s, w, S, ctr);
    else hipLaunchKernelGGL((wf_extend_kernel<false, false>), dim3(ext_blocks), block, 0, s, w, S, ctr);
    if (ev) (void)hipEventRecord(ev[1], s);
#define PT_STAGE(kernel, k, ...) do { \
        if (ctr) hipLaunchKernelGGL(kernel<true>, stage, block, 0, s, __VA_ARGS__); else hipLaunchKernelGGL(kernel<false>, stage, block, 0, s, __VA_ARGS__); \
        if (ev) (void)hipEventRecord(ev[k], s); } while (0)
    // path hits (timed in the place of the stage's first half; the second half's slot stays zero)
    {
        const bool direct = S.sky_mode == PT_SKY_SUNSKY;
        const bool tables = S.n_mats <= kLocalMats && S.n_tex <= kLocalTex && PT_PATHHIT_TABLES;
#define PT_PATHHIT(C, T, D) hipLaunchKernelGGL((wf_pathhit_kernel<C, T, D>), stage, block, 0, s, w, S, cam, ctr)
        if (direct) { if (tables) { if (ctr) PT_PATHHIT(true, true, true); else PT_PATHHIT(false, true, true); } else { if (ctr) PT_PATHHIT(true, false, true); else PT_PATHHIT(false, false, true); } }
        else { if (tables) { if (ctr) PT_PATHHIT(true, true, false); else PT_PATHHIT(false, true, false); } else { if (ctr) PT_PATHHIT(true, false, false); else PT_PATHHIT(false, false, false); } }
#undef PT_PATHHIT
        if (ev) (void)hipEventRecord(ev[2], s);
    }
    if (ev) (void)hipEventRecord(ev[3], s);
    PT_STAGE(wf_miss_kernel, 4, w, S, cam, ctr);
    if (PT_FUSED_LIGHT) {
        PT_STAGE(wf_light_kernel, 5, w, S, cam, ctr);
        if (ev) (void)hipEventRecord(ev[6], s);
    } else {
        PT_STAGE(wf_shadow_kernel, 5, w, S, cam, ctr);
        PT_STAGE(wf_bsdf_kernel, 6, w, S, cam, ctr);
    }
#undef PT_STAGE
}

// w.Pn: the path queue the next iteration will read (its tail word is still open)
void launch_wf_flush_leftover(const WfParams& w, const DCamera& cam, DCounters* ctr, hipStream_t s)
{
    if (ctr) hipLaunchKernelGGL(wf_flush_leftover_kernel<true>, dim3(w.stage_blocks), dim3(kBlock), 0, s, w, cam, ctr);
    else hipLaunchKernelGGL(wf_flush_leftover_kernel<false>, dim3(w.stage_blocks), dim3(kBlock), 0, s, w, cam, ctr);
}

// w as for the iteration that would come next; max_paths: an upper bound of the records in both queues
void launch_wf_drain(const WfParams& w, const DScene& S, const DCamera& cam, int max_paths, DCounters* ctr, hipStream_t s)
{
    hipLaunchKernelGGL(wf_plan_kernel, dim3(1), dim3(kBlock), 0, s, w.counters, (const uint32_t*)w.wg_chunk, w.stage_blocks, w.n_samples, S.sky_mode == PT_SKY_SUNSKY ? 1 : 0);
    // blocks of both queues at most: a stripe's blocks are kStripes apart, and in the worst case one stripe of each
    // queue holds every record
    int max_blocks = (max_paths / kBlock + 2) * kStripes;
    if (ctr && w.count_wide && S.wide) hipLaunchKernelGGL((wf_drain_kernel<true, true>), dim3(max_blocks), dim3(kBlock), 0, s, w, S, cam, ctr);
    else if (ctr) hipLaunchKernelGGL((wf_drain_kernel<true, false>), dim3(max_blocks), dim3(kBlock), 0, s, w, S, cam, ctr);
    else if (S.wide) hipLaunchKernelGGL((wf_drain_kernel<false, true>), dim3(max_blocks), dim3(kBlock), 0, s, w, S, cam, ctr);
    else hipLaunchKernelGGL((wf_drain_kernel<false, false>), dim3(max_blocks), dim3(kBlock), 0, s, w, S, cam, ctr);
}

// adds frames [first, first + count) of the lane's staged chunk to its pixels
void launch_wf_accumulate(const WfParams& w, const DCamera& cam, float* accum_dev, int first, int count, hipStream_t s)
{
    hipLaunchKernelGGL(wf_accumulate_kernel, dim3((w.n_pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, w, cam, accum_dev, first, count);
}

} // namespace ptamd
