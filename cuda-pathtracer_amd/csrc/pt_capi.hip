// C-ABI of the render path (include/pt_capi.h): context, uploads, BVH flattening, launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <atomic>
#include <vector>
#include <functional>
#include <new>
#include <chrono>
#include <string>
#include <thread>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "../../include/pt_capi.h"
#include "../../include/pt_detmath.h"
#include "pt_internal.h"
#include "pt_layout.h"
#include "pt_launch.h"

namespace ptamd { const char* last_error(); }
using namespace ptamd;

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return PT_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

// Host loops over n independent items in slices, one thread each (at most 16, the CPU share of one GPU on a
// shared machine; PT_HOST_THREADS overrides), inline when the loop is short. f(first, last, slice index).
static int host_threads()
{
    static const int n = [] {
        const char* e = getenv("PT_HOST_THREADS");
        int v = e ? atoi(e) : 0;
        if (v <= 0) { v = (int)std::thread::hardware_concurrency(); if (v > 16) v = 16; }
        return v < 1 ? 1 : v;
    }();
    return n;
}
template <typename F>
static void par_for(long long n, long long serial_below, F f)
{
    const int nt = n < serial_below ? 1 : host_threads();
    if (nt <= 1) { f(0LL, n, 0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back([=, &f] { f(n * t / nt, n * (t + 1) / nt, t); });
    f(0LL, n / nt, 0);
    for (std::thread& x : th) x.join();
}

// Device allocation that is released on every way out of a function.
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct pt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int W = 0, H = 0;
    float* accum = nullptr;         // 3*W*H, the buffer rendered into
    float* own_accum = nullptr;     // allocated by pt_init
    uint8_t* display = nullptr;     // 4*W*H
    // pt_render_frame_async: two display buffers on the device, two pinned ones on the host, a stream for the copies
    struct AsyncFrames {
        uint8_t* dev[2] = {nullptr, nullptr};
        uint8_t* host[2] = {nullptr, nullptr};
        hipEvent_t resolved[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
        bool pending[2] = {false, false};       // copied or being copied, not fetched yet
        int frame[2] = {0, 0};
        int next = 0, oldest = 0;
        hipStream_t copy_stream = nullptr;
        size_t bytes = 0;
    } af;
    // scene, host copies
    std::vector<pt_triangle> tris;
    std::vector<int> tri_idx;
    // scene, device
    pt_triangle* d_tri_ref = nullptr;
    uint32_t* d_tri_info = nullptr;         // per triangle: material index and its material's opaque flag (rebuilt when triangles or materials change)
    std::vector<uint8_t> mat_opaque;
    bool tri_flags_stale = true;
    DTri* d_tris = nullptr;
    DNode* d_nodes = nullptr;
    DNodeQ* d_qnodes = nullptr;
    DNode4* d_nodes4 = nullptr;
    bool wide_fits = false;
    int node_width = 4;
    int drain_rays = 1 << 18;
    int leaf_merge = 2;
    DMat* d_mats = nullptr;
    std::vector<uint8_t> tex_alpha_one;     // per uploaded texture: four components and alpha exactly 1 in every texel
    DTex* d_tex = nullptr;
    std::vector<DTex> tex_table;
    std::vector<void*> tex_data;
    bool pack_textures = true;
    float* d_env = nullptr;
    float* d_cdf = nullptr;
    float* d_env_tables = nullptr;   // row table [H] followed by column table [H*W]
    uint32_t* d_env_guides = nullptr; // row guide, column guides, one word of verification result
    uint2* d_env_texel_guide = nullptr; // table over the whole range of xi, one word of verification result
    float4* d_env_sample_table = nullptr; // per texel: what a light sample that lands on it returns
    bool env_guides_ok = false, use_env_guides = true, env_texel_guide_ok = false, use_env_texel_guide = true;
    DScene scene{};
    bool have_tris = false, have_bvh = false, have_mats = false, have_env = false;
    int max_bounces = 512;
    int counting = 0;               // 0 off, 1 the reference-order walk, 2 the production walk's own work
    DCounters* d_ctr = nullptr;
    pt_counters totals{};
    size_t n_nodes = 0, n_nodes4 = 0;
    // wavefront pool
    int mode = PT_MODE_WAVEFRONT;
    // Samples in flight at most: 456 bytes of queue arrays each, allocated for what a render needs (never more than its
    // samples). 2^27 holds a whole 64-spp 1080p chunk (61 GB of the 288): no sample waits for a slot, and the stage
    // loop takes 28 iterations instead of 38. Bench: 2^24 -6 %, 2^25 -4.4 %, 2^26 -2.6 %.
    int pool_slots = 1 << 27;
    int chunk_frames = 512;         // frames per staging chunk: every chunk pays one ramp-up and one run-down of the pool
    int refill_lanes = 8, leaf_vote = 24, n_cus = 256;
    int extend_blocks_per_cu = 0;   // 0: 5 when one lane runs (what its registers allow), 4 when two do
    // The wavefront loop can run as two independent lanes, each with its own sample queues, staging array,
    // counters and stream, on disjoint pixels (alternate blocks of 64): while one lane's traversal kernel (ALU bound)
    // runs, the other lane's shading kernels (memory bound) can share the GPU. One lane is the default: since the
    // traversal kernel fits five workgroups per CU it fills the chip alone, and two lanes are 1-2 % slower on the
    // bench scenes (pt_set_lanes selects two).
    struct WfLane {
        hipStream_t stream = nullptr;   // lane 0 runs on the context's stream, the others on their own
        void* block = nullptr;          // one allocation holding every slot array
        size_t block_bytes = 0;
        float* staging[2] = {nullptr, nullptr};     // two sample arrays: frames staged ahead are consumed from one while the next batch fills the other
        size_t staging_bytes[2] = {0, 0};
        uint32_t* counters = nullptr;
        uint32_t* host_counters = nullptr;   // pinned
        hipEvent_t st_ev[8 * kStageEvents] = {};    // per iteration of a batch: before extend, after each stage kernel
        hipEvent_t ready = nullptr;     // the last issued batch and its counter copy are done
        WfParams w{};
        int batch = 8;
        bool finished = true;
        WfPathQueue path_q[2];          // the two buffers of each queue: an iteration reads one and appends to the other
        WfShadowQueue shadow_q[2];
        int parity = 0;                 // which buffers the next iteration reads
        bool all_started = false;       // the host has seen every sample id handed out to some workgroup
        bool flushed = false;           // the ids workgroups still held then have been started (or there were none)
        uint64_t issued = 0, issue_cap = 0;   // stage iterations issued for the current chunk, and the most it can need
    };
    static constexpr int kMaxLanes = 2;
    WfLane lanes[kMaxLanes];
    int n_lanes = 1;
    bool lanes_forced = false;      // pt_set_lanes was called: use that count whatever the render size
    hipEvent_t ev_fork = nullptr;
    // pt_render_frame renders a few frames ahead into the staging arrays while the caller keeps asking for the next
    // frame of the same view; a staged frame is added to the accumulator when its call comes.
    uint64_t epoch = 0;             // bumped by every call that changes what a sample evaluates to
    int lookahead_max = 32;         // frames rendered per batch at most (1: no look-ahead)
    struct Ahead {
        bool valid = false;
        uint64_t epoch = 0;
        pt_params params{};         // the view the batch was rendered for
        int first = 0, count = 0, next = 0, lanes = 0;
        int streak = 0;             // consecutive calls that continued the previous one
        int last_frame = -2;
        int buf = 0;                // which staging arrays hold the batch being handed out
        WfParams w[2];              // that batch's lanes (pixels, staging array)
        // The batch after it is rendered meanwhile by a worker thread into the other staging arrays, on streams of its
        // own, so that no call pays for a batch: the caller's stream only ever runs one frame's addition, resolve and copy.
        std::thread worker;
        bool pending = false;       // a worker was started for [p_first, p_first + p_count)
        int p_first = 0, p_count = 0, p_lanes = 0, p_rc = 0;
        std::string p_error;
        WfParams p_w[2];
        hipStream_t stream0 = nullptr;      // stands in for the context's stream in the worker's renders
        hipEvent_t staged = nullptr;        // recorded by the worker behind its batch
        hipEvent_t consumed = nullptr;      // recorded on the context's stream behind the last addition from the arrays the worker fills next
    } ahead;
    void* static_scratch = nullptr; // this device's copy of the library's built-in block (pt_trace_closest_noalloc_dev)
    hipError_t static_scratch_error = hipSuccess;
    uint64_t wf_iterations = 0, wf_drains = 0;
    bool stage_timing = false;
    double extend_ms = 0.0;
    uint64_t extend_launches = 0;
    double stage_ms[PT_STAGE_KERNELS] = {};     // extend, resolve, vertex, miss, shadow, bsdf
};

// Ends the look-ahead of pt_render_frame: waits for the worker thread that renders the next batch (and for the work it
// queued), and drops what is staged. Every entry point that changes the scene, the settings or the pool, or renders
// by itself, starts with it; the frame-loop calls and the read-only image getters do not.
static void ahead_quiesce(pt_ctx* c)
{
    pt_ctx::Ahead& A = c->ahead;
    if (A.worker.joinable()) {
        A.worker.join();
        (void)hipSetDevice(c->device);
        if (A.stream0) (void)hipStreamSynchronize(A.stream0);
        for (int i = 1; i < pt_ctx::kMaxLanes; i++) if (c->lanes[i].stream) (void)hipStreamSynchronize(c->lanes[i].stream);
    }
    A.pending = false;
    A.valid = false;
}
// the first line of every entry point after which frames staged ahead would be stale
static void state_change(pt_ctx* c) { ahead_quiesce(c); c->epoch++; }

extern "C" {

const char* pt_last_error(void) { return ptamd::last_error(); }

int pt_create(pt_ctx** out, int device_id)
{
    if (!out) { set_error("pt_create: null output"); return PT_ERR_ARG; }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("pt_create: no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return PT_ERR_HIP;
    }
    if (device_id < 0 || device_id >= count) { set_error("pt_create: device %d out of range (%d devices)", device_id, count); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(device_id));
    pt_ctx* c = new (std::nothrow) pt_ctx();
    if (!c) { set_error("out of memory"); return PT_ERR_NOMEM; }
    c->device = device_id;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount; }
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    HIP_TRY(hipMalloc((void**)&c->d_ctr, sizeof(DCounters)));
    HIP_TRY(hipMemset(c->d_ctr, 0, sizeof(DCounters)));
    // the address of the built-in block on this device, looked up here so that the entry point that uses it
    // has no set-up left to do; a failure is reported by that entry point, everything else works without it
    c->static_scratch_error = wf_static_scratch(device_id, &c->static_scratch);
    c->scene.sky_mode = PT_SKY_ENVMAP;
    c->scene.sun[0] = 0.f; c->scene.sun[1] = 1.f; c->scene.sun[2] = 0.f;
    *out = c;
    return PT_OK;
}

static void free_async_frames(pt_ctx* c)
{
    pt_ctx::AsyncFrames& A = c->af;
    if (A.copy_stream) (void)hipStreamSynchronize(A.copy_stream);
    for (int b = 0; b < 2; b++) {
        if (A.dev[b]) (void)hipFree(A.dev[b]);
        if (A.host[b]) (void)hipHostFree(A.host[b]);
        if (A.resolved[b]) (void)hipEventDestroy(A.resolved[b]);
        if (A.copied[b]) (void)hipEventDestroy(A.copied[b]);
    }
    if (A.copy_stream) (void)hipStreamDestroy(A.copy_stream);
    A = pt_ctx::AsyncFrames{};
}

static void free_scene_geometry(pt_ctx* c)
{
    if (c->d_tris) { (void)hipFree(c->d_tris); c->d_tris = nullptr; }
    if (c->d_nodes) { (void)hipFree(c->d_nodes); c->d_nodes = nullptr; }
    if (c->d_qnodes) { (void)hipFree(c->d_qnodes); c->d_qnodes = nullptr; }
    if (c->d_nodes4) { (void)hipFree(c->d_nodes4); c->d_nodes4 = nullptr; }
    c->have_bvh = false;
}

void pt_destroy(pt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    ahead_quiesce(c);
    if (c->ahead.stream0) (void)hipStreamDestroy(c->ahead.stream0);
    if (c->ahead.staged) (void)hipEventDestroy(c->ahead.staged);
    if (c->ahead.consumed) (void)hipEventDestroy(c->ahead.consumed);
    (void)hipStreamSynchronize(c->stream);
    free_scene_geometry(c);
    if (c->d_tri_ref) (void)hipFree(c->d_tri_ref);
    if (c->d_tri_info) (void)hipFree(c->d_tri_info);
    if (c->d_mats) (void)hipFree(c->d_mats);
    if (c->d_tex) (void)hipFree(c->d_tex);
    for (void* p : c->tex_data) (void)hipFree(p);
    if (c->d_env) (void)hipFree(c->d_env);
    if (c->d_cdf) (void)hipFree(c->d_cdf);
    if (c->d_env_tables) (void)hipFree(c->d_env_tables);
    if (c->d_env_guides) (void)hipFree(c->d_env_guides);
    if (c->d_env_texel_guide) (void)hipFree(c->d_env_texel_guide);
    if (c->d_env_sample_table) (void)hipFree(c->d_env_sample_table);
    if (c->own_accum) (void)hipFree(c->own_accum);
    if (c->display) (void)hipFree(c->display);
    free_async_frames(c);
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    for (int i = 0; i < pt_ctx::kMaxLanes; i++) {
        pt_ctx::WfLane& L = c->lanes[i];
        for (hipEvent_t e : L.st_ev) if (e) (void)hipEventDestroy(e);
        if (L.ready) (void)hipEventDestroy(L.ready);
        if (L.block) (void)hipFree(L.block);
        for (float* p : L.staging) if (p) (void)hipFree(p);
        if (L.counters) (void)hipFree(L.counters);
        if (L.host_counters) (void)hipHostFree(L.host_counters);
        if (i > 0 && L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int pt_set_stream(pt_ctx* c, void* s)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return PT_OK;
}

int pt_init(pt_ctx* c, const pt_params* p)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || !p) { set_error("pt_init: null argument"); return PT_ERR_ARG; }
    int W = p->windowSize.x, H = p->windowSize.y;
    if (W <= 0 || H <= 0 || (int64_t)W * H > (1 << 28)) { set_error("pt_init: bad window size %dx%d", W, H); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    if (c->own_accum) { (void)hipFree(c->own_accum); c->own_accum = nullptr; c->accum = nullptr; }
    if (c->display) { (void)hipFree(c->display); c->display = nullptr; }
    free_async_frames(c);
    HIP_TRY(hipMalloc((void**)&c->own_accum, (size_t)W * H * 3 * sizeof(float)));
    c->accum = c->own_accum;
    HIP_TRY(hipMalloc((void**)&c->display, (size_t)W * H * 4));
    HIP_TRY(hipMemsetAsync(c->accum, 0, (size_t)W * H * 3 * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->display, 0, (size_t)W * H * 4, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->W = W; c->H = H;
    return PT_OK;
}

int pt_clear_accum(pt_ctx* c)
{
    if (!c || !c->accum) { set_error("pt_clear_accum: not initialised"); return PT_ERR_STATE; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->accum, 0, (size_t)c->W * c->H * 3 * sizeof(float), c->stream));
    return PT_OK;
}

int pt_upload_triangles(pt_ctx* c, const int* tri_idx, const pt_triangle* tris, int n)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || !tris || n <= 0) { set_error("pt_upload_triangles: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    c->tris.assign(tris, tris + n);
    c->tri_idx.resize(n);
    for (int i = 0; i < n; i++) {
        int v = tri_idx ? tri_idx[i] : i;
        if (v < 0 || v >= n) { set_error("pt_upload_triangles: index %d out of range at %d", v, i); return PT_ERR_ARG; }
        c->tri_idx[i] = v;
    }
    if (c->d_tri_ref) { (void)hipFree(c->d_tri_ref); c->d_tri_ref = nullptr; }
    free_scene_geometry(c);
    HIP_TRY(hipMalloc((void**)&c->d_tri_ref, (size_t)n * sizeof(pt_triangle)));
    HIP_TRY(hipMemcpy(c->d_tri_ref, tris, (size_t)n * sizeof(pt_triangle), hipMemcpyHostToDevice));
    c->scene.tri_ref = c->d_tri_ref;
    c->scene.n_tris = n;
    c->have_tris = true;
    c->tri_flags_stale = true; c->scene.tri_info = nullptr;
    return PT_OK;
}

static int tex_slot(const pt_ctx* c, uint64_t handle, uint8_t flag, int comps, const char* what, int mat, int* out)
{
    *out = 0;
    if (!flag || handle == 0) return PT_OK;
    if (handle > c->tex_table.size()) { set_error("material %d: %s handle %llu is not an uploaded texture", mat, what, (unsigned long long)handle); return PT_ERR_ARG; }
    if (c->tex_table[handle - 1].comps != comps) { set_error("material %d: %s needs a %d-component texture", mat, what, comps); return PT_ERR_ARG; }
    *out = (int)handle;
    return PT_OK;
}

int pt_upload_materials(pt_ctx* c, const pt_material* mats, int n)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || !mats || n <= 0) { set_error("pt_upload_materials: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    std::vector<DMat> dm(n);
    for (int i = 0; i < n; i++) {
        const pt_material& m = mats[i];
        DMat& d = dm[i];
        memset(&d, 0, sizeof(d));
        d.base[0] = m.baseColor.x; d.base[1] = m.baseColor.y; d.base[2] = m.baseColor.z;
        d.emis[0] = m.emissive.x; d.emis[1] = m.emissive.y; d.emis[2] = m.emissive.z;
        d.rough = m.roughness; d.aniso = m.anisotropic; d.metal = m.metallic; d.subsurf = m.subsurface; d.sheen = m.sheen;
        d.coat = m.clearcoat; d.coat_rough = m.clearcoatRoughness; d.strans = m.specTrans; d.alpha = m.alpha; d.ior = m.ior;
        d.G = m.medium.G;
        d.sigA[0] = m.medium.sigmaA.x; d.sigA[1] = m.medium.sigmaA.y; d.sigA[2] = m.medium.sigmaA.z;
        d.sigS[0] = m.medium.sigmaS.x; d.sigS[1] = m.medium.sigmaS.y; d.sigS[2] = m.medium.sigmaS.z;
        int rc;
        if ((rc = tex_slot(c, m.diffuseTexture, m.useTexture, 4, "map_Kd", i, &d.tex_diffuse)) != PT_OK) return rc;
        if ((rc = tex_slot(c, m.roughnessTexture, m.use_mapPr, 1, "map_Pr", i, &d.tex_rough)) != PT_OK) return rc;
        if ((rc = tex_slot(c, m.metallicTexture, m.use_mapPm, 1, "map_Pm", i, &d.tex_metal)) != PT_OK) return rc;
        if ((rc = tex_slot(c, m.emissiveTexture, m.use_mapKe, 1, "map_Ke", i, &d.tex_emis)) != PT_OK) return rc;
        if ((rc = tex_slot(c, m.normalTexture, m.use_mapNor, 4, "map_Bump", i, &d.tex_normal)) != PT_OK) return rc;
        // (a filtered lookup in a map whose alpha is 1 everywhere returns exactly 1: the weights of tex_mix are a and
        // 1 - a, whose rounded sum is 1 for every a in [0, 1])
        const bool alpha_blocks = d.tex_diffuse ? c->tex_alpha_one[(size_t)d.tex_diffuse - 1] != 0 : !(d.alpha < 1.0f);
        const bool never_refractive = !(d.strans > 0.f || d.strans < 0.f) || !(fabsf(d.ior - 1.f) <= 0.01f);
        d.opaque = alpha_blocks && never_refractive ? 1 : 0;
    }
    c->mat_opaque.resize((size_t)n);
    for (int i = 0; i < n; i++) c->mat_opaque[(size_t)i] = (uint8_t)dm[(size_t)i].opaque;
    c->tri_flags_stale = true; c->scene.tri_info = nullptr;
    if (c->have_tris) {
        for (size_t i = 0; i < c->tris.size(); i++)
            if (c->tris[i].matIndex < 0 || c->tris[i].matIndex >= n) { set_error("triangle %zu uses material %d of %d", i, c->tris[i].matIndex, n); return PT_ERR_ARG; }
    }
    if (c->d_mats) { (void)hipFree(c->d_mats); c->d_mats = nullptr; }
    HIP_TRY(hipMalloc((void**)&c->d_mats, (size_t)n * sizeof(DMat)));
    HIP_TRY(hipMemcpy(c->d_mats, dm.data(), (size_t)n * sizeof(DMat), hipMemcpyHostToDevice));
    if (c->d_tex) { (void)hipFree(c->d_tex); c->d_tex = nullptr; }
    if (!c->tex_table.empty()) {
        HIP_TRY(hipMalloc((void**)&c->d_tex, c->tex_table.size() * sizeof(DTex)));
        HIP_TRY(hipMemcpy(c->d_tex, c->tex_table.data(), c->tex_table.size() * sizeof(DTex), hipMemcpyHostToDevice));
    }
    c->scene.mats = c->d_mats;
    c->scene.tex = c->d_tex;
    c->scene.n_mats = n; c->scene.n_tex = (int)c->tex_table.size();
    c->have_mats = true;
    return PT_OK;
}

// Re-flattens the reference's 32-byte nodes (structs.cuh:130-137) into 64-byte two-box nodes and
// lays triangles out in leaf order. Interior nodes are numbered in depth-first order.
int pt_upload_bvh(pt_ctx* c, const pt_bvh_node* nodes, int root_idx, int nodes_used, int n_tri)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || !nodes || nodes_used <= 0) { set_error("pt_upload_bvh: bad argument"); return PT_ERR_ARG; }
    if (!c->have_tris || n_tri != (int)c->tris.size()) { set_error("pt_upload_bvh: upload %d triangles first", n_tri); return PT_ERR_STATE; }
    if (root_idx < 0 || root_idx >= nodes_used) { set_error("pt_upload_bvh: bad root"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    const bool timing = getenv("PT_BUILD_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto lap = [&](const char* what) { if (timing) { double t = now(); fprintf(stderr, "pt_upload_bvh: %-28s %.3f s\n", what, t - t_mark); t_mark = t; } };
    std::vector<DNode> dn;
    std::vector<DTri> dt;
    dn.reserve(nodes_used / 2 + 1);
    dt.reserve(n_tri);
    std::vector<char> seen(n_tri, 0);
    auto emit_leaf = [&](const pt_bvh_node& nd, int* ref) -> int {
        if (nd.triCount <= 0 || nd.leftFirst < 0 || (int64_t)nd.leftFirst + nd.triCount > n_tri) { set_error("pt_upload_bvh: leaf range [%d,+%d) out of bounds", nd.leftFirst, nd.triCount); return PT_ERR_ARG; }
        int first = (int)dt.size();
        for (int i = 0; i < nd.triCount; i++) {
            int id = c->tri_idx[nd.leftFirst + i];
            const pt_triangle& T = c->tris[id];
            seen[id] = 1;
            DTri r;
            r.q0 = make_float4(T.a.x, T.a.y, T.a.z, __builtin_bit_cast(float, (uint32_t)id << 1));
            r.q1 = make_float4(T.b.x - T.a.x, T.b.y - T.a.y, T.b.z - T.a.z, __builtin_bit_cast(float, i == 0 ? nd.triCount : 0));
            r.q2 = make_float4(T.c.x - T.a.x, T.c.y - T.a.y, T.c.z - T.a.z, 0.f);
            dt.push_back(r);
        }
        *ref = ~first;
        return PT_OK;
    };
    const pt_bvh_node& root = nodes[root_idx];
    if (root.triCount > 0) {
        int ref;
        int rc = emit_leaf(root, &ref);
        if (rc != PT_OK) return rc;
        c->scene.root_ref = ref;
        dn.push_back(DNode{});     // keep the node buffer non-empty
    } else {
        struct Item { int src; int dst; };
        std::vector<Item> st;
        dn.push_back(DNode{});
        st.push_back(Item{root_idx, 0});
        c->scene.root_ref = 0;
        while (!st.empty()) {
            Item it = st.back(); st.pop_back();
            const pt_bvh_node& nd = nodes[it.src];
            int l = nd.leftFirst, r = nd.leftFirst + 1;
            if (l <= 0 || r >= nodes_used) { set_error("pt_upload_bvh: node %d has children out of range", it.src); return PT_ERR_ARG; }
            const pt_bvh_node& L = nodes[l];
            const pt_bvh_node& R = nodes[r];
            DNode d{};
            d.lmin[0] = L.aabbMin.x; d.lmin[1] = L.aabbMin.y; d.lmin[2] = L.aabbMin.z;
            d.lmax[0] = L.aabbMax.x; d.lmax[1] = L.aabbMax.y; d.lmax[2] = L.aabbMax.z;
            d.rmin[0] = R.aabbMin.x; d.rmin[1] = R.aabbMin.y; d.rmin[2] = R.aabbMin.z;
            d.rmax[0] = R.aabbMax.x; d.rmax[1] = R.aabbMax.y; d.rmax[2] = R.aabbMax.z;
            // left subtree first, so it is contiguous right after its parent
            int rc;
            if (L.triCount > 0) { if ((rc = emit_leaf(L, &d.left)) != PT_OK) return rc; }
            else { d.left = (int)dn.size(); dn.push_back(DNode{}); }
            if (R.triCount > 0) { if ((rc = emit_leaf(R, &d.right)) != PT_OK) return rc; }
            else { d.right = (int)dn.size(); dn.push_back(DNode{}); }
            if ((int)dn.size() > nodes_used) { set_error("pt_upload_bvh: cyclic node graph"); return PT_ERR_ARG; }
            if (R.triCount <= 0) st.push_back(Item{r, d.right});
            if (L.triCount <= 0) st.push_back(Item{l, d.left});
            dn[it.dst] = d;
        }
    }
    c->scene.root_min[0] = root.aabbMin.x; c->scene.root_min[1] = root.aabbMin.y; c->scene.root_min[2] = root.aabbMin.z;
    c->scene.root_max[0] = root.aabbMax.x; c->scene.root_max[1] = root.aabbMax.y; c->scene.root_max[2] = root.aabbMax.z;
    // 16-bit grid over the (padded) root box for the compact nodes
    lap("binary nodes, leaf order");
    std::vector<DNodeQ> qn(dn.size());
    std::function<uint16_t(float, int)> qlo_f, qhi_f;
    {
        double org[3] = {root.aabbMin.x, root.aabbMin.y, root.aabbMin.z};
        double ext[3] = {(double)root.aabbMax.x - root.aabbMin.x, (double)root.aabbMax.y - root.aabbMin.y, (double)root.aabbMax.z - root.aabbMin.z};
        double cell[3];
        for (int a = 0; a < 3; a++) {
            if (!(ext[a] > 0.0)) ext[a] = 1.0;
            cell[a] = ext[a] / 65531.0;           // cells 2..65533 span the box, two cells of slack on each side
            c->scene.q_origin[a] = (float)(org[a] - 2.0 * cell[a]);
            c->scene.q_cell[a] = (float)cell[a];
            c->scene.q_inv_cell[a] = (float)(1.0 / cell[a]);
        }
        auto qlo = [&](float v, int a) { double q = std::floor(((double)v - (double)c->scene.q_origin[a]) / (double)c->scene.q_cell[a]) - 2.0; return (uint16_t)(q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q)); };
        auto qhi = [&](float v, int a) { double q = std::ceil(((double)v - (double)c->scene.q_origin[a]) / (double)c->scene.q_cell[a]) + 2.0; return (uint16_t)(q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q)); };
        qlo_f = qlo; qhi_f = qhi;
        for (size_t i = 0; i < dn.size(); i++) {
            const DNode& s = dn[i];
            DNodeQ& t = qn[i];
            for (int a = 0; a < 3; a++) {
                t.q[a] = qlo(s.lmin[a], a); t.q[3 + a] = qhi(s.lmax[a], a);
                t.q[6 + a] = qlo(s.rmin[a], a); t.q[9 + a] = qhi(s.rmax[a], a);
            }
            t.left = s.left; t.right = s.right;
        }
    }
    lap("binary quantized nodes");
    // four-wide nodes: each binary node's children are replaced by their own children, largest box first,
    // until four are present; absorbed binary nodes disappear
    std::vector<DNode4> n4;
    int stack_need_binary = 0, stack_need_wide = 0;
    // The sweeps below run level by level over the binary tree, each level in parallel: parents have smaller
    // indices than their children, so one forward pass gives every node's depth.
    const int nb = c->scene.root_ref >= 0 ? (int)dn.size() : 0;
    std::vector<int> depth(nb, 0), level_start, by_level(nb);
    {
        int max_depth = 0;
        for (int i = 0; i < nb; i++) {
            const int dch = depth[i] + 1;
            if (dn[i].left >= 0) depth[dn[i].left] = dch;
            if (dn[i].right >= 0) depth[dn[i].right] = dch;
            if (depth[i] > max_depth) max_depth = depth[i];
        }
        stack_need_binary = nb ? max_depth + 1 : 0;         // one pending sibling per level of the binary walk
        level_start.assign((size_t)max_depth + 2, 0);
        for (int i = 0; i < nb; i++) level_start[(size_t)depth[i] + 1]++;
        for (size_t l = 1; l < level_start.size(); l++) level_start[l] += level_start[l - 1];
        std::vector<int> cursor(level_start.begin(), level_start.end() - 1);
        for (int i = 0; i < nb; i++) by_level[(size_t)cursor[depth[i]]++] = i;
    }
    // Leaves of the wide tree: a binary subtree with at most leaf_merge triangles becomes one leaf. Its triangle
    // records are contiguous because subtrees are emitted whole. A wide leaf reference carries its own
    // count: ~(first | (count - 1) << 26).
    const int leaf_merge = c->leaf_merge;
    bool wide_ok = dt.size() < ((size_t)1 << 26);
    auto wide_leaf = [](int first, int count) { return ~(int)((uint32_t)first | ((uint32_t)(count - 1) << 26)); };
    std::vector<int> sub_first(dn.size(), 0), sub_count(dn.size(), 0);
    auto leaf_count = [&](int ref) { return __builtin_bit_cast(int, dt[(size_t)~ref].q1.w); };
    // A subtree becomes one leaf only where that saves work: its triangles' boxes must overlap so much that their
    // union is no larger (in surface area) than the two halves together, as for the two triangles of a quad. Unrelated
    // neighbours (a soup of small triangles) keep a box and a test each.
    std::vector<char> mergeable(dn.size(), 0);
    std::atomic<bool> big_leaf{false};       // set from the worker threads of the sweeps below
    auto wide_ref = [&](int ref) -> int {       // binary reference -> wide reference (interior stays >= 0, to be renumbered)
        if (ref < 0) { int n = leaf_count(ref); if (n > 16) big_leaf = true; return wide_leaf(~ref, n > 16 ? 16 : n); }
        if (mergeable[ref] && sub_count[ref] <= leaf_merge && sub_count[ref] <= 16) return wide_leaf(sub_first[ref], sub_count[ref]);
        return ref;
    };
    // Tight bounds of the triangles below each child of every binary node (the boxes the uploaded tree carries may
    // be padded, as the reference builder's are by 0.1 per vertex, BVH_builder.cpp:19-24), and the safety margin the
    // wide boxes get instead: 2^-19 of the largest coordinate, a few times the rounding of the box and triangle tests.
    struct TightBox { float lo[3], hi[3]; };
    lap("depths");
    std::vector<TightBox> tight_l(dn.size()), tight_r(dn.size());
    double max_coord = 0.0;
    {
        auto leaf_bounds = [&](int ref) {
            TightBox b; for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
            const int first = ~ref, count = leaf_count(ref);
            for (int j = first; j < first + count; j++) {
                const pt_triangle& T = c->tris[__builtin_bit_cast(uint32_t, dt[(size_t)j].q0.w) >> 1];
                const float* v[3] = {&T.a.x, &T.b.x, &T.c.x};
                for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) { b.lo[a] = std::fmin(b.lo[a], v[k][a]); b.hi[a] = std::fmax(b.hi[a], v[k][a]); }
            }
            return b;
        };
        auto join = [](const TightBox& x, const TightBox& y) { TightBox b; for (int a = 0; a < 3; a++) { b.lo[a] = std::fmin(x.lo[a], y.lo[a]); b.hi[a] = std::fmax(x.hi[a], y.hi[a]); } return b; };
        auto half_area = [](const TightBox& b) { double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2]; return x * y + y * z + z * x; };
        for (int l = (int)level_start.size() - 2; l >= 0; l--) {        // deepest level first: children before parents
            const int lo = level_start[(size_t)l], hi = level_start[(size_t)l + 1];
            par_for(hi - lo, 4096, [&](long long q0, long long q1, int) {
            for (long long q = lo + q0; q < lo + q1; q++) {
                const int i = by_level[(size_t)q];
                const int L = dn[i].left, R = dn[i].right;
                int fl, cl, fr, cr;
                if (L < 0) { fl = ~L; cl = leaf_count(L); } else { fl = sub_first[L]; cl = sub_count[L]; }
                if (R < 0) { fr = ~R; cr = leaf_count(R); } else { fr = sub_first[R]; cr = sub_count[R]; }
                sub_first[i] = fl < fr ? fl : fr;
                sub_count[i] = cl + cr;
                tight_l[i] = L < 0 ? leaf_bounds(L) : join(tight_l[L], tight_r[L]);
                tight_r[i] = R < 0 ? leaf_bounds(R) : join(tight_l[R], tight_r[R]);
                const bool parts_ok = (L < 0 || mergeable[L]) && (R < 0 || mergeable[R]);
                mergeable[i] = parts_ok && half_area(join(tight_l[i], tight_r[i])) <= half_area(tight_l[i]) + half_area(tight_r[i]);
            }
            });
        }
        std::vector<double> part_max((size_t)host_threads(), 0.0);
        par_for((long long)c->tris.size(), 65536, [&](long long t0, long long t1, int slice) {
            double mx = 0.0;
            for (long long t = t0; t < t1; t++) {
                const pt_triangle& T = c->tris[(size_t)t];
                const float* v[3] = {&T.a.x, &T.b.x, &T.c.x};
                for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) { double m = std::fabs((double)v[k][a]); if (m > mx && m < 1e30) mx = m; }
            }
            part_max[(size_t)slice] = mx;
        });
        for (double m : part_max) if (m > max_coord) max_coord = m;
    }
    lap("tight bounds");
    const double margin = std::ldexp(max_coord > 0.0 ? max_coord : 1.0, -19);
    if (c->scene.root_ref >= 0 && wide_ref(0) >= 0) {
        // level by level from the root, every node of a level in parallel; the children of a node get consecutive
        // indices, handed out by a running sum over the level so that the numbering does not depend on thread timing
        struct Cand { int ref; const float* lo; const float* hi; double area; };      // ref: wide leaf (< 0) or binary interior index
        struct Item { int src, dst, need; };
        struct Made { DNode4 d; int kid[4]; int n_kids; int need; };
        auto area = [](const float* lo, const float* hi) { double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2]; return x * y + y * z + z * x; };
        std::vector<Item> items, next;
        std::vector<Made> made;
        std::vector<int> base;
        n4.push_back(DNode4{});
        items.push_back(Item{0, 0, 0});
        bool bad_box = false;
        while (!items.empty()) {
            const long long n_items = (long long)items.size();
            made.resize(items.size());
            std::vector<int> part_need((size_t)host_threads(), 0);
            std::vector<char> part_bad((size_t)host_threads(), 0);
            par_for(n_items, 2048, [&](long long q0, long long q1, int slice) {
            int level_need = 0;
            bool bad_box = false;
            for (long long q = q0; q < q1; q++) {
                const Item it = items[(size_t)q];
                Cand cand[4];
                int nc = 2;
                cand[0] = Cand{wide_ref(dn[it.src].left), tight_l[it.src].lo, tight_l[it.src].hi, 0.0};
                cand[1] = Cand{wide_ref(dn[it.src].right), tight_r[it.src].lo, tight_r[it.src].hi, 0.0};
                for (int k = 0; k < 2; k++) cand[k].area = area(cand[k].lo, cand[k].hi);
                while (nc < 4) {
                    int pick = -1; double best = -1.0;
                    for (int k = 0; k < nc; k++) if (cand[k].ref >= 0 && cand[k].area > best) { best = cand[k].area; pick = k; }
                    if (pick < 0) break;
                    const int src = cand[pick].ref;
                    cand[pick] = Cand{wide_ref(dn[src].left), tight_l[src].lo, tight_l[src].hi, area(tight_l[src].lo, tight_l[src].hi)};
                    cand[nc++] = Cand{wide_ref(dn[src].right), tight_r[src].lo, tight_r[src].hi, area(tight_r[src].lo, tight_r[src].hi)};
                }
                Made& m = made[(size_t)q];
                m.d = DNode4{};
                m.need = it.need + nc - 1;
                if (m.need > level_need) level_need = m.need;
                // the node's grid: origin at the lowest corner of the widened boxes (rounded down to single precision),
                // cell = the smallest power of two with which the boxes span at most 255 cells
                uint32_t exps = 0;
                for (int a = 0; a < 3; a++) {
                    double lo = INFINITY, hi = -INFINITY;
                    for (int k = 0; k < nc; k++) { lo = std::fmin(lo, (double)cand[k].lo[a] - margin); hi = std::fmax(hi, (double)cand[k].hi[a] + margin); }
                    if (!(lo > -1e30 && hi < 1e30)) { bad_box = true; lo = 0.0; hi = 1.0; }      // non-finite vertices: the binary walk takes the scene
                    float org = (float)lo;
                    if ((double)org > lo) org = std::nextafterf(org, -INFINITY);
                    int e = 0;
                    (void)std::frexp((hi - (double)org) / 255.0, &e);        // extent / 255 = f * 2^e with f in [0.5, 1): 2^e is the smallest cell that fits, or one too large
                    if (e > -126 && hi - (double)org <= 255.0 * std::ldexp(1.0, e - 1)) e--;
                    e = e < -126 ? -126 : (e > 127 ? 127 : e);
                    const double cell = std::ldexp(1.0, e);
                    m.d.w[a] = __builtin_bit_cast(uint32_t, org);
                    exps |= (uint32_t)(uint8_t)(int8_t)e << (8 * a);
                    for (int k = 0; k < nc; k++) {
                        double ql = std::floor(((double)cand[k].lo[a] - margin - (double)org) / cell), qh = std::ceil(((double)cand[k].hi[a] + margin - (double)org) / cell);
                        ql = ql < 0.0 ? 0.0 : (ql > 255.0 ? 255.0 : ql); qh = qh < 0.0 ? 0.0 : (qh > 255.0 ? 255.0 : qh);
                        m.d.w[4 + a] |= (uint32_t)ql << (8 * k);
                        m.d.w[7 + a] |= (uint32_t)qh << (8 * k);
                    }
                    // a child that does not exist gets an inverted slab (low plane 255, high plane 0): its exit never
                    // lies beyond its entry, so the walk drops it without looking at its reference
                    for (int k = nc; k < 4; k++) m.d.w[4 + a] |= 255u << (8 * k);
                }
                m.d.w[3] = exps;
                m.n_kids = 0;
                for (int k = 0; k < 4; k++) {
                    if (k >= nc) { m.d.ref[k] = (int)0x80000000; continue; }
                    if (cand[k].ref >= 0) { m.kid[m.n_kids++] = cand[k].ref; m.d.ref[k] = 0; }       // numbered below
                    else m.d.ref[k] = cand[k].ref;
                }
            }
            part_need[(size_t)slice] = level_need; part_bad[(size_t)slice] = bad_box ? 1 : 0;
            });
            for (int v : part_need) if (v > stack_need_wide) stack_need_wide = v;
            for (char v : part_bad) if (v) bad_box = true;
            base.resize(items.size());
            int total_kids = 0;
            for (size_t q = 0; q < items.size(); q++) { base[q] = (int)n4.size() + total_kids; total_kids += made[q].n_kids; }
            const size_t first_new = n4.size();
            n4.resize(first_new + (size_t)total_kids);
            next.resize((size_t)total_kids);
            par_for(n_items, 2048, [&](long long q0, long long q1, int) {
            for (long long q = q0; q < q1; q++) {
                Made& m = made[(size_t)q];
                int j = 0;
                for (int k = 0; k < 4; k++) {
                    if (m.d.ref[k] != 0 || j >= m.n_kids) continue;        // interior children were marked 0 (no other reference is 0: leaves are negative)
                    m.d.ref[k] = base[(size_t)q] + j;
                    next[(size_t)(base[(size_t)q] - (int)first_new + j)] = Item{m.kid[j], base[(size_t)q] + j, m.need};
                    j++;
                }
                n4[(size_t)items[(size_t)q].dst] = m.d;
            }
            });
            items.swap(next);
        }
        if (bad_box) wide_ok = false;
        c->scene.root_ref4 = 0;
    } else {
        n4.push_back(DNode4{});
        c->scene.root_ref4 = wide_ref(c->scene.root_ref >= 0 ? 0 : c->scene.root_ref);
    }
    if (big_leaf) wide_ok = false;
    lap("wide nodes");
    const int stack_capacity = 16 + 48      /* LDS entries + private spill entries of the traversal kernels */;
    if (stack_need_binary > stack_capacity) { set_error("pt_upload_bvh: tree depth %d exceeds the traversal stack of %d entries", stack_need_binary, stack_capacity); return PT_ERR_ARG; }
    c->wide_fits = wide_ok && stack_need_wide <= stack_capacity;
    c->scene.wide = (c->wide_fits && c->node_width == 4) ? 1 : 0;
    free_scene_geometry(c);
    HIP_TRY(hipMalloc((void**)&c->d_qnodes, qn.size() * sizeof(DNodeQ)));
    HIP_TRY(hipMalloc((void**)&c->d_nodes4, n4.size() * sizeof(DNode4)));
    HIP_TRY(hipMemcpy(c->d_nodes4, n4.data(), n4.size() * sizeof(DNode4), hipMemcpyHostToDevice));
    c->scene.nodes4 = c->d_nodes4;
    HIP_TRY(hipMemcpy(c->d_qnodes, qn.data(), qn.size() * sizeof(DNodeQ), hipMemcpyHostToDevice));
    c->scene.qnodes = c->d_qnodes;
    HIP_TRY(hipMalloc((void**)&c->d_nodes, dn.size() * sizeof(DNode)));
    HIP_TRY(hipMemcpy(c->d_nodes, dn.data(), dn.size() * sizeof(DNode), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&c->d_tris, (dt.size() + 1) * sizeof(DTri)));
    HIP_TRY(hipMemcpy(c->d_tris, dt.data(), dt.size() * sizeof(DTri), hipMemcpyHostToDevice));
    c->scene.nodes = c->d_nodes;
    c->scene.tris = c->d_tris;
    c->n_nodes = dn.size();
    c->n_nodes4 = n4.size();
    c->have_bvh = true;
    c->tri_flags_stale = true; c->scene.tri_info = nullptr;       // (the new records carry no opaque marks yet)
    lap("device copies");
    return PT_OK;
}

int pt_build_bvh(pt_ctx* c, int flavour)
{
    if (!c || !c->have_tris) { set_error("pt_build_bvh: upload triangles first"); return PT_ERR_STATE; }
    if (flavour != PT_BVH_REFERENCE && flavour != PT_BVH_GPU_LBVH && flavour != PT_BVH_GPU_LBVH_VIA_HOST && flavour != PT_BVH_GPU_PLOC) { set_error("pt_build_bvh: unknown flavour %d", flavour); return PT_ERR_ARG; }
    int n = (int)c->tris.size();
    if (flavour == PT_BVH_GPU_PLOC && (n < 64 || (size_t)n >= ((size_t)1 << 26))) flavour = PT_BVH_GPU_LBVH;       // tiny and huge scenes: the radix tree through the host path
    if ((flavour == PT_BVH_GPU_LBVH || flavour == PT_BVH_GPU_PLOC) && n >= 64 && (size_t)n < ((size_t)1 << 26)) {
        // tree and traversal layouts are made on the device; the host receives the leaf order and a few numbers
        c->epoch++;
        HIP_TRY(hipSetDevice(c->device));
        std::vector<int> new_idx((size_t)n);
        GpuScene g;
        const int stack_capacity = 16 + 48      /* LDS entries + private spill entries of the traversal kernels */;
        int rc = gpu_build_scene(c->d_tri_ref, n, c->leaf_merge, flavour == PT_BVH_GPU_PLOC, flavour == PT_BVH_GPU_PLOC, &g, new_idx.data(), c->stream);
        if (rc != PT_OK && flavour != PT_BVH_GPU_PLOC) return rc;
        if (flavour == PT_BVH_GPU_PLOC && (rc != PT_OK || g.stack_need_binary > stack_capacity)) {
            // an agglomerative tree can come out lopsided (clusters swallowing their neighbours one by one) or, for
            // chains of nearest neighbours, not finish within its round limit; the radix tree over the same order is
            // bounded by the key length and always ends
            if (rc == PT_OK) { (void)hipFree(g.nodes); (void)hipFree(g.qnodes); (void)hipFree(g.nodes4); (void)hipFree(g.tris); }
            g = GpuScene{};
            rc = gpu_build_scene(c->d_tri_ref, n, c->leaf_merge, false, false, &g, new_idx.data(), c->stream);
            if (rc != PT_OK) return rc;
        }
        if (g.stack_need_binary > stack_capacity) {
            (void)hipFree(g.nodes); (void)hipFree(g.qnodes); (void)hipFree(g.nodes4); (void)hipFree(g.tris);
            set_error("pt_build_bvh: tree depth %d exceeds the traversal stack of %d entries", g.stack_need_binary, stack_capacity);
            return PT_ERR_ARG;
        }
        free_scene_geometry(c);
        c->tri_idx.swap(new_idx);
        c->d_nodes = g.nodes; c->d_qnodes = g.qnodes; c->d_nodes4 = g.nodes4; c->d_tris = g.tris;
        c->scene.nodes = g.nodes; c->scene.qnodes = g.qnodes; c->scene.nodes4 = g.nodes4; c->scene.tris = g.tris;
        c->scene.root_ref = 0; c->scene.root_ref4 = 0;
        for (int a = 0; a < 3; a++) {
            c->scene.root_min[a] = g.root_min[a]; c->scene.root_max[a] = g.root_max[a];
            c->scene.q_origin[a] = g.q_origin[a]; c->scene.q_cell[a] = g.q_cell[a]; c->scene.q_inv_cell[a] = g.q_inv_cell[a];
        }
        c->wide_fits = !g.bad_box && g.stack_need_wide <= stack_capacity;
        c->scene.wide = (c->wide_fits && c->node_width == 4) ? 1 : 0;
        c->n_nodes = g.n_nodes; c->n_nodes4 = g.n_nodes4;
        c->have_bvh = true;
        c->tri_flags_stale = true; c->scene.tri_info = nullptr;   // (the new records carry no opaque marks yet)
        return PT_OK;
    }
    std::vector<pt_bvh_node> nodes((size_t)2 * n - 1 + 1);
    if (flavour != PT_BVH_REFERENCE) {
        // the device-built tree in the reference's node layout, taken in like an uploaded one
        HIP_TRY(hipSetDevice(c->device));
        DevBuf d_nodes, d_idx;
        HIP_TRY(d_nodes.alloc(((size_t)2 * n - 1) * sizeof(pt_bvh_node)));
        HIP_TRY(d_idx.alloc((size_t)n * sizeof(int)));
        int rc = gpu_build_lbvh(c->d_tri_ref, n, d_nodes.as<pt_bvh_node>(), d_idx.as<int>(), c->stream);
        if (rc != PT_OK) return rc;
        std::vector<int> new_idx((size_t)n);
        HIP_TRY(hipMemcpy(nodes.data(), d_nodes.p, ((size_t)2 * n - 1) * sizeof(pt_bvh_node), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(new_idx.data(), d_idx.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) if (new_idx[i] < 0 || new_idx[i] >= n) { set_error("pt_build_bvh: the device build returned a bad index"); return PT_ERR_HIP; }
        // the index array and the device tree change together or not at all
        c->tri_idx.swap(new_idx);
        rc = pt_upload_bvh(c, nodes.data(), 0, 2 * n - 1, n);
        if (rc != PT_OK) c->tri_idx.swap(new_idx);
        return rc;
    }
    std::vector<int> new_idx((size_t)n);
    for (int i = 0; i < n; i++) new_idx[i] = i;
    int used = 0;
    int rc = build_bvh_reference(c->tris.data(), new_idx.data(), n, nodes.data(), &used);
    if (rc != PT_OK) return rc;
    c->tri_idx.swap(new_idx);
    rc = pt_upload_bvh(c, nodes.data(), 0, used, n);
    if (rc != PT_OK) c->tri_idx.swap(new_idx);
    return rc;
}

int pt_upload_texture(pt_ctx* c, const float* texels, int w, int h, int comps, uint64_t* handle_out)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || !texels || !handle_out || w <= 0 || h <= 0 || (comps != 1 && comps != 4)) { set_error("pt_upload_texture: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    // 8-bit image data (every value exactly k/255) is kept as bytes: the sampler rebuilds the same floats
    size_t n = (size_t)w * h * comps;
    std::vector<uint8_t> packed;
    bool bytes_ok = c->pack_textures;
    if (bytes_ok) {
        packed.resize(n);
        for (size_t i = 0; i < n; i++) {
            float v = texels[i];
            if (!(v >= 0.f && v <= 1.f)) { bytes_ok = false; break; }
            int k = (int)(v * 255.0f + 0.5f);
            float back = (float)k / 255.0f;
            if (memcmp(&back, &v, sizeof(float)) != 0) { bytes_ok = false; break; }       // bit equality: -0 stays a float texture
            packed[i] = (uint8_t)k;
        }
    }
    void* d = nullptr;
    size_t bytes = bytes_ok ? n : n * sizeof(float);
    HIP_TRY(hipMalloc(&d, bytes));
    HIP_TRY(hipMemcpy(d, bytes_ok ? (const void*)packed.data() : (const void*)texels, bytes, hipMemcpyHostToDevice));
    c->tex_data.push_back(d);
    bool alpha_one = comps == 4;
    const float one = 1.0f;
    for (size_t i = 3; alpha_one && i < n; i += 4) alpha_one = memcmp(&texels[i], &one, sizeof(float)) == 0;
    c->tex_alpha_one.push_back(alpha_one ? 1 : 0);
    DTex t; t.texels = d; t.w = w; t.h = h; t.comps = comps; t.format = bytes_ok ? 1 : 0;
    c->tex_table.push_back(t);
    *handle_out = (uint64_t)c->tex_table.size();
    return PT_OK;
}

int pt_set_texture_packing(pt_ctx* c, int on)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    c->pack_textures = on != 0;
    return PT_OK;
}

// device evaluation of the byte -> float conversion of packed textures, all 256 values
int pt_unorm8_probe(pt_ctx* c, float* out256)
{
    if (!c || !out256) { set_error("pt_unorm8_probe: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d;
    HIP_TRY(d.alloc(256 * sizeof(float)));
    launch_unorm8_probe(d.as<float>(), c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out256, d.p, 256 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    hipError_t es = hipStreamSynchronize(c->stream);        // nothing may still use the buffer when it is freed
    HIP_TRY(e);
    HIP_TRY(es);
    return PT_OK;
}

int pt_upload_envmap(pt_ctx* c, const float* rgba, const float* cdf, int w, int h, float sum, pt_params* params_io)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || !rgba || !cdf || w <= 0 || h <= 0) { set_error("pt_upload_envmap: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    c->have_env = false;            // until the whole upload has succeeded; every buffer below belongs to the context
    if (c->d_env) { (void)hipFree(c->d_env); c->d_env = nullptr; }
    if (c->d_cdf) { (void)hipFree(c->d_cdf); c->d_cdf = nullptr; }
    HIP_TRY(hipMalloc((void**)&c->d_env, (size_t)w * h * 4 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c->d_cdf, (size_t)w * h * sizeof(float)));
    HIP_TRY(hipMemcpy(c->d_env, rgba, (size_t)w * h * 4 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_cdf, cdf, (size_t)w * h * sizeof(float), hipMemcpyHostToDevice));
    if (c->d_env_tables) { (void)hipFree(c->d_env_tables); c->d_env_tables = nullptr; }
    HIP_TRY(hipMalloc((void**)&c->d_env_tables, ((size_t)h + (size_t)w * h) * sizeof(float)));
    launch_env_tables(c->d_cdf, w, h, c->d_env_tables, c->d_env_tables + h, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->scene.env_row_table = c->d_env_tables; c->scene.env_col_table = c->d_env_tables + h;
    c->scene.env_rgba = c->d_env; c->scene.env_cdf = c->d_cdf;
    c->scene.env_w = w; c->scene.env_h = h; c->scene.env_sum = sum;
    // what a light sample returns, per texel (32 bytes each; maps beyond 2^26 texels compute it per sample)
    c->scene.env_sample_table = nullptr;
    if (c->d_env_sample_table) { (void)hipFree(c->d_env_sample_table); c->d_env_sample_table = nullptr; }
    if ((size_t)w * h <= ((size_t)1 << 26)) {
        HIP_TRY(hipMalloc((void**)&c->d_env_sample_table, (size_t)w * h * 2 * sizeof(float4)));
        launch_env_sample_table(c->scene, c->d_env_sample_table, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->use_env_guides) c->scene.env_sample_table = c->d_env_sample_table;
    }
    // guide tables for the two CDF searches, trusted only if they reproduce the bisection at every breakpoint
    c->scene.env_fast = 0; c->scene.env_fast2 = 0;
    c->env_guides_ok = false; c->env_texel_guide_ok = false;
    if (c->d_env_texel_guide) { (void)hipFree(c->d_env_texel_guide); c->d_env_texel_guide = nullptr; }
    if (c->d_env_guides) { (void)hipFree(c->d_env_guides); c->d_env_guides = nullptr; }
    if (sum > 0.f && sum < 1e30f && w <= 65535 && h <= 65535) {
        size_t words = (size_t)kEnvRowBuckets + (size_t)h * kEnvColBuckets + 1;
        HIP_TRY(hipMalloc((void**)&c->d_env_guides, words * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(c->d_env_guides, 0, words * sizeof(uint32_t), c->stream));
        c->scene.env_row_guide = c->d_env_guides;
        c->scene.env_col_guide = c->d_env_guides + kEnvRowBuckets;
        c->scene.env_row_scale = (float)kEnvRowBuckets / sum;
        c->scene.env_col_scale = (float)kEnvColBuckets / sum;
        uint32_t* d_bad = c->d_env_guides + words - 1;
        launch_env_guides(c->scene, c->d_env_guides, c->d_env_guides + kEnvRowBuckets, d_bad, c->stream);
        HIP_TRY(hipGetLastError());
        uint32_t bad = 1;
        HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->env_guides_ok = bad == 0;
        c->scene.env_fast = (c->env_guides_ok && c->use_env_guides) ? 1 : 0;
        if (c->env_guides_ok) {
            // the table over the whole range of xi, checked the same way with the guides above as its fallback
            HIP_TRY(hipMalloc((void**)&c->d_env_texel_guide, (size_t)kEnvTexelBuckets * sizeof(uint2) + sizeof(uint32_t)));
            uint32_t* d_bad2 = (uint32_t*)(c->d_env_texel_guide + kEnvTexelBuckets);
            HIP_TRY(hipMemsetAsync(d_bad2, 0, sizeof(uint32_t), c->stream));
            DScene probe = c->scene;
            probe.env_fast = 1; probe.env_fast2 = 1;
            probe.env_texel_guide = c->d_env_texel_guide;
            probe.env_texel_scale = (float)kEnvTexelBuckets / sum;
            launch_env_texel_guide(probe, c->d_env_texel_guide, d_bad2, c->stream);
            HIP_TRY(hipGetLastError());
            uint32_t bad2 = 1;
            HIP_TRY(hipMemcpyAsync(&bad2, d_bad2, sizeof(bad2), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->env_texel_guide_ok = bad2 == 0;
            c->scene.env_texel_guide = c->d_env_texel_guide;
            c->scene.env_texel_scale = probe.env_texel_scale;
            c->scene.env_fast2 = (c->env_texel_guide_ok && c->use_env_guides && c->use_env_texel_guide) ? 1 : 0;
        }
    }
    c->have_env = true;
    if (params_io) {
        params_io->envMap_size.x = w; params_io->envMap_size.y = h; params_io->envmap_sum = sum;
        params_io->cubeMap = 1; params_io->envMap_cdf = 1;
    }
    return PT_OK;
}

int pt_set_env_guides(pt_ctx* c, int on, int* verified_out)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    c->use_env_guides = on != 0;
    c->use_env_texel_guide = on != 2;        // 2: row and column guides only
    c->scene.env_fast = (c->env_guides_ok && c->use_env_guides) ? 1 : 0;
    c->scene.env_fast2 = (c->env_texel_guide_ok && c->use_env_guides && c->use_env_texel_guide) ? 1 : 0;
    c->scene.env_sample_table = c->use_env_guides ? c->d_env_sample_table : nullptr;     // 0: every light sample is computed as the reference does
    if (verified_out) *verified_out = (c->env_guides_ok ? 1 : 0) | (c->env_texel_guide_ok ? 2 : 0);
    return PT_OK;
}

int pt_set_sky_mode(pt_ctx* c, int mode)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || (mode != PT_SKY_ENVMAP && mode != PT_SKY_SUNSKY)) { set_error("pt_set_sky_mode: bad argument"); return PT_ERR_ARG; }
    c->scene.sky_mode = mode;
    return PT_OK;
}

int pt_set_max_bounces(pt_ctx* c, int n)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || n < 0) { set_error("pt_set_max_bounces: bad argument"); return PT_ERR_ARG; }
    c->max_bounces = n;
    return PT_OK;
}

int pt_enable_counters(pt_ctx* c, int on)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    if (on < 0 || on > 2) { set_error("pt_enable_counters: 0, 1 or 2"); return PT_ERR_ARG; }
    c->counting = on;
    return PT_OK;
}

int pt_camera_basis(const pt_params* P, pt_float3* ww, pt_float3* uu, pt_float3* vv)
{
    if (!P || !ww || !uu || !vv) { set_error("pt_camera_basis: null argument"); return PT_ERR_ARG; }
    // pathtracer.cu:568-571 with the pinned sin/cos and rsqrtf = 1/sqrtf
    auto unit = [](float x, float y, float z, pt_float3* o) { float s = 1.0f / __builtin_sqrtf(x * x + y * y + z * z); o->x = x * s; o->y = y * s; o->z = z * s; };
    float ax = P->cameraAngle.x, ay = P->cameraAngle.y;
    float ox = P->rayOrigin.x, oy = P->rayOrigin.y, oz = P->rayOrigin.z;
    float tx = ox + pt_sinf(ax) * pt_cosf(ay), ty = oy + pt_sinf(ay), tz = oz + -pt_cosf(ax) * pt_cosf(ay);
    unit(tx - ox, ty - oy, tz - oz, ww);
    // cross(ww, (0,1,0)) and cross(uu, ww) component by component (float3_header.cuh:80-82)
    float cx = ww->y * 0.f - 1.f * ww->z, cy = ww->z * 0.f - 0.f * ww->x, cz = ww->x * 1.f - 0.f * ww->y;
    unit(cx, cy, cz, uu);
    float dx = uu->y * ww->z - ww->y * uu->z, dy = uu->z * ww->x - ww->z * uu->x, dz = uu->x * ww->y - ww->x * uu->y;
    unit(dx, dy, dz, vv);
    return PT_OK;
}

static int make_camera(pt_ctx* c, const pt_params* P, DCamera* cam)
{
    if (P->windowSize.x != c->W || P->windowSize.y != c->H) { set_error("window size %dx%d differs from pt_init's %dx%d", P->windowSize.x, P->windowSize.y, c->W, c->H); return PT_ERR_ARG; }
    pt_float3 ww, uu, vv;
    pt_camera_basis(P, &ww, &uu, &vv);
    cam->origin[0] = P->rayOrigin.x; cam->origin[1] = P->rayOrigin.y; cam->origin[2] = P->rayOrigin.z;
    cam->ww[0] = ww.x; cam->ww[1] = ww.y; cam->ww[2] = ww.z;
    cam->uu[0] = uu.x; cam->uu[1] = uu.y; cam->uu[2] = uu.z;
    cam->vv[0] = vv.x; cam->vv[1] = vv.y; cam->vv[2] = vv.z;
    cam->fov = P->fov; cam->focal = P->focalDistance; cam->dof = P->DOF_strenght;
    cam->W = c->W; cam->H = c->H;
    cam->is_rendering = P->isRendering ? 1 : 0;
    cam->max_bounces = c->max_bounces;
    c->scene.sun[0] = P->sunDirection.x; c->scene.sun[1] = P->sunDirection.y; c->scene.sun[2] = P->sunDirection.z;
    return PT_OK;
}

static int check_ready(pt_ctx* c, bool need_image)
{
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    if (!c->have_tris || !c->have_bvh) { set_error("scene not ready: upload triangles and a BVH first"); return PT_ERR_STATE; }
    if (need_image) {
        if (!c->accum) { set_error("pt_init has not been called"); return PT_ERR_STATE; }
        if (!c->have_mats) { set_error("scene not ready: upload materials first"); return PT_ERR_STATE; }
        if (c->scene.sky_mode == PT_SKY_ENVMAP && !c->have_env) { set_error("scene not ready: upload an environment map or select the sun/sky model"); return PT_ERR_STATE; }
    }
    if (need_image && c->tri_flags_stale) {
        // material index and opaque flag of every triangle, in triangle order (DScene::tri_info)
        std::vector<uint32_t> info(c->tris.size());
        for (size_t i = 0; i < c->tris.size(); i++) {
            const int m = c->tris[i].matIndex;
            const bool known = m >= 0 && (size_t)m < c->mat_opaque.size();
            info[i] = (uint32_t)(known ? m : 0) | ((known && c->mat_opaque[(size_t)m]) ? 0x80000000u : 0u);
        }
        HIP_TRY(hipSetDevice(c->device));
        if (c->d_tri_info) { (void)hipFree(c->d_tri_info); c->d_tri_info = nullptr; }
        HIP_TRY(hipMalloc((void**)&c->d_tri_info, info.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(c->d_tri_info, info.data(), info.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->scene.tri_info = c->d_tri_info;
        // the same flag into the leaf-order records the traversal kernel tests (DTri::q0.w)
        launch_mark_opaque_triangles(c->d_tris, (int)c->tris.size(), c->d_tri_info, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->tri_flags_stale = false;
    }
    return PT_OK;
}

static int fetch_counters(pt_ctx* c)
{
    DCounters h;
    HIP_TRY(hipMemcpyAsync(&h, c->d_ctr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->totals.rays += h.rays; c->totals.samples += h.samples; c->totals.path_rays += h.path_rays;
    const bool wide_walk = c->counting == 2 && c->scene.wide && c->mode == PT_MODE_WAVEFRONT;
    if (wide_walk) { c->totals.wide_nodes += h.box_tests; c->totals.wide_tris += h.tri_tests; }
    else { c->totals.box_tests += h.box_tests; c->totals.tri_tests += h.tri_tests; }
    c->totals.shade_vertices += h.shade_vertices; c->totals.env_samples += h.env_samples; c->totals.tex_taps += h.tex_taps; c->totals.tex_taps_r += h.tex_taps_r;
    HIP_TRY(hipMemsetAsync(c->d_ctr, 0, sizeof(DCounters), c->stream));
    return PT_OK;
}

static int lane_counters(pt_ctx::WfLane& L)
{
    if (!L.counters) {
        HIP_TRY(hipMalloc((void**)&L.counters, WF_CTR_WORDS * sizeof(uint32_t) + 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(L.counters, 0, WF_CTR_WORDS * sizeof(uint32_t) + 16 * sizeof(unsigned long long)));
        HIP_TRY(hipHostMalloc((void**)&L.host_counters, WF_CTR_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    }
    if (!L.ready) HIP_TRY(hipEventCreateWithFlags(&L.ready, hipEventDisableTiming));
    return PT_OK;
}

// Workgroups of a stage launch for a pool of `slots` samples in flight: enough to fill the GPU with the lightest
// stage kernel, and no more than the pool has 256-record segments.
#ifndef PT_STAGE_BLOCKS_PER_CU
#define PT_STAGE_BLOCKS_PER_CU 8
#endif
static int stage_blocks_for(const pt_ctx* c, size_t slots)
{
    size_t b = (size_t)PT_STAGE_BLOCKS_PER_CU * (size_t)c->n_cus, segs = slots / 256;
    if (segs < 1) segs = 1;
    return (int)(b < segs ? b : segs);
}

// Entries every queue array gets for `slots` samples in flight: the pool itself plus the sample ids the stage
// workgroups can hold in their chunks when the global hand-out runs dry (they are all started at once then). Both
// are dealt to the kStripes stripes of a queue in blocks of 256, in turn, and a sample stays in its stripe, so every
// stripe needs room for its share rounded up to whole blocks.
static size_t queue_capacity(const pt_ctx* c, size_t slots)
{
    const size_t per_stripe = ((slots + 255) / 256 + kStripes - 1) / kStripes + ((size_t)stage_blocks_for(c, slots) + kStripes - 1) / kStripes;
    return per_stripe * kStripes * 256;
}

// Bytes of a lane's queues: 28 arrays of 16 bytes per entry (two buffers each of the path queue's 3 and the shadow
// queue's 10, and the two hit arrays), the bsdf list, the miss list, and the workgroups' sample-id chunks.
static size_t pool_bytes(size_t capacity, int stage_blocks)
{
    size_t b = capacity * (28 * sizeof(float4) + 2 * sizeof(uint32_t)) + (size_t)stage_blocks * 2 * sizeof(uint32_t);
    return (b + 255) / 256 * 256;
}

// Points the lane's queue arrays into the block that starts at `base_ptr`.
static void pool_layout(pt_ctx::WfLane& L, void* base_ptr, size_t capacity, int stage_blocks, WfParams* w)
{
    float4* p = (float4*)base_ptr;
    auto take = [&]() { float4* r = p; p += capacity; return r; };
    for (int k = 0; k < 2; k++) {
        WfPathQueue& P = L.path_q[k];
        P.c0 = take(); P.c1 = take(); P.c2 = take();
        WfShadowQueue& S = L.shadow_q[k];
        S.so = take(); S.sd = take(); S.c0 = take(); S.c1 = take(); S.c2 = take(); S.d0 = take(); S.d1 = take();
        S.e0 = take(); S.e1 = take(); S.e4 = take();
    }
    w->hitP = take(); w->hitS = take();
    w->bsdf_list = (uint32_t*)p;
    w->miss_list = w->bsdf_list + capacity;
    w->wg_chunk = w->miss_list + capacity;
    w->counters = L.counters;
    w->capacity = (int)capacity;
    w->stage_blocks = stage_blocks;
    w->Pc = L.path_q[0]; w->Pn = L.path_q[1]; w->Sc = L.shadow_q[0]; w->Sn = L.shadow_q[1];
}

// Lays the queue arrays of a lane out in its block (grown if needed) for `slots` samples in flight.
static int lane_pool(pt_ctx* c, pt_ctx::WfLane& L, size_t slots, WfParams* w)
{
    int stage_blocks = stage_blocks_for(c, slots);
    size_t capacity = queue_capacity(c, slots);
    size_t need = pool_bytes(capacity, stage_blocks);
    while (need > L.block_bytes) {
        if (L.block) { (void)hipFree(L.block); L.block = nullptr; L.block_bytes = 0; }
        const hipError_t e = hipMalloc(&L.block, need);
        if (e == hipSuccess) { L.block_bytes = need; break; }
        L.block = nullptr;
        (void)hipGetLastError();
        // The device does not have that much memory free (the default asks for up to 61 GB): fewer samples in flight
        // change the number of stage iterations, never a result.
        if (e != hipErrorOutOfMemory || slots <= ((size_t)1 << 20)) { set_error("hipMalloc of %zu bytes for the sample queues: %s", need, hipGetErrorString(e)); return PT_ERR_HIP; }
        slots = (slots / 2 + 255) / 256 * 256;
        stage_blocks = stage_blocks_for(c, slots);
        capacity = queue_capacity(c, slots);
        need = pool_bytes(capacity, stage_blocks);
    }
    int rc = lane_counters(L);
    if (rc != PT_OK) return rc;
    pool_layout(L, L.block, capacity, stage_blocks, w);
    w->n_slots = (int)slots;
    w->refill_lanes = c->refill_lanes; w->leaf_vote = c->leaf_vote;
    w->count_wide = c->counting == 2 ? 1 : 0;
    w->extend_blocks = (c->extend_blocks_per_cu ? c->extend_blocks_per_cu : 5) * c->n_cus;
    return PT_OK;
}

// The lane's parameters as the iteration that comes next sees them: it reads the buffers `parity` names and
// appends to the other ones.
static WfParams lane_view(const pt_ctx::WfLane& L)
{
    WfParams v = L.w;
    v.Pc = L.path_q[L.parity]; v.Pn = L.path_q[L.parity ^ 1];
    v.Sc = L.shadow_q[L.parity]; v.Sn = L.shadow_q[L.parity ^ 1];
    return v;
}
// ... and as a kernel that appends to the queue that iteration will read sees them (the first samples, the
// leftover ids)
static WfParams lane_feed_view(const pt_ctx::WfLane& L)
{
    WfParams v = L.w;
    v.Pc = L.path_q[L.parity ^ 1]; v.Pn = L.path_q[L.parity];
    v.Sc = L.shadow_q[L.parity ^ 1]; v.Sn = L.shadow_q[L.parity];
    return v;
}

// Sizes a lane's sample queues and staging array for `frames` frames of the pixels dealt to lane `lane` of `n_lanes`.
static int wf_prepare(pt_ctx* c, pt_ctx::WfLane& L, int frames, int lane, int n_lanes, size_t pool_slots, int buf, WfParams* w)
{
    size_t npix = (size_t)c->W * c->H;
    size_t blocks = (npix + 63) / 64, rem = npix % 64;
    size_t mine = (blocks - (size_t)lane + (size_t)n_lanes - 1) / (size_t)n_lanes;       // blocks lane, lane + n_lanes, ...
    size_t n_pixels = mine * 64;
    if (rem != 0 && (blocks - 1) % (size_t)n_lanes == (size_t)lane) n_pixels -= 64 - rem;    // the last block is short and it is ours
    size_t n_samples = n_pixels * (size_t)frames;
    size_t slots = n_samples < pool_slots ? n_samples : pool_slots;
    slots = (slots + 255) / 256 * 256;
    int rc = lane_pool(c, L, slots, w);
    if (rc != PT_OK) return rc;
    if (!c->extend_blocks_per_cu) w->extend_blocks = (n_lanes > 1 ? 4 : 5) * c->n_cus;     // room for the other lane's kernels (3: +0.5 % time, 5: +0.2 %)
    size_t sbytes = n_samples * sizeof(float);
    if (sbytes > L.staging_bytes[buf]) {
        if (L.staging[buf]) { (void)hipFree(L.staging[buf]); L.staging[buf] = nullptr; L.staging_bytes[buf] = 0; }
        HIP_TRY(hipMalloc((void**)&L.staging[buf], sbytes));
        L.staging_bytes[buf] = sbytes;
    }
    w->staging = L.staging[buf];
    w->n_samples = (uint32_t)n_samples;
    w->n_pixels = (uint32_t)n_pixels;
    w->div_pixels = make_fast_div((uint32_t)n_pixels); w->div_tiles_x = make_fast_div((uint32_t)c->W >> 3); w->div_W = make_fast_div((uint32_t)c->W);
    w->px_lane = lane; w->px_lanes = n_lanes;
    w->W = c->W; w->H = c->H;
    w->tiled = (c->W % 8 == 0 && c->H % 8 == 0) ? 1 : 0;
    w->n_frames = frames;
    return PT_OK;
}

// The host looks at a lane's counters after every batch of iterations: 8 while new samples are still being
// started (a look costs a wait), 2 once all are in flight, so that the hand-over to the drain kernel comes
// soon after the pool has thinned out.
static const int kMaxBatch = 8;

static int lane_issue_batch(pt_ctx* c, pt_ctx::WfLane& L, const DCamera& cam, DCounters* ctr)
{
    if (c->stage_timing && !L.st_ev[0]) { for (int i = 0; i < kStageEvents * kMaxBatch; i++) HIP_TRY(hipEventCreate(&L.st_ev[i])); }
    if (L.all_started && !L.flushed) {
        // the global hand-out has run dry: start the ids workgroups still hold in their chunks, all at once
        launch_wf_flush_leftover(lane_feed_view(L), cam, ctr, L.stream);
        L.flushed = true;
    }
    for (int i = 0; i < L.batch; i++) {
        launch_wf_iteration(lane_view(L), c->scene, cam, ctr, L.stream, c->stage_timing ? L.st_ev + kStageEvents * i : nullptr);
        L.parity ^= 1;
    }
    HIP_TRY(hipGetLastError());
    c->wf_iterations += L.batch;
    L.issued += (uint64_t)L.batch;
    HIP_TRY(hipMemcpyAsync(L.host_counters, L.counters, WF_CTR_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipEventRecord(L.ready, L.stream));
    return PT_OK;
}

// Frames [f0, f1) through the stage kernels, in chunks that bound the staging memory. The lanes own disjoint
// pixels (blocks of 64 dealt in turn) and each adds its staged samples to its pixels frame by frame, so every
// pixel sees its additions in frame order whatever the lanes' timing was.
// Frames one staging chunk holds: pt_set_pool's chunk_frames, halved until a chunk is at most 2^30 samples
// (4 GiB of staging).
static int effective_chunk_frames(const pt_ctx* c)
{
    size_t npix = (size_t)c->W * c->H;
    int max_chunk = c->chunk_frames;
    while (max_chunk > 1 && npix * (size_t)max_chunk > ((size_t)1 << 30)) max_chunk /= 2;
    return max_chunk;
}

static int render_wavefront_chunks(pt_ctx* c, const DCamera& cam, int f0, int f1, bool hold, int* lanes_used, int buf, hipStream_t base);

// hold: leave the samples in the staging arrays instead of adding them to the accumulator (the caller adds
// frames one by one); the range must then fit one staging chunk, because every chunk overwrites the last.
// buf: which of the lanes' two staging arrays receives the samples; base: the stream that plays the context's stream
// (the first lane runs on it and the render is ordered behind it and ahead of it); null = the context's stream.
static int render_wavefront(pt_ctx* c, const DCamera& cam, int f0, int f1, bool hold = false, int* lanes_used = nullptr, int buf = 0, hipStream_t base = nullptr)
{
    if (hold && f1 - f0 > effective_chunk_frames(c)) { set_error("render_wavefront: %d held frames exceed the staging chunk of %d", f1 - f0, effective_chunk_frames(c)); return PT_ERR_ARG; }
    int rc = render_wavefront_chunks(c, cam, f0, f1, hold, lanes_used, buf, base ? base : c->stream);
    if (rc != PT_OK) {
        // kernels of either lane may still be queued on buffers a later call resizes or frees
        for (int i = 0; i < pt_ctx::kMaxLanes; i++) if (c->lanes[i].stream) (void)hipStreamSynchronize(c->lanes[i].stream);
    }
    return rc;
}

static int render_wavefront_chunks(pt_ctx* c, const DCamera& cam, int f0, int f1, bool hold, int* lanes_used, int buf, hipStream_t base)
{
    size_t npix = (size_t)c->W * c->H;
    const int max_chunk = effective_chunk_frames(c);
    DCounters* ctr = c->counting ? c->d_ctr : nullptr;
    if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    c->lanes[0].stream = base;
    for (int i = 1; i < pt_ctx::kMaxLanes; i++)
        if (!c->lanes[i].stream) HIP_TRY(hipStreamCreateWithFlags(&c->lanes[i].stream, hipStreamNonBlocking));
    size_t pixel_blocks = (npix + 63) / 64;
    for (int b = f0; b < f1; b += max_chunk) {
        int frames = (f1 - b) < max_chunk ? (f1 - b) : max_chunk;
        // two lanes, when asked for through the default count, only from about three million samples on (a 1080p
        // frame is faster on one lane); the split never changes a result, so small renders may use any count
        int n_lanes = (size_t)c->n_lanes < pixel_blocks ? c->n_lanes : (int)pixel_blocks;
        if (!c->lanes_forced && npix * (size_t)frames < ((size_t)3 << 20)) n_lanes = 1;
        size_t lane_pool_slots = (size_t)c->pool_slots / (size_t)n_lanes;
        if (lane_pool_slots < 256) lane_pool_slots = 256;
        // the other lanes start after everything queued on the context's stream so far (uploads, earlier renders)
        HIP_TRY(hipEventRecord(c->ev_fork, base));
        for (int i = 0; i < n_lanes; i++) {
            pt_ctx::WfLane& L = c->lanes[i];
            if (i > 0) HIP_TRY(hipStreamWaitEvent(L.stream, c->ev_fork, 0));
            int rc = wf_prepare(c, L, frames, i, n_lanes, lane_pool_slots, buf, &L.w);
            if (rc != PT_OK) return rc;
            L.w.frame_begin = b;
            HIP_TRY(hipMemsetAsync(L.counters, 0, WF_CTR_WORDS * sizeof(uint32_t), L.stream));
            HIP_TRY(hipMemsetAsync(L.w.wg_chunk, 0, (size_t)L.w.stage_blocks * 2 * sizeof(uint32_t), L.stream));
            L.parity = 0;
            launch_wf_generate(lane_feed_view(L), cam, ctr, L.stream);
            HIP_TRY(hipGetLastError());
            // a pool that holds every sample of the chunk starts them all here: nothing is left to hand out
            const bool all_at_once = (size_t)L.w.n_slots >= (size_t)L.w.n_samples;
            L.batch = all_at_once ? 2 : kMaxBatch;
            L.finished = false;
            L.all_started = all_at_once;
            L.flushed = all_at_once;
            // A sample visits at most (bounce cap + 1) vertices, each at most one path ray, one light sample of up to 32
            // visibility segments (pathtracer.cu:279) and one bsdf stage; a slot runs its share of the samples one after
            // the other. A loop that passes this bound is not going to end (a slot that keeps looking live): stop with
            // an error instead of occupying the GPU for ever.
            L.issued = 0;
            L.issue_cap = ((uint64_t)L.w.n_samples / (uint64_t)L.w.n_slots + 2u) * ((uint64_t)c->max_bounces + 1u) * 34u + 64u;
            rc = lane_issue_batch(c, L, cam, ctr);
            if (rc != PT_OK) return rc;
        }
        for (int remaining = n_lanes; remaining > 0;) {
            for (int i = 0; i < n_lanes; i++) {
                pt_ctx::WfLane& L = c->lanes[i];
                if (L.finished) continue;
                HIP_TRY(hipEventSynchronize(L.ready));
                if (c->stage_timing) {
                    for (int k = 0; k < L.batch; k++) {
                        hipEvent_t* ev = L.st_ev + kStageEvents * k;
                        for (int q = 0; q < PT_STAGE_KERNELS; q++) { float ms = 0.f; if (hipEventElapsedTime(&ms, ev[q], ev[q + 1]) == hipSuccess) c->stage_ms[q] += ms; }
                        float ms = 0.f;
                        if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) { c->extend_ms += ms; c->extend_launches++; }
                    }
                }
                const uint32_t* hc = L.host_counters;
                // what the last iteration of the batch left in the queues it appended to
                // (without light samples the shadow queue holds only records that went through the bsdf stage already)
                uint64_t live = 0;
                for (int r = 0; r < kStripes; r++)
                    live += (uint64_t)hc[WF_CTR_TAIL_P + r * kStripeWords] + (c->scene.sky_mode == PT_SKY_SUNSKY ? 0u : (uint64_t)hc[WF_CTR_TAIL_S + r * kStripeWords]);
                // Nothing more can start once the global hand-out had run dry before the batch was issued and the
                // batch began by starting the ids the workgroups still held.
                const bool settled = L.flushed;
                if (hc[WF_CTR_NEXT_SAMPLE] >= L.w.n_samples) { L.batch = 2; L.all_started = true; }
                bool done = settled && live == 0;
                // every sample started and few paths left: finish them in one kernel instead of one stage
                // launch sequence per remaining path vertex
                if (!done && settled && c->drain_rays > 0 && live <= (uint64_t)c->drain_rays) {
                    launch_wf_drain(lane_view(L), c->scene, cam, (int)live, ctr, L.stream);
                    HIP_TRY(hipGetLastError());
                    c->wf_drains++;
                    done = true;
                }
                if (!done && L.issued > L.issue_cap) {
                    set_error("render: lane %d still has live paths after %llu stage iterations (bound %llu)", i, (unsigned long long)L.issued, (unsigned long long)L.issue_cap);
                    return PT_ERR_STATE;
                }
                if (done) {
                    if (!hold) launch_wf_accumulate(L.w, cam, c->accum, 0, L.w.n_frames, L.stream);
                    HIP_TRY(hipGetLastError());
                    L.finished = true; remaining--;
                    continue;
                }
                int rc = lane_issue_batch(c, L, cam, ctr);
                if (rc != PT_OK) return rc;
            }
        }
        if (lanes_used) *lanes_used = n_lanes;
        // the context's stream continues after the other lanes
        for (int i = 1; i < n_lanes; i++) {
            HIP_TRY(hipEventRecord(c->lanes[i].ready, c->lanes[i].stream));
            HIP_TRY(hipStreamWaitEvent(base, c->lanes[i].ready, 0));
        }
    }
    return PT_OK;
}

int pt_stage_timing(pt_ctx* c, int on, double* extend_ms_out, uint64_t* extend_launches_out)
{
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    if (extend_ms_out) *extend_ms_out = c->extend_ms;
    if (extend_launches_out) *extend_launches_out = c->extend_launches;
    c->stage_timing = on != 0;
    c->extend_ms = 0.0; c->extend_launches = 0;
    for (double& v : c->stage_ms) v = 0.0;
    return PT_OK;
}

int pt_stage_times(pt_ctx* c, double* ms_out, uint64_t* iterations_out)
{
    if (!c || !ms_out) { set_error("pt_stage_times: bad argument"); return PT_ERR_ARG; }
    for (int q = 0; q < PT_STAGE_KERNELS; q++) ms_out[q] = c->stage_ms[q];
    if (iterations_out) *iterations_out = c->extend_launches;
    return PT_OK;
}

// diagnostic builds only: the 16 debug words the extend kernel accumulates (zeroed on read)
int pt_debug_words(pt_ctx* c, unsigned long long* out16)
{
    if (!c || !out16 || !c->lanes[0].counters) { set_error("pt_debug_words: nothing to read"); return PT_ERR_STATE; }
    HIP_TRY(hipDeviceSynchronize());
    for (int k = 0; k < 16; k++) out16[k] = 0;
    for (int i = 0; i < pt_ctx::kMaxLanes; i++) {
        if (!c->lanes[i].counters) continue;
        unsigned long long part[16];
        HIP_TRY(hipMemcpy(part, c->lanes[i].counters + WF_CTR_WORDS, sizeof(part), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(c->lanes[i].counters + WF_CTR_WORDS, 0, sizeof(part)));
        for (int k = 0; k < 16; k++) out16[k] += part[k];
    }
    return PT_OK;
}

int pt_wavefront_stats(pt_ctx* c, uint64_t* iterations_out, uint64_t* drains_out)
{
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    if (iterations_out) *iterations_out = c->wf_iterations;
    if (drains_out) *drains_out = c->wf_drains;
    return PT_OK;
}

int pt_set_extend_tuning(pt_ctx* c, int refill_lanes, int leaf_vote, int blocks_per_cu)
{
    if (!c || refill_lanes < 1 || refill_lanes > 64 || leaf_vote < 1 || leaf_vote > 64 || blocks_per_cu < 1 || blocks_per_cu > 5) { set_error("pt_set_extend_tuning: bad argument"); return PT_ERR_ARG; }
    c->refill_lanes = refill_lanes; c->leaf_vote = leaf_vote; c->extend_blocks_per_cu = blocks_per_cu;
    return PT_OK;
}

int pt_set_node_width(pt_ctx* c, int width, int leaf_merge)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || (width != 2 && width != 4) || leaf_merge < 1 || leaf_merge > 16) { set_error("pt_set_node_width: width must be 2 or 4, leaf_merge 1..16"); return PT_ERR_ARG; }
    c->node_width = width;
    c->leaf_merge = leaf_merge;         // takes effect at the next pt_upload_bvh / pt_build_bvh
    c->scene.wide = (c->wide_fits && width == 4) ? 1 : 0;
    return PT_OK;
}

int pt_set_mode(pt_ctx* c, int mode)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || (mode != PT_MODE_WAVEFRONT && mode != PT_MODE_SINGLE_KERNEL)) { set_error("pt_set_mode: bad argument"); return PT_ERR_ARG; }
    c->mode = mode;
    return PT_OK;
}

int pt_set_lanes(pt_ctx* c, int lanes)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || lanes < 1 || lanes > pt_ctx::kMaxLanes) { set_error("pt_set_lanes: 1 or 2 lanes"); return PT_ERR_ARG; }
    c->n_lanes = lanes;
    c->lanes_forced = true;
    return PT_OK;
}

int pt_set_drain(pt_ctx* c, int max_rays)
{
    if (!c || max_rays < 0) { set_error("pt_set_drain: bad argument"); return PT_ERR_ARG; }
    c->drain_rays = max_rays;
    return PT_OK;
}

int pt_set_pool(pt_ctx* c, int slots, int chunk_frames)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    if (!c || slots < 64 || chunk_frames < 1) { set_error("pt_set_pool: bad argument"); return PT_ERR_ARG; }
    c->pool_slots = slots;
    c->chunk_frames = chunk_frames;
    return PT_OK;
}

int pt_render_headless(pt_ctx* c, const pt_params* P, int frame_begin, int frame_end)
{
    int rc = check_ready(c, true);
    if (rc != PT_OK) return rc;
    if (!P || frame_begin < 0 || frame_end < frame_begin || frame_end > 524288) { set_error("pt_render_headless: bad frame range [%d,%d)", frame_begin, frame_end); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    ahead_quiesce(c);               // the lanes and their staging arrays are about to be used
    DCamera cam;
    if ((rc = make_camera(c, P, &cam)) != PT_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (c->mode == PT_MODE_WAVEFRONT && cam.is_rendering && frame_end > frame_begin) {
        if ((rc = render_wavefront(c, cam, frame_begin, frame_end)) != PT_OK) return rc;
    } else {
        launch_render_samples(c->scene, cam, frame_begin, frame_end, c->accum, c->counting ? c->d_ctr : nullptr, c->stream);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    if (c->counting) { if ((rc = fetch_counters(c)) != PT_OK) return rc; }
    return PT_OK;
}

int pt_sync(pt_ctx* c)
{
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->totals.last_ms = ms;
    else (void)hipGetLastError();       // no timed call yet: the events were never recorded; the error must not stick to the next launch check
    return PT_OK;
}

// Everything of params that a sample or the displayed image depends on, except the frame index and the
// caller's buffer.
static bool same_view(const pt_params& a, const pt_params& b)
{
    pt_params x = a, y = b;
    x.frameIndex = y.frameIndex = 0;
    x.pixelBuffer = y.pixelBuffer = nullptr;
    return memcmp(&x, &y, sizeof(pt_params)) == 0;
}

int pt_set_frame_lookahead(pt_ctx* c, int max_frames)
{
    if (!c || max_frames < 1 || max_frames > 64) { set_error("pt_set_frame_lookahead: bad argument"); return PT_ERR_ARG; }
    ahead_quiesce(c);
    c->lookahead_max = max_frames;
    return PT_OK;
}

// One frame of the reference's loop up to and including its addition to the accumulator (no resolve, no copy).
static int render_frame_core(pt_ctx* c, pt_params* P)
{
    int rc = check_ready(c, true);
    if (rc != PT_OK) return rc;
    if (!P) { set_error("pt_render_frame: null params"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    // pathtracer.cu:583-586: the first frame of a render and every preview frame overwrite
    if (P->frameIndex == 0 || !P->isRendering) { if ((rc = pt_clear_accum(c)) != PT_OK) return rc; }
    pt_ctx::Ahead& A = c->ahead;
    const bool can_stage = c->mode == PT_MODE_WAVEFRONT && P->isRendering && !c->counting && !c->stage_timing && c->lookahead_max > 1 &&
                           P->frameIndex >= 0 && P->frameIndex < 524288 - 128;
    // A caller that keeps asking for the next frame of one view (the reference's main loop while the camera rests,
    // main.cpp:163-169) gets frames rendered ahead: a batch fills the GPU where a single 1-spp frame mostly ramps up and
    // down. The first batch (4 frames) is rendered by the call that notices the pattern; from then on a worker thread
    // renders the next batch (8, 16, ... lookahead_max frames) into the lanes' other staging arrays on streams of its own
    // while the calls hand out the current one, so no call waits for a batch. A staged frame joins the accumulator only
    // when its call comes, in frame order, so every call returns exactly the image the one-frame-at-a-time loop would.
    const bool continues = A.last_frame >= 0 && P->frameIndex == A.last_frame + 1 && A.epoch == c->epoch && same_view(*P, A.params);
    A.streak = continues ? A.streak + 1 : 0;
    if (!continues || !can_stage) { if (A.valid || A.pending) ahead_quiesce(c); }
    DCamera cam;
    if ((rc = make_camera(c, P, &cam)) != PT_OK) return rc;
    const int chunk = effective_chunk_frames(c);        // a batch must fit one staging chunk
    auto batch_size = [&](int after) { int k = after * 2; k = k < 4 ? 4 : k; k = k < c->lookahead_max ? k : c->lookahead_max; return k < chunk ? k : chunk; };
    // the batch in hand is used up: the one the worker rendered takes its place
    if (can_stage && continues && A.valid && P->frameIndex == A.first + A.count && A.pending && P->frameIndex == A.p_first) {
        A.worker.join();
        A.pending = false;
        if (A.p_rc != PT_OK) { set_error("%s", A.p_error.c_str()); A.valid = false; return A.p_rc; }
        HIP_TRY(hipStreamWaitEvent(c->stream, A.staged, 0));
        A.buf ^= 1;
        A.first = A.next = A.p_first; A.count = A.p_count; A.lanes = A.p_lanes;
        for (int i = 0; i < A.lanes; i++) A.w[i] = A.p_w[i];
    }
    bool staged = can_stage && continues && A.valid && P->frameIndex == A.next && A.next < A.first + A.count;
    if (can_stage && !staged && A.streak >= 1) {
        // no batch in hand: render the first one here
        ahead_quiesce(c);
        const int k = batch_size(0);
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
        int lanes = 0;
        A.buf = 0;
        if ((rc = render_wavefront(c, cam, P->frameIndex, P->frameIndex + k, true, &lanes, A.buf)) != PT_OK) return rc;
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        A.valid = true; A.first = A.next = P->frameIndex; A.count = k; A.lanes = lanes;
        for (int i = 0; i < lanes; i++) A.w[i] = c->lanes[i].w;
        staged = true;
    }
    if (staged && !A.pending && A.first + A.count < 524288 - 128) {
        // start the worker on the batch that follows the one in hand
        if (!A.stream0) HIP_TRY(hipStreamCreateWithFlags(&A.stream0, hipStreamNonBlocking));
        if (!A.staged) HIP_TRY(hipEventCreateWithFlags(&A.staged, hipEventDisableTiming));
        if (!A.consumed) HIP_TRY(hipEventCreateWithFlags(&A.consumed, hipEventDisableTiming));
        // the arrays it fills were handed out before the batch in hand: their last addition is already queued on the
        // context's stream, and so is everything the scene depends on
        HIP_TRY(hipEventRecord(A.consumed, c->stream));
        HIP_TRY(hipStreamWaitEvent(A.stream0, A.consumed, 0));
        A.p_first = A.first + A.count; A.p_count = batch_size(A.count); A.p_rc = PT_OK; A.p_error.clear();
        A.pending = true;
        const int device = c->device, target = A.buf ^ 1;
        A.worker = std::thread([c, cam, device, target]() {
            pt_ctx::Ahead& W = c->ahead;
            int rc = hipSetDevice(device) == hipSuccess ? PT_OK : PT_ERR_HIP;
            int lanes = 0;
            if (rc == PT_OK) rc = render_wavefront(c, cam, W.p_first, W.p_first + W.p_count, true, &lanes, target, W.stream0);
            if (rc == PT_OK && hipEventRecord(W.staged, W.stream0) != hipSuccess) rc = PT_ERR_HIP;
            W.p_lanes = lanes;
            for (int i = 0; i < lanes; i++) W.p_w[i] = c->lanes[i].w;
            W.p_rc = rc;
            if (rc != PT_OK) W.p_error = pt_last_error();      // messages are per thread
        });
    }
    if (staged) {
        for (int i = 0; i < A.lanes; i++) launch_wf_accumulate(A.w[i], cam, c->accum, A.next - A.first, 1, c->stream);
        HIP_TRY(hipGetLastError());
        A.next++;
    } else {
        if ((rc = pt_render_headless(c, P, P->frameIndex, P->frameIndex + 1)) != PT_OK) return rc;
    }
    A.params = *P; A.epoch = c->epoch; A.last_frame = P->isRendering ? P->frameIndex : -2;
    return PT_OK;
}

int pt_render_frame(pt_ctx* c, pt_params* P)
{
    int rc = render_frame_core(c, P);
    if (rc != PT_OK) return rc;
    launch_resolve_display(c->accum, c->W * c->H, P->isRendering ? P->frameIndex + 1 : 1, c->display, c->stream);
    HIP_TRY(hipGetLastError());
    if (P->pixelBuffer) HIP_TRY(hipMemcpyAsync(P->pixelBuffer, c->display, (size_t)c->W * c->H * 4, hipMemcpyDeviceToHost, c->stream));
    return pt_sync(c);
}

// pt_render_frame without the wait: the frame is rendered and added as above, its display image is resolved into one
// of two device buffers and copied to one of two pinned host buffers on a stream of its own; the call returns when
// the work is queued. While the caller keeps a frame in flight, the copy of frame f runs beside the rendering of
// frame f + 1. pt_frame_fetch hands the frames out in order.
int pt_render_frame_async(pt_ctx* c, pt_params* P)
{
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    pt_ctx::AsyncFrames& F = c->af;
    const size_t bytes = (size_t)c->W * c->H * 4;
    if (F.pending[F.next]) { set_error("pt_render_frame_async: two frames are in flight, fetch one first"); return PT_ERR_STATE; }
    int rc = render_frame_core(c, P);
    if (rc != PT_OK) return rc;
    if (!F.copy_stream || F.bytes != bytes) {
        free_async_frames(c);
        HIP_TRY(hipStreamCreateWithFlags(&F.copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; b++) {
            HIP_TRY(hipMalloc((void**)&F.dev[b], bytes));
            HIP_TRY(hipHostMalloc((void**)&F.host[b], bytes, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&F.resolved[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&F.copied[b], hipEventDisableTiming));
        }
        F.bytes = bytes;
    }
    const int b = F.next;
    // (the buffer's last copy was fetched, hence finished, before this call could reuse it)
    launch_resolve_display(c->accum, c->W * c->H, P->isRendering ? P->frameIndex + 1 : 1, F.dev[b], c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(F.resolved[b], c->stream));
    HIP_TRY(hipStreamWaitEvent(F.copy_stream, F.resolved[b], 0));
    HIP_TRY(hipMemcpyAsync(F.host[b], F.dev[b], bytes, hipMemcpyDeviceToHost, F.copy_stream));
    HIP_TRY(hipEventRecord(F.copied[b], F.copy_stream));
    F.pending[b] = true; F.frame[b] = P->frameIndex;
    F.next = b ^ 1;
    return PT_OK;
}

// The oldest frame in flight: waits for its copy, then copies it into rgba_out (when given) and returns the pinned
// buffer itself in *pinned_out (when given; valid until the second pt_render_frame_async call from now).
int pt_frame_fetch(pt_ctx* c, uint8_t* rgba_out, const uint8_t** pinned_out, int* frame_index_out)
{
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    pt_ctx::AsyncFrames& F = c->af;
    const int b = F.oldest;
    if (!F.pending[b]) { set_error("pt_frame_fetch: no frame in flight"); return PT_ERR_STATE; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(F.copied[b]));
    if (rgba_out) memcpy(rgba_out, F.host[b], F.bytes);
    if (pinned_out) *pinned_out = F.host[b];
    if (frame_index_out) *frame_index_out = F.frame[b];
    F.pending[b] = false;
    F.oldest = b ^ 1;
    return PT_OK;
}

int pt_read_accum(pt_ctx* c, float* out)
{
    if (!c || !out || !c->accum) { set_error("pt_read_accum: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(out, c->accum, (size_t)c->W * c->H * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return pt_sync(c);
}

int pt_write_accum(pt_ctx* c, const float* in)
{
    if (!c || !in || !c->accum) { set_error("pt_write_accum: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->accum, in, (size_t)c->W * c->H * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_accum_device_ptr(pt_ctx* c, float** out)
{
    if (!c || !out || !c->accum) { set_error("pt_accum_device_ptr: bad argument"); return PT_ERR_ARG; }
    *out = c->accum;
    return PT_OK;
}

int pt_bind_accum(pt_ctx* c, float* accum_dev)
{
    if (!c || !c->own_accum) { set_error("pt_bind_accum: call pt_init first"); return PT_ERR_STATE; }
    c->accum = accum_dev ? accum_dev : c->own_accum;
    return PT_OK;
}

int pt_resolve_display(pt_ctx* c, int n_frames, uint8_t* rgba_out)
{
    if (!c || !c->accum || n_frames <= 0) { set_error("pt_resolve_display: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    launch_resolve_display(c->accum, c->W * c->H, n_frames, c->display, c->stream);
    HIP_TRY(hipGetLastError());
    if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, c->display, (size_t)c->W * c->H * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_save_png(const char* path, const uint8_t* rgba, int w, int h) { return save_png(path, rgba, w, h); }

// The image as the reference's window shows it: pt_resolve_display followed by the post-process shader
// (post_process.glsl:19-23) with the exposure / mult / saturation / gamma of params.
int pt_resolve_display_post(pt_ctx* c, int n_frames, const pt_params* P, uint8_t* rgba_out)
{
    if (!c || !c->accum || n_frames <= 0 || !P) { set_error("pt_resolve_display_post: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    launch_resolve_display(c->accum, c->W * c->H, n_frames, c->display, c->stream);
    float mult[3] = {P->mult.x, P->mult.y, P->mult.z};
    launch_post_process(c->display, c->W * c->H, P->exposure, mult, P->saturation, P->gamma, c->display, c->stream);
    HIP_TRY(hipGetLastError());
    if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, c->display, (size_t)c->W * c->H * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_load_texture_png(const char* path, int comps, float** texels_out, int* w, int* h)
{
    if (!path || !texels_out || !w || !h) { set_error("pt_load_texture_png: bad argument"); return PT_ERR_ARG; }
    return load_png_texture(path, comps, texels_out, w, h);
}
int pt_load_texture_image(const char* path, int comps, float** texels_out, int* w, int* h)
{
    if (!path || !texels_out || !w || !h) { set_error("pt_load_texture_image: bad argument"); return PT_ERR_ARG; }
    return load_image_texture(path, comps, texels_out, w, h);
}
int pt_save_pfm(const char* path, const float* rgb, int w, int h) { return save_pfm(path, rgb, w, h); }
int pt_save_exr(const char* path, const float* rgb, int w, int h, float scale) { return save_exr(path, rgb, w, h, scale); }
int pt_load_pfm(const char* path, float** rgb_out, int* w, int* h)
{
    if (!path || !rgb_out || !w || !h) { set_error("pt_load_pfm: bad argument"); return PT_ERR_ARG; }
    return load_pfm(path, rgb_out, w, h);
}

int pt_trace_closest_dev(pt_ctx* c, const pt_ray* rays_dev, int n, pt_hit* hits_dev)
{
    int rc = check_ready(c, false);
    if (rc != PT_OK) return rc;
    if (n < 0 || (n > 0 && (!rays_dev || !hits_dev))) { set_error("pt_trace_closest_dev: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    launch_trace_closest(c->scene, rays_dev, n, hits_dev, nullptr, nullptr, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    return PT_OK;
}

// Closest hits through the production traversal kernel (persistent waves, compact nodes): what the
// wavefront renderer's extend stage runs. Device pointers.
int pt_trace_closest_fast_dev(pt_ctx* c, const pt_ray* rays_dev, int n, pt_hit* hits_dev)
{
    if (c) state_change(c);     // frames staged ahead by pt_render_frame are stale after this call
    int rc = check_ready(c, false);
    if (rc != PT_OK) return rc;
    if (n <= 0 || !rays_dev || !hits_dev) { set_error("pt_trace_closest_fast_dev: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    // borrow lane 0's queue arrays: the batch is a shadow queue of n pending rays
    size_t slots = ((size_t)n + 255) / 256 * 256;
    WfParams w{};
    rc = lane_pool(c, c->lanes[0], slots, &w);
    if (rc != PT_OK) return rc;
    HIP_TRY(hipMemsetAsync(w.counters, 0, WF_CTR_WORDS * sizeof(uint32_t), c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    launch_wf_trace(w, c->scene, rays_dev, n, hits_dev, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    return PT_OK;
}

// The same batch with every intermediate in the library's built-in device block: no hipMalloc, hipFree or
// other allocating call on the way (events and streams were made by pt_create).
int pt_trace_closest_noalloc_dev(pt_ctx* c, const pt_ray* rays_dev, int n, pt_hit* hits_dev)
{
    int rc = check_ready(c, false);
    if (rc != PT_OK) return rc;
    if (n <= 0 || !rays_dev || !hits_dev) { set_error("pt_trace_closest_noalloc_dev: bad argument"); return PT_ERR_ARG; }
    if (n > PT_NOALLOC_MAX_RAYS) { set_error("pt_trace_closest_noalloc_dev: %d rays exceed the built-in block's %d; split the batch or use pt_trace_closest_fast_dev", n, PT_NOALLOC_MAX_RAYS); return PT_ERR_ARG; }
    if (!c->static_scratch) { set_error("pt_trace_closest_noalloc_dev: the built-in device block is not available on device %d (%s)", c->device, hipGetErrorString(c->static_scratch_error)); return PT_ERR_STATE; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(launch_wf_trace_static(c->static_scratch, c->scene, rays_dev, n, hits_dev, c->refill_lanes, c->leaf_vote,
                                   (c->extend_blocks_per_cu ? c->extend_blocks_per_cu : 5) * c->n_cus, c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    return PT_OK;
}

int pt_trace_closest_fast(pt_ctx* c, const pt_ray* rays, int n, pt_hit* hits)
{
    if (!c || n <= 0 || !rays || !hits) { set_error("pt_trace_closest_fast: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d_r, d_h;
    HIP_TRY(d_r.alloc((size_t)n * sizeof(pt_ray)));
    HIP_TRY(d_h.alloc((size_t)n * sizeof(pt_hit)));
    hipError_t e = hipMemcpyAsync(d_r.p, rays, (size_t)n * sizeof(pt_ray), hipMemcpyHostToDevice, c->stream);
    int rc = e == hipSuccess ? pt_trace_closest_fast_dev(c, d_r.as<pt_ray>(), n, d_h.as<pt_hit>()) : PT_OK;
    if (e == hipSuccess && rc == PT_OK) e = hipMemcpyAsync(hits, d_h.p, (size_t)n * sizeof(pt_hit), hipMemcpyDeviceToHost, c->stream);
    int rs = pt_sync(c);                                    // the stream is idle before the buffers are freed
    HIP_TRY(e);
    return rc != PT_OK ? rc : rs;
}

int pt_trace_closest_counted(pt_ctx* c, const pt_ray* rays, int n, pt_hit* hits, uint32_t* nbox, uint32_t* ntri)
{
    int rc = check_ready(c, false);
    if (rc != PT_OK) return rc;
    if (n < 0 || (n > 0 && (!rays || !hits))) { set_error("pt_trace_closest: bad argument"); return PT_ERR_ARG; }
    if ((nbox == nullptr) != (ntri == nullptr)) { set_error("pt_trace_closest_counted: pass both counter arrays or neither"); return PT_ERR_ARG; }
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d_r, d_h, d_b, d_t;
    HIP_TRY(d_r.alloc((size_t)n * sizeof(pt_ray)));
    HIP_TRY(d_h.alloc((size_t)n * sizeof(pt_hit)));
    if (nbox) { HIP_TRY(d_b.alloc((size_t)n * 4)); HIP_TRY(d_t.alloc((size_t)n * 4)); }
    auto run = [&]() -> hipError_t {
        hipError_t e;
        if ((e = hipMemcpyAsync(d_r.p, rays, (size_t)n * sizeof(pt_ray), hipMemcpyHostToDevice, c->stream)) != hipSuccess) return e;
        if ((e = hipEventRecord(c->ev0, c->stream)) != hipSuccess) return e;
        launch_trace_closest(c->scene, d_r.as<pt_ray>(), n, d_h.as<pt_hit>(), d_b.as<uint32_t>(), d_t.as<uint32_t>(), c->stream);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipEventRecord(c->ev1, c->stream)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(hits, d_h.p, (size_t)n * sizeof(pt_hit), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
        if (nbox) {
            if ((e = hipMemcpyAsync(nbox, d_b.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
            if ((e = hipMemcpyAsync(ntri, d_t.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
        }
        return hipSuccess;
    };
    hipError_t e = run();
    rc = pt_sync(c);                                        // the stream is idle before the buffers are freed
    HIP_TRY(e);
    return rc;
}

int pt_trace_closest(pt_ctx* c, const pt_ray* rays, int n, pt_hit* hits) { return pt_trace_closest_counted(c, rays, n, hits, nullptr, nullptr); }

int pt_get_counters(pt_ctx* c, pt_counters* out)
{
    if (!c || !out) { set_error("pt_get_counters: null argument"); return PT_ERR_ARG; }
    *out = c->totals;
    return PT_OK;
}

int pt_reset_counters(pt_ctx* c)
{
    if (!c) { set_error("null context"); return PT_ERR_ARG; }
    double ms = c->totals.last_ms;
    c->totals = pt_counters{};
    c->totals.last_ms = ms;
    return PT_OK;
}

int pt_scene_bytes(pt_ctx* c, uint64_t* node_bytes, uint64_t* tri_bytes)
{
    if (!c || !c->have_bvh) { set_error("pt_scene_bytes: no BVH"); return PT_ERR_STATE; }
    // the nodes the production traversal walks: 4-wide, or the quantized binary ones
    if (node_bytes) *node_bytes = c->scene.wide ? c->n_nodes4 * sizeof(DNode4) : c->n_nodes * sizeof(DNodeQ);
    if (tri_bytes) *tri_bytes = c->tris.size() * sizeof(DTri);
    return PT_OK;
}

// numerics probe: 12 outputs per input pair (see pt_kernels.hip)
int pt_detmath_probe(pt_ctx* c, const float* x, const float* y, int n, float* out)
{
    if (!c || !x || !y || !out || n <= 0) { set_error("pt_detmath_probe: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    DevBuf dx, dy, dout;
    HIP_TRY(dx.alloc((size_t)n * 4));
    HIP_TRY(dy.alloc((size_t)n * 4));
    HIP_TRY(dout.alloc((size_t)n * 48));
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy.p, y, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_detmath_probe(dx.as<float>(), dy.as<float>(), n, dout.as<float>(), c->stream);
    hipError_t e = hipGetLastError();
    hipError_t es = hipStreamSynchronize(c->stream);        // nothing may still use the buffers when they are freed
    HIP_TRY(e);
    HIP_TRY(es);
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * 48, hipMemcpyDeviceToHost));
    return PT_OK;
}

// BSDF probe: 12 outputs per item (see pt_kernels.hip)
int pt_bsdf_probe(pt_ctx* c, const pt_material* mats, const int* inside, const float* V, const float* N, const float* L, const uint32_t* pixel, int frame, int n, float* out12)
{
    if (!c || !mats || !inside || !V || !N || !L || !pixel || !out12 || n <= 0) { set_error("pt_bsdf_probe: bad argument"); return PT_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    DevBuf dm, di, dv, dn, dl, dp, dout;
    HIP_TRY(dm.alloc((size_t)n * sizeof(pt_material))); HIP_TRY(di.alloc((size_t)n * 4)); HIP_TRY(dv.alloc((size_t)n * 12)); HIP_TRY(dn.alloc((size_t)n * 12));
    HIP_TRY(dl.alloc((size_t)n * 12)); HIP_TRY(dp.alloc((size_t)n * 4)); HIP_TRY(dout.alloc((size_t)n * 48));
    HIP_TRY(hipMemcpy(dm.p, mats, (size_t)n * sizeof(pt_material), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(di.p, inside, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dv.p, V, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dn.p, N, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dl.p, L, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dp.p, pixel, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_bsdf_probe(dm.as<pt_material>(), di.as<int>(), dv.as<float>(), dn.as<float>(), dl.as<float>(), dp.as<uint32_t>(), frame, n, dout.as<float>(), c->stream);
    hipError_t e = hipGetLastError();
    hipError_t es = hipStreamSynchronize(c->stream);        // nothing may still use the buffers when they are freed
    HIP_TRY(e);
    HIP_TRY(es);
    HIP_TRY(hipMemcpy(out12, dout.p, (size_t)n * 48, hipMemcpyDeviceToHost));
    return PT_OK;
}

// ---------------------------------------------------------------- host-side producers
struct pt_host_scene { HostScene s; };

int pt_read_obj(const char* obj_path, const char* model_dir, const char* sub_dir, pt_host_scene** out)
{
    if (!obj_path || !out) { set_error("pt_read_obj: null argument"); return PT_ERR_ARG; }
    pt_host_scene* hs = new (std::nothrow) pt_host_scene();
    if (!hs) { set_error("out of memory"); return PT_ERR_NOMEM; }
    int rc = read_obj(obj_path, model_dir, sub_dir, hs->s);
    if (rc != PT_OK) { delete hs; *out = nullptr; return rc; }
    *out = hs;
    return PT_OK;
}
int pt_read_obj_general(const char* obj_path, const char* model_dir, const char* sub_dir, float scale, pt_host_scene** out)
{
    if (!obj_path || !out || !(scale > 0.f)) { set_error("pt_read_obj_general: bad argument"); return PT_ERR_ARG; }
    pt_host_scene* hs = new (std::nothrow) pt_host_scene();
    if (!hs) { set_error("out of memory"); return PT_ERR_NOMEM; }
    int rc = read_obj_general(obj_path, model_dir, sub_dir, scale, hs->s);
    if (rc != PT_OK) { delete hs; *out = nullptr; return rc; }
    *out = hs;
    return PT_OK;
}
void pt_host_scene_free(pt_host_scene* s) { delete s; }
int pt_host_scene_counts(const pt_host_scene* s, int* n_tri, int* n_mat, int* n_tex)
{
    if (!s) { set_error("null scene"); return PT_ERR_ARG; }
    if (n_tri) *n_tri = (int)s->s.tris.size();
    if (n_mat) *n_mat = (int)s->s.mats.size();
    if (n_tex) *n_tex = (int)s->s.textures.size();
    return PT_OK;
}
pt_triangle* pt_host_scene_triangles(pt_host_scene* s) { return s ? s->s.tris.data() : nullptr; }
int* pt_host_scene_tri_idx(pt_host_scene* s) { return s ? s->s.tri_idx.data() : nullptr; }
pt_material* pt_host_scene_materials(pt_host_scene* s) { return s ? s->s.mats.data() : nullptr; }
const char* pt_host_scene_material_name(const pt_host_scene* s, int i)
{
    if (!s || i < 0 || i >= (int)s->s.mat_names.size()) return nullptr;
    return s->s.mat_names[i].c_str();
}
int pt_host_scene_texture_info(const pt_host_scene* s, int i, const char** path, int* comps)
{
    if (!s || i < 0 || i >= (int)s->s.textures.size()) { set_error("texture index out of range"); return PT_ERR_ARG; }
    if (path) *path = s->s.textures[i].path.c_str();
    if (comps) *comps = s->s.textures[i].comps;
    return PT_OK;
}

int pt_host_build_bvh(pt_triangle* tris, int* tri_idx, int n_tri, int flavour, pt_bvh_node* nodes_out, int* nodes_used_out)
{
    if (!tris || !tri_idx || !nodes_out || !nodes_used_out || n_tri <= 0) { set_error("pt_host_build_bvh: bad argument"); return PT_ERR_ARG; }
    if (flavour != PT_BVH_REFERENCE) { set_error("pt_host_build_bvh: unknown flavour %d", flavour); return PT_ERR_ARG; }
    for (int i = 0; i < n_tri; i++) tri_idx[i] = i;
    return build_bvh_reference(tris, tri_idx, n_tri, nodes_out, nodes_used_out);
}

int pt_load_hdr(const char* path, float** rgba_out, int* w, int* h)
{
    if (!path || !rgba_out || !w || !h) { set_error("pt_load_hdr: null argument"); return PT_ERR_ARG; }
    return load_hdr(path, rgba_out, w, h);
}
int pt_build_env_cdf(const float* rgba, int w, int h, float* cdf_out, float* sum_out)
{
    if (!rgba || !cdf_out || !sum_out || w <= 0 || h <= 0) { set_error("pt_build_env_cdf: bad argument"); return PT_ERR_ARG; }
    build_env_cdf(rgba, w, h, cdf_out, sum_out);
    return PT_OK;
}
void pt_free(void* p) { free(p); }

// ---------------------------------------------------------------- several GPUs in one process
// The render shards by sample: every (pixel, frame) owns a random stream and frames meet only in the accumulator
// sum (pathtracer.cu:560,583-586). A group holds one context per GPU with the same scene uploaded to each; a render
// gives context r of n the frames [begin + r*F/n, begin + (r+1)*F/n), one host thread per context, and sums the
// fp32 accumulators onto the first context with ONE reduce over xGMI (RCCL); no other exchange exists. RCCL is
// looked up at run time when a group of more than one GPU is made, so single-GPU users never load it.
namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if (handle) return true;
        // PTAMD_RCCL_LIBRARY names the library to use instead (an installation whose RCCL lives elsewhere; the tests'
        // stand-in that sums through host memory on a one-GPU machine). Otherwise the name is looked up as the loader
        // does: a process that already holds an RCCL (torch brings its own, soname librccl.so.1) gets that one, so the
        // group and torch.distributed share one copy and one HIP runtime.
        const char* named = getenv("PTAMD_RCCL_LIBRARY");
        if (named && *named) handle = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        else for (const char* name : {"librccl.so.1", "librccl.so"}) { handle = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (handle) break; }
        if (!handle) { set_error("pt_group_create: cannot load RCCL (%s)", dlerror()); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(handle, n); if (!p) set_error("pt_group_create: RCCL lacks %s", n); return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Reduce = (decltype(Reduce))sym("ncclReduce");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        return CommInitAll && CommDestroy && Reduce && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl g_rccl;
}

struct pt_group {
    std::vector<pt_ctx*> ctx;
    std::vector<int> device;          // of each context, kept here so that the group can be destroyed after its contexts
    std::vector<ncclComm_t> comm;     // empty for a group of one
};

int pt_group_frames(int frame_begin, int frame_end, int n, int r, int* first_out, int* last_out)
{
    if (n < 1 || r < 0 || r >= n || frame_end < frame_begin || !first_out || !last_out) { set_error("pt_group_frames: bad argument"); return PT_ERR_ARG; }
    const long long F = (long long)frame_end - frame_begin;
    *first_out = frame_begin + (int)(F * r / n);
    *last_out = frame_begin + (int)(F * (r + 1) / n);
    return PT_OK;
}

int pt_group_create(pt_ctx** ctxs, int n, pt_group** out)
{
    if (!out) { set_error("pt_group_create: null output"); return PT_ERR_ARG; }
    *out = nullptr;
    if (!ctxs || n < 1 || n > 64) { set_error("pt_group_create: 1 to 64 contexts"); return PT_ERR_ARG; }
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || !ctxs[i]->accum) { set_error("pt_group_create: context %d is not initialised", i); return PT_ERR_STATE; }
        if (ctxs[i]->W != ctxs[0]->W || ctxs[i]->H != ctxs[0]->H) { set_error("pt_group_create: context %d has another image size", i); return PT_ERR_ARG; }
        // (PTAMD_GROUP_SHARED_GPU=1 lifts this for rehearsals on a one-GPU machine with a stand-in for RCCL; RCCL itself
        // refuses two ranks on one device)
        const char* shared = getenv("PTAMD_GROUP_SHARED_GPU");
        if (!(shared && shared[0] == '1'))
            for (int j = 0; j < i; j++) if (ctxs[j]->device == ctxs[i]->device) { set_error("pt_group_create: contexts %d and %d share GPU %d; a group takes one context per GPU", j, i, ctxs[i]->device); return PT_ERR_ARG; }
        for (int j = 0; j < i; j++) if (ctxs[j] == ctxs[i]) { set_error("pt_group_create: context %d is listed twice", i); return PT_ERR_ARG; }
    }
    pt_group* g = new (std::nothrow) pt_group();
    if (!g) { set_error("out of memory"); return PT_ERR_NOMEM; }
    g->ctx.assign(ctxs, ctxs + n);
    for (int i = 0; i < n; i++) g->device.push_back(ctxs[i]->device);
    if (n > 1) {
        if (!g_rccl.load()) { delete g; return PT_ERR_STATE; }
        std::vector<int> devs(n);
        for (int i = 0; i < n; i++) devs[i] = ctxs[i]->device;
        g->comm.resize(n);
        ncclResult_t r = g_rccl.CommInitAll(g->comm.data(), n, devs.data());
        if (r != ncclSuccess) { set_error("ncclCommInitAll failed: %s", g_rccl.GetErrorString(r)); delete g; return PT_ERR_HIP; }
    }
    *out = g;
    return PT_OK;
}

void pt_group_destroy(pt_group* g)
{
    if (!g) return;
    for (size_t i = 0; i < g->comm.size(); i++) { (void)hipSetDevice(g->device[i]); (void)g_rccl.CommDestroy(g->comm[i]); }
    delete g;
}

int pt_group_render(pt_group* g, const pt_params* P, int frame_begin, int frame_end)
{
    if (!g || !P || frame_end < frame_begin) { set_error("pt_group_render: bad argument"); return PT_ERR_ARG; }
    const int n = (int)g->ctx.size();
    // contexts 1.. start from zero: what they render is added onto context 0, which keeps what it holds (as
    // pt_render_headless adds onto the accumulator)
    std::vector<int> rc(n, PT_OK);
    std::vector<std::string> msg(n);
    auto work = [&](int r) {
        int f0 = 0, f1 = 0;
        int e = pt_group_frames(frame_begin, frame_end, n, r, &f0, &f1);
        if (e == PT_OK && r > 0) e = pt_clear_accum(g->ctx[r]);
        if (e == PT_OK) e = pt_render_headless(g->ctx[r], P, f0, f1);
        rc[r] = e;
        if (e != PT_OK) msg[r] = pt_last_error();       // messages are per thread
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < n; r++) threads.emplace_back(work, r);
    work(0);
    for (std::thread& t : threads) t.join();
    for (int r = 0; r < n; r++) if (rc[r] != PT_OK) { set_error("GPU %d: %s", g->ctx[r]->device, msg[r].c_str()); return rc[r]; }
    if (n > 1) {
        // one reduce, enqueued behind each context's render on that context's stream
        const size_t count = (size_t)g->ctx[0]->W * g->ctx[0]->H * 3;
        ncclResult_t r = g_rccl.GroupStart();
        for (int i = 0; i < n && r == ncclSuccess; i++) {
            if (hipSetDevice(g->ctx[i]->device) != hipSuccess) { r = ncclUnhandledCudaError; break; }
            r = g_rccl.Reduce(g->ctx[i]->accum, g->ctx[i]->accum, count, ncclFloat, ncclSum, 0, g->comm[i], g->ctx[i]->stream);
        }
        ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { set_error("ncclReduce failed: %s", g_rccl.GetErrorString(r)); return PT_ERR_HIP; }
    }
    for (int i = 0; i < n; i++) { int e = pt_sync(g->ctx[i]); if (e != PT_OK) return e; }
    return PT_OK;
}

} // extern "C"
