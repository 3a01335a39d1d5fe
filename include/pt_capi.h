/*
 * pt_capi.h — C-ABI of the MI355X-native render hot path.
 *
 * Every entry point replaces one function of the reference's device boundary
 * (namespace pathtracer, "Cuda Pathtracer/core/CUDA/pathtracer.cuh:48-54") or
 * one of the host-side producers the reference application calls directly
 * (file_reader.h, BVH_builder.cpp, envmap.cpp, image_loader.h).  The POD
 * structs keep the reference's field order and byte layout
 * (core/CUDA/utility/structs.cuh) so host arrays produced by either side can
 * be handed to the other unchanged.
 *
 * Conventions: every function returns PT_OK (0) or a negative error code;
 * pt_last_error() returns a thread-local message for the last failure.
 * Pointers are host pointers unless the parameter name ends in _dev.
 * A context is bound to one HIP device; calls on one context must not overlap.
 */
#ifndef PT_CAPI_H
#define PT_CAPI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK            0
#define PT_ERR_ARG      -1
#define PT_ERR_HIP      -2
#define PT_ERR_STATE    -3
#define PT_ERR_IO       -4
#define PT_ERR_NOMEM    -5

/* sky selection (pathtracer.cu:346-380 holds both an envmap and a sun/sky model) */
#define PT_SKY_ENVMAP    0
#define PT_SKY_SUNSKY    1

/* how pt_render_headless / pt_render_frame schedule the work */
#define PT_MODE_WAVEFRONT     0   /* stage kernels over dense queues of samples in flight (default) */
#define PT_MODE_SINGLE_KERNEL 1   /* one kernel, one pixel per lane, as pathtracer.cu:544-606 launches it */

/* BVH build flavours for pt_host_build_bvh / pt_build_bvh */
#define PT_BVH_REFERENCE 0   /* identical tree to BVH_builder.cpp:12-128, built with binned sweeps */
#define PT_BVH_GPU_LBVH  1   /* pt_build_bvh only: Morton-order radix tree built on the GPU in milliseconds; same images (closest hits do not depend on the tree), different test counters */
#define PT_BVH_GPU_LBVH_VIA_HOST 2   /* pt_build_bvh only: the same tree, copied to the host in the reference's node layout and taken in through pt_upload_bvh (cross-check of the device-side layouts) */
#define PT_BVH_GPU_PLOC  3   /* pt_build_bvh only: agglomerative tree built on the GPU over the Morton order (clusters merge with their nearest neighbour within a window), its first levels rebuilt on the host with a surface-area sweep; about the build time of PT_BVH_GPU_LBVH, traversal cost within a few percent of the SAH tree's; same images */

typedef struct pt_float2 { float x, y; } pt_float2;
typedef struct pt_float3 { float x, y, z; } pt_float3;
typedef struct pt_int2 { int x, y; } pt_int2;

/* structs.cuh:111-128 — 112 bytes */
typedef struct pt_triangle {
    pt_float3 a, b, c, origin;
    pt_float3 nA, nB, nC;
    pt_float2 tA, tB, tC;
    int matIndex;
} pt_triangle;

/* structs.cuh:130-137 — 32 bytes */
typedef struct pt_bvh_node {
    pt_float3 aabbMin, aabbMax;
    int leftFirst, triCount;
} pt_bvh_node;

/* structs.cuh:36-44 — 28 bytes */
typedef struct pt_medium {
    float G;
    pt_float3 sigmaA;
    pt_float3 sigmaS;
} pt_medium;

/* structs.cuh:45-95 — 176 bytes. Texture fields hold handles returned by
 * pt_upload_texture (0 = none), in place of cudaTextureObject_t. */
typedef struct pt_material {
    pt_float3 baseColor;
    pt_float3 emissive;
    float roughness;
    float anisotropic;
    float metallic;
    float subsurface;
    float sheen;
    float clearcoat;
    float clearcoatRoughness;
    float specTrans;
    float alpha;
    float ior;
    pt_medium medium;
    uint32_t _pad0;
    uint64_t diffuseTexture;   uint8_t useTexture; uint8_t _p1[7];
    uint64_t roughnessTexture; uint8_t use_mapPr;  uint8_t _p2[7];
    uint64_t metallicTexture;  uint8_t use_mapPm;  uint8_t _p3[7];
    uint64_t emissiveTexture;  uint8_t use_mapKe;  uint8_t _p4[7];
    uint64_t normalTexture;    uint8_t use_mapNor; uint8_t _p5[7];
} pt_material;

/* structs.cuh:149-181 (kernelParams) — 152 bytes */
typedef struct pt_params {
    pt_int2 windowSize;
    pt_float3 rayOrigin;
    pt_float3 rayDirectionZ;
    pt_float3 cameraAngle;
    float cameraSpeed;
    pt_float3 sunDirection;
    float focalDistance;
    float DOF_strenght;
    float fov;
    int frameIndex;
    uint32_t _pad0;
    uint8_t* pixelBuffer;
    uint8_t isRendering; uint8_t _p1[3];
    float envmap_sum;
    pt_int2 envMap_size;
    uint64_t cubeMap;
    uint64_t envMap_cdf;
    pt_float3 mult;
    float gamma;
    float contrast;
    float saturation;
    float exposure;
    uint32_t _pad1;
} pt_params;

/* one closest-hit query: pathtracer.cu:236-266 (map) */
typedef struct pt_ray { pt_float3 o; float _pad0; pt_float3 d; float _pad1; } pt_ray;      /* 32 bytes */
typedef struct pt_hit { float t; uint32_t tri; } pt_hit;   /* tri = original triangle index, 0xFFFFFFFF and t = 1e10 on a miss */

typedef struct pt_counters {
    uint64_t rays;          /* closest-hit queries: camera, bounce and shadow-walk segments */
    uint64_t samples;       /* renderPixel evaluations */
    uint64_t box_tests;     /* ray/box tests, 1 + 2 per expanded interior node per ray */
    uint64_t tri_tests;     /* ray/triangle tests */
    uint64_t shade_vertices;/* surface or medium events shaded */
    uint64_t env_samples;   /* envmap importance samples drawn */
    uint64_t tex_taps;      /* bilinear lookups into 4-channel textures (materials, environment radiance) */
    uint64_t tex_taps_r;    /* bilinear lookups into 1-channel textures (roughness/metallic/emissive maps, environment CDF) */
    uint64_t wide_nodes;    /* pt_enable_counters(ctx, 2) only: wide nodes the production traversal visited (box_tests and tri_tests stay 0 then) */
    uint64_t wide_tris;     /* pt_enable_counters(ctx, 2) only: triangle tests of the production traversal */
    uint64_t path_rays;     /* the closest-hit queries among `rays` that extend a path (camera and bounce rays); the rest are segments of visibility walks */
    double   last_ms;       /* device time of the last render / trace call (HIP events) */
} pt_counters;

/* The PODs are binary-compatible with the reference's structs as nvcc lays them out (structs.cuh:111-181). */
#ifdef __cplusplus
#define PT_SIZE_CHECK(T, N) static_assert(sizeof(T) == N, #T " must be " #N " bytes")
#else
#define PT_SIZE_CHECK(T, N) _Static_assert(sizeof(T) == N, #T " must be " #N " bytes")
#endif
PT_SIZE_CHECK(pt_triangle, 112);
PT_SIZE_CHECK(pt_bvh_node, 32);
PT_SIZE_CHECK(pt_material, 176);
PT_SIZE_CHECK(pt_params, 152);
PT_SIZE_CHECK(pt_ray, 32);
PT_SIZE_CHECK(pt_hit, 8);
#undef PT_SIZE_CHECK

typedef struct pt_ctx pt_ctx;

const char* pt_last_error(void);

/* ---- device boundary: pathtracer.cuh:48-54 ---- */
int pt_create(pt_ctx** out, int device_id);
void pt_destroy(pt_ctx* ctx);                                           /* endCuda, pathtracer.cu:101 */
int pt_set_stream(pt_ctx* ctx, void* hip_stream);                       /* launch on a caller-owned stream (NULL = the context's own) */
int pt_init(pt_ctx* ctx, const pt_params* params);                      /* initCuda, pathtracer.cu:37: sizes accum/display from windowSize */
int pt_upload_triangles(pt_ctx* ctx, const int* tri_idx, const pt_triangle* tris, int n_tri);   /* transferTriangles, pathtracer.cu:66 */
int pt_upload_materials(pt_ctx* ctx, const pt_material* mats, int n_mat);                       /* transfertMaterials, pathtracer.cu:87 */
int pt_upload_bvh(pt_ctx* ctx, const pt_bvh_node* nodes, int root_idx, int nodes_used, int n_tri); /* transfertBVH, pathtracer.cu:92; re-flattened on upload */
int pt_build_bvh(pt_ctx* ctx, int flavour);                             /* buildBVH (BVH_builder.cpp:115) on the uploaded triangles, then upload */
int pt_render_frame(pt_ctx* ctx, pt_params* params);                    /* render, pathtracer.cu:602: one sample per pixel, RGBA8 into params->pixelBuffer */

/* ---- textures and sky: image_loader.h:5-89, envmap.cpp:37-94 ---- */
int pt_render_frame_async(pt_ctx* ctx, pt_params* params);              /* pt_render_frame without the wait (no counterpart in the reference, whose render() blocks in cudaMemcpy, pathtracer.cu:613-615): the image goes to one of two pinned host buffers on a copy stream while the caller goes on; at most two frames in flight; params->pixelBuffer is not written */
int pt_frame_fetch(pt_ctx* ctx, uint8_t* rgba_out, const uint8_t** pinned_out, int* frame_index_out); /* the oldest frame in flight: waits for its copy; copies it to rgba_out and/or returns the pinned buffer (valid until the second pt_render_frame_async from now) */
int pt_set_frame_lookahead(pt_ctx* ctx, int max_frames);               /* pt_render_frame: while the caller keeps asking for the next frame of one view, up to max_frames (default 32, at most 64; 1 = never) are rendered per batch and each joins the accumulator when its call comes: same image per call, fewer ramp-ups */
int pt_upload_texture(pt_ctx* ctx, const float* texels, int w, int h, int comps, uint64_t* handle_out); /* comps 4 = genTexture, 1 = genTexture_float; rows already V-flipped */
int pt_set_texture_packing(pt_ctx* ctx, int on);                        /* on (default): textures whose values are all exactly k/255 are stored as bytes; the sampled values are the same */
int pt_upload_envmap(pt_ctx* ctx, const float* rgba, const float* cdf, int w, int h, float sum, pt_params* params_io); /* generateCUDAenvmap + transfertToParams */
int pt_set_sky_mode(pt_ctx* ctx, int mode);
int pt_set_env_guides(pt_ctx* ctx, int on, int* verified_out);           /* on (default 1): sampleEnvMap's two bisections (envmap_cuda.cu:28-57) are answered from guide tables (one over the whole range of the random number that names row and column range at once; a row table and per-row column tables behind it) when a check of every breakpoint at upload found the results identical; 2 = without the whole-range table, 0 = plain bisections. With 1 or 2 the values a light sample returns for the texel it lands on (radiance, direction, pdf: envmap_cuda.cu:62-73, functions of the texel alone) are read from a table computed per texel at upload by the same code; with 0 they are computed per sample. verified_out: bit 0 the row/column tables passed for the current map, bit 1 the whole-range table */

/* ---- headless rendering (replaces the window loop, main.cpp:247-296) ---- */
int pt_set_mode(pt_ctx* ctx, int mode);                                 /* PT_MODE_*; both modes give identical accumulators */
int pt_set_pool(pt_ctx* ctx, int slots, int chunk_frames);              /* samples in flight at most (default 134217728 = 2^27: 456 bytes of queue arrays each, allocated for what a render needs and never for more than its samples; a 64-spp 1080p render holds 61 GB) and frames per staging chunk (default 512; one float per pixel and frame of a chunk is held, and each chunk pays one ramp-up and run-down of the queues) */
int pt_set_drain(pt_ctx* ctx, int max_rays);                            /* wavefront mode: once every sample is started and at most max_rays paths remain, finish them in one kernel (0 = never) */
int pt_wavefront_stats(pt_ctx* ctx, uint64_t* iterations_out, uint64_t* drains_out); /* wavefront mode: stage-kernel iterations issued and drain kernels run since the context was created (either pointer may be null) */
int pt_set_lanes(pt_ctx* ctx, int lanes);                                /* wavefront mode: number of concurrent lanes (streams), each with a share of the samples in flight and of the pixels: 1 (default) or 2; same accumulator bits */
int pt_set_extend_tuning(pt_ctx* ctx, int refill_lanes, int leaf_vote, int blocks_per_cu); /* traversal kernel: idle lanes before a refill, lanes waiting at leaves before a leaf pass, resident workgroups per CU */
int pt_set_node_width(pt_ctx* ctx, int width, int leaf_merge);         /* production traversal: 4-wide (default) or binary compact nodes; subtrees of <= leaf_merge triangles (default 2) become one wide leaf at the next BVH upload; a tree too deep for the 4-wide stack uses binary */
int pt_set_max_bounces(pt_ctx* ctx, int max_bounces);                   /* pathtracer.cu:405 hard-codes 512 */
int pt_render_headless(pt_ctx* ctx, const pt_params* params, int frame_begin, int frame_end); /* frames [begin,end) are ADDED to the accumulator in frame order (the += of pathtracer.cu:585-586), also when frame_begin is 0: ranges compose across calls and across GPUs. The overwrite the reference does for frame 0 and for preview frames (pathtracer.cu:583-584) is the caller's pt_clear_accum; pt_render_frame does it itself */
int pt_read_accum(pt_ctx* ctx, float* rgb_out);                         /* 3*W*H floats */
int pt_accum_device_ptr(pt_ctx* ctx, float** accum_dev_out);            /* device address of the fp32 accumulator (for collectives) */
int pt_bind_accum(pt_ctx* ctx, float* accum_dev);                       /* render into a caller-owned device buffer of 3*W*H floats (NULL = the context's own) */
int pt_clear_accum(pt_ctx* ctx);                                        /* zero the accumulator (what frame 0 does at pathtracer.cu:583-584) */
int pt_write_accum(pt_ctx* ctx, const float* rgb);                      /* restore an accumulator saved with pt_read_accum */
int pt_sync(pt_ctx* ctx);                                               /* wait for the context's stream; updates last_ms */
int pt_resolve_display(pt_ctx* ctx, int n_frames, uint8_t* rgba_out);   /* accum/n -> AgX -> RGBA8 (pathtracer.cu:588-599); rgba_out may be NULL */
int pt_save_png(const char* path, const uint8_t* rgba, int w, int h);   /* saveToFile, main.cpp:96-114 */
int pt_resolve_display_post(pt_ctx* ctx, int n_frames, const pt_params* params, uint8_t* rgba_out); /* pt_resolve_display + the window's post-process shader, post_process.glsl:19-23 (exposure, mult, saturation, gamma of params) */
int pt_save_pfm(const char* path, const float* rgb, int w, int h);       /* fp32 accumulator to a portable float map: checkpoint of a progressive render (the reference has none; it saves PNG only, main.cpp:96-114) */
int pt_save_exr(const char* path, const float* rgb, int w, int h, float scale); /* rgb * scale (1 / frames: linear radiance) as an uncompressed 32-bit float scan-line OpenEXR file; the reference only saves the tone-mapped PNG (main.cpp:96-114) */
int pt_load_pfm(const char* path, float** rgb_out, int* w, int* h);      /* read it back (free with pt_free); feed pt_write_accum to resume */

/* ---- several GPUs in one process (the reference is single-GPU: pathtracer.cu:48 cudaSetDevice(0)) ----
 * A group holds one initialised context per GPU, each with the same scene uploaded. pt_group_render shards frames
 * [begin, end) by sample: context r of n renders pt_group_frames' range on its GPU (one host thread each); contexts 1..
 * are cleared first, and one RCCL reduce over xGMI adds every accumulator onto context 0's, which keeps what it held
 * (so calls compose like pt_render_headless). The sum order differs from one GPU's frame order: equal to rounding
 * (1e-4 relative), not bit for bit. Resolve, read or save through context 0. RCCL is loaded when a group of more than
 * one GPU is created. */
typedef struct pt_group pt_group;
int pt_group_create(pt_ctx** ctxs, int n, pt_group** out);
int pt_group_frames(int frame_begin, int frame_end, int n, int r, int* first_out, int* last_out); /* [begin + r*F/n, begin + (r+1)*F/n), F = end - begin */
int pt_group_render(pt_group* g, const pt_params* params, int frame_begin, int frame_end);        /* returns when every GPU is done and context 0 holds the sum */
void pt_group_destroy(pt_group* g);                                      /* the contexts stay the caller's */

/* ---- closest-hit batch (test and measurement entry) ---- */
int pt_trace_closest(pt_ctx* ctx, const pt_ray* rays, int n, pt_hit* hits_out);
int pt_trace_closest_dev(pt_ctx* ctx, const pt_ray* rays_dev, int n, pt_hit* hits_dev);
int pt_trace_closest_fast(pt_ctx* ctx, const pt_ray* rays, int n, pt_hit* hits_out);      /* same results through the traversal kernel the renderer uses (persistent waves, compact nodes) */
int pt_trace_closest_fast_dev(pt_ctx* ctx, const pt_ray* rays_dev, int n, pt_hit* hits_dev);
/* No counterpart in the reference. pt_trace_closest_fast_dev for callers that cannot allocate (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the same kernels and results, with
 * every intermediate (packed rays, hits, block lists, counters) in a device block of about 12 MiB that is part
 * of the library image on each GPU, so no allocating call is made. At most PT_NOALLOC_MAX_RAYS rays per call
 * (larger n is PT_ERR_ARG). The block is shared by every context on a GPU and keeps the last call's contents:
 * calls that overlap in time on one GPU (different streams or contexts) race; calls in stream order do not. */
#define PT_NOALLOC_MAX_RAYS 262144
int pt_trace_closest_noalloc_dev(pt_ctx* ctx, const pt_ray* rays_dev, int n, pt_hit* hits_dev);
int pt_trace_closest_counted(pt_ctx* ctx, const pt_ray* rays, int n, pt_hit* hits_out, uint32_t* box_tests_out, uint32_t* tri_tests_out);

int pt_get_counters(pt_ctx* ctx, pt_counters* out);
int pt_reset_counters(pt_ctx* ctx);
int pt_enable_counters(pt_ctx* ctx, int on);                            /* instrumented kernels (slower); off by default. 1: the reference-order binary walk is traced and its box and triangle tests counted (equal to the reference's counts, pathtracer.cu:207,230); 2: the production walk over the wide nodes is traced and its own node visits and triangle tests counted (wide_nodes, wide_tris) */
int pt_stage_timing(pt_ctx* ctx, int on, double* extend_ms_out, uint64_t* extend_launches_out); /* HIP-event time of the extend kernel: returns the totals since the last call, then switches timing on/off and clears them */
#define PT_STAGE_KERNELS 6
int pt_stage_times(pt_ctx* ctx, double* ms_out, uint64_t* iterations_out);  /* HIP-event times of the stage kernels accumulated since timing was last switched on: ms_out[PT_STAGE_KERNELS] = extend, path hit, (not in use: zero), miss, shadow, bsdf (on the launch streams; with two lanes the kernels of one lane run beside the other's, so the durations do not add up to wall time) and the stage iterations they cover; call before pt_stage_timing clears them */
int pt_debug_words(pt_ctx* ctx, unsigned long long* out16);                 /* diagnostic builds only: 16 statistics words of the traversal kernel, zeroed on read (all zero in a regular build) */
int pt_scene_bytes(pt_ctx* ctx, uint64_t* node_bytes, uint64_t* tri_bytes); /* bytes of the nodes the production traversal walks (4-wide, or quantized binary) and of the leaf-ordered triangle records */
int pt_detmath_probe(pt_ctx* ctx, const float* x, const float* y, int n, float* out12); /* device evaluation of include/pt_detmath.h, 12 results per input pair */
int pt_bsdf_probe(pt_ctx* ctx, const pt_material* mats, const int* inside, const float* V3, const float* N3, const float* L3, const uint32_t* pixel, int frame, int n, float* out12); /* device evaluation of the BSDF alone, item i on material record i: DisneyEval_2 (disney.cuh:234-347) for (V, N, L) -> f.xyz, pdf; DisneySample_2 (disney.cuh:348-460) for (V, N) with the random stream of (pixel[i], frame) -> L.xyz, f.xyz, pdf, inside flag afterwards */
int pt_unorm8_probe(pt_ctx* ctx, float* out256);                         /* device evaluation of the byte -> float conversion of packed textures */

/* ---- host-side producers (no GPU needed) ---- */
typedef struct pt_host_scene pt_host_scene;
/* readObjFile, file_reader.h:217; model_dir replaces the hard-wired "assets/models/" prefix (file_reader.h:51,147) */
int pt_read_obj(const char* obj_path, const char* model_dir, const char* sub_dir, pt_host_scene** out);
int pt_read_obj_general(const char* obj_path, const char* model_dir, const char* sub_dir, float scale, pt_host_scene** out); /* same output for files the reference reader (file_reader.h:217-425) cannot take: polygons, mixed corner syntax, negative indices, more vertices than faces, no normals (face normal); scale 100 is the reference's */
void pt_host_scene_free(pt_host_scene* s);
int pt_host_scene_counts(const pt_host_scene* s, int* n_tri, int* n_mat, int* n_tex);
pt_triangle* pt_host_scene_triangles(pt_host_scene* s);
int* pt_host_scene_tri_idx(pt_host_scene* s);
pt_material* pt_host_scene_materials(pt_host_scene* s);
const char* pt_host_scene_material_name(const pt_host_scene* s, int i);
/* texture i (0-based) as requested by the MTL: path, kind (4 = RGBA, 1 = R). Material texture fields hold i+1 until uploaded. */
int pt_host_scene_texture_info(const pt_host_scene* s, int i, const char** path, int* comps);

/* buildBVH, BVH_builder.cpp:115: fills tris[i].origin, permutes tri_idx, writes up to 2n-1 nodes */
int pt_host_build_bvh(pt_triangle* tris, int* tri_idx, int n_tri, int flavour, pt_bvh_node* nodes_out, int* nodes_used_out);

/* load_hdr_float4 (hdr_loader.h:253) + envMap::buildCDF (envmap.cpp:11-28). Caller frees with pt_free. */
int pt_load_hdr(const char* path, float** rgba_out, int* w, int* h);
int pt_load_texture_png(const char* path, int comps, float** texels_out, int* w, int* h); /* genTexture (comps 4) / genTexture_float (comps 1), image_loader.h:5-89, for PNG files: rows flipped, values / 255; free with pt_free, hand to pt_upload_texture */
int pt_load_texture_image(const char* path, int comps, float** texels_out, int* w, int* h); /* the same for a PNG or a JPEG file (by signature): baseline and progressive Huffman JPEG, 1 or 3 components; sf::Image::loadFromFile, image_loader.h:8,52 */
int pt_build_env_cdf(const float* rgba, int w, int h, float* cdf_out, float* sum_out);
void pt_free(void* p);

/* camera basis from cameraAngle (pathtracer.cu:568-571), computed once on the host */
int pt_camera_basis(const pt_params* params, pt_float3* ww, pt_float3* uu, pt_float3* vv);

#ifdef __cplusplus
}
#endif
#endif /* PT_CAPI_H */
