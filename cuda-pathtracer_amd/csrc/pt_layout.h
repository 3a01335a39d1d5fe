// Device-memory layouts of the render path, shared by the host-side uploaders and the kernels.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/pt_capi.h"

namespace ptamd {

// ---------------------------------------------------------------- device data layout
// Interior node, 64 bytes = four 16-byte loads: both children's boxes and their references.
// ref >= 0: interior node index; ref < 0: leaf whose first triangle record is ~ref (the record
// carries the leaf's triangle count).
struct __attribute__((aligned(64))) DNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int left, right;
    int pad0, pad1;
};
static_assert(sizeof(DNode) == 64, "node must be one 64-byte record");

// Compact interior node for the production traversal kernel, 32 bytes = two 16-byte loads. Child boxes
// are stored as 16-bit grid coordinates over the scene box, rounded outward by two cells, so a box
// here always contains the exact one; closest hits do not depend on which (conservative) boxes guide
// the search, only the number of nodes visited does.
struct __attribute__((aligned(32))) DNodeQ {
    uint16_t q[12];       // lmin.xyz lmax.xyz rmin.xyz rmax.xyz
    int left, right;      // as DNode
};
static_assert(sizeof(DNodeQ) == 32, "compact node must be 32 bytes");

// Four-wide interior node for the production traversal kernel, 64 bytes = one cache line: up to four child
// boxes and their references (PT_TRAV_EMPTY marks an unused child). Made by pulling grandchildren of the
// binary tree up into their grandparent, largest box first, which halves the number of dependent node fetches
// per ray. The boxes are the tight bounds of the triangles below each child (recomputed, whatever boxes the
// uploaded tree carried), widened by a small safety margin and stored as 8-bit coordinates on a grid local to
// the node: a single-precision origin and a power-of-two cell size per axis, chosen so that the four boxes
// span at most 255 cells. A flat wall therefore keeps a box a few thousandths of a unit thick, thinner than
// the 0.01-unit offset at which shadow and bounce rays leave a surface (pathtracer.cu:467,480): such rays do not
// start inside the boxes of the surface they leave, and a ray that runs along a wall does not cross the box
// of every triangle of that wall. Rounded outward, the boxes always contain the exact ones, so closest hits
// do not depend on them, only the number of nodes visited does.
struct __attribute__((aligned(64))) DNode4 {
    uint32_t w[12];       // w[0..2]: origin x, y, z (float bits); w[3]: log2 of the cell size per axis, one signed byte each (x | y << 8 | z << 16);
                          // w[4..9]: byte k of each word belongs to child k: min.x, min.y, min.z, max.x, max.y, max.z in cells from the origin (a missing child: min 255, max 0, never entered); w[10..11]: 0
    int ref[4];           // as DNode
};
static_assert(sizeof(DNode4) == 64, "wide node must be 64 bytes");

// Triangle record for intersection, 48 bytes = three 16-byte loads, stored in leaf order.
// q0 = (a.xyz, original index), q1 = (b-a, triangles in this leaf [first record only]), q2 = (c-a, 0)
struct __attribute__((aligned(16))) DTri { float4 q0, q1, q2; };
static_assert(sizeof(DTri) == 48, "triangle record must be 48 bytes");

// Material record for shading, 128 bytes. Texture slots are 1-based indices into the texture table.
struct __attribute__((aligned(16))) DMat {
    float base[3]; float emis[3];
    float rough, aniso, metal, subsurf, sheen, coat, coat_rough, strans, alpha, ior;
    float G; float sigA[3]; float sigS[3];
    int tex_diffuse, tex_rough, tex_metal, tex_emis, tex_normal;
    int pad[4];
};
static_assert(sizeof(DMat) == 128, "material record must be 128 bytes");

// format 0: fp32 texels as uploaded; 1: one byte per channel, used when every uploaded value is exactly k/255
// (what an 8-bit image file decodes to), a quarter of the memory traffic for the same sampled values
struct DTex { const void* texels; int w, h, comps, format; };

struct DScene {
    const DNode* nodes;
    const DNodeQ* qnodes;         // same numbering as nodes
    const DNode4* nodes4;         // own numbering; root is node 0
    int wide;                     // 1: the production kernel walks nodes4 (its stack need fits), 0: qnodes
    int root_ref4;
    float q_origin[3], q_cell[3], q_inv_cell[3];
    const DTri* tris;             // leaf order
    const pt_triangle* tri_ref;   // original order, reference layout (shading attributes)
    const DMat* mats;
    const DTex* tex;
    const float* env_rgba;
    const float* env_cdf;
    const float* env_row_table;   // [H]   filtered CDF lookups of the row search (envmap_cuda.cu:33)
    const float* env_col_table;   // [H*W] filtered CDF lookups of the column search (envmap_cuda.cu:46)
    // Guide tables of the two searches: for a bucket of xi (xi * scale, truncated) the range of table
    // indices the search can end in, lo | hi << 16. Used only when a check of every breakpoint at upload
    // found the guided search equal to the reference's bisection (env_fast).
    const uint32_t* env_row_guide;   // [kEnvRowBuckets]
    const uint32_t* env_col_guide;   // [H][kEnvColBuckets]
    float env_row_scale, env_col_scale;
    int env_fast;
    // One table over the whole range of xi: the bucket of xi gives the row and the column range directly
    // when the bucket lies inside one row (x = row, y = first | last << 16), else x = 0xFFFFFFFF.
    const uint2* env_texel_guide;    // [kEnvTexelBuckets]
    const float4* env_sample_table;  // [H*W][2] (direction, pdf) and (radiance, 0) of a light sample that lands on the texel, or null
    float env_texel_scale;
    int env_fast2;
    int env_w, env_h;
    float env_sum;
    int sky_mode;
    float sun[3];
    float root_min[3], root_max[3];
    int root_ref;
    int n_tris;
};

constexpr int kEnvRowBuckets = 512;
constexpr int kEnvColBuckets = 128;
constexpr int kEnvTexelBuckets = 1 << 19;      // 4 MB: about one bucket per texel of a 1024x512 map

struct DCamera {
    float origin[3], ww[3], uu[3], vv[3];
    float fov, focal, dof;
    int W, H;
    int is_rendering;
    int max_bounces;
};

struct DCounters {
    unsigned long long rays, samples, box_tests, tri_tests, shade_vertices, env_samples, tex_taps, tex_taps_r, path_rays;
};

// ---------------------------------------------------------------- wavefront queues
// A sample in flight is one record in one of two queues, by what it waits for: the closest hit of its path ray
// (path queue, 48 bytes) or of a segment of the visibility walk of its light sample (shadow queue, 192 bytes: the
// path state, the light sample and the vertex the walk started from). Both are structures of 16-byte arrays indexed
// by queue position, and both are dense: every stage reads the current queues front to back and appends what it
// produces to the next ones (positions handed out by one atomic per workgroup and queue), so a wave's 64 records
// are consecutive and of one kind whatever the paths did before.
struct WfPathQueue { float4 *c0, *c1, *c2; };                  // (o, thr) (d, pdf) (result, flags + bounce, draws, sample id)
struct WfShadowQueue {
    float4 *so, *sd;                // pending segment of the visibility walk; sd.w != 0 while the record waits for its hit
    float4 *c0, *c1, *c2;           // path state as in the path queue (o is the vertex position)
    float4 *d0, *d1;                // (transmittance so far, light pdf) (light radiance, 0)
    float4 *e0, *e1, *e2, *e3, *e4; // vertex context: material values and shading normal
};

// Device counters of the wavefront loop, one 32-bit word each. The words that take atomics from every workgroup
// of a launch sit 256 bytes apart: atomics on one cache line are served one after the other.
enum : int {
    WF_CTR_NEXT_SAMPLE = 0,     // next sample id the global hand-out gives to a workgroup (in chunks)
    WF_CTR_TAIL_P = 64,         // entries appended to the next path queue so far
    WF_CTR_TAIL_S = 128,        // entries appended to the next shadow queue so far
    WF_CTR_TAIL_B = 192,        // entries of the bsdf list
    WF_CTR_TAIL_M = 576,        // entries of the miss list
    WF_CTR_SEG_MISS = 640,      // next segment of the miss list
    WF_CTR_SEG_VERTEX = 704,    // next segment of the records the resolve stage placed in the next shadow queue
    WF_CTR_RAY_NEXT = 256,      // next block of 256 rays the extend kernel hands to a wave
    WF_CTR_SEG_PATH = 320,      // next 256-record segment of the current path queue / shadow queue / bsdf list a
    WF_CTR_SEG_SHADOW = 384,    //   workgroup of the stage kernel takes
    WF_CTR_SEG_BSDF = 448,
    WF_CTR_N_P = 512,           // records in the current path queue and shadow queue (set by the planning pass)
    WF_CTR_N_S = 513,
    WF_CTR_RAY_CHUNKS = 514,    // blocks of 256 rays in both queues together; the first RAY_P_CHUNKS are path rays
    WF_CTR_RAY_P_CHUNKS = 515,
    WF_CTR_LEFTOVER = 516,      // sample ids workgroups still hold in their chunks at the last planning pass
    WF_CTR_WORDS = 712
};
// Division of a 32-bit unsigned number by a divisor fixed at launch (Granlund and Montgomery, "Division by invariant
// integers using multiplication", figure 4.1): exact for every numerator, five integer instructions instead of the
// thirty-odd of a run-time division.
struct FastDiv { uint32_t d, m, sh1, sh2; };
inline FastDiv make_fast_div(uint32_t d)
{
    FastDiv f; f.d = d ? d : 1u;
    uint32_t l = 0;
    while (l < 32 && ((uint64_t)1 << l) < (uint64_t)f.d) l++;              // ceil(log2 d)
    f.m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - (uint64_t)f.d)) / (uint64_t)f.d + 1u);
    f.sh1 = l < 1 ? l : 1u; f.sh2 = l > 0 ? l - 1u : 0u;
    return f;
}
struct WfParams {
    WfPathQueue Pc, Pn;             // current and next path queue
    WfShadowQueue Sc, Sn;           // current and next shadow queue
    float4 *hitP, *hitS;            // (t, triangle, u, v) for the records of the current queues
    uint32_t* bsdf_list;            // positions in the current shadow queue (bit 31 set: in the next one) whose light sample is finished
    uint32_t* miss_list;            // positions in the current path queue whose ray left the scene
    uint32_t* wg_chunk;             // [stage_blocks][2] cursor/end of the sample-id chunk each persistent workgroup owns
    uint32_t* counters;
    float* staging;                 // n_frames * W * H sample values of the current chunk
    int capacity;                   // entries every queue array has room for (a multiple of 256)
    int n_slots;                    // samples in flight at most (before the workgroups' leftover ids are started)
    int stage_blocks;               // workgroups of a stage launch
    uint32_t n_samples;
    int W, H, tiled, frame_begin, n_frames;
    // The pixels of the image, in sample order (8x8 tiles when the size allows, else rows), are dealt to the lanes in
    // blocks of 64: this lane owns blocks px_lane, px_lane + px_lanes, ...; n_pixels of them in all.
    int px_lane, px_lanes;
    uint32_t n_pixels;
    FastDiv div_pixels, div_tiles_x, div_W;       // divisions by n_pixels, W / 8 and W
    int refill_lanes, leaf_vote, extend_blocks;   // extend-kernel tuning (see wf_extend_kernel)
    int count_wide;                               // counting builds: tally the production (wide) walk's own work instead of the reference-order walk
};

} // namespace ptamd
