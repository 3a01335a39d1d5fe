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
// q0 = (a.xyz, id word), q1 = (b-a, triangles in this leaf [first record only]), q2 = (c-a, 0). The id word is twice
// the triangle's original index, plus 1 when its material ends every visibility walk that hits it (DMat::opaque; set
// once triangles, tree and materials are all there). Ordered like the indices, so ties between equal distances fall
// as they do by index; the production traversal kernel reports the word as it is, and the shadow stage reads the mark
// off it instead of looking anything up behind the hit.
struct __attribute__((aligned(16))) DTri { float4 q0, q1, q2; };
static_assert(sizeof(DTri) == 48, "triangle record must be 48 bytes");

// Material record for shading, 128 bytes. Texture slots are 1-based indices into the texture table.
struct __attribute__((aligned(16))) DMat {
    float base[3]; float emis[3];
    float rough, aniso, metal, subsurf, sheen, coat, coat_rough, strans, alpha, ior;
    float G; float sigA[3]; float sigS[3];
    int tex_diffuse, tex_rough, tex_metal, tex_emis, tex_normal;
    // 1 when a visibility walk that hits this material always ends there (pathtracer.cu:279-295): its alpha (the
    // material's, or that of every texel of its colour map) is 1 or more, so no random number passes the alpha test,
    // and it cannot count as refractive whatever its metallic map holds (specTrans 0, or an ior away from 1)
    int opaque;
    int pad[3];
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
    const uint32_t* tri_info;     // original order: the triangle's material index, bit 31 = DMat::opaque of that material; or null.
                                  // 4 bytes a triangle stay in the caches: the material's record can be fetched beside the 112-byte shading
                                  // triangle instead of after it, and the visibility walk reads the flag with one dependent load instead of two
    const DMat* mats;
    const DTex* tex;
    int n_mats, n_tex;            // entries of the two tables
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
// (path queue, 48 bytes) or of a segment of the visibility walk of its light sample (shadow queue, 128 bytes as a
// vertex writes it, 160 with the arrays few records need: the path state, the light sample and the vertex the walk started from). Both are structures of 16-byte arrays indexed
// by queue position, and both are dense: every stage reads the current queues front to back and appends what it
// produces to the next ones (positions handed out by one atomic per workgroup and queue), so a wave's 64 records
// are consecutive and of one kind whatever the paths did before.
struct WfPathQueue { float4 *c0, *c1, *c2; };                  // (o, thr) (d, pdf) (result, flags + bounce, draws, sample id)
struct WfShadowQueue {
    float4 *so, *sd;                // pending segment of the visibility walk; so.w: the vertex's material; sd.w: 1 while the record waits for
                                    // its hit, 2 when d0 holds the walk's transmittance as well
    float4 *c0, *c1, *c2;           // path state as in the path queue (o is the vertex position), with the vertex's metallic value in the place
                                    // of the last bounce's pdf, which nothing reads at a vertex (pathtracer.cu:449-480 set the next)
    float4 *d0, *d1;                // (transmittance so far, -): written by a walk's first segment that crosses a medium, (1, 1, 1) otherwise;
                                    // (light radiance, light pdf)
    float4 *e0, *e1, *e4;           // vertex context, what differs from hit to hit: (base colour, roughness), (emission, -) only where the material
                                    // has an emission map, (shading normal, alpha); the material's other parameters are read from its record
};

// Striped queues. One 32-bit word takes about 90 returning atomic adds per microsecond from the whole chip
// (tools/micro/atomic_rate.hip: 11.4 ns each, whatever the number of workgroups; words 256 bytes apart are served
// side by side), and a stage launch over six million records reserves its output positions workgroup by workgroup:
// 25 000 atomics on one tail word are 0.29 ms, more than the memory traffic of the launch takes. So the position
// space of every queue and list is dealt in blocks of 256 to kStripes stripes (block B belongs to stripe B mod
// kStripes), each with a tail word of its own: entry l of stripe r sits at position ((l / 256) * kStripes + r) * 256
// + l mod 256. A workgroup appends what it makes of an input block to the output stripe with the input block's
// number, so a sample never leaves its stripe and no stripe ever holds more than it was dealt when the samples
// were started (the arrays need no slack beyond rounding). Consumers take whole blocks; the blocks at the end of
// the shorter stripes are partly or wholly empty.
#ifndef PT_STRIPE_SHIFT
#define PT_STRIPE_SHIFT 4
#endif
constexpr int kStripeShift = PT_STRIPE_SHIFT;
constexpr int kStripes = 1 << kStripeShift;
constexpr int kStripeWords = 64;        // words between the counters of neighbouring stripes (256 bytes)

// Device counters of the wavefront loop, one 32-bit word each. The words that take atomics from every workgroup
// of a launch sit 256 bytes apart: atomics on one cache line are served one after the other.
enum : int {
    WF_CTR_NEXT_SAMPLE = 0,     // next sample id the global hand-out gives to a workgroup (in chunks)
    WF_CTR_RAY_NEXT = 64,       // next block of 256 rays the extend kernel hands to a wave
    WF_CTR_N_P = 128,           // [kStripes] records per stripe of the current path queue (set by the planning pass)
    WF_CTR_N_S = 128 + kStripes,        // [kStripes] ... of the current shadow queue
    WF_CTR_RAY_CHUNKS = 128 + 2 * kStripes,     // blocks of 256 positions in both queues together; the first RAY_P_CHUNKS belong to the path queue
    WF_CTR_RAY_P_CHUNKS = WF_CTR_RAY_CHUNKS + 1,
    WF_CTR_LEFTOVER = WF_CTR_RAY_CHUNKS + 2,    // sample ids workgroups still hold in their chunks at the last planning pass
    // families of kStripes words, kStripeWords apart: word of stripe r = family + r * kStripeWords
    WF_CTR_TAIL_P = 192,                                        // entries appended to the next path queue so far
    WF_CTR_TAIL_S = WF_CTR_TAIL_P + kStripes * kStripeWords,    // ... to the next shadow queue
    WF_CTR_TAIL_B = WF_CTR_TAIL_S + kStripes * kStripeWords,    // entries of the bsdf list
    WF_CTR_TAIL_M = WF_CTR_TAIL_B + kStripes * kStripeWords,    // entries of the miss list
    WF_CTR_SEG_PATH = WF_CTR_TAIL_M + kStripes * kStripeWords,  // next block of the stripe a workgroup of the path-hit stage takes
    WF_CTR_SEG_VERTEX = WF_CTR_SEG_PATH + kStripes * kStripeWords,          // (not in use)
    WF_CTR_SEG_MISS = WF_CTR_SEG_VERTEX + kStripes * kStripeWords,
    WF_CTR_SEG_SHADOW = WF_CTR_SEG_MISS + kStripes * kStripeWords,
    WF_CTR_SEG_BSDF = WF_CTR_SEG_SHADOW + kStripes * kStripeWords,
    WF_CTR_WORDS = WF_CTR_SEG_BSDF + kStripes * kStripeWords
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
    int capacity;                   // entries every queue array has room for (a multiple of 256 * kStripes)
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
