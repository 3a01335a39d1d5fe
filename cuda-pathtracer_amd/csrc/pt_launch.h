// Kernel launchers (defined in pt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "pt_layout.h"

namespace ptamd {
void launch_trace_closest(const DScene& S, const pt_ray* rays_dev, int n, pt_hit* hits_dev, uint32_t* nbox_dev, uint32_t* ntri_dev, hipStream_t stream);
void launch_render_samples(const DScene& S, const DCamera& cam, int frame_begin, int frame_end, float* accum_dev, DCounters* ctr_dev, hipStream_t stream);
void launch_mark_opaque_triangles(DTri* tris_dev, int n, const uint32_t* tri_info_dev, hipStream_t stream);
void launch_resolve_display(const float* accum_dev, int n_pixels, int n_frames, uint8_t* rgba_dev, hipStream_t stream);
void launch_post_process(const uint8_t* in_dev, int n_pixels, float exposure, const float* mult3, float saturation, float gamma, uint8_t* out_dev, hipStream_t stream);
void launch_wf_generate(const WfParams& w, const DCamera& cam, DCounters* ctr, hipStream_t s);
constexpr int kStageEvents = 7;
void launch_wf_iteration(const WfParams& w, const DScene& S, const DCamera& cam, DCounters* ctr, hipStream_t s, hipEvent_t* ev);
void launch_wf_flush_leftover(const WfParams& w, const DCamera& cam, DCounters* ctr, hipStream_t s);
void launch_wf_drain(const WfParams& w, const DScene& S, const DCamera& cam, int max_paths, DCounters* ctr, hipStream_t s);
void launch_wf_accumulate(const WfParams& w, const DCamera& cam, float* accum_dev, int first, int count, hipStream_t s);
void launch_wf_trace(const WfParams& w, const DScene& S, const pt_ray* rays_dev, int n, pt_hit* hits_dev, hipStream_t s);
// closest-hit batch over the library's built-in device block (no allocation anywhere on the way)
constexpr int kStaticMaxRays = PT_NOALLOC_MAX_RAYS;
constexpr int kStaticMaxDevices = 64;
hipError_t wf_static_scratch(int device, void** ptr_out);      // `device` must be the current device
hipError_t launch_wf_trace_static(void* base, const DScene& S, const pt_ray* rays_dev, int n, pt_hit* hits_dev, int refill_lanes, int leaf_vote,
                                  int extend_blocks, hipStream_t s);
int gpu_build_lbvh(const pt_triangle* tris_dev, int n, pt_bvh_node* nodes_dev, int* tri_idx_dev, hipStream_t stream);
// The traversal layouts of a tree built on the device. On success the caller owns the four arrays (hipFree).
struct GpuScene {
    DNode* nodes = nullptr; DNodeQ* qnodes = nullptr; DNode4* nodes4 = nullptr; DTri* tris = nullptr;
    size_t n_nodes = 0, n_nodes4 = 0;
    float root_min[3], root_max[3], q_origin[3], q_cell[3], q_inv_cell[3];
    int stack_need_binary = 0, stack_need_wide = 0;
    bool bad_box = false;
};
int gpu_build_scene(const pt_triangle* tris_dev, int n, int leaf_merge, bool agglomerative, bool top_sah, GpuScene* out, int* tri_idx_host, hipStream_t stream);
void launch_env_tables(const float* cdf_dev, int w, int h, float* row_table_dev, float* col_table_dev, hipStream_t stream);
void launch_env_sample_table(const DScene& S, float4* table_dev, hipStream_t stream);
void launch_env_texel_guide(const DScene& S, uint2* guide_dev, uint32_t* mismatches_dev, hipStream_t stream);
void launch_env_guides(const DScene& S, uint32_t* row_guide_dev, uint32_t* col_guide_dev, uint32_t* mismatches_dev, hipStream_t stream);
void launch_bsdf_probe(const pt_material* mats_dev, const int* inside_dev, const float* V_dev, const float* N_dev, const float* L_dev, const uint32_t* pixel_dev,
                       int frame, int n, float* out_dev, hipStream_t stream);
void launch_unorm8_probe(float* out_dev, hipStream_t stream);
void launch_detmath_probe(const float* x_dev, const float* y_dev, int n, float* out_dev, hipStream_t stream);
}
