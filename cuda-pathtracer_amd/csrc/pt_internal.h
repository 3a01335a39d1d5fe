// Internal declarations shared by the host-side translation units of libptamd.
#ifndef PT_INTERNAL_H
#define PT_INTERNAL_H
#include <string>
#include <vector>
#include "../../include/pt_capi.h"

namespace ptamd {

void set_error(const char* fmt, ...);

int build_bvh_reference(pt_triangle* tris, int* idx, int n, pt_bvh_node* nodes_out, int* nodes_used_out);

struct TextureRequest { std::string path; int comps; };

struct HostScene {
    std::vector<pt_triangle> tris;
    std::vector<int> tri_idx;
    std::vector<pt_material> mats;
    std::vector<std::string> mat_names;
    std::vector<TextureRequest> textures;
};

int read_obj(const char* obj_path, const char* model_dir, const char* sub_dir, HostScene& out);
int read_obj_general(const char* obj_path, const char* model_dir, const char* sub_dir, float scale, HostScene& out);
int load_hdr(const char* path, float** rgba_out, int* w, int* h);
void build_env_cdf(const float* rgba, int w, int h, float* cdf_out, float* sum_out);
int save_png(const char* path, const uint8_t* rgba, int w, int h);
int load_png_texture(const char* path, int comps, float** texels_out, int* w, int* h);
int load_image_texture(const char* path, int comps, float** texels_out, int* w, int* h);   // PNG or JPEG by signature
int decode_jpeg(const uint8_t* data, size_t size, int* w, int* h, std::vector<uint8_t>& rgb);
int save_pfm(const char* path, const float* rgb, int w, int h);
int save_exr(const char* path, const float* rgb, int w, int h, float scale);
int load_pfm(const char* path, float** rgb_out, int* w, int* h);
void default_material(pt_material& m);

} // namespace ptamd
#endif
