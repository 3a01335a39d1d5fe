// ptamd_render — headless front end of libptamd, written against the C-ABI only (include/pt_capi.h).
//
// It walks the sequence the reference application runs at start-up and per frame, without the window:
//   initScene (main.cpp:116-152): readObjFile -> buildBVH -> initCuda -> transferTriangles ->
//                                 transfertMaterials -> transfertBVH
//   updateSkyMap (main.cpp:153-174): loadMap -> buildCDF -> generateCUDAenvmap -> transfertToParams
//   frame loop (main.cpp:274): render(params) with frameIndex advancing
//   saveToFile (main.cpp:96-114): PNG of the displayed image
// Texture files named by the material library are decoded here (PNG only) and uploaded.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <chrono>
#include <string>
#include <vector>
#include "../../include/pt_capi.h"

namespace {

void die(const char* what)
{
    std::fprintf(stderr, "ptamd_render: %s: %s\n", what, pt_last_error());
    std::exit(1);
}
void must(int rc, const char* what) { if (rc != PT_OK) die(what); }

void usage()
{
    std::fprintf(stderr,
        "usage: ptamd_render --obj FILE --dir MODEL_DIR [--subdir SUB] --hdr FILE [options]\n"
        "  --size WxH        image size (default 1920x1080)\n"
        "  --spp N           samples per pixel (default 64)\n"
        "  --bounces N       bounce cap (default 512, the reference's constant)\n"
        "  --origin X,Y,Z    camera position (default 0,1.5,5 as in main.cpp:186)\n"
        "  --angle AX,AY     camera angles in radians (default 0,0)\n"
        "  --fov F           field-of-view factor (default 1)\n"
        "  --dof S --focal D thin lens strength and focal distance (default 0, 200)\n"
        "  --general-obj     read polygons, mixed corner syntax, negative indices (default: the reference's reader)\n"
        "  --scale S         model scale with --general-obj (default 100, the reference's)\n"
        "  --preview         normal-dot-view preview instead of path tracing\n"
        "  --sunsky          sun/sky gradient instead of the environment map\n"
        "  --frames-per-call N   1 = call pt_render_frame per sample like the application; 0 = one headless call (default)\n"
        "  --post E,G,S      exposure, gamma, saturation of the window's post-process (default: off)\n"
        "  --out FILE.png    displayed image (default out.png)    --pfm FILE.pfm   fp32 accumulator\n"
        "  --exr FILE.exr    linear radiance (accumulator / frames), 32-bit float OpenEXR\n"
        "  --resume FILE.pfm --resume-frames K   continue from a saved accumulator of K frames\n"
        "  --tree KIND       host (the reference's surface-area builder, default), lbvh (radix tree built on the GPU) or\n"
        "                    ploc (agglomerative tree built on the GPU, close to the host tree in traversal cost)\n"
        "  --gpu ID          device (default 0); with --gpus the first of the devices used\n"
        "  --gpus N          render on N GPUs: the scene goes to each, GPU r renders samples [r*spp/N, (r+1)*spp/N), and one\n"
        "                    RCCL reduce sums the accumulators onto the first GPU, which writes the outputs\n");
    std::exit(2);
}

bool parse_floats(const char* s, float* out, int n)
{
    for (int i = 0; i < n; i++) {
        char* e = nullptr;
        out[i] = std::strtof(s, &e);
        if (e == s) return false;
        s = e;
        if (i + 1 < n) { if (*s != ',') return false; s++; }
    }
    return *s == 0;
}

} // namespace

int main(int argc, char** argv)
{
    std::string obj, dir, subdir, hdr, out = "out.png", pfm, exr, resume, tree = "host";
    int W = 1920, H = 1080, spp = 64, bounces = 512, gpu = 0, per_call = 0, resume_frames = 0, gpus = 1;
    float origin[3] = {0.f, 1.5f, 5.f}, angle[2] = {0.f, 0.f}, fov = 1.f, dof = 0.f, focal = 200.f, post[3] = {1.f, 1.f, 1.f};
    bool preview = false, sunsky = false, use_post = false, general = false;
    float scale = 100.f;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) usage(); return argv[++i]; };
        if (a == "--obj") obj = val();
        else if (a == "--dir") dir = val();
        else if (a == "--subdir") subdir = val();
        else if (a == "--hdr") hdr = val();
        else if (a == "--out") out = val();
        else if (a == "--pfm") pfm = val();
        else if (a == "--exr") exr = val();
        else if (a == "--resume") resume = val();
        else if (a == "--resume-frames") resume_frames = std::atoi(val());
        else if (a == "--size") { if (std::sscanf(val(), "%dx%d", &W, &H) != 2) usage(); }
        else if (a == "--spp") spp = std::atoi(val());
        else if (a == "--bounces") bounces = std::atoi(val());
        else if (a == "--tree") tree = val();
        else if (a == "--gpu") gpu = std::atoi(val());
        else if (a == "--gpus") gpus = std::atoi(val());
        else if (a == "--frames-per-call") per_call = std::atoi(val());
        else if (a == "--origin") { if (!parse_floats(val(), origin, 3)) usage(); }
        else if (a == "--angle") { if (!parse_floats(val(), angle, 2)) usage(); }
        else if (a == "--fov") fov = std::strtof(val(), nullptr);
        else if (a == "--dof") dof = std::strtof(val(), nullptr);
        else if (a == "--focal") focal = std::strtof(val(), nullptr);
        else if (a == "--post") { if (!parse_floats(val(), post, 3)) usage(); use_post = true; }
        else if (a == "--general-obj") general = true;
        else if (a == "--scale") scale = std::strtof(val(), nullptr);
        else if (a == "--preview") preview = true;
        else if (a == "--sunsky") sunsky = true;
        else usage();
    }
    if (obj.empty() || (hdr.empty() && !sunsky) || W <= 0 || H <= 0 || spp <= 0 || gpus < 1 || gpus > 64) usage();
    if (gpus > 1 && per_call == 1) { std::fprintf(stderr, "ptamd_render: --frames-per-call 1 renders on one GPU\n"); return 2; }

    // kernel parameters as main.cpp:181-202 sets them
    pt_params P;
    std::memset(&P, 0, sizeof(P));
    P.windowSize.x = W; P.windowSize.y = H;
    P.rayOrigin = pt_float3{origin[0], origin[1], origin[2]};
    P.rayDirectionZ = pt_float3{0.f, 0.f, -1.f};
    P.cameraAngle = pt_float3{angle[0], angle[1], 0.f};
    P.cameraSpeed = 720.f;
    P.focalDistance = focal; P.DOF_strenght = dof; P.fov = fov;
    P.isRendering = preview ? 0 : 1;
    P.mult = pt_float3{1.f, 1.f, 1.f};
    P.gamma = P.saturation = P.contrast = P.exposure = 1.f;
    {
        float n = 1.0f / std::sqrt(2.49f);          // normalize(1, 0.7, -1), main.cpp:202
        P.sunDirection = pt_float3{n, 0.7f * n, -n};
    }

    auto t0 = std::chrono::steady_clock::now();
    auto seconds = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };

    // ---- initScene
    pt_host_scene* hs = nullptr;
    if (general) must(pt_read_obj_general(obj.c_str(), dir.c_str(), subdir.c_str(), scale, &hs), "reading the model");
    else must(pt_read_obj(obj.c_str(), dir.c_str(), subdir.c_str(), &hs), "reading the model");
    int n_tri = 0, n_mat = 0, n_tex = 0;
    must(pt_host_scene_counts(hs, &n_tri, &n_mat, &n_tex), "model");
    std::printf("model: %d triangles, %d materials, %d texture files (%.2f s)\n", n_tri, n_mat, n_tex, seconds());
    if (tree != "host" && tree != "lbvh" && tree != "ploc") usage();
    std::vector<pt_bvh_node> nodes;
    int nodes_used = 0;
    if (tree == "host") {
        nodes.resize((size_t)2 * n_tri);
        must(pt_host_build_bvh(pt_host_scene_triangles(hs), pt_host_scene_tri_idx(hs), n_tri, PT_BVH_REFERENCE, nodes.data(), &nodes_used), "BVH build");
        std::printf("BVH: %d nodes (%.2f s)\n", nodes_used, seconds());
    }

    // textures and the environment are decoded once, then every GPU gets the whole scene
    struct Tex { float* texels; int w, h, comps; };
    std::vector<Tex> tex(n_tex);
    for (int i = 0; i < n_tex; i++) {           // handles are issued in request order, which is what the materials hold
        const char* path = nullptr;
        must(pt_host_scene_texture_info(hs, i, &path, &tex[i].comps), "texture list");
        must(pt_load_texture_image(path, tex[i].comps, &tex[i].texels, &tex[i].w, &tex[i].h), "decoding a texture");
    }
    float* env_rgba = nullptr; int ew = 0, eh = 0; float env_sum = 0.f;
    std::vector<float> cdf;
    if (!hdr.empty()) {             // updateSkyMap
        must(pt_load_hdr(hdr.c_str(), &env_rgba, &ew, &eh), "environment map");
        cdf.resize((size_t)ew * eh);
        must(pt_build_env_cdf(env_rgba, ew, eh, cdf.data(), &env_sum), "environment CDF");
    }
    std::vector<pt_ctx*> ctxs(gpus, nullptr);
    for (int g = 0; g < gpus; g++) {
        pt_ctx*& c = ctxs[g];
        must(pt_create(&c, gpu + g), "opening the GPU");
        must(pt_init(c, &P), "init");
        for (int i = 0; i < n_tex; i++) {
            uint64_t handle = 0;
            must(pt_upload_texture(c, tex[i].texels, tex[i].w, tex[i].h, tex[i].comps, &handle), "texture upload");
            if (handle != (uint64_t)(i + 1)) { std::fprintf(stderr, "ptamd_render: texture handle order changed\n"); return 1; }
        }
        must(pt_upload_triangles(c, pt_host_scene_tri_idx(hs), pt_host_scene_triangles(hs), n_tri), "triangles");
        must(pt_upload_materials(c, pt_host_scene_materials(hs), n_mat), "materials");
        if (tree == "host") must(pt_upload_bvh(c, nodes.data(), 0, nodes_used, n_tri), "BVH upload");
        else {
            must(pt_build_bvh(c, tree == "ploc" ? PT_BVH_GPU_PLOC : PT_BVH_GPU_LBVH), "BVH build on the GPU");
            if (g == 0) std::printf("BVH: built on the GPU (%s, %.2f s)\n", tree.c_str(), seconds());
        }
        if (env_rgba) must(pt_upload_envmap(c, env_rgba, cdf.data(), ew, eh, env_sum, &P), "environment upload");
        must(pt_set_sky_mode(c, sunsky ? PT_SKY_SUNSKY : PT_SKY_ENVMAP), "sky mode");
        must(pt_set_max_bounces(c, bounces), "bounce cap");
    }
    for (Tex& t : tex) pt_free(t.texels);
    if (env_rgba) pt_free(env_rgba);
    pt_host_scene_free(hs);
    pt_ctx* ctx = ctxs[0];
    pt_group* group = nullptr;
    if (gpus > 1) must(pt_group_create(ctxs.data(), gpus, &group), "grouping the GPUs");
    std::printf("scene on %d GPU%s (%.2f s)\n", gpus, gpus > 1 ? "s" : "", seconds());

    // ---- frames
    int first = 0;
    if (!resume.empty()) {
        float* acc = nullptr; int aw = 0, ah = 0;
        must(pt_load_pfm(resume.c_str(), &acc, &aw, &ah), "checkpoint");
        if (aw != W || ah != H || resume_frames <= 0) { std::fprintf(stderr, "ptamd_render: checkpoint is %dx%d, --resume-frames must be given\n", aw, ah); return 1; }
        must(pt_write_accum(ctx, acc), "restoring the accumulator");
        pt_free(acc);
        first = resume_frames;
    }
    double t_render = seconds();
    std::vector<uint8_t> rgba((size_t)W * H * 4);
    if (per_call == 1) {
        P.pixelBuffer = rgba.data();
        for (int f = first; f < first + spp; f++) { P.frameIndex = f; must(pt_render_frame(ctx, &P), "render"); }
    } else if (group) {
        must(pt_group_render(group, &P, first, first + spp), "render");
    } else {
        must(pt_render_headless(ctx, &P, first, first + spp), "render");
        must(pt_sync(ctx), "render");
    }
    t_render = seconds() - t_render;
    int total = first + spp;
    if (use_post) {
        P.exposure = post[0]; P.gamma = post[1]; P.saturation = post[2];
        must(pt_resolve_display_post(ctx, total, &P, rgba.data()), "display");
    } else {
        must(pt_resolve_display(ctx, total, rgba.data()), "display");
    }
    std::printf("rendered %d frames of %dx%d in %.3f s (%.2f Msamples/s)\n", spp, W, H, t_render, (double)W * H * spp / t_render / 1e6);
    must(pt_save_png(out.c_str(), rgba.data(), W, H), "writing the image");
    if (!pfm.empty() || !exr.empty()) {
        std::vector<float> acc((size_t)W * H * 3);
        must(pt_read_accum(ctx, acc.data()), "reading the accumulator");
        if (!pfm.empty()) must(pt_save_pfm(pfm.c_str(), acc.data(), W, H), "writing the accumulator");
        if (!exr.empty()) must(pt_save_exr(exr.c_str(), acc.data(), W, H, 1.0f / (float)total), "writing the radiance image");
    }
    pt_group_destroy(group);
    for (pt_ctx* c : ctxs) pt_destroy(c);
    return 0;
}
