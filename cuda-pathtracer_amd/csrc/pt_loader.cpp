// Host-side asset ingest for the render path: triangulated OBJ + MTL with the PBR keys, Radiance
// RGBE environment maps and their sampling table, PNG screenshots.
//
// Behaviour follows the reference loaders so that a scene loads to the same arrays:
//   OBJ/MTL  "Cuda Pathtracer/core/file_reader.h:35-425"
//   HDR      "Cuda Pathtracer/core/hdr_loader.h:47-281"
//   CDF      "Cuda Pathtracer/core/envmap.cpp:11-28"
// including its quirks (single-space token splitting, the face layout chosen by whether any vn/vt
// line was seen so far, vertex indices above the triangle count reset to 0, usemtl names read from
// column 7, unknown material names mapped to material 0, vertices scaled by 100).  Where the
// reference would read out of bounds or throw, this loader reports an error instead.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <fstream>
#include <string>
#include <vector>
#include <zlib.h>
#include "pt_internal.h"

namespace ptamd {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

void default_material(pt_material& m)     // structs.cuh:75-94
{
    memset(&m, 0, sizeof(m));
    m.roughness = 0.01f;
    m.ior = 1.5f;
    m.alpha = 1.f;
}

namespace {

// One text line with the reference's token rule (file_reader.h:35-46): a token runs from a start
// column to the next occurrence of a stop character; the cursor is left ON the stop character (or on
// the last column when the stop character does not occur; untouched when the start is past the end).
struct Line {
    const std::string& s;
    int end = 0;
    explicit Line(const std::string& str) : s(str) {}
    std::string upto(int first, char stop)
    {
        std::string out;
        int n = (int)s.size();
        for (int i = first; i < n; i++) {
            end = i;
            if (s[i] == stop) return out;
            out.push_back(s[i]);
        }
        return out;
    }
    std::string next(char stop) { return upto(end + 1, stop); }
    std::string key() { end = 0; return upto(0, ' '); }
};

bool to_float(const std::string& tok, float& v)
{
    const char* c = tok.c_str();
    char* e = nullptr;
    v = strtof(c, &e);
    return e != c;
}
bool to_int(const std::string& tok, int& v)
{
    const char* c = tok.c_str();
    char* e = nullptr;
    long l = strtol(c, &e, 10);
    v = (int)l;
    return e != c;
}

struct MtlParse {
    HostScene& sc;
    std::string dir;
    int request_texture(const std::string& rel, int comps)
    {
        sc.textures.push_back(TextureRequest{dir + rel, comps});
        return (int)sc.textures.size();      // handle placeholder = index + 1
    }
};

bool read3(Line& ln, pt_float3& v)
{
    return to_float(ln.next(' '), v.x) && to_float(ln.next(' '), v.y) && to_float(ln.next('\n'), v.z);
}

// file_reader.h:48-208
int read_mtl(const std::string& path, const std::string& tex_dir, HostScene& sc)
{
    std::ifstream f(path);
    if (!f.is_open()) { set_error("cannot open material library %s", path.c_str()); return PT_ERR_IO; }
    MtlParse mp{sc, tex_dir};
    pt_material cur; default_material(cur);
    std::string cur_name;
    std::string text;
    int lineno = 0;
    while (std::getline(f, text)) {
        lineno++;
        Line ln(text);
        std::string k = ln.key();
        bool ok = true;
        if (k == "newmtl") {
            if (!cur_name.empty()) { sc.mats.push_back(cur); sc.mat_names.push_back(cur_name); default_material(cur); }
            cur_name = ln.next('\n');
        } else if (k == "Kd") { ok = read3(ln, cur.baseColor); }
        else if (k == "Ke") { ok = read3(ln, cur.emissive); }
        else if (k == "d") { float v; ok = to_float(ln.next('\n'), v); if (ok) cur.specTrans = (float)(1. - v); }
        else if (k == "Tf") { pt_float3 v; ok = read3(ln, v); if (ok) cur.medium.sigmaA = pt_float3{1.f - v.x, 1.f - v.y, 1.f - v.z}; }
        else if (k == "Ts") { ok = read3(ln, cur.medium.sigmaS); }
        else if (k == "Pg") { ok = to_float(ln.next(' '), cur.medium.G); }
        else if (k == "Ni") { ok = to_float(ln.next(' '), cur.ior); }
        else if (k == "Pr") { ok = to_float(ln.next('\n'), cur.roughness); }
        else if (k == "Pm") { ok = to_float(ln.next('\n'), cur.metallic); }
        else if (k == "Ps") { ok = to_float(ln.next('\n'), cur.sheen); }
        else if (k == "aniso") { ok = to_float(ln.next('\n'), cur.anisotropic); }
        else if (k == "map_Kd") { cur.diffuseTexture = mp.request_texture(ln.next('\n'), 4); cur.useTexture = 1; }
        else if (k == "map_Pr") { cur.roughnessTexture = mp.request_texture(ln.next('\n'), 1); cur.use_mapPr = 1; }
        else if (k == "map_Pm") { cur.metallicTexture = mp.request_texture(ln.next('\n'), 1); cur.use_mapPm = 1; }
        else if (k == "map_Bump") {
            ln.next(' '); ln.next(' ');                      // skips "-bm <scale>" (file_reader.h:165-166)
            cur.normalTexture = mp.request_texture(ln.next('\n'), 4); cur.use_mapNor = 1;
        }
        else if (k == "map_Ke") { cur.emissiveTexture = mp.request_texture(ln.next('\n'), 1); cur.use_mapKe = 1; }
        if (!ok) { set_error("%s:%d: malformed number", path.c_str(), lineno); return PT_ERR_IO; }
    }
    sc.mats.push_back(cur);
    sc.mat_names.push_back(cur_name);
    return PT_OK;
}

inline pt_float3 unit3(pt_float3 a)       // float3_header.cuh:72-76 with rsqrtf = 1/sqrtf
{
    float s = 1.0f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return pt_float3{a.x * s, a.y * s, a.z * s};
}

struct Corner { int v[3]; };

} // namespace

// file_reader.h:217-425
int read_obj(const char* obj_path, const char* model_dir, const char* sub_dir, HostScene& sc)
{
    const float size = 100.f;
    std::ifstream f(obj_path);
    if (!f.is_open()) { set_error("cannot open model %s", obj_path); return PT_ERR_IO; }
    std::string mdir = model_dir ? model_dir : "";
    if (!mdir.empty() && mdir.back() != '/') mdir.push_back('/');
    std::string sdir = sub_dir ? sub_dir : "";
    std::vector<pt_float3> pts, nrm;
    std::vector<pt_float2> uvs;
    std::vector<Corner> fpos, fnrm, fuv;
    std::vector<int> fmat;
    bool anyN = false, anyT = false;
    int cur_mat = 0;
    std::string text;
    int lineno = 0;
    auto idx_tok = [](const std::string& tok, int& dst) { int v; if (!tok.empty() && to_int(tok, v)) dst = v - 1; };
    while (std::getline(f, text)) {
        lineno++;
        Line ln(text);
        std::string k = ln.key();
        if (k == "f") {
            Corner p{{0, 0, 0}}, t{{0, 0, 0}}, n{{0, 0, 0}};
            if (anyN) {
                idx_tok(ln.next('/'), p.v[0]); idx_tok(ln.next('/'), t.v[0]); idx_tok(ln.next(' '), n.v[0]);
                idx_tok(ln.next('/'), p.v[1]); idx_tok(ln.next('/'), t.v[1]); idx_tok(ln.next(' '), n.v[1]);
                idx_tok(ln.next('/'), p.v[2]); idx_tok(ln.next('/'), t.v[2]); idx_tok(ln.next('\n'), n.v[2]);
            } else if (anyT) {
                idx_tok(ln.next('/'), p.v[0]); idx_tok(ln.next(' '), t.v[0]);
                idx_tok(ln.next('/'), p.v[1]); idx_tok(ln.next(' '), t.v[1]);
                idx_tok(ln.next('/'), p.v[2]); idx_tok(ln.next('\n'), t.v[2]);
            } else {
                idx_tok(ln.next(' '), p.v[0]); idx_tok(ln.next(' '), p.v[1]); idx_tok(ln.next('\n'), p.v[2]);
            }
            fpos.push_back(p); fnrm.push_back(n); fuv.push_back(t); fmat.push_back(cur_mat);
        } else if (k == "v") {
            pt_float3 p;
            if (!read3(ln, p)) { set_error("%s:%d: malformed vertex", obj_path, lineno); return PT_ERR_IO; }
            pts.push_back(pt_float3{p.x * size + 0.f, p.y * size + 0.f, p.z * size + 0.f});
        } else if (k == "vn") {
            pt_float3 p;
            if (!read3(ln, p)) { set_error("%s:%d: malformed normal", obj_path, lineno); return PT_ERR_IO; }
            nrm.push_back(p); anyN = true;
        } else if (k == "vt") {
            pt_float2 c;
            if (!(to_float(ln.next(' '), c.x) && to_float(ln.next('\n'), c.y))) { set_error("%s:%d: malformed texcoord", obj_path, lineno); return PT_ERR_IO; }
            uvs.push_back(c); anyT = true;
        } else if (k == "usemtl") {
            std::string name = ln.upto(7, '\n');
            cur_mat = 0;
            for (size_t i = 0; i < sc.mat_names.size(); i++) if (sc.mat_names[i] == name) { cur_mat = (int)i; break; }
        } else if (k == "mtllib") {
            std::string lib = ln.next('\n');
            // the reference replaces the material list on every mtllib (file_reader.h:185-193)
            sc.mats.clear(); sc.mat_names.clear();
            int rc = read_mtl(mdir + sdir + lib, mdir + sdir, sc);
            if (rc != PT_OK) return rc;
        }
    }
    int nbTri = (int)fpos.size();
    if (nbTri == 0) { set_error("%s: no faces", obj_path); return PT_ERR_IO; }
    if (sc.mats.empty()) { pt_material m; default_material(m); sc.mats.push_back(m); sc.mat_names.push_back(""); }
    sc.tris.assign(nbTri, pt_triangle{});
    sc.tri_idx.resize(nbTri);
    for (int i = 0; i < nbTri; i++) {
        int a[3], n[3], t[3];
        for (int c = 0; c < 3; c++) {
            a[c] = fpos[i].v[c]; n[c] = fnrm[i].v[c]; t[c] = fuv[i].v[c];
            if (a[c] > nbTri) a[c] = 0;                                   // file_reader.h:386-397
            if (n[c] < 0) n[c] = (int)nrm.size() - n[c] - 1;              // :398-400
            if (a[c] < 0) a[c] = (int)pts.size() - a[c] - 1;              // :402-404
            if (a[c] < 0 || a[c] >= (int)pts.size()) { set_error("%s: face %d references vertex %d of %zu", obj_path, i, a[c] + 1, pts.size()); return PT_ERR_IO; }
            if (anyN && (n[c] < 0 || n[c] >= (int)nrm.size())) { set_error("%s: face %d references normal %d of %zu", obj_path, i, n[c] + 1, nrm.size()); return PT_ERR_IO; }
            if (anyT && (t[c] < 0 || t[c] >= (int)uvs.size())) { set_error("%s: face %d references texcoord %d of %zu", obj_path, i, t[c] + 1, uvs.size()); return PT_ERR_IO; }
        }
        pt_triangle& T = sc.tris[i];
        memset(&T, 0, sizeof(T));
        T.a = pts[a[0]]; T.b = pts[a[1]]; T.c = pts[a[2]];
        T.matIndex = fmat[i];
        if (anyN) { T.nA = unit3(nrm[n[0]]); T.nB = unit3(nrm[n[1]]); T.nC = unit3(nrm[n[2]]); }
        if (anyT) { T.tA = uvs[t[0]]; T.tB = uvs[t[1]]; T.tC = uvs[t[2]]; }
        sc.tri_idx[i] = i;
    }
    return PT_OK;
}

// ---------------------------------------------------------------- general OBJ reader
// The reference reader above accepts only what the reference accepts (triangles, one face syntax per file,
// the index reset of file_reader.h:386-397). This one reads what exporters actually write: polygons (fan
// triangulation), "v", "v/vt", "v//vn" and "v/vt/vn" corners mixed freely, negative (relative) indices,
// any whitespace, several material libraries; a corner without a normal gets the face normal. Same output
// records, same material library reader.
int read_obj_general(const char* obj_path, const char* model_dir, const char* sub_dir, float scale, HostScene& sc)
{
    std::ifstream f(obj_path);
    if (!f.is_open()) { set_error("cannot open model %s", obj_path); return PT_ERR_IO; }
    std::string mdir = model_dir ? model_dir : "";
    if (!mdir.empty() && mdir.back() != '/') mdir.push_back('/');
    std::string sdir = sub_dir ? sub_dir : "";
    std::vector<pt_float3> pts, nrm;
    std::vector<pt_float2> uvs;
    int cur_mat = 0, lineno = 0;
    std::string text;
    auto trim = [](std::string v) { size_t a = v.find_first_not_of(" \t\r"); size_t b = v.find_last_not_of(" \t\r"); return a == std::string::npos ? std::string() : v.substr(a, b - a + 1); };
    struct Corner { int p, t, n; };
    std::vector<Corner> poly;
    while (std::getline(f, text)) {
        lineno++;
        const char* c = text.c_str();
        while (*c == ' ' || *c == '\t') c++;
        if (*c == '#' || *c == 0 || *c == '\r') continue;
        const char* k = c;
        while (*c && *c != ' ' && *c != '\t' && *c != '\r') c++;
        std::string key(k, c);
        if (key == "v" || key == "vn" || key == "vt") {
            float v[3] = {0.f, 0.f, 0.f};
            int got = 0;
            for (; got < 3; got++) { char* e = nullptr; v[got] = strtof(c, &e); if (e == c) break; c = e; }
            if (got < (key == "vt" ? 2 : 3)) { set_error("%s:%d: malformed %s", obj_path, lineno, key.c_str()); return PT_ERR_IO; }
            if (key == "v") pts.push_back(pt_float3{v[0] * scale, v[1] * scale, v[2] * scale});
            else if (key == "vn") nrm.push_back(pt_float3{v[0], v[1], v[2]});
            else uvs.push_back(pt_float2{v[0], v[1]});
        } else if (key == "f") {
            poly.clear();
            for (;;) {
                while (*c == ' ' || *c == '\t') c++;
                if (*c == 0 || *c == '\r') break;
                Corner q{0, 0, 0};
                bool hp = false, ht = false, hn = false;
                char* e = nullptr;
                long v = strtol(c, &e, 10);
                if (e == c) { set_error("%s:%d: malformed face", obj_path, lineno); return PT_ERR_IO; }
                q.p = (int)v; hp = true; c = e;
                if (*c == '/') {
                    c++;
                    if (*c != '/') { v = strtol(c, &e, 10); if (e != c) { q.t = (int)v; ht = true; c = e; } }
                    if (*c == '/') { c++; v = strtol(c, &e, 10); if (e != c) { q.n = (int)v; hn = true; c = e; } }
                }
                auto fix = [&](int idx, size_t count, bool have, const char* what) -> int {
                    if (!have) return -1;
                    long r = idx > 0 ? idx - 1 : (long)count + idx;
                    if (idx == 0 || r < 0 || r >= (long)count) { set_error("%s:%d: %s index %d out of range (%zu defined)", obj_path, lineno, what, idx, count); return -2; }
                    return (int)r;
                };
                q.p = fix(q.p, pts.size(), hp, "vertex"); q.t = fix(q.t, uvs.size(), ht, "texcoord"); q.n = fix(q.n, nrm.size(), hn, "normal");
                if (q.p == -2 || q.t == -2 || q.n == -2) return PT_ERR_IO;
                poly.push_back(q);
            }
            if (poly.size() < 3) { set_error("%s:%d: face with fewer than three corners", obj_path, lineno); return PT_ERR_IO; }
            for (size_t j = 1; j + 1 < poly.size(); j++) {
                const Corner q[3] = {poly[0], poly[j], poly[j + 1]};
                pt_triangle T;
                memset(&T, 0, sizeof(T));
                T.a = pts[q[0].p]; T.b = pts[q[1].p]; T.c = pts[q[2].p];
                T.matIndex = cur_mat;
                pt_float3 e1{T.b.x - T.a.x, T.b.y - T.a.y, T.b.z - T.a.z}, e2{T.c.x - T.a.x, T.c.y - T.a.y, T.c.z - T.a.z};
                pt_float3 g{e1.y * e2.z - e2.y * e1.z, e1.z * e2.x - e2.z * e1.x, e1.x * e2.y - e2.x * e1.y};
                float gl = g.x * g.x + g.y * g.y + g.z * g.z;
                pt_float3 face = gl > 0.f ? unit3(g) : pt_float3{0.f, 1.f, 0.f};
                pt_float3* ns[3] = {&T.nA, &T.nB, &T.nC};
                pt_float2* ts[3] = {&T.tA, &T.tB, &T.tC};
                for (int m = 0; m < 3; m++) {
                    if (q[m].n >= 0) { pt_float3 v = nrm[q[m].n]; *ns[m] = (v.x * v.x + v.y * v.y + v.z * v.z) > 0.f ? unit3(v) : face; }
                    else *ns[m] = face;
                    if (q[m].t >= 0) *ts[m] = uvs[q[m].t];
                }
                sc.tris.push_back(T);
            }
        } else if (key == "usemtl") {
            std::string name = trim(c);
            cur_mat = 0;
            for (size_t i = 0; i < sc.mat_names.size(); i++) if (sc.mat_names[i] == name) { cur_mat = (int)i; break; }
        } else if (key == "mtllib") {
            std::string lib = trim(c);
            int rc = read_mtl(mdir + sdir + lib, mdir + sdir, sc);
            if (rc != PT_OK) return rc;
        }
    }
    if (sc.tris.empty()) { set_error("%s: no faces", obj_path); return PT_ERR_IO; }
    if (sc.mats.empty()) { pt_material m; default_material(m); sc.mats.push_back(m); sc.mat_names.push_back(""); }
    sc.tri_idx.resize(sc.tris.size());
    for (size_t i = 0; i < sc.tris.size(); i++) sc.tri_idx[i] = (int)i;
    return PT_OK;
}

// ---------------------------------------------------------------- Radiance RGBE (the format hdr_loader.h:47-281 reads)
// The file is read whole and decoded from memory with every access bounds-checked. What is accepted matches
// the reference's reader: a "#?" signature line, FORMAT=32-bit_rle_rgbe|xyze, a resolution line "-Y h +X w",
// then h scanlines, each either flat (4 bytes per pixel) or new-style run-length coded per component (marker
// 2 2 hi lo with hi:lo == w, only for 8 <= w <= 32767); a pixel is (mantissa + 0.5) * 2^(e - 136), zero when e == 0.
namespace {
struct Bytes {
    const uint8_t* p; const uint8_t* end;
    size_t left() const { return (size_t)(end - p); }
    // next text line without its terminator; false at the end of the data
    bool line(std::string& out)
    {
        if (p >= end) return false;
        const uint8_t* nl = (const uint8_t*)memchr(p, '\n', left());
        const uint8_t* stop = nl ? nl : end;
        out.assign((const char*)p, (size_t)(stop - p));
        p = nl ? nl + 1 : end;
        return true;
    }
};

// the unsigned number that follows the axis letter on the resolution line
bool axis_size(const std::string& line, char axis, unsigned* out)
{
    size_t at = line.find(axis);
    if (at == std::string::npos || at == 0) return false;
    at++;
    while (at < line.size() && line[at] == ' ') at++;
    uint64_t v = 0; size_t digits = 0;
    while (at < line.size() && line[at] >= '0' && line[at] <= '9' && v < (1ull << 32)) { v = v * 10 + (uint64_t)(line[at] - '0'); at++; digits++; }
    if (digits == 0 || v >= (1ull << 32)) return false;
    *out = (unsigned)v;
    return true;
}

// one scanline of `width` pixels into row[4 * width]
bool scanline(Bytes& in, uint8_t* row, unsigned width, bool may_be_coded)
{
    const size_t flat = (size_t)width * 4;
    const bool coded = may_be_coded && in.left() >= 4 && in.p[0] == 2 && in.p[1] == 2 && !(in.p[2] & 128);
    if (!coded) {
        if (in.left() < flat) return false;
        memcpy(row, in.p, flat);
        in.p += flat;
        return true;
    }
    if ((((unsigned)in.p[2] << 8) | in.p[3]) != width) return false;
    in.p += 4;
    for (unsigned comp = 0; comp < 4; comp++) {
        unsigned filled = 0;
        while (filled < width) {
            if (in.left() < 2) return false;
            unsigned code = *in.p++;
            const bool run = code > 128;
            const unsigned count = run ? code - 128 : code;
            if (count > width - filled || (!run && in.left() < count)) return false;
            for (unsigned k = 0; k < count; k++) row[(size_t)(filled + k) * 4 + comp] = run ? in.p[0] : in.p[k];
            in.p += run ? 1 : count;
            filled += count;
        }
    }
    return true;
}
}

int load_hdr(const char* path, float** rgba_out, int* w_out, int* h_out)
{
    *rgba_out = nullptr; *w_out = *h_out = 0;
    std::vector<uint8_t> file;
    {
        FILE* fp = fopen(path, "rb");
        if (!fp) { set_error("cannot open %s", path); return PT_ERR_IO; }
        uint8_t chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof(chunk), fp)) > 0) file.insert(file.end(), chunk, chunk + got);
        fclose(fp);
    }
    Bytes in{file.data(), file.data() + file.size()};
    std::string text;
    if (!in.line(text) || (text.size() < 2) || (text[0] != '#' && text[1] != '?')) { set_error("%s: not a Radiance file", path); return PT_ERR_IO; }
    unsigned width = 0, height = 0;
    for (;;) {
        if (!in.line(text)) { set_error("%s: truncated header", path); return PT_ERR_IO; }
        if (text.empty() || text[0] == '#') continue;
        if (text.compare(0, 7, "FORMAT=") == 0) {
            if (text.compare(7, 15, "32-bit_rle_xyze") != 0 && text.compare(7, 15, "32-bit_rle_rgbe") != 0) { set_error("%s: unknown FORMAT", path); return PT_ERR_IO; }
        } else if (text[0] == '-' || text[0] == '+') {
            if (!axis_size(text, 'X', &width) || !axis_size(text, 'Y', &height)) { set_error("%s: bad resolution line", path); return PT_ERR_IO; }
            break;
        }
    }
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 30)) { set_error("%s: bad size", path); return PT_ERR_IO; }
    float* px = (float*)calloc((size_t)width * height, 4 * sizeof(float));
    if (!px) { set_error("out of memory"); return PT_ERR_NOMEM; }
    std::vector<uint8_t> row((size_t)width * 4);
    const bool may_be_coded = width >= 8 && width <= 0x7fff;
    for (unsigned y = 0; y < height; y++) {
        if (!scanline(in, row.data(), width, may_be_coded)) { free(px); set_error("%s: pixel data truncated", path); return PT_ERR_IO; }
        float* out = px + 4 * (size_t)y * width;
        for (unsigned x = 0; x < width; x++, out += 4) {
            const uint8_t* e = &row[4 * (size_t)x];
            if (e[3] == 0) continue;                                              // calloc left the pixel black
            uint32_t scale_bits = (uint32_t)(((int)e[3] - 9) << 23) & 0x7f800000u;    // 2^(e - 136) (hdr_loader.h:210)
            float scale; memcpy(&scale, &scale_bits, 4);
            out[0] = (float)(e[0] + 0.5f) * scale; out[1] = (float)(e[1] + 0.5f) * scale; out[2] = (float)(e[2] + 0.5f) * scale;
        }
    }
    *rgba_out = px; *w_out = (int)width; *h_out = (int)height;
    return PT_OK;
}

// envmap.cpp:7-28: running sum of luminance over all texels, row-major
void build_env_cdf(const float* rgba, int w, int h, float* cdf, float* sum)
{
    size_t n = (size_t)w * h;
    float run = 0.f;
    for (size_t i = 0; i < n; i++) {
        float wgt = (float)(0.212671 * rgba[4 * i] + 0.715160 * rgba[4 * i + 1] + 0.072169 * rgba[4 * i + 2]);
        run = i == 0 ? wgt : run + wgt;
        cdf[i] = run;
    }
    *sum = cdf[n - 1];
}

// main.cpp:96-114 saves the RGBA8 display buffer as PNG through SFML; this writes the same pixels
// with zlib (8-bit RGBA, no interlace, filter 0 on every row).
int save_png(const char* path, const uint8_t* rgba, int w, int h)
{
    if (!path || !rgba || w <= 0 || h <= 0) { set_error("save_png: bad argument"); return PT_ERR_ARG; }
    std::vector<uint8_t> raw((size_t)h * ((size_t)w * 4 + 1));
    for (int y = 0; y < h; y++) {
        raw[(size_t)y * (w * 4 + 1)] = 0;
        memcpy(&raw[(size_t)y * (w * 4 + 1) + 1], rgba + (size_t)y * w * 4, (size_t)w * 4);
    }
    uLongf clen = compressBound(raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), raw.size(), 6) != Z_OK) { set_error("save_png: deflate failed"); return PT_ERR_IO; }
    FILE* fp = fopen(path, "wb");
    if (!fp) { set_error("cannot write %s", path); return PT_ERR_IO; }
    auto be32 = [](uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
        uint8_t hdr[8]; be32(hdr, len); memcpy(hdr + 4, type, 4);
        fwrite(hdr, 1, 8, fp);
        if (len) fwrite(data, 1, len, fp);
        uLong crc = crc32(0L, (const Bytef*)type, 4);
        if (len) crc = crc32(crc, data, len);
        uint8_t c[4]; be32(c, (uint32_t)crc); fwrite(c, 1, 4, fp);
    };
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    fwrite(sig, 1, 8, fp);
    uint8_t ihdr[13]; be32(ihdr, (uint32_t)w); be32(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    fclose(fp);
    return PT_OK;
}

// ---------------------------------------------------------------- PNG decode -> texture texels
// What genTexture / genTexture_float hand to the device (image_loader.h:5-30,49-72): the image read as 8-bit
// RGBA, rows flipped (row y of the array is image row h-1-y), values / 255, four channels or the red channel
// alone. The reference decodes through sf::Image; here a PNG reader on zlib: bit depths 8 and 16 (the high
// byte), colour types grey, RGB, palette, grey+alpha, RGBA, non-interlaced.
namespace {
inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
inline int paeth(int a, int b, int c) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
}

int load_png_texture(const char* path, int comps, float** texels_out, int* w_out, int* h_out)
{
    *texels_out = nullptr; *w_out = *h_out = 0;
    if (comps != 1 && comps != 4) { set_error("load_png_texture: comps must be 1 or 4"); return PT_ERR_ARG; }
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return PT_ERR_IO; }
    std::vector<uint8_t> file;
    {
        uint8_t buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
        fclose(f);
    }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) { set_error("%s: not a PNG file", path); return PT_ERR_IO; }
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false, have_end = false;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        uint32_t len = be32(&file[pos]);
        if ((uint64_t)pos + 12 + len > file.size()) { set_error("%s: truncated chunk", path); return PT_ERR_IO; }
        const uint8_t* type = &file[pos + 4];
        const uint8_t* body = &file[pos + 8];
        if (be32(body + len) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4)) { set_error("%s: chunk checksum mismatch", path); return PT_ERR_IO; }
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) { have_end = true; break; }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_end || w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) { set_error("%s: bad PNG header", path); return PT_ERR_IO; }
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (channels == 0 || interlace != 0 || !(depth == 8 || (depth == 16 && ctype != 3) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4)) || (ctype == 0 && (depth == 1 || depth == 2 || depth == 4)))) {
        set_error("%s: unsupported PNG (colour type %d, depth %d, interlace %d)", path, ctype, depth, interlace); return PT_ERR_IO;
    }
    size_t bpp_bits = (size_t)channels * depth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) { set_error("%s: bad image data", path); return PT_ERR_IO; }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    float* px = (float*)malloc((size_t)w * h * comps * sizeof(float));
    if (!px) { set_error("out of memory"); return PT_ERR_NOMEM; }
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* line = &raw[(stride + 1) * y];
        int filter = line[0];
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = line[1 + i];
            int v = filter == 0 ? x : filter == 1 ? x + a : filter == 2 ? x + b : filter == 3 ? x + ((a + b) >> 1) : filter == 4 ? x + paeth(a, b, c) : -1;
            if (v < 0) { free(px); set_error("%s: bad row filter", path); return PT_ERR_IO; }
            cur[i] = (uint8_t)v;
        }
        float* dst = px + (size_t)(h - 1 - y) * w * comps;          // image_loader.h:15: rows flipped
        for (uint32_t x = 0; x < w; x++) {
            uint8_t r, g, b, a = 255;
            auto sample = [&](int ch) -> uint8_t {      // 8 most significant bits of channel ch of pixel x
                if (depth == 16) return cur[((size_t)x * channels + ch) * 2];
                if (depth == 8) return cur[(size_t)x * channels + ch];
                size_t bit = (size_t)x * depth; int v = (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1);
                return ctype == 3 ? (uint8_t)v : (uint8_t)(v * 255 / ((1 << depth) - 1));
            };
            if (ctype == 3) {
                size_t k = sample(0);
                if (3 * k + 2 >= plte.size()) { r = g = b = 0; }
                else { r = plte[3 * k]; g = plte[3 * k + 1]; b = plte[3 * k + 2]; }
                if (k < trns.size()) a = trns[k];
            } else if (ctype == 0) { r = g = b = sample(0); }
            else if (ctype == 4) { r = g = b = sample(0); a = sample(1); }
            else if (ctype == 2) { r = sample(0); g = sample(1); b = sample(2); }
            else { r = sample(0); g = sample(1); b = sample(2); a = sample(3); }
            if (comps == 4) { dst[4 * x] = (float)r / 255.0f; dst[4 * x + 1] = (float)g / 255.0f; dst[4 * x + 2] = (float)b / 255.0f; dst[4 * x + 3] = (float)a / 255.0f; }
            else dst[x] = (float)r / 255.0f;
        }
        prev.swap(cur);
    }
    *texels_out = px; *w_out = (int)w; *h_out = (int)h;
    return PT_OK;
}

// genTexture / genTexture_float for either format the reference's assets use: the file's signature decides.
// A JPEG becomes the same array a PNG of the decoded pixels would (rows flipped, / 255, alpha 1).
int load_image_texture(const char* path, int comps, float** texels_out, int* w_out, int* h_out)
{
    *texels_out = nullptr; *w_out = *h_out = 0;
    if (comps != 1 && comps != 4) { set_error("load_image_texture: comps must be 1 or 4"); return PT_ERR_ARG; }
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return PT_ERR_IO; }
    uint8_t sig[2] = {0, 0};
    size_t got = fread(sig, 1, 2, f);
    if (got != 2 || !(sig[0] == 0xFF && sig[1] == 0xD8)) { fclose(f); return load_png_texture(path, comps, texels_out, w_out, h_out); }
    std::vector<uint8_t> file(sig, sig + 2);
    {
        uint8_t buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
        fclose(f);
    }
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;
    int rc = decode_jpeg(file.data(), file.size(), &w, &h, rgb);
    if (rc != PT_OK) return rc;
    float* px = (float*)malloc((size_t)w * h * comps * sizeof(float));
    if (!px) { set_error("out of memory"); return PT_ERR_NOMEM; }
    for (int y = 0; y < h; y++) {
        const uint8_t* src = &rgb[(size_t)y * w * 3];
        float* dst = px + (size_t)(h - 1 - y) * w * comps;          // image_loader.h:15: rows flipped
        for (int x = 0; x < w; x++) {
            if (comps == 4) { dst[4 * x] = (float)src[3 * x] / 255.0f; dst[4 * x + 1] = (float)src[3 * x + 1] / 255.0f; dst[4 * x + 2] = (float)src[3 * x + 2] / 255.0f; dst[4 * x + 3] = 1.0f; }
            else dst[x] = (float)src[3 * x] / 255.0f;
        }
    }
    *texels_out = px; *w_out = w; *h_out = h;
    return PT_OK;
}

// Portable float map, little endian, three channels, rows bottom-up as the format prescribes: lossless
// storage of the fp32 accumulator (checkpoint / resume of a progressive render, HDR output).
int save_pfm(const char* path, const float* rgb, int w, int h)
{
    if (!path || !rgb || w <= 0 || h <= 0) { set_error("save_pfm: bad argument"); return PT_ERR_ARG; }
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("cannot write %s", path); return PT_ERR_IO; }
    fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
    bool ok = true;
    for (int y = h - 1; y >= 0 && ok; y--) ok = fwrite(rgb + (size_t)y * w * 3, sizeof(float), (size_t)w * 3, f) == (size_t)w * 3;
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("short write to %s", path); return PT_ERR_IO; }
    return PT_OK;
}

// OpenEXR, the simplest conforming form: single part, scan lines in increasing y, no compression, three 32-bit
// float channels (stored in the alphabetical order the format prescribes: B, G, R). Linear radiance for
// compositing tools: the accumulator times `scale` (1 / frames).
int save_exr(const char* path, const float* rgb, int w, int h, float scale)
{
    if (!path || !rgb || w <= 0 || h <= 0) { set_error("save_exr: bad argument"); return PT_ERR_ARG; }
    std::vector<uint8_t> hd;
    auto put = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; hd.insert(hd.end(), b, b + n); };
    auto str = [&](const char* t) { put(t, strlen(t) + 1); };
    auto i32 = [&](int32_t v) { put(&v, 4); };
    auto f32 = [&](float v) { put(&v, 4); };
    auto attr = [&](const char* name, const char* type, int32_t size) { str(name); str(type); i32(size); };
    i32(20000630); i32(2);
    attr("channels", "chlist", 3 * 18 + 1);
    for (const char* ch : {"B", "G", "R"}) { str(ch); i32(2); uint8_t lin[4] = {0, 0, 0, 0}; put(lin, 4); i32(1); i32(1); }
    hd.push_back(0);
    attr("compression", "compression", 1); hd.push_back(0);
    attr("dataWindow", "box2i", 16); i32(0); i32(0); i32(w - 1); i32(h - 1);
    attr("displayWindow", "box2i", 16); i32(0); i32(0); i32(w - 1); i32(h - 1);
    attr("lineOrder", "lineOrder", 1); hd.push_back(0);
    attr("pixelAspectRatio", "float", 4); f32(1.f);
    attr("screenWindowCenter", "v2f", 8); f32(0.f); f32(0.f);
    attr("screenWindowWidth", "float", 4); f32(1.f);
    hd.push_back(0);
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("cannot write %s", path); return PT_ERR_IO; }
    const uint64_t row_bytes = (uint64_t)w * 3 * sizeof(float), block = 8 + row_bytes;
    bool ok = fwrite(hd.data(), 1, hd.size(), f) == hd.size();
    const uint64_t first = hd.size() + (uint64_t)h * 8;
    for (int y = 0; y < h && ok; y++) { uint64_t off = first + (uint64_t)y * block; ok = fwrite(&off, 8, 1, f) == 1; }
    std::vector<float> line((size_t)w * 3);
    for (int y = 0; y < h && ok; y++) {
        const float* src = rgb + (size_t)y * w * 3;
        for (int x = 0; x < w; x++) { line[x] = src[3 * x + 2] * scale; line[(size_t)w + x] = src[3 * x + 1] * scale; line[2 * (size_t)w + x] = src[3 * x] * scale; }
        int32_t head[2] = {y, (int32_t)row_bytes};
        ok = fwrite(head, 4, 2, f) == 2 && fwrite(line.data(), sizeof(float), line.size(), f) == line.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("short write to %s", path); return PT_ERR_IO; }
    return PT_OK;
}

int load_pfm(const char* path, float** rgb_out, int* w_out, int* h_out)
{
    *rgb_out = nullptr; *w_out = *h_out = 0;
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return PT_ERR_IO; }
    char magic[3] = {0, 0, 0};
    int w = 0, h = 0;
    float scale = 0.f;
    if (fscanf(f, "%2s %d %d %f", magic, &w, &h, &scale) != 4 || strcmp(magic, "PF") != 0 || w <= 0 || h <= 0 || (uint64_t)w * h > (1ull << 30) || !(scale < 0.f)) {
        fclose(f); set_error("%s: not a little-endian three-channel PFM", path); return PT_ERR_IO;
    }
    if (fgetc(f) == EOF) { fclose(f); set_error("%s: truncated", path); return PT_ERR_IO; }       // the single whitespace after the header
    float* px = (float*)malloc((size_t)w * h * 3 * sizeof(float));
    if (!px) { fclose(f); set_error("out of memory"); return PT_ERR_NOMEM; }
    for (int y = h - 1; y >= 0; y--) {
        if (fread(px + (size_t)y * w * 3, sizeof(float), (size_t)w * 3, f) != (size_t)w * 3) { free(px); fclose(f); set_error("%s: pixel data truncated", path); return PT_ERR_IO; }
    }
    fclose(f);
    float s = -scale;
    if (s != 1.0f) for (size_t i = 0; i < (size_t)w * h * 3; i++) px[i] *= s;
    *rgb_out = px; *w_out = w; *h_out = h;
    return PT_OK;
}

} // namespace ptamd
