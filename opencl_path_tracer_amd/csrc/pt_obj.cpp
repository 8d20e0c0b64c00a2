// pt_obj.cpp -- OBJ/MTL import with the reference's conventions (Scene::add_Obj,
// main.cpp:552-617).  The reference parses with the vendored tiny_obj_loader.h v1.0.3; this is
// an own minimal reader that reproduces the parts of its behaviour add_Obj depends on:
//   * v / f records; f accepts i, i/j, i//k, i/j/k; negative (relative) indices
//     (tiny_obj_loader.h:410-414); polygons are fan-triangulated (tiny_obj_loader.h:893-916)
//   * shapes are split on g / o (tiny_obj_loader.h:1509-1567); usemtl switches the per-face
//     material without starting a new shape (tiny_obj_loader.h:1452-1478)
//   * MTL: newmtl, Kd, Ks, Ke, Ns; every other "key value" line is kept as a string, first
//     occurrence wins (tiny_obj_loader.h:1258-1269); defaults per tiny_obj_loader.h:838-873
//   * add_Obj itself: custom keys Kn / Kk (three floats, split on single blanks, atof) and
//     Tp (atoi) are REQUIRED (main.cpp:568-571 uses .at()); x is negated, then rotate_x(pitch),
//     rotate_y(yaw), scale, translate (main.cpp:598-606); one end_Obj per shape (main.cpp:615).
// Deviations, all turning undefined behaviour of the reference into errors: a missing
// Kn/Kk/Tp, a face without usemtl (material id -1) and an empty shape return PT_EIO.
#include "pt_api.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace ptamd {
int fail_ctx(pt_context* ctx, int code, const std::string& msg);   // pt_host.cpp
}

namespace {

struct MtlRec {
    std::string name;
    float diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0}, emission[3] = {0, 0, 0};
    float shininess = 1.0f;                         // tiny_obj_loader.h:858
    std::map<std::string, std::string> unknown;
};

inline bool is_space(char c) { return c == ' ' || c == '\t'; }

bool read_lines(const std::string& path, std::vector<std::string>* lines) {
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) return false;
    std::string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::string cur;
    for (size_t i = 0; i < all.size(); ++i) {
        char c = all[i];
        if (c == '\n') { lines->push_back(cur); cur.clear(); }
        else if (c == '\r') { if (i + 1 < all.size() && all[i + 1] == '\n') ++i; lines->push_back(cur); cur.clear(); }
        else cur.push_back(c);
    }
    if (!cur.empty()) lines->push_back(cur);
    return true;
}

float parse_float(const char** tok) {
    *tok += std::strspn(*tok, " \t");
    const char* end = *tok + std::strcspn(*tok, " \t\r");
    std::string s(*tok, end);
    *tok = end;
    if (s.empty()) return 0.0f;
    char* e = nullptr;
    double d = std::strtod(s.c_str(), &e);
    if (e == s.c_str()) return 0.0f;
    return (float)d;
}

void parse_float3(float out[3], const char** tok) {
    out[0] = parse_float(tok);
    out[1] = parse_float(tok);
    out[2] = parse_float(tok);
}

bool load_mtl(const std::string& path, std::vector<MtlRec>* mats, std::map<std::string, int>* index) {
    std::vector<std::string> lines;
    if (!read_lines(path, &lines)) return false;
    MtlRec cur;
    bool have = false;
    for (const std::string& raw : lines) {
        const char* t = raw.c_str();
        t += std::strspn(t, " \t");
        if (*t == '\0' || *t == '#') continue;
        if (0 == std::strncmp(t, "newmtl", 6) && is_space(t[6])) {
            if (have || !cur.name.empty()) {
                index->insert(std::make_pair(cur.name, (int)mats->size()));
                mats->push_back(cur);
            }
            cur = MtlRec();
            t += 7;
            t += std::strspn(t, " \t");
            cur.name = std::string(t, std::strcspn(t, " \t\r"));
            have = true;
            continue;
        }
        if (t[0] == 'K' && t[1] == 'd' && is_space(t[2])) { t += 2; parse_float3(cur.diffuse, &t); continue; }
        if (t[0] == 'K' && t[1] == 's' && is_space(t[2])) { t += 2; parse_float3(cur.specular, &t); continue; }
        if (t[0] == 'K' && t[1] == 'e' && is_space(t[2])) { t += 2; parse_float3(cur.emission, &t); continue; }
        if (t[0] == 'N' && t[1] == 's' && is_space(t[2])) { t += 2; cur.shininess = parse_float(&t); continue; }
        // keys tinyobj knows but add_Obj never reads
        static const char* known[] = {"Ka", "Kt", "Tf", "Ni", "illum", "d", "Tr", "Pr", "Pm", "Ps", "Pc", "Pcr", "aniso", "anisor",
                                      "map_Ka", "map_Kd", "map_Ks", "map_Ns", "map_bump", "bump", "map_d", "disp", "refl",
                                      "map_Pr", "map_Pm", "map_Ps", "map_Ke", "norm"};
        bool skip = false;
        for (const char* k : known) {
            size_t n = std::strlen(k);
            if (0 == std::strncmp(t, k, n) && is_space(t[n])) { skip = true; break; }
        }
        if (skip) continue;
        const char* sp = std::strchr(t, ' ');
        if (!sp) sp = std::strchr(t, '\t');
        if (sp) cur.unknown.insert(std::make_pair(std::string(t, (size_t)(sp - t)), std::string(sp + 1)));
    }
    index->insert(std::make_pair(cur.name, (int)mats->size()));   // tinyobj flushes the last material unconditionally
    mats->push_back(cur);
    return true;
}

// main.cpp:72-86: split on single blanks, atof the first three fields
bool str_to_float3(const std::string& s, float out[3]) {
    std::vector<std::string> arr;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, ' ')) arr.push_back(item);
    if (arr.size() < 3) return false;
    for (int i = 0; i < 3; ++i) out[i] = (float)std::atof(arr[i].c_str());
    return true;
}

int fix_index(int idx, int n) {        // tiny_obj_loader.h:410-414
    if (idx > 0) return idx - 1;
    if (idx == 0) return 0;
    return n + idx;
}

void rot_x(float v[3], float gamma) {   // main.cpp:63-70
    gamma = gamma / 180.0f * 3.141593f;
    const double c = std::cos((double)gamma), s = std::sin((double)gamma);
    const float r1 = (float)((double)v[1] * c - (double)v[2] * s);
    const float r2 = (float)((double)v[1] * s + (double)v[2] * c);
    v[1] = r1; v[2] = r2;
}
void rot_y(float v[3], float beta) {    // main.cpp:55-62
    beta = beta / 180.0f * 3.141593f;
    const double c = std::cos((double)beta), s = std::sin((double)beta);
    const float r0 = (float)((double)v[0] * c + (double)v[2] * s);
    const float r2 = (float)(-(double)v[0] * s + (double)v[2] * c);
    v[0] = r0; v[2] = r2;
}

struct Face3 { int v[3]; int mat; };
struct Shape { std::vector<Face3> faces; };

}  // namespace

extern "C" int pt_add_obj(pt_context* ctx, const char* file, const float pos[3], const float scale[3], float pitch, float yaw) {
    using ptamd::fail_ctx;
    if (!ctx || !file || !pos || !scale) return PT_EINVAL;
    const std::string path(file);
    const std::string matpath = path.substr(0, path.find_last_of('/') + 1);   // main.cpp:553
    std::vector<std::string> lines;
    if (!read_lines(path, &lines)) return fail_ctx(ctx, PT_EIO, "cannot open OBJ file: " + path);

    std::vector<float> v;
    std::vector<MtlRec> mtls;
    std::map<std::string, int> mtl_index;
    std::vector<Shape> shapes;
    Shape shape;
    std::vector<std::vector<int>> group;   // pending faces (vertex indices) of the current material
    int material = -1;

    auto flush_group = [&]() -> bool {      // exportFaceGroupToShape with triangulate = true
        if (group.empty()) return false;
        for (const std::vector<int>& face : group) {
            if (face.size() < 3) continue;
            for (size_t k = 2; k < face.size(); ++k) {
                Face3 f;
                f.v[0] = face[0]; f.v[1] = face[k - 1]; f.v[2] = face[k];
                f.mat = material;
                shape.faces.push_back(f);
            }
        }
        return true;
    };

    for (const std::string& raw : lines) {
        const char* t = raw.c_str();
        t += std::strspn(t, " \t");
        if (*t == '\0' || *t == '#') continue;
        if (t[0] == 'v' && is_space(t[1])) {
            t += 2;
            float p[3];
            parse_float3(p, &t);
            v.push_back(p[0]); v.push_back(p[1]); v.push_back(p[2]);
            continue;
        }
        if (t[0] == 'f' && is_space(t[1])) {
            t += 2;
            t += std::strspn(t, " \t");
            std::vector<int> face;
            while (*t != '\0' && *t != '\r' && *t != '\n') {
                int idx = std::atoi(t);                       // the v of v, v/vt, v//vn, v/vt/vn
                face.push_back(fix_index(idx, (int)(v.size() / 3)));
                t += std::strcspn(t, " \t\r");
                t += std::strspn(t, " \t\r");
            }
            group.push_back(face);
            continue;
        }
        if (0 == std::strncmp(t, "usemtl", 6) && is_space(t[6])) {
            t += 7;
            t += std::strspn(t, " \t");
            std::string name(t, std::strcspn(t, " \t\r"));
            int id = -1;
            auto it = mtl_index.find(name);
            if (it != mtl_index.end()) id = it->second;
            if (id != material) { flush_group(); group.clear(); material = id; }
            continue;
        }
        if (0 == std::strncmp(t, "mtllib", 6) && is_space(t[6])) {
            t += 7;
            t += std::strspn(t, " \t");
            std::string name(t, std::strcspn(t, " \t\r"));
            if (!load_mtl(matpath + name, &mtls, &mtl_index)) return fail_ctx(ctx, PT_EIO, "cannot open MTL file: " + matpath + name);
            continue;
        }
        if ((t[0] == 'g' || t[0] == 'o') && is_space(t[1])) {
            bool ret = flush_group();
            if (ret) shapes.push_back(shape);
            shape = Shape();
            group.clear();
            continue;
        }
        // vn, vt, s, t ...: nothing add_Obj reads
    }
    {   // end of file: tiny_obj_loader.h flushes the pending group
        bool ret = flush_group();
        if (ret || !shape.faces.empty()) shapes.push_back(shape);
        group.clear();
    }

    // ---- materials, main.cpp:562-581
    int base = -1;                                   // mat_offset, main.cpp:562
    for (size_t i = 0; i < mtls.size(); ++i) {
        const MtlRec& m = mtls[i];
        auto kn = m.unknown.find("Kn"), kk = m.unknown.find("Kk"), tp = m.unknown.find("Tp");
        if (kn == m.unknown.end() || kk == m.unknown.end() || tp == m.unknown.end())
            return fail_ctx(ctx, PT_EIO, "material '" + m.name + "' lacks Kn/Kk/Tp (the reference's .at() would throw, main.cpp:568-571)");
        float N[3], K[3];
        if (!str_to_float3(kn->second, N) || !str_to_float3(kk->second, K))
            return fail_ctx(ctx, PT_EIO, "material '" + m.name + "': Kn/Kk need three blank-separated numbers");
        pt_material pm;
        pt_material_init(&pm, m.diffuse, m.specular, m.emission, N, K, m.shininess, (int32_t)std::atoi(tp->second.c_str()));
        int idx = pt_add_material(ctx, &pm);
        if (idx < 0) return idx;
        if (base < 0) base = idx;
    }
    if (base < 0) base = 0;

    // ---- shapes, main.cpp:587-616
    for (const Shape& sh : shapes) {
        if (sh.faces.empty()) return fail_ctx(ctx, PT_EIO, "OBJ shape without faces (the reference would call end_Obj on an empty object)");
        for (const Face3& f : sh.faces) {
            if (f.mat < 0) return fail_ctx(ctx, PT_EIO, "OBJ face without a known usemtl material (the reference would index materials[-1])");
            float vert[3][3];
            for (int k = 0; k < 3; ++k) {
                const int vi = f.v[k];
                if (vi < 0 || (size_t)vi * 3 + 2 >= v.size()) return fail_ctx(ctx, PT_EIO, "OBJ face references a vertex that does not exist");
                vert[k][0] = -v[3 * vi + 0];                 // main.cpp:598
                vert[k][1] = v[3 * vi + 1];
                vert[k][2] = v[3 * vi + 2];
                rot_x(vert[k], pitch);                       // main.cpp:602-603
                rot_y(vert[k], yaw);
                for (int i = 0; i < 3; ++i) vert[k][i] = vert[k][i] * scale[i] + pos[i];   // main.cpp:604-606
            }
            pt_triangle tri;
            pt_triangle_init(&tri, vert[0], vert[1], vert[2], (uint16_t)(base + f.mat));
            int rc = pt_add_triangle(ctx, &tri);
            if (rc != PT_OK) return rc;
        }
        int rc = pt_end_obj(ctx);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}
