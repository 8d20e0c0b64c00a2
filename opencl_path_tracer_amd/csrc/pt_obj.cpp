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
#include "pt_internal.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace ptamd {
int fail_ctx(pt_context* ctx, int code, const std::string& msg);   // pt_host.cpp
int append_triangles(pt_context* ctx, int64_t n, pt_triangle** tail);
}

namespace {

struct MtlRec {
    std::string name;
    float diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0}, emission[3] = {0, 0, 0};
    float shininess = 1.0f;                         // tiny_obj_loader.h:858
    std::map<std::string, std::string> unknown;
};

inline bool is_space(char c) { return c == ' ' || c == '\t'; }

// The whole file in one buffer, NUL-terminated; lines are visited in place (a 1M-triangle OBJ is 50 MB and 1.5 M lines:
// one std::string per line and per number was most of pt_add_obj's 1.3 s, profiles/r03/e_*).
bool read_file(const std::string& path, std::vector<char>* buf) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); return false; }
    buf->resize((size_t)n + 1);
    const size_t got = n ? std::fread(buf->data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    if (got != (size_t)n) return false;
    (*buf)[(size_t)n] = '\0';
    return true;
}

// Next line of the buffer: [*cur, end) with the terminator (\n, \r\n or a lone \r) replaced by NUL; false at the end.
// (An empty last line after the final terminator is not a line, as before.)
bool next_line(char** cur, char* end, char** line) {
    char* p = *cur;
    if (p >= end) return false;
    *line = p;
    while (p < end && *p != '\n' && *p != '\r') ++p;
    if (p < end) {
        const char c = *p;
        *p = '\0';
        ++p;
        if (c == '\r' && p < end && *p == '\n') ++p;
    }
    *cur = p;
    return true;
}

// One number: the token runs to the next blank / tab (lines are NUL-terminated in the buffer); its value is what strtod
// makes of the token's prefix, 0 if nothing converts.  strtod runs on the buffer itself -- it stops at the blank or NUL that
// ends the token, so the copy into a std::string that used to precede it bought nothing.  (std::from_chars is no
// alternative with this toolchain: libstdc++ 11 implements it as newlocale + strtod per call, slower and serialised on
// the locale lock -- the parallel parse did not scale at all with it, profiles/r03/e_*.)
float parse_float(const char** tok) {
    const char* t = *tok;
    t += std::strspn(t, " \t");
    const char* end = t + std::strcspn(t, " \t");
    *tok = end;
    if (t == end) return 0.0f;
    char* e = nullptr;
    const double d = std::strtod(t, &e);
    if (e == t) return 0.0f;
    return (float)d;
}

void parse_float3(float out[3], const char** tok) {
    out[0] = parse_float(tok);
    out[1] = parse_float(tok);
    out[2] = parse_float(tok);
}

// atoi: optional blanks, optional sign, digits; stops at the first other character
inline int parse_int(const char* t) {
    while (*t == ' ' || *t == '\t' || *t == '\n' || *t == '\v' || *t == '\f' || *t == '\r') ++t;
    bool neg = false;
    if (*t == '-' || *t == '+') { neg = *t == '-'; ++t; }
    unsigned v = 0;
    while (*t >= '0' && *t <= '9') { v = v * 10u + (unsigned)(*t - '0'); ++t; }
    return neg ? (int)(0u - v) : (int)v;
}

bool load_mtl(const std::string& path, std::vector<MtlRec>* mats, std::map<std::string, int>* index) {
    std::vector<char> buf;
    if (!read_file(path, &buf)) return false;
    char* cur_p = buf.data();
    char* const end_p = buf.data() + buf.size() - 1;
    char* raw = nullptr;
    MtlRec cur;
    bool have = false;
    while (next_line(&cur_p, end_p, &raw)) {
        const char* t = raw;
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
            cur.name = std::string(t, std::strcspn(t, " \t"));
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

// run fn(begin, end) over [0, n) on up to 16 threads (element-wise work: any split gives the same result)
template <class F>
void parallel_ranges(size_t n, size_t grain, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), (n + grain - 1) / std::max<size_t>(grain, 1));
    if (nt <= 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (size_t k = 0; k < nt; ++k) {
        const size_t b = k * per, e = std::min(n, b + per);
        if (b < e) th.emplace_back([=]() { fn(b, e); });
    }
    for (std::thread& t : th) t.join();
}

}  // namespace

extern "C" int pt_add_obj(pt_context* ctx, const char* file, const float pos[3], const float scale[3], float pitch, float yaw) {
    using ptamd::fail_ctx;
    if (!ctx || !file || !pos || !scale) return PT_EINVAL;
    const std::string path(file);
    const std::string matpath = path.substr(0, path.find_last_of('/') + 1);   // main.cpp:553
    std::vector<char> buf;
    ptamd::PhaseClock clk("pt_add_obj");
    if (!read_file(path, &buf)) return fail_ctx(ctx, PT_EIO, "cannot open OBJ file: " + path);
    clk.lap("read file");

    // ---- parse.  Three steps, so that a 50-MB file is read by every core and still means what tinyobj's single pass says:
    //  A (parallel)  the buffer is cut at line ends into pieces; each piece turns its `v` lines into floats and its `f`
    //                lines into fan triangles of RAW indices (tiny_obj_loader.h:893-916) and notes, in order, the lines
    //                that change state: usemtl, mtllib, g / o, and faces of fewer than three vertices;
    //  B (serial)    the state machine over those notes: current material, whether the face group is open, which
    //                shape a run of triangles belongs to and whether that shape survives (below);
    //  C (parallel)  relative indices are resolved against the number of vertices read in front of each face
    //                (tiny_obj_loader.h:410-414) and the runs are copied to their shapes.
    // tinyobj collects the faces of the current material in a group and exports it (exportFaceGroupToShape with
    // triangulate = true) when the material changes, at g / o and at the end of the file; g / o keeps the shape only if
    // the group it closes is not empty (tiny_obj_loader.h:1509-1567) -- faces exported by an earlier usemtl are lost
    // with it, which B reproduces.  A face of fewer than three vertices opens the group and exports nothing.
    struct RawTri { int idx[3]; int nv_local; };
    enum { kEvUsemtl = 0, kEvMtllib = 1, kEvShape = 2, kEvShortFace = 3 };
    struct Event { int kind; size_t tri_pos; std::string name; };
    struct Piece {
        char* begin = nullptr;
        char* end = nullptr;
        std::vector<float> v;
        std::vector<RawTri> tris;
        std::vector<Event> ev;
    };
    char* const file_begin = buf.data();
    char* const file_end = buf.data() + buf.size() - 1;
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_pieces = std::max<size_t>(1, std::min<size_t>((size_t)std::min<unsigned>(hw ? hw : 1u, 16u) * 4, (size_t)(file_end - file_begin) >> 16));
    std::vector<Piece> pieces(n_pieces);
    {
        char* start = file_begin;
        for (size_t k = 0; k < n_pieces; ++k) {
            pieces[k].begin = start;
            char* stop = file_end;
            if (k + 1 < n_pieces) {
                stop = std::max(start, file_begin + (size_t)(file_end - file_begin) * (k + 1) / n_pieces);
                while (stop < file_end && *stop != '\n' && *stop != '\r') ++stop;
                if (stop < file_end) {
                    const char c = *stop++;
                    if (c == '\r' && stop < file_end && *stop == '\n') ++stop;
                }
            }
            pieces[k].end = stop;
            start = stop;
        }
    }
    auto parse_piece = [](Piece& pc) {
        pc.v.reserve((size_t)(pc.end - pc.begin) / 24);
        pc.tris.reserve((size_t)(pc.end - pc.begin) / 20);
        char* cur_p = pc.begin;
        char* raw = nullptr;
        while (next_line(&cur_p, pc.end, &raw)) {
            const char* t = raw;
            t += std::strspn(t, " \t");
            if (*t == '\0' || *t == '#') continue;
            if (t[0] == 'v' && is_space(t[1])) {
                t += 2;
                float p[3];
                parse_float3(p, &t);
                pc.v.push_back(p[0]); pc.v.push_back(p[1]); pc.v.push_back(p[2]);
                continue;
            }
            if (t[0] == 'f' && is_space(t[1])) {
                t += 2;
                t += std::strspn(t, " \t");
                const int nv_local = (int)(pc.v.size() / 3);
                int first = 0, prev = 0, count = 0;
                while (*t != '\0') {
                    const int idx = parse_int(t);                         // the v of v, v/vt, v//vn, v/vt/vn
                    if (count == 0) first = idx;
                    else if (count >= 2) pc.tris.push_back(RawTri{{first, prev, idx}, nv_local});   // fan: (first, previous, this)
                    prev = idx;
                    ++count;
                    t += std::strcspn(t, " \t");
                    t += std::strspn(t, " \t");
                }
                if (count < 3) pc.ev.push_back(Event{kEvShortFace, pc.tris.size(), std::string()});
                continue;
            }
            const bool usemtl = 0 == std::strncmp(t, "usemtl", 6) && is_space(t[6]);
            if (usemtl || (0 == std::strncmp(t, "mtllib", 6) && is_space(t[6]))) {
                t += 7;
                t += std::strspn(t, " \t");
                pc.ev.push_back(Event{usemtl ? kEvUsemtl : kEvMtllib, pc.tris.size(), std::string(t, std::strcspn(t, " \t"))});
                continue;
            }
            if ((t[0] == 'g' || t[0] == 'o') && is_space(t[1])) pc.ev.push_back(Event{kEvShape, pc.tris.size(), std::string()});
            // vn, vt, s, t ...: nothing add_Obj reads
        }
    };
    {
        std::atomic<size_t> next(0);
        std::vector<std::thread> th;
        const size_t nt = std::min<size_t>(n_pieces, std::min<unsigned>(hw ? hw : 1u, 16u));
        auto worker = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= n_pieces) return;
                parse_piece(pieces[k]);
            }
        };
        for (size_t k = 1; k < nt; ++k) th.emplace_back(worker);
        worker();
        for (std::thread& t : th) t.join();
    }
    clk.lap("parse: pieces");

    // B: the state machine
    std::vector<MtlRec> mtls;
    std::map<std::string, int> mtl_index;
    struct Run { size_t piece, tri_begin, tri_end; int material; size_t shape; size_t dst; };
    std::vector<Run> runs;
    std::vector<size_t> shape_tris;      // triangles per provisional shape
    std::vector<char> shape_kept;
    shape_tris.push_back(0);
    shape_kept.push_back(0);
    bool group_open = false;
    int material = -1;
    std::vector<size_t> vbase(n_pieces + 1, 0);
    for (size_t k = 0; k < n_pieces; ++k) {
        const Piece& pc = pieces[k];
        vbase[k + 1] = vbase[k] + pc.v.size() / 3;
        size_t at = 0;
        auto close_run = [&](size_t upto) {
            if (upto > at) {
                runs.push_back(Run{k, at, upto, material, shape_tris.size() - 1, shape_tris.back()});
                shape_tris.back() += upto - at;
                group_open = true;
                at = upto;
            }
        };
        for (const Event& e : pc.ev) {
            close_run(e.tri_pos);
            if (e.kind == kEvShortFace) {
                group_open = true;
            } else if (e.kind == kEvUsemtl) {
                int id = -1;
                auto it = mtl_index.find(e.name);
                if (it != mtl_index.end()) id = it->second;
                if (id != material) { group_open = false; material = id; }
            } else if (e.kind == kEvMtllib) {
                if (!load_mtl(matpath + e.name, &mtls, &mtl_index)) return fail_ctx(ctx, PT_EIO, "cannot open MTL file: " + matpath + e.name);
            } else {      // g / o
                shape_kept.back() = group_open ? 1 : 0;
                shape_tris.push_back(0);
                shape_kept.push_back(0);
                group_open = false;
            }
        }
        close_run(pc.tris.size());
    }
    shape_kept.back() = (group_open || shape_tris.back() != 0) ? 1 : 0;     // end of file: tiny_obj_loader.h flushes the pending group

    // C: vertices in file order, triangles into their shapes
    std::vector<float> v(vbase[n_pieces] * 3);
    std::vector<Shape> shapes;
    std::vector<size_t> shape_slot(shape_tris.size(), (size_t)-1);
    for (size_t sidx = 0; sidx < shape_tris.size(); ++sidx)
        if (shape_kept[sidx]) {
            shape_slot[sidx] = shapes.size();
            shapes.emplace_back();
            shapes.back().faces.resize(shape_tris[sidx]);
        }
    parallel_ranges(n_pieces, 1, [&](size_t b, size_t e) {
        for (size_t k = b; k < e; ++k)
            if (!pieces[k].v.empty()) std::memcpy(&v[vbase[k] * 3], pieces[k].v.data(), sizeof(float) * pieces[k].v.size());
    });
    parallel_ranges(runs.size(), 1, [&](size_t b, size_t e) {
        for (size_t r = b; r < e; ++r) {
            const Run& run = runs[r];
            if (shape_slot[run.shape] == (size_t)-1) continue;
            Face3* out = shapes[shape_slot[run.shape]].faces.data() + run.dst;
            const Piece& pc = pieces[run.piece];
            const int vb = (int)vbase[run.piece];
            for (size_t i = run.tri_begin; i < run.tri_end; ++i, ++out) {
                const RawTri& rt = pc.tris[i];
                for (int c = 0; c < 3; ++c) out->v[c] = fix_index(rt.idx[c], vb + rt.nv_local);
                out->mat = run.material;
            }
        }
    });
    pieces.clear();
    clk.lap("parse");

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

    // ---- vertices, main.cpp:598-606: negate x, rotate_x(pitch), rotate_y(yaw), scale, translate -- once per VERTEX of
    // the file (the reference redoes it for every corner of every face, with the same arithmetic and so the same bits)
    const size_t nvert = v.size() / 3;
    std::vector<float> w(v.size());
    parallel_ranges(nvert, 1 << 14, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            float q[3] = {-v[3 * i + 0], v[3 * i + 1], v[3 * i + 2]};       // main.cpp:598
            rot_x(q, pitch);                                                // main.cpp:602-603
            rot_y(q, yaw);
            for (int a = 0; a < 3; ++a) w[3 * i + a] = q[a] * scale[a] + pos[a];   // main.cpp:604-606
        }
    });

    clk.lap("transform vertices");
    // ---- shapes, main.cpp:587-616: one add_Triangle per face, one end_Obj per shape
    for (const Shape& sh : shapes) {
        if (sh.faces.empty()) return fail_ctx(ctx, PT_EIO, "OBJ shape without faces (the reference would call end_Obj on an empty object)");
        const size_t nf = sh.faces.size();
        int bad = 0;       // 1: face without material, 2: vertex out of range
        for (size_t i = 0; i < nf && !bad; ++i) {
            const Face3& f = sh.faces[i];
            if (f.mat < 0) bad = 1;
            for (int k = 0; k < 3 && !bad; ++k)
                if (f.v[k] < 0 || (size_t)f.v[k] >= nvert) bad = 2;
        }
        if (bad == 1) return fail_ctx(ctx, PT_EIO, "OBJ face without a known usemtl material (the reference would index materials[-1])");
        if (bad == 2) return fail_ctx(ctx, PT_EIO, "OBJ face references a vertex that does not exist");
        pt_triangle* tris = nullptr;                 // the records are built in place at the end of the scene's triangle list
        int rc = ptamd::append_triangles(ctx, (int64_t)nf, &tris);
        if (rc != PT_OK) return rc;
        parallel_ranges(nf, 1 << 14, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const Face3& f = sh.faces[i];
                pt_triangle_init(&tris[i], &w[3 * (size_t)f.v[0]], &w[3 * (size_t)f.v[1]], &w[3 * (size_t)f.v[2]], (uint16_t)(base + f.mat));
            }
        });
        clk.lap("triangle records (add_Triangle)");
        rc = pt_end_obj(ctx);
        if (rc != PT_OK) return rc;
        clk.lap("pt_end_obj");
    }
    return PT_OK;
}
