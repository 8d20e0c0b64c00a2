// ref_tinyobj_dump.cpp -- harness around the REFERENCE's own vendored parser
// (/root/reference/tiny_obj_loader.h, compiled where it lies; nothing is copied).  It calls
// tinyobj::LoadObj exactly like Scene::add_Obj does (main.cpp:553-558) and dumps what add_Obj
// reads from the result (main.cpp:564-613) as JSON.  Used only by tests/golden/make_obj_golden.py
// in the build container to produce fixtures for the product's own OBJ reader.
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"

#include <cstdio>
#include <string>

static void jstr(const std::string& s) {
    std::putchar('"');
    for (char c : s) {
        if (c == '"' || c == '\\') { std::putchar('\\'); std::putchar(c); }
        else if (c == '\r') std::printf("\\r");
        else if (c == '\n') std::printf("\\n");
        else if (c == '\t') std::printf("\\t");
        else std::putchar(c);
    }
    std::putchar('"');
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::string file = argv[1];
    std::string matpath = file.substr(0, file.find_last_of("/") + 1);        // main.cpp:553
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    bool ret = tinyobj::LoadObj(&attrib, &shapes, &materials, &err, file.c_str(), matpath.c_str());   // main.cpp:558
    std::printf("{\"ret\": %s, \"err\": ", ret ? "true" : "false");
    jstr(err);
    std::printf(",\n \"vertices\": [");
    for (size_t i = 0; i < attrib.vertices.size(); ++i) std::printf("%s%.9g", i ? ", " : "", attrib.vertices[i]);
    std::printf("],\n \"materials\": [");
    for (size_t i = 0; i < materials.size(); ++i) {
        const tinyobj::material_t& m = materials[i];
        std::printf("%s\n  {\"name\": ", i ? "," : "");
        jstr(m.name);
        std::printf(", \"diffuse\": [%.9g, %.9g, %.9g], \"specular\": [%.9g, %.9g, %.9g], \"emission\": [%.9g, %.9g, %.9g], \"shininess\": %.9g, \"unknown\": {",
                    m.diffuse[0], m.diffuse[1], m.diffuse[2], m.specular[0], m.specular[1], m.specular[2],
                    m.emission[0], m.emission[1], m.emission[2], m.shininess);
        bool first = true;
        for (auto& kv : m.unknown_parameter) {
            std::printf("%s", first ? "" : ", ");
            jstr(kv.first);
            std::printf(": ");
            jstr(kv.second);
            first = false;
        }
        std::printf("}}");
    }
    std::printf("],\n \"shapes\": [");
    for (size_t s = 0; s < shapes.size(); ++s) {
        std::printf("%s\n  {\"name\": ", s ? "," : "");
        jstr(shapes[s].name);
        std::printf(", \"num_face_vertices\": [");
        for (size_t f = 0; f < shapes[s].mesh.num_face_vertices.size(); ++f) std::printf("%s%d", f ? ", " : "", (int)shapes[s].mesh.num_face_vertices[f]);
        std::printf("], \"material_ids\": [");
        for (size_t f = 0; f < shapes[s].mesh.material_ids.size(); ++f) std::printf("%s%d", f ? ", " : "", shapes[s].mesh.material_ids[f]);
        std::printf("], \"vertex_index\": [");
        for (size_t k = 0; k < shapes[s].mesh.indices.size(); ++k) std::printf("%s%d", k ? ", " : "", shapes[s].mesh.indices[k].vertex_index);
        std::printf("]}");
    }
    std::printf("]}\n");
    return 0;
}
