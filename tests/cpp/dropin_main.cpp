// A host program written the way the reference's onInitialization()/onIdle() are
// (main.cpp:749-816, 1015-1016, 1226), against include/pt_scene.hpp.  Prints a checksum of
// colors and the LCG state; tests/test_gpu_parity.py compares them with the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pt_scene.hpp"

using namespace ptamd_dropin;

int main(int argc, char** argv) {
    const int W = argc > 1 ? std::atoi(argv[1]) : 64, H = argc > 2 ? std::atoi(argv[2]) : 64;
    const int samples = argc > 3 ? std::atoi(argv[3]) : 3;
    try {
        Scene scene;
        scene.globals.screen_width = W;
        scene.globals.screen_height = H;
        scene.globals.iterations = 4;
        scene.init_Scene();
        std::printf("# %s\n", scene.list_info().c_str());
        unsigned short LAMP, WHITE_DIFFUSE;
        LAMP = 0;          scene.add_Material(Material((cl_float3){0.0f, 0.0f, 0.0f}, (cl_float3){0.0f, 0.0f, 0.0f}, (cl_float3){60.0f * 2, 50.0f * 2, 40.0f * 2}, (cl_float3){0.00f, 0.00f, 0.00f}, (cl_float3){0.0f, 0.0f, 0.0f}, 0, 3));
        WHITE_DIFFUSE = 1; scene.add_Material(Material((cl_float3){0.3f, 0.3f, 0.3f}, (cl_float3){0.0f, 0.0f, 0.0f}, (cl_float3){0.0f, 0.0f, 0.0f}, (cl_float3){0.00f, 0.00f, 0.00f}, (cl_float3){0.0f, 0.0f, 0.0f}, 50, 0));
        // lamp + floor, coordinates as in main.cpp:765-766, 814-815
        scene.add_Triangle(Triangle((cl_float3){300.0f, 999.9f, 700.0f}, (cl_float3){300.0f, 999.9f, 300.0f}, (cl_float3){700.0f, 999.9f, 700.0f}, LAMP));
        scene.add_Triangle(Triangle((cl_float3){700.0f, 999.9f, 700.0f}, (cl_float3){300.0f, 999.9f, 300.0f}, (cl_float3){700.0f, 999.9f, 300.0f}, LAMP));
        scene.end_Obj();
        scene.add_Triangle(Triangle((cl_float3){-10000.0f, 0.0f, -10000.0f}, (cl_float3){-10000.0f, 0.0f, 10000.0f}, (cl_float3){10000.0f, 0.0f, 10000.0f}, WHITE_DIFFUSE));
        scene.add_Triangle(Triangle((cl_float3){10000.0f, 0.0f, 10000.0f}, (cl_float3){10000.0f, 0.0f, -10000.0f}, (cl_float3){-10000.0f, 0.0f, -10000.0f}, WHITE_DIFFUSE));
        scene.end_Obj();
        scene.upload_Triangles();
        scene.upload_Materials();
        for (int i = 0; i < samples; ++i) scene.render();          // onIdle, main.cpp:1226
        scene.finish();
        std::vector<cl_float3> colors = scene.download_colors();
        if (argc > 4) scene.write_ppm(argv[4]);                      // what onDisplay() would have shown (main.cpp:1019-1039)
        unsigned long long h = 1469598103934665603ull;
        for (const cl_float3& c : colors)
            for (int k = 0; k < 3; ++k) {
                unsigned u;
                std::memcpy(&u, &c.s[k], 4);
                h = (h ^ u) * 1099511628211ull;
            }
        std::printf("samples %d colors_fnv %016llx\n", scene.current_sample(), h);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    return 0;
}
