// host-only stress of the threaded scene path under ThreadSanitizer: pt_add_obj (parallel parse, threaded encounter ranks),
// pt_upload_triangles (thread pool, parallel SAH top, splice, 4-wide collapse) on a generated OBJ
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include "pt_api.h"
int main(int argc, char** argv) {
    const char* obj = argv[1];
    auto run = [&](int id) {
        pt_context* ctx = nullptr;
        if (pt_create(-1, 64, 64, &ctx) != PT_OK) { std::printf("create failed\n"); std::exit(1); }
        float kd[3] = {.3f, .3f, .3f}, z[3] = {0, 0, 0};
        pt_material m;
        pt_material_init(&m, kd, z, z, z, z, 50.f, 0);
        for (int k = 0; k < 10; ++k) pt_add_material(ctx, &m);
        float pos[3] = {40, -15, 25}, sc[3] = {2, 2, 2};
        int rc = pt_add_obj(ctx, obj, pos, sc, 10.f, 30.f);
        if (rc != PT_OK) { std::printf("[%d] add_obj rc %d %s\n", id, rc, pt_last_error(ctx)); std::exit(1); }
        rc = pt_upload_triangles(ctx);
        std::printf("[%d] upload rc %d (%s)\n", id, rc, rc ? pt_last_error(ctx) : "ok");
        pt_destroy(ctx);
    };
    run(0);
    std::thread a(run, 1), b(run, 2);      // two contexts at once: the pool is busy for one of them
    a.join();
    b.join();
    return 0;
}
