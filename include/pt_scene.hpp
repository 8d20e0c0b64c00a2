// pt_scene.hpp -- header-only C++ mirror of the reference's host interface over the C ABI
// (pt_api.h).  Same class names, method names, argument meaning and call order as
// /root/reference/main.cpp:92-182 (Material, Triangle), 306-348 (Camera) and 363-742 (Scene), so
// that the reference's onInitialization()/onIdle() bodies (main.cpp:749-1016, 1226) compile
// against it with `using namespace ptamd_dropin;` once cl_float3 is spelled pt_float3.
//
// Differences forced by leaving OpenCL/GLUT behind:
//   * the globals the reference's Camera() and trace_rays() read (screen_width/height,
//     iterations, current_sample, global_fov/yaw/pitch/shift and the per-frame movement
//     global_forward/rightward/upward: main.cpp:20-39) are members of Scene (`globals`);
//     Camera(Globals&) has the reference constructor's side effect -- it adds the movement into
//     global_shift along the rotated axes (main.cpp:334-336) -- so the key handlers of
//     main.cpp:1189-1224 move the camera exactly as they do there; init_Scene takes the frame size;
//   * failures throw std::runtime_error instead of exit(1) (main.cpp:502, 560);
//   * the radiance is read back with download_colors() instead of being blitted from a GL
//     texture (main.cpp:519, 1019-1039).
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "pt_api.h"

namespace ptamd_dropin {

typedef pt_float3 cl_float3;   // 16 bytes, like CL/cl_platform.h

struct Material : pt_material {                        // main.cpp:92-112
    Material() { type = -1; }
    Material(cl_float3 kd, cl_float3 ks, cl_float3 emission, cl_float3 N, cl_float3 K, float shininess, int type_) {
        pt_material_init(this, kd.s, ks.s, emission.s, N.s, K.s, shininess, type_);
    }
};

struct Triangle : pt_triangle {                        // main.cpp:139-182
    Triangle(cl_float3 r1_, cl_float3 r2_, cl_float3 r3_, unsigned short mati_) { pt_triangle_init(this, r1_.s, r2_.s, r3_.s, mati_); }
};

struct Globals {                                       // the shipped values of main.cpp:20-39
    int screen_width = 192 * 8, screen_height = 108 * 8;                                 // main.cpp:20-21
    int iterations = 1;                                                                  // main.cpp:27
    float global_fov = 75.0f;                                                            // main.cpp:30
    float global_yaw = (float)(-13.800002 - 50), global_pitch = (float)(5.599997 + 10);   // main.cpp:31-32 (double arithmetic, narrowed)
    float global_forward = 0, global_rightward = 0, global_upward = 0;                   // main.cpp:36-38: this frame's movement (speed * dt or 0, main.cpp:1189-1209)
    cl_float3 global_shift = {{265.055481f, 162.305969f, 360.414001f, 0.0f}};            // main.cpp:39
    // (the "canonical" view the reference keeps in comments, main.cpp:33-35,40: fov 60, yaw 0, pitch 0, shift 0)
};

struct Camera : pt_camera {                            // main.cpp:306-348
    Camera() { XM = YM = 0; }
    // the reference's Camera(): FIRST the movement accumulates into global_shift (main.cpp:334-336), then the eye is placed
    explicit Camera(Globals& g) {
        pt_camera_move(g.global_shift.s, g.global_yaw, g.global_pitch, g.global_forward, g.global_rightward, g.global_upward);
        pt_camera_init(this, g.global_fov, g.global_yaw, g.global_pitch, g.global_shift.s, g.screen_width, g.screen_height);
    }
    // a view of the globals as they stand (no movement applied)
    explicit Camera(const Globals& g) { pt_camera_init(this, g.global_fov, g.global_yaw, g.global_pitch, g.global_shift.s, g.screen_width, g.screen_height); }
};

class Scene {                                          // main.cpp:363-742
public:
    Globals globals;

    Scene() {}
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene() { if (ctx) pt_destroy(ctx); }

    std::string list_info() {                          // main.cpp:389-455
        char buf[256];
        ck(pt_device_info(ctx, buf, sizeof buf));
        return buf;
    }
    void init_Scene(int device = 0) {                  // main.cpp:456-528
        if (pt_create(device, globals.screen_width, globals.screen_height, &ctx) != PT_OK)
            throw std::runtime_error(std::string("init_Scene: ") + pt_last_error(nullptr));
    }
    void add_Triangle(const Triangle& tri) { ck(pt_add_triangle(ctx, &tri)); }                 // main.cpp:529
    int add_Material(const Material& mat) { return ck(pt_add_material(ctx, &mat)); }           // main.cpp:532
    void end_Obj() { ck(pt_end_obj(ctx)); }                                                    // main.cpp:536
    void add_Obj(const std::string& file, cl_float3 pos, cl_float3 scale, float pitch, float yaw) {   // main.cpp:552
        ck(pt_add_obj(ctx, file.c_str(), pos.s, scale.s, pitch, yaw));
    }
    void upload_Triangles() { ck(pt_upload_triangles(ctx)); }                                  // main.cpp:618
    void upload_Materials() { ck(pt_upload_materials(ctx)); }                                  // main.cpp:631
    void generate_rays() {                                                                     // main.cpp:635
        camera = Camera(globals);
        ck(pt_generate_rays(ctx, &camera));
    }
    void trace_rays() {                                                                        // main.cpp:645
        int32_t s = 0;
        ck(pt_get_current_sample(ctx, &s));
        ck(pt_trace_rays(ctx, &camera, globals.iterations, s));
    }
    void render() {                                                                            // main.cpp:683
        generate_rays();
        trace_rays();
        ck(pt_set_current_sample(ctx, current_sample() + 1));
    }
    // nsamples x render() as one persistent launch (same result, bit for bit, for a camera at rest; a moving camera -- a
    // non-zero global_forward / rightward / upward -- moves ONCE per call here, where n x render() would move it n times)
    void render(int nsamples) {
        camera = Camera(globals);
        ck(pt_render(ctx, &camera, globals.iterations, nsamples));
    }
    int current_sample() { int32_t s = 0; ck(pt_get_current_sample(ctx, &s)); return s; }
    void reset_samples() { ck(pt_set_current_sample(ctx, 0)); }                                // main.cpp:1046
    void finish() { ck(pt_sync(ctx)); }                                                        // queue.finish(), main.cpp:675
    std::vector<cl_float3> download_colors() {                                                 // (commented out in the reference: main.cpp:727)
        int64_t n = 0;
        ck(pt_local_pixel_count(ctx, &n));
        std::vector<cl_float3> out((size_t)n);
        ck(pt_read_colors(ctx, &out[0].s[0], n));
        return out;
    }
    // what the reference shows through its GL blit (main.cpp:1019-1039), as files
    void write_ppm(const std::string& path, int which = 0) { ck(pt_write_ppm(ctx, path.c_str(), which)); }
    void write_pfm(const std::string& path) { ck(pt_write_pfm(ctx, path.c_str())); }
    // multi-GPU hosts: init_Scene_tiled instead of init_Scene, then comm_init with the
    // 128-byte id rank 0 obtained from Scene::comm_unique_id(), render as usual, gather_frame() + download_frame()
    void init_Scene_tiled(int device, int rank, int world, int rows_per_block = 8) {
        if (pt_create_tiled(device, globals.screen_width, globals.screen_height, rank, world, rows_per_block, &ctx) != PT_OK)
            throw std::runtime_error(std::string("init_Scene_tiled: ") + pt_last_error(nullptr));
    }
    static std::vector<unsigned char> comm_unique_id() {
        std::vector<unsigned char> id(PT_COMM_ID_BYTES);
        if (pt_comm_unique_id(id.data()) != PT_OK) throw std::runtime_error(std::string("comm_unique_id: ") + pt_last_error(nullptr));
        return id;
    }
    void comm_init(const std::vector<unsigned char>& id) { ck(pt_comm_init(ctx, id.data())); }
    void gather_frame() { ck(pt_gather_frame(ctx)); }
    std::vector<cl_float3> download_frame() {
        int64_t n = 0;
        ck(pt_frame_size(ctx, nullptr, nullptr, &n));
        std::vector<cl_float3> out((size_t)n);
        ck(pt_read_frame(ctx, &out[0].s[0], n));
        return out;
    }
    pt_context* handle() { return ctx; }

private:
    int ck(int rc) {
        if (rc < 0) throw std::runtime_error(std::string("libptamd: ") + pt_last_error(ctx));
        return rc;
    }
    pt_context* ctx = nullptr;
    Camera camera;
};

}  // namespace ptamd_dropin
