// pt_host.cpp -- host side of libptamd.so: the C ABI of include/pt_api.h (context, authoring, read-back, options, statistics).
//
// What lives here (all of it host work the reference also does on the host):
//   * value-type constructors            main.cpp:101-111, 144-166, 311-347
//   * context life cycle, Scene authoring (add_Material / add_Triangle), seeds, materials
//   * read-back, frame assembly entry points, options, statistics, debug getters
// Scene -> tree is pt_builder.cpp (pt_end_obj, pt_upload_triangles), tree -> launches pt_launch.cpp (pt_render ...); what the
// three share is pt_context.hpp.  There is no CPU render path in this library: every pt_render / pt_trace_rays /
// pt_generate_rays call launches HIP kernels or fails.
//
// Compiled with -ffp-contract=off: the reference's host arithmetic is plain x86-64 g++
// (no fused multiply-add), and the results of these constructors feed bit-exact parity tests.
#include "pt_context.hpp"

namespace {
thread_local std::string g_create_error;
}  // namespace

namespace ptamd {

int fail(pt_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

// threads of the host-side scene path (option build_threads; 0: the machine's, at most 16)
int host_threads(const pt_context* ctx) {
    const unsigned hw = std::thread::hardware_concurrency();
    return ctx->build_threads > 0 ? ctx->build_threads : (int)std::min(16u, std::max(1u, hw));
}

int fail_ctx(pt_context* ctx, int code, const std::string& msg) { return fail(ctx, code, msg); }   // for pt_obj.cpp, pt_image.cpp
// pt_add_triangles without the copy, for pt_obj.cpp: n records are appended and returned for the caller to fill in place
// (every one of them, before anything else touches the context)
int append_triangles(pt_context* ctx, int64_t n, pt_triangle** tail) {
    if (!ctx || n < 0 || !tail) return PT_EINVAL;
    if ((int64_t)ctx->tris.size() + n > kMaxTriangles) return fail(ctx, PT_EINVAL, "more than 2^26 triangles");
    const size_t old = ctx->tris.size();
    ctx->tris.resize(old + (size_t)n);
    ctx->tris_uploaded = false;
    *tail = ctx->tris.data() + old;
    return PT_OK;
}
// pt_comm.hip
hipError_t launch_deinterleave(const float4* gathered, float4* frame, int W, int H, int world, int rb, long long slab_stride, hipStream_t stream);
void gather_source_index(int W, int H, int world, int rb, long long slab_stride, int64_t* out);
int comm_available(std::string* err);
int comm_unique_id(void* id128, std::string* err);
int comm_init(const void* id128, int rank, int world, void** comm_out, std::string* err);
void comm_destroy(void* comm);
int comm_all_gather(void* comm, const void* send, void* recv, size_t floats_per_rank, hipStream_t stream, std::string* err);

}  // namespace ptamd

extern "C" {

void pt_material_init(pt_material* m, const float kd[3], const float ks[3], const float emission[3],
                      const float N[3], const float K[3], float shininess, int32_t type) {
    std::memset(m, 0, sizeof *m);
    for (int i = 0; i < 3; ++i) { m->kd.s[i] = kd[i]; m->ks.s[i] = ks[i]; m->emission.s[i] = emission[i]; }
    m->shininess = shininess;
    m->type = type;
    m->n = (N[0] + N[1] + N[2]) / 3.0f;                       // main.cpp:103
    for (int i = 0; i < 3; ++i) {                              // main.cpp:105-109
        float a = (N[i] - 1) * (N[i] - 1);
        float b = (N[i] + 1) * (N[i] + 1);
        m->F0.s[i] = (K[i] * K[i] + a) / (K[i] * K[i] + b);
    }
}

void pt_triangle_init(pt_triangle* t, const float r1[3], const float r2[3], const float r3[3], uint16_t mati) {
    std::memset(t, 0, sizeof *t);
    float v1[3], v2[3], n[3];
    for (int i = 0; i < 3; ++i) {
        t->r1.s[i] = r1[i]; t->r2.s[i] = r2[i]; t->r3.s[i] = r3[i];
        v1[i] = r2[i] - r1[i];
        v2[i] = r3[i] - r1[i];
    }
    t->mati = mati;
    n[0] = v1[1] * v2[2] - v1[2] * v2[1];
    n[1] = v1[2] * v2[0] - v1[0] * v2[2];
    n[2] = v1[0] * v2[1] - v1[1] * v2[0];
    // main.cpp:160: unqualified sqrt on a float -> the double routine, narrowed
    float length = (float)std::sqrt((double)(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]));
    for (int i = 0; i < 3; ++i) t->N.s[i] = n[i] / length;
}

void pt_triangles_init(pt_triangle* out, const float* verts, const uint16_t* mati, int64_t n) {
    for (int64_t i = 0; i < n; ++i) pt_triangle_init(&out[i], verts + 9 * i, verts + 9 * i + 3, verts + 9 * i + 6, mati[i]);
}

static void rotate_x_ref(float v[3], float gamma) {  // main.cpp:63-70 (trig in double)
    gamma = gamma / 180.0f * 3.141593f;
    const double c = std::cos((double)gamma), s = std::sin((double)gamma);
    const float r1 = (float)((double)v[1] * c - (double)v[2] * s);
    const float r2 = (float)((double)v[1] * s + (double)v[2] * c);
    v[1] = r1;
    v[2] = r2;
}
static void rotate_y_ref(float v[3], float beta) {  // main.cpp:55-62
    beta = beta / 180.0f * 3.141593f;
    const double c = std::cos((double)beta), s = std::sin((double)beta);
    const float r0 = (float)((double)v[0] * c + (double)v[2] * s);
    const float r2 = (float)(-(double)v[0] * s + (double)v[2] * c);
    v[0] = r0;
    v[2] = r2;
}

void pt_camera_init(pt_camera* c, float fov, float yaw, float pitch, const float shift[3], int32_t width, int32_t height) {
    std::memset(c, 0, sizeof *c);
    c->XM = (float)width;
    c->YM = (float)height;
    const float up_length = c->YM / 2.0f;
    const float right_length = c->XM / 2.0f;
    const float ahead_length = (float)((double)right_length / std::tan((double)(fov / 2.0f / 180.0f * 3.141593f)));
    float up[3] = {0.0f, 1.0f, 0.0f}, right[3] = {1.0f, 0.0f, 0.0f}, ahead[3] = {0.0f, 0.0f, 1.0f};
    rotate_x_ref(up, pitch); rotate_y_ref(up, yaw);
    rotate_x_ref(right, pitch); rotate_y_ref(right, yaw);
    rotate_x_ref(ahead, pitch); rotate_y_ref(ahead, yaw);
    for (int i = 0; i < 3; ++i) { up[i] *= up_length; right[i] *= right_length; ahead[i] *= ahead_length; }
    c->eye.s[0] = 500.0f + shift[0];
    c->eye.s[1] = 500.0f + shift[1];
    c->eye.s[2] = -1299.037842f + shift[2];
    for (int i = 0; i < 3; ++i) { c->up.s[i] = up[i]; c->right.s[i] = right[i]; c->lookat.s[i] = c->eye.s[i] + ahead[i]; }
}

// The side effect of the reference's Camera(): main.cpp:334-336 adds this frame's movement along the ROTATED unit axes into
// global_shift before the eye is placed (the key handlers of main.cpp:1189-1209 set global_forward / rightward / upward to
// speed * dt or 0).  Same rotations as pt_camera_init, float arithmetic in the reference's order, no fused operations.
void pt_camera_move(float shift[3], float yaw, float pitch, float forward, float rightward, float upward) {
    float up[3] = {0.0f, 1.0f, 0.0f}, right[3] = {1.0f, 0.0f, 0.0f}, ahead[3] = {0.0f, 0.0f, 1.0f};
    rotate_x_ref(up, pitch); rotate_y_ref(up, yaw);
    rotate_x_ref(right, pitch); rotate_y_ref(right, yaw);
    rotate_x_ref(ahead, pitch); rotate_y_ref(ahead, yaw);
    for (int i = 0; i < 3; ++i) {
        const float a = ahead[i] * forward, r = right[i] * rightward, u = up[i] * upward;
        shift[i] = ((shift[i] + a) + r) + u;
    }
}

int pt_create_tiled(int device, int32_t width, int32_t height, int32_t rank, int32_t world, int32_t rows_per_block, pt_context** out) {
    if (!out) return fail(nullptr, PT_EINVAL, "out is NULL");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(nullptr, PT_EINVAL, "bad frame size");
    if (width > 65535 || height > 65535) return fail(nullptr, PT_EINVAL, "bad frame size: at most 65,535 pixels per side (the kernels pack a pixel's coordinates into 2 x 16 bits)");
    if (world < 1 || rank < 0 || rank >= world || rows_per_block < 1) return fail(nullptr, PT_EINVAL, "bad rank/world/rows_per_block");
    pt_context* ctx = new pt_context();
    ctx->W = width;
    ctx->H = height;
    ctx->rank = rank;
    ctx->world = world;
    ctx->rows_per_block = rows_per_block;
    ctx->local_rows = count_local_rows(height, rank, world, rows_per_block);
    ctx->npix = (int64_t)ctx->local_rows * width;
    for (int32_t r = 0; r < world; ++r) ctx->slab_pix = std::max(ctx->slab_pix, (int64_t)count_local_rows(height, r, world, rows_per_block) * width);
    ctx->device = device;
    if (device < 0) {  // host-only context: authoring + BVH build + debug getters, nothing renders
        std::snprintf(ctx->info, sizeof ctx->info, "host-only context (no device)");
        *out = ctx;
        return PT_OK;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0 || device >= count) {
        delete ctx;
        return fail(nullptr, PT_ENODEVICE, std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device index out of range"));
    }
    auto bail = [&](const char* what, hipError_t err) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(err);
        pt_destroy(ctx);
        return fail(nullptr, PT_EHIP, msg);
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    ctx->cu_count = prop.multiProcessorCount;
    std::snprintf(ctx->info, sizeof ctx->info, "%s (%s), %d CUs, %.1f GiB, wave %d", prop.name, prop.gcnArchName,
                  prop.multiProcessorCount, (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), prop.warpSize);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string msg = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only";
        pt_destroy(ctx);
        return fail(nullptr, PT_ENODEVICE, msg);
    }
    ctx->has_device = true;
    const size_t np = (size_t)std::max<int64_t>(ctx->npix, 1);
    const size_t nslab = (size_t)std::max<int64_t>(ctx->slab_pix, 1);      // colors is this rank's slab of the all-gather
    if ((e = hipMalloc((void**)&ctx->d_rays, sizeof(pt_ray) * np)) != hipSuccess) return bail("hipMalloc(rays)", e);      // main.cpp:508
    if ((e = hipMalloc((void**)&ctx->d_rnds, sizeof(int32_t) * np)) != hipSuccess) return bail("hipMalloc(rnds)", e);     // main.cpp:509
    if ((e = hipMalloc((void**)&ctx->d_colors, sizeof(float4) * nslab)) != hipSuccess) return bail("hipMalloc(colors)", e);  // main.cpp:520
    if ((e = hipMalloc((void**)&ctx->d_stats, sizeof(unsigned long long) * kStatCols * kStatRows)) != hipSuccess) return bail("hipMalloc(stats)", e);
    if ((e = hipMalloc((void**)&ctx->d_tile_counter, sizeof(uint32_t) * kTileCounterWords)) != hipSuccess) return bail("hipMalloc(tile counter)", e);
    if ((e = hipMemset(ctx->d_tile_counter, 0, sizeof(uint32_t) * kTileCounterWords)) != hipSuccess) return bail("hipMemset", e);      // zeroed ONCE: the last wave of a launch leaves it zero (k_render)
    if ((e = hipMemset(ctx->d_rays, 0, sizeof(pt_ray) * np)) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipMemset(ctx->d_colors, 0, sizeof(float4) * nslab)) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipMemset(ctx->d_stats, 0, sizeof(unsigned long long) * kStatCols * kStatRows)) != hipSuccess) return bail("hipMemset", e);
    int rc = pt_seed_default(ctx);                                                                                    // main.cpp:522-527
    if (rc != PT_OK) {
        std::string msg = ctx->err;
        pt_destroy(ctx);
        return fail(nullptr, rc, msg);
    }
    *out = ctx;
    return PT_OK;
}

int pt_create(int device, int32_t width, int32_t height, pt_context** out) {
    return pt_create_tiled(device, width, height, 0, 1, 8, out);
}

void pt_destroy(pt_context* ctx) {
    if (!ctx) return;
    if (ctx->has_device) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        for (auto& e : ctx->events) { if (e.a) (void)hipEventDestroy(e.a); if (e.b) (void)hipEventDestroy(e.b); }
        if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
        if (ctx->d_nodes4) (void)hipFree(ctx->d_nodes4);
        if (ctx->d_stack_ovf) (void)hipFree(ctx->d_stack_ovf);
        if (ctx->d_tris) (void)hipFree(ctx->d_tris);
        if (ctx->d_meta) (void)hipFree(ctx->d_meta);
        if (ctx->d_mats) (void)hipFree(ctx->d_mats);
        if (ctx->d_rays) (void)hipFree(ctx->d_rays);
        if (ctx->d_ldr) (void)hipFree(ctx->d_ldr);
        if (ctx->d_stats) (void)hipFree(ctx->d_stats);
        if (ctx->d_tile_counter) (void)hipFree(ctx->d_tile_counter);
        if (ctx->d_tile_done) (void)hipFree(ctx->d_tile_done);
        if (ctx->d_tile_cost) (void)hipFree(ctx->d_tile_cost);
        if (ctx->d_wf_state) (void)hipFree(ctx->d_wf_state);
        if (ctx->d_wf_queues) (void)hipFree(ctx->d_wf_queues);
        if (ctx->d_wf_counters) (void)hipFree(ctx->d_wf_counters);
        for (int c = 0; c < kWfMaxChains; ++c) {
            if (ctx->wf_stream[c]) (void)hipStreamDestroy(ctx->wf_stream[c]);
            if (ctx->wf_event[c]) (void)hipEventDestroy(ctx->wf_event[c]);
        }
        if (ctx->comm) comm_destroy(ctx->comm);
        if (ctx->d_gathered) (void)hipFree(ctx->d_gathered);
        if (ctx->d_frame) (void)hipFree(ctx->d_frame);
        if (ctx->own_rnds && ctx->d_rnds) (void)hipFree(ctx->d_rnds);
        if (ctx->own_colors && ctx->d_colors) (void)hipFree(ctx->d_colors);
    }
    delete ctx;
}

const char* pt_last_error(const pt_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int pt_device_info(const pt_context* ctx, char* buf, int32_t buflen) {
    if (!ctx || !buf || buflen <= 0) return PT_EINVAL;
    std::snprintf(buf, (size_t)buflen, "%s", ctx->info);
    return PT_OK;
}

int pt_add_material(pt_context* ctx, const pt_material* m) {
    if (!ctx || !m) return PT_EINVAL;
    if (ctx->mats.size() >= 65536) return fail(ctx, PT_EINVAL, "more than 65536 materials (mati is a ushort, prog.cl:20)");
    ctx->mats.push_back(*m);
    ctx->mats_uploaded = false;
    return (int)ctx->mats.size() - 1;
}

int pt_add_triangle(pt_context* ctx, const pt_triangle* t) { return pt_add_triangles(ctx, t, 1); }

int pt_add_triangles(pt_context* ctx, const pt_triangle* t, int64_t n) {
    if (!ctx || (!t && n) || n < 0) return PT_EINVAL;
    if ((int64_t)ctx->tris.size() + n > kMaxTriangles)      // checked before anything is read: 32-bit device offsets (pt_internal.hpp)
        return fail(ctx, PT_EINVAL, "more than 2^26 triangles");
    ctx->tris.insert(ctx->tris.end(), t, t + n);
    ctx->tris_uploaded = false;
    return PT_OK;
}

int pt_upload_materials(pt_context* ctx) {
    if (!ctx) return PT_EINVAL;
    for (const pt_triangle& t : ctx->tris)
        if (t.mati >= ctx->mats.size()) return fail(ctx, PT_EINVAL, "a triangle references a material index that was never added");
    if (ctx->has_device) {
        PT_HIP(ctx, hipSetDevice(ctx->device));
        // device copy: _pad marks materials whose specular lobe is identically zero (ks == 0, finite
        // shininess >= 0): the kernel then skips pow(), the product ks*pow being +0 either way
        std::vector<pt_material> dm(ctx->mats);
        for (pt_material& m : dm)
            m._pad = (m.ks.s[0] == 0.0f && m.ks.s[1] == 0.0f && m.ks.s[2] == 0.0f && std::isfinite(m.shininess) && m.shininess >= 0.0f) ? 1 : 0;
        int rc = upload_vec(ctx, &ctx->d_mats, dm.data(), sizeof(pt_material) * dm.size());
        if (rc != PT_OK) return rc;
    }
    ctx->mats_uploaded = true;
    return PT_OK;
}

int pt_seed_default(pt_context* ctx) {
    PT_NEED_DEVICE(ctx);
    // std::minstd_rand0, default seed 1, drawn in GLOBAL pixel order (main.cpp:45, 522-527)
    const size_t n = (size_t)ctx->W * (size_t)ctx->H;
    std::vector<int32_t> g(n);
    uint64_t x = 1;
    for (size_t i = 0; i < n; ++i) {
        x = (x * 16807ull) % 2147483647ull;
        g[i] = (int32_t)x;
    }
    return seed_upload(ctx, g.data());
}

int pt_upload_seeds(pt_context* ctx, const int32_t* seeds, int64_t n) {
    PT_NEED_DEVICE(ctx);
    if (!seeds || n != (int64_t)ctx->W * ctx->H) return fail(ctx, PT_EINVAL, "seeds must hold width*height ints (global frame)");
    return seed_upload(ctx, seeds);
}

int pt_local_pixel_count(const pt_context* ctx, int64_t* out) {
    if (!ctx || !out) return PT_EINVAL;
    *out = ctx->npix;
    return PT_OK;
}

int pt_slab_pixel_count(const pt_context* ctx, int64_t* out) {
    if (!ctx || !out) return PT_EINVAL;
    *out = ctx->slab_pix;
    return PT_OK;
}

int pt_frame_size(const pt_context* ctx, int32_t* width, int32_t* height, int64_t* npix) {
    if (!ctx) return PT_EINVAL;
    if (width) *width = ctx->W;
    if (height) *height = ctx->H;
    if (npix) *npix = (int64_t)ctx->W * ctx->H;
    return PT_OK;
}

// ---- frame assembly over RCCL (pt_comm.hip)
int pt_comm_available(void) {
    std::string err;
    const int rc = comm_available(&err);
    return rc == PT_OK ? rc : fail(nullptr, rc, err);
}

int pt_comm_unique_id(void* id128) {
    if (!id128) return PT_EINVAL;
    std::string err;
    const int rc = comm_unique_id(id128, &err);
    return rc == PT_OK ? rc : fail(nullptr, rc, err);
}

int pt_comm_init(pt_context* ctx, const void* id128) {
    PT_NEED_DEVICE(ctx);
    if (!id128) return fail(ctx, PT_EINVAL, "id is NULL");
    if (ctx->comm) return fail(ctx, PT_EINVAL, "the context already has a communicator");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    std::string err;
    const int rc = comm_init(id128, ctx->rank, ctx->world, &ctx->comm, &err);
    return rc == PT_OK ? rc : fail(ctx, rc, err);
}

int pt_gather_frame(pt_context* ctx) {
    PT_NEED_DEVICE(ctx);
    if (!ctx->comm) {
        if (ctx->world == 1) return PT_OK;             // the colors buffer IS the frame
        return fail(ctx, PT_EINVAL, "pt_comm_init has not been called on this tiled context");
    }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t slab = (size_t)ctx->slab_pix;
    if (!ctx->d_gathered) PT_HIP(ctx, hipMalloc((void**)&ctx->d_gathered, sizeof(float4) * slab * (size_t)ctx->world));
    if (!ctx->d_frame) PT_HIP(ctx, hipMalloc((void**)&ctx->d_frame, sizeof(float4) * (size_t)ctx->W * (size_t)ctx->H));
    std::string err;
    const int rc = comm_all_gather(ctx->comm, ctx->d_colors, ctx->d_gathered, slab * 4, ctx->stream, &err);
    if (rc != PT_OK) return fail(ctx, rc, err);
    PT_HIP(ctx, launch_deinterleave(ctx->d_gathered, ctx->d_frame, ctx->W, ctx->H, ctx->world, ctx->rows_per_block, (long long)slab, ctx->stream));
    ctx->frame_epoch = ctx->render_epoch;
    return PT_OK;
}

// The assembled frame: the gathered copy while nothing has been rendered since the gather; for a one-rank context the
// colors buffer otherwise (it IS the frame); for a tiled context nothing -- a frame older than colors is never served.
static const float4* current_frame(const pt_context* ctx) {
    if (ctx->d_frame && ctx->frame_epoch == ctx->render_epoch) return ctx->d_frame;
    return ctx->world == 1 ? ctx->d_colors : nullptr;
}

void* pt_device_frame(pt_context* ctx) { return !ctx ? nullptr : (void*)current_frame(ctx); }

int pt_read_frame(pt_context* ctx, float* out, int64_t npix) {
    PT_NEED_DEVICE(ctx);
    if (!out || npix != (int64_t)ctx->W * ctx->H) return fail(ctx, PT_EINVAL, "npix must equal width * height of the global frame");
    const float4* src = current_frame(ctx);
    if (!src) return fail(ctx, PT_EINVAL, ctx->d_frame ? "the gathered frame is older than colors: call pt_gather_frame again" : "pt_gather_frame has not been called");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(out, src, sizeof(float4) * (size_t)npix, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_debug_gather_index(int32_t width, int32_t height, int32_t world, int32_t rows_per_block, int64_t slab_stride, int64_t* out) {
    if (width <= 0 || height <= 0 || world < 1 || rows_per_block < 1 || !out) return PT_EINVAL;
    gather_source_index(width, height, world, rows_per_block, (long long)slab_stride, out);
    return PT_OK;
}

int pt_local_pixel_ids(const pt_context* ctx, int32_t* out, int64_t n) {
    if (!ctx || !out || n != ctx->npix) return PT_EINVAL;
    for (int32_t lr = 0; lr < ctx->local_rows; ++lr) {
        const int32_t gr = global_row(ctx, lr);
        for (int32_t x = 0; x < ctx->W; ++x) out[(size_t)lr * ctx->W + x] = gr * ctx->W + x;
    }
    return PT_OK;
}

static int read_back(pt_context* ctx, void* dst, const void* src, size_t bytes) {
    const int rc = sync_and_check(ctx);
    if (rc != PT_OK) return rc;
    if (bytes) PT_HIP(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_read_colors(pt_context* ctx, float* out, int64_t npix) {
    PT_NEED_DEVICE(ctx);
    if (!out || npix != ctx->npix) return fail(ctx, PT_EINVAL, "npix must equal the local pixel count");
    return read_back(ctx, out, ctx->d_colors, sizeof(float4) * (size_t)npix);
}
int pt_read_rnds(pt_context* ctx, int32_t* out, int64_t npix) {
    PT_NEED_DEVICE(ctx);
    if (!out || npix != ctx->npix) return fail(ctx, PT_EINVAL, "npix must equal the local pixel count");
    return read_back(ctx, out, ctx->d_rnds, sizeof(int32_t) * (size_t)npix);
}
int pt_read_rays(pt_context* ctx, pt_ray* out, int64_t npix) {
    PT_NEED_DEVICE(ctx);
    if (!out || npix != ctx->npix) return fail(ctx, PT_EINVAL, "npix must equal the local pixel count");
    return read_back(ctx, out, ctx->d_rays, sizeof(pt_ray) * (size_t)npix);
}

int pt_resolve_ldr(pt_context* ctx, int32_t which, float* out, int64_t npix) {
    PT_NEED_DEVICE(ctx);
    if (!out || npix != ctx->npix) return fail(ctx, PT_EINVAL, "npix must equal the local pixel count");
    if (which != 0 && which != 1) return fail(ctx, PT_EINVAL, "which must be 0 (Reinhard) or 1 (filt_im)");
    if (which == 1 && ctx->world != 1) return fail(ctx, PT_EINVAL, "filt_im needs the whole frame on one context (3x3 stencil)");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_ldr) PT_HIP(ctx, hipMalloc((void**)&ctx->d_ldr, sizeof(float4) * (size_t)std::max<int64_t>(npix, 1)));
    if (which == 0) {
        PT_HIP(ctx, launch_resolve_reinhard(ctx->d_colors, ctx->d_ldr, npix, ctx->stream));
    } else {
        PT_HIP(ctx, hipMemsetAsync(ctx->d_ldr, 0, sizeof(float4) * (size_t)npix, ctx->stream));
        PT_HIP(ctx, launch_filt_im(ctx->d_colors, ctx->d_ldr, ctx->W, ctx->H, ctx->stream));
    }
    return read_back(ctx, out, ctx->d_ldr, sizeof(float4) * (size_t)npix);
}

int pt_bind_framebuffer(pt_context* ctx, void* d_colors, void* d_rnds) {
    PT_NEED_DEVICE(ctx);
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (d_colors && d_colors != ctx->d_colors) {
        PT_HIP(ctx, hipMemcpy(d_colors, ctx->d_colors, sizeof(float4) * (size_t)ctx->npix, hipMemcpyDeviceToDevice));
        if (ctx->own_colors) PT_HIP(ctx, hipFree(ctx->d_colors));
        ctx->d_colors = (float4*)d_colors;
        ctx->own_colors = false;
    }
    if (d_rnds && d_rnds != ctx->d_rnds) {
        PT_HIP(ctx, hipMemcpy(d_rnds, ctx->d_rnds, sizeof(int32_t) * (size_t)ctx->npix, hipMemcpyDeviceToDevice));
        if (ctx->own_rnds) PT_HIP(ctx, hipFree(ctx->d_rnds));
        ctx->d_rnds = (int32_t*)d_rnds;
        ctx->own_rnds = false;
    }
    return PT_OK;
}

void* pt_device_colors(pt_context* ctx) { return ctx ? (void*)ctx->d_colors : nullptr; }
void* pt_device_rnds(pt_context* ctx) { return ctx ? (void*)ctx->d_rnds : nullptr; }

int pt_set_stream(pt_context* ctx, void* s) {
    PT_NEED_DEVICE(ctx);
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)s;
    return PT_OK;
}

int pt_set_option(pt_context* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return PT_EINVAL;
    std::string k(key);
    if (k == "variant") {
        if (value != 0 && value != 1) return fail(ctx, PT_EINVAL, "variant must be 0 (megakernel) or 1 (wavefront)");
        ctx->variant = (int)value;
    } else if (k == "lds_scene") {
        if (value != 0 && value != 2) return fail(ctx, PT_EINVAL, "lds_scene: 0 every node through L1/L2, 2 stage the tree (or its top) in LDS");
        ctx->lds_scene = (int)value;
    } else if (k == "treelet") {
        if (value < -1 || value > 2048) return fail(ctx, PT_EINVAL, "treelet: 0 off, -1 as many nodes as fit, 2..2048 nodes");
        ctx->treelet = (int)value;
        ctx->tris_uploaded = false;                  // the tree is re-indexed at upload
    } else if (k == "lbvh_cluster") {
        if (value < 0 || value > (1 << 20)) return fail(ctx, PT_EINVAL, "lbvh_cluster: 0 off, 1..2^20 triangles");
        ctx->lbvh_cluster = (int)value;
        ctx->tris_uploaded = false;
    } else if (k == "build_threads") {
        if (value < 0 || value > 256) return fail(ctx, PT_EINVAL, "build_threads: 0 automatic, 1..256");
        ctx->build_threads = (int)value;
    } else if (k == "wide_nodes") {
        if (value < 0 || value > 2) return fail(ctx, PT_EINVAL, "wide_nodes: 0 never, 1 for trees that do not fit LDS, 2 always");
        ctx->wide_nodes = (int)value;
        ctx->tris_uploaded = false;                  // the wide nodes are built at upload
    } else if (k == "wide_lds_entries") {
        if (value < 4 || value > kWideLdsEntries || (value & 1)) return fail(ctx, PT_EINVAL, "wide_lds_entries: even, 4..20");
        ctx->wide_lds_entries = (int)value;
        ctx->tris_uploaded = false;                  // the global part of the stacks is sized at upload
    } else if (k == "timing") {
        ctx->timing = value ? 1 : 0;
    } else if (k == "count_work") {
        ctx->count_work = value ? 1 : 0;
    } else if (k == "chunk_taper") {
        if (value < -1 || value > 1 << 15) return fail(ctx, PT_EINVAL, "chunk_taper: -1 default, 0 off, else the shortest pass");
        ctx->chunk_taper = (int)value;
    } else if (k == "chunk_spp") {
        if (value < -1 || value > 1 << 20) return fail(ctx, PT_EINVAL, "chunk_spp out of range");
        ctx->chunk_spp = (int)value;
    } else if (k == "persistent") {
        ctx->persistent = value ? 1 : 0;
    } else if (k == "sah_visit_cost") {
        if (value < 0 || value > 1000) return fail(ctx, PT_EINVAL, "sah_visit_cost: tenths of a triangle test, 0..1000");
        ctx->sah_visit_cost = (int)value;
    } else if (k == "schedule") {
        if (value < -1 || value > 2) return fail(ctx, PT_EINVAL, "schedule: -1 automatic, 0 lockstep per sample, 1 restart + tail suspension, 2 the same with lanes moving on to the wave's next work item");
        ctx->schedule = (int)value;
    } else if (k == "flat_list") {
        if (value < 0 || value > 32) return fail(ctx, PT_EINVAL, "flat_list: 0..32 big triangles tested before the tree (a 32-bit candidate mask per lane)");
        ctx->flat_list = (int)value;
        ctx->tris_uploaded = false;
    } else if (k == "poll_timeout_ms") {
        if (value < 1 || value > 40000) return fail(ctx, PT_EINVAL, "poll_timeout_ms: 1..40000");
        ctx->poll_timeout_ms = (int)value;
    } else if (k == "debug_stall_tile") {
        if (value < -1 || value > 0x7fffffff) return fail(ctx, PT_EINVAL, "debug_stall_tile: -1 none, or a tile index");
        ctx->debug_stall_tile = (int)value;
    } else if (k == "wf_streams") {
        if (value < -1 || value == 0 || value > kWfMaxChains) return fail(ctx, PT_EINVAL, "wf_streams: -1 default, 1..8");
        ctx->wf_streams = (int)value;
    } else if (k == "node_min_lanes" || k == "leaf_min_lanes") {
        if (value < -1 || value > 63) return fail(ctx, PT_EINVAL, k + ": -1 default, 0..63");
        (k == "node_min_lanes" ? ctx->node_min_lanes : ctx->leaf_min_lanes) = (int)value;
    } else if (k == "migrate_lanes") {
        if (value != -1 && (value < 1 || value > 64)) return fail(ctx, PT_EINVAL, "migrate_lanes: -1 default, 1..64");
        ctx->migrate_lanes = (int)value;
    } else if (k == "suspend_lanes") {
        if (value < -1 || value > 63) return fail(ctx, PT_EINVAL, "suspend_lanes: -1 default, 0..63");
        ctx->suspend_lanes = (int)value;
    } else if (k == "waves_per_simd") {
        if (value != -1 && (value < 4 || value > 8)) return fail(ctx, PT_EINVAL, "waves_per_simd: -1 automatic (at most 7), 4..8 (kernels that read nodes from global memory)");
        ctx->waves_per_simd = (int)value;
    } else if (k == "bvh_device") {
        if (value < -1 || value > 1) return fail(ctx, PT_EINVAL, "bvh_device: -1 (by scene size), 0 (host) or 1 (device)");
        ctx->bvh_device = (int)value;
    } else if (k == "wide_on_device") {
        if (value != 0 && value != 1) return fail(ctx, PT_EINVAL, "wide_on_device: 0 or 1");
        ctx->wide_on_device = (int)value;
    } else if (k == "sah_grain") {
        if (value < 8 || value > (1 << 16)) return fail(ctx, PT_EINVAL, "sah_grain: 8..65536 triangles");
        ctx->sah_grain = (int)value;
    } else if (k == "lbvh_ploc") {
        if (value != 0 && value != 8 && value != 16 && value != 32) return fail(ctx, PT_EINVAL, "lbvh_ploc: 0 (radix tree), 8, 16 or 32 (PLOC search radius)");
        ctx->lbvh_ploc = (int)value;
    } else if (k == "lds_block") {
        if (value != -1 && value != kLdsBlockBase && value != kLdsBlockWide) return fail(ctx, PT_EINVAL, "lds_block: -1 automatic, 512 or 768");
        ctx->lds_block = (int)value;
    } else if (k == "debug_repeat") {
        if (value < 0 || value > 1000) return fail(ctx, PT_EINVAL, "debug_repeat: 0..1000 extra timed launches");
        ctx->debug_repeat = (int)value;
    } else if (k == "cost_binning") {
        ctx->cost_binning = value ? 1 : 0;
    } else if (k == "bvh_policy") {
        if (value < 0 || value > 5) return fail(ctx, PT_EINVAL, "bvh_policy must be 0..5 (4 = device LBVH, 5 = the SAH tree built on the device)");
        ctx->bvh_policy = (int)value;
        ctx->tris_uploaded = false;
    } else if (k == "reset_stats") {
        if (ctx->has_device) {
            PT_HIP(ctx, hipSetDevice(ctx->device));
            PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            PT_HIP(ctx, hipMemset(ctx->d_stats, 0, sizeof(unsigned long long) * kStatCols * kStatRows));
        }
        int rc = time_collect(ctx);
        if (rc != PT_OK) return rc;
        ctx->kernel_ms_acc = 0.0;
        ctx->kernel_launches = 0;
    } else {
        return fail(ctx, PT_EINVAL, "unknown option: " + k);
    }
    return PT_OK;
}

int pt_get_stat(pt_context* ctx, const char* key, double* out) {
    if (!ctx || !key || !out) return PT_EINVAL;
    std::string k(key);
    if (k == "bvh_nodes") { *out = (double)ctx->nodes.size(); return PT_OK; }
    if (k == "bvh_depth") { *out = (double)ctx->bvh_depth; return PT_OK; }
    if (k == "stack_entries") { *out = (double)stack_entries_for(ctx->interior_depth); return PT_OK; }
    if (k == "bvh_build_ms") { *out = ctx->bvh_build_ms; return PT_OK; }
    if (k == "bvh_on_device") { *out = (double)ctx->bvh_on_device; return PT_OK; }
    if (k == "triangles") { *out = (double)ctx->orig.size(); return PT_OK; }
    if (k == "lds_bytes") { *out = (double)ctx->last_lds_bytes; return PT_OK; }
    if (k == "waves_per_simd") { *out = (double)ctx->last_waves_per_simd; return PT_OK; }
    if (k == "treelet_nodes") { *out = (double)ctx->treelet_nodes; return PT_OK; }
    if (k == "wide_nodes") { *out = (double)ctx->nodes4.size(); return PT_OK; }
    if (k == "wide_pending") { *out = (double)ctx->wide_pending; return PT_OK; }
    if (k == "flat_triangles") { *out = (double)ctx->n_flat; return PT_OK; }
    if (k == "flat_boxes") { *out = (double)ctx->n_fbox; return PT_OK; }
    if (k == "node_mode") {      // what the next launch will use: 0 whole tree in LDS, 1 L1/L2 only, 2 treelet
        pt_camera cam;
        std::memset(&cam, 0, sizeof cam);
        RenderParams p;
        fill_params(ctx, &cam, &p);
        *out = (double)p.node_mode;
        return PT_OK;
    }
    if (k == "kernel_launches") { *out = (double)ctx->kernel_launches; return PT_OK; }
    PT_NEED_DEVICE(ctx);
    PT_HIP(ctx, hipSetDevice(ctx->device));
    if (k == "kernel_ms") {
        int rc = time_collect(ctx);
        if (rc != PT_OK) return rc;
        *out = ctx->kernel_ms_acc;
        return PT_OK;
    }
    if (k == "segments" || k == "samples" || k == "node_visits" || k == "tri_tests" || k == "wave_node_steps" || k == "wave_tri_steps" || k == "tile_lane_steps" ||
        k == "wave_shade_steps" || k == "wave_trips" || k == "wave_rounds" || k.rfind("low_", 0) == 0) {
        std::vector<unsigned long long> rows((size_t)kStatCols * kStatRows);
        unsigned long long h[kStatCols] = {};
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PT_HIP(ctx, hipMemcpy(rows.data(), ctx->d_stats, sizeof(unsigned long long) * rows.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < kStatRows; ++r)
            for (int c = 0; c < kStatCols; ++c) h[c] += rows[(size_t)r * kStatCols + c];
        // low_node / low_tri / low_exact / low_shade: executions of that body for at most 8 lanes; exact_steps: all executions of the
        // exact part of the triangle test (counting instances)
        if (k.rfind("low_", 0) == 0 || k == "exact_steps") {
            const int slot = k == "low_node" ? 10 : k == "low_tri" ? 11 : k == "low_exact" ? 12 : k == "low_shade" ? 13 : k == "low_exact_all" ? 15 : -1;
            if (slot < 0) return fail(ctx, PT_EINVAL, "unknown stat " + k);
            *out = (double)h[slot];
            return PT_OK;
        }
        *out = (double)h[k == "segments" ? 0 : k == "samples" ? 1 : k == "node_visits" ? 2 : k == "tri_tests" ? 3 : k == "wave_node_steps" ? 4 : k == "wave_tri_steps" ? 5 : k == "tile_lane_steps" ? 6 : k == "wave_shade_steps" ? 7 : k == "wave_trips" ? 8 : 9];
        return PT_OK;
    }
    return fail(ctx, PT_EINVAL, "unknown stat: " + k);
}

int pt_debug_bvh_sizes(const pt_context* ctx, int64_t* nnodes, int64_t* ntris) {
    if (!ctx) return PT_EINVAL;
    if (nnodes) *nnodes = (int64_t)ctx->nodes.size();
    if (ntris) *ntris = (int64_t)ctx->orig.size();
    return PT_OK;
}

int pt_debug_wide_nodes(const pt_context* ctx, void* out, int64_t capacity, int64_t* count) {
    if (!ctx || !count || capacity < 0) return PT_EINVAL;
    *count = (int64_t)ctx->nodes4.size();
    if (out) std::memcpy(out, ctx->nodes4.data(), sizeof(Node4q) * (size_t)std::min<int64_t>(capacity, *count));
    return PT_OK;
}

int pt_debug_bvh_copy(const pt_context* cctx, float* nodes, float* tris, int32_t* meta, int32_t* orig) {
    if (!cctx || !cctx->tris_uploaded) return PT_EINVAL;
    pt_context* ctx = const_cast<pt_context*>(cctx);            // (the host mirror of a device-built tree is filled on demand)
    if (ctx->host_packets_stale && (tris || meta)) {
        const size_t m = ctx->orig.size();
        ctx->packets.resize(m);
        ctx->meta.resize(m);
        PT_HIP(ctx, hipSetDevice(ctx->device));
        PT_HIP(ctx, hipMemcpy(ctx->packets.data(), ctx->d_tris, sizeof(TriPacket) * m, hipMemcpyDeviceToHost));
        PT_HIP(ctx, hipMemcpy(ctx->meta.data(), ctx->d_meta, sizeof(TriMeta) * m, hipMemcpyDeviceToHost));
        ctx->host_packets_stale = false;
    }
    if (nodes) std::memcpy(nodes, ctx->nodes.data(), sizeof(Node64) * ctx->nodes.size());
    if (tris) std::memcpy(tris, ctx->packets.data(), sizeof(TriPacket) * ctx->orig.size());
    if (meta) std::memcpy(meta, ctx->meta.data(), sizeof(TriMeta) * ctx->orig.size());
    if (orig) std::memcpy(orig, ctx->orig.data(), sizeof(int32_t) * ctx->orig.size());
    return PT_OK;
}

namespace {
struct DeviceBuf {          // frees on every exit path
    void* p = nullptr;
    ~DeviceBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
};
struct EventOwner {
    hipEvent_t e = nullptr;
    ~EventOwner() { if (e) (void)hipEventDestroy(e); }
};
}  // namespace

int pt_debug_closest_hit(pt_context* ctx, const pt_ray* rays, int64_t n, float* out_t, int32_t* out_tri) {
    PT_NEED_DEVICE(ctx);
    if (!rays || !out_t || !out_tri || n < 0) return fail(ctx, PT_EINVAL, "bad arguments");
    if (!ctx->tris_uploaded) return fail(ctx, PT_EINVAL, "pt_upload_triangles has not been called");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    pt_camera cam;
    std::memset(&cam, 0, sizeof cam);
    RenderParams p;
    fill_params(ctx, &cam, &p);         // same node placement as the render kernels would use
    DeviceBuf d_rays, d_t, d_tri;
    PT_HIP(ctx, d_rays.alloc(sizeof(pt_ray) * (size_t)n));
    PT_HIP(ctx, d_t.alloc(sizeof(float) * (size_t)n));
    PT_HIP(ctx, d_tri.alloc(sizeof(int32_t) * (size_t)n));
    if (n) PT_HIP(ctx, hipMemcpy(d_rays.p, rays, sizeof(pt_ray) * (size_t)n, hipMemcpyHostToDevice));
    ctx->last_lds_bytes = traversal_lds_bytes(p, traversal_block(p.node_mode, false));
    PT_HIP(ctx, launch_debug_closest_hit(p, (const pt_ray*)d_rays.p, n, (float*)d_t.p, (int32_t*)d_tri.p, ctx->cu_count, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->debug_repeat > 0) {      // traversal-only timing: the same launch, debug_repeat times
        EventOwner e0, e1;
        PT_HIP(ctx, hipEventCreate(&e0.e));
        PT_HIP(ctx, hipEventCreate(&e1.e));
        PT_HIP(ctx, hipEventRecord(e0.e, ctx->stream));
        for (int r = 0; r < ctx->debug_repeat; ++r)
            PT_HIP(ctx, launch_debug_closest_hit(p, (const pt_ray*)d_rays.p, n, (float*)d_t.p, (int32_t*)d_tri.p, ctx->cu_count, ctx->stream));
        PT_HIP(ctx, hipEventRecord(e1.e, ctx->stream));
        PT_HIP(ctx, hipEventSynchronize(e1.e));
        float ms = 0.f;
        PT_HIP(ctx, hipEventElapsedTime(&ms, e0.e, e1.e));
        ctx->kernel_ms_acc += ms;
        ctx->kernel_launches += ctx->debug_repeat;
    }
    if (n) {
        PT_HIP(ctx, hipMemcpy(out_t, d_t.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
        PT_HIP(ctx, hipMemcpy(out_tri, d_tri.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    }
    for (int64_t i = 0; i < n; ++i)
        if (out_tri[i] >= 0) out_tri[i] = ctx->orig[(size_t)out_tri[i]];      // packed -> add order
    return PT_OK;
}

int pt_debug_deinterleave(pt_context* ctx, const float* gathered, int64_t n_pixels, float* out_frame) {
    PT_NEED_DEVICE(ctx);
    if (!gathered || !out_frame || n_pixels != ctx->slab_pix * ctx->world) return fail(ctx, PT_EINVAL, "gathered must hold world x slab pixels");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    DeviceBuf d_g, d_f;
    const size_t nf = (size_t)ctx->W * (size_t)ctx->H;
    PT_HIP(ctx, d_g.alloc(sizeof(float4) * (size_t)n_pixels));
    PT_HIP(ctx, d_f.alloc(sizeof(float4) * nf));
    PT_HIP(ctx, hipMemcpy(d_g.p, gathered, sizeof(float4) * (size_t)n_pixels, hipMemcpyHostToDevice));
    PT_HIP(ctx, launch_deinterleave((const float4*)d_g.p, (float4*)d_f.p, ctx->W, ctx->H, ctx->world, ctx->rows_per_block, (long long)ctx->slab_pix, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(out_frame, d_f.p, sizeof(float4) * nf, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_debug_scene_sizes(const pt_context* ctx, int64_t* ntris, int64_t* nmats, int64_t* nobjs) {
    if (!ctx) return PT_EINVAL;
    if (ntris) *ntris = (int64_t)ctx->tris.size();
    if (nmats) *nmats = (int64_t)ctx->mats.size();
    if (nobjs) *nobjs = (int64_t)ctx->obj_begin.size();
    return PT_OK;
}

int pt_debug_scene_copy(const pt_context* ctx, pt_triangle* tris, pt_material* mats, int32_t* obj_begin) {
    if (!ctx) return PT_EINVAL;
    if (tris && !ctx->tris.empty()) std::memcpy(tris, ctx->tris.data(), sizeof(pt_triangle) * ctx->tris.size());
    if (mats && !ctx->mats.empty()) std::memcpy(mats, ctx->mats.data(), sizeof(pt_material) * ctx->mats.size());
    if (obj_begin && !ctx->obj_begin.empty()) std::memcpy(obj_begin, ctx->obj_begin.data(), sizeof(int32_t) * ctx->obj_begin.size());
    return PT_OK;
}

int pt_debug_tile_cost(pt_context* ctx, uint32_t* out, int64_t n) {
    PT_NEED_DEVICE(ctx);
    const int64_t n_tiles = (int64_t)((ctx->W + 7) / 8) * ((ctx->local_rows + 7) / 8);
    if (!out || n != n_tiles) return fail(ctx, PT_EINVAL, "pt_debug_tile_cost: n must be the number of 8x8 tiles of the local frame");
    if (!ctx->d_tile_cost) return fail(ctx, PT_EINVAL, "pt_debug_tile_cost: no counting launch yet (option count_work, then pt_render)");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(out, ctx->d_tile_cost, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_debug_encounter_rank(const pt_context* ctx, int32_t* out, int64_t n) {
    if (!ctx || !out || n != (int64_t)ctx->enc_rank.size()) return PT_EINVAL;
    std::memcpy(out, ctx->enc_rank.data(), sizeof(int32_t) * (size_t)n);
    return PT_OK;
}

}  // extern "C"
