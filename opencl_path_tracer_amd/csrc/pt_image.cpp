// pt_image.cpp -- image files of the radiance buffer.  Replaces what the reference
// shows through its GL blit (main.cpp:1019-1039: the RGBA32F texture trace_ray writes with
// write_imagef, prog.cl:380, drawn with texture row 0 at the BOTTOM of the window).
//   PFM: the HDR running mean `colors` itself, 3 x f32 per pixel, little-endian, rows bottom-to-top --
//        which is exactly the buffer's own row order (camera_get_ray, prog.cl:82-92: row 0 is the
//        bottom of the view), so the file is a dump of the parity target.
//   PPM: an LDR resolve (reinhard_tone_map + sRGB of prog.cl:247-269, or filt_im), binary P6, rows
//        top-to-bottom.  Channels are clamped to [0, 1] as the 8-bit GL framebuffer does; the NaN the
//        reference's tone map produces for black pixels (L = 0: 0 * 0 / 0, prog.cl:265-267) is written
//        as 0.
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "pt_internal.hpp"

namespace ptamd {
int fail_ctx(pt_context* ctx, int code, const std::string& msg);   // pt_host.cpp
}

extern "C" {

int pt_image_write_pfm(const char* path, const float* rgba, int32_t width, int32_t height) {
    if (!path || !rgba || width <= 0 || height <= 0) return PT_EINVAL;
    FILE* f = std::fopen(path, "wb");
    if (!f) return PT_EIO;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
    std::vector<float> row((size_t)width * 3);
    bool ok = true;
    for (int32_t y = 0; y < height && ok; ++y) {
        const float* src = rgba + (size_t)y * width * 4;
        for (int32_t x = 0; x < width; ++x) { row[3 * x] = src[4 * x]; row[3 * x + 1] = src[4 * x + 1]; row[3 * x + 2] = src[4 * x + 2]; }
        ok = std::fwrite(row.data(), sizeof(float), row.size(), f) == row.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? PT_OK : PT_EIO;
}

int pt_image_write_ppm(const char* path, const float* rgba, int32_t width, int32_t height) {
    if (!path || !rgba || width <= 0 || height <= 0) return PT_EINVAL;
    FILE* f = std::fopen(path, "wb");
    if (!f) return PT_EIO;
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::vector<unsigned char> row((size_t)width * 3);
    bool ok = true;
    for (int32_t y = height - 1; y >= 0 && ok; --y) {             // buffer row 0 is the bottom of the view
        const float* src = rgba + (size_t)y * width * 4;
        for (int32_t x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) {
                float v = src[4 * x + c];
                if (!(v > 0.0f)) v = 0.0f;                         // NaN (black pixel, prog.cl:265-267) and negatives
                if (v > 1.0f) v = 1.0f;
                row[3 * x + c] = (unsigned char)std::lrintf(v * 255.0f);
            }
        ok = std::fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? PT_OK : PT_EIO;
}

int pt_write_pfm(pt_context* ctx, const char* path) {
    if (!ctx || !path) return PT_EINVAL;
    int64_t n = 0;
    int32_t W = 0, H = 0;
    int rc = pt_frame_size(ctx, &W, &H, &n);
    if (rc != PT_OK) return rc;
    std::vector<float> buf((size_t)n * 4);
    if ((rc = pt_read_frame(ctx, buf.data(), n)) != PT_OK) return rc;
    rc = pt_image_write_pfm(path, buf.data(), W, H);
    return rc == PT_OK ? rc : ptamd::fail_ctx(ctx, rc, std::string("cannot write ") + path);
}

int pt_write_ppm(pt_context* ctx, const char* path, int32_t which) {
    if (!ctx || !path) return PT_EINVAL;
    int64_t n = 0;
    int32_t W = 0, H = 0;
    int rc = pt_frame_size(ctx, &W, &H, &n);
    if (rc != PT_OK) return rc;
    int64_t local = 0;
    if ((rc = pt_local_pixel_count(ctx, &local)) != PT_OK) return rc;
    if (local != n) return ptamd::fail_ctx(ctx, PT_EINVAL, "pt_write_ppm resolves the local pixels: it needs a context that owns the whole frame (world = 1)");
    std::vector<float> buf((size_t)n * 4);
    if ((rc = pt_resolve_ldr(ctx, which, buf.data(), n)) != PT_OK) return rc;
    rc = pt_image_write_ppm(path, buf.data(), W, H);
    return rc == PT_OK ? rc : ptamd::fail_ctx(ctx, rc, std::string("cannot write ") + path);
}

}  // extern "C"
