/*
 * pt_oracle.h -- CPU ORACLE for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (opencl_path_tracer_amd/) never links, imports or calls it.
 *
 * What it is: a plain-C restatement of the reference's algorithm for the path
 *   gen_ray -> trace_ray (prog.cl:384-389, prog.cl:292-381) and of the host-side data
 *   producers that feed it (main.cpp:92-348, 522-551).  Every function cites the
 *   reference lines it follows ("prog.cl:NN" = /root/reference/prog.cl).
 *
 * PARITY UNPINNED (vs an execution of the reference): the reference ships no tests,
 *   fixtures or golden images, and it cannot be built in this image (main.cpp needs
 *   windows.h / GL/glut.h / a GL+OpenCL runtime; prog.cl needs an OpenCL C builtin
 *   library and a CPU OpenCL device -- none exist here).  What IS pinned, by
 *   tests/test_oracle_*.py: the seed sequence against libstdc++'s std::minstd_rand0
 *   (the very generator main.cpp:45 uses), the C++-standard known answers for both
 *   LCGs, the OpenCL struct layouts, analytic radiance cases read off prog.cl,
 *   internal consistency (heap-array traversal == pointer traversal == brute force), and
 *   -- round 3 -- an independent float64 numpy model of the whole path, replayed per pixel
 *   from the LCG stream on the Cornell box at eight bounces (tests/test_gpu_closed_form.py
 *   replay_model): same colours within 2e-4 and the same final LCG state in every pixel
 *   clear of float32 / float64 ties.  A second reading of prog.cl, not a run of it.
 *
 * Arithmetic contract ("the spec").  OpenCL C leaves the
 * precision of '/', sqrt, sin, cos, pow and the placement of fused multiply-adds to
 * the implementation.  Paths are chaotic (one flipped branch permanently forks a
 * pixel's RNG stream), so this oracle fixes ONE conforming choice, and the HIP kernels
 * follow the same choice independently so that results are comparable bit for bit:
 *   - all f32 add/sub/mul/div/sqrt are IEEE-754 round-to-nearest-even, denormals kept;
 *   - dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x));
 *   - cross(a,b) = (fma(a.y,b.z,-(a.z*b.y)), fma(a.z,b.x,-(a.x*b.z)), fma(a.x,b.y,-(a.y*b.x)));
 *   - normalize(v) = v * (1.0f / sqrtf(dot(v,v)));
 *   - a literal "u*s + w" / "w + u*s" / "w - u*s" in prog.cl is one fma;
 *   - half_sqrt = IEEE sqrtf; pow(x,5) = (x*x)*(x*x)*x; general pow, sin, cos are the
 *     double-precision polynomial routines orc_spec_powf / orc_spec_sincosf below
 *     (<= 1 ulp of the exact value; OpenCL allows 16 / 4 / 4 ulp);
 *   - max(0,c) is (c > 0 ? c : +0) so the sign of zero is defined;
 *   - host-side code (main.cpp) is plain x86-64 g++ arithmetic: no fma, and
 *     unqualified sin/cos/tan/sqrt on float arguments evaluate in double.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device-mirror PODs: prog.cl:1-35, main.cpp:92-193, 306-348.  cl_float3 = 16 B. */
typedef struct { float x, y, z, w; } orc_f3;                       /* float3 (16 B) */
typedef struct { orc_f3 kd, ks, emission, F0; float n, shininess; int32_t type; int32_t _pad; } orc_material; /* 80 B */
typedef struct { orc_f3 P, D; } orc_ray;                            /* 32 B */
typedef struct { orc_f3 r1, r2, r3, N; uint16_t mati; uint8_t _pad[14]; } orc_triangle;   /* 80 B */
typedef struct { orc_f3 bl, tr; } orc_bbox;                         /* 32 B */
typedef struct { int32_t trii[2]; int32_t _pad[2]; orc_bbox bbox; } orc_node;             /* 48 B */
typedef struct { orc_f3 eye, lookat, up, right; float XM, YM; float _pad[2]; } orc_camera; /* 80 B */
typedef struct { float t; float _p0[3]; orc_f3 P, N; uint16_t mati; uint8_t _p1[14]; orc_material mat; } orc_hit; /* 144 B */

typedef struct orc_scene orc_scene;   /* host Scene: main.cpp:363-387 */
typedef struct orc_frame orc_frame;   /* rays / rnds / colors buffers: main.cpp:508-527 */

/* sizes/offsets table for the layout test */
int  orc_layout(int which);           /* see pt_oracle.c for the index list */

/* ---- RNG: main.cpp:45,522-527 (host seeds) and prog.cl:72-77 (device LCG) */
void  orc_seed_sequence(int32_t* out, int64_t n);          /* i-th output of default minstd_rand0 */
float orc_rand(int32_t* seed);

/* ---- spec math */
void  orc_spec_sincosf(float theta, float* s, float* c);
float orc_spec_powf(float x, float y);
float orc_spec_pow5(float x);

/* ---- host-side constructors */
void orc_material_make(orc_material* m, const float kd[3], const float ks[3], const float em[3],
                       const float N[3], const float K[3], float shininess, int type);   /* main.cpp:101-111 */
void orc_triangle_make(orc_triangle* t, const float r1[3], const float r2[3], const float r3[3], int mati); /* main.cpp:144-166 */
void orc_camera_make(orc_camera* c, float fov, float yaw, float pitch, const float shift[3],
                     int width, int height);                                           /* main.cpp:311-347 */
void orc_camera_move(float shift[3], float yaw, float pitch, float forward, float rightward, float upward);   /* main.cpp:323-336 */

void orc_obj_vertex(float out[3], const float v[3], const float pos[3], const float scale[3],
                    float pitch, float yaw);                                         /* main.cpp:598-606 */

/* ---- Scene: main.cpp:529-551, 618-634 */
orc_scene* orc_scene_create(void);
void orc_scene_destroy(orc_scene*);
int  orc_add_material(orc_scene*, const orc_material*);
void orc_add_triangle(orc_scene*, const orc_triangle*);
void orc_add_triangles(orc_scene*, const float* verts9, const uint16_t* mati, int64_t n);   /* n x Triangle(...) + add_Triangle */
int  orc_end_obj(orc_scene*);                 /* 0 ok; <0 = reference would not terminate / overflow */
int  orc_scene_counts(const orc_scene*, int* ntris, int* nnodes, int* nobj, int* nmats);
const orc_triangle* orc_scene_tris(const orc_scene*);      /* leaf-ordered, main.cpp:548-549 */
const orc_node*     orc_scene_nodes(const orc_scene*);     /* heap-indexed kd_tree (may be NULL if infeasible) */
const int32_t*      orc_scene_shifts(const orc_scene*);
const orc_material* orc_scene_mats(const orc_scene*);
/* position of every ORIGINAL triangle (in add order) in the reference's traversal
 * encounter order (object, then depth-first left-first leaf order, then index in leaf) */
void orc_scene_encounter_rank(const orc_scene*, int32_t* rank_out);

/* ---- unit-level device functions (for known-answer tests) */
void orc_camera_get_ray(orc_ray* out, int id, const orc_camera* cam, float rnd1, float rnd2); /* prog.cl:82-92 */
void orc_triangle_intersect(orc_hit* out, const orc_triangle* tri, const orc_ray* ray);       /* prog.cl:94-112 */
int  orc_bbox_intersection(const orc_bbox* box, const orc_ray* ray, float* tmin, float* tmax); /* prog.cl:123-143 */
void orc_kd_intersect(orc_hit* out, const orc_scene*, const orc_ray* ray, int mode);
        /* mode 0: heap array as prog.cl:144-184; 1: pointer tree, same order;
           2: brute force over all triangles, ties -> lowest encounter rank */
void orc_new_ray_diffuse(orc_ray* out, const orc_f3* P, const orc_f3* N, float rnd1, float rnd2); /* prog.cl:205-218 */
void orc_new_ray_specular(orc_ray* out, const orc_f3* P, const orc_f3* N, const orc_ray* old);     /* prog.cl:223-227 */
void orc_new_ray_refractive(orc_ray* out, const orc_f3* P, const orc_f3* N, const orc_f3* F0, float n,
                            const orc_ray* old, int* in, float rnd);                            /* prog.cl:228-245 */
void orc_fresnel(orc_f3* out, const orc_f3* F0, const orc_f3* N, const orc_f3* D);               /* prog.cl:219-222 */
void orc_reinhard_tone_map(float out[4], const float c[3]);                                      /* prog.cl:264-269 */
void orc_filmic_tone_map(float out[4], const float c[3]);                                        /* prog.cl:259-263 */

/* ---- frame buffers + kernels */
orc_frame* orc_frame_create(int width, int height);
void orc_frame_destroy(orc_frame*);
void orc_frame_seed_default(orc_frame*);                 /* main.cpp:522-527 */
int32_t* orc_frame_rnds(orc_frame*);
orc_ray* orc_frame_rays(orc_frame*);
orc_f3*  orc_frame_colors(orc_frame*);
float*   orc_frame_tex(orc_frame*);                      /* RGBA32F, what write_imagef stored */
/* one kernel launch each, over ALL pixels, on nthreads host threads (rows are independent) */
void orc_gen_ray(orc_frame*, const orc_camera*, int nthreads);                                     /* prog.cl:384-389 */
void orc_trace_ray(orc_frame*, const orc_scene*, const orc_camera*, int iterations,
                   int current_sample, int mode, int nthreads);                                    /* prog.cl:292-381 */
/* Scene::render() x nsamples (main.cpp:683-687), current_sample = first_sample .. +nsamples-1.
 * Returns the number of path segments executed (closest-hit queries), for d-bar.       */
int64_t orc_render(orc_frame*, const orc_scene*, const orc_camera*, int iterations,
                   int first_sample, int nsamples, int mode, int nthreads);
/* filt_im (prog.cl:391-427) with the out-of-range reads at the right/top edge skipped */
void orc_filt_im(orc_frame*, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
