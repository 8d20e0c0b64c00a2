/*
 * pt_api.h -- C ABI of the MI355X-native path-tracing hot path (libptamd.so).
 *
 * Drop-in boundary: the reference's host class `Scene` (main.cpp:363-742) and the value
 * types it is fed (main.cpp:92-193, 306-348).  Every entry point below names the
 * reference interface it replaces ("main.cpp:NN" = /root/reference/main.cpp,
 * "prog.cl:NN" = /root/reference/prog.cl).  Call order is the reference's:
 *
 *   pt_create                       (Scene::init_Scene)
 *   pt_add_material*                (Scene::add_Material)
 *   (pt_add_triangle* pt_end_obj)*  (Scene::add_Triangle / Scene::end_Obj)   | pt_add_obj*
 *   pt_upload_triangles             (Scene::upload_Triangles)
 *   pt_upload_materials             (Scene::upload_Materials)
 *   pt_render* | (pt_generate_rays pt_trace_rays)*        (Scene::render / generate_rays / trace_rays)
 *   pt_read_colors / pt_read_rnds / pt_read_rays / pt_resolve_ldr
 *   pt_destroy
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  Every function returns
 * PT_OK (0) or a negative PT_E* code and never calls exit(); the message is available from
 * pt_last_error().  Upload/add functions copy (the caller keeps ownership of its arrays);
 * read functions fill caller-provided host buffers.  One host thread per context; one
 * context per GPU.  Nothing here runs on the CPU as a fallback: without a usable HIP
 * device pt_create fails with PT_ENODEVICE.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_EINVAL (-1)     /* bad argument / call order */
#define PT_ENODEVICE (-2)  /* no HIP device, or the device is not gfx950-compatible */
#define PT_EHIP (-3)       /* a HIP runtime call failed (text in pt_last_error) */
#define PT_ESCENE (-4)     /* scene cannot be built (e.g. the reference's build would not terminate) */
#define PT_EIO (-5)        /* a file could not be read, parsed or written */
#define PT_ECOMM (-6)      /* RCCL is unavailable or a collective failed (text in pt_last_error) */

/* ---- value types, byte-compatible with the reference's device structs ------------ */
typedef struct { float s[4]; } pt_float3;                 /* cl_float3: 16 B, .s[3] is padding */
typedef struct {                                           /* Material: prog.cl:1-5, main.cpp:92-112 (80 B) */
    pt_float3 kd, ks, emission, F0;
    float n, shininess;
    int32_t type;                                          /* 0 diffuse, 1 mirror, 2 dielectric, 3 emitter */
    int32_t _pad;
} pt_material;
typedef struct { pt_float3 P, D; } pt_ray;                 /* Ray: prog.cl:7-9 (32 B) */
typedef struct {                                           /* Triangle: prog.cl:18-21, main.cpp:139-182 (80 B) */
    pt_float3 r1, r2, r3, N;
    uint16_t mati;
    uint8_t _pad[14];
} pt_triangle;
typedef struct {                                           /* Camera: prog.cl:32-35, main.cpp:306-348 (80 B) */
    pt_float3 eye, lookat, up, right;
    float XM, YM;
    float _pad[2];
} pt_camera;

typedef struct pt_context pt_context;

/* ---- value-type constructors (host arithmetic of the reference's constructors) --- */
/* Material(kd,ks,emission,N,K,shininess,type): main.cpp:101-111 */
void pt_material_init(pt_material* m, const float kd[3], const float ks[3], const float emission[3],
                      const float N[3], const float K[3], float shininess, int32_t type);
/* Triangle(r1,r2,r3,mati): main.cpp:144-166 (precomputes the unit geometric normal) */
void pt_triangle_init(pt_triangle* t, const float r1[3], const float r2[3], const float r3[3], uint16_t mati);
/* n x Triangle(...): verts holds 9 floats per triangle (r1, r2, r3), mati one index each */
void pt_triangles_init(pt_triangle* out, const float* verts, const uint16_t* mati, int64_t n);
/* Camera(): main.cpp:311-347, with the globals it reads (global_fov/yaw/pitch/shift,
 * screen_width/height: main.cpp:20-21,30-39) passed as arguments */
void pt_camera_init(pt_camera* c, float fov, float yaw, float pitch, const float shift[3],
                    int32_t width, int32_t height);
/* The side effect of the reference's Camera() (main.cpp:334-336): shift += ahead * forward + right * rightward + up * upward
 * along the rotated unit axes, in place -- what moves the camera when the key handlers (main.cpp:1189-1209) set the three
 * globals.  Call it before pt_camera_init, once per Camera() the reference would have constructed. */
void pt_camera_move(float shift[3], float yaw, float pitch, float forward, float rightward, float upward);

/* ---- context: Scene::init_Scene, main.cpp:456-528 --------------------------------- */
/* Selects HIP device `device`, allocates rays (32 B/px), rnds (4 B/px), colors (16 B/px)
 * and seeds rnds from std::minstd_rand0 in pixel order (main.cpp:508-527). */
int pt_create(int device, int32_t width, int32_t height, pt_context** out);
/* Multi-GPU variant: this context owns the rows r with (r / rows_per_block) % world == rank
 * of the global width x height frame.  Seeds and pixel ids stay those of the
 * GLOBAL frame, so the union of all ranks' pixels equals a 1-GPU render bit for bit. */
int pt_create_tiled(int device, int32_t width, int32_t height, int32_t rank, int32_t world,
                    int32_t rows_per_block, pt_context** out);
void pt_destroy(pt_context* ctx);
/* Message of the last failure on ctx (ctx == NULL: last failure of pt_create*). */
const char* pt_last_error(const pt_context* ctx);
/* Scene::list_info, main.cpp:389-455: one line describing the device into buf. */
int pt_device_info(const pt_context* ctx, char* buf, int32_t buflen);

/* ---- scene authoring: main.cpp:529-617 --------------------------------------------- */
int pt_add_material(pt_context* ctx, const pt_material* m);          /* Scene::add_Material: returns the index (>= 0) */
int pt_add_triangle(pt_context* ctx, const pt_triangle* t);          /* Scene::add_Triangle */
int pt_add_triangles(pt_context* ctx, const pt_triangle* t, int64_t n); /* n x add_Triangle */
int pt_end_obj(pt_context* ctx);                                     /* Scene::end_Obj: closes one object */
/* Scene::add_Obj(file,pos,scale,pitch,yaw), main.cpp:552-617: OBJ+MTL import with the
 * reference's conventions (x negated, rotate_x(pitch), rotate_y(yaw), scale, translate;
 * first three vertices of each face; MTL keys Kd Ks Ke Ns + custom Kn Kk Tp; one object
 * per shape). */
int pt_add_obj(pt_context* ctx, const char* file, const float pos[3], const float scale[3],
               float pitch, float yaw);
int pt_upload_triangles(pt_context* ctx);                            /* Scene::upload_Triangles, main.cpp:618-630 */
int pt_upload_materials(pt_context* ctx);                            /* Scene::upload_Materials, main.cpp:631-634 */

/* ---- RNG state: main.cpp:522-527 ---------------------------------------------------- */
int pt_seed_default(pt_context* ctx);                                /* re-seed from minstd_rand0 */
/* seeds[] holds one int per pixel of the GLOBAL frame (n = width*height) */
int pt_upload_seeds(pt_context* ctx, const int32_t* seeds, int64_t n);

/* ---- the hot path: main.cpp:635-687 -> prog.cl:384-389, 292-381 --------------------- */
int pt_generate_rays(pt_context* ctx, const pt_camera* cam);         /* Scene::generate_rays -> gen_ray */
int pt_trace_rays(pt_context* ctx, const pt_camera* cam, int32_t iterations, int32_t current_sample); /* Scene::trace_rays -> trace_ray */
/* nsamples x Scene::render(): for current_sample = s0 .. s0+nsamples-1 (s0 = the context's
 * sample counter, main.cpp:28) do generate_rays + trace_rays; the counter advances by
 * nsamples.  Fused on the device: one launch, path state in registers. */
int pt_render(pt_context* ctx, const pt_camera* cam, int32_t iterations, int32_t nsamples);
int pt_set_current_sample(pt_context* ctx, int32_t current_sample);  /* main.cpp:1046 etc.: key events reset it to 0 */
int pt_get_current_sample(const pt_context* ctx, int32_t* out);
int pt_sync(pt_context* ctx);                                        /* queue.finish(), main.cpp:675 */

/* ---- readback (the reference never reads back; its only output is a GL texture) ----- */
int pt_local_pixel_count(const pt_context* ctx, int64_t* out);
int pt_local_pixel_ids(const pt_context* ctx, int32_t* out_ids, int64_t n);  /* global pixel id of each local pixel */
int pt_slab_pixel_count(const pt_context* ctx, int64_t* out);       /* max over ranks of the local pixel count */
int pt_read_colors(pt_context* ctx, float* out_rgba, int64_t npix);  /* buffer_colors: float3 @ 16 B stride */
int pt_read_rnds(pt_context* ctx, int32_t* out, int64_t npix);       /* buffer_rnds */
int pt_read_rays(pt_context* ctx, pt_ray* out, int64_t npix);        /* buffer_rays */
/* reinhard_tone_map + sRGB of colors (prog.cl:247-269, the value write_imagef stores at
 * prog.cl:380); which = 0 Reinhard, 1 = filt_im (3x3 median + filmic, prog.cl:391-427). */
int pt_resolve_ldr(pt_context* ctx, int32_t which, float* out_rgba, int64_t npix);

/* ---- multi-GPU frame assembly (SURVEY 8b "RCCL communicator per context", 8e) ----------
 * The reference is single-device (main.cpp:466-476); a host that tiles the frame over N contexts with
 * pt_create_tiled assembles it with these.  One process (or thread) per GPU:
 *   rank 0: pt_comm_unique_id(id); every rank receives the 128 bytes by the host's own means (MPI,
 *   a file, torch.distributed ...); every rank: pt_comm_init(ctx, id); after rendering, every rank:
 *   pt_gather_frame(ctx) -- ONE ncclAllGather of the ranks' radiance slabs over RCCL/xGMI on the context's
 *   stream + a de-interleave kernel -- leaves the whole width x height frame (float3 @ 16 B, global pixel
 *   order) in device memory on every rank.  world = 1 needs no communicator.
 * The gathered frame is served (pt_device_frame, pt_read_frame, pt_write_*) only until the next call that renders: after
 * that a one-rank context serves its colors buffer (which IS the frame) and a tiled context returns NULL / PT_EINVAL
 * until pt_gather_frame has run again -- never a frame older than colors. */
#define PT_COMM_ID_BYTES 128
/* PT_OK if librccl could be bound in this process (dlopen + the five entry points), PT_ECOMM otherwise (pt_last_error(NULL)
 * says why).  Touches no collective: every rank can ask BEFORE any of them enters ncclCommInitRank, so that a rank that cannot
 * take part is found while the others can still be told (a rank missing from ncclCommInitRank blocks all the others). */
int pt_comm_available(void);
int pt_comm_unique_id(void* id128);                                  /* ncclGetUniqueId */
int pt_comm_init(pt_context* ctx, const void* id128);                /* ncclCommInitRank(world, id, rank) of pt_create_tiled */
int pt_gather_frame(pt_context* ctx);
void* pt_device_frame(pt_context* ctx);                              /* the assembled frame (world = 1 without a fresh gather: the colors buffer); NULL if stale */
int pt_frame_size(const pt_context* ctx, int32_t* width, int32_t* height, int64_t* npix);
int pt_read_frame(pt_context* ctx, float* out_rgba, int64_t npix);   /* npix = width * height of the GLOBAL frame */

/* ---- image files: what the reference shows through its GL blit, main.cpp:1019-1039 -------
 * PFM = the HDR `colors` (the parity target) of the assembled frame; PPM = an LDR resolve (`which` as in
 * pt_resolve_ldr; the tone map's NaN for black pixels, prog.cl:265-267, is written as 0), world = 1 only.
 * The pt_image_* forms write a caller-supplied host buffer (float3 @ 16 B, row 0 = bottom of the view). */
int pt_write_pfm(pt_context* ctx, const char* path);
int pt_write_ppm(pt_context* ctx, const char* path, int32_t which);
int pt_image_write_pfm(const char* path, const float* rgba, int32_t width, int32_t height);
int pt_image_write_ppm(const char* path, const float* rgba, int32_t width, int32_t height);

/* ---- plumbing: device memory, streams, options, statistics --------------------------- */
/* Use caller-owned device buffers (e.g. torch tensors) for colors (16 B/px) and rnds (4 B/px)
 * of the LOCAL pixels; current contents are copied in.  NULL keeps the internal buffer.  In a tiled
 * context the colors buffer must hold pt_slab_pixel_count() pixels (the largest rank's count: the
 * all-gather sends equal slabs). */
int pt_bind_framebuffer(pt_context* ctx, void* d_colors, void* d_rnds);
void* pt_device_colors(pt_context* ctx);
void* pt_device_rnds(pt_context* ctx);
int pt_set_stream(pt_context* ctx, void* hip_stream);                /* hipStream_t; NULL = default stream */
/* options (key, value):
 *   "variant"      0 megakernel (default), 1 wavefront (stream-compacted, path state in HBM)
 *   "bvh_policy"   WHICH tree: 0 binned SAH with SAH leaf termination (default), 2 / 3 the same with every subtree of <= 4 /
 *                  <= 8 triangles forced into a leaf, 4 Morton order + PLOC merges (device only, a cheaper and worse tree),
 *                  5 = 0 built on the device whatever the scene size; set before the triangles are uploaded
 *   "bvh_device"   WHERE the SAH tree of policies 0..3 is built -- the result is the same, node for node: -1 (default) on the
 *                  device for scenes of >= 16,384 triangles, 0 on the host, 1 on the device.  The host builds whatever the
 *                  device hands back (non-finite triangles, ranges that need the median split)
 *   "sah_grain"    device SAH builder: ranges of at most this many triangles are finished by one wave each (default 256; 8..65536)
 *   "wide_on_device" device-built trees: 1 (default) the 4-wide collapse runs on the device too (the same nodes), 0 on the host
 *   "lds_scene"    2 (default) every workgroup stages BVH nodes in LDS: the whole tree when it fits (<= 64 KB,
 *                  <= 4096 triangles), otherwise its top if "treelet" asks for one (else the nodes are read through
 *                  L1/L2, see "wide_nodes"); 0 every node through L1/L2
 *   "treelet"      nodes of a LARGE tree to stage in LDS (the ones with the biggest boxes, renumbered to the front):
 *                  0 (default) none -- slower than 4-wide nodes through L1/L2 --, -1 what fits next to one
 *                  1,024-thread workgroup's stacks (~750-1,000), 2..2048; set before the triangles are uploaded
 *   "schedule"     megakernel: 1 a lane whose path ended starts its next sample at once and the wave leaves a traversal
 *                  when at most "suspend_lanes" lanes are unfinished (they resume in the next trip); 0 lockstep: all
 *                  lanes of a wave start a sample together; 2 like 1, and a lane whose pixel has had its samples of the
 *                  wave's work item moves on to its pixel of the wave's next item instead of waiting for the item's
 *                  slowest pixel (persistent launches; "migrate_lanes": how many such lanes must have gathered, default 1);
 *                  -1 (default) by tiles per resident wave: from 3 (one or two GPUs at 1080p) 2 -- for a tree in LDS only from
 *                  64 samples per launch, below that 1 --, else 0
 *   "suspend_lanes" -1 (default: 16 for a tree in LDS, else 24), 0..63
 *   "chunk_taper"  persistent launches with chained passes: > 0 the last "chunk_spp" samples of a launch are cut in halves down
 *                  to this many (64 samples in passes of 32, taper 8: 32, 16, 8, 8), so that the launch ends on short work items;
 *                  0 all passes "chunk_spp" long; -1 (default) 8 / 4 under schedule 2 from 64 / 32 samples per launch, else 0
 *   "lbvh_cluster" device-built trees (bvh_policy 4): the top of the tree above clusters of at most this many triangles is
 *                  rebuilt with the host's SAH over the cluster boxes (default 64; 0: the LBVH as the device built it);
 *                  set before the triangles are uploaded
 *   "build_threads" host SAH builder: 0 (default) as many threads as the machine has (at most 16), 1..256; the tree is
 *                  the same, node for node, for any number
 *   "wide_nodes"   trees read from global memory as 4-wide nodes with 8-bit child boxes (one 64-byte fetch decides two
 *                  BVH2 levels): 1 (default) when the tree does not fit LDS, 0 never, 2 every tree; set before the
 *                  triangles are uploaded
 *   "wide_lds_entries" 4-wide traversal: per-lane stack entries kept in LDS (default 20; the rest of the worst case lives in
 *                  global memory and is touched only by rays that get there); even, 4..20; set before the triangles are uploaded
 *   "waves_per_simd" kernels that read nodes from global memory: register budget for 4, 5, 6 or 7 resident waves per SIMD
 *                  (128 / 96 / 80 / 72 VGPRs); -1 (default) the most that the per-lane stacks in LDS leave room for
 *   "persistent"   1 (default) megakernel grid only fills the chip and every wave pulls its next 8x8
 *                  tile from a global counter; 0 one workgroup per group of tiles
 *   "chunk_spp"    persistent megakernel work items: n > 0 (pass, tile) items of n samples, chained per tile
 *                  through memory inside ONE launch; 0 whole tiles; -1 (default) by tiles per resident wave: 32 from 5
 *                  (64 when a launch has >= 256 samples), 16 from 3, 8 above 1.5, otherwise whole tiles
 *   "sah_visit_cost"  SAH price of one node visit in tenths of a triangle test (default 10; set before
 *                  the triangles are uploaded.  Measured: 5 / 10 / 15 / 20 -> 1006 / 1448 / 1393 / 1266 Msamples/s)
 *   "cost_binning" 0/1 wavefront: separate ray streams for rays touching a complex object's box
 *   "timing"       0/1 record HIP events around the dominant kernel ("kernel_ms" statistic)
 *   "count_work"   0/1 also count node visits / triangle tests (slower kernel instance)
 *   "reset_stats"  1 zero all statistics
 *   "debug_repeat" 0..1000 extra timed launches in pt_debug_closest_hit */
int pt_set_option(pt_context* ctx, const char* key, int64_t value);
/* stats: "segments" path segments executed since the last reset, "samples", "kernel_ms" (sum of
 * HIP-event durations of the dominant kernel), "kernel_launches", "bvh_nodes", "bvh_depth", "stack_entries"
 * (per-lane BVH2 traversal stack: deepest interior node + 2), "bvh_build_ms" (pt_upload_triangles as a whole),
 * "bvh_on_device", "triangles", "lds_bytes", "waves_per_simd", "node_mode" (0 whole tree in LDS, 1 BVH2 nodes through
 * L1/L2, 2 treelet, 3 4-wide nodes through L1/L2), "treelet_nodes", "wide_nodes" (how many 4-wide nodes),
 * "wide_pending" (their worst-case stack), "flat_triangles", "flat_boxes" (their distinct bounding boxes), and with
 * count_work: "node_visits", "tri_tests", "wave_node_steps", "wave_tri_steps", "tile_lane_steps", "wave_shade_steps",
 * "wave_trips", "wave_rounds" */
int pt_get_stat(pt_context* ctx, const char* key, double* out);

/* ---- introspection for tests (host data; no device work) ----------------------------- */
/* Packed BVH as uploaded: nodes (64 B each), triangle packets (48 B each), per-triangle
 * {rank, mati} pairs, and the add-order index of every packed triangle. */
int pt_debug_bvh_sizes(const pt_context* ctx, int64_t* nnodes, int64_t* ntris);
int pt_debug_bvh_copy(const pt_context* ctx, float* nodes, float* tris, int32_t* meta, int32_t* orig);
/* the 4-wide quantised nodes built from that tree (64 B each: {origin.xyz, exponents | nchild << 24}, {qlo_x, qhi_x,
 * qlo_y, qhi_y}, {qlo_z, qhi_z, -, -}, {4 child references}); *count = how many there are (0: not built), at most
 * `capacity` are copied */
int pt_debug_wide_nodes(const pt_context* ctx, void* out, int64_t capacity, int64_t* count);
/* Closest hit of n caller-supplied rays through the device traversal (kd_intersect, prog.cl:144-184):
 * out_t[i] = t (-1 on a miss), out_tri[i] = add-order index of the triangle hit (-1 on a miss). */
int pt_debug_closest_hit(pt_context* ctx, const pt_ray* rays, int64_t n, float* out_t, int32_t* out_tri);
/* The authored scene (what the reference keeps in Scene::tris / Scene::mats, main.cpp:366-371):
 * triangles in add order, materials, and the first triangle of every object. */
int pt_debug_scene_sizes(const pt_context* ctx, int64_t* ntris, int64_t* nmats, int64_t* nobjs);
int pt_debug_scene_copy(const pt_context* ctx, pt_triangle* tris, pt_material* mats, int32_t* obj_begin);
/* the reference's traversal encounter rank of each triangle, in add order */
int pt_debug_encounter_rank(const pt_context* ctx, int32_t* out, int64_t n);
/* after a counting launch (option count_work = 1, pt_render; schedules 0 and 1 -- under schedule 2 a wave works on two tiles at
 * once and leaves this zero): per 8x8 tile of the local frame, shader-clock cycles / 64 spent on it */
int pt_debug_tile_cost(pt_context* ctx, uint32_t* out, int64_t n_tiles);
/* What pt_render(nsamples) would launch on a device of cu_count compute units (0: the context's own; works on a host-only
 * context): out[8] = { threads per workgroup, waves per SIMD, schedule (0 lockstep, 1 suspend, 2 migrate), samples per (pass, tile)
 * work item (0: whole tiles), resident waves, tiles, node mode, dynamic LDS bytes }.  Lets CPU tests pin the launch policy
 * of a rank of an N-GPU job. */
int pt_debug_launch_plan(pt_context* ctx, int32_t nsamples, int32_t cu_count, int64_t out[8]);
/* frame assembly: out[gid] = index into the rank-major all-gather buffer (slab_stride pixels per rank) that
 * global pixel gid is read from -- the host statement of the de-interleave kernel's map (no device work) */
int pt_debug_gather_index(int32_t width, int32_t height, int32_t world, int32_t rows_per_block, int64_t slab_stride, int64_t* out);
/* runs ONLY the de-interleave kernel of pt_gather_frame on a caller-supplied all-gather buffer
 * (world x slab pixels, float4 each) with the context's frame size and tiling */
int pt_debug_deinterleave(pt_context* ctx, const float* gathered_rgba, int64_t n_pixels, float* out_frame_rgba);

#ifdef __cplusplus
}
#endif
#endif
