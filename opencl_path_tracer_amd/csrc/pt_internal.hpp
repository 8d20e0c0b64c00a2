// pt_internal.hpp -- shared between the host side (pt_host.cpp) and the device side
// (pt_kernels.hip) of libptamd.so.  Not part of the public ABI (that is include/pt_api.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "pt_api.h"

namespace ptamd {

// ---- packed scene layout in HBM ----------------------------------
// BVH2 node, 64 B = 4 x float4.  A node stores the boxes of BOTH children so one fetch
// decides both descents:
//   q[a] (a = 0,1,2 = x,y,z) = { Lmin[a], Lmax[a], Rmin[a], Rmax[a] }
//   q[3] = { bits(left), bits(right), 0, 0 }
// child reference >= 0: interior node index; < 0: leaf, ~ref = (first << 3) | (count - 1).
struct Node64 {
    float q[3][4];
    int32_t left, right;
    int32_t pad[2];
};
static_assert(sizeof(Node64) == 64, "Node64 must be 64 B");

// Triangle packet, 48 B = 3 x float4: { r1.xyz r2.x | r2.yz r3.xy | r3.z N.xyz } -- exactly the
// twelve floats prog.cl:94-112 reads, nothing derived (the hit test is bit-defined by them).
struct TriPacket {
    float v[12];
};
static_assert(sizeof(TriPacket) == 48, "TriPacket must be 48 B");

// per packed triangle: encounter rank in the reference's traversal (tie-break), material index
struct TriMeta {
    int32_t rank;
    int32_t mati;
};

constexpr int kMaxLeaf = 4;        // triangles per leaf (<= 8 by the reference encoding)
constexpr int kMaxDepth = 30;      // builder guarantees depth <= kMaxDepth
constexpr int kStatRows = 256;     // statistics counters are spread over this many rows of 8
constexpr int kStackEntries = 36;  // upper bound of the per-lane traversal stack: sentinel + far children + one slot above the top

// ---- kernel parameter block (passed by value, like `Camera` in prog.cl:292-304) ----------
struct RenderParams {
    const float4* nodes;
    const float4* tris;
    const TriMeta* meta;
    const pt_material* mats;
    int32_t* rnds;          // local pixels
    float4* colors;         // local pixels, float3 @ 16 B
    pt_ray* rays;           // local pixels
    unsigned long long* stats;   // [kStatRows][8]: 0 segments, 1 samples, 2 node visits, 3 tri tests, 4/5 wave-level body runs
    pt_camera cam;
    int32_t width, height;       // GLOBAL frame
    int32_t local_rows;          // rows owned by this context
    int32_t rank, world, rows_per_block;
    int32_t iterations, first_sample, nsamples;
    int32_t n_nodes, n_tris;
    int32_t lds_scene;           // 1: stage nodes+triangles in LDS
    int32_t stack_entries;       // per-lane stack depth actually needed (BVH depth + 1)
    uint32_t* tile_counter;      // != 0: persistent launch, waves pull tile indices from this counter
    int32_t n_tiles;
    int32_t chunk_spp;           // > 0: a work item is (pass, tile) = chunk_spp samples of a tile; passes of one
                                 // tile are chained through tile_done[] (agent-scope release / acquire)
    uint32_t* tile_done;         // [n_tiles] number of passes completed, zeroed before the launch
    int32_t pixel_map;           // 0: one wave = one 8x8 tile; 1: lane l of wave w owns pixel l*n_waves + w
};

// ---- wavefront (stream-compacted) formulation
// Per local pixel (index li), 64 B:  sC = {fL.xyz, fB.x}  sD = {fB.yz, fS.xy}  sE = {fS.z, fR.xyz}
//                                    sF = {color.xyz, bits(seed | inside << 31)}
// Ray streams, one per (bounce parity, cost class), addressed by POSITION (compact, written and
// read coalesced), 32 B per ray:     rsA = {P.xyz, D.x}  rsB = {D.y, D.z, bits(li), 0}
// Hit stream, parallel to the ray stream of the current bounce, 8 B:  hit = {t, bits(tri)}
// Class queues (shade input): entries (cost << 31) | position.
// Counter rows (kWfCounterStride words each), see wf_row(): 0 n_ray cheap, 1 n_ray expensive,
// 2..4 n_class A/B/C.  Row 0 belongs to bounce 0 (filled by wf_generate, cleared by a memset in
// front of it); bounce b >= 1 uses row b + 1 (cleared by wf_generate).
constexpr int kWfCounterStride = 8;
constexpr int kWfMaxBounces = 1023;
constexpr int kWfGenRow = 0;
__host__ __device__ inline int wf_row(int bounce) { return bounce == 0 ? kWfGenRow : bounce + 1; }
constexpr int kWfMaxCostBoxes = 8;
struct WfParams {
    RenderParams rp;
    float4 *sC, *sD, *sE, *sF;
    float4* rsA[2][2];     // [bounce parity][cost class]
    float4* rsB[2][2];
    float2* hit[2];        // [cost class]
    int32_t* q_cls[3];     // shade input queues: 0 diffuse/emitter, 1 mirror/dielectric/other, 2 miss
    uint32_t* counters;    // [(iterations + 3) * kWfCounterStride]
    float cbox[kWfMaxCostBoxes][6];   // bounding boxes of the complex objects (min xyz, max xyz)
    int32_t n_cbox;
    int32_t npix;
    int32_t sample;        // current_sample of this pass
};

// device-side BVH construction (pt_lbvh.hip); all pointers are device memory owned by the caller
struct LbvhResult {
    float4* d_nodes = nullptr;
    int n_nodes = 0;
    float4* d_tris = nullptr;
    TriMeta* d_meta = nullptr;
    int32_t* d_orig = nullptr;
    int depth = 0;
};
hipError_t lbvh_build(const pt_triangle* h_tris, const int32_t* h_rank, int n, hipStream_t stream, LbvhResult* out);

struct LaunchConfig {
    int block = 256;
    size_t lds_bytes = 0;
    int min_waves = 1;         // __launch_bounds__ second argument (waves per SIMD the allocator must allow)
    int traversal = 0;         // 0 while-while rounds, 1 wave-voting single steps
    int persistent_blocks = 1 << 30;   // grid size of a persistent launch (workgroups that fit the chip)
    bool count_work = false;   // also count node visits / triangle tests into stats[2], stats[3]
};

// launchers implemented in pt_kernels.hip; all asynchronous on `stream`
hipError_t launch_gen_ray(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream);
hipError_t launch_trace_ray(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream);
hipError_t launch_render_mega(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream);
hipError_t launch_resolve_reinhard(const float4* colors, float4* out, int64_t n, hipStream_t stream);
hipError_t launch_filt_im(const float4* colors, float4* out, int32_t width, int32_t height, hipStream_t stream);
hipError_t launch_wf_generate(const WfParams& p, hipStream_t stream);
hipError_t launch_wf_intersect(const WfParams& p, int bounce, hipStream_t stream);
hipError_t launch_wf_shade(const WfParams& p, int bounce, hipStream_t stream);
hipError_t launch_debug_closest_hit(const RenderParams& p, const pt_ray* rays, int64_t n, float* out_t, int32_t* out_tri, hipStream_t stream, size_t lds_pad = 0);
size_t mega_lds_bytes(const RenderParams& p, int block);
int mega_max_lds_scene_bytes();

}  // namespace ptamd
