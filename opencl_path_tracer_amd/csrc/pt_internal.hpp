// pt_internal.hpp -- shared between the host side (pt_host.cpp, pt_builder.cpp, pt_launch.cpp: see pt_context.hpp) and the device side
// (pt_kernels.hip) of libptamd.so.  Not part of the public ABI (that is include/pt_api.h).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

// 4-wide node, 64 B = 4 x float4 (pt_wide.cpp builds it from the BVH2; Trav<kNodesWide>::wide_step reads it):
//   { origin.xyz, exp_x | exp_y << 8 | exp_z << 16 | nchild << 24 }      grid of the node: plane = origin + q * 2^(exp - 127)
//   { qlo_x, qhi_x, qlo_y, qhi_y }   { qlo_z, qhi_z, -, - }              byte k of each word = child k's plane on that grid
//   { ref[0..3] }                    child references as in Node64; kWideNoChild (and an inverted box) where there is no child
struct Node4q {
    float origin[3];
    uint8_t exp[3];
    uint8_t nchild;
    uint32_t qlo_x, qhi_x, qlo_y, qhi_y;
    uint32_t qlo_z, qhi_z;
    uint32_t spare[2];
    int32_t ref[4];
};
static_assert(sizeof(Node4q) == 64, "Node4q must be 64 B");
constexpr int32_t kWideNoChild = ~0;      // the leaf {packet 0, 1 triangle}: always present (an empty scene holds one all-zero packet)
constexpr int kWideLdsEntries = 20;   // 20 x 4 B x 256 lanes = 20 KB per workgroup: seven workgroups per CU next to the big-triangle list
bool build_wide_nodes(const std::vector<Node64>& bvh2, std::vector<Node4q>* out, int* max_pending, int threads);   // pt_wide.cpp

// Where the traversal reads BVH nodes from (Trav<MODE> in pt_device.hpp)
enum : int { kNodesLds = 0, kNodesGlobal = 1, kNodesTreelet = 2, kNodesWide = 3 };

// Packets and nodes are addressed with 32-bit byte offsets on the device (index * 48, index << 6) and a
// leaf reference holds first << 3 in 31 bits: 2^26 triangles (hence < 2^26 nodes) keep all three in range.
constexpr int64_t kMaxTriangles = (int64_t)1 << 26;

constexpr int64_t kDeviceBuildFrom = 16384;   // SAH trees of scenes this big are built on the device by default (same tree; mesh6k: 2.4 vs 2.8 ms, 1M: 15 vs 99)
constexpr int kMaxLeaf = 4;        // triangles per leaf (<= 8 by the reference encoding)
constexpr int kMaxDepth = 30;      // builder guarantees depth <= kMaxDepth
constexpr int kTileCounterError = 16;  // word of RenderParams::tile_counter that holds 1 + tile of a lost hand-over (word 17: the pass) -- on a cache
                                       // line of its own: the waves that poll for a hand-over look at it now and then, and the line of words 0 / 1
                                       // takes every work-item fetch of the launch
constexpr int kMaxTaperPasses = 30;   // passes of a launch whose lengths are tabled (RenderParams::taper_end)
constexpr int kTileCounterWords = 32;  // words of the work counter (two 64-byte lines)
constexpr int kStatRows = 256;     // statistics counters are spread over this many rows of kStatCols
constexpr int kStatCols = 16;
// k_render instances whose register budget is set for this many waves per SIMD or more carry nothing across a traversal
// that can be recomputed or fetched (LEAN, pt_kernels.hip) -- A/B switch: 8 = never.  7: at 80 VGPRs (six waves) the two
// forms measure the same (2,347 / 2,354 on the Cornell box, 787 / 794 on MESH-100k, profiles/r03/r_*), and the plain one
// leaves the frame buffer alone until the end of a pass.
// Round 4: 6.  Under schedule 2 the plain 80-VGPR instance spills four registers of path state around every traversal -- stores in
// the hot loop, whose dirty lines every agent-scope release of a chained pass writes back: 17.3 GB of WRITE_SIZE per 64-spp launch
// of the Cornell box, against 4.0 GB for the lean form (whose frame-buffer lines are what is flushed), at +1 % (profiles/r04/u_*).
#ifndef PT_LEAN_FROM_WPS
#define PT_LEAN_FROM_WPS 6
#endif
constexpr int kLeanFromWps = PT_LEAN_FROM_WPS;
// The k_render instances for a tree staged whole in LDS (two workgroups per CU either way): 2 x 768 threads at an 80-VGPR
// budget = six waves per SIMD when the LDS has room for it (12 waves per workgroup = 3 per SIMD: any placement of the two
// workgroups fits; 2 x 640 at 96 VGPRs does not -- 10 waves land 3+3+2+2 and the second workgroup finds two SIMDs full,
// which is what made "five waves" lose 30 % in round 2), otherwise 2 x 512 at 128 VGPRs.  Cornell box: 2,050 -> 2,349
// Msamples/s (profiles/r03/f_*).
constexpr int kLdsBlockWide = 768, kLdsWpsWide = 6;
constexpr int kLdsBlockBase = 512, kLdsWpsBase = 4;
// a node of a tree staged whole in LDS (kNodesLds): three swizzled box quads + both 16-bit child references in one word,
// 52 B padded to 56.  Not 64: a 14-dword stride puts the same field of 32 different nodes on 32 different bank pairs
// (a 16-dword stride on 4), and 7.5 KB less per workgroup is what lets two 768-thread workgroups share a CU
#ifndef PT_LDS_NODE_BYTES
#define PT_LDS_NODE_BYTES 56
#endif
constexpr int kLdsNodeBytes = PT_LDS_NODE_BYTES;
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
    unsigned long long* stats;   // [kStatRows][kStatCols]: 0 segments, 1 samples, 2 node visits, 3 tri tests, 4/5 wave-level body runs,
                                 // 6 tile lane steps, 7 wave-level shade runs, 8 wave-level trips of the segment loop, 9 wave-level rounds
    pt_camera cam;
    int32_t width, height;       // GLOBAL frame
    int32_t local_rows;          // rows owned by this context
    int32_t rank, world, rows_per_block;
    int32_t iterations, first_sample, nsamples;
    int32_t n_nodes, n_tris;
    int32_t n_flat;              // packets [0, n_flat) are the big-triangle list: tested by every ray before the tree
    int32_t n_fbox;              // their distinct boxes (the two triangles of a wall share one): box b is that of packet fbox_rep[b]
    uint8_t fbox_rep[32];        // and covers the listed triangles in fbox_mask[b]
    uint32_t fbox_mask[32];
    int32_t node_mode;           // kNodesLds / kNodesGlobal / kNodesTreelet / kNodesWide: where the traversal reads BVH nodes from (and which)
    int32_t treelet_nodes;       // kNodesTreelet: nodes [0, treelet_nodes) are staged in LDS
    int32_t stack_entries;       // per-lane stack entries in LDS: all a BVH2 traversal can need (sentinel + deepest interior node + the slot
                                 // above the top); kNodesWide: at most kWideLdsEntries, the rest of the worst case in stack_ovf
    uint32_t* stack_ovf;         // kNodesWide: [entries past the LDS part][lane of the grid], or null when LDS holds the worst case
    int32_t stack_ovf_lanes;     // lanes stack_ovf has room for (every launch's grid must fit)
    uint32_t* tile_counter;      // != 0: persistent launch, waves pull tile indices from word 0; word 1 counts the waves that left (the last resets
                                 // both); words 16 / 17: 1 + tile and pass of a hand-over that never came (kTileCounterError), left for the host
    uint32_t poll_ticks;         // chained passes: how long a wave waits for a tile's previous pass, in 10-ns ticks of s_memrealtime, before it
                                 // reports the tile in word 4 and the launch winds down (pt_sync then returns PT_EHIP)
    int32_t debug_stall_tile;    // tests: pass 0 of this tile is rendered but never published (-1: none)
    int32_t n_tiles;
    // passes of DIFFERENT lengths (option chunk_taper): n_taper > 0 -> pass k of a tile is samples [taper_end[k-1], taper_end[k]) of the
    // launch (taper_end[-1] = 0), n_taper passes in all; the last passes are short, so that the launch does not end on whole long items
    int32_t n_taper;
    uint16_t taper_end[kMaxTaperPasses];
    int32_t chunk_spp;           // > 0: a work item is (pass, tile) = chunk_spp samples of a tile; passes of one
                                 // tile are chained through tile_done[] (agent-scope release / acquire)
    uint32_t* tile_done;         // [n_tiles] number of passes completed, zeroed before the launch
    int32_t suspend_lanes;       // megakernel: leave the traversal when at most this many lanes are unfinished (0: never)
    int32_t node_min_lanes;      // Trav::round: the node phase ends early when at most this many lanes still descend and another holds a leaf (0: never)
    int32_t leaf_min_lanes;      // ... and the leaf phase when at most this many lanes still hold leaves and another has a node (0: never)
    int32_t migrate_lanes;       // kSchedMigrate: lanes that are done with the wave's current work item move to the next one when this many have gathered
    uint32_t* tile_cost;         // counting instances only, or null: [n_tiles] += shader-clock cycles / 64 the wave spent on each work item of the tile (pt_debug_tile_cost)
};

// ---- wavefront (stream-compacted) formulation
// Ray streams, one per (bounce parity, cost class), addressed by POSITION (compact, written and read coalesced), 48 B per
// ray:   rsA = {P.xyz, D.x}   rsB = {D.y, D.z, bits(li), bits(flags)}   rsC = {factor_L.xyz, bits(LCG state)}
// What changes at (almost) every segment rides the stream: the ray, factor_L (every diffuse hit multiplies it) and the LCG
// state (every diffuse / emitter / glass hit draws) -- round 3 kept the last two in per-PIXEL arrays, 12-B and 4-B accesses
// scattered by the compaction order, which cost a 64-byte sector each way each.  wf_intersect reads rsA and half of rsB only.
// Per local pixel (index li): the other three factors and the colour, arrays of 12-B records sP[field][li] (kWfB .. kWfC; the
// kWfL slot is unused).  A field is read only if it was written in this sample and written only by the material that changes it
// (PathInHbm, pt_wavefront.hip): mirrors factor_S, glass factor_R, emitters the colour; factor_B only where a material has a
// specular lobe -- with ks = 0 it becomes +0 at the first diffuse hit and stays there, which is one flag bit, not 12 bytes.
// rnds[li] is read by wf_generate and written when the path ends.
//                                    flags: bit f (f < 5) = field f of sP is valid; bit 5 = the path is inside glass;
//                                    bit 6 = factor_B is +0 in all three components (nothing stored)
// Hit stream, parallel to the ray stream of the current bounce, 8 B:  hit = {t, bits(tri)}
// Class queues (shade input): entries (cost << 31) | position.
// Counter rows (kWfCounterStride words each), see wf_row(): 0 n_ray cheap, 1 n_ray expensive,
// 2..4 n_class A/B/C, 5 / 6 next unassigned ray of the cheap / expensive stream (wf_intersect's work fetch).
// Row 0 belongs to bounce 0 (filled by wf_generate, cleared by a memset in
// front of it); bounce b >= 1 uses row b + 1 (cleared by wf_generate).
constexpr int kWfCounterStride = 8;
constexpr int kWfMaxBounces = 1023;
constexpr int kWfGenRow = 0;
__host__ __device__ inline int wf_row(int bounce) { return bounce == 0 ? kWfGenRow : bounce + 1; }
constexpr int kWfMaxCostBoxes = 8;
constexpr int kWfMaxChains = 8;
constexpr int kWfDefaultChains = 4;
enum : int { kWfL = 0, kWfB = 1, kWfS = 2, kWfR = 3, kWfC = 4, kWfFields = 5, kWfInsideBit = 1 << 5, kWfBZeroBit = 1 << 6 };
struct WfParams {
    RenderParams rp;
    float* sP;             // [kWfFields][npix] x 3 floats
    float4* rsA[2][2];     // [bounce parity][cost class]
    float4* rsB[2][2];
    float4* rsC[2][2];
    float2* hit[2];        // [cost class]
    int32_t* q_cls[3];     // shade input queues: 0 diffuse/emitter, 1 mirror/dielectric/other, 2 miss
    uint32_t* counters;    // [(iterations + 3) * kWfCounterStride]
    float cbox[kWfMaxCostBoxes][6];   // bounding boxes of the complex objects (min xyz, max xyz)
    int32_t n_cbox;
    int32_t npix;          // pixels of this CHAIN (the local pixels are cut into wf_streams contiguous chains, each with its own
                           // streams, queues and counters, run on its own HIP stream: one chain's launch tails under another's work)
    int32_t pix0;          // first local pixel of the chain
    int32_t npix_all;      // local pixels of the context (stride of sP's fields)
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
// all triangles of a scene in device memory, for the device builders (pt_sahdev.hip: stage_*)
struct DeviceStage {
    char* base = nullptr;            // the one allocation behind the pointers below
    pt_triangle* d_tris = nullptr;   // add order
    int32_t* d_rank = nullptr;       // encounter ranks, add order
    int32_t* d_sel = nullptr;        // add-order indices of the triangles that go into the tree (stage_select)
    float* d_area = nullptr;         // half area of every triangle's padded bounds
    int* d_misc = nullptr;
    int n = 0;
};
hipError_t stage_upload(const pt_triangle* h_tris, const int32_t* h_rank, int n, hipStream_t stream, DeviceStage* st, float* h_area, int* nonfinite);
hipError_t stage_rest_box(const DeviceStage& st, const int32_t* h_top, int cand, hipStream_t stream, float box[6]);
hipError_t stage_select(const DeviceStage& st, const int32_t* h_flat, int nf, hipStream_t stream);
void stage_free(DeviceStage* st);
hipError_t lbvh_build(const pt_triangle* d_tris, const int32_t* d_rank, int n_all, const int32_t* d_sel, int n, int ploc_radius, hipStream_t stream, LbvhResult* out);
// the host builder's binned-SAH tree, built on the device (pt_sahdev.hip); *unsupported: a range needs the host's median split
hipError_t sah_device_build(const pt_triangle* d_tris, const int32_t* d_rank, int n_all, const int32_t* d_sel, int n, int max_leaf, bool force_leaf, float visit_cost, int grain,
                            hipStream_t stream, LbvhResult* out, bool* unsupported);

// build_wide_nodes() on the device (pt_widedev.hip): the same 4-wide nodes for a BVH2 that is in device memory
hipError_t wide_device_build(const float4* d_bvh2, int n_nodes, hipStream_t stream, float4** d_out, int* n_out, int* max_pending, bool* failed);

struct LaunchConfig {
    int block = 256;                   // traversal_block(node_mode)
    size_t lds_bytes = 0;              // traversal_lds_bytes()
    int persistent_blocks = 1 << 30;   // grid size of a persistent launch (workgroups that fit the chip): what launch_cfg expects to be resident
    int cu_count = 0;                  // compute units of the device (> 0: the grid is also capped by what the runtime says can co-reside)
    bool count_work = false;           // also count node visits / triangle tests into stats[2..9]
    int schedule = 0;                  // megakernel: 0 lockstep per sample, 1 restart + tail suspension (pt_kernels.hip)
    int waves_per_simd = 4;            // register budget of the k_render instance: 4, or 5 / 6 / 7 for nodes from global memory
};

// Dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel.  Set ONCE per kernel instance
// and device (a high-water mark), not per launch: the reference's own loop shape launches a kernel per sample
// (main.cpp:683-687) and a per-launch runtime call shows there.  `mark` = one static array per template instantiation.
constexpr int kMaxDevicesPerProcess = 16;
struct LdsMark {
    std::atomic<size_t> bytes[kMaxDevicesPerProcess];
};
inline hipError_t ensure_dynamic_lds(const void* kern, LdsMark& mark, size_t lds_bytes) {
    if (lds_bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevicesPerProcess) return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (mark.bytes[dev].load(std::memory_order_acquire) >= lds_bytes) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    size_t seen = mark.bytes[dev].load(std::memory_order_relaxed);
    while (seen < lds_bytes && !mark.bytes[dev].compare_exchange_weak(seen, lds_bytes, std::memory_order_release)) {}
    return hipSuccess;
}

// Workgroups of `kern` (BLOCK threads, lds_bytes of dynamic LDS) that can be resident on one CU according to the runtime, asked
// once per kernel instance, device and LDS size.  A persistent grid is capped by it: launch_cfg's own figure is derived from the
// register budget the instance was compiled for, and a grid LARGER than what is co-resident would leave workgroups queued behind
// waves that poll for a hand-over (harmless for correctness -- a producer is always dequeued before its consumer -- but it is not
// the launch shape the schedules were tuned for).  0: the runtime gave no answer.
struct OccMark {
    std::atomic<long long> key[kMaxDevicesPerProcess];      // (lds_bytes << 8) | blocks per CU
};
inline int resident_blocks_per_cu(const void* kern, OccMark& mark, int block, size_t lds_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevicesPerProcess) return 0;
    const long long seen = mark.key[dev].load(std::memory_order_acquire);
    if (seen != 0 && (size_t)(seen >> 8) == lds_bytes) return (int)(seen & 0xff);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, block, lds_bytes) != hipSuccess || n <= 0) return 0;
    mark.key[dev].store(((long long)lds_bytes << 8) | (long long)std::min(n, 255), std::memory_order_release);
    return n;
}

// PTAMD_TRACE=1: phase times of the host-side scene path on stderr (pt_add_obj, pt_upload_triangles; tools/obj_load_time.py)
struct PhaseClock {
    const char* who;
    const bool on = std::getenv("PTAMD_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit PhaseClock(const char* w) : who(w) {}
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[%s] %-32s %8.1f ms\n", who, what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// launchers (pt_kernels.hip, pt_wavefront.hip, pt_debug.hip); all asynchronous on `stream`
int traversal_block(int node_mode, bool wide_lds_block);       // threads per workgroup of k_render
size_t traversal_lds_bytes(const RenderParams& p, int block);  // stacks + staged nodes
hipError_t launch_gen_ray(const RenderParams& p, hipStream_t stream);
hipError_t launch_trace_ray(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream);
hipError_t launch_render_mega(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream);
hipError_t launch_resolve_reinhard(const float4* colors, float4* out, int64_t n, hipStream_t stream);
hipError_t launch_filt_im(const float4* colors, float4* out, int32_t width, int32_t height, hipStream_t stream);
hipError_t launch_wf_generate(const WfParams& p, hipStream_t stream);
hipError_t launch_wf_intersect(const WfParams& p, int bounce, int cu_count, hipStream_t stream);
hipError_t launch_wf_shade(const WfParams& p, int bounce, hipStream_t stream);
hipError_t launch_debug_closest_hit(const RenderParams& p, const pt_ray* rays, int64_t n, float* out_t, int32_t* out_tri, int cu_count, hipStream_t stream);

}  // namespace ptamd
