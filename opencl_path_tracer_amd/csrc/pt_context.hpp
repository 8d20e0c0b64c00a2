// pt_context.hpp -- what the host-side translation units of libptamd.so share: the context behind the opaque pt_context* of
// include/pt_api.h, the error / HIP-call conventions, the host thread pool, and the functions that cross file boundaries.
//   pt_host.cpp     the C ABI: constructors, context life cycle, authoring, read-back, options, statistics, debug entry points
//   pt_builder.cpp  scene -> tree: encounter ranks of end_Obj, the binned-SAH builder (host) and its orchestration on the device,
//                   big-triangle list, packets, cost boxes; pt_end_obj, pt_upload_triangles
//   pt_launch.cpp   tree -> launches: node placement (LDS / treelet / 4-wide), stack sizing, kernel parameters, the launch policy
//                   (instance, schedule, pass length), the wavefront chains; pt_generate_rays, pt_trace_rays, pt_render, pt_sync
// Internal: not part of the ABI.  (`using namespace ptamd` below is deliberate: these files ARE the library.)
#pragma once

#include "pt_internal.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <queue>
#include <thread>

using namespace ptamd;

namespace ptamd {
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
};
}  // namespace ptamd

struct pt_context {
    int device = -1;
    bool has_device = false;
    int32_t W = 0, H = 0;
    int32_t rank = 0, world = 1, rows_per_block = 8;
    int32_t local_rows = 0;
    int64_t npix = 0;  // local pixels
    int64_t slab_pix = 0;  // max over ranks of the local pixel count: what every rank contributes to the all-gather

    // ---- frame assembly (pt_comm.hip)
    void* comm = nullptr;          // ncclComm_t
    float4* d_gathered = nullptr;  // world x slab_pix
    float4* d_frame = nullptr;     // W x H
    uint64_t render_epoch = 0;     // bumped by every call that writes colors
    uint64_t frame_epoch = ~0ull;  // render_epoch at the last pt_gather_frame: d_frame is served only while they are equal

    // ---- authoring state (Scene members, main.cpp:365-372)
    std::vector<pt_triangle> tris;  // add order
    std::vector<int32_t> obj_begin;
    int32_t tri_shift = 0;
    std::vector<pt_material> mats;
    std::vector<int32_t> enc_rank;  // per add-order triangle
    int32_t next_rank = 0;
    bool tris_uploaded = false, mats_uploaded = false;

    // ---- packed scene (host copies kept for the debug getters)
    std::vector<Node64> nodes;
    std::vector<Node4q> nodes4;   // the same tree collapsed to 4-wide quantised nodes (empty: not built), pt_wide.cpp
    int wide_pending = 0;         // most entries a wide traversal can have pushed when it visits an interior node
    std::vector<TriPacket> packets;
    bool host_packets_stale = false;   // device-built tree: packets / meta behind the big-triangle list live on the device only until a debug getter asks
    std::vector<TriMeta> meta;
    std::vector<int32_t> orig;
    int bvh_depth = 0;
    int interior_depth = 0;   // depth of the deepest interior node of the packed tree (root: 0): sizes the traversal stacks
    int n_flat = 0;             // packed triangles [0, n_flat): the big-triangle list tested before the tree
    int n_fbox = 0;             // its distinct bounding boxes: representative packet and the listed triangles each one covers
    uint8_t fbox_rep[32] = {};
    uint32_t fbox_mask[32] = {};

    // ---- device buffers
    float4* d_nodes = nullptr;
    float4* d_nodes4 = nullptr;
    uint32_t* d_stack_ovf = nullptr;   // kNodesWide: stack entries past the LDS part, [entry][lane of the grid]
    size_t stack_ovf_lanes = 0;
    float4* d_tris = nullptr;
    TriMeta* d_meta = nullptr;
    pt_material* d_mats = nullptr;
    int32_t* d_rnds = nullptr;
    float4* d_colors = nullptr;
    pt_ray* d_rays = nullptr;
    float4* d_ldr = nullptr;
    unsigned long long* d_stats = nullptr;
    // wavefront variant: path state + queues (allocated on first use)
    float4* d_wf_state = nullptr;   // per local pixel: 4 float4 worth of path factors + colour (5 x 12 B), 8 + 4 float4 of ray streams (rsA, rsB; rsC), 2 float2 of hits
    int32_t* d_wf_queues = nullptr; // 3 x npix int32 (class queues)
    std::vector<float> cost_boxes;  // 6 floats per complex object (wavefront cost classes)
    uint32_t* d_wf_counters = nullptr;   // kWfMaxChains x (kWfMaxBounces + 4) rows
    hipStream_t wf_stream[kWfMaxChains] = {};   // chains 1.. of the wavefront variant (chain 0 runs on `stream`)
    hipEvent_t wf_event[kWfMaxChains] = {};
    int poll_timeout_ms = 10000;         // chained passes: a wave gives a tile's previous pass this long before it reports the hand-over lost
    int debug_stall_tile = -1;           // tests: pass 0 of this tile is never published
    bool counters_suspect = false;
    bool launched_since_check = false;   // a persistent launch has been enqueued since the work counter's error word was last read       // a launch failed or lost a hand-over: word 0 / 1 of d_tile_counter may not be back at zero
    int wf_streams = -1;                 // option wf_streams: chains of the wavefront variant (-1: kWfDefaultChains)
    bool own_rnds = true, own_colors = true;
    hipStream_t stream = nullptr;

    int32_t current_sample = 0;  // main.cpp:28

    // ---- options
    int variant = 0;
    int lds_scene = 2;   // 2: stage BVH nodes in LDS -- the whole tree when it fits next to two 512-thread blocks per
                         // CU, otherwise its top (`treelet`); 0: every node through L1/L2
    int treelet = 0;     // nodes of a large tree to stage in LDS: 0 none (default: with the big-triangle list in place the
                         // treelet no longer pays, profiles/r02/s_*), -1 what fits one 1,024-thread block per CU, n
    int treelet_nodes = 0;   // decided at upload: nodes [0, treelet_nodes) are the re-indexed top of the tree
    int timing = 0;
    int count_work = 0;
    int bvh_on_device = 0;
    double bvh_build_ms = 0.0;
    int cu_count = 256;
    int persistent = 1;   // 1: megakernel waves pull tiles from a counter (grid = what fits the chip)
    uint32_t* d_tile_counter = nullptr;
    uint32_t* d_tile_done = nullptr;
    uint32_t* d_tile_cost = nullptr;   // count_work: per tile, cycles / 64 its waves spent on it in the last launch (pt_debug_tile_cost)
    int chunk_taper = -1;  // option chunk_taper: shortest pass of a launch whose last passes taper off (0: all passes chunk_spp long; -1 default)
    int chunk_spp = -1;   // persistent megakernel work items: > 0 (pass, tile) items of that many samples, 0 whole
                          // tiles, -1 automatic (4 when the context has clearly more tiles than resident waves)
    int sah_visit_cost = 10;   // tenths of a triangle test (option sah_visit_cost)
    int flat_list = 16;        // at most this many big triangles go to the flat list (option flat_list; 0: none)
    int schedule = -1;     // megakernel: 0 lockstep per sample, 1 restart + tail suspension, -1 by the number of tiles per resident wave
    int suspend_lanes = -1; // tail suspension threshold of schedule 1 (-1: 24)
    int migrate_lanes = -1;                         // option migrate_lanes (kSchedMigrate; -1: 1)
    int node_min_lanes = -1, leaf_min_lanes = -1;   // phase switching of the while-while rounds (-1: by node path, fill_params)
    int lbvh_ploc = 16;     // device-built trees: PLOC search radius (8 / 16 / 32); 0: Karras' radix tree over the Morton codes
    int lbvh_cluster = 64;  // device-built trees: the top above clusters of this many triangles is rebuilt with the host SAH (0: not)
    int build_threads = 0; // host SAH builder: threads (0: the machine's, at most 16); the tree is the same for any number
    int wide_nodes = 1;    // 4-wide quantised nodes: 0 never, 1 for trees that do not fit LDS, 2 for every tree (tests)
    int wide_lds_entries = kWideLdsEntries;   // 4-wide traversal: stack entries per lane kept in LDS (tests lower it to force the global part)
    int lds_block = -1;      // whole tree in LDS: threads per workgroup of k_render: -1 768 where two such workgroups fit a CU, 512 (tests)
    int waves_per_simd = -1; // nodes from global memory: register budget for 4 / 5 / 6 / 7 waves per SIMD (-1: the most the LDS stacks allow)
    int debug_repeat = 0; // pt_debug_closest_hit: extra timed launches
    int cost_binning = 1; // wavefront: separate ray queues for rays that touch a complex object's box
    int bvh_policy = 0;   // 0/1 host SAH with SAH leaf termination, 2 leaves of <= 4, 3 leaves of <= 8, 4 device LBVH, 5 the SAH tree built on the device
    int sah_grain = 256;  // device SAH builder: ranges of at most this many triangles are finished by one wave each
    int wide_on_device = 1;  // device-built trees: the 4-wide collapse runs on the device too (0: on the host)
    int bvh_device = -1;     // SAH policies 0..3: build on the device (the same tree)?  -1: scenes of >= kDeviceBuildFrom triangles, 0 never, 1 always

    // ---- statistics
    std::vector<EventPair> events;
    size_t events_used = 0;
    double kernel_ms_acc = 0.0;
    int64_t kernel_launches = 0;
    size_t last_lds_bytes = 0;
    int last_waves_per_simd = 4;

    std::string err;
    char info[256] = {0};
};

namespace ptamd {

constexpr size_t kLdsPerCu = 160 * 1024;
// next to the stacks and the staged nodes: the big-triangle list (96 B each) and, in wf_intersect, one class byte per ray
// of a trip (16 waves x 256) + the compaction counters
constexpr size_t kLdsSlack = 32 * 100 + 4096 + 1024 + 256;

int fail(pt_context* ctx, int code, const std::string& msg);      // pt_host.cpp: records the text behind pt_last_error
int host_threads(const pt_context* ctx);                          // threads of the host-side scene path (option build_threads)

#define PT_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ctx, PT_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

#define PT_NEED_DEVICE(ctx)                                                                 \
    do {                                                                                    \
        if (!(ctx)) return PT_EINVAL;                                                       \
        if (!(ctx)->has_device)                                                             \
            return fail(ctx, PT_ENODEVICE, "context was created without a HIP device (host-only); no CPU render path exists"); \
    } while (0)
// A small persistent pool for the host-side scene path: the threaded builders issue hundreds of short parallel regions
// (a 1M-triangle SAH build: ~150 at the top of the tree), and spawning 15 threads for each cost more than the regions did.
// One region at a time; a second caller (another context on another host thread) simply runs its region on fresh threads.
class HostPool {
public:
    static HostPool& get() { static HostPool p; return p; }
    // fn(k) for k in [0, chunks), on up to `threads` threads including the caller
    template <class F>
    void run(size_t chunks, int threads, F fn) {
        if (chunks == 0) return;
        if (threads <= 1 || chunks == 1) { for (size_t k = 0; k < chunks; ++k) fn(k); return; }
        std::unique_lock<std::mutex> region(region_mu_, std::try_to_lock);
        if (!region.owns_lock()) {                       // pool busy: plain threads
            std::atomic<size_t> next(0);
            auto work = [&]() { for (size_t k = next.fetch_add(1); k < chunks; k = next.fetch_add(1)) fn(k); };
            std::vector<std::thread> th;
            for (int t = 1; t < std::min<int>(threads, (int)chunks); ++t) th.emplace_back(work);
            work();
            for (std::thread& t : th) t.join();
            return;
        }
        grow(std::min<int>(threads, (int)chunks) - 1);
        std::function<void(size_t)> f = fn;
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &f;
            chunks_ = chunks;
            next_.store(0);
            helpers_wanted_ = std::min<int>(threads, (int)chunks) - 1;
            helpers_in_ = 0;
            helpers_done_ = 0;
            ++generation_;
        }
        cv_.notify_all();
        for (size_t k = next_.fetch_add(1); k < chunks; k = next_.fetch_add(1)) fn(k);
        std::unique_lock<std::mutex> lk(mu_);
        job_ = nullptr;                                  // no helper may start on this job any more
        done_cv_.wait(lk, [&]() { return helpers_done_ == helpers_in_; });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
    }

private:
    void grow(int n) {
        while ((int)workers_.size() < n) workers_.emplace_back([this]() { loop(); });
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            std::function<void(size_t)>* job = nullptr;
            size_t chunks = 0;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&]() { return stop_ || (generation_ != seen && job_ != nullptr && helpers_in_ < helpers_wanted_); });
                if (stop_) return;
                seen = generation_;
                job = job_;
                chunks = chunks_;
                ++helpers_in_;
            }
            for (size_t k = next_.fetch_add(1); k < chunks; k = next_.fetch_add(1)) (*job)(k);
            {
                std::lock_guard<std::mutex> lk(mu_);
                ++helpers_done_;
            }
            done_cv_.notify_all();
        }
    }
    std::mutex region_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    std::function<void(size_t)>* job_ = nullptr;
    size_t chunks_ = 0;
    std::atomic<size_t> next_{0};
    int helpers_wanted_ = 0, helpers_in_ = 0, helpers_done_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false;
};

// fn(begin, end) over [0, n) on up to `threads` threads (element-wise work: any split gives the same result)
template <class F>
void parallel_for(size_t n, size_t grain, int threads, F fn) {
    const size_t nt = std::min<size_t>((size_t)std::max(threads, 1), (n + grain - 1) / std::max<size_t>(grain, 1));
    if (nt <= 1) { fn((size_t)0, n); return; }
    const size_t per = (n + nt - 1) / nt;
    HostPool::get().run(nt, (int)nt, [&](size_t k) {
        const size_t b = k * per, e = std::min(n, b + per);
        if (b < e) fn(b, e);
    });
}
// ---- pt_builder.cpp
int deepest_interior_node(const std::vector<Node64>& nodes);
int reindex_treelet(std::vector<Node64>& nodes, int interior_depth, int want);

// ---- pt_launch.cpp
int32_t count_local_rows(int32_t H, int32_t rank, int32_t world, int32_t rb);
inline int32_t global_row(const pt_context* c, int32_t lrow) {
    return ((lrow / c->rows_per_block) * c->world + c->rank) * c->rows_per_block + (lrow % c->rows_per_block);
}
int wide_stack_entries(int pending);
int stack_entries_for(int interior_depth);
bool whole_tree_fits_lds(size_t n_nodes, size_t n_tris, int interior_depth, int n_flat);
int plan_node_placement(pt_context* ctx, const float4* d_bvh2 = nullptr, bool* wide_on_device = nullptr);
int alloc_stack_overflow(pt_context* ctx);
int seed_upload(pt_context* ctx, const int32_t* global_seeds);
void fill_params(const pt_context* ctx, const pt_camera* cam, RenderParams* p);
int check_ready(pt_context* ctx, const pt_camera* cam);
int time_begin(pt_context* ctx, EventPair** ep);
int time_end(pt_context* ctx, EventPair* ep);
int time_collect(pt_context* ctx);
int sync_and_check(pt_context* ctx);
template <class T>
int upload_vec(pt_context* ctx, T** dptr, const void* src, size_t bytes) {
    if (*dptr) { PT_HIP(ctx, hipFree(*dptr)); *dptr = nullptr; }
    PT_HIP(ctx, hipMalloc((void**)dptr, std::max<size_t>(bytes, 64)));
    // an (almost) empty array still has one readable, all-zero record: a zero packet can never be
    // hit, so a leaf reference into an empty scene (the wrapped root's ~0) stays harmless
    if (bytes < 64) PT_HIP(ctx, hipMemset(*dptr, 0, 64));
    if (bytes) PT_HIP(ctx, hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
    return PT_OK;
}

// kNodesWide: room for the stack entries past the LDS part, for every lane of the largest grid a traversal kernel of
// this context is launched with (256-thread workgroups: persistent <= 6 per CU, wf_intersect 2 x 8 per CU, the debug
// kernel 8 per CU, a non-persistent render one wave per tile)

}  // namespace ptamd
