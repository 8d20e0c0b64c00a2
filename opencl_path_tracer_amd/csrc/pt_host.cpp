// pt_host.cpp -- host side of libptamd.so: the C ABI of include/pt_api.h.
//
// What lives here (all of it host work the reference also does on the host):
//   * value-type constructors            main.cpp:101-111, 144-166, 311-347
//   * Scene authoring + end_Obj           main.cpp:529-551   (encounter order of the
//     reference's per-object tree, needed only to break exact-t ties like prog.cl:113-184)
//   * own BVH: binned-SAH BVH2 over ALL objects, two child boxes per 64-B node, 48-B triangle
//     packets -- replaces NodeOnHost::convert's heap array, main.cpp:263-303
//   * buffer management + launches        main.cpp:456-528, 618-687
// There is no CPU render path in this library: every pt_render/pt_trace_rays/pt_generate_rays
// call launches HIP kernels or fails.
//
// Compiled with -ffp-contract=off: the reference's host arithmetic is plain x86-64 g++
// (no fused multiply-add), and the results of these constructors feed bit-exact parity tests.
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

namespace {

thread_local std::string g_create_error;

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
};

}  // namespace

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
    int poll_timeout_ms = 2000;          // chained passes: a wave gives a tile's previous pass this long before it reports the hand-over lost
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
    int chunk_spp = -1;   // persistent megakernel work items: > 0 (pass, tile) items of that many samples, 0 whole
                          // tiles, -1 automatic (4 when the context has clearly more tiles than resident waves)
    int sah_visit_cost = 10;   // tenths of a triangle test (option sah_visit_cost)
    int flat_list = 16;        // at most this many big triangles go to the flat list (option flat_list; 0: none)
    int schedule = -1;     // megakernel: 0 lockstep per sample, 1 restart + tail suspension, -1 by the number of tiles per resident wave
    int suspend_lanes = -1; // tail suspension threshold of schedule 1 (-1: 24)
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

namespace {

int fail(pt_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

// threads of the host-side scene path (option build_threads; 0: the machine's, at most 16)
int host_threads(const pt_context* ctx) {
    const unsigned hw = std::thread::hardware_concurrency();
    return ctx->build_threads > 0 ? ctx->build_threads : (int)std::min(16u, std::max(1u, hw));
}

}  // namespace

namespace ptamd {
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
}

namespace {

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
int host_threads(const pt_context* ctx);

// rows owned by `rank`: r with (r / rb) % world == rank
int32_t count_local_rows(int32_t H, int32_t rank, int32_t world, int32_t rb) {
    int32_t n = 0;
    for (int32_t r = 0; r < H; ++r)
        if ((r / rb) % world == rank) ++n;
    return n;
}
inline int32_t global_row(const pt_context* c, int32_t lrow) {
    return ((lrow / c->rows_per_block) * c->world + c->rank) * c->rows_per_block + (lrow % c->rows_per_block);
}

// ------------------------------------------------------------------------------------------
// Encounter order of the reference's per-object tree (NodeOnHost::build, main.cpp:210-262):
// leaf when <= 6 triangles; otherwise split at the MEAN of the centroids on axis depth%3
// (centroid <= mean goes right, main.cpp:241-244), rotating the axis while one side is empty.
// prog.cl:159-181 always descends left first, so the order in which it can meet triangles is
// the depth-first, left-first concatenation of the leaves.  Only that order is needed here.
// The recursion is flattened: ONE index array is partitioned in place, stably (left part first, as the reference visits
// it), the centroids are computed once, and the leaf order that results IS the encounter order.  The mean of a node is
// still summed sequentially in index order -- float addition, the reference's rounding (main.cpp:224-236) -- but disjoint
// subtrees are independent and run on separate threads (1M triangles: 228 -> ~40 ms, profiles/r03/e_*).
struct RefOrder {
    const std::vector<pt_triangle>& tris;
    std::vector<int32_t>& rank;
    int32_t& next_rank;
    bool degenerate = false;

    struct Range { int32_t begin, end, depth; };

    // partitions [r.begin, r.end) of idx; returns the size of the left part, 0 for a leaf, -1 for the reference's endless loop
    static int32_t split(const Range& r, int32_t* idx, int32_t* tmp, const float* cx, const float* cy, const float* cz) {
        const int32_t n = r.end - r.begin;
        if (n <= 6) return 0;
        int32_t* ix = idx + r.begin;
        float mx = cx[ix[0]], my = cy[ix[0]], mz = cz[ix[0]];
        for (int32_t i = 1; i < n; ++i) {
            mx = mx + cx[ix[i]];
            my = my + cy[ix[i]];
            mz = mz + cz[ix[i]];
        }
        const float m[3] = {mx / (float)(unsigned long)n, my / (float)(unsigned long)n, mz / (float)(unsigned long)n};
        const float* c[3] = {cx, cy, cz};
        int axis = r.depth % 3;
        for (int tries = 0;; ++tries) {
            const float* ca = c[axis];
            const float ma = m[axis];
            int32_t nl = 0, nr = 0;
            for (int32_t i = 0; i < n; ++i) {
                const int32_t t = ix[i];
                if (ma >= ca[t]) tmp[r.begin + nr++] = t;      // centroid <= mean goes right (main.cpp:241-244)
                else ix[nl++] = t;                              // (nl <= i: never overtakes the read position)
            }
            if (nl != 0 && nr != 0) {
                std::memcpy(ix + nl, tmp + r.begin, sizeof(int32_t) * (size_t)nr);
                return nl;
            }
            if (nl == 0) std::memcpy(ix, tmp + r.begin, sizeof(int32_t) * (size_t)nr);    // everything went right: restore the order
            if (tries == 2) return -1;          // the reference loops forever here (main.cpp:246-257)
            axis = (axis + 1) % 3;
        }
    }

    // triangles [first, first + n) of `tris` (one object, add order)
    void run(int32_t first, int32_t n, int threads) {
        std::vector<float> cx((size_t)n), cy((size_t)n), cz((size_t)n);
        std::vector<int32_t> idx((size_t)n), tmp((size_t)n);
        for (int32_t i = 0; i < n; ++i) {
            const pt_triangle& t = tris[(size_t)(first + i)];
            cx[(size_t)i] = (t.r1.s[0] + t.r2.s[0] + t.r3.s[0]) / 3.0f;
            cy[(size_t)i] = (t.r1.s[1] + t.r2.s[1] + t.r3.s[1]) / 3.0f;
            cz[(size_t)i] = (t.r1.s[2] + t.r2.s[2] + t.r3.s[2]) / 3.0f;
            idx[(size_t)i] = i;
        }
        std::atomic<bool> bad(false);
        auto descend = [&](Range root) {          // depth-first over an explicit stack (the reference's tree can be very deep)
            std::vector<Range> st;
            st.push_back(root);
            while (!st.empty() && !bad.load(std::memory_order_relaxed)) {
                const Range r = st.back();
                st.pop_back();
                const int32_t nl = split(r, idx.data(), tmp.data(), cx.data(), cy.data(), cz.data());
                if (nl < 0) { bad.store(true); return; }
                if (nl == 0) continue;
                st.push_back(Range{r.begin + nl, r.end, r.depth + 1});
                st.push_back(Range{r.begin, r.begin + nl, r.depth + 1});
            }
        };
        // the top of the tree serially, until there are enough independent ranges; then one range per task
        std::vector<Range> open;
        open.push_back(Range{0, n, 0});
        const int32_t grain = std::max<int32_t>(n / (8 * std::max(threads, 1)), 4096);
        std::vector<Range> tasks;
        while (!open.empty() && !bad.load()) {
            const Range r = open.back();
            open.pop_back();
            if (threads <= 1 || r.end - r.begin <= grain) { tasks.push_back(r); continue; }
            const int32_t nl = split(r, idx.data(), tmp.data(), cx.data(), cy.data(), cz.data());
            if (nl < 0) { bad.store(true); break; }
            if (nl == 0) continue;
            open.push_back(Range{r.begin + nl, r.end, r.depth + 1});
            open.push_back(Range{r.begin, r.begin + nl, r.depth + 1});
        }
        if (!bad.load()) {
            const int nt = std::max(1, std::min<int>(threads, (int)tasks.size()));
            if (nt == 1) {
                for (const Range& r : tasks) descend(r);
            } else {
                std::sort(tasks.begin(), tasks.end(), [](const Range& a, const Range& b) { return a.end - a.begin > b.end - b.begin; });
                std::atomic<size_t> next(0);
                std::vector<std::thread> th;
                for (int k = 0; k < nt; ++k)
                    th.emplace_back([&]() {
                        for (;;) {
                            const size_t i = next.fetch_add(1);
                            if (i >= tasks.size()) return;
                            descend(tasks[i]);
                        }
                    });
                for (std::thread& t : th) t.join();
            }
        }
        if (bad.load()) { degenerate = true; return; }
        for (int32_t k = 0; k < n; ++k) rank[(size_t)(first + idx[(size_t)k])] = next_rank + k;
        next_rank += n;
    }
};

// ------------------------------------------------------------------------------------------
// Own BVH: binned SAH, BVH2, child boxes stored in the parent.
struct Aabb {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const Aabb& o) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], o.lo[a]); hi[a] = std::max(hi[a], o.hi[a]); }
    }
    void grow(const float p[3]) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct BuildPrim {
    Aabb box;
    float c[3];
    int32_t tri;  // add-order index
};

struct BvhBuilder {
    std::vector<BuildPrim> prims;
    std::vector<Node64> nodes;
    std::vector<int32_t> order;  // packed triangle order (add-order indices)
    int max_depth_seen = 0;
    int max_leaf = kMaxLeaf;     // leaf size limit of this attempt
    bool force_leaf = false;     // true: every subtree of <= max_leaf triangles becomes a leaf
    float visit_cost = 1.0f;     // SAH price of one node visit, in exact triangle tests

    static int need_levels(size_t n) {  // levels a median-split subtree of n prims needs
        size_t leaves = (n + 3) / 4;
        int l = 0;
        while ((size_t(1) << l) < leaves) ++l;
        return l;
    }

    // A leaf's packed position follows from its range: the recursion visits [lo, hi) ranges in ascending order and
    // every primitive ends up in exactly one leaf, so leaf [lo, hi) holds packed triangles order_base + lo ...
    // (`order` itself is filled from the final arrangement of `prims`, finish_order()).
    int32_t order_base = 0;
    int32_t make_leaf(size_t lo, size_t hi) const {
        const int32_t first = order_base + (int32_t)lo;
        const int32_t count = (int32_t)(hi - lo);
        return ~((first << 3) | (count - 1));
    }
    void finish_order() {
        order.resize((size_t)order_base + prims.size());
        for (size_t i = 0; i < prims.size(); ++i) order[(size_t)order_base + i] = prims[i].tri;
    }

    // Bounds of [lo, hi) and, unless the range becomes a leaf (returns false), its partition point.
    // The partition is STABLE (both sides keep their order), through `scratch`: the arrangement of `prims` -- hence the
    // packed triangle order -- is then the same however a range was split: serially, or, for the big ranges at the top of
    // the tree, with bounds, bins and partition spread over `split_threads` threads (min / max / counts: any grouping
    // gives the same bins).  1M triangles: the serial top of the tree was half of the threaded build's time.
    std::unique_ptr<BuildPrim[]> scratch;   // as long as prims (uninitialised); a range only ever uses its own slice
    size_t scratch_len = 0;
    void need_scratch() {
        if (scratch_len < prims.size()) { scratch.reset(new BuildPrim[prims.size()]); scratch_len = prims.size(); }
    }
    static constexpr int NB = 16;
    struct Bins {
        Aabb bb[3][NB];
        int cnt[3][NB];
        void reset() {
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < NB; ++k) { bb[a][k].reset(); cnt[a][k] = 0; }
        }
    };
    static int bin_of(float c, float lo, float scale) {
        int k = (int)((c - lo) * scale);
        return std::min(std::max(k, 0), NB - 1);
    }
    bool split(size_t lo, size_t hi, int depth, Aabb* box, size_t* mid_out, int split_threads = 1) {
        const size_t n = hi - lo;
        const int mt = (split_threads > 1 && n >= 65536) ? split_threads : 1;
        Aabb b, cb;
        b.reset();
        cb.reset();
        if (mt > 1) {
            std::vector<Aabb> pb((size_t)mt), pc((size_t)mt);
            for (int k = 0; k < mt; ++k) { pb[(size_t)k].reset(); pc[(size_t)k].reset(); }
            std::atomic<int> slot(0);
            parallel_for(n, 1 << 14, mt, [&](size_t cbeg, size_t cend) {
                Aabb x, y;
                x.reset();
                y.reset();
                for (size_t i = lo + cbeg; i < lo + cend; ++i) { x.grow(prims[i].box); y.grow(prims[i].c); }
                const size_t sidx = (size_t)slot.fetch_add(1);
                pb[sidx] = x;
                pc[sidx] = y;
            });
            for (int k = 0; k < mt; ++k) { b.grow(pb[(size_t)k]); cb.grow(pc[(size_t)k]); }
        } else {
            for (size_t i = lo; i < hi; ++i) { b.grow(prims[i].box); cb.grow(prims[i].c); }
        }
        *box = b;
        if (n <= 1) return false;

        // --- binned SAH over the three axes
        float ext[3], scale[3];
        for (int a = 0; a < 3; ++a) {
            ext[a] = cb.hi[a] - cb.lo[a];
            scale[a] = ext[a] > 0.f ? (float)NB / ext[a] : 0.f;
        }
        Bins bins;
        bins.reset();
        auto bin_range = [&](Bins& out, size_t ibeg, size_t iend) {
            for (int a = 0; a < 3; ++a) {
                if (!(ext[a] > 0.f)) continue;
                for (size_t i = ibeg; i < iend; ++i) {
                    const int k = bin_of(prims[i].c[a], cb.lo[a], scale[a]);
                    out.bb[a][k].grow(prims[i].box);
                    out.cnt[a][k]++;
                }
            }
        };
        if (mt > 1) {
            std::vector<Bins> part((size_t)mt);
            for (Bins& p : part) p.reset();
            std::atomic<int> slot(0);
            parallel_for(n, 1 << 14, mt, [&](size_t cbeg, size_t cend) { bin_range(part[(size_t)slot.fetch_add(1)], lo + cbeg, lo + cend); });
            for (const Bins& p : part)
                for (int a = 0; a < 3; ++a)
                    for (int k = 0; k < NB; ++k) { bins.bb[a][k].grow(p.bb[a][k]); bins.cnt[a][k] += p.cnt[a][k]; }
        } else {
            bin_range(bins, lo, hi);
        }
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3; ++a) {
            if (!(ext[a] > 0.f)) continue;
            const Aabb* bb = bins.bb[a];
            const int* cnt = bins.cnt[a];
            float la[NB], ra[NB];
            int lc[NB], rc[NB];
            Aabb acc;
            acc.reset();
            int c = 0;
            for (int k = 0; k < NB; ++k) { acc.grow(bb[k]); c += cnt[k]; la[k] = acc.half_area(); lc[k] = c; }
            acc.reset();
            c = 0;
            for (int k = NB - 1; k >= 0; --k) { acc.grow(bb[k]); c += cnt[k]; ra[k] = acc.half_area(); rc[k] = c; }
            for (int k = 0; k < NB - 1; ++k) {
                if (lc[k] == 0 || rc[k + 1] == 0) continue;
                float cost = la[k] * (float)lc[k] + ra[k + 1] * (float)rc[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
            }
        }
        // SAH termination: a node visit (64 B, two slab tests) is priced like one exact triangle test
        const float leaf_cost = b.half_area() * (float)n;
        if (n <= (size_t)max_leaf && (force_leaf || !(best_cost + visit_cost * b.half_area() < leaf_cost))) return false;

        size_t mid = lo;
        bool median = (best_axis < 0);
        if (!median) {
            const float blo = cb.lo[best_axis], bsc = scale[best_axis];
            const int ax = best_axis, bbin = best_bin;
            auto goes_left = [=](const BuildPrim& p) { return bin_of(p.c[ax], blo, bsc) <= bbin; };
            need_scratch();
            if (mt > 1) {
                std::vector<size_t> nl((size_t)mt + 1, 0), bounds_((size_t)mt + 1, 0);
                const size_t per = (n + (size_t)mt - 1) / (size_t)mt;
                for (int k = 0; k <= mt; ++k) bounds_[(size_t)k] = std::min(n, (size_t)k * per);
                HostPool::get().run((size_t)mt, mt, [&](size_t k) {
                    size_t c = 0;
                    for (size_t i = lo + bounds_[k]; i < lo + bounds_[k + 1]; ++i) c += goes_left(prims[i]) ? 1 : 0;
                    nl[k + 1] = c;
                });
                for (int k = 0; k < mt; ++k) nl[(size_t)k + 1] += nl[(size_t)k];       // left elements in front of chunk k
                const size_t total_left = nl[(size_t)mt];
                HostPool::get().run((size_t)mt, mt, [&](size_t k) {
                    size_t l = lo + nl[k];
                    size_t r = lo + total_left + (bounds_[k] - nl[k]);
                    for (size_t i = lo + bounds_[k]; i < lo + bounds_[k + 1]; ++i) {
                        if (goes_left(prims[i])) scratch[l++] = prims[i]; else scratch[r++] = prims[i];
                    }
                });
                parallel_for(n, 1 << 14, mt, [&](size_t cbeg, size_t cend) { std::memcpy(&prims[lo + cbeg], &scratch[lo + cbeg], sizeof(BuildPrim) * (cend - cbeg)); });
                mid = lo + total_left;
            } else {
                size_t w = lo, r = lo;
                for (size_t i = lo; i < hi; ++i) {
                    if (goes_left(prims[i])) { if (w != i) prims[w] = prims[i]; ++w; }
                    else scratch[r++] = prims[i];
                }
                if (r > lo) std::memcpy(&prims[w], &scratch[lo], sizeof(BuildPrim) * (r - lo));
                mid = w;
            }
            size_t big = std::max(mid - lo, hi - mid);
            if (mid == lo || mid == hi || depth + 1 + need_levels(big) > kMaxDepth) median = true;
        }
        if (median) {
            int a = 0;
            float e = -1.f;
            for (int k = 0; k < 3; ++k) { float ex = cb.hi[k] - cb.lo[k]; if (ex > e) { e = ex; a = k; } }
            mid = lo + n / 2;
            std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi,
                             [a](const BuildPrim& x, const BuildPrim& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.tri < y.tri); });
        }
        *mid_out = mid;
        return true;
    }

    static void set_children(Node64& nd, int32_t l, int32_t r, const Aabb& lb, const Aabb& rb) {
        for (int a = 0; a < 3; ++a) { nd.q[a][0] = lb.lo[a]; nd.q[a][1] = lb.hi[a]; nd.q[a][2] = rb.lo[a]; nd.q[a][3] = rb.hi[a]; }
        nd.left = l;
        nd.right = r;
        nd.pad[0] = nd.pad[1] = 0;
    }

    // Subtree of [lo, hi) appended to `out` in preorder; returns the child reference (index into `out`, or a leaf),
    // *box receives the bounds, *deepest the depth of the deepest range.
    int32_t build_into(std::vector<Node64>& out, int* deepest, size_t lo, size_t hi, int depth, Aabb* box) {
        *deepest = std::max(*deepest, depth);
        size_t mid;
        if (!split(lo, hi, depth, box, &mid)) return make_leaf(lo, hi);
        const int32_t me = (int32_t)out.size();
        out.emplace_back();
        Aabb lb, rb;
        const int32_t l = build_into(out, deepest, lo, mid, depth + 1, &lb);
        const int32_t r = build_into(out, deepest, mid, hi, depth + 1, &rb);
        set_children(out[(size_t)me], l, r, lb, rb);
        return me;
    }
    int32_t build(size_t lo, size_t hi, int depth, Aabb* box) { return build_into(nodes, &max_depth_seen, lo, hi, depth, box); }

    // The same tree, node for node, from several threads: the top of the tree is split serially down to ranges of at most
    // `grain` primitives, the ranges are built concurrently (they are disjoint slices of `prims`) into private node
    // arrays, and a last preorder walk splices them into `nodes` (interior references move by the splice offset; leaf
    // references are positions and do not move).
    struct Part {
        int kind;               // 0 leaf reference, 1 top node, 2 task
        int32_t v;              // the reference / index into tops / index into tasks
        Aabb box;
    };
    struct TopNode { Part l, r; };
    struct Task {
        size_t lo, hi;
        int depth;
        std::vector<Node64> out;
        int32_t root = 0;
        int deepest = 0;
        Aabb box;
    };
    // The top of the tree, level by level: the ranges of a level are independent, so while they are fewer than the threads
    // each is split ON all threads (bounds, bins and partition in parallel), and once they are more, each BY one thread, side by
    // side.  Ranges of at most `grain` primitives become tasks.  Which node gets which index in `tops` is irrelevant: splice()
    // numbers the final nodes in preorder.
    Part split_top(std::vector<TopNode>& tops, std::vector<Task>& tasks, size_t grain, size_t n, int threads) {
        struct Open { size_t lo, hi; int depth; int32_t parent; int side; };      // parent -1: the root
        struct Res { bool inner; Aabb box; size_t mid; };
        Part root;
        root.kind = 0;
        root.v = 0;
        auto place = [&](const Open& o, const Part& p) {
            if (o.parent < 0) root = p;
            else if (o.side == 0) tops[(size_t)o.parent].l = p;
            else tops[(size_t)o.parent].r = p;
        };
        auto as_task = [&](const Open& o) {
            Part p;
            p.kind = 2;
            p.v = (int32_t)tasks.size();
            tasks.emplace_back();
            tasks.back().lo = o.lo;
            tasks.back().hi = o.hi;
            tasks.back().depth = o.depth;
            place(o, p);
        };
        std::vector<Open> level, nextl;
        {
            const Open o{0, n, 0, -1, 0};
            if (n <= grain) as_task(o); else level.push_back(o);
        }
        while (!level.empty()) {
            std::vector<Res> res(level.size());
            if ((int)level.size() < threads) {
                for (size_t i = 0; i < level.size(); ++i)
                    res[i].inner = split(level[i].lo, level[i].hi, level[i].depth, &res[i].box, &res[i].mid, threads);
            } else {
                HostPool::get().run(level.size(), threads, [&](size_t i) {
                    res[i].inner = split(level[i].lo, level[i].hi, level[i].depth, &res[i].box, &res[i].mid, 1);
                });
            }
            nextl.clear();
            for (size_t i = 0; i < level.size(); ++i) {
                const Open& o = level[i];
                max_depth_seen = std::max(max_depth_seen, o.depth);
                Part p;
                p.box = res[i].box;
                if (!res[i].inner) {
                    p.kind = 0;
                    p.v = make_leaf(o.lo, o.hi);
                    place(o, p);
                    continue;
                }
                p.kind = 1;
                p.v = (int32_t)tops.size();
                tops.emplace_back();
                place(o, p);
                const Open kids[2] = {{o.lo, res[i].mid, o.depth + 1, p.v, 0}, {res[i].mid, o.hi, o.depth + 1, p.v, 1}};
                for (const Open& k : kids) {
                    if (k.hi - k.lo <= grain) as_task(k); else nextl.push_back(k);
                }
            }
            level.swap(nextl);
        }
        return root;
    }
    // Preorder numbering of the final tree: top nodes are written as the walk passes them, a task's block is only given its
    // place (task_off) -- the blocks are copied afterwards, side by side (copy_tasks).
    int32_t splice(const std::vector<TopNode>& tops, const std::vector<Task>& tasks, std::vector<int32_t>& task_off, int32_t* next_index, const Part& p, Aabb* box) {
        if (p.kind == 0) { *box = p.box; return p.v; }
        if (p.kind == 2) {
            const Task& t = tasks[(size_t)p.v];
            *box = t.box;
            if (t.root < 0) return t.root;
            const int32_t off = *next_index;
            task_off[(size_t)p.v] = off;
            *next_index += (int32_t)t.out.size();
            return off + t.root;
        }
        const int32_t me = (*next_index)++;
        Aabb lb, rb;
        const int32_t l = splice(tops, tasks, task_off, next_index, tops[(size_t)p.v].l, &lb);
        const int32_t r = splice(tops, tasks, task_off, next_index, tops[(size_t)p.v].r, &rb);
        if (nodes.size() < (size_t)*next_index) nodes.resize((size_t)*next_index);
        set_children(nodes[(size_t)me], l, r, lb, rb);
        Aabb b = lb;
        b.grow(rb);
        *box = b;
        return me;
    }
    void copy_tasks(const std::vector<Task>& tasks, const std::vector<int32_t>& task_off, int threads) {
        HostPool::get().run(tasks.size(), threads, [&](size_t i) {
            const Task& t = tasks[i];
            if (t.root < 0) return;
            const int32_t off = task_off[i];
            for (size_t k = 0; k < t.out.size(); ++k) {
                Node64 nd = t.out[k];
                if (nd.left >= 0) nd.left += off;
                if (nd.right >= 0) nd.right += off;
                nodes[(size_t)off + k] = nd;
            }
        });
    }
    int32_t build_parallel(int threads, Aabb* box) {
        const size_t n = prims.size();
        const size_t grain = std::max<size_t>(4096, n / ((size_t)threads * 8));
        std::vector<TopNode> tops;
        std::vector<Task> tasks;
        PhaseClock clk("sah build");
        need_scratch();
        const Part root = split_top(tops, tasks, grain, n, threads);
        clk.lap("top of the tree");
        HostPool::get().run(tasks.size(), threads, [&](size_t i) {
            Task& t = tasks[i];
            t.out.reserve(t.hi - t.lo);
            t.root = build_into(t.out, &t.deepest, t.lo, t.hi, t.depth, &t.box);
        });
        clk.lap("subtrees");
        for (const Task& t : tasks) max_depth_seen = std::max(max_depth_seen, t.deepest);
        std::vector<int32_t> task_off(tasks.size(), 0);
        int32_t total = 0;
        size_t upper = tops.size();
        for (const Task& t : tasks) upper += t.out.size();
        nodes.resize(upper);                                     // (all of them are written below)
        const int32_t r = splice(tops, tasks, task_off, &total, root, box);
        nodes.resize((size_t)total);
        copy_tasks(tasks, task_off, threads);
        clk.lap("splice");
        return r;
    }
};

// Triangle bounds, padded: prog.cl:104-106 accepts points a few rounding errors outside the
// exact triangle, and the box test must never reject a ray the triangle test would accept.
Aabb padded_bounds(const pt_triangle& t) {
    Aabb b;
    b.reset();
    b.grow(t.r1.s);
    b.grow(t.r2.s);
    b.grow(t.r3.s);
    float m = 0.f;
    for (int a = 0; a < 3; ++a) m = std::max(m, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    float pad = m * 1e-5f + 1e-6f;
    for (int a = 0; a < 3; ++a) { b.lo[a] -= pad; b.hi[a] += pad; }
    return b;
}

// One build attempt.  Returns PT_OK and fills bld.
void compute_cost_boxes_impl(pt_context* ctx);
void compute_cost_boxes_from(pt_context* ctx, const Aabb* boxes);

int build_attempt(pt_context* ctx, BvhBuilder& bld, std::vector<BuildPrim>& prims, const std::vector<int32_t>& flat, int max_leaf, bool force_leaf) {
    bld = BvhBuilder();
    bld.prims.swap(prims);         // (the caller has no further use for them)
    bld.max_leaf = max_leaf;
    bld.force_leaf = force_leaf;
    bld.visit_cost = (float)ctx->sah_visit_cost * 0.1f;
    bld.nodes.reserve(bld.prims.size());
    bld.order.reserve(bld.prims.size() + flat.size());
    bld.order = flat;            // the flat list comes first in packed order; leaf ranges start behind it
    bld.order_base = (int32_t)flat.size();
    // The root must be an interior node: wrap a leaf / an empty scene.
    Aabb lb, rb;
    lb.reset();
    rb.reset();
    if (bld.prims.size() <= (size_t)max_leaf) {
        bld.nodes.emplace_back();
        bld.force_leaf = true;
        int32_t l = bld.prims.empty() ? ~0 : bld.build(0, bld.prims.size(), 1, &lb);
        if (bld.prims.empty()) lb.reset();
        Node64& nd = bld.nodes[0];
        for (int a = 0; a < 3; ++a) { nd.q[a][0] = lb.lo[a]; nd.q[a][1] = lb.hi[a]; nd.q[a][2] = rb.lo[a]; nd.q[a][3] = rb.hi[a]; }
        nd.left = l;
        nd.right = ~0;
        nd.pad[0] = nd.pad[1] = 0;
        if (l >= 0) return fail(ctx, PT_ESCENE, "internal: small scene did not become a leaf");
    } else {
        Aabb box;
        int threads = ctx->build_threads > 0 ? ctx->build_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (bld.prims.size() < 32768) threads = 1;
        int32_t root = threads > 1 ? bld.build_parallel(threads, &box) : bld.build(0, bld.prims.size(), 0, &box);
        if (root != 0) return fail(ctx, PT_ESCENE, "internal: BVH root is not node 0");
    }
    bld.finish_order();
    return PT_OK;
}

// The big-triangle list: removes the chosen primitives from `prims` and returns their
// add-order triangle indices (in add order).
std::vector<int32_t> select_flat_list(const pt_context* ctx, std::vector<BuildPrim>& prims) {
    std::vector<int32_t> flat;
    if (ctx->flat_list > 0 && !prims.empty()) {
        const int threads = host_threads(ctx);
        // only the `cand` biggest need to be in order (ties: add order), the others only need their common box
        const size_t n = prims.size();
        std::vector<float> area(n);
        parallel_for(n, 1 << 15, threads, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) area[i] = prims[i].box.half_area(); });
        const size_t cand = std::min<size_t>((size_t)ctx->flat_list, n);
        auto bigger = [&](size_t a, size_t b) { return area[a] > area[b] || (area[a] == area[b] && a < b); };
        std::vector<size_t> top;                     // the cand biggest, in order: one pass with a small sorted buffer
        top.reserve(cand + 1);
        for (size_t i = 0; i < n; ++i) {
            if (top.size() == cand && !bigger(i, top.back())) continue;
            top.insert(std::upper_bound(top.begin(), top.end(), i, bigger), i);
            if (top.size() > cand) top.pop_back();
        }
        std::vector<char> in_top(n, 0);
        for (size_t k : top) in_top[k] = 1;
        // box of everything that is not among them (min / max: any grouping gives the same box)
        std::vector<Aabb> part((size_t)threads);
        for (Aabb& p : part) p.reset();
        {
            std::atomic<int> slot(0);
            parallel_for(n, 1 << 15, threads, [&](size_t b, size_t e) {
                Aabb acc;
                acc.reset();
                for (size_t i = b; i < e; ++i)
                    if (!in_top[i]) acc.grow(prims[i].box);
                part[(size_t)slot.fetch_add(1)] = acc;
            });
        }
        std::vector<Aabb> rest(cand + 1);             // rest[k] = box of top[k..] and all the others
        Aabb tail;
        tail.reset();
        for (const Aabb& p : part) tail.grow(p);
        rest[cand] = tail;
        for (size_t k = cand; k-- > 0;) { tail.grow(prims[top[k]].box); rest[k] = tail; }
        // the largest m such that each of the m biggest is >= 1/16 of the box around all the others
        std::vector<char> is_flat(n, 0);
        size_t n_flat = 0;
        for (size_t m = cand; m > 0; --m) {
            const float smallest = prims[top[m - 1]].box.half_area(), others = rest[m].half_area();
            if (smallest >= others * (1.0f / 16.0f)) {
                for (size_t k = 0; k < m; ++k) is_flat[top[k]] = 1;
                n_flat = m;
                break;
            }
        }
        if (n_flat > 0) {                             // take them out in place, add order kept on both sides
            size_t w = 0;
            for (size_t i = 0; i < n; ++i) {
                if (is_flat[i]) flat.push_back(prims[i].tri);
                else { if (w != i) prims[w] = prims[i]; ++w; }
            }
            prims.resize(w);
        }
    }
    return flat;
}

// Which SAH tree a bvh_policy stands for, read by the host builder (build_and_pack) AND the device builder (build_on_device) so that
// a build that falls back from the device to the host -- a host-only context, non-finite triangles, a median split, eight
// triangles or fewer -- gives the tree the option names: 5 ("the SAH tree, built on the device") is policy 0's tree; 4 (device
// LBVH) has no host form and falls back to policy 0's tree as well.
int tree_policy(const pt_context* ctx) { return ctx->bvh_policy >= 4 ? 0 : ctx->bvh_policy; }

int build_and_pack(pt_context* ctx) {
    PhaseClock clk("pt_upload_triangles");
    const size_t n = ctx->tris.size();
    const int threads = host_threads(ctx);
    // padded boxes of ALL triangles once (the builder's primitives and the wavefront's cost boxes both come from them)
    std::vector<Aabb> boxes(n);
    std::vector<char> finite(n);
    parallel_for(n, 1 << 14, threads, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const pt_triangle& t = ctx->tris[i];
            bool f = true;
            for (int a = 0; a < 3; ++a)
                f = f && std::isfinite(t.r1.s[a]) && std::isfinite(t.r2.s[a]) && std::isfinite(t.r3.s[a]);
            finite[i] = f ? 1 : 0;      // a non-finite triangle cannot be hit (prog.cl:99-106 compares NaN) and has no box
            boxes[i] = padded_bounds(t);
        }
    });
    std::vector<BuildPrim> prims(n);
    parallel_for(n, 1 << 14, threads, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            BuildPrim& p = prims[i];
            p.box = boxes[i];
            for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
            p.tri = (int32_t)i;
        }
    });
    {
        size_t w = 0;
        for (size_t i = 0; i < n; ++i)
            if (finite[i]) { if (w != i) prims[w] = prims[i]; ++w; }
        prims.resize(w);
    }
    clk.lap("primitive boxes");
    std::vector<int32_t> flat = select_flat_list(ctx, prims);
    clk.lap("big-triangle list");
    BvhBuilder bld;
    // (the SAH tree of a policy is the same on the host and on the device: tree_policy() is read by both builders)
    const int tp = tree_policy(ctx);
    int rc = tp <= 1 ? build_attempt(ctx, bld, prims, flat, 4, false) : build_attempt(ctx, bld, prims, flat, tp == 2 ? 4 : 8, true);
    if (rc != PT_OK) return rc;
    clk.lap("SAH build");
    ctx->n_flat = (int)flat.size();
    if (bld.max_depth_seen > kMaxDepth) return fail(ctx, PT_ESCENE, "internal: BVH deeper than the traversal stack");
    ctx->bvh_depth = bld.max_depth_seen;
    ctx->nodes.swap(bld.nodes);
    ctx->orig.swap(bld.order);
    const size_t m = ctx->orig.size();
    ctx->packets.resize(std::max<size_t>(m, 1));
    ctx->meta.resize(std::max<size_t>(m, 1));
    std::memset(ctx->packets.data(), 0, sizeof(TriPacket) * ctx->packets.size());
    std::memset(ctx->meta.data(), 0, sizeof(TriMeta) * ctx->meta.size());
    compute_cost_boxes_from(ctx, boxes.data());
    clk.lap("cost boxes");
    parallel_for(m, 1 << 14, threads, [&](size_t kb, size_t ke) {
        for (size_t k = kb; k < ke; ++k) {
            const pt_triangle& t = ctx->tris[ctx->orig[k]];
            float* v = ctx->packets[k].v;
            v[0] = t.r1.s[0]; v[1] = t.r1.s[1]; v[2] = t.r1.s[2];
            v[3] = t.r2.s[0]; v[4] = t.r2.s[1]; v[5] = t.r2.s[2];
            v[6] = t.r3.s[0]; v[7] = t.r3.s[1]; v[8] = t.r3.s[2];
            v[9] = t.N.s[0]; v[10] = t.N.s[1]; v[11] = t.N.s[2];
            ctx->meta[k].rank = ctx->enc_rank[ctx->orig[k]];
            ctx->meta[k].mati = t.mati;
        }
    });
    clk.lap("packets + meta");
    return PT_OK;
}

// bounding boxes of the complex objects (more than 16 triangles), for the wavefront cost classes
// boxes of the objects with more than 16 triangles (the wavefront variant's ray cost classes); `boxes` = the padded bounds of
// every add-order triangle if the caller has them already
void compute_cost_boxes_from(pt_context* ctx, const Aabb* boxes) {
    struct OB { Aabb b; size_t n; };
    std::vector<OB> obs;
    const int threads = host_threads(ctx);
    for (size_t o = 0; o < ctx->obj_begin.size(); ++o) {
        const size_t lo = (size_t)ctx->obj_begin[o], hi = o + 1 < ctx->obj_begin.size() ? (size_t)ctx->obj_begin[o + 1] : ctx->tris.size();
        if (hi - lo <= 16) continue;
        OB ob;
        ob.b.reset();
        ob.n = hi - lo;
        std::vector<Aabb> part((size_t)threads);          // (min / max: any grouping gives the same box)
        for (Aabb& p : part) p.reset();
        std::atomic<int> slot(0);
        parallel_for(hi - lo, 1 << 15, threads, [&](size_t b, size_t e) {
            Aabb acc;
            acc.reset();
            for (size_t i = lo + b; i < lo + e; ++i) acc.grow(boxes ? boxes[i] : padded_bounds(ctx->tris[i]));
            part[(size_t)slot.fetch_add(1)] = acc;
        });
        for (const Aabb& p : part) ob.b.grow(p);
        if (std::isfinite(ob.b.half_area())) obs.push_back(ob);
    }
    std::sort(obs.begin(), obs.end(), [](const OB& x, const OB& y) { return x.n > y.n; });
    while (obs.size() > (size_t)kWfMaxCostBoxes) {       // fold the smallest objects into one box
        obs[obs.size() - 2].b.grow(obs.back().b);
        obs[obs.size() - 2].n += obs.back().n;
        obs.pop_back();
    }
    ctx->cost_boxes.clear();
    for (const OB& ob : obs) {
        for (int a = 0; a < 3; ++a) ctx->cost_boxes.push_back(ob.b.lo[a]);
        for (int a = 0; a < 3; ++a) ctx->cost_boxes.push_back(ob.b.hi[a]);
    }
}
void compute_cost_boxes_impl(pt_context* ctx) { compute_cost_boxes_from(ctx, nullptr); }

// Stack entries a traversal of this tree needs: sentinel + one far child per level + the slot above the top
// that Trav::node_step writes unconditionally (+ 2 spare), rounded to even.
// Entries of a lane's traversal stack.  A visit of an interior node at depth d (root: 0) finds at most d far children
// pushed by its ancestors above the sentinel (entry 0) and stores its own far child one above the top, at index
// <= d + 1, whether or not it keeps it (Trav::node_step); leaves store nothing.  So the deepest interior node's
// depth + 2 entries suffice; rounded up to an even count.  Every entry is LDS that bounds the resident waves of the
// kernels reading nodes from global memory (launch_cfg), so the bound is the exact one, measured on the packed tree.
int deepest_interior_node(const std::vector<Node64>& nodes) {
    // Every builder here numbers a child behind its parent (preorder, or merge order counted downwards): one pass in index
    // order then knows every depth (1M triangles: ~1 ms; the walk below took 8).
    {
        std::vector<uint8_t> depth(nodes.size(), 0);
        int deepest = 0;
        bool ordered = true;
        for (size_t i = 0; i < nodes.size() && ordered; ++i) {
            const int d = depth[i];
            deepest = std::max(deepest, d);
            const int32_t kids[2] = {nodes[i].left, nodes[i].right};
            for (int32_t c : kids)
                if (c >= 0) {
                    if ((size_t)c <= i || (size_t)c >= nodes.size() || d >= 254) { ordered = false; break; }
                    depth[(size_t)c] = (uint8_t)(d + 1);
                }
        }
        if (ordered) return deepest;
    }
    int deepest = 0;
    std::vector<std::pair<int32_t, int>> todo;
    if (!nodes.empty()) todo.emplace_back(0, 0);
    while (!todo.empty()) {
        const std::pair<int32_t, int> it = todo.back();
        todo.pop_back();
        deepest = std::max(deepest, it.second);
        const Node64& nd = nodes[(size_t)it.first];
        if (nd.left >= 0) todo.emplace_back(nd.left, it.second + 1);
        if (nd.right >= 0) todo.emplace_back(nd.right, it.second + 1);
    }
    return deepest;
}
// 4-wide nodes: a visit finds at most `pending` entries above the sentinel and stores its three other children above
// the top, kept or not (Trav::wide_step)
int wide_stack_entries(int pending) { return ((pending + 4) + 1) & ~1; }
int stack_entries_for(int interior_depth) { return std::min(kStackEntries, ((interior_depth + 2) + 1) & ~1); }

constexpr size_t kLdsPerCu = 160 * 1024;
// next to the stacks and the staged nodes: the big-triangle list (96 B each) and, in wf_intersect, one class byte per ray
// of a trip (16 waves x 256) + the compaction counters
constexpr size_t kLdsSlack = 32 * 100 + 4096 + 1024 + 256;

// Does the whole tree fit next to two 512-thread workgroups per CU (kNodesLds: 16-bit references)?
bool whole_tree_fits_lds(size_t n_nodes, size_t n_tris, int interior_depth, int n_flat) {
    const bool s16 = n_nodes <= 32767 && n_tris <= 4096;
    const size_t block = (size_t)kLdsBlockBase;      // (the 768-thread k_render instance is taken only where it fits too)
    return s16 && (size_t)kLdsNodeBytes * n_nodes + 16 + (size_t)stack_entries_for(interior_depth) * 2 * block + (size_t)n_flat * 100 + 64 <= kLdsPerCu / 2;   // (+ flat list: packet + box + group mask per triangle)
}

// Treelet: when the tree is too large for LDS, the T nodes with the largest boxes --
// the ones a ray is most likely to visit; a child's box lies inside its parent's, so they form a connected
// top of the tree -- are renumbered to [0, T) and every workgroup stages exactly those.  T is what one
// 1,024-thread workgroup per CU has left next to its 32-bit stacks.  The rest keeps its depth-first order.
// Returns T (0: no treelet).
int reindex_treelet(std::vector<Node64>& nodes, int interior_depth, int want) {
    const size_t n = nodes.size();
    const size_t stacks = (size_t)stack_entries_for(interior_depth) * 4 * 1024;
    if (stacks + kLdsSlack + 64 * sizeof(Node64) > kLdsPerCu) return 0;
    size_t cap = (kLdsPerCu - kLdsSlack - stacks) / sizeof(Node64);
    if (want > 0) cap = std::min(cap, (size_t)want);
    const size_t T = std::min(cap, n);
    if (T < 2) return 0;
    auto area = [&](int32_t i) {
        const Node64& nd = nodes[(size_t)i];
        float d[3];
        for (int a = 0; a < 3; ++a) d[a] = std::max(nd.q[a][1], nd.q[a][3]) - std::min(nd.q[a][0], nd.q[a][2]);
        const float h = d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
        return std::isfinite(h) ? h : std::numeric_limits<float>::infinity();
    };
    typedef std::pair<float, int32_t> Item;        // (area, -index): ties go to the lower index
    std::priority_queue<Item> pq;
    std::vector<int32_t> newidx(n, -1);
    pq.push(Item(area(0), 0));
    int32_t next = 0;
    while (!pq.empty() && (size_t)next < T) {
        const int32_t i = -pq.top().second;
        pq.pop();
        newidx[(size_t)i] = next++;
        const Node64& nd = nodes[(size_t)i];
        if (nd.left >= 0) pq.push(Item(area(nd.left), -nd.left));
        if (nd.right >= 0) pq.push(Item(area(nd.right), -nd.right));
    }
    const int32_t t_final = next;
    for (size_t i = 0; i < n; ++i)
        if (newidx[i] < 0) newidx[i] = next++;
    std::vector<Node64> out(n);
    for (size_t i = 0; i < n; ++i) {
        Node64 nd = nodes[i];
        if (nd.left >= 0) nd.left = newidx[(size_t)nd.left];
        if (nd.right >= 0) nd.right = newidx[(size_t)nd.right];
        out[(size_t)newidx[i]] = nd;
    }
    nodes.swap(out);
    return t_final;
}

// The listed triangles whose bounding boxes coincide -- the two halves of an axis-aligned wall -- are culled with ONE
// box test (Trav::flat_pass): group them by the vertex extremes the kernel pads into that box.
static void group_flat_boxes(pt_context* ctx) {
    ctx->n_fbox = 0;
    float ext[32][6];
    for (int k = 0; k < ctx->n_flat; ++k) {
        const float* v = ctx->packets[(size_t)k].v;          // r1, r2, r3
        float e[6];
        for (int a = 0; a < 3; ++a) {
            e[a] = std::min(std::min(v[a], v[3 + a]), v[6 + a]);
            e[3 + a] = std::max(std::max(v[a], v[3 + a]), v[6 + a]);
        }
        int b = 0;
        for (; b < ctx->n_fbox; ++b)
            if (std::memcmp(ext[b], e, sizeof e) == 0) break;
        if (b == ctx->n_fbox) {
            std::memcpy(ext[b], e, sizeof e);
            ctx->fbox_rep[b] = (uint8_t)k;
            ctx->fbox_mask[b] = 0;
            ctx->n_fbox++;
        }
        ctx->fbox_mask[b] |= 1u << k;
    }
}

// d_bvh2: the tree as it is in ctx->nodes, already in device memory (a device-built tree) -- the 4-wide collapse then runs
// there too (pt_widedev.hip: the same nodes) and *wide_on_device reports that ctx->d_nodes4 is in place.
int plan_node_placement(pt_context* ctx, const float4* d_bvh2 = nullptr, bool* wide_on_device = nullptr) {
    PhaseClock clk("node placement");
    if (wide_on_device) *wide_on_device = false;
    ctx->treelet_nodes = 0;
    group_flat_boxes(ctx);
    ctx->interior_depth = deepest_interior_node(ctx->nodes);
    clk.lap("list boxes + interior depth");
    if (ctx->interior_depth + 2 > kStackEntries) return fail(ctx, PT_ESCENE, "internal: BVH deeper than the traversal stack");
    const bool fits = whole_tree_fits_lds(ctx->nodes.size(), ctx->orig.size(), ctx->interior_depth, ctx->n_flat);
    if (ctx->treelet != 0 && !fits) ctx->treelet_nodes = reindex_treelet(ctx->nodes, ctx->interior_depth, ctx->treelet);
    // 4-wide nodes for trees read from global memory -- unless their worst-case stack would not leave room for four
    // 256-thread workgroups per CU (then the BVH2 path stays)
    ctx->nodes4.clear();
    ctx->wide_pending = 0;
    if (ctx->wide_nodes == 2 || (ctx->wide_nodes == 1 && !fits && ctx->treelet_nodes == 0)) {
        if (d_bvh2 && ctx->treelet_nodes == 0 && ctx->wide_on_device != 0) {
            float4* d4 = nullptr;
            int n4 = 0, pending = 0;
            bool failed = false;
            PT_HIP(ctx, wide_device_build(d_bvh2, (int)ctx->nodes.size(), ctx->stream, &d4, &n4, &pending, &failed));
            if (!failed) {
                if (ctx->d_nodes4) (void)hipFree(ctx->d_nodes4);
                ctx->d_nodes4 = d4;
                ctx->nodes4.resize((size_t)n4);          // host copy for the debug getter and the stack sizing
                ctx->wide_pending = pending;
                PT_HIP(ctx, hipMemcpy(ctx->nodes4.data(), d4, sizeof(Node4q) * (size_t)n4, hipMemcpyDeviceToHost));
                if (wide_on_device) *wide_on_device = true;
            }
            clk.lap("4-wide nodes (device)");
            return PT_OK;
        }
        const unsigned hw = std::thread::hardware_concurrency();
        const int threads = ctx->build_threads > 0 ? ctx->build_threads : (int)std::min(16u, std::max(1u, hw));
        if (!build_wide_nodes(ctx->nodes, &ctx->nodes4, &ctx->wide_pending, threads)) ctx->nodes4.clear();
        clk.lap("4-wide nodes (host)");
    }
    return PT_OK;
}

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
int alloc_stack_overflow(pt_context* ctx) {
    if (ctx->d_stack_ovf) { PT_HIP(ctx, hipFree(ctx->d_stack_ovf)); ctx->d_stack_ovf = nullptr; }
    ctx->stack_ovf_lanes = 0;
    const int extra = ctx->nodes4.empty() ? 0 : wide_stack_entries(ctx->wide_pending) - ctx->wide_lds_entries;
    if (extra <= 0) return PT_OK;
    const size_t n_tiles = (size_t)((ctx->W + 7) / 8) * (size_t)((ctx->local_rows + 7) / 8);
    // every grid that can be in flight at once: k_render (one thread per pixel at most), or kWfDefaultChains concurrent wf_intersect
    // launches of the wavefront variant (2 cost classes x 6 x 256 threads per CU each, render_wavefront)
    const size_t lanes = 256 * std::max<size_t>((size_t)ctx->cu_count * 12 * kWfDefaultChains, (n_tiles + 3) / 4);
    PT_HIP(ctx, hipMalloc((void**)&ctx->d_stack_ovf, lanes * (size_t)extra * sizeof(uint32_t)));
    ctx->stack_ovf_lanes = lanes;
    return PT_OK;
}

int seed_upload(pt_context* ctx, const int32_t* global_seeds) {
    std::vector<int32_t> local((size_t)ctx->npix);
    for (int32_t lr = 0; lr < ctx->local_rows; ++lr) {
        const int32_t gr = global_row(ctx, lr);
        std::memcpy(&local[(size_t)lr * ctx->W], &global_seeds[(size_t)gr * ctx->W], sizeof(int32_t) * (size_t)ctx->W);
    }
    if (ctx->npix) PT_HIP(ctx, hipMemcpyAsync(ctx->d_rnds, local.data(), sizeof(int32_t) * local.size(), hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

void fill_params(const pt_context* ctx, const pt_camera* cam, RenderParams* p) {
    std::memset(p, 0, sizeof *p);
    p->nodes = ctx->d_nodes;
    p->tris = ctx->d_tris;
    p->meta = ctx->d_meta;
    p->mats = ctx->d_mats;
    p->rnds = ctx->d_rnds;
    p->colors = ctx->d_colors;
    p->rays = ctx->d_rays;
    p->stats = ctx->d_stats;
    p->cam = *cam;
    p->width = ctx->W;
    p->height = ctx->H;
    p->local_rows = ctx->local_rows;
    p->rank = ctx->rank;
    p->world = ctx->world;
    p->rows_per_block = ctx->rows_per_block;
    p->n_nodes = (int32_t)ctx->nodes.size();
    p->n_tris = (int32_t)ctx->orig.size();
    p->n_flat = ctx->n_flat;
    p->n_fbox = ctx->n_fbox;
    std::memcpy(p->fbox_rep, ctx->fbox_rep, sizeof p->fbox_rep);
    std::memcpy(p->fbox_mask, ctx->fbox_mask, sizeof p->fbox_mask);
    p->stack_entries = stack_entries_for(ctx->interior_depth);
    p->stack_ovf = nullptr;
    p->stack_ovf_lanes = 0;
    // where the traversal reads nodes from: the whole tree staged in LDS, its re-indexed top, or L1/L2 only
    p->node_mode = kNodesGlobal;
    p->treelet_nodes = 0;
    if (ctx->lds_scene) {
        if (whole_tree_fits_lds(ctx->nodes.size(), ctx->orig.size(), ctx->interior_depth, ctx->n_flat)) {
            p->node_mode = kNodesLds;
        } else if (ctx->treelet_nodes > 0) {
            p->node_mode = kNodesTreelet;
            p->treelet_nodes = ctx->treelet_nodes;
        }
    }
    if (!ctx->nodes4.empty() && (p->node_mode == kNodesGlobal || ctx->wide_nodes == 2)) {
        p->node_mode = kNodesWide;
        p->treelet_nodes = 0;
        p->nodes = ctx->d_nodes4;
        p->stack_entries = std::min(ctx->wide_lds_entries, wide_stack_entries(ctx->wide_pending));
        p->stack_ovf = ctx->d_stack_ovf;
        p->stack_ovf_lanes = (int32_t)ctx->stack_ovf_lanes;
    }
    p->tile_counter = nullptr;
    p->poll_ticks = (uint32_t)std::min<int64_t>((int64_t)ctx->poll_timeout_ms * 100000, 0xffffffffll);      // s_memrealtime: 100 MHz
    p->debug_stall_tile = ctx->debug_stall_tile;
    p->chunk_spp = 0;
    p->tile_done = nullptr;
    p->n_tiles = ((ctx->W + 7) / 8) * ((ctx->local_rows + 7) / 8);
    // measured with the big-triangle list in place (profiles/r02/i_*): Cornell box in LDS 8 / 16 / 24 / 32 -> 1669 / 1690 /
    // 1680 / 1669 Msamples/s (lockstep 1671); MESH-100k 24 / 32 / 48 -> 600 / 599 / 595 (lockstep 571); MESH-1M 211 / 210 / 205 (191)
    // (re-swept at the end of the round: Cornell box 8 / 12 / 16 / 20 / 24 / 32 -> 1957 / 2033 / 2066 / 2074 / 2078 / 2070)
    // (round 4, with the phase switching below: tree in LDS 12 / 16 / 20 / 24 / 32 -> 2637 / 2638 / 2621 / 2607 / 2554; from global
    // memory the rate is flat from 16 to 24: profiles/r04/)
    p->suspend_lanes = ctx->suspend_lanes >= 0 ? ctx->suspend_lanes : (p->node_mode == kNodesLds ? 16 : 24);
    // a phase of a while-while round ends early when at most this many lanes are still in it and some lane has left it
    // (Trav::round).  1080p, node_min / leaf_min (profiles/r04/c_*): tree in LDS (Cornell box) 0/0 2436, 3/4 2589, 4/8 2590, 8/4 2506
    // Msamples/s; 4-wide nodes from global memory 0/0 839 | 294, 4/4 1012 | 367, 6/4 1028 | 373, 8/8 1031 | 371 (MESH-100k | MESH-1M)
    p->node_min_lanes = ctx->node_min_lanes >= 0 ? ctx->node_min_lanes : (p->node_mode == kNodesLds ? 3 : 6);
    p->leaf_min_lanes = ctx->leaf_min_lanes >= 0 ? ctx->leaf_min_lanes : 4;
}

int check_ready(pt_context* ctx, const pt_camera* cam) {
    if (!cam) return fail(ctx, PT_EINVAL, "camera is NULL");
    if (!ctx->tris_uploaded) return fail(ctx, PT_EINVAL, "pt_upload_triangles has not been called");
    if (!ctx->mats_uploaded) return fail(ctx, PT_EINVAL, "pt_upload_materials has not been called");
    if ((int32_t)cam->XM != ctx->W || (int32_t)cam->YM != ctx->H)
        return fail(ctx, PT_EINVAL, "camera XM/YM do not match the context's frame size");
    return PT_OK;
}

int time_begin(pt_context* ctx, EventPair** ep) {
    *ep = nullptr;
    if (!ctx->timing) return PT_OK;
    if (ctx->events_used == ctx->events.size()) {
        EventPair e;
        PT_HIP(ctx, hipEventCreate(&e.a));
        PT_HIP(ctx, hipEventCreate(&e.b));
        ctx->events.push_back(e);
    }
    *ep = &ctx->events[ctx->events_used++];
    PT_HIP(ctx, hipEventRecord((*ep)->a, ctx->stream));
    return PT_OK;
}
int time_end(pt_context* ctx, EventPair* ep) {
    ctx->kernel_launches++;
    if (ep) PT_HIP(ctx, hipEventRecord(ep->b, ctx->stream));
    return PT_OK;
}
int time_collect(pt_context* ctx) {
    for (size_t i = 0; i < ctx->events_used; ++i) {
        float ms = 0.f;
        PT_HIP(ctx, hipEventSynchronize(ctx->events[i].b));
        PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->events[i].a, ctx->events[i].b));
        ctx->kernel_ms_acc += ms;
    }
    ctx->events_used = 0;
    return PT_OK;
}

}  // namespace

// =============================================================================== C ABI
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
    if ((e = hipMalloc((void**)&ctx->d_tile_counter, 64)) != hipSuccess) return bail("hipMalloc(tile counter)", e);
    if ((e = hipMemset(ctx->d_tile_counter, 0, 64)) != hipSuccess) return bail("hipMemset", e);      // zeroed ONCE: the last wave of a launch leaves it zero (k_render)
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

int pt_end_obj(pt_context* ctx) {
    if (!ctx) return PT_EINVAL;
    const int32_t n = (int32_t)ctx->tris.size() - ctx->tri_shift;
    if (n <= 0) return fail(ctx, PT_ESCENE, "end_Obj on an empty object (the reference reads tris[0] of an empty vector, main.cpp:216)");
    ctx->enc_rank.resize(ctx->tris.size(), -1);
    RefOrder ro{ctx->tris, ctx->enc_rank, ctx->next_rank};
    const unsigned hw = std::thread::hardware_concurrency();
    ro.run(ctx->tri_shift, n, ctx->build_threads > 0 ? ctx->build_threads : (int)std::min<unsigned>(hw ? hw : 1u, 16u));
    if (ro.degenerate) {
        ctx->tris.resize((size_t)ctx->tri_shift);
        ctx->enc_rank.resize((size_t)ctx->tri_shift);
        return fail(ctx, PT_ESCENE, "object has more than 6 triangles sharing one centroid: the reference's NodeOnHost::build (main.cpp:246-257) never terminates on it");
    }
    ctx->obj_begin.push_back(ctx->tri_shift);
    ctx->tri_shift = (int32_t)ctx->tris.size();
    ctx->tris_uploaded = false;
    return PT_OK;
}

// Device-built trees (bvh_policy 4): the LBVH splits by Morton code, which is good inside small clusters and poor at the top,
// where boxes overlap most.  The tree is cut into clusters of at most `cluster` triangles (maximal subtrees of the radix
// tree), and the top above the cut is rebuilt with the host's binned SAH over the cluster boxes -- a few thousand
// primitives, milliseconds -- and spliced onto the untouched cluster subtrees.  Leaves, packets and their order stay as the
// device emitted them.  Returns false (tree unchanged) when there is nothing to gain or the result would be too deep.
static bool sah_top_rebuild(pt_context* ctx, int cluster) {
    std::vector<Node64>& old = ctx->nodes;
    if (cluster <= 0 || old.size() < 64) return false;
    PhaseClock clk("sah top");
    // triangles below every node (post-order over an explicit stack; children are visited before their parent is closed)
    std::vector<int32_t> count(old.size(), 0);
    {
        std::vector<std::pair<int32_t, int>> st;
        st.emplace_back(0, 0);
        while (!st.empty()) {
            const int32_t i = st.back().first;
            const int phase = st.back().second;
            const Node64& nd = old[(size_t)i];
            if (phase == 0) {
                st.back().second = 1;
                if (nd.left >= 0) st.emplace_back(nd.left, 0);
                if (nd.right >= 0) st.emplace_back(nd.right, 0);
            } else {
                const int32_t cl = nd.left >= 0 ? count[(size_t)nd.left] : ((~nd.left) & 7) + 1;
                const int32_t cr = nd.right >= 0 ? count[(size_t)nd.right] : ((~nd.right) & 7) + 1;
                count[(size_t)i] = cl + cr;
                st.pop_back();
            }
        }
    }
    clk.lap("triangle counts");
    if (count[0] <= cluster * 4) return false;
    // the cut: children that are leaves or small enough become clusters
    struct Cluster { int32_t ref; Aabb box; };
    std::vector<Cluster> clusters;
    {
        std::vector<int32_t> st(1, 0);
        while (!st.empty()) {
            const int32_t i = st.back();
            st.pop_back();
            const Node64& nd = old[(size_t)i];
            for (int side = 0; side < 2; ++side) {
                const int32_t c = side ? nd.right : nd.left;
                Cluster cl;
                cl.ref = c;
                for (int a = 0; a < 3; ++a) { cl.box.lo[a] = nd.q[a][2 * side]; cl.box.hi[a] = nd.q[a][2 * side + 1]; }
                if (!(cl.box.lo[0] <= cl.box.hi[0] && cl.box.lo[1] <= cl.box.hi[1] && cl.box.lo[2] <= cl.box.hi[2])) continue;   // empty child
                if (c >= 0 && count[(size_t)c] > cluster) st.push_back(c);
                else clusters.push_back(cl);
            }
        }
    }
    clk.lap("cut");
    if (clusters.size() < 4) return false;
    BvhBuilder top;
    top.prims.resize(clusters.size());
    for (size_t k = 0; k < clusters.size(); ++k) {
        BuildPrim& p = top.prims[k];
        p.box = clusters[k].box;
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
        p.tri = (int32_t)k;
    }
    top.max_leaf = 1;            // one cluster per leaf of the top tree
    top.force_leaf = false;
    top.visit_cost = (float)ctx->sah_visit_cost * 0.1f;
    Aabb box;
    if (top.build(0, top.prims.size(), 0, &box) != 0) return false;
    clk.lap("SAH over the clusters");
    // splice: top nodes in preorder, every cluster's subtree copied right where the top tree refers to it
    std::vector<Node64> out;
    out.reserve(old.size() + top.nodes.size());
    struct Copy {
        const std::vector<Node64>& old;
        std::vector<Node64>& out;
        int32_t subtree(int32_t ref) {
            if (ref < 0) return ref;
            const int32_t me = (int32_t)out.size();
            out.push_back(old[(size_t)ref]);
            const int32_t l = subtree(old[(size_t)ref].left), r = subtree(old[(size_t)ref].right);
            out[(size_t)me].left = l;
            out[(size_t)me].right = r;
            return me;
        }
    } copy{old, out};
    struct Emit {
        const BvhBuilder& top;
        const std::vector<Cluster>& clusters;
        Copy& copy;
        std::vector<Node64>& out;
        int32_t child(int32_t ref) {
            if (ref >= 0) return node(ref);
            const int32_t pos = (~ref) >> 3;                              // a top leaf holds one primitive: the cluster at that position
            return copy.subtree(clusters[(size_t)top.prims[(size_t)pos].tri].ref);
        }
        int32_t node(int32_t t) {
            const int32_t me = (int32_t)out.size();
            out.push_back(top.nodes[(size_t)t]);
            const int32_t l = child(top.nodes[(size_t)t].left), r = child(top.nodes[(size_t)t].right);
            out[(size_t)me].left = l;
            out[(size_t)me].right = r;
            return me;
        }
    } emit{top, clusters, copy, out};
    emit.node(0);
    clk.lap("splice");
    if (deepest_interior_node(out) + 2 > kStackEntries) return false;
    clk.lap("depth check");
    old.swap(out);
    return true;
}

// bvh_policy 4: build the tree on the device (pt_lbvh.hip); host copies are kept for the debug getters
// The big-triangle list of select_flat_list(), from the half areas of ALL triangles (add order) and a callback for the
// bounds of everything but the candidates: the same choice, without the host builder's primitive array.
static int choose_flat_list(const pt_context* ctx, const std::vector<float>& area, const std::function<int(const std::vector<int32_t>&, Aabb*)>& rest_box,
                            std::vector<int32_t>* flat) {
    flat->clear();
    const size_t n = area.size();
    if (ctx->flat_list <= 0 || n == 0) return PT_OK;
    const int threads = host_threads(ctx);
    const size_t cand = std::min<size_t>((size_t)ctx->flat_list, n);
    auto bigger = [&](size_t a, size_t b) { return area[a] > area[b] || (area[a] == area[b] && a < b); };
    // the cand biggest in order (ties: add order): per chunk, then merged -- `bigger` is a total order, so any grouping agrees
    const size_t chunks = std::max<size_t>(1, std::min<size_t>((size_t)threads, n / 65536 + 1));
    std::vector<std::vector<size_t>> part(chunks);
    parallel_for(chunks, 1, threads, [&](size_t cb, size_t ce) {
        for (size_t c = cb; c < ce; ++c) {
            std::vector<size_t>& top = part[c];
            top.reserve(cand + 1);
            const size_t lo = n * c / chunks, hi = n * (c + 1) / chunks;
            for (size_t i = lo; i < hi; ++i) {
                if (top.size() == cand && !bigger(i, top.back())) continue;
                top.insert(std::upper_bound(top.begin(), top.end(), i, bigger), i);
                if (top.size() > cand) top.pop_back();
            }
        }
    });
    std::vector<size_t> top;
    for (const std::vector<size_t>& p : part) top.insert(top.end(), p.begin(), p.end());
    std::sort(top.begin(), top.end(), bigger);
    top.resize(cand);
    std::vector<int32_t> top32(cand);
    for (size_t k = 0; k < cand; ++k) top32[k] = (int32_t)top[k];
    Aabb tail;
    int rc = rest_box(top32, &tail);
    if (rc != PT_OK) return rc;
    std::vector<Aabb> rest(cand + 1);                 // rest[k] = box of top[k..] and all the others
    rest[cand] = tail;
    for (size_t k = cand; k-- > 0;) { tail.grow(padded_bounds(ctx->tris[top[k]])); rest[k] = tail; }
    for (size_t m = cand; m > 0; --m) {               // the largest m such that each of the m biggest is >= 1/16 of the box around all the others
        const float smallest = area[top[m - 1]], others = rest[m].half_area();
        if (smallest >= others * (1.0f / 16.0f)) {
            flat->assign(top32.begin(), top32.begin() + (std::ptrdiff_t)m);
            std::sort(flat->begin(), flat->end());    // add order
            break;
        }
    }
    return PT_OK;
}

static int build_on_device(pt_context* ctx, bool* done) {
    *done = false;
    PhaseClock clk("pt_upload_triangles/device");
    const int n = (int)ctx->tris.size();
    if (!ctx->has_device || n <= 2 * kMaxLeaf) return PT_OK;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    // The triangles go to the device as they are; the areas of their padded bounds come back for the big-triangle list,
    // which is chosen on the host (<= 32 entries) from bounds the device reduces.
    DeviceStage st;
    struct StageGuard {
        DeviceStage* s;
        ~StageGuard() { stage_free(s); }
    } guard{&st};
    std::vector<float> area((size_t)n);
    int nonfinite = 0;
    PT_HIP(ctx, stage_upload(ctx->tris.data(), ctx->enc_rank.data(), n, ctx->stream, &st, area.data(), &nonfinite));
    if (nonfinite) return PT_OK;                // the host path handles those
    clk.lap("upload + areas");
    std::vector<int32_t> flat;
    int frc = choose_flat_list(ctx, area, [&](const std::vector<int32_t>& top, Aabb* box) {
        float b[6];
        PT_HIP(ctx, stage_rest_box(st, top.data(), (int)top.size(), ctx->stream, b));
        for (int a = 0; a < 3; ++a) { box->lo[a] = b[a]; box->hi[a] = b[3 + a]; }
        return (int)PT_OK;
    }, &flat);
    if (frc != PT_OK) return frc;
    std::vector<float>().swap(area);
    const int nf = (int)flat.size(), ns = n - nf;
    if (ns <= 2 * kMaxLeaf) return PT_OK;
    if (nf > 0) PT_HIP(ctx, stage_select(st, flat.data(), nf, ctx->stream));
    const int32_t* d_sel = nf > 0 ? st.d_sel : nullptr;
    const int threads = host_threads(ctx);
    clk.lap("big-triangle list");
    LbvhResult r;
    const bool sah = ctx->bvh_policy != 4;
    if (sah) {                                  // the host builder's tree, node for node; the host builds what the device cannot
        bool unsupported = false;
        const int tp = tree_policy(ctx);
        const bool forced = tp == 2 || tp == 3;            // (as build_and_pack reads the policy)
        PT_HIP(ctx, sah_device_build(st.d_tris, st.d_rank, n, d_sel, ns, tp == 3 ? 8 : 4, forced, (float)ctx->sah_visit_cost * 0.1f, ctx->sah_grain, ctx->stream, &r,
                                     &unsupported));
        if (unsupported) return PT_OK;
        clk.lap("sah_device_build");
    } else {
        PT_HIP(ctx, lbvh_build(st.d_tris, st.d_rank, n, d_sel, ns, ctx->lbvh_ploc, ctx->stream, &r));
        clk.lap("lbvh_build");
    }
    stage_free(&st);
    auto drop = [&]() { (void)hipFree(r.d_nodes); (void)hipFree(r.d_tris); (void)hipFree(r.d_meta); (void)hipFree(r.d_orig); };
    if (!sah && r.depth + 5 > kStackEntries) {  // deeper than the traversal stack: let the host builder do it
        drop();
        return PT_OK;
    }
    // The device arrays are final but for the list's nf slots in front, which are written here (leaf references already count
    // from behind them).  The host keeps the nodes (SAH top, 4-wide collapse, debug getters) and the order; packets and meta
    // of the tree's triangles stay on the device until a debug getter asks for them (host_packets_stale).
    ctx->nodes.resize((size_t)r.n_nodes);
    ctx->orig.resize((size_t)n);
    ctx->packets.assign((size_t)std::max(nf, 1), TriPacket());
    ctx->meta.assign((size_t)std::max(nf, 1), TriMeta());
    hipError_t e = hipMemcpy(ctx->nodes.data(), r.d_nodes, sizeof(Node64) * (size_t)r.n_nodes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ctx->orig.data() + nf, reinterpret_cast<int32_t*>(r.d_orig) + nf, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost);
    for (int k = 0; k < nf; ++k) {
        const pt_triangle& t = ctx->tris[(size_t)flat[(size_t)k]];
        float* v = ctx->packets[(size_t)k].v;
        v[0] = t.r1.s[0]; v[1] = t.r1.s[1]; v[2] = t.r1.s[2];
        v[3] = t.r2.s[0]; v[4] = t.r2.s[1]; v[5] = t.r2.s[2];
        v[6] = t.r3.s[0]; v[7] = t.r3.s[1]; v[8] = t.r3.s[2];
        v[9] = t.N.s[0]; v[10] = t.N.s[1]; v[11] = t.N.s[2];
        ctx->meta[(size_t)k].rank = ctx->enc_rank[(size_t)flat[(size_t)k]];
        ctx->meta[(size_t)k].mati = t.mati;
        ctx->orig[(size_t)k] = flat[(size_t)k];
    }
    if (e == hipSuccess && nf > 0) e = hipMemcpy(r.d_tris, ctx->packets.data(), sizeof(TriPacket) * (size_t)nf, hipMemcpyHostToDevice);
    if (e == hipSuccess && nf > 0) e = hipMemcpy(r.d_meta, ctx->meta.data(), sizeof(TriMeta) * (size_t)nf, hipMemcpyHostToDevice);
    if (e != hipSuccess) { drop(); return fail(ctx, PT_EHIP, std::string("device BVH download: ") + hipGetErrorString(e)); }
    if (sah) {
        // The SAH tree arrives in preorder: a child's index is larger than its parent's and every leaf lies inside the packed
        // triangles behind the list.  Checked before any host code walks the tree (a malformed tree must fail here, loudly,
        // not loop there).
        std::atomic<bool> ok(true);
        const int32_t nn = (int32_t)r.n_nodes;
        parallel_for((size_t)nn, 1 << 15, threads, [&](size_t b, size_t e2) {
            bool good = true;
            for (size_t i = b; i < e2 && good; ++i) {
                const int32_t refs[2] = {ctx->nodes[i].left, ctx->nodes[i].right};
                for (int32_t ref : refs) {
                    if (ref >= 0) good = good && ref > (int32_t)i && ref < nn;
                    else {
                        const int32_t first = (~ref) >> 3, count = ((~ref) & 7) + 1;
                        good = good && first >= nf && first + count <= n;
                    }
                }
            }
            if (!good) ok.store(false);
        });
        if (!ok.load()) { drop(); return fail(ctx, PT_EHIP, "internal: the device SAH builder returned a malformed tree"); }
    }
    (void)hipFree(r.d_orig);
    r.d_orig = nullptr;
    ctx->host_packets_stale = true;
    ctx->bvh_depth = sah ? r.depth : r.depth + 1;
    ctx->n_flat = nf;
    clk.lap("download + list in front");
    const bool retopped = !sah && sah_top_rebuild(ctx, ctx->lbvh_cluster);
    clk.lap("SAH top over clusters");
    bool wide_done = false;
    int rc = plan_node_placement(ctx, retopped ? nullptr : r.d_nodes, &wide_done);
    clk.lap("node placement + 4-wide nodes");
    if (rc != PT_OK) { drop(); return rc; }
    if (retopped) ctx->bvh_depth = ctx->interior_depth + 1;
    if (ctx->d_tris) (void)hipFree(ctx->d_tris);
    if (ctx->d_meta) (void)hipFree(ctx->d_meta);
    ctx->d_tris = r.d_tris;
    ctx->d_meta = r.d_meta;
    if (retopped) {               // the nodes were recomposed on the host
        (void)hipFree(r.d_nodes);
        if ((rc = upload_vec(ctx, &ctx->d_nodes, ctx->nodes.data(), sizeof(Node64) * ctx->nodes.size())) != PT_OK) return rc;
    } else {
        if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
        ctx->d_nodes = r.d_nodes;
        if (ctx->treelet_nodes > 0) PT_HIP(ctx, hipMemcpy(ctx->d_nodes, ctx->nodes.data(), sizeof(Node64) * ctx->nodes.size(), hipMemcpyHostToDevice));
    }
    if (!wide_done && (rc = upload_vec(ctx, &ctx->d_nodes4, ctx->nodes4.data(), sizeof(Node4q) * ctx->nodes4.size())) != PT_OK) return rc;
    if ((rc = alloc_stack_overflow(ctx)) != PT_OK) return rc;
    clk.lap("uploads");
    *done = true;
    return PT_OK;
}

static void compute_cost_boxes(pt_context* ctx) { compute_cost_boxes_impl(ctx); }   // defined in the anonymous namespace above

int pt_upload_triangles(pt_context* ctx) {
    if (!ctx) return PT_EINVAL;
    if (ctx->tri_shift != (int32_t)ctx->tris.size())
        return fail(ctx, PT_EINVAL, "triangles were added after the last end_Obj; close the object first (main.cpp:536)");
    const auto t0 = std::chrono::steady_clock::now();
    // Policies 0..3 name a TREE (binned SAH, leaf rule); where it is built does not change it: on the device (pt_sahdev.hip, the
    // same nodes) for scenes big enough to repay the launches, on the host otherwise and for whatever the device hands back.
    const bool on_device = ctx->has_device && (ctx->bvh_policy >= 4 || ctx->bvh_device == 1 || (ctx->bvh_device < 0 && (int64_t)ctx->tris.size() >= kDeviceBuildFrom));
    if (on_device) {
        bool done = false;
        int rcd = build_on_device(ctx, &done);
        if (rcd != PT_OK) return rcd;
        if (done) {
            compute_cost_boxes(ctx);
            ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            ctx->bvh_on_device = 1;
            ctx->tris_uploaded = true;
            return PT_OK;
        }
    }
    ctx->bvh_on_device = 0;
    ctx->host_packets_stale = false;
    int rc = build_and_pack(ctx);
    if (rc != PT_OK) return rc;
    PhaseClock clk("pt_upload_triangles");
    if ((rc = plan_node_placement(ctx)) != PT_OK) return rc;
    clk.lap("node placement + 4-wide nodes");
    ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ctx->has_device) {
        PT_HIP(ctx, hipSetDevice(ctx->device));
        if ((rc = upload_vec(ctx, &ctx->d_nodes, ctx->nodes.data(), sizeof(Node64) * ctx->nodes.size())) != PT_OK) return rc;
        if ((rc = upload_vec(ctx, &ctx->d_nodes4, ctx->nodes4.data(), sizeof(Node4q) * ctx->nodes4.size())) != PT_OK) return rc;
        if ((rc = alloc_stack_overflow(ctx)) != PT_OK) return rc;
        if ((rc = upload_vec(ctx, &ctx->d_tris, ctx->packets.data(), sizeof(TriPacket) * ctx->packets.size())) != PT_OK) return rc;
        if ((rc = upload_vec(ctx, &ctx->d_meta, ctx->meta.data(), sizeof(TriMeta) * ctx->meta.size())) != PT_OK) return rc;
        clk.lap("device allocations + copies");
    }
    ctx->tris_uploaded = true;
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

// The work counter of the persistent launches (d_tile_counter) is zeroed once, at pt_create: every launch that runs to its end leaves
// words 0 / 1 at zero (the last wave out resets them).  INVARIANT: a launch that did NOT run to its end -- a failed launch call, a
// failed synchronize, a lost hand-over -- sets counters_suspect, and the next launch clears the counter first.
static int prepare_work_counter(pt_context* ctx) {
    if (ctx->counters_suspect) {
        PT_HIP(ctx, hipMemsetAsync(ctx->d_tile_counter, 0, 64, ctx->stream));
        ctx->counters_suspect = false;
    }
    ctx->launched_since_check = true;
    return PT_OK;
}

static int ptamd_resident_waves(const pt_context*, const LaunchConfig& lc) { return lc.persistent_blocks * (lc.block / 64); }

// Samples per (pass, tile) work item of a persistent launch, by tiles per resident wave (0: whole tiles).
// fewer tiles per resident wave (1080p over 2 / 4 / 8 GPUs: 4.0 / 2.0 / 1.0): suspend with passes of 16, lockstep with
// passes of 8, lockstep with whole tiles (profiles/r02/q_*: 95.7 % / 87.5 % / 61.8 % of the one-GPU rate per GPU)
static int auto_chunk_spp(int n_tiles, int resident_waves, int nsamples) {
    return n_tiles >= 5 * resident_waves ? (nsamples >= 256 ? 64 : 32)
         : n_tiles >= 3 * resident_waves ? 16
         : n_tiles > resident_waves + resident_waves / 4 ? 8 : 0;
}

static void launch_cfg(pt_context* ctx, const RenderParams& p, LaunchConfig* lc) {
    // whole tree in LDS: two 768-thread workgroups per CU (six waves per SIMD) if their LDS fits, else two of 512
    // -- and if the launch has a tile for each of their waves: with fewer (a 1080p frame over 8 GPUs: 4,050 tiles for
    // 6,144 waves) the extra waves stay empty and the 128-VGPR instance runs each tile faster (profiles/r03/g_*)
    const bool wide_fits = p.node_mode == kNodesLds && 2 * (traversal_lds_bytes(p, kLdsBlockWide) + 512) <= kLdsPerCu;
    const bool wide_block = wide_fits && ctx->lds_block != kLdsBlockBase &&
                            (ctx->lds_block == kLdsBlockWide || p.n_tiles >= ctx->cu_count * 2 * (kLdsBlockWide / 64));
    lc->block = traversal_block(p.node_mode, wide_block);
    lc->lds_bytes = traversal_lds_bytes(p, lc->block);
    lc->count_work = ctx->count_work != 0;
    // Restart + tail suspension wins when a wave works through many tiles (one GPU, 1080p: 7.9 per resident wave:
    // Cornell +2.4 %, mesh scenes +13-18 %); with few tiles per wave the end of a tile -- its slowest pixels finishing
    // their last samples alone -- is on the critical path and lockstep, whose lanes finish together, wins clearly
    // (1080p over 4 / 8 ranks: 87 % / 62 % strong-scaling efficiency against 73 % / 42 %; over 2 ranks suspend wins again:
    // 95.7 % against 88 %; profiles/r02/e_*, q_*).
    // resident workgroups at 4 waves per SIMD: 2 x 512 threads (whole tree in LDS), 1 x 1024 (treelet) per CU; nodes through
    // L1/L2 (256 threads): as many waves per SIMD -- 7, 6, 5 or 4 -- as the stacks in LDS leave room for
    lc->waves_per_simd = wide_block ? kLdsWpsWide : 4;
    if (p.node_mode == kNodesGlobal || p.node_mode == kNodesWide) {
        const int want = ctx->waves_per_simd > 0 ? ctx->waves_per_simd : 7;
        for (int w = std::min(want, 8); w > 4; --w)
            if ((size_t)w * lc->lds_bytes + 1024 <= kLdsPerCu) { lc->waves_per_simd = w; break; }
    }
    lc->persistent_blocks = ctx->cu_count * std::max(1, 256 * lc->waves_per_simd / lc->block);
    lc->cu_count = ctx->cu_count;
    // ... and with few samples per launch: a lane has no next sample to start while the others finish, and lanes that run
    // ahead give up the coherence of a tile's camera rays -- lockstep up to 4 samples per launch with the tree in LDS
    // (render(1): 1,810 against 1,492 Msamples/s), for one sample otherwise (profiles/r03/q_*)
    const bool few_samples = p.nsamples <= (p.node_mode == kNodesLds ? 4 : 1);
    lc->schedule = ctx->schedule >= 0 ? ctx->schedule : (!few_samples && p.n_tiles >= 3 * ptamd_resident_waves(ctx, *lc) ? 1 : 0);
    ctx->last_lds_bytes = lc->lds_bytes;
    ctx->last_waves_per_simd = lc->waves_per_simd;
}

int pt_generate_rays(pt_context* ctx, const pt_camera* cam) {
    PT_NEED_DEVICE(ctx);
    int rc = check_ready(ctx, cam);
    if (rc != PT_OK) return rc;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    RenderParams p;
    fill_params(ctx, cam, &p);
    PT_HIP(ctx, launch_gen_ray(p, ctx->stream));
    return PT_OK;
}

int pt_trace_rays(pt_context* ctx, const pt_camera* cam, int32_t iterations, int32_t current_sample) {
    PT_NEED_DEVICE(ctx);
    int rc = check_ready(ctx, cam);
    if (rc != PT_OK) return rc;
    if (iterations < 0 || current_sample < 0) return fail(ctx, PT_EINVAL, "iterations/current_sample must be >= 0");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    RenderParams p;
    fill_params(ctx, cam, &p);
    p.iterations = iterations;
    p.first_sample = current_sample;
    p.nsamples = 1;
    LaunchConfig lc;
    launch_cfg(ctx, p, &lc);
    if (ctx->persistent) {      // the grid only fills the chip: a workgroup stages the tree once, not once per eight tiles
        if (int rc2 = prepare_work_counter(ctx)) return rc2;
        p.tile_counter = ctx->d_tile_counter;      // (zero: the previous launch's last wave reset it)
    }
    EventPair* ep;
    if ((rc = time_begin(ctx, &ep)) != PT_OK) return rc;
    {
        const hipError_t le = launch_trace_ray(p, lc, ctx->stream);
        if (le != hipSuccess) { ctx->counters_suspect = true; return fail(ctx, PT_EHIP, std::string("launch_trace_ray: ") + hipGetErrorString(le)); }
    }
    ctx->render_epoch++;
    rc = time_end(ctx, ep);
    if (rc != PT_OK) ctx->counters_suspect = true;
    return rc;
}

// The stream-compacted variant.  The local pixels are cut into `wf_streams` contiguous chains; every chain owns its ray streams,
// hit stream, class queues and counters and runs its passes -- wf_generate, then per bounce wf_intersect and wf_shade -- on a HIP
// stream of its own.  A pass is a chain of 17 dependent launches of ~100 us, and a persistent wf_intersect launch ends on its
// longest ray: ~60 us of the 133 us a 1080p launch took were the same at a quarter and at four times the rays
// (profiles/r04/).  With two chains the tail of one runs under the body of the other.
static int render_wavefront(pt_context* ctx, const RenderParams& rp, int32_t nsamples) {
    if (rp.iterations > kWfMaxBounces) return fail(ctx, PT_EINVAL, "wavefront variant supports at most 1023 iterations");
    const size_t np = (size_t)std::max<int64_t>(ctx->npix, 1);
    constexpr size_t kCounterWords = (size_t)kWfCounterStride * (kWfMaxBounces + 4);
    if (!ctx->d_wf_state) {
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_wf_state, sizeof(float4) * 17 * np));
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_wf_queues, sizeof(int32_t) * 3 * np));
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_wf_counters, sizeof(uint32_t) * kCounterWords * kWfMaxChains));
    }
    if (ctx->npix == 0) return PT_OK;
    // chains of whole 8,192-pixel units, at least ~64k pixels each (a chain of a few thousand rays is all launch overhead)
    const int want = ctx->wf_streams > 0 ? ctx->wf_streams : kWfDefaultChains;
    int chains = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(want, kWfMaxChains), ctx->npix / 65536));
    // 4-wide traversal: the part of a lane's stack that lies in global memory is addressed by the lane's place in the GRID; chains
    // run concurrently, so each gets its own range of that buffer (the widest wf_intersect grid: 2 cost classes x 6 x 256 threads per CU)
    const size_t chain_lanes = (size_t)2 * 6 * 256 * (size_t)ctx->cu_count;
    if (rp.stack_ovf) chains = (int)std::max<size_t>(1, std::min<size_t>((size_t)chains, (size_t)rp.stack_ovf_lanes / chain_lanes));
    for (int c = 1; c < chains; ++c)
        if (!ctx->wf_stream[c]) {
            PT_HIP(ctx, hipStreamCreateWithFlags(&ctx->wf_stream[c], hipStreamNonBlocking));
            PT_HIP(ctx, hipEventCreateWithFlags(&ctx->wf_event[c], hipEventDisableTiming));
        }
    if (!ctx->wf_event[0]) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->wf_event[0], hipEventDisableTiming));
    WfParams w[kWfMaxChains];
    const int64_t unit = 8192;
    const int64_t per = ((ctx->npix + chains - 1) / chains + unit - 1) / unit * unit;
    for (int c = 0; c < chains; ++c) {
        WfParams& wc = w[c];
        const int64_t p0 = std::min<int64_t>((int64_t)c * per, ctx->npix), p1 = std::min<int64_t>(p0 + per, ctx->npix);
        wc.rp = rp;
        if (rp.stack_ovf) wc.rp.stack_ovf = rp.stack_ovf + (size_t)c * chain_lanes;      // (stack_ovf_lanes stays the stride between entries)
        wc.sP = reinterpret_cast<float*>(ctx->d_wf_state);        // kWfFields x np x 12 B <= 4 x np x 16 B, indexed by local pixel
        for (int par = 0; par < 2; ++par)
            for (int k = 0; k < 2; ++k) {
                wc.rsA[par][k] = ctx->d_wf_state + (size_t)(4 + (par * 2 + k) * 2 + 0) * np + p0;      // a chain's streams: its slice of each array
                wc.rsB[par][k] = ctx->d_wf_state + (size_t)(4 + (par * 2 + k) * 2 + 1) * np + p0;
                wc.rsC[par][k] = ctx->d_wf_state + (size_t)(13 + par * 2 + k) * np + p0;
            }
        wc.hit[0] = reinterpret_cast<float2*>(ctx->d_wf_state + 12 * np) + p0;
        wc.hit[1] = reinterpret_cast<float2*>(ctx->d_wf_state + 12 * np) + np + p0;
        for (int k = 0; k < 3; ++k) wc.q_cls[k] = ctx->d_wf_queues + (size_t)k * np + p0;
        wc.counters = ctx->d_wf_counters + (size_t)c * kCounterWords;
        wc.npix = (int32_t)(p1 - p0);
        wc.pix0 = (int32_t)p0;
        wc.npix_all = (int32_t)ctx->npix;
        wc.n_cbox = ctx->cost_binning ? (int32_t)(ctx->cost_boxes.size() / 6) : 0;
        for (int b = 0; b < wc.n_cbox; ++b)
            for (int k = 0; k < 6; ++k) wc.cbox[b][k] = ctx->cost_boxes[(size_t)b * 6 + k];
    }
    // the other chains' streams start behind whatever the context's stream holds, and the context's stream ends behind them
    PT_HIP(ctx, hipEventRecord(ctx->wf_event[0], ctx->stream));
    for (int c = 1; c < chains; ++c) PT_HIP(ctx, hipStreamWaitEvent(ctx->wf_stream[c], ctx->wf_event[0], 0));
    for (int32_t k = 0; k < nsamples; ++k) {
        for (int c = 0; c < chains; ++c) {
            if (w[c].npix == 0) continue;
            hipStream_t st = c == 0 ? ctx->stream : ctx->wf_stream[c];
            w[c].sample = rp.first_sample + k;
            PT_HIP(ctx, hipMemsetAsync(w[c].counters, 0, sizeof(uint32_t) * kWfCounterStride, st));
            PT_HIP(ctx, launch_wf_generate(w[c], st));
        }
        for (int32_t b = 0; b < rp.iterations; ++b)
            for (int c = 0; c < chains; ++c) {
                if (w[c].npix == 0) continue;
                hipStream_t st = c == 0 ? ctx->stream : ctx->wf_stream[c];
                EventPair* ep = nullptr;
                int rc = PT_OK;
                if (c == 0 && (rc = time_begin(ctx, &ep)) != PT_OK) return rc;      // kernel_ms: wf_intersect of chain 0 (the others overlap it)
                PT_HIP(ctx, launch_wf_intersect(w[c], b, ctx->cu_count, st));
                if (c == 0 && (rc = time_end(ctx, ep)) != PT_OK) return rc;
                PT_HIP(ctx, launch_wf_shade(w[c], b, st));
            }
        if (ctx->timing && ctx->events_used >= 4096) {   // bound the event pool
            int rc = time_collect(ctx);
            if (rc != PT_OK) return rc;
        }
    }
    for (int c = 1; c < chains; ++c) {
        PT_HIP(ctx, hipEventRecord(ctx->wf_event[c], ctx->wf_stream[c]));
        PT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->wf_event[c], 0));
    }
    return PT_OK;
}

int pt_render(pt_context* ctx, const pt_camera* cam, int32_t iterations, int32_t nsamples) {
    PT_NEED_DEVICE(ctx);
    int rc = check_ready(ctx, cam);
    if (rc != PT_OK) return rc;
    if (iterations < 0 || nsamples < 0) return fail(ctx, PT_EINVAL, "iterations/nsamples must be >= 0");
    if (nsamples == 0) return PT_OK;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    RenderParams p;
    fill_params(ctx, cam, &p);
    p.iterations = iterations;
    p.first_sample = ctx->current_sample;
    p.nsamples = nsamples;
    ctx->render_epoch++;
    if (ctx->variant == 1) {
        if ((rc = render_wavefront(ctx, p, nsamples)) != PT_OK) return rc;
        ctx->current_sample += nsamples;
        return PT_OK;
    }
    LaunchConfig lc;
    launch_cfg(ctx, p, &lc);
    if (ctx->persistent) {
        if (int rc2 = prepare_work_counter(ctx)) return rc2;
        p.tile_counter = ctx->d_tile_counter;      // (zero: the previous launch's last wave reset it)
        // automatic: chaining passes only pays when there are enough tiles to re-balance; with about one
        // tile per resident wave (a 1080p frame over 8 GPUs) the most expensive tile is the critical path
        // either way and the extra hand-offs cost 2-4 %.  With many tiles per wave (one GPU, 1080p: 7.9) the
        // launch runs the suspend schedule, whose items end with their slowest pixels finishing alone: long
        // passes amortise that tail, and half a launch's samples still re-balance the chip -- 64 spp per launch:
        // 8 / 16 / 32 / 64 -> 1689 / 1760 / 1787 / 1761 Msamples/s; 256 spp: 1706 / 1789 / 1843 / 1868
        // (profiles/r02/n_*; the lockstep kernel of round 1 peaked at 8).
        const int resident_waves = ptamd_resident_waves(ctx, lc);
        const int auto_chunk = auto_chunk_spp(p.n_tiles, resident_waves, nsamples);
        const int chunk = ctx->chunk_spp >= 0 ? ctx->chunk_spp : auto_chunk;
        // work items are numbered with an int on the device: passes x tiles + one failed fetch per resident wave
        const int64_t items = chunk > 0 ? ((int64_t)nsamples + chunk - 1) / chunk * p.n_tiles : p.n_tiles;
        if (items + (int64_t)resident_waves + 64 >= ((int64_t)1 << 31))
            return fail(ctx, PT_EINVAL, "nsamples / chunk_spp x tiles does not fit the 31-bit work-item counter of one launch: render in several calls");
        if (chunk > 0 && nsamples > chunk && p.n_tiles > 0) {
            if (!ctx->d_tile_done) PT_HIP(ctx, hipMalloc((void**)&ctx->d_tile_done, sizeof(uint32_t) * (size_t)p.n_tiles));
            PT_HIP(ctx, hipMemsetAsync(ctx->d_tile_done, 0, sizeof(uint32_t) * (size_t)p.n_tiles, ctx->stream));
            p.tile_done = ctx->d_tile_done;
            p.chunk_spp = chunk;
        }
    }
    if (lc.count_work && p.n_tiles > 0) {      // per-tile cost of this launch (pt_debug_tile_cost)
        if (!ctx->d_tile_cost) PT_HIP(ctx, hipMalloc((void**)&ctx->d_tile_cost, sizeof(uint32_t) * (size_t)p.n_tiles));
        PT_HIP(ctx, hipMemsetAsync(ctx->d_tile_cost, 0, sizeof(uint32_t) * (size_t)p.n_tiles, ctx->stream));
        p.tile_cost = ctx->d_tile_cost;
    }
    EventPair* ep;
    if ((rc = time_begin(ctx, &ep)) != PT_OK) return rc;
    {
        const hipError_t le = launch_render_mega(p, lc, ctx->stream);
        if (le != hipSuccess) { ctx->counters_suspect = true; return fail(ctx, PT_EHIP, std::string("launch_render_mega: ") + hipGetErrorString(le)); }
    }
    if ((rc = time_end(ctx, ep)) != PT_OK) { ctx->counters_suspect = true; return rc; }
    ctx->current_sample += nsamples;  // main.cpp:686
    return PT_OK;
}

int pt_set_current_sample(pt_context* ctx, int32_t s) {
    if (!ctx || s < 0) return PT_EINVAL;
    ctx->current_sample = s;
    return PT_OK;
}
int pt_get_current_sample(const pt_context* ctx, int32_t* out) {
    if (!ctx || !out) return PT_EINVAL;
    *out = ctx->current_sample;
    return PT_OK;
}

// Wait for the context's stream, then look at what the kernels left behind: a launch that lost a hand-over between chained passes
// (k_render) has written the tile into the work counter's error word.  That, like any failed HIP call on the way, becomes PT_EHIP --
// and the work counter is cleared before the next launch (a launch that did not run to its end does not leave words 0 / 1 at zero).
static int sync_and_check(pt_context* ctx) {
    PT_HIP(ctx, hipSetDevice(ctx->device));
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->counters_suspect = true;
        return fail(ctx, PT_EHIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    if (!ctx->launched_since_check || !ctx->d_tile_counter) return PT_OK;
    ctx->launched_since_check = false;
    uint32_t words[8] = {};
    PT_HIP(ctx, hipMemcpy(words, ctx->d_tile_counter, sizeof words, hipMemcpyDeviceToHost));
    if (words[kTileCounterError] != 0) {
        ctx->counters_suspect = true;
        PT_HIP(ctx, hipMemset(ctx->d_tile_counter, 0, 64));
        ctx->counters_suspect = false;
        char msg[256];
        std::snprintf(msg, sizeof msg, "k_render: pass %u of tile %u waited more than %d ms for the tile's previous pass to be handed over (the "
                      "wave rendering it was lost); the launch was wound down, the frame is incomplete", words[kTileCounterError + 1],
                      words[kTileCounterError] - 1, ctx->poll_timeout_ms);
        return fail(ctx, PT_EHIP, msg);
    }
    return PT_OK;
}

int pt_sync(pt_context* ctx) {
    PT_NEED_DEVICE(ctx);
    return sync_and_check(ctx);
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
    } else if (k == "chunk_spp") {
        if (value < -1 || value > 1 << 20) return fail(ctx, PT_EINVAL, "chunk_spp out of range");
        ctx->chunk_spp = (int)value;
    } else if (k == "persistent") {
        ctx->persistent = value ? 1 : 0;
    } else if (k == "sah_visit_cost") {
        if (value < 0 || value > 1000) return fail(ctx, PT_EINVAL, "sah_visit_cost: tenths of a triangle test, 0..1000");
        ctx->sah_visit_cost = (int)value;
    } else if (k == "schedule") {
        if (value < -1 || value > 1) return fail(ctx, PT_EINVAL, "schedule: -1 automatic, 0 lockstep per sample, 1 restart + tail suspension");
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
        k == "wave_shade_steps" || k == "wave_trips" || k == "wave_rounds") {
        std::vector<unsigned long long> rows((size_t)kStatCols * kStatRows);
        unsigned long long h[kStatCols] = {};
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PT_HIP(ctx, hipMemcpy(rows.data(), ctx->d_stats, sizeof(unsigned long long) * rows.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < kStatRows; ++r)
            for (int c = 0; c < kStatCols; ++c) h[c] += rows[(size_t)r * kStatCols + c];
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

// counting launches (option count_work): per 8x8 tile of the local frame, the shader-clock cycles / 64 its wave(s) spent on it,
// summed over the work items of the last pt_render call -- the latency a launch with one tile per wave ends on
// What pt_render(nsamples) would launch on a device of `cu_count` compute units (0: the context's own): works on a host-only
// context too, so that the launch policy -- which kernel instance, which schedule, which pass length a rank of an N-GPU job gets
// -- can be pinned by CPU tests.  out[8] = { threads per workgroup, waves per SIMD, schedule (0 lockstep, 1 suspend), samples per
// (pass, tile) work item (0: whole tiles), resident waves, tiles, node mode, dynamic LDS bytes }.
int pt_debug_launch_plan(pt_context* ctx, int32_t nsamples, int32_t cu_count, int64_t out[8]) {
    if (!ctx || !out || nsamples < 1 || cu_count < 0) return PT_EINVAL;
    if (!ctx->tris_uploaded) return fail(ctx, PT_EINVAL, "upload_Triangles first");
    pt_camera cam;
    const float shift[3] = {0.f, 0.f, 0.f};
    pt_camera_init(&cam, 60.0f, 0.0f, 0.0f, shift, ctx->W, ctx->H);
    RenderParams p;
    fill_params(ctx, &cam, &p);
    p.nsamples = nsamples;
    const int saved_cu = ctx->cu_count;
    const size_t saved_lds = ctx->last_lds_bytes;
    const int saved_wps = ctx->last_waves_per_simd;
    if (cu_count > 0) ctx->cu_count = cu_count;
    LaunchConfig lc;
    launch_cfg(ctx, p, &lc);
    const int resident = ptamd_resident_waves(ctx, lc);
    ctx->cu_count = saved_cu;
    ctx->last_lds_bytes = saved_lds;
    ctx->last_waves_per_simd = saved_wps;
    const int chunk = ctx->chunk_spp >= 0 ? ctx->chunk_spp : auto_chunk_spp(p.n_tiles, resident, nsamples);
    out[0] = lc.block; out[1] = lc.waves_per_simd; out[2] = lc.schedule; out[3] = (chunk > 0 && nsamples > chunk) ? chunk : 0;
    out[4] = resident; out[5] = p.n_tiles; out[6] = p.node_mode; out[7] = (int64_t)lc.lds_bytes;
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
