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
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>

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
    std::vector<TriPacket> packets;
    std::vector<TriMeta> meta;
    std::vector<int32_t> orig;
    int bvh_depth = 0;

    // ---- device buffers
    float4* d_nodes = nullptr;
    float4* d_tris = nullptr;
    TriMeta* d_meta = nullptr;
    pt_material* d_mats = nullptr;
    int32_t* d_rnds = nullptr;
    float4* d_colors = nullptr;
    pt_ray* d_rays = nullptr;
    float4* d_ldr = nullptr;
    unsigned long long* d_stats = nullptr;
    // wavefront variant: path state + queues (allocated on first use)
    float4* d_wf_state = nullptr;   // 4 (state) + 8 (ray streams) x npix float4, + 2 x npix float2 (hits)
    int32_t* d_wf_queues = nullptr; // 3 x npix int32 (class queues)
    std::vector<float> cost_boxes;  // 6 floats per complex object (wavefront cost classes)
    uint32_t* d_wf_counters = nullptr;
    bool own_rnds = true, own_colors = true;
    hipStream_t stream = nullptr;

    int32_t current_sample = 0;  // main.cpp:28

    // ---- options
    int variant = 0;
    int block = 256;
    int lds_scene = 2;   // 2: stage the BVH nodes in LDS when they fit next to two 512-thread blocks per CU (+5 %
                         // measured); 1: nodes and packets (costs occupancy and needs a fat-leaf tree: slower); 0: off
    int timing = 0;
    int count_work = 0;
    int traversal = 0;    // 0 while-while, 1 voting
    int bvh_on_device = 0;
    double bvh_build_ms = 0.0;
    int cu_count = 256;
    int persistent = 1;   // 1: megakernel waves pull tiles from a counter (grid = what fits the chip)
    uint32_t* d_tile_counter = nullptr;
    uint32_t* d_tile_done = nullptr;
    int chunk_spp = -1;   // persistent megakernel work items: > 0 (pass, tile) items of that many samples, 0 whole
                          // tiles, -1 automatic (4 when the context has clearly more tiles than resident waves)
    int sah_visit_cost = 10;   // tenths of a triangle test (option sah_visit_cost)
    int pixel_map = 0;    // 0 tiles of 8x8 per wave, 1 strided (balances waves; for ranks with few waves)
    int debug_lds_pad = 0; // extra LDS bytes per block of the timed debug launches (limits occupancy)
    int debug_repeat = 0; // pt_debug_closest_hit: extra timed launches
    int cost_binning = 1; // wavefront: separate ray queues for rays that touch a complex object's box
    int min_waves = 4;    // k_render at 128 VGPRs (4 waves/SIMD) measured fastest
    int bvh_policy = 0;   // 0 auto (SAH termination; LDS fit when lds_scene is on), 1 SAH termination, 2 leaves of <= 4, 3 leaves of <= 8

    // ---- statistics
    std::vector<EventPair> events;
    size_t events_used = 0;
    double kernel_ms_acc = 0.0;
    int64_t kernel_launches = 0;
    size_t last_lds_bytes = 0;

    std::string err;
    char info[256] = {0};
};

namespace {

int fail(pt_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

}  // namespace

namespace ptamd {
int fail_ctx(pt_context* ctx, int code, const std::string& msg) { return fail(ctx, code, msg); }   // for pt_obj.cpp
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
struct RefOrder {
    const std::vector<pt_triangle>& tris;
    std::vector<int32_t>& rank;
    int32_t& next_rank;
    bool degenerate = false;

    static inline float mid(const pt_triangle& t, int a) { return (t.r1.s[a] + t.r2.s[a] + t.r3.s[a]) / 3.0f; }

    void run(std::vector<int32_t>& idx, int depth) {
        const size_t n = idx.size();
        if (n <= 6) {
            for (int32_t i : idx) rank[i] = next_rank++;
            return;
        }
        float m[3] = {mid(tris[idx[0]], 0), mid(tris[idx[0]], 1), mid(tris[idx[0]], 2)};
        for (size_t i = 1; i < n; ++i)
            for (int a = 0; a < 3; ++a) m[a] = m[a] + mid(tris[idx[i]], a);
        for (int a = 0; a < 3; ++a) m[a] = m[a] / (float)(unsigned long)n;
        int axis = depth % 3;
        std::vector<int32_t> left, right;
        for (int tries = 0;; ++tries) {
            left.clear();
            right.clear();
            for (int32_t i : idx) {
                if (m[axis] >= mid(tris[i], axis)) right.push_back(i); else left.push_back(i);
            }
            if (!left.empty() && !right.empty()) break;
            if (tries == 2) {  // the reference loops forever here (main.cpp:246-257)
                degenerate = true;
                return;
            }
            axis = (axis + 1) % 3;
        }
        std::vector<int32_t>().swap(idx);
        run(left, depth + 1);
        run(right, depth + 1);
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

    int32_t make_leaf(size_t lo, size_t hi) {
        int32_t first = (int32_t)order.size();
        for (size_t i = lo; i < hi; ++i) order.push_back(prims[i].tri);
        int32_t count = (int32_t)(hi - lo);
        return ~((first << 3) | (count - 1));
    }

    // returns child reference; *box receives the bounds of the subtree
    int32_t build(size_t lo, size_t hi, int depth, Aabb* box) {
        max_depth_seen = std::max(max_depth_seen, depth);
        const size_t n = hi - lo;
        Aabb b, cb;
        b.reset();
        cb.reset();
        for (size_t i = lo; i < hi; ++i) { b.grow(prims[i].box); cb.grow(prims[i].c); }
        *box = b;
        if (n <= 1) return make_leaf(lo, hi);

        // --- binned SAH over the three axes
        constexpr int NB = 16;
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3; ++a) {
            float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.f)) continue;
            Aabb bb[NB];
            int cnt[NB];
            for (int k = 0; k < NB; ++k) { bb[k].reset(); cnt[k] = 0; }
            float scale = (float)NB / ext;
            for (size_t i = lo; i < hi; ++i) {
                int k = (int)((prims[i].c[a] - cb.lo[a]) * scale);
                k = std::min(std::max(k, 0), NB - 1);
                bb[k].grow(prims[i].box);
                cnt[k]++;
            }
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
        if (n <= (size_t)max_leaf && (force_leaf || !(best_cost + visit_cost * b.half_area() < leaf_cost))) return make_leaf(lo, hi);

        size_t mid = lo;
        bool median = (best_axis < 0);
        if (!median) {
            float ext = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = (float)NB / ext;
            auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [&](const BuildPrim& p) {
                int k = (int)((p.c[best_axis] - cb.lo[best_axis]) * scale);
                k = std::min(std::max(k, 0), NB - 1);
                return k <= best_bin;
            });
            mid = (size_t)(it - prims.begin());
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
        int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Aabb lb, rb;
        int32_t l = build(lo, mid, depth + 1, &lb);
        int32_t r = build(mid, hi, depth + 1, &rb);
        Node64& nd = nodes[me];
        for (int a = 0; a < 3; ++a) { nd.q[a][0] = lb.lo[a]; nd.q[a][1] = lb.hi[a]; nd.q[a][2] = rb.lo[a]; nd.q[a][3] = rb.hi[a]; }
        nd.left = l;
        nd.right = r;
        nd.pad[0] = nd.pad[1] = 0;
        return me;
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

int build_attempt(pt_context* ctx, BvhBuilder& bld, const std::vector<BuildPrim>& prims, int max_leaf, bool force_leaf) {
    bld = BvhBuilder();
    bld.prims = prims;
    bld.max_leaf = max_leaf;
    bld.force_leaf = force_leaf;
    bld.visit_cost = (float)ctx->sah_visit_cost * 0.1f;
    bld.nodes.reserve(prims.size());
    bld.order.reserve(prims.size());
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
        int32_t root = bld.build(0, bld.prims.size(), 0, &box);
        if (root != 0) return fail(ctx, PT_ESCENE, "internal: BVH root is not node 0");
    }
    return PT_OK;
}

int build_and_pack(pt_context* ctx) {
    const size_t n = ctx->tris.size();
    std::vector<BuildPrim> prims;
    prims.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const pt_triangle& t = ctx->tris[i];
        bool finite = true;
        for (int a = 0; a < 3; ++a)
            finite = finite && std::isfinite(t.r1.s[a]) && std::isfinite(t.r2.s[a]) && std::isfinite(t.r3.s[a]);
        if (!finite) continue;  // cannot be hit (prog.cl:99-106 compares NaN) and has no box
        BuildPrim p;
        p.box = padded_bounds(t);
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
        p.tri = (int32_t)i;
        prims.push_back(p);
    }
    // Attempts, best traversal quality first; a later (fatter-leaved, smaller) tree is taken only
    // if it makes the whole scene fit the LDS of one CU next to the traversal stacks.
    const size_t lds_budget = 160 * 1024;
    auto footprint = [&](const BvhBuilder& b) {
        int entries = std::min(kStackEntries, ((b.max_depth_seen + 4) + 1) & ~1);
        return sizeof(Node64) * b.nodes.size() + sizeof(TriPacket) * b.order.size() + (size_t)entries * 4 * 256;
    };
    BvhBuilder bld;
    int rc = ctx->bvh_policy <= 1 ? build_attempt(ctx, bld, prims, 4, false)
                                  : build_attempt(ctx, bld, prims, ctx->bvh_policy == 2 ? 4 : 8, true);
    if (rc != PT_OK) return rc;
    if (ctx->bvh_policy == 0 && ctx->lds_scene == 1 && footprint(bld) > lds_budget && sizeof(TriPacket) * prims.size() < lds_budget) {
        const int tries[2][2] = {{4, 1}, {8, 1}};
        for (auto& t : tries) {
            BvhBuilder alt;
            rc = build_attempt(ctx, alt, prims, t[0], t[1] != 0);
            if (rc != PT_OK) return rc;
            if (footprint(alt) <= lds_budget) { bld = std::move(alt); break; }
        }
    }
    if (bld.max_depth_seen > kMaxDepth) return fail(ctx, PT_ESCENE, "internal: BVH deeper than the traversal stack");
    ctx->bvh_depth = bld.max_depth_seen;
    ctx->nodes.swap(bld.nodes);
    ctx->orig.swap(bld.order);
    const size_t m = ctx->orig.size();
    ctx->packets.resize(std::max<size_t>(m, 1));
    ctx->meta.resize(std::max<size_t>(m, 1));
    std::memset(ctx->packets.data(), 0, sizeof(TriPacket) * ctx->packets.size());
    std::memset(ctx->meta.data(), 0, sizeof(TriMeta) * ctx->meta.size());
    compute_cost_boxes_impl(ctx);
    for (size_t k = 0; k < m; ++k) {
        const pt_triangle& t = ctx->tris[ctx->orig[k]];
        float* v = ctx->packets[k].v;
        v[0] = t.r1.s[0]; v[1] = t.r1.s[1]; v[2] = t.r1.s[2];
        v[3] = t.r2.s[0]; v[4] = t.r2.s[1]; v[5] = t.r2.s[2];
        v[6] = t.r3.s[0]; v[7] = t.r3.s[1]; v[8] = t.r3.s[2];
        v[9] = t.N.s[0]; v[10] = t.N.s[1]; v[11] = t.N.s[2];
        ctx->meta[k].rank = ctx->enc_rank[ctx->orig[k]];
        ctx->meta[k].mati = t.mati;
    }
    return PT_OK;
}

// bounding boxes of the complex objects (more than 16 triangles), for the wavefront cost classes
void compute_cost_boxes_impl(pt_context* ctx) {
    {
        struct OB { Aabb b; size_t n; };
        std::vector<OB> obs;
        for (size_t o = 0; o < ctx->obj_begin.size(); ++o) {
            const size_t lo = (size_t)ctx->obj_begin[o], hi = o + 1 < ctx->obj_begin.size() ? (size_t)ctx->obj_begin[o + 1] : ctx->tris.size();
            if (hi - lo <= 16) continue;
            OB ob;
            ob.b.reset();
            ob.n = hi - lo;
            for (size_t i = lo; i < hi; ++i) ob.b.grow(padded_bounds(ctx->tris[i]));
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
    p->stack_entries = std::min(kStackEntries, ((ctx->bvh_depth + 4) + 1) & ~1);   // sentinel + far children + the slot above the top (Trav::node_step)
    p->pixel_map = ctx->pixel_map;
    p->tile_counter = nullptr;
    p->chunk_spp = 0;
    p->tile_done = nullptr;
    p->n_tiles = ((ctx->W + 7) / 8) * ((ctx->local_rows + 7) / 8);
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

int pt_create_tiled(int device, int32_t width, int32_t height, int32_t rank, int32_t world, int32_t rows_per_block, pt_context** out) {
    if (!out) return fail(nullptr, PT_EINVAL, "out is NULL");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(nullptr, PT_EINVAL, "bad frame size");
    if (world < 1 || rank < 0 || rank >= world || rows_per_block < 1) return fail(nullptr, PT_EINVAL, "bad rank/world/rows_per_block");
    pt_context* ctx = new pt_context();
    ctx->W = width;
    ctx->H = height;
    ctx->rank = rank;
    ctx->world = world;
    ctx->rows_per_block = rows_per_block;
    ctx->local_rows = count_local_rows(height, rank, world, rows_per_block);
    ctx->npix = (int64_t)ctx->local_rows * width;
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
    if ((e = hipMalloc((void**)&ctx->d_rays, sizeof(pt_ray) * np)) != hipSuccess) return bail("hipMalloc(rays)", e);      // main.cpp:508
    if ((e = hipMalloc((void**)&ctx->d_rnds, sizeof(int32_t) * np)) != hipSuccess) return bail("hipMalloc(rnds)", e);     // main.cpp:509
    if ((e = hipMalloc((void**)&ctx->d_colors, sizeof(float4) * np)) != hipSuccess) return bail("hipMalloc(colors)", e);  // main.cpp:520
    if ((e = hipMalloc((void**)&ctx->d_stats, sizeof(unsigned long long) * 8 * kStatRows)) != hipSuccess) return bail("hipMalloc(stats)", e);
    if ((e = hipMalloc((void**)&ctx->d_tile_counter, 64)) != hipSuccess) return bail("hipMalloc(tile counter)", e);
    if ((e = hipMemset(ctx->d_rays, 0, sizeof(pt_ray) * np)) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipMemset(ctx->d_colors, 0, sizeof(float4) * np)) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipMemset(ctx->d_stats, 0, sizeof(unsigned long long) * 8 * kStatRows)) != hipSuccess) return bail("hipMemset", e);
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
        if (ctx->d_tris) (void)hipFree(ctx->d_tris);
        if (ctx->d_meta) (void)hipFree(ctx->d_meta);
        if (ctx->d_mats) (void)hipFree(ctx->d_mats);
        if (ctx->d_rays) (void)hipFree(ctx->d_rays);
        if (ctx->d_ldr) (void)hipFree(ctx->d_ldr);
        if (ctx->d_stats) (void)hipFree(ctx->d_stats);
        if (ctx->d_tile_counter) (void)hipFree(ctx->d_tile_counter);
        if (ctx->d_tile_done) (void)hipFree(ctx->d_tile_done);
        if (ctx->d_wf_state) (void)hipFree(ctx->d_wf_state);
        if (ctx->d_wf_queues) (void)hipFree(ctx->d_wf_queues);
        if (ctx->d_wf_counters) (void)hipFree(ctx->d_wf_counters);
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
    if ((int64_t)ctx->tris.size() + n > ((int64_t)1 << 27)) return fail(ctx, PT_EINVAL, "more than 2^27 triangles");
    ctx->tris.insert(ctx->tris.end(), t, t + n);
    ctx->tris_uploaded = false;
    return PT_OK;
}

int pt_end_obj(pt_context* ctx) {
    if (!ctx) return PT_EINVAL;
    const int32_t n = (int32_t)ctx->tris.size() - ctx->tri_shift;
    if (n <= 0) return fail(ctx, PT_ESCENE, "end_Obj on an empty object (the reference reads tris[0] of an empty vector, main.cpp:216)");
    ctx->enc_rank.resize(ctx->tris.size(), -1);
    std::vector<int32_t> idx((size_t)n);
    std::iota(idx.begin(), idx.end(), ctx->tri_shift);
    RefOrder ro{ctx->tris, ctx->enc_rank, ctx->next_rank};
    ro.run(idx, 0);
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

// bvh_policy 4: build the tree on the device (pt_lbvh.hip); host copies are kept for the debug getters
static int build_on_device(pt_context* ctx, bool* done) {
    *done = false;
    const int n = (int)ctx->tris.size();
    if (!ctx->has_device || n <= 2 * kMaxLeaf) return PT_OK;
    for (const pt_triangle& t : ctx->tris)
        for (int a = 0; a < 3; ++a)
            if (!std::isfinite(t.r1.s[a]) || !std::isfinite(t.r2.s[a]) || !std::isfinite(t.r3.s[a])) return PT_OK;   // host path handles those
    PT_HIP(ctx, hipSetDevice(ctx->device));
    LbvhResult r;
    PT_HIP(ctx, lbvh_build(ctx->tris.data(), ctx->enc_rank.data(), n, ctx->stream, &r));
    if (r.depth + 5 > kStackEntries) {          // deeper than the traversal stack: let the host builder do it
        (void)hipFree(r.d_nodes); (void)hipFree(r.d_tris); (void)hipFree(r.d_meta); (void)hipFree(r.d_orig);
        return PT_OK;
    }
    if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
    if (ctx->d_tris) (void)hipFree(ctx->d_tris);
    if (ctx->d_meta) (void)hipFree(ctx->d_meta);
    ctx->d_nodes = r.d_nodes;
    ctx->d_tris = r.d_tris;
    ctx->d_meta = r.d_meta;
    ctx->nodes.resize((size_t)r.n_nodes);
    ctx->packets.resize((size_t)n);
    ctx->meta.resize((size_t)n);
    ctx->orig.resize((size_t)n);
    PT_HIP(ctx, hipMemcpy(ctx->nodes.data(), r.d_nodes, sizeof(Node64) * (size_t)r.n_nodes, hipMemcpyDeviceToHost));
    PT_HIP(ctx, hipMemcpy(ctx->packets.data(), r.d_tris, sizeof(TriPacket) * (size_t)n, hipMemcpyDeviceToHost));
    PT_HIP(ctx, hipMemcpy(ctx->meta.data(), r.d_meta, sizeof(TriMeta) * (size_t)n, hipMemcpyDeviceToHost));
    PT_HIP(ctx, hipMemcpy(ctx->orig.data(), r.d_orig, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(r.d_orig);
    ctx->bvh_depth = r.depth + 1;
    *done = true;
    return PT_OK;
}

static void compute_cost_boxes(pt_context* ctx) { compute_cost_boxes_impl(ctx); }   // defined in the anonymous namespace above

int pt_upload_triangles(pt_context* ctx) {
    if (!ctx) return PT_EINVAL;
    if (ctx->tri_shift != (int32_t)ctx->tris.size())
        return fail(ctx, PT_EINVAL, "triangles were added after the last end_Obj; close the object first (main.cpp:536)");
    const auto t0 = std::chrono::steady_clock::now();
    if (ctx->bvh_policy == 4) {
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
    int rc = build_and_pack(ctx);
    ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != PT_OK) return rc;
    if (ctx->has_device) {
        PT_HIP(ctx, hipSetDevice(ctx->device));
        if ((rc = upload_vec(ctx, &ctx->d_nodes, ctx->nodes.data(), sizeof(Node64) * ctx->nodes.size())) != PT_OK) return rc;
        if ((rc = upload_vec(ctx, &ctx->d_tris, ctx->packets.data(), sizeof(TriPacket) * ctx->packets.size())) != PT_OK) return rc;
        if ((rc = upload_vec(ctx, &ctx->d_meta, ctx->meta.data(), sizeof(TriMeta) * ctx->meta.size())) != PT_OK) return rc;
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

static int launch_cfg(pt_context* ctx, const RenderParams& p, LaunchConfig* lc, int block) {
    lc->block = block;
    lc->lds_bytes = mega_lds_bytes(p, block);
    lc->count_work = ctx->count_work != 0;
    lc->min_waves = ctx->min_waves;
    lc->traversal = ctx->traversal;
    // resident workgroups: 256 CUs x (2048 threads at 4 waves/SIMD ... the launcher's MINW decides; use
    // the LDS/VGPR-limited count of the default configurations: 4 x 256-thread or 2 x 512-thread blocks)
    lc->persistent_blocks = ctx->cu_count * std::max(1, 1024 / block);
    ctx->last_lds_bytes = lc->lds_bytes;
    return PT_OK;
}

static void decide_lds_scene(const pt_context* ctx, RenderParams* p, int* block_out) {
    *block_out = ctx->block;
    if (ctx->lds_scene == 2) {      // nodes only: needs the 16-bit stack encoding and two 512-thread blocks per CU
        const bool s16 = ctx->nodes.size() <= 32767 && ctx->orig.size() <= 4096;
        const size_t need = sizeof(Node64) * ctx->nodes.size() + (size_t)p->stack_entries * 2 * 512 + 32;
        p->lds_scene = (s16 && need <= 80 * 1024) ? 2 : 0;
        if (p->lds_scene) *block_out = 512;
        return;
    }
    size_t scene = sizeof(Node64) * ctx->nodes.size() + sizeof(TriPacket) * ctx->orig.size();
    const bool s16 = ctx->nodes.size() <= 32767 && ctx->orig.size() <= 4096;
    size_t stack = (size_t)p->stack_entries * (s16 ? 2 : 4) * (size_t)ctx->block + 16;
    p->lds_scene = (ctx->lds_scene && scene + stack <= (size_t)mega_max_lds_scene_bytes()) ? 1 : 0;
}

int pt_generate_rays(pt_context* ctx, const pt_camera* cam) {
    PT_NEED_DEVICE(ctx);
    int rc = check_ready(ctx, cam);
    if (rc != PT_OK) return rc;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    RenderParams p;
    fill_params(ctx, cam, &p);
    LaunchConfig lc;
    lc.block = 256;
    PT_HIP(ctx, launch_gen_ray(p, lc, ctx->stream));
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
    int blk = ctx->block;
    decide_lds_scene(ctx, &p, &blk);
    LaunchConfig lc;
    launch_cfg(ctx, p, &lc, blk);
    EventPair* ep;
    if ((rc = time_begin(ctx, &ep)) != PT_OK) return rc;
    PT_HIP(ctx, launch_trace_ray(p, lc, ctx->stream));
    return time_end(ctx, ep);
}

static int render_wavefront(pt_context* ctx, const RenderParams& rp, int32_t nsamples) {
    if (rp.iterations > kWfMaxBounces) return fail(ctx, PT_EINVAL, "wavefront variant supports at most 1023 iterations");
    const size_t np = (size_t)std::max<int64_t>(ctx->npix, 1);
    if (!ctx->d_wf_state) {
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_wf_state, sizeof(float4) * 13 * np));
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_wf_queues, sizeof(int32_t) * 3 * np));
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_wf_counters, sizeof(uint32_t) * kWfCounterStride * (kWfMaxBounces + 4)));
    }
    WfParams w;
    w.rp = rp;
    w.sC = ctx->d_wf_state + 0 * np; w.sD = ctx->d_wf_state + 1 * np;
    w.sE = ctx->d_wf_state + 2 * np; w.sF = ctx->d_wf_state + 3 * np;
    for (int par = 0; par < 2; ++par)
        for (int c = 0; c < 2; ++c) {
            w.rsA[par][c] = ctx->d_wf_state + (size_t)(4 + (par * 2 + c) * 2 + 0) * np;
            w.rsB[par][c] = ctx->d_wf_state + (size_t)(4 + (par * 2 + c) * 2 + 1) * np;
        }
    w.hit[0] = reinterpret_cast<float2*>(ctx->d_wf_state + 12 * np);
    w.hit[1] = w.hit[0] + np;
    w.q_cls[0] = ctx->d_wf_queues + 0 * np; w.q_cls[1] = ctx->d_wf_queues + 1 * np; w.q_cls[2] = ctx->d_wf_queues + 2 * np;
    w.counters = ctx->d_wf_counters;
    w.npix = (int32_t)ctx->npix;
    w.n_cbox = ctx->cost_binning ? (int32_t)(ctx->cost_boxes.size() / 6) : 0;
    for (int b = 0; b < w.n_cbox; ++b)
        for (int k = 0; k < 6; ++k) w.cbox[b][k] = ctx->cost_boxes[(size_t)b * 6 + k];
    if (ctx->npix == 0) return PT_OK;
    for (int32_t k = 0; k < nsamples; ++k) {
        w.sample = rp.first_sample + k;
        PT_HIP(ctx, hipMemsetAsync(ctx->d_wf_counters, 0, sizeof(uint32_t) * kWfCounterStride, ctx->stream));
        PT_HIP(ctx, launch_wf_generate(w, ctx->stream));
        for (int32_t b = 0; b < rp.iterations; ++b) {
            EventPair* ep;
            int rc = time_begin(ctx, &ep);
            if (rc != PT_OK) return rc;
            PT_HIP(ctx, launch_wf_intersect(w, b, ctx->stream));
            if ((rc = time_end(ctx, ep)) != PT_OK) return rc;
            PT_HIP(ctx, launch_wf_shade(w, b, ctx->stream));
        }
        if (ctx->timing && ctx->events_used >= 4096) {   // bound the event pool
            int rc = time_collect(ctx);
            if (rc != PT_OK) return rc;
        }
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
    if (ctx->variant == 1) {
        p.lds_scene = 0;
        if ((rc = render_wavefront(ctx, p, nsamples)) != PT_OK) return rc;
        ctx->current_sample += nsamples;
        return PT_OK;
    }
    int blk = ctx->block;
    decide_lds_scene(ctx, &p, &blk);
    if (ctx->persistent) {
        PT_HIP(ctx, hipMemsetAsync(ctx->d_tile_counter, 0, sizeof(uint32_t), ctx->stream));
        p.tile_counter = ctx->d_tile_counter;
        // automatic: chaining passes only pays when there are enough tiles to re-balance; with about one
        // tile per resident wave (a 1080p frame over 8 GPUs) the most expensive tile is the critical path
        // either way and the extra hand-offs cost 2-4 %.  With many tiles per wave (one GPU, 1080p: 7.9)
        // passes of 8 samples re-balance just as well as passes of 4 and halve the hand-overs
        // (measured 0 / 2 / 4 / 8 / 16 -> 1409 / 1406 / 1456 / 1470 / 1465 Msamples/s).
        const int resident_waves = ctx->cu_count * 16;
        const int auto_chunk = p.n_tiles >= 6 * resident_waves ? 8 : (p.n_tiles > resident_waves + resident_waves / 2 ? 4 : 0);
        const int chunk = ctx->chunk_spp >= 0 ? ctx->chunk_spp : auto_chunk;
        if (chunk > 0 && nsamples > chunk && p.n_tiles > 0) {
            if (!ctx->d_tile_done) PT_HIP(ctx, hipMalloc((void**)&ctx->d_tile_done, sizeof(uint32_t) * (size_t)p.n_tiles));
            PT_HIP(ctx, hipMemsetAsync(ctx->d_tile_done, 0, sizeof(uint32_t) * (size_t)p.n_tiles, ctx->stream));
            p.tile_done = ctx->d_tile_done;
            p.chunk_spp = chunk;
        }
    }
    LaunchConfig lc;
    launch_cfg(ctx, p, &lc, blk);
    EventPair* ep;
    if ((rc = time_begin(ctx, &ep)) != PT_OK) return rc;
    PT_HIP(ctx, launch_render_mega(p, lc, ctx->stream));
    if ((rc = time_end(ctx, ep)) != PT_OK) return rc;
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

int pt_sync(pt_context* ctx) {
    PT_NEED_DEVICE(ctx);
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_local_pixel_count(const pt_context* ctx, int64_t* out) {
    if (!ctx || !out) return PT_EINVAL;
    *out = ctx->npix;
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
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    } else if (k == "block") {
        if (value != 64 && value != 128 && value != 256 && value != 512 && value != 1024) return fail(ctx, PT_EINVAL, "block must be 64..1024, power of two");
        ctx->block = (int)value;
    } else if (k == "lds_scene") {
        if (value < 0 || value > 2) return fail(ctx, PT_EINVAL, "lds_scene: 0 off, 1 nodes + packets in LDS, 2 nodes in LDS");
        if ((value == 1) != (ctx->lds_scene == 1) && ctx->bvh_policy == 0 && ctx->tris_uploaded) {
            ctx->lds_scene = (int)value;             // the automatic BVH policy depends on it: rebuild
            int rc = pt_upload_triangles(ctx);
            if (rc != PT_OK) return rc;
        }
        ctx->lds_scene = (int)value;
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
    } else if (k == "pixel_map") {
        ctx->pixel_map = value ? 1 : 0;
    } else if (k == "debug_lds_pad") {
        ctx->debug_lds_pad = (int)value;
    } else if (k == "debug_repeat") {
        ctx->debug_repeat = (int)value;
    } else if (k == "cost_binning") {
        ctx->cost_binning = value ? 1 : 0;
    } else if (k == "traversal") {
        if (value < 0 || value > 64) return fail(ctx, PT_EINVAL, "traversal: 0 while-while, 1 voting, n >= 2 sliced with n-1 rounds per trip");
        ctx->traversal = (int)value;
    } else if (k == "min_waves") {
        ctx->min_waves = (int)value;
    } else if (k == "bvh_policy") {
        if (value < 0 || value > 4) return fail(ctx, PT_EINVAL, "bvh_policy must be 0..4 (4 = device LBVH)");
        ctx->bvh_policy = (int)value;
        ctx->tris_uploaded = false;
    } else if (k == "reset_stats") {
        if (ctx->has_device) {
            PT_HIP(ctx, hipSetDevice(ctx->device));
            PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            PT_HIP(ctx, hipMemset(ctx->d_stats, 0, sizeof(unsigned long long) * 8 * kStatRows));
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
    if (k == "bvh_build_ms") { *out = ctx->bvh_build_ms; return PT_OK; }
    if (k == "bvh_on_device") { *out = (double)ctx->bvh_on_device; return PT_OK; }
    if (k == "triangles") { *out = (double)ctx->orig.size(); return PT_OK; }
    if (k == "lds_bytes") { *out = (double)ctx->last_lds_bytes; return PT_OK; }
    if (k == "kernel_launches") { *out = (double)ctx->kernel_launches; return PT_OK; }
    PT_NEED_DEVICE(ctx);
    PT_HIP(ctx, hipSetDevice(ctx->device));
    if (k == "kernel_ms") {
        int rc = time_collect(ctx);
        if (rc != PT_OK) return rc;
        *out = ctx->kernel_ms_acc;
        return PT_OK;
    }
    if (k == "segments" || k == "samples" || k == "node_visits" || k == "tri_tests" || k == "wave_node_steps" || k == "wave_tri_steps" || k == "tile_lane_steps") {
        std::vector<unsigned long long> rows((size_t)8 * kStatRows);
        unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PT_HIP(ctx, hipMemcpy(rows.data(), ctx->d_stats, sizeof(unsigned long long) * rows.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < kStatRows; ++r)
            for (int c = 0; c < 8; ++c) h[c] += rows[(size_t)r * 8 + c];
        *out = (double)h[k == "segments" ? 0 : k == "samples" ? 1 : k == "node_visits" ? 2 : k == "tri_tests" ? 3 : k == "wave_node_steps" ? 4 : k == "wave_tri_steps" ? 5 : 6];
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

int pt_debug_bvh_copy(const pt_context* ctx, float* nodes, float* tris, int32_t* meta, int32_t* orig) {
    if (!ctx || !ctx->tris_uploaded) return PT_EINVAL;
    if (nodes) std::memcpy(nodes, ctx->nodes.data(), sizeof(Node64) * ctx->nodes.size());
    if (tris) std::memcpy(tris, ctx->packets.data(), sizeof(TriPacket) * ctx->orig.size());
    if (meta) std::memcpy(meta, ctx->meta.data(), sizeof(TriMeta) * ctx->orig.size());
    if (orig) std::memcpy(orig, ctx->orig.data(), sizeof(int32_t) * ctx->orig.size());
    return PT_OK;
}

int pt_debug_closest_hit(pt_context* ctx, const pt_ray* rays, int64_t n, float* out_t, int32_t* out_tri) {
    PT_NEED_DEVICE(ctx);
    if (!rays || !out_t || !out_tri || n < 0) return fail(ctx, PT_EINVAL, "bad arguments");
    if (!ctx->tris_uploaded) return fail(ctx, PT_EINVAL, "pt_upload_triangles has not been called");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    pt_camera cam;
    std::memset(&cam, 0, sizeof cam);
    RenderParams p;
    fill_params(ctx, &cam, &p);
    pt_ray* d_rays = nullptr;
    float* d_t = nullptr;
    int32_t* d_tri = nullptr;
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    PT_HIP(ctx, hipMalloc((void**)&d_rays, sizeof(pt_ray) * nn));
    PT_HIP(ctx, hipMalloc((void**)&d_t, sizeof(float) * nn));
    PT_HIP(ctx, hipMalloc((void**)&d_tri, sizeof(int32_t) * nn));
    PT_HIP(ctx, hipMemcpy(d_rays, rays, sizeof(pt_ray) * (size_t)n, hipMemcpyHostToDevice));
    // same node path as the render kernel would take: nodes staged in LDS (swizzled quads, 16-bit
    // references, one-fma slab test) when the scene qualifies, otherwise straight from global memory
    int blk = 0;
    decide_lds_scene(ctx, &p, &blk);
    PT_HIP(ctx, launch_debug_closest_hit(p, d_rays, n, d_t, d_tri, ctx->stream, p.lds_scene == 2 ? 2 : 0));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->debug_repeat > 0) {      // traversal-only timing: the same launch, debug_repeat times
        hipEvent_t e0, e1;
        PT_HIP(ctx, hipEventCreate(&e0));
        PT_HIP(ctx, hipEventCreate(&e1));
        PT_HIP(ctx, hipEventRecord(e0, ctx->stream));
        for (int r = 0; r < ctx->debug_repeat; ++r) PT_HIP(ctx, launch_debug_closest_hit(p, d_rays, n, d_t, d_tri, ctx->stream, (size_t)ctx->debug_lds_pad));
        PT_HIP(ctx, hipEventRecord(e1, ctx->stream));
        PT_HIP(ctx, hipEventSynchronize(e1));
        float ms = 0.f;
        PT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        ctx->kernel_ms_acc += ms;
        ctx->kernel_launches += ctx->debug_repeat;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    PT_HIP(ctx, hipMemcpy(out_t, d_t, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    PT_HIP(ctx, hipMemcpy(out_tri, d_tri, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i)
        if (out_tri[i] >= 0) out_tri[i] = ctx->orig[(size_t)out_tri[i]];      // packed -> add order
    (void)hipFree(d_rays);
    (void)hipFree(d_t);
    (void)hipFree(d_tri);
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

int pt_debug_encounter_rank(const pt_context* ctx, int32_t* out, int64_t n) {
    if (!ctx || !out || n != (int64_t)ctx->enc_rank.size()) return PT_EINVAL;
    std::memcpy(out, ctx->enc_rank.data(), sizeof(int32_t) * (size_t)n);
    return PT_OK;
}

}  // extern "C"
