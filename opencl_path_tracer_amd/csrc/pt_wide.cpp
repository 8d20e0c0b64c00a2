// pt_wide.cpp -- collapses the packed BVH2 into 4-wide nodes with child boxes quantised to 8 bits per plane
// (Node4q, pt_internal.hpp).  Host code; the traversal is Trav<kNodesWide>::wide_step in pt_device.hpp.
//
// Why: for scenes whose tree lives in global memory every node visit is a dependent memory round trip that the
// whole wave waits out.  A 4-wide node decides two BVH2 levels with ONE 64-byte fetch --
// the same bytes a BVH2 node takes -- so a ray makes about half as many round trips and moves about half as many
// node bytes.  Only culling changes: a child's box may only grow (the planes are rounded outwards, and checked
// here with the very fma the kernel decodes them with), the exact triangle test and the tie-break by encounter
// rank decide the hit as before, so results stay bit-identical.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

#include "pt_internal.hpp"

namespace ptamd {

namespace {

struct Kid {
    float lo[3], hi[3];
    int32_t ref;
};

inline void kids_of(const Node64& nd, Kid out[2]) {
    for (int a = 0; a < 3; ++a) {
        out[0].lo[a] = nd.q[a][0];
        out[0].hi[a] = nd.q[a][1];
        out[1].lo[a] = nd.q[a][2];
        out[1].hi[a] = nd.q[a][3];
    }
    out[0].ref = nd.left;
    out[1].ref = nd.right;
}

inline bool empty_box(const Kid& k) { return !(k.lo[0] <= k.hi[0] && k.lo[1] <= k.hi[1] && k.lo[2] <= k.hi[2]); }

inline double half_area(const Kid& k) {
    const double dx = (double)k.hi[0] - k.lo[0], dy = (double)k.hi[1] - k.lo[1], dz = (double)k.hi[2] - k.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

inline float step_of(int biased_exp) {
    const uint32_t bits = (uint32_t)biased_exp << 23;
    float f;
    std::memcpy(&f, &bits, sizeof f);
    return f;
}

// One axis of one node: grid origin o = the lowest child plane, step 2^(e - 127) with the smallest e for which every
// child's planes fit in a byte.  A plane decodes as fmaf((float)q, step, o) -- on the device with the same single
// rounding -- and q is moved until the decoded low plane is <= the child's and the decoded high plane >= it.
bool quantise_axis(const Kid* kids, int n, int a, float* origin, int* biased_exp, uint8_t qlo[4], uint8_t qhi[4]) {
    float o = std::numeric_limits<float>::infinity(), top = -std::numeric_limits<float>::infinity();
    for (int k = 0; k < n; ++k) { o = std::min(o, kids[k].lo[a]); top = std::max(top, kids[k].hi[a]); }
    if (!std::isfinite(o) || !std::isfinite(top)) return false;
    const double ext = (double)top - (double)o;
    int e = 1;
    if (ext > 0.0) e = std::max(1, std::min(254, (int)std::ilogb(ext / 255.0) + 127));
    for (; e <= 254; ++e) {
        const float step = step_of(e);
        bool ok = true;
        for (int k = 0; k < n && ok; ++k) {
            long lo = (long)std::floor(((double)kids[k].lo[a] - (double)o) / (double)step);
            lo = std::max(0l, std::min(255l, lo));
            while (lo > 0 && std::fmaf((float)lo, step, o) > kids[k].lo[a]) --lo;
            if (std::fmaf((float)lo, step, o) > kids[k].lo[a]) { ok = false; break; }
            long hi = (long)std::ceil(((double)kids[k].hi[a] - (double)o) / (double)step);
            hi = std::max(0l, hi);
            while (hi <= 255 && std::fmaf((float)hi, step, o) < kids[k].hi[a]) ++hi;
            if (hi > 255) { ok = false; break; }
            qlo[k] = (uint8_t)lo;
            qhi[k] = (uint8_t)hi;
        }
        if (ok) {
            *origin = o;
            *biased_exp = e;
            return true;
        }
    }
    return false;
}

}  // namespace

namespace {

struct Work { int32_t src; int32_t dst; int pending; };

// One 4-wide node from BVH2 node `src`: its children (opened top-down), quantised.  Interior children are reported in
// kid_src[] (their BVH2 node) with ref[k] left for the caller to fill; returns the number of children, or -1 when the
// boxes cannot be quantised.
int make_wide_node(const std::vector<Node64>& bvh2, int32_t src, Node4q* out, int32_t kid_src[4]) {
    Kid kids[4];
    int n = 0;
    {
        Kid two[2];
        kids_of(bvh2[(size_t)src], two);
        for (int k = 0; k < 2; ++k)
            if (!empty_box(two[k])) kids[n++] = two[k];
    }
    // open the interior child with the largest box until there are four children
    while (n < 4) {
        int pick = -1;
        double best = -1.0;
        for (int k = 0; k < n; ++k)
            if (kids[k].ref >= 0 && half_area(kids[k]) > best) { best = half_area(kids[k]); pick = k; }
        if (pick < 0) break;
        Kid two[2];
        kids_of(bvh2[(size_t)kids[pick].ref], two);
        int live = 0;
        Kid keep[2];
        for (int k = 0; k < 2; ++k)
            if (!empty_box(two[k])) keep[live++] = two[k];
        if (live == 0) { kids[pick] = kids[--n]; continue; }
        kids[pick] = keep[0];
        if (live == 2) kids[n++] = keep[1];
    }
    Node4q nd;
    std::memset(&nd, 0, sizeof nd);
    uint8_t qlo[3][4] = {}, qhi[3][4] = {};
    int e[3] = {1, 1, 1};
    if (n > 0) {
        for (int a = 0; a < 3; ++a)
            if (!quantise_axis(kids, n, a, &nd.origin[a], &e[a], qlo[a], qhi[a])) return -1;
    }
    for (int a = 0; a < 3; ++a) nd.exp[a] = (uint8_t)e[a];
    nd.nchild = (uint8_t)n;
    for (int k = 0; k < 4; ++k) {
        kid_src[k] = -1;
        if (k >= n) {                       // no child: an inverted box, which a ray can only "hit" through the widening of
            nd.ref[k] = kWideNoChild;       // far exit distances, and a reference that is harmless then (packet 0 as a leaf)
            for (int a = 0; a < 3; ++a) { qlo[a][k] = 255; qhi[a][k] = 0; }
            continue;
        }
        if (kids[k].ref >= 0) kid_src[k] = kids[k].ref;      // interior: the caller numbers it
        else nd.ref[k] = kids[k].ref;
    }
    for (int k = 0; k < 4; ++k) {
        nd.qlo_x |= (uint32_t)qlo[0][k] << (8 * k);
        nd.qhi_x |= (uint32_t)qhi[0][k] << (8 * k);
        nd.qlo_y |= (uint32_t)qlo[1][k] << (8 * k);
        nd.qhi_y |= (uint32_t)qhi[1][k] << (8 * k);
        nd.qlo_z |= (uint32_t)qlo[2][k] << (8 * k);
        nd.qhi_z |= (uint32_t)qhi[2][k] << (8 * k);
    }
    *out = nd;
    return n;
}

// the subtree below one 4-wide node, depth first into `out` (indices local to `out`; the subtree's root is written to *root)
bool collapse_subtree(const std::vector<Node64>& bvh2, const Work& top, Node4q* root, std::vector<Node4q>* out, int* max_pending) {
    std::vector<Work> todo;
    todo.push_back({top.src, -1, top.pending});          // dst -1 = *root
    while (!todo.empty()) {
        const Work w = todo.back();
        todo.pop_back();
        *max_pending = std::max(*max_pending, w.pending);
        Node4q nd;
        int32_t kid_src[4];
        const int n = make_wide_node(bvh2, w.src, &nd, kid_src);
        if (n < 0) return false;
        for (int k = 0; k < 4; ++k)
            if (kid_src[k] >= 0) {
                const int32_t dst = (int32_t)out->size();
                out->emplace_back();
                nd.ref[k] = dst;
                todo.push_back({kid_src[k], dst, w.pending + n - 1});
            }
        if (w.dst < 0) *root = nd; else (*out)[(size_t)w.dst] = nd;
    }
    return true;
}

}  // namespace

// Returns false when the tree cannot be expressed (non-finite boxes): the caller keeps the BVH2 path.
// *max_pending = the most stack entries a traversal can hold when it visits an interior node: every ancestor may
// have left all its other children on the stack.
// The top of the tree is collapsed breadth first (its nodes come first, level by level) until there are a few hundred
// open subtrees; those are collapsed depth first on `threads` threads, each into a block of its own, and appended in a
// fixed order -- the numbering does not depend on the number of threads.
bool build_wide_nodes(const std::vector<Node64>& bvh2, std::vector<Node4q>* out, int* max_pending, int threads) {
    out->clear();
    *max_pending = 0;
    if (bvh2.empty()) return false;
    constexpr size_t kFrontier = 512;
    std::vector<Work> frontier;      // FIFO
    size_t head = 0;
    out->emplace_back();
    frontier.push_back({0, 0, 0});
    while (head < frontier.size() && frontier.size() - head < kFrontier) {
        const Work w = frontier[head++];
        *max_pending = std::max(*max_pending, w.pending);
        Node4q nd;
        int32_t kid_src[4];
        const int n = make_wide_node(bvh2, w.src, &nd, kid_src);
        if (n < 0) return false;
        for (int k = 0; k < 4; ++k)
            if (kid_src[k] >= 0) {
                const int32_t dst = (int32_t)out->size();
                out->emplace_back();
                nd.ref[k] = dst;
                frontier.push_back({kid_src[k], dst, w.pending + n - 1});
            }
        (*out)[(size_t)w.dst] = nd;
    }
    const size_t n_tasks = frontier.size() - head;
    if (n_tasks == 0) return true;
    std::vector<std::vector<Node4q>> blocks(n_tasks);
    std::vector<Node4q> roots(n_tasks);
    std::vector<int> pend(n_tasks, 0);
    std::atomic<size_t> next(0);
    std::atomic<bool> ok(true);
    auto work = [&]() {
        for (size_t t = next.fetch_add(1); t < n_tasks; t = next.fetch_add(1))
            if (!collapse_subtree(bvh2, frontier[head + t], &roots[t], &blocks[t], &pend[t])) ok.store(false);
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < std::max(1, std::min<int>(threads, (int)n_tasks)); ++k) pool.emplace_back(work);
    work();
    for (std::thread& th : pool) th.join();
    if (!ok.load()) { out->clear(); return false; }
    size_t total = out->size();
    std::vector<size_t> base(n_tasks);
    for (size_t t = 0; t < n_tasks; ++t) { base[t] = total; total += blocks[t].size(); }
    out->resize(total);
    auto place = [&](size_t t) {
        const int32_t b = (int32_t)base[t];
        Node4q r = roots[t];
        for (int k = 0; k < 4; ++k)
            if (r.ref[k] >= 0) r.ref[k] += b;
        (*out)[(size_t)frontier[head + t].dst] = r;
        for (size_t i = 0; i < blocks[t].size(); ++i) {
            Node4q nd = blocks[t][i];
            for (int k = 0; k < 4; ++k)
                if (nd.ref[k] >= 0) nd.ref[k] += b;
            (*out)[base[t] + i] = nd;
        }
    };
    next.store(0);
    auto work2 = [&]() {
        for (size_t t = next.fetch_add(1); t < n_tasks; t = next.fetch_add(1)) place(t);
    };
    pool.clear();
    for (int k = 1; k < std::max(1, std::min<int>(threads, (int)n_tasks)); ++k) pool.emplace_back(work2);
    work2();
    for (std::thread& th : pool) th.join();
    for (size_t t = 0; t < n_tasks; ++t) *max_pending = std::max(*max_pending, pend[t]);
    return true;
}

}  // namespace ptamd
