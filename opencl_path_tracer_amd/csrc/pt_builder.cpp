// pt_builder.cpp -- scene -> tree (host side of libptamd.so; shared declarations: pt_context.hpp).
//   * end_Obj's encounter ranks: the order in which the reference's per-object mean-split tree meets triangles (main.cpp:210-262,
//     prog.cl:113-184), needed only to break exact-t ties
//   * the own BVH: binned SAH over ALL objects, threaded, with the big-triangle list kept out of the tree;
//     packets, meta records, cost boxes of the wavefront variant; treelet re-indexing
//   * the same tree built on the device (pt_sahdev.hip / pt_lbvh.hip / pt_widedev.hip): staging, list selection, hand-back
//   * pt_end_obj, pt_upload_triangles
// Compiled with -ffp-contract=off: the encounter ranks replay the reference's float arithmetic (plain x86-64 g++, no fma).
#include "pt_context.hpp"

namespace ptamd {

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


}  // namespace ptamd

extern "C" {

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


}  // extern "C"
