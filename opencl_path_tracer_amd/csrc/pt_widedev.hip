// pt_widedev.hip -- build_wide_nodes() of pt_wide.cpp ON THE DEVICE: the packed BVH2 collapsed into 4-wide nodes with child
// boxes quantised to 8 bits per plane, the same nodes in the same places (tests/test_gpu_parity.py::
// test_device_sah_builder_same_tree compares them byte for byte).  For trees that were built on the device
// (pt_sahdev.hip, pt_lbvh.hip) this was the largest piece of host work left in pt_upload_triangles: 21-27 ms for 1M triangles.
//
// The host version is a breadth-first pass over the top of the tree until 512 subtrees are open, then a depth-first
// collapse of each subtree into a block of its own.  Which BVH2 nodes become 4-wide nodes, and what their children are,
// only depends on the BVH2 (make_wide_node: open the interior child with the largest box until there are four); WHERE a
// node goes follows from the sizes of the subtrees.  So here:
//   1. expansion, level by level in breadth-first order: every 4-wide node finds its children; an exclusive scan over the
//      number of interior children gives the next level its places (= the order the host's FIFO would visit them in);
//   2. bottom-up: the number of 4-wide descendants of every node;
//   3. the host's numbering in closed form: a node visited before the FIFO held 512 open subtrees keeps its breadth-first
//      number; those 512 subtrees are the blocks, laid down one after the other; inside a block a node's interior children
//      take the next free numbers when the node is visited and the LAST child is visited first (the host's LIFO), so child j
//      starts allocating at  F + m + sum of the descendants of the children after j;
//   4. every node is quantised and written at its number (independent threads).
#include "pt_internal.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

namespace ptamd {

namespace {

constexpr int kFrontier = 512;        // build_wide_nodes: open subtrees at which the breadth-first pass stops

struct WKid {
    float lo[3], hi[3];
    int ref;
};

__device__ __forceinline__ WKid kid_of(const Node64& nd, int side) {
    WKid k;
    for (int a = 0; a < 3; ++a) {
        k.lo[a] = nd.q[a][side * 2];
        k.hi[a] = nd.q[a][side * 2 + 1];
    }
    k.ref = side ? nd.right : nd.left;
    return k;
}
__device__ __forceinline__ bool empty_box(const WKid& k) { return !(k.lo[0] <= k.hi[0] && k.lo[1] <= k.hi[1] && k.lo[2] <= k.hi[2]); }
__device__ __forceinline__ double half_area_d(const WKid& k) {
    const double dx = (double)k.hi[0] - k.lo[0], dy = (double)k.hi[1] - k.lo[1], dz = (double)k.hi[2] - k.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ void put(WKid kids[4], int at, const WKid& v) {       // kids[at] = v without indexed registers
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j == at) kids[j] = v;
}

// make_wide_node(): the children of the 4-wide node that BVH2 node `src` becomes, in the host's order
__device__ int open_kids(const Node64* bvh2, int src, WKid kids[4]) {
    int n = 0;
    {
        const Node64 nd = bvh2[src];
        const WKid l = kid_of(nd, 0), r = kid_of(nd, 1);
        if (!empty_box(l)) put(kids, n++, l);
        if (!empty_box(r)) put(kids, n++, r);
    }
    while (n < 4) {
        int pick = -1;
        double best = -1.0;
        int pick_ref = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n && kids[j].ref >= 0) {
                const double h = half_area_d(kids[j]);
                if (h > best) { best = h; pick = j; pick_ref = kids[j].ref; }
            }
        if (pick < 0) break;
        const Node64 nd = bvh2[pick_ref];
        const WKid l = kid_of(nd, 0), r = kid_of(nd, 1);
        const bool le = !empty_box(l), re = !empty_box(r);
        if (!le && !re) {                     // kids[pick] = kids[--n]
            --n;
            WKid last = kids[0];
#pragma unroll
            for (int j = 1; j < 4; ++j)
                if (j == n) last = kids[j];
            put(kids, pick, last);
            continue;
        }
        put(kids, pick, le ? l : r);
        if (le && re) put(kids, n++, r);
    }
    return n;
}

__device__ __forceinline__ float step_of(int biased_exp) { return __uint_as_float((unsigned)biased_exp << 23); }

// quantise_axis() of pt_wide.cpp
__device__ bool quantise_axis(const WKid kids[4], int n, int a, float* origin, int* biased_exp, unsigned* qlo_out, unsigned* qhi_out) {
    float o = __builtin_inff(), top = -__builtin_inff();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < n) {
            const float lo = a == 0 ? kids[k].lo[0] : (a == 1 ? kids[k].lo[1] : kids[k].lo[2]);
            const float hi = a == 0 ? kids[k].hi[0] : (a == 1 ? kids[k].hi[1] : kids[k].hi[2]);
            o = fminf(o, lo);
            top = fmaxf(top, hi);
        }
    if (!isfinite(o) || !isfinite(top)) return false;
    const double ext = (double)top - (double)o;
    int e = 1;
    if (ext > 0.0) e = max(1, min(254, (int)ilogb(ext / 255.0) + 127));
    for (; e <= 254; ++e) {
        const float step = step_of(e);
        bool ok = true;
        unsigned ql = 0, qh = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= n || !ok) continue;
            const float klo = a == 0 ? kids[k].lo[0] : (a == 1 ? kids[k].lo[1] : kids[k].lo[2]);
            const float khi = a == 0 ? kids[k].hi[0] : (a == 1 ? kids[k].hi[1] : kids[k].hi[2]);
            long long lo = (long long)floor(((double)klo - (double)o) / (double)step);
            lo = max(0ll, min(255ll, lo));
            while (lo > 0 && fmaf((float)lo, step, o) > klo) --lo;
            if (fmaf((float)lo, step, o) > klo) { ok = false; continue; }
            long long hi = (long long)ceil(((double)khi - (double)o) / (double)step);
            hi = max(0ll, hi);
            while (hi <= 255 && fmaf((float)hi, step, o) < khi) ++hi;
            if (hi > 255) { ok = false; continue; }
            ql |= (unsigned)lo << (8 * k);
            qh |= (unsigned)hi << (8 * k);
        }
        if (ok) {
            *origin = o;
            *biased_exp = e;
            *qlo_out = ql;
            *qhi_out = qh;
            return true;
        }
    }
    return false;
}

struct WideArrays {
    int* src;          // BVH2 node of a 4-wide node (breadth-first index)
    int* nk;           // number of children
    int* m;            // number of interior children
    int* first;        // breadth-first index of the first interior child (= entries the host's FIFO holds when it reaches this node)
    int* desc;         // 4-wide descendants
    int* id;           // final index
    int* F;            // inside a block: the next free index when the node is visited
    int* pend;         // stack entries a traversal can hold when it visits the node
    int* kref;         // [4] child references (BVH2)
    float* kbox;       // [4][6] child boxes
};

// step 1a: the children of the nodes [begin, begin + count)
__global__ void __launch_bounds__(256) k_wide_expand(const Node64* bvh2, WideArrays w, int begin, int count, int* mlevel) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const int i = begin + t;
    WKid kids[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        for (int a = 0; a < 3; ++a) { kids[j].lo[a] = 0.f; kids[j].hi[a] = 0.f; }
        kids[j].ref = kWideNoChild;
    }
    const int n = open_kids(bvh2, w.src[i], kids);
    int m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w.kref[i * 4 + j] = j < n ? kids[j].ref : kWideNoChild;
        for (int a = 0; a < 3; ++a) {
            w.kbox[(i * 4 + j) * 6 + a] = kids[j].lo[a];
            w.kbox[(i * 4 + j) * 6 + 3 + a] = kids[j].hi[a];
        }
        if (j < n && kids[j].ref >= 0) ++m;
    }
    w.nk[i] = n;
    w.m[i] = m;
    mlevel[t] = m;
}

// step 1b: the next level's nodes, in the order of their parents (scan = exclusive sum of mlevel)
__global__ void __launch_bounds__(256) k_wide_children(WideArrays w, int begin, int count, const int* scan, int next_begin, int* next_count) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const int i = begin + t;
    const int first = next_begin + scan[t];
    w.first[i] = first;
    int j = 0;
    for (int k = 0; k < 4; ++k) {
        const int ref = w.kref[i * 4 + k];
        if (k < w.nk[i] && ref >= 0) {
            w.src[first + j] = ref;
            w.pend[first + j] = w.pend[i] + w.nk[i] - 1;
            ++j;
        }
    }
    if (t == count - 1) *next_count = scan[t] + w.m[i];
}

// step 2 (levels in reverse): 4-wide descendants
__global__ void __launch_bounds__(256) k_wide_desc(WideArrays w, int begin, int count) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const int i = begin + t;
    int d = 0;
    for (int j = 0; j < w.m[i]; ++j) d += 1 + w.desc[w.first[i] + j];
    w.desc[i] = d;
}

// step 3a: where the host's breadth-first pass stops, and the deepest stack
__global__ void __launch_bounds__(256) k_wide_head(WideArrays w, int total, int* head, int* max_pending) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    int p = 0;
    if (i < total) {
        if (w.first[i] - i >= kFrontier) atomicMin(head, i);
        p = w.pend[i];
        w.id[i] = i;                        // (nodes numbered breadth first keep this; the others are overwritten in step 3b)
    }
    for (int off = 32; off > 0; off >>= 1) p = max(p, __shfl_xor(p, off, 64));
    if ((threadIdx.x & 63) == 0 && p > 0) atomicMax(max_pending, p);
}

// step 3b (levels top-down): numbers inside the blocks
__global__ void __launch_bounds__(256) k_wide_number(WideArrays w, int begin, int count, int head) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const int i = begin + t;
    if (i < head) return;                   // visited breadth first: its children keep their breadth-first numbers
    const int m = w.m[i], first = w.first[i], F = w.F[i];
    int after = 0;                          // descendants of the children behind child j
    for (int j = m - 1; j >= 0; --j) {
        w.id[first + j] = F + j;
        w.F[first + j] = F + m + after;
        after += w.desc[first + j];
    }
}

// step 4: quantise and write
__global__ void __launch_bounds__(256) k_wide_emit(WideArrays w, int total, Node4q* out, int* failed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    WKid kids[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        for (int a = 0; a < 3; ++a) {
            kids[j].lo[a] = w.kbox[(i * 4 + j) * 6 + a];
            kids[j].hi[a] = w.kbox[(i * 4 + j) * 6 + 3 + a];
        }
        kids[j].ref = w.kref[i * 4 + j];
    }
    const int n = w.nk[i];
    Node4q nd;
    nd.spare[0] = nd.spare[1] = 0;
    unsigned qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    int e[3] = {1, 1, 1};
    for (int a = 0; a < 3; ++a) nd.origin[a] = 0.f;
    bool ok = true;
    if (n > 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
            ok = ok && quantise_axis(kids, n, a, &nd.origin[a], &e[a], &qlo[a], &qhi[a]);
    }
    if (!ok) { atomicMax(failed, 1); return; }
    for (int a = 0; a < 3; ++a) nd.exp[a] = (uint8_t)e[a];
    nd.nchild = (uint8_t)n;
    int j = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= n) {                           // no child: an inverted box and the harmless reference
            nd.ref[k] = kWideNoChild;
            for (int a = 0; a < 3; ++a) {
                qlo[a] |= 255u << (8 * k);
                qhi[a] &= ~(255u << (8 * k));
            }
        } else if (kids[k].ref >= 0) {
            nd.ref[k] = w.id[w.first[i] + j];
            ++j;
        } else {
            nd.ref[k] = kids[k].ref;
        }
    }
    nd.qlo_x = qlo[0];
    nd.qhi_x = qhi[0];
    nd.qlo_y = qlo[1];
    nd.qhi_y = qhi[1];
    nd.qlo_z = qlo[2];
    nd.qhi_z = qhi[2];
    out[w.id[i]] = nd;
}

#define WD_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) { cleanup(); return e_; }                       \
    } while (0)

inline size_t round256(size_t x) { return (x + 255) & ~size_t(255); }

}  // namespace

// d_bvh2: the packed BVH2 (n_nodes Node64, root 0) in device memory.  On success *d_out holds *n_out 4-wide nodes (the caller
// frees it) and *max_pending what build_wide_nodes() reports; *failed: a box could not be quantised (non-finite), nothing is
// returned and the caller keeps the BVH2 path, as with the host version.
hipError_t wide_device_build(const float4* d_bvh2, int n_nodes, hipStream_t stream, float4** d_out, int* n_out, int* max_pending, bool* failed) {
    *failed = false;
    *d_out = nullptr;
    *n_out = 0;
    *max_pending = 0;
    PhaseClock clk("device 4-wide");
    const Node64* bvh2 = reinterpret_cast<const Node64*>(d_bvh2);
    const size_t cap = (size_t)n_nodes;                    // a 4-wide node per BVH2 node at most
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += round256(bytes); return at; };
    const size_t o_int = carve(sizeof(int) * cap * 8), o_kref = carve(sizeof(int) * cap * 4), o_kbox = carve(sizeof(float) * cap * 24);
    const size_t o_ml = carve(sizeof(int) * cap), o_sc = carve(sizeof(int) * cap), o_misc = carve(sizeof(int) * 8);
    size_t scan_bytes = 0;
    char* d_all = nullptr;
    Node4q* d_nodes4 = nullptr;
    auto cleanup = [&]() {
        if (d_all) (void)hipFree(d_all);
        if (d_nodes4) (void)hipFree(d_nodes4);
        d_all = nullptr;
        d_nodes4 = nullptr;
    };
    WD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, (int)cap, stream));
    const size_t o_temp = carve(scan_bytes + 256);
    WD_HIP(hipMalloc((void**)&d_all, off));
    WideArrays w;
    int* ints = reinterpret_cast<int*>(d_all + o_int);
    w.src = ints;
    w.nk = ints + cap;
    w.m = ints + 2 * cap;
    w.first = ints + 3 * cap;
    w.desc = ints + 4 * cap;
    w.id = ints + 5 * cap;
    w.F = ints + 6 * cap;
    w.pend = ints + 7 * cap;
    w.kref = reinterpret_cast<int*>(d_all + o_kref);
    w.kbox = reinterpret_cast<float*>(d_all + o_kbox);
    int* d_ml = reinterpret_cast<int*>(d_all + o_ml);
    int* d_sc = reinterpret_cast<int*>(d_all + o_sc);
    int* d_misc = reinterpret_cast<int*>(d_all + o_misc);          // [0] next level's count, [1] head, [2] max pending, [3] failed
    void* d_temp = d_all + o_temp;
    const int init[8] = {0, 0x7fffffff, 0, 0, 0, 0, 0, 0};       // (function scope: alive until the copies have run)
    const int zero = 0;
    WD_HIP(hipMemcpyAsync(d_misc, init, sizeof init, hipMemcpyHostToDevice, stream));
    WD_HIP(hipMemcpyAsync(w.src, &zero, sizeof(int), hipMemcpyHostToDevice, stream));           // node 0 = BVH2 root
    WD_HIP(hipMemcpyAsync(w.pend, &zero, sizeof(int), hipMemcpyHostToDevice, stream));
    // 1. expansion
    std::vector<int> level_begin;
    int begin = 0, count = 1;
    while (count > 0) {
        level_begin.push_back(begin);
        if ((size_t)begin + (size_t)count > cap) { cleanup(); return hipErrorUnknown; }
        const int blocks = (count + 255) / 256;
        hipLaunchKernelGGL(k_wide_expand, dim3(blocks), dim3(256), 0, stream, bvh2, w, begin, count, d_ml);
        size_t tb = scan_bytes;
        WD_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_ml, d_sc, count, stream));
        hipLaunchKernelGGL(k_wide_children, dim3(blocks), dim3(256), 0, stream, w, begin, count, d_sc, begin + count, d_misc);
        WD_HIP(hipGetLastError());
        int next = 0;
        WD_HIP(hipMemcpyAsync(&next, d_misc, sizeof(int), hipMemcpyDeviceToHost, stream));
        WD_HIP(hipStreamSynchronize(stream));
        begin += count;
        count = next;
    }
    const int total = begin;
    level_begin.push_back(total);
    const int levels = (int)level_begin.size() - 1;
    clk.lap("expansion");
    // 2. descendants, 3a. head
    for (int l = levels - 1; l >= 0; --l) {
        const int c = level_begin[(size_t)l + 1] - level_begin[(size_t)l];
        hipLaunchKernelGGL(k_wide_desc, dim3((c + 255) / 256), dim3(256), 0, stream, w, level_begin[(size_t)l], c);
    }
    hipLaunchKernelGGL(k_wide_head, dim3((total + 255) / 256), dim3(256), 0, stream, w, total, d_misc + 1, d_misc + 2);
    WD_HIP(hipGetLastError());
    int misc[4];
    WD_HIP(hipMemcpyAsync(misc, d_misc, sizeof misc, hipMemcpyDeviceToHost, stream));
    WD_HIP(hipStreamSynchronize(stream));
    const int head = std::min(misc[1], total);
    *max_pending = misc[2];
    // 3b. the blocks: subtrees head .. T-1 in that order behind the breadth-first part
    std::vector<int> d, F;           // function scope: F is the source of an asynchronous copy and must outlive it
    if (head < total) {
        int T = 0;
        WD_HIP(hipMemcpy(&T, w.first + head, sizeof(int), hipMemcpyDeviceToHost));
        const int nt = T - head;
        d.resize((size_t)nt);
        F.resize((size_t)nt);
        WD_HIP(hipMemcpy(d.data(), w.desc + head, sizeof(int) * (size_t)nt, hipMemcpyDeviceToHost));
        int at = T;
        for (int t = 0; t < nt; ++t) { F[(size_t)t] = at; at += d[(size_t)t]; }
        if (at != total) { cleanup(); return hipErrorUnknown; }
        WD_HIP(hipMemcpyAsync(w.F + head, F.data(), sizeof(int) * (size_t)nt, hipMemcpyHostToDevice, stream));
        for (int l = 0; l < levels; ++l) {
            const int b = level_begin[(size_t)l], c = level_begin[(size_t)l + 1] - b;
            if (b + c <= head) continue;
            hipLaunchKernelGGL(k_wide_number, dim3((c + 255) / 256), dim3(256), 0, stream, w, b, c, head);
        }
    }
    clk.lap("numbering");
    // 4. the nodes
    WD_HIP(hipMalloc((void**)&d_nodes4, sizeof(Node4q) * (size_t)total));
    hipLaunchKernelGGL(k_wide_emit, dim3((total + 255) / 256), dim3(256), 0, stream, w, total, d_nodes4, d_misc + 3);
    WD_HIP(hipGetLastError());
    int bad = 0;
    WD_HIP(hipMemcpyAsync(&bad, d_misc + 3, sizeof(int), hipMemcpyDeviceToHost, stream));
    WD_HIP(hipStreamSynchronize(stream));
    clk.lap("quantise + write");
    if (bad) {
        cleanup();
        *failed = true;
        return hipSuccess;
    }
    (void)hipFree(d_all);
    *d_out = reinterpret_cast<float4*>(d_nodes4);
    *n_out = total;
    return hipSuccess;
}

}  // namespace ptamd
