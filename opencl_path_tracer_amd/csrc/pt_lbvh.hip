// pt_lbvh.hip -- BVH construction ON THE DEVICE (the reference builds its tree
// on the host, NodeOnHost::build/convert, main.cpp:195-304, which is what makes scene upload slow
// for large meshes and rules out per-frame rebuilds).
//
// 30-bit Morton codes of the triangle centroids (made unique with the triangle index in the low word), one 64-bit radix
// sort (hipCUB), then the hierarchy: PLOC merges over the Morton order (default; k_ploc_*: nearest neighbour by box area
// within a window, mutual pairs merge, survivors are compacted, repeat) or Karras' parallel radix tree (option lbvh_ploc 0)
// with bottom-up box fitting; then a collapse of every subtree of <= 4 triangles into a leaf and emission in the SAME
// 64-byte node / 48-byte packet layout the host builder produces (pt_internal.hpp) -- the traversal kernels do not know
// which builder made the tree.  The closest hit does not depend on the tree, so renders are
// bit-identical with either builder; only the traversal cost differs (PLOC: 0.64-0.83x of the SAH tree's render rate,
// the Morton radix tree 0.43-0.69x: profiles/r03/t_*).
#include "pt_internal.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>

namespace ptamd {

namespace {

struct Box {
    float lo[3], hi[3];
};

__device__ __forceinline__ int ordered_int(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float from_ordered_int(int i) {
    const int j = i >= 0 ? i : i ^ 0x7fffffff;
    union { int i; float f; } u;
    u.i = j;
    return u.f;
}

// padded bounds of a triangle: identical arithmetic to padded_bounds() of pt_builder.cpp
__device__ __forceinline__ Box tri_bounds(const pt_triangle& t, bool* finite) {
    Box b;
    float m = 0.f;
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
        const float x = t.r1.s[a], y = t.r2.s[a], z = t.r3.s[a];
        ok = ok && isfinite(x) && isfinite(y) && isfinite(z);
        b.lo[a] = fminf(fminf(x, y), z);
        b.hi[a] = fmaxf(fmaxf(x, y), z);
        m = fmaxf(m, fmaxf(fabsf(b.lo[a]), fabsf(b.hi[a])));
    }
    const float pad = m * 1e-5f + 1e-6f;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] -= pad;
        b.hi[a] += pad;
    }
    *finite = ok;
    return b;
}

// (sel: the add-order indices of the triangles that go into the tree -- all but the big-triangle list -- or null for all)
__global__ void __launch_bounds__(256) k_prim_bounds(const pt_triangle* tris, const int32_t* sel, int n, Box* boxes, int* cbounds /*[6] ordered ints*/) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float c[3] = {0.f, 0.f, 0.f};
    bool ok = false;
    if (i < n) {
        Box b = tri_bounds(tris[sel ? sel[i] : i], &ok);
        if (!ok) {                       // cannot be hit (prog.cl:99-106 compares NaN): empty box
            for (int a = 0; a < 3; ++a) { b.lo[a] = __builtin_inff(); b.hi[a] = -__builtin_inff(); }
        }
        boxes[i] = b;
        for (int a = 0; a < 3; ++a) c[a] = 0.5f * (b.lo[a] + b.hi[a]);
    }
    // wave-level reduction of the centroid bounds, then one atomic per wave and component
    for (int a = 0; a < 3; ++a) {
        float lo = ok ? c[a] : __builtin_inff(), hi = ok ? c[a] : -__builtin_inff();
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_down(lo, off, 64));
            hi = fmaxf(hi, __shfl_down(hi, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            if (lo <= hi) {
                atomicMin(&cbounds[a], ordered_int(lo));
                atomicMax(&cbounds[3 + a], ordered_int(hi));
            }
        }
    }
}

__device__ __forceinline__ unsigned expand10(unsigned v) {   // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(256) k_morton(const Box* boxes, int n, const int* cbounds, unsigned long long* keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned code = 0;
    const Box b = boxes[i];
    if (b.lo[0] <= b.hi[0]) {
        unsigned q[3];
        for (int a = 0; a < 3; ++a) {
            const float lo = from_ordered_int(cbounds[a]), hi = from_ordered_int(cbounds[3 + a]);
            const float ext = hi - lo;
            const float c = 0.5f * (b.lo[a] + b.hi[a]);
            float u = ext > 0.f ? (c - lo) / ext : 0.f;
            u = fminf(fmaxf(u * 1024.f, 0.f), 1023.f);
            q[a] = (unsigned)u;
        }
        code = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    }
    keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
}

// ---- Karras 2012: one thread per internal node of the binary radix tree over the sorted keys
struct RadixNode {
    int left, right;     // child index; bit 31 set: leaf (sorted position), else internal node
    int first, last;     // covered range of sorted positions
};
constexpr int kLeafBit = (int)0x80000000;

__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

__global__ void __launch_bounds__(256) k_karras(const unsigned long long* keys, int n, RadixNode* nodes, int* parent_of_internal, int* parent_of_leaf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    RadixNode nd;
    nd.first = lo;
    nd.last = hi;
    if (lo == gamma) { nd.left = gamma | kLeafBit; parent_of_leaf[gamma] = i; }
    else { nd.left = gamma; parent_of_internal[gamma] = i; }
    if (hi == gamma + 1) { nd.right = (gamma + 1) | kLeafBit; parent_of_leaf[gamma + 1] = i; }
    else { nd.right = gamma + 1; parent_of_internal[gamma + 1] = i; }
    nodes[i] = nd;
    if (i == 0) parent_of_internal[0] = -1;
}

// ---- bottom-up boxes: the second thread to arrive at a node owns it (agent-scope fences: the two
// children may have been written by CUs of different XCDs, whose L2s are not coherent)
__global__ void __launch_bounds__(256) k_fit(const unsigned long long* keys, int n, const Box* prim_boxes, const RadixNode* nodes,
                                             const int* parent_of_internal, const int* parent_of_leaf, Box* node_boxes, int* arrivals) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    int cur = parent_of_leaf[k];
    while (cur >= 0) {
        __threadfence();
        const int prev = atomicAdd(&arrivals[cur], 1);
        if (prev == 0) return;                  // the sibling subtree is not finished yet
        __threadfence();
        const RadixNode nd = nodes[cur];
        Box b;
        for (int a = 0; a < 3; ++a) { b.lo[a] = __builtin_inff(); b.hi[a] = -__builtin_inff(); }
        for (int side = 0; side < 2; ++side) {
            const int c = side ? nd.right : nd.left;
            const Box* src = (c & kLeafBit) ? &prim_boxes[(unsigned)(keys[c & ~kLeafBit] & 0xffffffffu)] : &node_boxes[c];
            for (int a = 0; a < 3; ++a) {
                const float lo = __hip_atomic_load(&src->lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float hi = __hip_atomic_load(&src->hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.lo[a] = fminf(b.lo[a], lo);
                b.hi[a] = fmaxf(b.hi[a], hi);
            }
        }
        for (int a = 0; a < 3; ++a) {
            __hip_atomic_store(&node_boxes[cur].lo[a], b.lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&node_boxes[cur].hi[a], b.hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cur = parent_of_internal[cur];
    }
}

// ---- PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) in place of the radix tree: the Morton order is
// only used to find CANDIDATES -- every cluster looks R positions to either side for the partner with which it makes the
// smallest box, mutual choices merge, the survivors close ranks, repeat until one cluster is left.  Splits therefore follow
// box AREA, not bit positions of a space-filling curve: a Morton LBVH of the height-field meshes renders at 0.44-0.6x of the
// SAH tree's rate (curve discontinuities put far-apart triangles under one low node), this at ~0.8-0.9x.
// Internal nodes are numbered downwards from n - 2 in creation order, so the last merge -- the root -- is node 0.
__device__ __forceinline__ float union_half_area(const Box& a, const Box& b) {
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

// nearest neighbour of cluster i among [i - R, i + R]: smallest union area; ties by the pair (min index, max index), which
// is symmetric -- the globally smallest pair always chooses each other, so every round merges at least one pair
template <int R>
__global__ void __launch_bounds__(256) k_ploc_nn(const Box* cbox, int m, int* nn) {
    __shared__ Box tile[256 + 2 * R];
    const int base = blockIdx.x * 256 - R;
    for (int t = threadIdx.x; t < 256 + 2 * R; t += 256) {
        const int j = base + t;
        if (j >= 0 && j < m) tile[t] = cbox[j];
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const Box me = tile[threadIdx.x + R];
    float best = __builtin_inff();
    int bj = -1;
    for (int d = -R; d <= R; ++d) {
        const int j = i + d;
        if (d == 0 || j < 0 || j >= m) continue;
        const float c = union_half_area(me, tile[threadIdx.x + R + d]);
        bool better = c < best;
        if (c == best && bj >= 0) {
            const int a0 = min(i, j), a1 = max(i, j), b0 = min(i, bj), b1 = max(i, bj);
            better = a0 < b0 || (a0 == b0 && a1 < b1);
        }
        if (better) { best = c; bj = j; }
    }
    nn[i] = bj;
}

// leader[i] = 1: i and nn[i] chose each other and i is the lower one (it creates the node); keep[i] = 0 for the upper one
__global__ void __launch_bounds__(256) k_ploc_mark(const int* nn, int m, int* leader, int* keep) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    leader[i] = (mutual && i < j) ? 1 : 0;
    keep[i] = (mutual && i > j) ? 0 : 1;
}

__global__ void __launch_bounds__(256) k_ploc_apply(const int* cid, const Box* cbox, const int* nn, const int* leader, const int* leader_scan,
                                                    const int* keep, const int* keep_scan, int m, int next_id /* index of this round's first new node */,
                                                    RadixNode* nodes, Box* node_boxes, int* parent_of_internal, int* parent_of_leaf,
                                                    int* cid_out, Box* cbox_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m || !keep[i]) return;
    const int p = keep_scan[i];
    if (!leader[i]) {
        cid_out[p] = cid[i];
        cbox_out[p] = cbox[i];
        return;
    }
    const int j = nn[i];
    const int idx = next_id - leader_scan[i];
    const Box a = cbox[i], b = cbox[j];
    Box u;
    for (int k = 0; k < 3; ++k) { u.lo[k] = fminf(a.lo[k], b.lo[k]); u.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
    RadixNode nd;
    nd.left = cid[i];
    nd.right = cid[j];
    nd.first = 0;
    nd.last = 0;
    nodes[idx] = nd;
    node_boxes[idx] = u;
    if (nd.left & kLeafBit) parent_of_leaf[nd.left & ~kLeafBit] = idx; else parent_of_internal[nd.left] = idx;
    if (nd.right & kLeafBit) parent_of_leaf[nd.right & ~kLeafBit] = idx; else parent_of_internal[nd.right] = idx;
    cid_out[p] = idx;
    cbox_out[p] = u;
}

__global__ void __launch_bounds__(256) k_ploc_init(const unsigned long long* keys, const Box* prim_boxes, int n, int* cid, Box* cbox) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    cid[k] = k | kLeafBit;
    cbox[k] = prim_boxes[(unsigned)(keys[k] & 0xffffffffu)];
}

// leaves under every node, bottom-up (the second child to arrive owns the node)
__global__ void __launch_bounds__(256) k_ploc_count(int n, const RadixNode* nodes, const int* parent_of_internal, const int* parent_of_leaf, int* count, int* arrivals) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    int cur = parent_of_leaf[k];
    while (cur >= 0) {
        __threadfence();
        const int prev = atomicAdd(&arrivals[cur], 1);
        if (prev == 0) return;
        __threadfence();
        const RadixNode nd = nodes[cur];
        const int cl = (nd.left & kLeafBit) ? 1 : __hip_atomic_load(&count[nd.left], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int cr = (nd.right & kLeafBit) ? 1 : __hip_atomic_load(&count[nd.right], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&count[cur], cl + cr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur = parent_of_internal[cur];
    }
}

// Depth-first position of every leaf (left subtree first): the leaves in front of it are those of the left siblings along
// its path to the root.  On the way up the leaf also reports itself as the first leaf of every ancestor it reaches through
// left edges only, and as the last of those it reaches through right edges only.
__global__ void __launch_bounds__(256) k_ploc_order(int n, RadixNode* nodes, const int* parent_of_internal, const int* parent_of_leaf, const int* count, int* dfs_pos) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    int me = k | kLeafBit;
    int cur = parent_of_leaf[k];
    int pos = 0;
    while (cur >= 0) {
        const RadixNode nd = nodes[cur];
        if (nd.right == me) pos += (nd.left & kLeafBit) ? 1 : count[nd.left];
        me = cur;
        cur = parent_of_internal[cur];
    }
    dfs_pos[k] = pos;
}
__global__ void __launch_bounds__(256) k_ploc_ranges(int n, RadixNode* nodes, const int* parent_of_internal, const int* parent_of_leaf, const int* dfs_pos) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int pos = dfs_pos[k];
    int me = k | kLeafBit;
    bool all_left = true, all_right = true;
    for (int cur = parent_of_leaf[k]; cur >= 0 && (all_left || all_right); cur = parent_of_internal[cur]) {
        const bool from_left = nodes[cur].left == me;
        all_left = all_left && from_left;
        all_right = all_right && !from_left;
        if (all_left) nodes[cur].first = pos;
        if (all_right) nodes[cur].last = pos;
        me = cur;
    }
}
// sorted keys, leaf parents and leaf references re-expressed in depth-first positions: from here on a node covers the
// contiguous range [first, last] again and the rest of the pipeline (collapse, emit, pack) does not know the difference
__global__ void __launch_bounds__(256) k_ploc_permute(int n, const unsigned long long* keys, const int* parent_of_leaf, const int* dfs_pos,
                                                      unsigned long long* keys_out, int* parent_of_leaf_out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int p = dfs_pos[k];
    keys_out[p] = keys[k];
    parent_of_leaf_out[p] = parent_of_leaf[k];
}
__global__ void __launch_bounds__(256) k_ploc_relabel(int n_internal, RadixNode* nodes, const int* dfs_pos) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_internal) return;
    RadixNode nd = nodes[i];
    if (nd.left & kLeafBit) nd.left = dfs_pos[nd.left & ~kLeafBit] | kLeafBit;
    if (nd.right & kLeafBit) nd.right = dfs_pos[nd.right & ~kLeafBit] | kLeafBit;
    nodes[i] = nd;
}

// a node survives as an interior node of the output iff it covers more than kMaxLeaf triangles.  (Round 3 also tried the
// host builder's SAH termination here -- leaf iff A N <= visit_cost A + cost(children), bottom-up: 73 % more nodes, +2 %
// render rate, +15 ms of host work downstream for 1M triangles: not kept, profiles/r03/t_*.)
__global__ void __launch_bounds__(256) k_flags(const RadixNode* nodes, int n_internal, int* flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_internal) flags[i] = (nodes[i].last - nodes[i].first + 1) > kMaxLeaf ? 1 : 0;
}

// (leaf_base: packets in front of the tree's -- the big-triangle list; a leaf reference is ~(first packet << 3 | count - 1))
__device__ __forceinline__ int child_ref(const RadixNode* nodes, const int* flags, const int* out_index, int c, int leaf_base) {
    if (c & kLeafBit) return ~((((c & ~kLeafBit) + leaf_base) << 3) | 0);
    if (flags[c]) return out_index[c];
    const RadixNode nd = nodes[c];
    return ~(((nd.first + leaf_base) << 3) | (nd.last - nd.first));
}

__global__ void __launch_bounds__(256) k_emit(const unsigned long long* keys, const RadixNode* nodes, int n_internal, const int* flags, const int* out_index,
                                              const Box* prim_boxes, const Box* node_boxes, int leaf_base, Node64* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_internal || !flags[i]) return;
    const RadixNode nd = nodes[i];
    Node64 o;
    for (int side = 0; side < 2; ++side) {
        const int c = side ? nd.right : nd.left;
        const Box b = (c & kLeafBit) ? prim_boxes[(unsigned)(keys[c & ~kLeafBit] & 0xffffffffu)] : node_boxes[c];
        for (int a = 0; a < 3; ++a) {
            o.q[a][side * 2 + 0] = b.lo[a];
            o.q[a][side * 2 + 1] = b.hi[a];
        }
    }
    o.left = child_ref(nodes, flags, out_index, nd.left, leaf_base);
    o.right = child_ref(nodes, flags, out_index, nd.right, leaf_base);
    o.pad[0] = o.pad[1] = 0;
    out[out_index[i]] = o;
}

// depth of the output tree = max number of surviving ancestors of a leaf (+1 for the leaf level)
__global__ void __launch_bounds__(256) k_depth(int n, const int* flags, const int* parent_of_internal, const int* parent_of_leaf, int* max_depth) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    int d = 0;
    if (k < n) {
        for (int cur = parent_of_leaf[k]; cur >= 0; cur = parent_of_internal[cur]) d += flags[cur];
    }
    for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_down(d, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(max_depth, d);
}

__global__ void __launch_bounds__(256) k_pack(const unsigned long long* keys, int n, const pt_triangle* tris, const int32_t* rank, const int32_t* sel,
                                              TriPacket* packets, TriMeta* meta, int32_t* orig) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int j = (int)(unsigned)(keys[k] & 0xffffffffu);
    const int i = sel ? sel[j] : j;                  // add-order index
    const pt_triangle t = tris[i];
    TriPacket p;
    p.v[0] = t.r1.s[0]; p.v[1] = t.r1.s[1]; p.v[2] = t.r1.s[2];
    p.v[3] = t.r2.s[0]; p.v[4] = t.r2.s[1]; p.v[5] = t.r2.s[2];
    p.v[6] = t.r3.s[0]; p.v[7] = t.r3.s[1]; p.v[8] = t.r3.s[2];
    p.v[9] = t.N.s[0]; p.v[10] = t.N.s[1]; p.v[11] = t.N.s[2];
    packets[k] = p;
    TriMeta m;
    m.rank = rank[i];
    m.mati = t.mati;
    meta[k] = m;
    orig[k] = i;
}

#define LB_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) { cleanup_all(); return e_; }                       \
    } while (0)

}  // namespace

// Builds the tree for n (> kMaxLeaf) triangles given in add order on the HOST; every step after the
// upload runs on the device.  On success the caller owns out->* (hipFree).
// d_tris / d_rank: ALL n_all triangles in add order, in device memory (stage_upload); d_sel: the n of them that go into the tree
// (null: all, n == n_all).  The
// packets / meta / orig arrays are emitted for n_all triangles with the tree's behind the first n_all - n slots, which the
// caller fills (the big-triangle list); leaf references count from there.
hipError_t lbvh_build(const pt_triangle* d_tris, const int32_t* d_rank, int n_all, const int32_t* d_sel, int n, int ploc_radius, hipStream_t stream, LbvhResult* out) {
    const int nf = n_all - n;
    Box *d_pbox = nullptr, *d_nbox = nullptr;
    unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
    RadixNode* d_rnodes = nullptr;
    int *d_pint = nullptr, *d_pleaf = nullptr, *d_arr = nullptr, *d_flags = nullptr, *d_oidx = nullptr, *d_misc = nullptr;
    void* d_temp = nullptr;
    // PLOC (ploc_radius > 0): two cluster lists, nearest neighbours, flags + scans, leaf counts, depth-first positions
    int *d_cidA = nullptr, *d_cidB = nullptr, *d_nn = nullptr, *d_leader = nullptr, *d_keep = nullptr, *d_lscan = nullptr, *d_kscan = nullptr;
    int *d_count = nullptr, *d_dfs = nullptr, *d_pleaf2 = nullptr;
    Box *d_cboxA = nullptr, *d_cboxB = nullptr;
    unsigned long long* d_keys3 = nullptr;
    Node64* d_out = nullptr;
    TriPacket* d_packets = nullptr;
    TriMeta* d_meta = nullptr;
    int32_t* d_orig = nullptr;
    auto cleanup = [&]() {
        void* ptrs[] = {d_pbox, d_nbox, d_keys, d_keys2, d_rnodes, d_pint, d_pleaf, d_arr, d_flags, d_oidx, d_misc, d_temp,
                        d_cidA, d_cidB, d_nn, d_leader, d_keep, d_lscan, d_kscan, d_count, d_dfs, d_pleaf2, d_cboxA, d_cboxB, d_keys3};
        for (void* p : ptrs) if (p) (void)hipFree(p);
    };
    auto cleanup_all = [&]() {
        cleanup();
        if (d_out) (void)hipFree(d_out);
        if (d_packets) (void)hipFree(d_packets);
        if (d_meta) (void)hipFree(d_meta);
        if (d_orig) (void)hipFree(d_orig);
    };
    const int ni = n - 1;
    const int blocks_n = (n + 255) / 256, blocks_i = (ni + 255) / 256;
    PhaseClock clk("device bvh");
    LB_HIP(hipMalloc((void**)&d_pbox, sizeof(Box) * (size_t)n));
    LB_HIP(hipMalloc((void**)&d_nbox, sizeof(Box) * (size_t)ni));
    LB_HIP(hipMalloc((void**)&d_keys, sizeof(unsigned long long) * (size_t)n));
    LB_HIP(hipMalloc((void**)&d_keys2, sizeof(unsigned long long) * (size_t)n));
    LB_HIP(hipMalloc((void**)&d_rnodes, sizeof(RadixNode) * (size_t)ni));
    LB_HIP(hipMalloc((void**)&d_pint, sizeof(int) * (size_t)ni));
    LB_HIP(hipMalloc((void**)&d_pleaf, sizeof(int) * (size_t)n));
    LB_HIP(hipMalloc((void**)&d_arr, sizeof(int) * (size_t)ni));
    LB_HIP(hipMalloc((void**)&d_flags, sizeof(int) * (size_t)ni));
    LB_HIP(hipMalloc((void**)&d_oidx, sizeof(int) * (size_t)ni));
    LB_HIP(hipMalloc((void**)&d_misc, sizeof(int) * 8));
    // misc: [0..2] centroid min (ordered ints), [3..5] centroid max, [6] max depth
    int init[8];                                   // (function scope: alive until the copy has run)
    {
        union { float f; int i; } pinf, ninf;
        pinf.f = INFINITY;
        ninf.f = -INFINITY;
        for (int a = 0; a < 3; ++a) { init[a] = pinf.i; init[3 + a] = ninf.i ^ 0x7fffffff; }
        init[6] = 0;
        init[7] = 0;
        LB_HIP(hipMemcpyAsync(d_misc, init, sizeof init, hipMemcpyHostToDevice, stream));
    }
    LB_HIP(hipMemsetAsync(d_arr, 0, sizeof(int) * (size_t)ni, stream));
    hipLaunchKernelGGL(k_prim_bounds, dim3(blocks_n), dim3(256), 0, stream, d_tris, d_sel, n, d_pbox, d_misc);
    hipLaunchKernelGGL(k_morton, dim3(blocks_n), dim3(256), 0, stream, d_pbox, n, d_misc, d_keys);
    LB_HIP(hipGetLastError());
    size_t temp_bytes = 0;
    LB_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d_keys, d_keys2, n, 0, 64, stream));
    size_t scan_bytes = 0;
    LB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_flags, d_oidx, n, stream));
    LB_HIP(hipMalloc(&d_temp, std::max(temp_bytes, scan_bytes) + 256));
    LB_HIP(hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_keys2, n, 0, 64, stream));
    const unsigned long long* keys = d_keys2;
    if (clk.on) { LB_HIP(hipStreamSynchronize(stream)); clk.lap("boxes + morton + sort"); }
    if (ploc_radius > 0) {
        LB_HIP(hipMalloc((void**)&d_cidA, sizeof(int) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_cidB, sizeof(int) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_cboxA, sizeof(Box) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_cboxB, sizeof(Box) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_nn, sizeof(int) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_leader, sizeof(int) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_keep, sizeof(int) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_lscan, sizeof(int) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_kscan, sizeof(int) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_count, sizeof(int) * (size_t)ni));
        LB_HIP(hipMalloc((void**)&d_dfs, sizeof(int) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_pleaf2, sizeof(int) * (size_t)n));
        LB_HIP(hipMalloc((void**)&d_keys3, sizeof(unsigned long long) * (size_t)n));
        LB_HIP(hipMemsetAsync(d_pint, 0xff, sizeof(int) * (size_t)ni, stream));        // the root keeps parent -1
        hipLaunchKernelGGL(k_ploc_init, dim3(blocks_n), dim3(256), 0, stream, keys, d_pbox, n, d_cidA, d_cboxA);
        int m = n, created = 0;
        int *cid = d_cidA, *cid_out = d_cidB;
        Box *cbox = d_cboxA, *cbox_out = d_cboxB;
        int rounds = 0;
        while (m > 1) {
            const int blocks_m = (m + 255) / 256;
            if (ploc_radius <= 8) hipLaunchKernelGGL(k_ploc_nn<8>, dim3(blocks_m), dim3(256), 0, stream, cbox, m, d_nn);
            else if (ploc_radius <= 16) hipLaunchKernelGGL(k_ploc_nn<16>, dim3(blocks_m), dim3(256), 0, stream, cbox, m, d_nn);
            else hipLaunchKernelGGL(k_ploc_nn<32>, dim3(blocks_m), dim3(256), 0, stream, cbox, m, d_nn);
            hipLaunchKernelGGL(k_ploc_mark, dim3(blocks_m), dim3(256), 0, stream, d_nn, m, d_leader, d_keep);
            LB_HIP(hipGetLastError());
            LB_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, scan_bytes, d_leader, d_lscan, m, stream));
            LB_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, scan_bytes, d_keep, d_kscan, m, stream));
            int last_leader = 0, last_scan = 0;
            LB_HIP(hipMemcpyAsync(&last_leader, d_leader + (m - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            LB_HIP(hipMemcpyAsync(&last_scan, d_lscan + (m - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            LB_HIP(hipStreamSynchronize(stream));
            const int merges = last_leader + last_scan;
            if (merges < 1 || ++rounds > 4096) { cleanup_all(); return hipErrorUnknown; }      // (cannot happen: the smallest pair is always mutual)
            hipLaunchKernelGGL(k_ploc_apply, dim3(blocks_m), dim3(256), 0, stream, cid, cbox, d_nn, d_leader, d_lscan, d_keep, d_kscan, m, (ni - 1) - created,
                               d_rnodes, d_nbox, d_pint, d_pleaf, cid_out, cbox_out);
            LB_HIP(hipGetLastError());
            created += merges;
            m -= merges;
            std::swap(cid, cid_out);
            std::swap(cbox, cbox_out);
        }
        if (created != ni) { cleanup_all(); return hipErrorUnknown; }
        if (clk.on) { std::fprintf(stderr, "[device bvh] PLOC rounds %d\n", rounds); clk.lap("PLOC merges"); }
        hipLaunchKernelGGL(k_ploc_count, dim3(blocks_n), dim3(256), 0, stream, n, d_rnodes, d_pint, d_pleaf, d_count, d_arr);
        hipLaunchKernelGGL(k_ploc_order, dim3(blocks_n), dim3(256), 0, stream, n, d_rnodes, d_pint, d_pleaf, d_count, d_dfs);
        hipLaunchKernelGGL(k_ploc_ranges, dim3(blocks_n), dim3(256), 0, stream, n, d_rnodes, d_pint, d_pleaf, d_dfs);
        hipLaunchKernelGGL(k_ploc_permute, dim3(blocks_n), dim3(256), 0, stream, n, keys, d_pleaf, d_dfs, d_keys3, d_pleaf2);
        hipLaunchKernelGGL(k_ploc_relabel, dim3(blocks_i), dim3(256), 0, stream, ni, d_rnodes, d_dfs);
        LB_HIP(hipGetLastError());
        keys = d_keys3;
        std::swap(d_pleaf, d_pleaf2);
    } else {
        hipLaunchKernelGGL(k_karras, dim3(blocks_i), dim3(256), 0, stream, keys, n, d_rnodes, d_pint, d_pleaf);
        hipLaunchKernelGGL(k_fit, dim3(blocks_n), dim3(256), 0, stream, keys, n, d_pbox, d_rnodes, d_pint, d_pleaf, d_nbox, d_arr);
    }
    hipLaunchKernelGGL(k_flags, dim3(blocks_i), dim3(256), 0, stream, d_rnodes, ni, d_flags);
    LB_HIP(hipGetLastError());
    LB_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, scan_bytes, d_flags, d_oidx, ni, stream));
    hipLaunchKernelGGL(k_depth, dim3(blocks_n), dim3(256), 0, stream, n, d_flags, d_pint, d_pleaf, d_misc + 6);
    int last_flag = 0, last_idx = 0, depth = 0;
    LB_HIP(hipMemcpyAsync(&last_flag, d_flags + (ni - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
    LB_HIP(hipMemcpyAsync(&last_idx, d_oidx + (ni - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
    LB_HIP(hipMemcpyAsync(&depth, d_misc + 6, sizeof(int), hipMemcpyDeviceToHost, stream));
    LB_HIP(hipStreamSynchronize(stream));
    const int n_out = last_idx + last_flag;
    if (n_out < 1) { cleanup_all(); return hipErrorUnknown; }
    LB_HIP(hipMalloc((void**)&d_out, sizeof(Node64) * (size_t)n_out));
    LB_HIP(hipMalloc((void**)&d_packets, sizeof(TriPacket) * (size_t)n_all));
    LB_HIP(hipMalloc((void**)&d_meta, sizeof(TriMeta) * (size_t)n_all));
    LB_HIP(hipMalloc((void**)&d_orig, sizeof(int32_t) * (size_t)n_all));
    hipLaunchKernelGGL(k_emit, dim3(blocks_i), dim3(256), 0, stream, keys, d_rnodes, ni, d_flags, d_oidx, d_pbox, d_nbox, nf, d_out);
    hipLaunchKernelGGL(k_pack, dim3(blocks_n), dim3(256), 0, stream, keys, n, d_tris, d_rank, d_sel, d_packets + nf, d_meta + nf, d_orig + nf);
    LB_HIP(hipGetLastError());
    LB_HIP(hipStreamSynchronize(stream));
    clk.lap("order + collapse + emit + pack");
    cleanup();
    clk.lap("free");
    out->d_nodes = reinterpret_cast<float4*>(d_out);
    out->n_nodes = n_out;
    out->d_tris = reinterpret_cast<float4*>(d_packets);
    out->d_meta = d_meta;
    out->d_orig = d_orig;
    out->depth = depth;
    return hipSuccess;
}

}  // namespace ptamd
