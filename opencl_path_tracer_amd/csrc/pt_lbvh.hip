// pt_lbvh.hip -- BVH construction ON THE DEVICE (the reference builds its tree
// on the host, NodeOnHost::build/convert, main.cpp:195-304, which is what makes scene upload slow
// for large meshes and rules out per-frame rebuilds).
//
// Linear BVH: 30-bit Morton codes of the triangle centroids (made unique with the triangle index in
// the low word), one 64-bit radix sort (hipCUB), Karras' parallel radix-tree construction, bottom-up
// box fitting with one arrival counter per node, then a collapse of every subtree of <= 4 triangles
// into a leaf and emission in the SAME 64-byte node / 48-byte packet layout the host builder
// produces (pt_internal.hpp) -- the traversal kernels do not know which builder made the tree.
// The closest hit does not depend on the tree, so renders are bit-identical
// with either builder; only the traversal cost differs (an LBVH is looser than the SAH tree).
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

// padded bounds of a triangle: identical arithmetic to padded_bounds() of pt_host.cpp
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

__global__ void __launch_bounds__(256) k_prim_bounds(const pt_triangle* tris, int n, Box* boxes, int* cbounds /*[6] ordered ints*/) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float c[3] = {0.f, 0.f, 0.f};
    bool ok = false;
    if (i < n) {
        Box b = tri_bounds(tris[i], &ok);
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

// a radix node survives as an interior node of the output iff it covers more than kMaxLeaf triangles
__global__ void __launch_bounds__(256) k_flags(const RadixNode* nodes, int n_internal, int* flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_internal) flags[i] = (nodes[i].last - nodes[i].first + 1) > kMaxLeaf ? 1 : 0;
}

__device__ __forceinline__ int child_ref(const RadixNode* nodes, const int* out_index, int c) {
    if (c & kLeafBit) return ~(((c & ~kLeafBit) << 3) | 0);
    const RadixNode nd = nodes[c];
    const int len = nd.last - nd.first + 1;
    if (len <= kMaxLeaf) return ~((nd.first << 3) | (len - 1));
    return out_index[c];
}

__global__ void __launch_bounds__(256) k_emit(const unsigned long long* keys, const RadixNode* nodes, int n_internal, const int* flags, const int* out_index,
                                              const Box* prim_boxes, const Box* node_boxes, Node64* out) {
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
    o.left = child_ref(nodes, out_index, nd.left);
    o.right = child_ref(nodes, out_index, nd.right);
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

__global__ void __launch_bounds__(256) k_pack(const unsigned long long* keys, int n, const pt_triangle* tris, const int32_t* rank,
                                              TriPacket* packets, TriMeta* meta, int32_t* orig) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int i = (int)(unsigned)(keys[k] & 0xffffffffu);
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
hipError_t lbvh_build(const pt_triangle* h_tris, const int32_t* h_rank, int n, hipStream_t stream, LbvhResult* out) {
    pt_triangle* d_tris = nullptr;
    int32_t* d_rank = nullptr;
    Box *d_pbox = nullptr, *d_nbox = nullptr;
    unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
    RadixNode* d_rnodes = nullptr;
    int *d_pint = nullptr, *d_pleaf = nullptr, *d_arr = nullptr, *d_flags = nullptr, *d_oidx = nullptr, *d_misc = nullptr;
    void* d_temp = nullptr;
    Node64* d_out = nullptr;
    TriPacket* d_packets = nullptr;
    TriMeta* d_meta = nullptr;
    int32_t* d_orig = nullptr;
    auto cleanup = [&]() {
        void* ptrs[] = {d_tris, d_rank, d_pbox, d_nbox, d_keys, d_keys2, d_rnodes, d_pint, d_pleaf, d_arr, d_flags, d_oidx, d_misc, d_temp};
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
    LB_HIP(hipMalloc((void**)&d_tris, sizeof(pt_triangle) * (size_t)n));
    LB_HIP(hipMalloc((void**)&d_rank, sizeof(int32_t) * (size_t)n));
    LB_HIP(hipMemcpyAsync(d_tris, h_tris, sizeof(pt_triangle) * (size_t)n, hipMemcpyHostToDevice, stream));
    LB_HIP(hipMemcpyAsync(d_rank, h_rank, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, stream));
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
    {
        int init[8];
        union { float f; int i; } pinf, ninf;
        pinf.f = INFINITY;
        ninf.f = -INFINITY;
        for (int a = 0; a < 3; ++a) { init[a] = pinf.i; init[3 + a] = ninf.i ^ 0x7fffffff; }
        init[6] = 0;
        init[7] = 0;
        LB_HIP(hipMemcpyAsync(d_misc, init, sizeof init, hipMemcpyHostToDevice, stream));
    }
    LB_HIP(hipMemsetAsync(d_arr, 0, sizeof(int) * (size_t)ni, stream));
    hipLaunchKernelGGL(k_prim_bounds, dim3(blocks_n), dim3(256), 0, stream, d_tris, n, d_pbox, d_misc);
    hipLaunchKernelGGL(k_morton, dim3(blocks_n), dim3(256), 0, stream, d_pbox, n, d_misc, d_keys);
    LB_HIP(hipGetLastError());
    size_t temp_bytes = 0;
    LB_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d_keys, d_keys2, n, 0, 64, stream));
    size_t scan_bytes = 0;
    LB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_flags, d_oidx, ni, stream));
    LB_HIP(hipMalloc(&d_temp, std::max(temp_bytes, scan_bytes) + 256));
    LB_HIP(hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_keys2, n, 0, 64, stream));
    const unsigned long long* keys = d_keys2;
    hipLaunchKernelGGL(k_karras, dim3(blocks_i), dim3(256), 0, stream, keys, n, d_rnodes, d_pint, d_pleaf);
    hipLaunchKernelGGL(k_fit, dim3(blocks_n), dim3(256), 0, stream, keys, n, d_pbox, d_rnodes, d_pint, d_pleaf, d_nbox, d_arr);
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
    LB_HIP(hipMalloc((void**)&d_packets, sizeof(TriPacket) * (size_t)n));
    LB_HIP(hipMalloc((void**)&d_meta, sizeof(TriMeta) * (size_t)n));
    LB_HIP(hipMalloc((void**)&d_orig, sizeof(int32_t) * (size_t)n));
    hipLaunchKernelGGL(k_emit, dim3(blocks_i), dim3(256), 0, stream, keys, d_rnodes, ni, d_flags, d_oidx, d_pbox, d_nbox, d_out);
    hipLaunchKernelGGL(k_pack, dim3(blocks_n), dim3(256), 0, stream, keys, n, d_tris, d_rank, d_packets, d_meta, d_orig);
    LB_HIP(hipGetLastError());
    LB_HIP(hipStreamSynchronize(stream));
    cleanup();
    out->d_nodes = reinterpret_cast<float4*>(d_out);
    out->n_nodes = n_out;
    out->d_tris = reinterpret_cast<float4*>(d_packets);
    out->d_meta = d_meta;
    out->d_orig = d_orig;
    out->depth = depth;
    return hipSuccess;
}

}  // namespace ptamd
