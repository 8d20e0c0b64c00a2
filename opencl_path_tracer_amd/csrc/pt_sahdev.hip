// pt_sahdev.hip -- the host builder's binned-SAH tree (BvhBuilder::split, pt_builder.cpp), built ON THE DEVICE, node for node.
//
// The reference builds its tree on the host (NodeOnHost::build / convert, main.cpp:195-304); so does this library by
// default, and its other device builder (pt_lbvh.hip: Morton order + PLOC merges) is quick but renders 0.6-0.8x as fast as
// the SAH tree.  This file runs the SAME top-down algorithm on the GPU: 16 bins x 3 axes over the centroid bounds of a
// range, the split that minimises area x count, the leaf test against one node visit, a STABLE partition -- with the same
// float expressions in the same order (-ffp-contract=off on both sides; min / max / counts do not depend on grouping), so
// every range makes the decision the host makes and the tree, the packed triangle order and the boxes come out identical
// (tests/test_gpu_parity.py::test_device_sah_builder_same_tree).  Two phases:
//   top    level-synchronous over the open ranges of more than `grain` primitives: per level one pass that bins every
//          primitive of an open range (LDS-privatised for the range a block mostly sits in, global atomics otherwise), one
//          wave per range that evaluates the split and opens the children, and a flag / scan / scatter stable partition of
//          the whole index array at once;
//   bottom one wave per range of at most `grain` primitives builds that subtree depth first (explicit stack in LDS), in its
//          own slice of a node array, in local preorder;
// a walk over the few top nodes on the host numbers everything in preorder and two kernels splice the pieces into the
// final array.  Children's bounds AND centroid bounds come out of the parent's bins (the union over the bins on either side of
// the split), so a range is read once per level.
// What the host builder does with std::nth_element (the median split: a range of more than `max_leaf` coincident
// centroids, or a depth the traversal stack could not take) has no order the device could reproduce: the build reports
// `unsupported` and the caller builds on the host.
#include "pt_internal.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

namespace ptamd {

namespace {

constexpr int NB = 16;                            // BvhBuilder::NB
constexpr int kBinWords = 13;                     // per (axis, bin): box lo xyz / hi xyz, centroid lo xyz / hi xyz (ordered ints), count
constexpr int kBinsWords = 3 * NB * kBinWords;    // 624
constexpr int kTopChunk = 256;                    // positions per workgroup of the binning pass of the top phase
constexpr int kTopSlots = 4;                      // ranges with private bins in LDS per workgroup of that pass
constexpr int kOrdPosInf = 0x7f800000;            // ordered_int(+inf)
constexpr int kOrdNegInf = (int)0x807fffffu;      // ordered_int(-inf)

struct Seg {                 // a range of the index array that still has to be looked at (64 B)
    int lo, hi;
    int depth;
    int slot;                // where its reference goes: node * 2 + side (top nodes or, inside a task, local nodes); -1: the root
    float b[6];              // bounds of the primitives' boxes: lo xyz, hi xyz
    float cb[6];             // bounds of their centroids
};

struct SegBin {              // what binning a primitive of an open range needs (32 B)
    float cblo[3];
    float scale[3];
    int open;                // bit a: the centroids have an extent along axis a
    int pad;
};

struct Decision {            // how an open range of the top phase was split (axis -1: not at all)
    int axis, bin, nl, lo;
    int child[2];            // the children's indices in the next level's list, -1: a task (closed)
    int pad[2];
};

struct PBox {                // a primitive: its padded bounds (the centroid is 0.5f * (lo + hi), as on the host)
    float lo[3], hi[3];
    int pad[2];
};

struct Counters {
    int next_open;           // ranges opened for the next level
    int tasks;
    int tops;
    int bad;                 // 1: a range needs the host's median split; 2: a capacity ran out
    int deepest;
    int root[12];            // bounds of everything (ordered ints): box lo, hi, centroid lo, hi
    int pad[3];
};

__device__ __forceinline__ int ordered_int(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float from_ordered_int(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// padded_bounds() of pt_builder.cpp, the same arithmetic
__device__ __forceinline__ PBox tri_bounds(const pt_triangle& t) {
    PBox b;
    float m = 0.f;
    for (int a = 0; a < 3; ++a) {
        const float x = t.r1.s[a], y = t.r2.s[a], z = t.r3.s[a];
        b.lo[a] = fminf(fminf(x, y), z);
        b.hi[a] = fmaxf(fmaxf(x, y), z);
        m = fmaxf(m, fmaxf(fabsf(b.lo[a]), fabsf(b.hi[a])));
    }
    const float pad = m * 1e-5f + 1e-6f;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] -= pad;
        b.hi[a] += pad;
    }
    b.pad[0] = b.pad[1] = 0;
    return b;
}

__device__ __forceinline__ float half_area6(const float* lo, const float* hi) {      // Aabb::half_area
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ int bin_of(float c, float lo, float scale) {               // BvhBuilder::bin_of
    const int k = (int)((c - lo) * scale);
    return min(max(k, 0), NB - 1);
}

__device__ __forceinline__ SegBin segbin_of(const float* cb) {
    SegBin s;
    s.open = 0;
    for (int a = 0; a < 3; ++a) {
        const float ext = cb[3 + a] - cb[a];
        s.cblo[a] = cb[a];
        s.scale[a] = ext > 0.f ? (float)NB / ext : 0.f;
        if (ext > 0.f) s.open |= 1 << a;
    }
    s.pad = 0;
    return s;
}

template <class T> __device__ __forceinline__ T pick3(int a, T x, T y, T z) { return a == 0 ? x : (a == 1 ? y : z); }   // (no indexed registers)

__device__ __forceinline__ int need_levels(int n) {                                   // BvhBuilder::need_levels
    const int leaves = (n + 3) / 4;
    int l = 0;
    while ((1 << l) < leaves) ++l;
    return l;
}

__device__ __forceinline__ void bins_reset(int* bins, int lane, int stride) {
    for (int w = lane; w < kBinsWords; w += stride) {
        const int j = w % kBinWords;
        bins[w] = j == 12 ? 0 : ((j < 3 || (j >= 6 && j < 9)) ? kOrdPosInf : kOrdNegInf);
    }
}

// one primitive into the bins of its range (LDS or global memory)
__device__ __forceinline__ void bin_prim(int* bins, const PBox& p, const SegBin& sb, int* packed) {
    float c[3];
    for (int a = 0; a < 3; ++a) c[a] = 0.5f * (p.lo[a] + p.hi[a]);
    int pk = 0;
    for (int a = 0; a < 3; ++a) {
        if (!((sb.open >> a) & 1)) continue;
        const int k = bin_of(c[a], sb.cblo[a], sb.scale[a]);
        pk |= k << (4 * a);
        int* w = bins + (a * NB + k) * kBinWords;
        for (int j = 0; j < 3; ++j) {
            atomicMin(w + j, ordered_int(p.lo[j]));
            atomicMax(w + 3 + j, ordered_int(p.hi[j]));
            const int oc = ordered_int(c[j]);
            atomicMin(w + 6 + j, oc);
            atomicMax(w + 9 + j, oc);
        }
        atomicAdd(w + 12, 1);
    }
    *packed = pk;
}

// ---- the split decision of one range, by one wave -------------------------------------------------------------------
// LDS the wave works in: its bins and the two children.
struct WaveLds {
    int bins[kBinsWords];
    float child[2][12];             // per side: box lo, hi, centroid lo, hi
    int nl;                         // primitives on the left side
};

struct SplitOut {
    int axis, bin, nl;              // axis -1: a leaf
    bool bad;                       // the host would split at the median here
};

// lane i takes the value of lane i - D (CTRL = 0x110 + D, row_shr) or i + D (0x100 + D, row_shl) of its row of 16 lanes; a lane
// whose source lies outside the row keeps `old`
template <int CTRL> __device__ __forceinline__ float dpp_f(float old, float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, 0xf, false); }

struct BinAcc {                     // what the host's sweep carries: the box of the bins so far, their centroid bounds, the count
    float lo[3], hi[3], clo[3], chi[3];
    int cnt;
};
template <int CTRL> __device__ __forceinline__ void scan_step(BinAcc& x) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        x.lo[j] = fminf(x.lo[j], dpp_f<CTRL>(x.lo[j], x.lo[j]));
        x.hi[j] = fmaxf(x.hi[j], dpp_f<CTRL>(x.hi[j], x.hi[j]));
        x.clo[j] = fminf(x.clo[j], dpp_f<CTRL>(x.clo[j], x.clo[j]));
        x.chi[j] = fmaxf(x.chi[j], dpp_f<CTRL>(x.chi[j], x.chi[j]));
    }
    x.cnt += dpp_i<CTRL>(0, x.cnt);
}

// All 64 lanes call this with the same arguments; `w.bins` is filled and comes back EMPTY (ready for the next range).  The
// workgroup IS the wave wherever this is called (64 threads), so __syncthreads() is a wave barrier.
// Lane a * 16 + k holds bin k of axis a; the host's prefix / suffix sweeps over the 16 bins of an axis are inclusive scans
// along a row of 16 lanes (min / max / integer sums: any order gives the host's values).
__device__ SplitOut eval_split(WaveLds& w, const Seg& s, int open, int max_leaf, bool force_leaf, float visit_cost) {
    const int lane = threadIdx.x & 63;
    const int n = s.hi - s.lo;
    BinAcc pre;
    {
        int* b = w.bins + min(lane, 3 * NB - 1) * kBinWords;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            pre.lo[j] = from_ordered_int(b[j]);
            pre.hi[j] = from_ordered_int(b[3 + j]);
            pre.clo[j] = from_ordered_int(b[6 + j]);
            pre.chi[j] = from_ordered_int(b[9 + j]);
        }
        pre.cnt = b[12];
        if (lane < 3 * NB) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { b[j] = kOrdPosInf; b[3 + j] = kOrdNegInf; b[6 + j] = kOrdPosInf; b[9 + j] = kOrdNegInf; }
            b[12] = 0;
        }
    }
    BinAcc suf = pre;
    scan_step<0x111>(pre);
    scan_step<0x112>(pre);
    scan_step<0x114>(pre);
    scan_step<0x118>(pre);
    scan_step<0x101>(suf);
    scan_step<0x102>(suf);
    scan_step<0x104>(suf);
    scan_step<0x108>(suf);
    const float la = half_area6(pre.lo, pre.hi), ra = half_area6(suf.lo, suf.hi);
    const float ra_next = dpp_f<0x101>(0.f, ra);                   // the suffix from bin k + 1 on
    const int rc_next = dpp_i<0x101>(0, suf.cnt);
    float cost = __builtin_inff();
    bool valid = false;
    {
        const int a = lane / NB, k = lane % NB;
        if (lane < 3 * NB && k < NB - 1 && ((open >> a) & 1) && pre.cnt != 0 && rc_next != 0) {
            const float t = la * (float)pre.cnt + ra_next * (float)rc_next;
            if (t < __builtin_inff()) { cost = t; valid = true; }      // (the host's `cost < best_cost` never takes inf or NaN)
        }
    }
    float best = cost;
    for (int off = 32; off > 0; off >>= 1) best = fminf(best, __shfl_xor(best, off, 64));
    const unsigned long long who = __ballot(valid && cost == best);       // the first (axis, bin) with the least cost
    SplitOut o;
    o.axis = -1;
    o.bin = -1;
    o.nl = 0;
    o.bad = false;
    int pick = -1;
    if (who != 0ull) pick = __ffsll((long long)who) - 1;
    const float area = half_area6(s.b, s.b + 3);
    const float leaf_cost = area * (float)n;
    const float best_cost = pick >= 0 ? best : __builtin_inff();
    if (n <= max_leaf && (force_leaf || !(best_cost + visit_cost * area < leaf_cost))) return o;      // a leaf
    if (pick < 0) { o.bad = true; return o; }
    // the children's bounds: the union of the bins on either side = the prefix at the split bin, the suffix behind it
    if (lane == pick) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { w.child[0][j] = pre.lo[j]; w.child[0][3 + j] = pre.hi[j]; w.child[0][6 + j] = pre.clo[j]; w.child[0][9 + j] = pre.chi[j]; }
        w.nl = pre.cnt;
    }
    if (lane == pick + 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { w.child[1][j] = suf.lo[j]; w.child[1][3 + j] = suf.hi[j]; w.child[1][6 + j] = suf.clo[j]; w.child[1][9 + j] = suf.chi[j]; }
    }
    __syncthreads();
    o.axis = pick / NB;
    o.bin = pick % NB;
    o.nl = w.nl;
    const int big = max(o.nl, n - o.nl);
    if (s.depth + 1 + need_levels(big) > kMaxDepth) { o.bad = true; o.axis = -1; return o; }
    return o;
}

// the 12 box words of a node from the two children (Node64: q[a] = { Lmin, Lmax, Rmin, Rmax })
__device__ __forceinline__ void write_child_boxes(Node64* nd, const WaveLds& w, int lane) {
    if (lane < 12) {
        const int a = lane >> 2, j = lane & 3;
        nd->q[a][j] = w.child[j >> 1][(j & 1) * 3 + a];
    } else if (lane < 14) {
        nd->pad[lane - 12] = 0;
    }
}

__device__ __forceinline__ Seg child_seg(const WaveLds& w, const Seg& s, const SplitOut& o, int side, int slot) {
    Seg c;
    c.lo = side ? s.lo + o.nl : s.lo;
    c.hi = side ? s.hi : s.lo + o.nl;
    c.depth = s.depth + 1;
    c.slot = slot;
    for (int j = 0; j < 6; ++j) { c.b[j] = w.child[side][j]; c.cb[j] = w.child[side][6 + j]; }
    return c;
}

// ---- setup ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sah_prims(const pt_triangle* tris, const int32_t* sel, int n, PBox* pbox, int* idx, int* owner, Counters* cnt) {
    __shared__ int s_red[12];
    if (threadIdx.x < 12) s_red[threadIdx.x] = (threadIdx.x < 3 || (threadIdx.x >= 6 && threadIdx.x < 9)) ? kOrdPosInf : kOrdNegInf;
    __syncthreads();
    float v[12];
    for (int j = 0; j < 3; ++j) { v[j] = __builtin_inff(); v[3 + j] = -__builtin_inff(); v[6 + j] = __builtin_inff(); v[9 + j] = -__builtin_inff(); }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const PBox b = tri_bounds(tris[sel ? sel[i] : i]);
        pbox[i] = b;
        idx[i] = i;
        owner[i] = 0;
        for (int j = 0; j < 3; ++j) {
            const float c = 0.5f * (b.lo[j] + b.hi[j]);
            v[j] = fminf(v[j], b.lo[j]);
            v[3 + j] = fmaxf(v[3 + j], b.hi[j]);
            v[6 + j] = fminf(v[6 + j], c);
            v[9 + j] = fmaxf(v[9 + j], c);
        }
    }
    // one LDS atomic per wave and component, then one global atomic per block and component
    for (int j = 0; j < 12; ++j) {
        const bool is_min = j < 3 || (j >= 6 && j < 9);
        float x = v[j];
        for (int off = 32; off > 0; off >>= 1) {
            const float y = __shfl_xor(x, off, 64);
            x = is_min ? fminf(x, y) : fmaxf(x, y);
        }
        if ((threadIdx.x & 63) == 0) {
            if (is_min) atomicMin(&s_red[j], ordered_int(x)); else atomicMax(&s_red[j], ordered_int(x));
        }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const bool is_min = threadIdx.x < 3 || (threadIdx.x >= 6 && threadIdx.x < 9);
        if (is_min) atomicMin(&cnt->root[threadIdx.x], s_red[threadIdx.x]); else atomicMax(&cnt->root[threadIdx.x], s_red[threadIdx.x]);
    }
}

// the root range: open if it is bigger than the grain, else the one task
__global__ void k_sah_root(int n, int grain, Counters* cnt, Seg* segs, SegBin* segbins, Seg* tasks) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Seg s;
    s.lo = 0;
    s.hi = n;
    s.depth = 0;
    s.slot = -1;
    for (int j = 0; j < 6; ++j) { s.b[j] = from_ordered_int(cnt->root[j]); s.cb[j] = from_ordered_int(cnt->root[6 + j]); }
    if (n > grain) {
        segs[0] = s;
        segbins[0] = segbin_of(s.cb);
        cnt->next_open = 1;
    } else {
        tasks[0] = s;
        cnt->tasks = 1;
        cnt->next_open = 0;
    }
}

__global__ void __launch_bounds__(256) k_sah_bins_init(int* gbins, int nsegs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nsegs * kBinsWords) return;
    const int j = (i % kBinsWords) % kBinWords;
    gbins[i] = j == 12 ? 0 : ((j < 3 || (j >= 6 && j < 9)) ? kOrdPosInf : kOrdNegInf);
}

// ---- top phase ----------------------------------------------------------------------------------------------------
// Every primitive of an open range into the range's bins; the bin indices are kept for the partition.  One position per thread;
// the open ranges a block's 256 positions touch are runs of consecutive positions, each gets a private copy of its bins in
// LDS (up to kTopSlots of them: ranges are longer than the grain), flushed to the range's bins in global memory at the end.
__global__ void __launch_bounds__(256) k_top_bin(const PBox* pbox, const int* idx, const int* owner, int n, const SegBin* segbins, int* gbins, uint16_t* packed) {
    __shared__ int s_bins[kTopSlots][kBinsWords];
    __shared__ int s_seg[kTopSlots];
    __shared__ int s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = blockIdx.x * kTopChunk + tid;
    const int s = pos < n ? owner[pos] : -1;
    const int prev = (tid == 0 || pos >= n) ? -2 : owner[pos - 1];
    const bool start = s >= 0 && s != prev;                   // the first position of a run in this block
    const unsigned long long m = __ballot(start);
    const int incl = __popcll(m & ((2ull << lane) - 1ull));   // run starts at or before this lane, in this wave
    if (lane == 0) s_wave[wave] = __popcll(m);
    for (int w = tid; w < kTopSlots * kBinsWords; w += 256) {
        const int j = (w % kBinsWords) % kBinWords;
        (&s_bins[0][0])[w] = j == 12 ? 0 : ((j < 3 || (j >= 6 && j < 9)) ? kOrdPosInf : kOrdNegInf);
    }
    __syncthreads();
    int before = 0, runs = 0;
    for (int k = 0; k < 4; ++k) {
        if (k < wave) before += s_wave[k];
        runs += s_wave[k];
    }
    const int slot = before + incl - 1;
    if (start && slot < kTopSlots) s_seg[slot] = s;
    if (s >= 0) {
        const SegBin sb = segbins[s];
        const PBox p = pbox[idx[pos]];
        int pk;
        bin_prim(slot < kTopSlots ? s_bins[slot] : gbins + (size_t)s * kBinsWords, p, sb, &pk);
        packed[pos] = (uint16_t)pk;
    }
    __syncthreads();
    const int used = min(runs, kTopSlots);
    for (int w = tid; w < used * kBinsWords; w += 256) {
        const int sl = w / kBinsWords, word = w % kBinsWords, j = word % kBinWords;
        int* g = gbins + (size_t)s_seg[sl] * kBinsWords + word;
        const int v = s_bins[sl][word];
        if (j == 12) { if (v != 0) atomicAdd(g, v); }
        else if (j < 3 || (j >= 6 && j < 9)) { if (v != kOrdPosInf) atomicMin(g, v); }
        else { if (v != kOrdNegInf) atomicMax(g, v); }
    }
}

// one wave per open range: the split, the top node, the children (open again, or a task)
__global__ void __launch_bounds__(64) k_top_eval(const Seg* segs, const SegBin* segbins, int nsegs, int* gbins, int grain, int max_leaf, int force_leaf, float visit_cost,
                                                 Seg* next_segs, SegBin* next_segbins, int cap_open, Seg* tasks, int cap_tasks, Node64* tops, int* top_link, int cap_tops,
                                                 Decision* dec, Counters* cnt) {
    __shared__ WaveLds w;
    __shared__ int s_ids[3];
    const int si = blockIdx.x, lane = threadIdx.x;
    if (si >= nsegs) return;
    const Seg s = segs[si];
    int* g = gbins + (size_t)si * kBinsWords;
    for (int i = lane; i < kBinsWords; i += 64) w.bins[i] = g[i];
    __syncthreads();
    bins_reset(g, lane, 64);                          // (the next level's range number si starts from empty bins)
    const SplitOut o = eval_split(w, s, segbins[si].open, max_leaf, force_leaf != 0, visit_cost);
    Decision d;
    d.axis = -1;
    d.bin = 0;
    d.nl = 0;
    d.lo = s.lo;
    d.child[0] = d.child[1] = -1;
    d.pad[0] = d.pad[1] = 0;
    if (o.bad || o.axis < 0) {                       // (a leaf cannot be: the range is bigger than the grain >= max_leaf)
        if (lane == 0) { atomicMax(&cnt->bad, 1); dec[si] = d; }
        return;
    }
    const int nc[2] = {o.nl, (s.hi - s.lo) - o.nl};
    if (lane == 0) {
        s_ids[0] = atomicAdd(&cnt->tops, 1);
        for (int side = 0; side < 2; ++side)
            s_ids[1 + side] = nc[side] > grain ? atomicAdd(&cnt->next_open, 1) : -1 - atomicAdd(&cnt->tasks, 1);
        atomicMax(&cnt->deepest, s.depth);
    }
    __syncthreads();
    const int me = s_ids[0];
    bool full = me >= cap_tops;
    for (int side = 0; side < 2; ++side) full = full || (s_ids[1 + side] >= 0 ? s_ids[1 + side] >= cap_open : -1 - s_ids[1 + side] >= cap_tasks);
    if (full) {
        if (lane == 0) { atomicMax(&cnt->bad, 2); dec[si] = d; }
        return;
    }
    write_child_boxes(&tops[me], w, lane);
    if (lane == 0) {
        if (s.slot >= 0) top_link[s.slot] = me;
        for (int side = 0; side < 2; ++side) {
            const Seg c = child_seg(w, s, o, side, me * 2 + side);
            const int id = s_ids[1 + side];
            if (id >= 0) {
                next_segs[id] = c;
                next_segbins[id] = segbin_of(c.cb);
                d.child[side] = id;
            } else {
                tasks[-1 - id] = c;
                top_link[me * 2 + side] = id;         // -1 - task
            }
        }
        d.axis = o.axis;
        d.bin = o.bin;
        d.nl = o.nl;
        dec[si] = d;
    }
}

__global__ void __launch_bounds__(256) k_top_flags(const int* owner, const uint16_t* packed, int n, const Decision* dec, int* flags) {
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const int s = owner[pos];
    int f = 0;
    if (s >= 0) {
        const Decision d = dec[s];
        if (d.axis >= 0) f = (int)((packed[pos] >> (4 * d.axis)) & 15) <= d.bin ? 1 : 0;
    }
    flags[pos] = f;
}

// the stable partition of every split range at once: scan[] = exclusive sum of flags[] over ALL positions
__global__ void __launch_bounds__(256) k_top_scatter(const int* idx, const int* owner, const int* flags, const int* scan, int n, const Decision* dec, int* idx_out, int* owner_out) {
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n) return;
    const int s = owner[pos];
    if (s < 0) {
        idx_out[pos] = idx[pos];
        owner_out[pos] = -1;
        return;
    }
    const Decision d = dec[s];
    if (d.axis < 0) {
        idx_out[pos] = idx[pos];
        owner_out[pos] = -1;
        return;
    }
    const int rank = scan[pos] - scan[d.lo];          // left-goers of this range in front of pos
    const int f = flags[pos];
    const int dst = f ? d.lo + rank : d.lo + d.nl + (pos - d.lo - rank);
    idx_out[dst] = idx[pos];
    owner_out[dst] = d.child[f ? 0 : 1];
}

// ---- bottom phase: one wave per task ---------------------------------------------------------------------------------
constexpr int kStackSegs = kMaxDepth + 4;

__global__ void __launch_bounds__(64) k_sah_bottom(const PBox* pbox, int* idx, int* tmp, const Seg* tasks, int ntasks, int max_leaf, int force_leaf, float visit_cost, int leaf_base,
                                                   Node64* tnodes, int* task_root, int* task_count, Counters* cnt) {
    __shared__ WaveLds w;
    __shared__ Seg stack[kStackSegs];
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= ntasks) return;
    int sp = 0;
    if (lane == 0) {
        stack[0] = tasks[t];
        stack[0].slot = -1;                           // (the task's own slot is a TOP node's: the splice resolves that one)
    }
    sp = 1;
    bins_reset(w.bins, lane, 64);                     // (eval_split hands them back empty)
    __syncthreads();
    Node64* mine = tnodes + stack[0].lo;              // at most n - 1 interior nodes for n primitives: the task's own slice
    int made = 0, deepest = 0;
    bool bad = false;
    while (sp > 0) {
        const Seg s = stack[--sp];
        __syncthreads();                              // (everyone has read the entry before it is overwritten)
        const int n = s.hi - s.lo;
        deepest = max(deepest, s.depth);
        SplitOut o;
        o.axis = -1;
        o.bin = -1;
        o.nl = 0;
        o.bad = false;
        int my_k[3] = {0, 0, 0};                      // (n <= 64: this lane's primitive stays in registers for the partition)
        int my_idx = 0;
        if (n > 1) {
            const SegBin sb = segbin_of(s.cb);
            for (int i = s.lo + lane; i < s.hi; i += 64) {
                my_idx = idx[i];
                const PBox p = pbox[my_idx];
                int pk;
                bin_prim(w.bins, p, sb, &pk);
                my_k[0] = pk & 15;
                my_k[1] = (pk >> 4) & 15;
                my_k[2] = (pk >> 8) & 15;
            }
            __syncthreads();
            o = eval_split(w, s, sb.open, max_leaf, force_leaf != 0, visit_cost);
            if (o.bad) { bad = true; break; }
        }
        int ref;
        if (o.axis < 0) {
            ref = ~(((s.lo + leaf_base) << 3) | (n - 1));
        } else {
            const int me = made++;
            ref = me;
            write_child_boxes(&mine[me], w, lane);
            if (lane == 0) {
                stack[sp] = child_seg(w, s, o, 1, me * 2 + 1);
                stack[sp + 1] = child_seg(w, s, o, 0, me * 2 + 0);
            }
            sp += 2;
            // stable partition of [lo, hi) by `bin on the split axis <= split bin`
            const SegBin sb = segbin_of(s.cb);
            if (n <= 64) {
                const bool valid = lane < n;
                const bool left = valid && pick3(o.axis, my_k[0], my_k[1], my_k[2]) <= o.bin;
                const unsigned long long lm = __ballot(left), vm = __ballot(valid);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int dst = left ? s.lo + __popcll(lm & below) : s.lo + o.nl + __popcll((vm & ~lm) & below);
                if (valid) idx[dst] = my_idx;
            } else {
                int nl = 0, nr = 0;
                for (int base = s.lo; base < s.hi; base += 64) {
                    const int i = base + lane;
                    const bool valid = i < s.hi;
                    int v = 0;
                    bool left = false;
                    if (valid) {
                        v = idx[i];
                        const PBox p = pbox[v];
                        const float c = 0.5f * (pick3(o.axis, p.lo[0], p.lo[1], p.lo[2]) + pick3(o.axis, p.hi[0], p.hi[1], p.hi[2]));
                        left = bin_of(c, pick3(o.axis, sb.cblo[0], sb.cblo[1], sb.cblo[2]), pick3(o.axis, sb.scale[0], sb.scale[1], sb.scale[2])) <= o.bin;
                    }
                    const unsigned long long lm = __ballot(left), vm = __ballot(valid);
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const int dst = left ? s.lo + nl + __popcll(lm & below) : s.lo + o.nl + nr + __popcll((vm & ~lm) & below);
                    if (valid) tmp[dst] = v;
                    nl += __popcll(lm);
                    nr += __popcll(vm & ~lm);
                }
                __syncthreads();
                for (int i = s.lo + lane; i < s.hi; i += 64) idx[i] = tmp[i];
            }
            __syncthreads();
        }
        if (lane == 0) {
            if (s.slot < 0) task_root[t] = ref;
            else if (s.slot & 1) mine[s.slot >> 1].right = ref;
            else mine[s.slot >> 1].left = ref;
        }
    }
    if (lane == 0) {
        task_count[t] = made;
        atomicMax(&cnt->deepest, deepest);
        if (bad) atomicMax(&cnt->bad, 1);
    }
}

// ---- splice: everything into the final preorder ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_splice_tops(const Node64* tops, const int* top_link, int ntops, const int* fin_top, const int* task_off, const int* task_root,
                                                     const int* task_count, Node64* out) {
    const int me = blockIdx.x * 256 + threadIdx.x;
    if (me >= ntops) return;
    Node64 nd = tops[me];
    int ref[2];
    for (int side = 0; side < 2; ++side) {
        const int link = top_link[me * 2 + side];
        if (link >= 0) ref[side] = fin_top[link];
        else {
            const int t = -1 - link;
            ref[side] = task_count[t] > 0 ? task_off[t] : task_root[t];
        }
    }
    nd.left = ref[0];
    nd.right = ref[1];
    nd.pad[0] = nd.pad[1] = 0;
    out[fin_top[me]] = nd;
}

__global__ void __launch_bounds__(64) k_splice_tasks(const Node64* tnodes, const Seg* tasks, int ntasks, const int* task_off, const int* task_count, Node64* out) {
    const int t = blockIdx.x;
    if (t >= ntasks) return;
    const int cnt = task_count[t], off = task_off[t];
    const Node64* src = tnodes + tasks[t].lo;
    for (int j = threadIdx.x; j < cnt; j += 64) {
        Node64 nd = src[j];
        if (nd.left >= 0) nd.left += off;
        if (nd.right >= 0) nd.right += off;
        out[off + j] = nd;
    }
}

__global__ void __launch_bounds__(256) k_sah_pack(const int* idx, int n, const pt_triangle* tris, const int32_t* rank, const int32_t* sel, TriPacket* packets, TriMeta* meta,
                                                  int32_t* orig) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int j = idx[k];
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

#define SD_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) { cleanup_all(); return e_; }                   \
    } while (0)

inline size_t round256(size_t x) { return (x + 255) & ~size_t(255); }

}  // namespace

// Builds the host builder's tree for the n triangles d_sel selects (null: all n_all) of d_tris (device memory: stage_upload).  The packets / meta
// / orig arrays are emitted for n_all triangles with the tree's behind the first n_all - n slots (the big-triangle list, which
// the caller fills); leaf references count from there.  *unsupported is set (and nothing returned) when a range needs the
// host builder's median split.  On success the caller owns out->* (hipFree).
hipError_t sah_device_build(const pt_triangle* d_tris, const int32_t* d_rank, int n_all, const int32_t* d_sel, int n, int max_leaf, bool force_leaf, float visit_cost, int grain,
                            hipStream_t stream, LbvhResult* out, bool* unsupported) {
    *unsupported = false;
    PhaseClock clk("device sah");
    const int nf = n_all - n;
    grain = std::max(grain, std::max(max_leaf, 8));
    const int cap_open = n / grain + 2;
    const int cap_tops = std::min(n, 8 * (n / grain + 2));
    const int cap_tasks = cap_tops + 2;
    // one allocation for everything that is scratch
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += round256(bytes); return at; };
    const size_t o_pbox = carve(sizeof(PBox) * (size_t)n), o_idxA = carve(sizeof(int) * (size_t)n), o_idxB = carve(sizeof(int) * (size_t)n);
    const size_t o_ownA = carve(sizeof(int) * (size_t)n), o_ownB = carve(sizeof(int) * (size_t)n), o_packed = carve(sizeof(uint16_t) * (size_t)n);
    const size_t o_flags = carve(sizeof(int) * (size_t)n), o_scan = carve(sizeof(int) * (size_t)n);
    const size_t o_segA = carve(sizeof(Seg) * (size_t)cap_open), o_segB = carve(sizeof(Seg) * (size_t)cap_open);
    const size_t o_sbA = carve(sizeof(SegBin) * (size_t)cap_open), o_sbB = carve(sizeof(SegBin) * (size_t)cap_open);
    const size_t o_dec = carve(sizeof(Decision) * (size_t)cap_open), o_gbins = carve(sizeof(int) * (size_t)kBinsWords * (size_t)cap_open);
    const size_t o_tasks = carve(sizeof(Seg) * (size_t)cap_tasks), o_troot = carve(sizeof(int) * (size_t)cap_tasks), o_tcount = carve(sizeof(int) * (size_t)cap_tasks);
    const size_t o_toff = carve(sizeof(int) * (size_t)cap_tasks), o_tops = carve(sizeof(Node64) * (size_t)cap_tops), o_link = carve(sizeof(int) * 2 * (size_t)cap_tops);
    const size_t o_fin = carve(sizeof(int) * (size_t)cap_tops), o_tnodes = carve(sizeof(Node64) * (size_t)n), o_cnt = carve(sizeof(Counters));
    size_t scan_bytes = 0;
    char* d_all = nullptr;
    Node64* d_out = nullptr;
    TriPacket* d_packets = nullptr;
    TriMeta* d_meta = nullptr;
    int32_t* d_orig = nullptr;
    auto cleanup = [&]() { if (d_all) (void)hipFree(d_all); d_all = nullptr; };
    auto cleanup_all = [&]() {
        cleanup();
        if (d_out) (void)hipFree(d_out);
        if (d_packets) (void)hipFree(d_packets);
        if (d_meta) (void)hipFree(d_meta);
        if (d_orig) (void)hipFree(d_orig);
    };
    SD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, n, stream));
    const size_t o_temp = carve(scan_bytes + 256);
    SD_HIP(hipMalloc((void**)&d_all, off));
    PBox* d_pbox = reinterpret_cast<PBox*>(d_all + o_pbox);
    int* d_idx = reinterpret_cast<int*>(d_all + o_idxA);
    int* d_idx2 = reinterpret_cast<int*>(d_all + o_idxB);
    int* d_own = reinterpret_cast<int*>(d_all + o_ownA);
    int* d_own2 = reinterpret_cast<int*>(d_all + o_ownB);
    uint16_t* d_packed = reinterpret_cast<uint16_t*>(d_all + o_packed);
    int* d_flags = reinterpret_cast<int*>(d_all + o_flags);
    int* d_scan = reinterpret_cast<int*>(d_all + o_scan);
    Seg* d_seg = reinterpret_cast<Seg*>(d_all + o_segA);
    Seg* d_seg2 = reinterpret_cast<Seg*>(d_all + o_segB);
    SegBin* d_sb = reinterpret_cast<SegBin*>(d_all + o_sbA);
    SegBin* d_sb2 = reinterpret_cast<SegBin*>(d_all + o_sbB);
    Decision* d_dec = reinterpret_cast<Decision*>(d_all + o_dec);
    int* d_gbins = reinterpret_cast<int*>(d_all + o_gbins);
    Seg* d_tasks = reinterpret_cast<Seg*>(d_all + o_tasks);
    int* d_troot = reinterpret_cast<int*>(d_all + o_troot);
    int* d_tcount = reinterpret_cast<int*>(d_all + o_tcount);
    int* d_toff = reinterpret_cast<int*>(d_all + o_toff);
    Node64* d_tops = reinterpret_cast<Node64*>(d_all + o_tops);
    int* d_link = reinterpret_cast<int*>(d_all + o_link);
    int* d_fin = reinterpret_cast<int*>(d_all + o_fin);
    Node64* d_tnodes = reinterpret_cast<Node64*>(d_all + o_tnodes);
    Counters* d_cnt = reinterpret_cast<Counters*>(d_all + o_cnt);
    void* d_temp = d_all + o_temp;

    Counters h_cnt;
    std::memset(&h_cnt, 0, sizeof h_cnt);
    for (int j = 0; j < 12; ++j) h_cnt.root[j] = (j < 3 || (j >= 6 && j < 9)) ? kOrdPosInf : kOrdNegInf;
    SD_HIP(hipMemcpyAsync(d_cnt, &h_cnt, sizeof h_cnt, hipMemcpyHostToDevice, stream));
    const int blocks_n = (n + 255) / 256, blocks_c = (n + kTopChunk - 1) / kTopChunk;
    hipLaunchKernelGGL(k_sah_prims, dim3(std::min(blocks_n, 1024)), dim3(256), 0, stream, d_tris, d_sel, n, d_pbox, d_idx, d_own, d_cnt);
    hipLaunchKernelGGL(k_sah_root, dim3(1), dim3(64), 0, stream, n, grain, d_cnt, d_seg, d_sb, d_tasks);
    hipLaunchKernelGGL(k_sah_bins_init, dim3((cap_open * kBinsWords + 255) / 256), dim3(256), 0, stream, d_gbins, cap_open);
    SD_HIP(hipGetLastError());
    if (clk.on) { SD_HIP(hipStreamSynchronize(stream)); clk.lap("boxes"); }

    int nsegs = n > grain ? 1 : 0, levels = 0;
    while (nsegs > 0) {
        // (next_open restarts from zero every level; tasks / tops / deepest / bad accumulate)
        SD_HIP(hipMemsetAsync(&d_cnt->next_open, 0, sizeof(int), stream));
        hipLaunchKernelGGL(k_top_bin, dim3(blocks_c), dim3(256), 0, stream, d_pbox, d_idx, d_own, n, d_sb, d_gbins, d_packed);
        hipLaunchKernelGGL(k_top_eval, dim3(nsegs), dim3(64), 0, stream, d_seg, d_sb, nsegs, d_gbins, grain, max_leaf, force_leaf ? 1 : 0, visit_cost, d_seg2, d_sb2, cap_open,
                           d_tasks, cap_tasks, d_tops, d_link, cap_tops, d_dec, d_cnt);
        hipLaunchKernelGGL(k_top_flags, dim3(blocks_n), dim3(256), 0, stream, d_own, d_packed, n, d_dec, d_flags);
        SD_HIP(hipGetLastError());
        size_t tb = scan_bytes;
        SD_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_flags, d_scan, n, stream));
        hipLaunchKernelGGL(k_top_scatter, dim3(blocks_n), dim3(256), 0, stream, d_idx, d_own, d_flags, d_scan, n, d_dec, d_idx2, d_own2);
        SD_HIP(hipGetLastError());
        SD_HIP(hipMemcpyAsync(&h_cnt, d_cnt, sizeof(int) * 5, hipMemcpyDeviceToHost, stream));
        SD_HIP(hipStreamSynchronize(stream));
        if (h_cnt.bad) { cleanup_all(); *unsupported = true; return hipSuccess; }
        std::swap(d_idx, d_idx2);
        std::swap(d_own, d_own2);
        std::swap(d_seg, d_seg2);
        std::swap(d_sb, d_sb2);
        nsegs = h_cnt.next_open;
        ++levels;
    }
    SD_HIP(hipMemcpyAsync(&h_cnt, d_cnt, sizeof(int) * 5, hipMemcpyDeviceToHost, stream));
    SD_HIP(hipStreamSynchronize(stream));
    const int ntasks = h_cnt.tasks, ntops = h_cnt.tops;
    if (clk.on) { std::fprintf(stderr, "[device sah] levels %d, top nodes %d, tasks %d\n", levels, ntops, ntasks); clk.lap("top phase"); }
    hipLaunchKernelGGL(k_sah_bottom, dim3(ntasks), dim3(64), 0, stream, d_pbox, d_idx, d_idx2, d_tasks, ntasks, max_leaf, force_leaf ? 1 : 0, visit_cost, nf, d_tnodes, d_troot,
                       d_tcount, d_cnt);
    SD_HIP(hipGetLastError());
    std::vector<int> link((size_t)ntops * 2), tcount((size_t)ntasks), troot((size_t)ntasks);
    if (ntops > 0) SD_HIP(hipMemcpyAsync(link.data(), d_link, sizeof(int) * link.size(), hipMemcpyDeviceToHost, stream));
    SD_HIP(hipMemcpyAsync(tcount.data(), d_tcount, sizeof(int) * tcount.size(), hipMemcpyDeviceToHost, stream));
    SD_HIP(hipMemcpyAsync(troot.data(), d_troot, sizeof(int) * troot.size(), hipMemcpyDeviceToHost, stream));
    SD_HIP(hipMemcpyAsync(&h_cnt, d_cnt, sizeof(int) * 5, hipMemcpyDeviceToHost, stream));
    SD_HIP(hipStreamSynchronize(stream));
    clk.lap("bottom phase");
    if (h_cnt.bad) { cleanup_all(); *unsupported = true; return hipSuccess; }
    // preorder over the top: a top node, then everything under its left child, then under its right child
    std::vector<int> fin((size_t)ntops), toff((size_t)ntasks, 0);
    int total = 0;
    {
        std::vector<int> todo;                        // top node (>= 0) or -1 - task
        todo.push_back(ntops > 0 ? 0 : -1);
        while (!todo.empty()) {
            const int x = todo.back();
            todo.pop_back();
            if (x >= 0) {
                fin[(size_t)x] = total++;
                todo.push_back(link[(size_t)x * 2 + 1]);
                todo.push_back(link[(size_t)x * 2 + 0]);
            } else {
                toff[(size_t)(-1 - x)] = total;
                total += tcount[(size_t)(-1 - x)];
            }
        }
    }
    if (total < 1) { cleanup_all(); *unsupported = true; return hipSuccess; }      // the whole scene is one leaf: the host wraps that
    SD_HIP(hipMalloc((void**)&d_out, sizeof(Node64) * (size_t)total));
    SD_HIP(hipMalloc((void**)&d_packets, sizeof(TriPacket) * (size_t)n_all));
    SD_HIP(hipMalloc((void**)&d_meta, sizeof(TriMeta) * (size_t)n_all));
    SD_HIP(hipMalloc((void**)&d_orig, sizeof(int32_t) * (size_t)n_all));
    if (ntops > 0) SD_HIP(hipMemcpyAsync(d_fin, fin.data(), sizeof(int) * fin.size(), hipMemcpyHostToDevice, stream));
    SD_HIP(hipMemcpyAsync(d_toff, toff.data(), sizeof(int) * toff.size(), hipMemcpyHostToDevice, stream));
    if (ntops > 0)
        hipLaunchKernelGGL(k_splice_tops, dim3((ntops + 255) / 256), dim3(256), 0, stream, d_tops, d_link, ntops, d_fin, d_toff, d_troot, d_tcount, d_out);
    hipLaunchKernelGGL(k_splice_tasks, dim3(ntasks), dim3(64), 0, stream, d_tnodes, d_tasks, ntasks, d_toff, d_tcount, d_out);
    hipLaunchKernelGGL(k_sah_pack, dim3(blocks_n), dim3(256), 0, stream, d_idx, n, d_tris, d_rank, d_sel, d_packets + nf, d_meta + nf, d_orig + nf);
    SD_HIP(hipGetLastError());
    SD_HIP(hipStreamSynchronize(stream));
    clk.lap("splice + pack");
    cleanup();
    clk.lap("free");
    out->d_nodes = reinterpret_cast<float4*>(d_out);
    out->n_nodes = total;
    out->d_tris = reinterpret_cast<float4*>(d_packets);
    out->d_meta = d_meta;
    out->d_orig = d_orig;
    out->depth = h_cnt.deepest;
    return hipSuccess;
}


// ---- staging for the device builders: the triangles in device memory, and the big-triangle list chosen there ----------------
namespace {

__global__ void __launch_bounds__(256) k_stage_area(const pt_triangle* tris, int n, float* area, int* nonfinite) {
    int bad = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const pt_triangle t = tris[i];
        bool ok = true;
        for (int a = 0; a < 3; ++a) ok = ok && isfinite(t.r1.s[a]) && isfinite(t.r2.s[a]) && isfinite(t.r3.s[a]);
        const PBox b = tri_bounds(t);
        area[i] = half_area6(b.lo, b.hi);
        if (!ok) bad = 1;
    }
    if (__ballot(bad != 0) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(nonfinite, 1);
}

// bounds of all triangles but the `cand` listed ones (ordered ints in box[0..5], preset to +inf / -inf)
__global__ void __launch_bounds__(256) k_stage_rest_box(const pt_triangle* tris, int n, const int32_t* top, int cand, int* box) {
    __shared__ int s_top[32];
    __shared__ int s_red[6];
    if (threadIdx.x < 32) s_top[threadIdx.x] = (int)threadIdx.x < cand ? top[threadIdx.x] : -1;
    if (threadIdx.x < 6) s_red[threadIdx.x] = threadIdx.x < 3 ? kOrdPosInf : kOrdNegInf;
    __syncthreads();
    float v[6] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        bool listed = false;
        for (int k = 0; k < cand; ++k) listed = listed || s_top[k] == i;
        if (listed) continue;
        const PBox b = tri_bounds(tris[i]);
        for (int a = 0; a < 3; ++a) {
            v[a] = fminf(v[a], b.lo[a]);
            v[3 + a] = fmaxf(v[3 + a], b.hi[a]);
        }
    }
    for (int j = 0; j < 6; ++j) {
        float x = v[j];
        for (int off = 32; off > 0; off >>= 1) {
            const float y = __shfl_xor(x, off, 64);
            x = j < 3 ? fminf(x, y) : fmaxf(x, y);
        }
        if ((threadIdx.x & 63) == 0) {
            if (j < 3) atomicMin(&s_red[j], ordered_int(x)); else atomicMax(&s_red[j], ordered_int(x));
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        if (threadIdx.x < 3) atomicMin(&box[threadIdx.x], s_red[threadIdx.x]); else atomicMax(&box[threadIdx.x], s_red[threadIdx.x]);
    }
}

// sel[k] = add-order index of the k-th triangle that is NOT on the list (flat: sorted ascending)
__global__ void __launch_bounds__(256) k_stage_select(int ns, const int32_t* flat, int nf, int32_t* sel) {
    __shared__ int s_flat[32];
    if (threadIdx.x < 32) s_flat[threadIdx.x] = (int)threadIdx.x < nf ? flat[threadIdx.x] : 0x7fffffff;
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns) return;
    int j = 0;
    while (j < nf && s_flat[j] - j <= k) ++j;
    sel[k] = k + j;
}

}  // namespace

hipError_t stage_upload(const pt_triangle* h_tris, const int32_t* h_rank, int n, hipStream_t stream, DeviceStage* st, float* h_area, int* nonfinite) {
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += round256(bytes); return at; };
    const size_t o_tris = carve(sizeof(pt_triangle) * (size_t)n), o_rank = carve(sizeof(int32_t) * (size_t)n), o_sel = carve(sizeof(int32_t) * (size_t)n);
    const size_t o_area = carve(sizeof(float) * (size_t)n), o_misc = carve(sizeof(int) * 48);
    char* base = nullptr;
    hipError_t e = hipMalloc((void**)&base, off);
    if (e != hipSuccess) return e;
    st->base = base;
    st->d_tris = reinterpret_cast<pt_triangle*>(base + o_tris);
    st->d_rank = reinterpret_cast<int32_t*>(base + o_rank);
    st->d_sel = reinterpret_cast<int32_t*>(base + o_sel);
    st->d_area = reinterpret_cast<float*>(base + o_area);
    st->d_misc = reinterpret_cast<int*>(base + o_misc);
    st->n = n;
    int init[48];
    for (int j = 0; j < 48; ++j) init[j] = 0;
    for (int j = 0; j < 3; ++j) { init[8 + j] = kOrdPosInf; init[11 + j] = kOrdNegInf; }
    if ((e = hipMemcpyAsync(st->d_tris, h_tris, sizeof(pt_triangle) * (size_t)n, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(st->d_rank, h_rank, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(st->d_misc, init, sizeof init, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_stage_area, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, stream, st->d_tris, n, st->d_area, st->d_misc);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(h_area, st->d_area, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(nonfinite, st->d_misc, sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    return hipStreamSynchronize(stream);
}

// bounds (lo xyz, hi xyz) of all staged triangles except the `cand` (<= 32) listed in h_top
hipError_t stage_rest_box(const DeviceStage& st, const int32_t* h_top, int cand, hipStream_t stream, float box[6]) {
    hipError_t e;
    int32_t* d_top = reinterpret_cast<int32_t*>(st.d_misc + 16);
    if (cand > 0 && (e = hipMemcpyAsync(d_top, h_top, sizeof(int32_t) * (size_t)cand, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_stage_rest_box, dim3(std::min((st.n + 255) / 256, 1024)), dim3(256), 0, stream, st.d_tris, st.n, d_top, cand, st.d_misc + 8);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    int ord[6];
    if ((e = hipMemcpyAsync(ord, st.d_misc + 8, sizeof ord, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    for (int j = 0; j < 6; ++j) {
        const int i = ord[j] >= 0 ? ord[j] : ord[j] ^ 0x7fffffff;
        std::memcpy(&box[j], &i, sizeof(float));
    }
    return hipSuccess;
}

// fills st.d_sel with the add-order indices of the triangles that are not on the (ascending) list
hipError_t stage_select(const DeviceStage& st, const int32_t* h_flat, int nf, hipStream_t stream) {
    hipError_t e;
    int32_t* d_flat = reinterpret_cast<int32_t*>(st.d_misc + 16);
    if (nf > 0 && (e = hipMemcpyAsync(d_flat, h_flat, sizeof(int32_t) * (size_t)nf, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    const int ns = st.n - nf;
    hipLaunchKernelGGL(k_stage_select, dim3((ns + 255) / 256), dim3(256), 0, stream, ns, d_flat, nf, st.d_sel);
    return hipGetLastError();
}

void stage_free(DeviceStage* st) {
    if (st->base) (void)hipFree(st->base);
    *st = DeviceStage();
}

}  // namespace ptamd
