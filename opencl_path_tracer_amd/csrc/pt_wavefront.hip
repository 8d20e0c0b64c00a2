// pt_wavefront.hip -- the stream-compacted ("wavefront") formulation of the same path
// (BASELINE north_star; option variant = 1).  Device helpers: pt_device.hpp.
#include "pt_device.hpp"

#include <algorithm>

namespace ptamd {

// ============================================================================ wavefront
// Stream-compacted formulation of the same path (BASELINE north_star): one pass = one sample of
// every local pixel.  generate -> for each bounce { intersect ; shade } with the path state SoA in
// HBM (WfParams) and index queues between the stages.
//   wf_generate : 2 LCG draws + camera ray per pixel (prog.cl:384-389), state init (prog.cl:307-316)
//   wf_intersect: each wave owns 256 consecutive entries of a ray queue; it runs the big-triangle list over
//                 them with all lanes busy, then traverses in trips of while-while rounds, refilling the
//                 lanes whose traversal ended between trips (__ballot/__popcll rank inside the wave's
//                 range) while the stragglers carry on.  At the end of its range the block
//                 compacts its rays into three class queues by the material type they hit
//                 (order-preserving ballot scan through LDS, 3 global atomics per 1,024 rays).
//   wf_shade    : one block row per class -> waves are material-coherent.  The ray, factor_L and the LCG state ride the
//                 position-addressed stream (48 B read + 48 B written per surviving ray, coalesced); per PIXEL a hit moves only
//                 what its material touches (PathInHbm): mirror fS, dielectric fR, a specular lobe fB, an emitter reads what
//                 was ever written and adds to the colour.  Survivors go to the next bounce's ray queues; paths that
//                 end (miss / last bounce) fold their colour into the running mean (prog.cl:379) and store their LCG state.
// Ray queues come in two COST classes: a ray that misses the bounding boxes of every complex
// object (more than 16 triangles) can only hit the few large triangles around them and finishes
// in a handful of steps; mixing it into a wave with rays that walk a 1,000-triangle object leaves
// its lane idle for most of the wave's life (measured: 16 % lane utilisation in the node loop).
PT_DEV unsigned long long lanemask_lt() {
    const unsigned lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// Order-preserving slot reservation: every thread with cls in [0, NCLS) gets the next free
// position of stream/queue `cls` (count at counters[cls]); returns it, or ~0u.  Every thread of the
// block must call it.  scratch: NCLS*(WAVES+1) words.
template <int NCLS, int BLOCK>
PT_DEV unsigned block_reserve(int cls, unsigned* counters, unsigned* scratch) {
    constexpr int WAVES = BLOCK / 64;
    const unsigned wave = threadIdx.x >> 6;
    const unsigned long long lt = lanemask_lt();
    unsigned myoff = 0;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if ((threadIdx.x & 63) == 0) scratch[c * (WAVES + 1) + wave] = (unsigned)__popcll(m);
        if (cls == c) myoff = (unsigned)__popcll(m & lt);
    }
    __syncthreads();
    if (threadIdx.x < NCLS) {
        unsigned* row = scratch + threadIdx.x * (WAVES + 1);
        unsigned tot = 0;
        for (int k = 0; k < WAVES; ++k) { const unsigned v = row[k]; row[k] = tot; tot += v; }
        row[WAVES] = tot ? atomicAdd(&counters[threadIdx.x], tot) : 0u;
    }
    __syncthreads();
    unsigned pos = ~0u;
    if (cls >= 0 && cls < NCLS) {
        const unsigned* row = scratch + cls * (WAVES + 1);
        pos = row[WAVES] + row[wave] + myoff;
    }
    __syncthreads();
    return pos;
}

// 1 = the ray touches the box of a complex object (expensive traversal ahead), 0 = it cannot
PT_DEV int ray_cost_class(const WfParams& w, f3 P, f3 D) {
    const f3 inv = mk(__builtin_amdgcn_rcpf(D.x), __builtin_amdgcn_rcpf(D.y), __builtin_amdgcn_rcpf(D.z));
    int cost = 0;
    for (int b = 0; b < w.n_cbox; ++b) {
        const float x0 = (w.cbox[b][0] - P.x) * inv.x, x1 = (w.cbox[b][3] - P.x) * inv.x;
        const float y0 = (w.cbox[b][1] - P.y) * inv.y, y1 = (w.cbox[b][4] - P.y) * inv.y;
        const float z0 = (w.cbox[b][2] - P.z) * inv.z, z1 = (w.cbox[b][5] - P.z) * inv.z;
        const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
        const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * 1.0000005f;
        if (tf >= tn && tf >= 0.0f) cost = 1;
    }
    return cost;
}

PT_DEV void wf_finalize(const WfParams& w, int li, f3 color) {
    f3 acc = mk(0.0f, 0.0f, 0.0f);
    if (w.sample != 0) {
        const float4 c = w.rp.colors[li];
        acc = mk(c.x, c.y, c.z);
    }
    acc = running_mean(acc, color, w.sample);
    w.rp.colors[li] = make_float4(acc.x, acc.y, acc.z, 0.0f);
}

// The four path factors and the colour of pixel li in HBM, sP[field][li] x 12 B, behind the accessors shade_hit uses
// (PathRegs, pt_device.hpp).  `flags` travels with the ray: bit f = field f was written in this sample -- a field that was
// not is its initial value (1, or 0 for the colour) and costs no read; nothing is initialised when a sample starts.
struct PathInHbm {
    float* base;         // &sP[0][li]
    size_t fstride;      // floats between fields
    unsigned flags;
    f3 fL;               // factor_L: rides the ray stream (rsC), not sP
    PT_DEV f3 get(int f, float init) const {
        f3 v = mk(init, init, init);
        if (flags & (1u << f)) {
            const float* q = base + (size_t)f * fstride;
            v = mk(q[0], q[1], q[2]);
        }
        return v;
    }
    PT_DEV void put(int f, f3 v) {
        float* q = base + (size_t)f * fstride;
        q[0] = v.x;
        q[1] = v.y;
        q[2] = v.z;
        flags |= 1u << f;
    }
    PT_DEV f3 L() const { return fL; }
    // factor_B: a value that is +0 in all three components (what ks = 0 makes of it at the first diffuse hit, for good) is a
    // flag, not a record -- the reader gets the very bits that would have been stored
    PT_DEV f3 B() const { return (flags & (unsigned)kWfBZeroBit) ? mk(0.f, 0.f, 0.f) : get(kWfB, 1.f); }
    PT_DEV f3 S() const { return get(kWfS, 1.f); }
    PT_DEV f3 R() const { return get(kWfR, 1.f); }
    PT_DEV f3 C() const { return get(kWfC, 0.f); }
    PT_DEV void setL(f3 v) { fL = v; }
    PT_DEV void setB(f3 v) {
        if ((__float_as_int(v.x) | __float_as_int(v.y) | __float_as_int(v.z)) == 0) {
            flags = (flags | (unsigned)kWfBZeroBit) & ~(1u << kWfB);
        } else {
            flags &= ~(unsigned)kWfBZeroBit;
            put(kWfB, v);
        }
    }
    PT_DEV void setS(f3 v) { put(kWfS, v); }
    PT_DEV void setR(f3 v) { put(kWfR, v); }
    PT_DEV void setC(f3 v) { put(kWfC, v); }
};

// The same for K items per thread (item k of every thread before item k + 1): ONE atomic per class and block, whatever K.
// A returning atomic on one address completes at ~88 per microsecond on MI355X: wf_generate with one pixel per thread and
// 256-thread blocks (8,100 blocks at 1080p, two counters) spent its whole 99 us waiting for them (profiles/r03/h_*).
// scratch: NCLS * (K * WAVES + 1) words.
template <int NCLS, int BLOCK, int K>
PT_DEV void block_reserve_multi(const int (&cls)[K], unsigned (&pos)[K], unsigned* counters, unsigned* scratch) {
    constexpr int WAVES = BLOCK / 64;
    constexpr int ROW = K * WAVES + 1;
    const unsigned wave = threadIdx.x >> 6;
    const unsigned long long lt = lanemask_lt();
    unsigned myoff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        myoff[k] = 0;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            const unsigned long long m = __ballot(cls[k] == c);
            if ((threadIdx.x & 63) == 0) scratch[c * ROW + k * WAVES + wave] = (unsigned)__popcll(m);
            if (cls[k] == c) myoff[k] = (unsigned)__popcll(m & lt);
        }
    }
    __syncthreads();
    if (threadIdx.x < NCLS) {
        unsigned* row = scratch + threadIdx.x * ROW;
        unsigned tot = 0;
        for (int i = 0; i < K * WAVES; ++i) { const unsigned v = row[i]; row[i] = tot; tot += v; }
        row[K * WAVES] = tot ? atomicAdd(&counters[threadIdx.x], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        pos[k] = ~0u;
        if (cls[k] >= 0 && cls[k] < NCLS) {
            const unsigned* row = scratch + cls[k] * ROW;
            pos[k] = row[K * WAVES] + row[k * WAVES + wave] + myoff[k];
        }
    }
    __syncthreads();
}

constexpr int kWfGenBlock = 1024, kWfGenPerThread = 4;

__global__ void __launch_bounds__(kWfGenBlock) wf_generate(WfParams w) {
    __shared__ unsigned s_scratch[2 * (kWfGenPerThread * (kWfGenBlock / 64) + 1)];
    const RenderParams& p = w.rp;
    // rows >= 1 (bounces >= 1) are cleared here; row 0 (bounce 0, filled by THIS launch) is cleared by
    // a memset the host enqueues in front of the kernel
    const int ci = blockIdx.x * kWfGenBlock + threadIdx.x;
    if (ci >= kWfCounterStride && ci < (p.iterations + 3) * kWfCounterStride) w.counters[ci] = 0u;
    int cost[kWfGenPerThread];
    f3 P[kWfGenPerThread], D[kWfGenPerThread];
    int lis[kWfGenPerThread];
    int seeds[kWfGenPerThread];
    unsigned live = 0;
#pragma unroll
    for (int k = 0; k < kWfGenPerThread; ++k) {
        const int idx = (blockIdx.x * kWfGenPerThread + k) * kWfGenBlock + threadIdx.x;      // pixel of the chain
        const int li = w.pix0 + idx;
        lis[k] = li;
        cost[k] = -1;
        seeds[k] = 0;
        P[k] = mk(0.f, 0.f, 0.f);
        D[k] = mk(0.f, 0.f, 1.f);
        if (idx < w.npix) {
            ++live;
            const int lrow = li / p.width, x = li - lrow * p.width;
            const int grow = ((lrow / p.rows_per_block) * p.world + p.rank) * p.rows_per_block + (lrow % p.rows_per_block);
            const int gid = grow * p.width + x;
            int seed = p.rnds[li];
            const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
            camera_get_ray(gid, p.cam, rnd1, rnd2, &P[k], &D[k]);
            seeds[k] = seed;
            if (p.iterations <= 0) {
                wf_finalize(w, li, mk(0.0f, 0.0f, 0.0f));
                p.rnds[li] = seed;
            } else {
                cost[k] = ray_cost_class(w, P[k], D[k]);
            }
        }
    }
    if (p.stats) {
        unsigned long long n = wave_sum((unsigned long long)live);
        if ((threadIdx.x & 63) == 0 && n) stat_add(p, 1, n);
    }
    unsigned pos[kWfGenPerThread];
    block_reserve_multi<2, kWfGenBlock, kWfGenPerThread>(cost, pos, w.counters + kWfGenRow * kWfCounterStride, s_scratch);
#pragma unroll
    for (int k = 0; k < kWfGenPerThread; ++k)
        if (cost[k] >= 0) {
            w.rsA[0][cost[k]][pos[k]] = make_float4(P[k].x, P[k].y, P[k].z, D[k].x);
            w.rsB[0][cost[k]][pos[k]] = make_float4(D[k].y, D[k].z, __int_as_float(lis[k]), 0.0f);
            w.rsC[0][cost[k]][pos[k]] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(seeds[k]));      // factor_L = 1 (prog.cl:307), LCG state
        }
}

// Rays per wave: each wave owns a contiguous range of the bounce's ray stream (no global atomics
// on the fetch side; blocks that finish early are replaced by the dispatcher).
#ifndef PT_WF_RPW
#define PT_WF_RPW 256
#endif
#ifndef PT_WF_RPW_LDS
#define PT_WF_RPW_LDS 128
#endif
constexpr int kWfRaysPerWave = PT_WF_RPW;
constexpr int kWfRaysPerWaveLds = PT_WF_RPW_LDS;      // the 768-thread instance (whole tree in LDS)
constexpr int kWfSuspendLanes = 48;     // (round 4, with the phase switching of Trav::round: 24 / 36 / 48 / 56 all within 1 %)     // 8 / 16 / 32 / 48 -> 749 / 752 / 763 / 773 Msamples/s (flat loop 700)

// Traversal with lane refill: a trip = while-while rounds until most lanes are done; a lane whose ray is
// finished takes the next ray of the wave's range before the next trip (the next ray's 32 B are prefetched one
// assignment ahead, so the switch costs no memory round trip); the stragglers simply carry on.
// WPS = waves per SIMD the register budget is set for, RPW = rays per wave and trip (instances: launch_wf_intersect)
template <int MODE, int BLOCK, int WPS, int RPW>
__global__ void __launch_bounds__(BLOCK, WPS) wf_intersect(WfParams w, int bounce) {
    typedef typename StackOf<MODE>::type StackT;
    constexpr int WAVES = BLOCK / 64;
    constexpr int RPB = WAVES * RPW;                     // rays per block and trip
    constexpr int CHUNKS = RPB / 64;
    const RenderParams& p = w.rp;
    const int cost = blockIdx.y;
    unsigned* ctr = w.counters + wf_row(bounce) * kWfCounterStride;
    const unsigned n = ctr[cost];
    if (blockIdx.x * RPB >= n) return;                    // uniform for the whole block
    // dynamic LDS: [traversal stacks][staged nodes][class byte per ray of the trip][CHUNKS x 3 counts][3 bases]
    LaneStack<StackT> stk;
    SceneView sv;
    setup_traversal<MODE, BLOCK>(p, &sv, &stk);
    unsigned char* lds_cls = pt_lds_raw + traversal_lds_bytes_dev<MODE, BLOCK>(p);
    unsigned* lds_cnt = reinterpret_cast<unsigned*>(lds_cls + RPB);     // [CHUNKS][3]
    unsigned* lds_base = lds_cnt + CHUNKS * 3;                          // [3]
    const float4* __restrict__ rsA = w.rsA[bounce & 1][cost];
    const float4* __restrict__ rsB = w.rsB[bounce & 1][cost];
    float2* __restrict__ hits = w.hit[cost];
    const unsigned long long lt = lanemask_lt();
    // wave-uniform loop state (wave index, the cursor cbase / cend of the wave's ray range) is kept in scalar registers
    // -- readfirstlane at every redefinition -- so that it can never be spilled lane by lane under a partial exec mask
    // (the hang class of profiles/r02/v_*; tools/check_isa.py checks every instance)
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WorkCount wc;
    // the grid only fills the chip (the nodes are staged once per block, not once per trip); a block's first trip is
    // block_base = blockIdx.x * RPB, every further one is fetched from a counter (word 5 / 6 of the bounce's row), so that
    // a block whose rays were cheap takes over from one whose rays were not (static striding left 1.7 trips per block with
    // nothing to balance them: profiles/r03/b_*)
    __shared__ unsigned s_next_base;
    // ... and INSIDE a trip the block's waves share its RPB rays dynamically: a wave takes the next 64 of them from a cursor in
    // LDS whenever its own range has run dry, so the waves of a block finish a trip together whatever their rays cost (with a
    // fixed 128 rays per wave the trip ended on its slowest wave, eleven others waiting at the barrier below)
    __shared__ unsigned s_cursor;
    if (threadIdx.x == 0) s_cursor = 0;
    __syncthreads();
    unsigned block_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * RPB));
    const unsigned first_dynamic = gridDim.x * RPB;
    for (;;) {
        const unsigned nblock = (unsigned)__builtin_amdgcn_readfirstlane((int)min((unsigned)RPB, n - block_base));
        unsigned cbase = 0, cend = 0;        // uniform per wave: the unassigned rest of the range the wave holds
        bool exhausted = false;              // the block's trip has no ray left to hand out
        Trav<MODE> tr;
        tr.begin(mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f), stk);
        tr.idle();
        unsigned pos = ~0u;          // stream position of the ray in flight (~0u: none)
        unsigned npos = ~0u;         // prefetched next ray (~0u: none)
        float4 nA = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 nB = make_float2(0.f, 1.f);
        float2 nH = make_float2(0.f, 0.f);
        for (;;) {
            // ---- the wave's range has run dry: take the next 64 rays of the block's trip
            if (cbase >= cend && !exhausted) {
                unsigned off = 0;
                if ((threadIdx.x & 63) == 0) off = atomicAdd(&s_cursor, 64u);
                off = (unsigned)__builtin_amdgcn_readfirstlane((int)off);
                if (off < nblock) {
                    cbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(block_base + off));
                    cend = (unsigned)__builtin_amdgcn_readfirstlane((int)(block_base + min(off + 64u, nblock)));
                    // the big-triangle list first, for all 64 with every lane busy: the hit record of a ray starts out as its
                    // closest hit among them (per refilled lane it would run for the few lanes that switch rays together).  The
                    // lanes are in mid-traversal: what a resumed traversal cannot recompute from its ray is set aside
                    if (sv.n_flat > 0) {
                        const f3 sP_ = tr.P, sD_ = tr.D;
                        const float s_bt = tr.best_t;
                        const int s_b = tr.best, s_cur = tr.cur, s_pend = tr.pend;
                        char* const s_tos = tr.tos;
                        const unsigned r = cbase + (threadIdx.x & 63);
                        if (r < cend) {
                            const float4 A = rsA[r];
                            const float2 Bq = *reinterpret_cast<const float2*>(&rsB[r]);
                            tr.setup(mk(A.x, A.y, A.z), mk(A.w, Bq.x, Bq.y));
                            tr.best_t = __builtin_inff();
                            tr.best = -1;
                            tr.template flat_pass<false>(sv, &wc);
                            hits[r] = make_float2(tr.best_t, __int_as_float(tr.best));
                        }
                        __threadfence_block();       // the records are read back by other lanes of this wave
                        tr.setup(sP_, sD_);
                        tr.best_t = s_bt;
                        tr.best = s_b;
                        tr.cur = s_cur;
                        tr.pend = s_pend;
                        tr.tos = s_tos;
                    }
                } else {
                    exhausted = true;
                }
            }
            // ---- lanes whose ray is finished switch to their prefetched ray
            if (tr.done() && npos != ~0u) {
                pos = npos;
                npos = ~0u;
                tr.begin(mk(nA.x, nA.y, nA.z), mk(nA.w, nB.x, nB.y), stk);
                if (sv.n_flat > 0) {
                    tr.best_t = nH.x;
                    tr.best = __float_as_int(nH.y);
                }
            }
            // ---- lanes without a prefetched ray reserve the next positions of the wave's range
            const unsigned long long want = __ballot(npos == ~0u);
            if (want != 0 && cbase < cend) {
                const unsigned my = cbase + (unsigned)__popcll(want & lt);
                if (npos == ~0u && my < cend) {
                    npos = my;
                    nA = rsA[my];
                    nB = *reinterpret_cast<const float2*>(&rsB[my]);
                    if (sv.n_flat > 0) nH = hits[my];
                }
                cbase = (unsigned)__builtin_amdgcn_readfirstlane((int)min(cbase + (unsigned)__popcll(want), cend));
            }
            if (__ballot(!tr.done() || npos != ~0u) == 0) {
                if (exhausted) break;
                continue;                    // nothing in flight, but the block may still have rays: take the next 64
            }
            // ---- while-while rounds until at most kWfSuspendLanes lanes are unfinished (and one has finished): the
            // megakernel's tail suspension, with the refill from the ray stream above in the place of shading.
            // (A flat loop -- one node visit and one triangle test per lane and iteration -- ran both bodies every
            // iteration: intersect-only rate 1,256 -> 1,501 Msamples/s-equivalent with rounds, profiles/r02/u_*.)
            for (;;) {
                tr.template round<false>(sv, &wc);
                const unsigned long long unfinished = __ballot(!tr.done());
                if (unfinished == 0) break;
                if (__popcll(unfinished) <= kWfSuspendLanes && __ballot(tr.done() && pos != ~0u) != 0) break;
            }
            // ---- finished: hit record + class byte
            if (tr.done() && pos != ~0u) {
                hits[pos] = make_float2(tr.best_t, __int_as_float(tr.best));
                int cls = 2;
                if (tr.best >= 0) {
                    const int type = p.mats[sv.meta[tr.best].mati].type;
                    cls = (type == 0 || type == 3) ? 0 : 1;
                }
                lds_cls[pos - block_base] = (unsigned char)cls;
                pos = ~0u;
            }
        }
        // ---- order-preserving compaction of the trip's rays into the three class queues
        __syncthreads();
        unsigned off[RPB / BLOCK];
        int cl[RPB / BLOCK];
#pragma unroll
        for (int k = 0; k < RPB / BLOCK; ++k) {
            const unsigned r = k * BLOCK + threadIdx.x;
            cl[k] = r < nblock ? (int)lds_cls[r] : -1;
            off[k] = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned long long m = __ballot(cl[k] == c);
                if ((threadIdx.x & 63) == 0) lds_cnt[(r >> 6) * 3 + c] = (unsigned)__popcll(m);
                if (cl[k] == c) off[k] = (unsigned)__popcll(m & lt);
            }
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            unsigned tot = 0;
            for (int ch = 0; ch < CHUNKS; ++ch) { const unsigned v = lds_cnt[ch * 3 + threadIdx.x]; lds_cnt[ch * 3 + threadIdx.x] = tot; tot += v; }
            lds_base[threadIdx.x] = tot ? atomicAdd(&ctr[2 + threadIdx.x], tot) : 0u;
        }
        if (threadIdx.x == 0 && p.stats) stat_add(p, 0, (unsigned long long)nblock);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPB / BLOCK; ++k) {
            const unsigned r = k * BLOCK + threadIdx.x;
            if (cl[k] >= 0) w.q_cls[cl[k]][lds_base[cl[k]] + lds_cnt[(r >> 6) * 3 + cl[k]] + off[k]] = (int)(((unsigned)cost << 31) | (block_base + r));
        }
        if (wave == 0) {            // (the fetched range is wave-uniform state: through scalar registers, as everywhere)
            unsigned t = 0;
            if ((threadIdx.x & 63) == 0) t = atomicAdd(&ctr[5 + cost], (unsigned)RPB);
            t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
            if ((threadIdx.x & 63) == 0) { s_next_base = first_dynamic + t; s_cursor = 0; }
        }
        __syncthreads();            // (also: lds_cls / lds_cnt are rewritten by the next trip)
        block_base = (unsigned)__builtin_amdgcn_readfirstlane((int)s_next_base);
        if (block_base >= n) break;
    }
}

constexpr int kWfShadeBlock = 1024, kWfShadePerThread = 1;      // (two rays per thread: 100 -> 109 us per launch, profiles/r03/h_*)

__global__ void __launch_bounds__(kWfShadeBlock) wf_shade(WfParams w, int bounce) {
    __shared__ unsigned s_scratch[2 * (kWfShadePerThread * (kWfShadeBlock / 64) + 1)];
    const RenderParams& p = w.rp;
    const int cls = blockIdx.y;
    unsigned* ctr = w.counters + wf_row(bounce) * kWfCounterStride;
    const unsigned n = ctr[2 + cls];
    constexpr unsigned per_block = kWfShadeBlock * kWfShadePerThread;
    if (blockIdx.x * per_block >= n) return;              // whole block idle
    int li[kWfShadePerThread];
    int cost[kWfShadePerThread];                          // >= 0: the path continues with a ray of that cost class
    unsigned flags[kWfShadePerThread];
    int seeds[kWfShadePerThread];
    f3 rP[kWfShadePerThread], rD[kWfShadePerThread], fLs[kWfShadePerThread];
#pragma unroll
    for (int k = 0; k < kWfShadePerThread; ++k) {
        const unsigned i = (blockIdx.x * kWfShadePerThread + k) * kWfShadeBlock + threadIdx.x;
        li[k] = 0;
        cost[k] = -1;
        flags[k] = 0;
        seeds[k] = 0;
        fLs[k] = mk(1.f, 1.f, 1.f);
        rP[k] = mk(0.f, 0.f, 0.f);
        rD[k] = mk(0.f, 0.f, 1.f);
        if (i < n) {
            const unsigned e = (unsigned)w.q_cls[cls][i];
            const int c_in = (int)(e >> 31);
            const unsigned pos = e & 0x7fffffffu;
            const float4 A = w.rsA[bounce & 1][c_in][pos], B = w.rsB[bounce & 1][c_in][pos], Cq = w.rsC[bounce & 1][c_in][pos];
            li[k] = __float_as_int(B.z);
            PathInHbm st;
            st.base = w.sP + (size_t)li[k] * 3;
            st.fstride = (size_t)w.npix_all * 3;
            st.flags = (unsigned)__float_as_int(B.w);
            st.fL = mk(Cq.x, Cq.y, Cq.z);
            int seed = __float_as_int(Cq.w);
            if (cls == 2) {                                   // miss: black environment, prog.cl:367-376
                wf_finalize(w, li[k], st.C());
                p.rnds[li[k]] = seed;
            } else {
                const float2 h = w.hit[c_in][pos];
                rP[k] = mk(A.x, A.y, A.z);
                rD[k] = mk(A.w, B.x, B.y);
                bool inside = (st.flags & (unsigned)kWfInsideBit) != 0;
                shade_hit<false>(rP[k], rD[k], st, seed, inside, p, p.tris, p.meta, __float_as_int(h.y), h.x);
                if (bounce + 1 >= p.iterations) {
                    wf_finalize(w, li[k], st.C());
                    p.rnds[li[k]] = seed;
                } else {
                    flags[k] = (st.flags & ~(unsigned)kWfInsideBit) | (inside ? (unsigned)kWfInsideBit : 0u);
                    cost[k] = ray_cost_class(w, rP[k], rD[k]);
                    fLs[k] = st.fL;
                    seeds[k] = seed;
                }
            }
        }
    }
    unsigned npos[kWfShadePerThread];
    block_reserve_multi<2, kWfShadeBlock, kWfShadePerThread>(cost, npos, w.counters + wf_row(bounce + 1) * kWfCounterStride, s_scratch);
#pragma unroll
    for (int k = 0; k < kWfShadePerThread; ++k)
        if (cost[k] >= 0) {
            w.rsA[(bounce + 1) & 1][cost[k]][npos[k]] = make_float4(rP[k].x, rP[k].y, rP[k].z, rD[k].x);
            w.rsB[(bounce + 1) & 1][cost[k]][npos[k]] = make_float4(rD[k].y, rD[k].z, __int_as_float(li[k]), __int_as_float((int)flags[k]));
            w.rsC[(bounce + 1) & 1][cost[k]][npos[k]] = make_float4(fLs[k].x, fLs[k].y, fLs[k].z, __int_as_float(seeds[k]));
        }
}

hipError_t launch_wf_generate(const WfParams& w, hipStream_t stream) {
    constexpr int per_block = kWfGenBlock * kWfGenPerThread;
    const int blocks = std::max((w.npix + per_block - 1) / per_block, ((w.rp.iterations + 3) * kWfCounterStride + kWfGenBlock - 1) / kWfGenBlock);
    hipLaunchKernelGGL(wf_generate, dim3(blocks), dim3(kWfGenBlock), 0, stream, w);
    return hipGetLastError();
}

template <int MODE, int BLOCK, int WPS, int RPW>
static size_t wf_intersect_lds(const WfParams& w) {
    constexpr int RPB = (BLOCK / 64) * RPW;
    return traversal_lds_bytes(w.rp, BLOCK) + RPB + (RPB / 64) * 3 * 4 + 32;
}

template <int MODE, int BLOCK, int WPS, int RPW>
static hipError_t launch_wf_intersect_t(const WfParams& w, int bounce, int resident_blocks, hipStream_t stream) {
    constexpr int RPB = (BLOCK / 64) * RPW;
    const size_t lds = wf_intersect_lds<MODE, BLOCK, WPS, RPW>(w);
    auto kern = wf_intersect<MODE, BLOCK, WPS, RPW>;
    static LdsMark mark;
    const hipError_t e = ensure_dynamic_lds((const void*)kern, mark, lds);
    if (e != hipSuccess) return e;
    const int blocks = std::min((w.npix + RPB - 1) / RPB, resident_blocks);
    if (w.rp.stack_ovf && 2ll * blocks * BLOCK > (long long)w.rp.stack_ovf_lanes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3(blocks, 2), dim3(BLOCK), lds, stream, w, bounce);
    return hipGetLastError();
}

hipError_t launch_wf_intersect(const WfParams& w, int bounce, int cu_count, hipStream_t stream) {
    // grid: what is resident at once (per cost class row; the rows of a launch share the chip).  Register budgets as for
    // k_render: whole tree in LDS -> two 768-thread workgroups per CU at 80 VGPRs (six waves per SIMD) where their LDS
    // fits, else two of 512; nodes from global memory -> six / seven 256-thread workgroups per CU
    switch (w.rp.node_mode) {
    case kNodesLds:
        if (2 * (wf_intersect_lds<kNodesLds, kLdsBlockWide, kLdsWpsWide, kWfRaysPerWaveLds>(w) + 512) <= 160 * 1024)
            return launch_wf_intersect_t<kNodesLds, kLdsBlockWide, kLdsWpsWide, kWfRaysPerWaveLds>(w, bounce, cu_count * 2, stream);
        return launch_wf_intersect_t<kNodesLds, 512, 4, kWfRaysPerWave>(w, bounce, cu_count * 2, stream);
    case kNodesGlobal: return launch_wf_intersect_t<kNodesGlobal, 256, 6, kWfRaysPerWave>(w, bounce, cu_count * 6, stream);
    case kNodesWide: return launch_wf_intersect_t<kNodesWide, 256, 6, kWfRaysPerWave>(w, bounce, cu_count * 6, stream);
    case kNodesTreelet: return launch_wf_intersect_t<kNodesTreelet, 1024, 4, kWfRaysPerWave>(w, bounce, cu_count, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_wf_shade(const WfParams& w, int bounce, hipStream_t stream) {
    hipLaunchKernelGGL(wf_shade, dim3((w.npix + kWfShadeBlock * kWfShadePerThread - 1) / (kWfShadeBlock * kWfShadePerThread), 3), dim3(kWfShadeBlock), 0, stream, w, bounce);
    return hipGetLastError();
}

}  // namespace ptamd
