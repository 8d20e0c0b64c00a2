// pt_device.hpp -- gfx950 device code shared by the kernel translation units of libptamd.so
// (pt_kernels.hip: megakernel; pt_wavefront.hip: stream-compacted variant; pt_debug.hip: test entry
// points).  Everything here is __device__ __forceinline__.
//
// Arithmetic: compiled with -ffp-contract=off; the only fused operations are the explicit fma calls,
// placed as DESIGN.md section 3 prescribes, so that results can be compared bit for bit with the CPU
// oracle (-fno-slp-vectorize: packed f32 ops cost what two scalar ones do, plus the shuffles).  '/'
// and sqrt are IEEE (hipcc default for HIP), sin/cos/pow are the double-precision polynomial routines
// below.  Box tests are NOT part of that contract: they are conservative (padded boxes, widened
// slabs) and only ever cull what the exact triangle test would reject.
#pragma once

#include "pt_internal.hpp"

// A/B switch of the slab test for nodes read from global memory (1: one fma per plane, planes picked
// by the sign of the direction; 0: round 1's (plane - P) * inv with min/max)
#ifndef PT_GLOBAL_SLAB_FMA
#define PT_GLOBAL_SLAB_FMA 1
#endif
// A/B switch of the 4-wide node visit (1: round 3's form -- decode, then slab test; 0: round 4's, see wide_step)
#ifndef PT_WIDE_STEP_V1
#define PT_WIDE_STEP_V1 0
#endif
// A/B switches: Trav::round ends a phase early when few lanes are still in it (see round()), per node path
#ifndef PT_PHASE_SWITCH_GLOBAL
#define PT_PHASE_SWITCH_GLOBAL 1
#endif
#ifndef PT_PHASE_SWITCH_LDS
#define PT_PHASE_SWITCH_LDS 1
#endif
#ifndef PT_WIDE_PREDICATED_STORES
#define PT_WIDE_PREDICATED_STORES 1
#endif

namespace ptamd {

// ---------------------------------------------------------------------------- small math
struct f3 {
    float x, y, z;
};
#define PT_DEV __device__ __forceinline__

PT_DEV f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 ldf3(const pt_float3& p) { return mk(p.s[0], p.s[1], p.s[2]); }
PT_DEV f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
PT_DEV f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
PT_DEV float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_DEV double fmad_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// u*s + w, one fma per component
PT_DEV f3 madd(f3 u, float s, f3 w) { return mk(fmaf_(u.x, s, w.x), fmaf_(u.y, s, w.y), fmaf_(u.z, s, w.z)); }
PT_DEV float dot3(f3 a, f3 b) { return fmaf_(a.z, b.z, fmaf_(a.y, b.y, a.x * b.x)); }
PT_DEV f3 cross3(f3 a, f3 b) {
    return mk(fmaf_(a.y, b.z, -(a.z * b.y)), fmaf_(a.z, b.x, -(a.x * b.z)), fmaf_(a.x, b.y, -(a.y * b.x)));
}
PT_DEV f3 normalize3(f3 a) {
    const float s = 1.0f / __builtin_sqrtf(dot3(a, a));
    return a * s;
}
PT_DEV float max0(float c) { return c > 0.0f ? c : 0.0f; }

// ---- spec math: double polynomials, rounded once to float
// The polynomial coefficients are 64-bit literals, which no gfx950 VALU instruction can encode: left alone, the compiler
// keeps each in a VGPR pair for the whole kernel (hoisted out of every loop) and, in the instances with 72-96 VGPRs,
// spills them -- ten 8-byte scratch reloads per lane in front of every cosine-sampled direction, each wave with its own
// copy of the same constants (~180 KB per CU fighting the nodes for L1 / L2).  KC<true>() pins a coefficient to a scalar
// register pair at its point of use instead: two s_mov_b32 with literal operands, then v_fma_f64 reads the pair directly.
// SK = false leaves the constants to the compiler: the 128-VGPR instance (whole tree in LDS) has room for them and is
// 1 % faster that way (profiles/r03/d_*).
template <bool SK, unsigned long long BITS>
PT_DEV double kc_bits() {
    if (SK) {
        // (the literal is written by the asm itself: with the constant as an INPUT of an empty asm the compiler hoisted its
        // materialisation out of the sample loop, ran out of scalar registers and kept it in VGPR lanes -- two v_readlane and two
        // v_writelane per use, ~50 VALU instructions per cosine-sampled direction, profiles/r03/y_*)
        unsigned lo, hi;
        asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"((unsigned)(BITS & 0xffffffffull)), "i"((unsigned)(BITS >> 32)));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
    }
    return __builtin_bit_cast(double, BITS);
}
#define KC_(SK, x) kc_bits<SK, __builtin_bit_cast(unsigned long long, (double)(x))>()
// a * b + K.  (Left to the compiler, a Horner step becomes v_fmac_f64 with the constant as the accumulator -- which has to be a VGPR
// pair: two v_mov_b32 from the scalar pair per step, 46 per cosine-sampled direction.  v_fma_f64 takes the pair as it is.)
template <bool SK, unsigned long long BITS>
PT_DEV double fmad_k(double a, double b) {
    if (SK) {
        const double k = kc_bits<SK, BITS>();
        double r;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
        return r;
    }
    return fmad_(a, b, __builtin_bit_cast(double, BITS));
}
#define FMADK(SK, a, b, x) fmad_k<SK, __builtin_bit_cast(unsigned long long, (double)(x))>(a, b)
template <bool SK>
PT_DEV void spec_sincos(float theta, float* s, float* c) {
    const double t = (double)theta;
    const int q = (int)fmad_(t, KC_(SK, 0.63661977236758138), 0.5);
    const double qd = (double)q;
    double r = fmad_(qd, KC_(SK, -1.5707963267948966), t);
    r = fmad_(qd, KC_(SK, -6.123233995736766e-17), r);
    const double z = r * r;
    double ps = KC_(SK, 1.6059043836821613e-10);
    ps = FMADK(SK, ps, z, -2.505210838544172e-08);
    ps = FMADK(SK, ps, z, 2.7557319223985893e-06);
    ps = FMADK(SK, ps, z, -0.0001984126984126984);
    ps = FMADK(SK, ps, z, 0.008333333333333333);
    ps = FMADK(SK, ps, z, -0.16666666666666666);
    const double sr = fmad_(r * z, ps, r);
    double pc = KC_(SK, -1.1470745597729725e-11);
    pc = FMADK(SK, pc, z, 2.08767569878681e-09);
    pc = FMADK(SK, pc, z, -2.755731922398589e-07);
    pc = FMADK(SK, pc, z, 2.48015873015873e-05);
    pc = FMADK(SK, pc, z, -0.001388888888888889);
    pc = FMADK(SK, pc, z, 0.041666666666666664);
    pc = fmad_(pc, z, -0.5);
    const double cr = fmad_(z, pc, 1.0);
    const int k = q & 3;
    const double sv = (k == 0) ? sr : (k == 1) ? cr : (k == 2) ? -sr : -cr;
    const double cv = (k == 0) ? cr : (k == 1) ? -sr : (k == 2) ? -cr : sr;
    *s = (float)sv;
    *c = (float)cv;
}

PT_DEV float spec_pow5(float x) {
    const float x2 = x * x;
    const float x4 = x2 * x2;
    return x4 * x;
}

template <bool SK>
PT_DEV float spec_pow(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return __builtin_nanf("");
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return y > 0.0f ? 0.0f : __builtin_inff();
    if (__builtin_isinf(x)) return y > 0.0f ? __builtin_inff() : 0.0f;
    const double xd = (double)x;
    unsigned long long bits = (unsigned long long)__double_as_longlong(xd);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = __longlong_as_double((long long)bits);
    if (m > KC_(SK, 1.4142135623730951)) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double sdiv = f / (2.0 + f);
    const double z = sdiv * sdiv;
    double p = KC_(SK, 0.10526315789473684);
    p = FMADK(SK, p, z, 0.11764705882352941);
    p = FMADK(SK, p, z, 0.13333333333333333);
    p = FMADK(SK, p, z, 0.15384615384615385);
    p = FMADK(SK, p, z, 0.18181818181818182);
    p = FMADK(SK, p, z, 0.22222222222222221);
    p = FMADK(SK, p, z, 0.2857142857142857);
    p = FMADK(SK, p, z, 0.4);
    p = FMADK(SK, p, z, 0.66666666666666663);
    p = fmad_(p, z, 2.0);
    const double lnm = sdiv * p;
    const double lg2 = fmad_(lnm, KC_(SK, 1.4426950408889634), (double)e);
    const double w = (double)y * lg2;
    if (!(w > -126.0)) return 0.0f;
    if (w >= 128.0) return __builtin_inff();
    const double nd = __builtin_floor(w + 0.5);
    const double g = (w - nd) * KC_(SK, 0.6931471805599453);
    double q = KC_(SK, 2.08767569878681e-09);
    q = FMADK(SK, q, g, 2.505210838544172e-08);
    q = FMADK(SK, q, g, 2.755731922398589e-07);
    q = FMADK(SK, q, g, 2.7557319223985893e-06);
    q = FMADK(SK, q, g, 2.48015873015873e-05);
    q = FMADK(SK, q, g, 0.0001984126984126984);
    q = FMADK(SK, q, g, 0.001388888888888889);
    q = FMADK(SK, q, g, 0.008333333333333333);
    q = FMADK(SK, q, g, 0.041666666666666664);
    q = FMADK(SK, q, g, 0.16666666666666666);
    q = fmad_(q, g, 0.5);
    q = fmad_(q, g, 1.0);
    q = fmad_(q, g, 1.0);
    const unsigned long long sb = (unsigned long long)((long long)nd + 1023) << 52;
    const double sc = __longlong_as_double((long long)sb);
    return (float)(q * sc);
}

// ---- LCG, prog.cl:72-77: n = (ulong)seed * 48271 % 2147483647.
// For seed >= 0 the product is < 2^47 and the modulus is the Mersenne number 2^31 - 1:
// n = hi * 2^31 + lo = hi + lo (mod M) with hi < 2^16, so one conditional subtraction finishes it
// (8 32-bit instructions instead of the ~30 of a 64-bit multiply and remainder).  A negative seed
// (only possible for a seed the caller uploaded; every output is in [0, M)) sign-extends to 64 bits
// as in the reference and takes the generic path.
PT_DEV float lcg_rand(int& seed) {
    unsigned n;
    if (seed >= 0) {
        const unsigned s = (unsigned)seed;
        const unsigned plo = s * 48271u, phi = __umulhi(s, 48271u);
        const unsigned t = (plo & 0x7fffffffu) + ((phi << 1) | (plo >> 31));
        n = min(t, t - 2147483647u);
    } else {
        unsigned long long w = (unsigned long long)(long long)seed;
        w = (w * 48271ull) % 2147483647ull;
        n = (unsigned)w;
    }
    seed = (int)n;
    return (float)n / 2147483648.0f;
}

// ---------------------------------------------------------------------------- pixel map
// One wave covers an 8x8 pixel tile of the LOCAL frame (width x local_rows).
struct PixelId {
    int li;   // local pixel index (buffer index), -1 = none
    int gid;  // global pixel id (what prog.cl calls id)
};
PT_DEV PixelId pixel_of_wave(const RenderParams& p, int wave) {
    const int lane = threadIdx.x & 63;
    const int tiles_x = (p.width + 7) >> 3;
    const int ty = wave / tiles_x, tx = wave - ty * tiles_x;
    const int x = tx * 8 + (lane & 7);
    const int lrow = ty * 8 + (lane >> 3);
    PixelId r;
    if (x >= p.width || lrow >= p.local_rows) {
        r.li = -1;
        r.gid = 0;
        return r;
    }
    const int grow = ((lrow / p.rows_per_block) * p.world + p.rank) * p.rows_per_block + (lrow % p.rows_per_block);
    r.li = lrow * p.width + x;
    r.gid = grow * p.width + x;
    return r;
}

PT_DEV PixelId pixel_of_thread(const RenderParams& p) {
    return pixel_of_wave(p, (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
}

// ---------------------------------------------------------------------------- camera, prog.cl:82-92
// (px, py) = (float)(id % X), (float)(id / X): fixed per pixel, so the render kernel computes them once
PT_DEV void camera_get_ray_xy(float px, float py, const pt_camera& cam, float rnd1, float rnd2, f3* P, f3* D) {
    const int X = (int)cam.XM;
    const int Y = (int)cam.YM;
    const float x = px + rnd1;
    const float y = py + rnd2;
    const f3 right = ldf3(cam.right) * ((2.0f * x) / (float)X - 1.0f);
    const f3 up = ldf3(cam.up) * ((2.0f * y) / (float)Y - 1.0f);
    const f3 pp = (ldf3(cam.lookat) + right) + up;
    const f3 eye = ldf3(cam.eye);
    *P = eye;
    *D = normalize3(pp - eye);
}
PT_DEV void camera_get_ray(int id, const pt_camera& cam, float rnd1, float rnd2, f3* P, f3* D) {
    const int X = (int)cam.XM;
    camera_get_ray_xy((float)(id % X), (float)(id / X), cam, rnd1, rnd2, P, D);
}

// ---------------------------------------------------------------------------- traversal
// dynamic LDS of every traversal kernel: [per-lane stacks][staged nodes][big-triangle list] (setup_traversal)
extern __shared__ __attribute__((aligned(16))) unsigned char pt_lds_raw[];

// Where the BVH nodes are read from (template parameter of Trav and of the kernels):
//   kNodesLds      every node is staged in LDS by the workgroup (stage_nodes), re-laid out for the
//                  ray (swizzled box quads, 16-bit child references, 16-bit stack entries).  Scenes
//                  of up to ~1,000 nodes (Cornell box: 941).
//   kNodesGlobal   nodes are read from global memory (L1/L2) as the host packed them.
//   kNodesTreelet  the top `treelet` nodes of a large tree -- the host re-indexes the tree so that
//                  the nodes with the largest boxes are nodes [0, treelet) -- are staged in LDS AS THEY
//                  ARE, the rest stays in global memory, and a visit reads its node through a generic
//                  (flat) pointer that each lane points at its LDS copy or at global memory: one
//                  instruction stream for both, no divergence between lanes above and below the cut.
//                  References stay 32-bit.  MESH-100k / MESH-1M (50 k / 505 k nodes).
struct SceneView {
    const float4* nodes;       // global memory, host layout (unused by kNodesLds)
    const float4* lds_nodes;   // staged copy (kNodesLds: all nodes; kNodesTreelet: nodes [0, treelet))
    const float4* tris;        // packets, global memory
    const TriMeta* meta;       // global memory
    unsigned treelet;          // kNodesTreelet: node indices below this are read from lds_nodes
    int n_flat;                // packets [0, n_flat): big triangles kept out of the tree, tested first (flat_pass)
    const float4* lds_flat;    // their packets, staged in LDS
    int n_fbox;                // their distinct padded boxes (the halves of a wall share one) ...
    const char* lds_fbox;      // ... 48 B each: per axis {lo, hi, hi, lo} (planes picked by address, like a staged node)
    const unsigned* lds_fmask; // ... and which listed triangles each one covers
    // kNodesWide: the lane's stack continues in global memory past its LDS entries (a 4-wide traversal can have three
    // children pending per level, far more than it usually has; LDS holds what keeps six waves per SIMD resident)
    // (all three wave-uniform, i.e. scalar registers: which entry an address is, and whose, is read off the address
    // itself -- the stacks start at LDS offset 0, entry k of lane l at (k * BLOCK + l) * 4 -- so that no per-lane
    // pointer rides through the traversal)
    // phase switching of Trav::round (wave-uniform): the node phase ends early when at most node_min_lanes lanes still descend
    // and another lane holds a leaf; the leaf phase when at most leaf_min_lanes lanes still hold leaves and another has a node
    int node_min_lanes, leaf_min_lanes;
    int lds_entries;           // entries per lane in LDS
    unsigned* ovf;             // entry (lds_entries + j) of lane l of this workgroup = ovf[j * ovf_stride + l]
    unsigned ovf_stride;
};

struct WorkCount {
    unsigned nodes, tris;     // per-lane visits / tests
    unsigned wnodes, wtris;   // wave-level executions of the two bodies (counted by the first active lane)
    unsigned wshade, wtrips, wrounds;   // wave-level executions of shade_hit, of the segment loop body, of Trav::round
    unsigned low[6];                    // ... of those that ran for at most 8 lanes: node body, triangle body, its exact part, shade_hit, the
                                        // big-triangle list's exact tests (all of them in low[5])
};
// counting instances: one more execution of body `which` (first active lane only), noting whether it ran for at most 8 lanes
PT_DEV void count_low(WorkCount* wc, int which) {
    const unsigned long long m = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1 && __popcll(m) <= 8) wc->low[which]++;
}
PT_DEV bool first_active_lane() {
    const unsigned long long m = __ballot(1);
    return (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1;
}

// prog.cl:94-112 on one packet; returns t (> 0) or -1.  `limit` is the current closest t: a
// triangle whose t is clearly larger can neither win nor tie, so it is dropped before the exact
// (IEEE-divide) evaluation.  Both early-outs are conservative: whatever the exact test would
// accept with t <= best_t passes them (t > 0 needs num and den of the same non-zero sign; the
// reciprocal estimate is within 2 ulp and the margin is 16 ulp).
// QUOT picks how the two early-outs are evaluated (same accepted set up to harmless extras):
//   true : from the estimated quotient alone, q = num * rcp(den): not negative and not above the
//          limit (q = 0, -0 and NaN pass and are sorted out by the exact evaluation) -- fewest VALU
//          instructions, +2.7 % on the VALU-bound LDS node path;
//   false: four sign compares first, the reciprocal only for same-sign pairs -- 3 % faster on the
//          L1/L2 node path, where the 8-clock v_rcp on every test costs more than it saves.
template <bool QUOT>
PT_DEV float tri_test(const float4 a, const float4 b, const float4 c, f3 P, f3 Vd, float limit, WorkCount* wc = nullptr) {
    const f3 r1 = mk(a.x, a.y, a.z), r2 = mk(a.w, b.x, b.y), r3 = mk(b.z, b.w, c.x), N = mk(c.y, c.z, c.w);
    const float num = dot3(r1 - P, N), den = dot3(Vd, N);
    float res = -1.0f;
    bool cand;
    if (QUOT) {
        const float q = num * __builtin_amdgcn_rcpf(den);
        cand = !(q < 0.0f) && !(q > limit);
    } else {
        const bool same_sign = (num > 0.0f && den > 0.0f) || (num < 0.0f && den < 0.0f);
        cand = same_sign && num * __builtin_amdgcn_rcpf(den) <= limit;
    }
    if (cand) {
        if (wc) { count_low(wc, 2); if (first_active_lane()) wc->low[5]++; }
        const float t = num / den;
        const f3 pt = madd(Vd, t, P);
        const float c1 = dot3(cross3(r2 - r1, pt - r1), N);
        const float c2 = dot3(cross3(r3 - r2, pt - r2), N);
        const float c3 = dot3(cross3(r1 - r3, pt - r3), N);
        const bool ok = !(t < 0.0f) && (c1 >= 0.0f) && (c2 >= 0.0f) && (c3 >= 0.0f) && (t > 0.0f);
        res = ok ? t : -1.0f;
    }
    return res;
}

// Per-lane traversal stack in LDS, laid out [entry][lane] (consecutive lanes -> consecutive
// banks).  Entry 0 holds a "finished" sentinel, so popping never has to test for an empty stack.
template <class T>
struct LaneStack {
    T* base;        // already offset by the lane
    int stride;     // entries are `stride` elements apart
};

enum : int { kVisitLds = 0, kVisitGlobal = 1, kVisitFlat = 2 };      // Trav::node_step

template <int MODE> struct StackOf { typedef unsigned type; };
template <> struct StackOf<kNodesLds> { typedef unsigned short type; };

// Closest hit over the whole scene; ties in t go to the lower encounter rank (the triangle the
// reference's traversal, prog.cl:113-184, meets first).
// While-while traversal, one "round" at a time: every lane first descends through interior nodes
// until it holds a leaf (or has finished), then all lanes holding leaves intersect them.  The
// per-lane state survives between rounds so that a flat loop can hand a finished lane its next ray
// while the others keep going (wf_intersect).
//
// Node references (`cur`, child slots, stack entries):
//   kNodesGlobal / kNodesTreelet   as the host packs them: >= 0 interior index, < 0 leaf
//                                  ~(first << 3 | count - 1), kDone finished; 32-bit stack entries
//   kNodesLds                      re-encoded to 16 bits by stage_nodes(): < 0x7fff interior index,
//                                  0x7fff finished, 0x8000 | (first << 3 | count - 1) leaf -- nothing
//                                  to encode or decode on a push or pop; 16-bit stack entries
template <int MODE>
struct Trav {
    typedef typename StackOf<MODE>::type StackT;
    static constexpr bool kRef16 = MODE == kNodesLds;
    static constexpr bool kDefer = MODE != kNodesLds;     // see node_step
    static constexpr bool kPhaseSwitch = MODE == kNodesLds ? (PT_PHASE_SWITCH_LDS != 0) : (PT_PHASE_SWITCH_GLOBAL != 0);   // see round()
    static constexpr int kDone = kRef16 ? 0x7fff : 0x7fffffff;
    f3 P, D, inv;
    float best_t;
    int best;      // packed triangle index of the closest hit so far, -1 none
    char* tos;     // top of this lane's stack (entry 0 = sentinel kDone), as a byte address
    int stride;    // bytes between entries
    int cur;
    int pend;      // round(): a leaf met during the node phase and not yet intersected (0: none)
    int onx, ony, onz;   // kNodesLds only: byte offset of the entry-plane pair of each axis inside a staged node; 8 / 24 / 40 = direction
                         // negative.  The other modes read the sign bits of `inv` where they need them (ox / oy / oz, negx / negy /
                         // negz): three registers fewer across a traversal, at the same instruction count
    f3 cn, cf;           // -(P * inv) widened down / up (entry / exit distance = fma(plane, inv, c))

    PT_DEV static bool is_node(int c) { return kRef16 ? c < 0x7fff : (unsigned)c < 0x7fffffffu; }
    PT_DEV static bool is_leaf(int c) { return kRef16 ? c > 0x7fff : c < 0; }
    PT_DEV static int leaf_bits(int c) { return kRef16 ? (c & 0x7fff) : ~c; }

    PT_DEV void begin(f3 P_, f3 D_, const LaneStack<StackT> stk) {
        setup(P_, D_);
        restart(stk);
    }
    // the traversal state proper: {best_t, best, tos, cur} (+ the stack itself); everything setup() computes can
    // be recomputed from the ray, which is what lets a suspended traversal resume from four saved values
    PT_DEV void restart(const LaneStack<StackT> stk) {
        best_t = __builtin_inff();
        best = -1;
        tos = reinterpret_cast<char*>(stk.base);
        stride = stk.stride * (int)sizeof(StackT);
        *reinterpret_cast<StackT*>(tos) = (StackT)kDone;
        cur = 0;        // the root is always an interior node
        pend = 0;
    }
    PT_DEV void setup(f3 P_, f3 D_) {
        P = P_;
        D = D_;
        inv = mk(__builtin_amdgcn_rcpf(D_.x), __builtin_amdgcn_rcpf(D_.y), __builtin_amdgcn_rcpf(D_.z));
        if (kRef16) {
            onx = __float_as_int(inv.x) < 0 ? 8 : 0;
            ony = __float_as_int(inv.y) < 0 ? 24 : 16;
            onz = __float_as_int(inv.z) < 0 ? 40 : 32;
        }
        // Distance to a plane as ONE fma: plane * inv - P * inv.  The product P * inv is rounded
        // (half an ulp of |P * inv|, which can dwarf the distance itself), so the entry constant
        // is lowered and the exit constant raised by 4 such half-ulps: entry distances come out
        // too small, exit distances too large, never the other way round.  inf - inf = NaN
        // (direction component 0 or underflowing) is ignored by max3/min3: that slab counts as
        // entered, which only costs work.  This needs the entry / exit plane of an axis to be picked
        // by the SIGN of the direction (by address in LDS, by select for nodes from global memory):
        // with min/max picking them a NaN would be replaced by the other plane's distance and cull
        // real hits (tests/test_gpu_parity.py::test_closest_hit_adversarial_rays).
        const float px = P.x * inv.x, py = P.y * inv.y, pz = P.z * inv.z;
        const float k22 = 2.3841858e-07f;    // 2^-22
        const float ex = __builtin_fabsf(px) * k22, ey = __builtin_fabsf(py) * k22, ez = __builtin_fabsf(pz) * k22;
        cn = mk(-(px + ex), -(py + ey), -(pz + ez));
        cf = mk(-(px - ex), -(py - ey), -(pz - ez));
    }
    PT_DEV bool negx() const { return __float_as_int(inv.x) < 0; }
    PT_DEV bool negy() const { return __float_as_int(inv.y) < 0; }
    PT_DEV bool negz() const { return __float_as_int(inv.z) < 0; }
    PT_DEV int ox() const { return kRef16 ? onx : (negx() ? 8 : 0); }
    PT_DEV int oy() const { return kRef16 ? ony : (negy() ? 24 : 16); }
    PT_DEV int oz() const { return kRef16 ? onz : (negz() ? 40 : 32); }
    PT_DEV void idle() { cur = kDone; }
    PT_DEV bool done() const { return cur == kDone; }

    // One interior-node visit: slab-test both children, descend into the nearer hit one, push the
    // other.  The visit is ONE basic block with one stack round trip: the top of the stack is fetched
    // together with the node (it is the next node if neither child is hit), and the far child is
    // stored above the top unconditionally; only the stack pointer moves conditionally.
    //
    // KIND = kVisitLds: the node is read from the staged copy, whose box quads are swizzled to
    // {L.lo, R.lo, L.hi, R.hi}: the entry / exit planes of both children are picked by ADDRESS from
    // the sign of the ray direction -- 7 8-byte LDS reads, no select (v_min/v_max/v_cndmask are
    // 4-cycle ops on gfx950, tools/micro/exec_ops.hip; the adds that form the addresses are 2-cycle
    // ops).  kVisitGlobal / kVisitFlat: the node comes as four 16-byte loads {L.lo, L.hi, R.lo, R.hi}
    // (narrow per-lane global loads cost per instruction and lane, not per byte: 7 of them ran at
    // 0.74-0.80x, profiles/r01/r_*) and the planes are picked by 12 selects; kVisitFlat loads through a
    // generic pointer that is the lane's LDS copy for nodes of the treelet and global memory below it.
    //
    // DEFER (nodes from global memory): when the nearer child is a leaf and no leaf is pending, the
    // leaf is remembered and the descent goes on with the other child or the stack; round()
    // intersects it after the node phase.  A lane then goes through about half as many node-phase /
    // leaf-phase alternations, each of which the whole wave waits out -- at the price of 3 % more
    // node visits (the pending leaf cannot prune yet).  Measured: L1/L2 node path +5..8 %, LDS node
    // path +-0 (the 5 extra VALU instructions per visit cost what the saved alternations bring).
    template <bool COUNT, bool DEFER, int KIND>
    PT_DEV void node_step(const SceneView& sv, WorkCount* wc) {
        const float kWiden = 1.0000005f;   // > 4 ulp: covers rcp + fma (begin() covers P * inv)
        const int top = (int)*reinterpret_cast<const StackT*>(tos);
        float ln, lf, rn, rf;
        int li, ri;
        if (COUNT) { wc->nodes++; if (first_active_lane()) wc->wnodes++; count_low(wc, 0); }
        if (KIND == kVisitLds) {
            const char* nb = reinterpret_cast<const char*>(sv.lds_nodes) + (unsigned)cur * (unsigned)kLdsNodeBytes;
            const float2 ex = *reinterpret_cast<const float2*>(nb + onx), xx = *reinterpret_cast<const float2*>(nb + (onx ^ 8));
            const float2 ey = *reinterpret_cast<const float2*>(nb + ony), xy = *reinterpret_cast<const float2*>(nb + (ony ^ 8));
            const float2 ez = *reinterpret_cast<const float2*>(nb + onz), xz = *reinterpret_cast<const float2*>(nb + (onz ^ 8));
            const unsigned chw = *reinterpret_cast<const unsigned*>(nb + 48);       // left | right << 16
            ln = fmaxf(fmaxf(fmaf_(ex.x, inv.x, cn.x), fmaf_(ey.x, inv.y, cn.y)), fmaf_(ez.x, inv.z, cn.z));
            rn = fmaxf(fmaxf(fmaf_(ex.y, inv.x, cn.x), fmaf_(ey.y, inv.y, cn.y)), fmaf_(ez.y, inv.z, cn.z));
            lf = fminf(fminf(fmaf_(xx.x, inv.x, cf.x), fmaf_(xy.x, inv.y, cf.y)), fmaf_(xz.x, inv.z, cf.z)) * kWiden;
            rf = fminf(fminf(fmaf_(xx.y, inv.x, cf.x), fmaf_(xy.y, inv.y, cf.y)), fmaf_(xz.y, inv.z, cf.z)) * kWiden;
            li = (int)(chw & 0xffffu);
            ri = (int)(chw >> 16);
        } else {
            // (unsigned 32-bit byte offsets: scalar base + vector offset addressing, no 64-bit address
            // math; pt_add_triangles caps the scene so that they cannot wrap)
            const unsigned off = (unsigned)cur << 6;
            float4 qx, qy, qz, qr;
            if (KIND == kVisitFlat) {
                const char* nb = ((unsigned)cur < sv.treelet ? reinterpret_cast<const char*>(sv.lds_nodes) : reinterpret_cast<const char*>(sv.nodes)) + off;
                qx = *reinterpret_cast<const float4*>(nb);
                qy = *reinterpret_cast<const float4*>(nb + 16);
                qz = *reinterpret_cast<const float4*>(nb + 32);
                qr = *reinterpret_cast<const float4*>(nb + 48);
            } else {
                const char* nb = reinterpret_cast<const char*>(sv.nodes);
                qx = *reinterpret_cast<const float4*>(nb + off);
                qy = *reinterpret_cast<const float4*>(nb + (off + 16u));
                qz = *reinterpret_cast<const float4*>(nb + (off + 32u));
                qr = *reinterpret_cast<const float4*>(nb + (off + 48u));
            }
#if PT_GLOBAL_SLAB_FMA
            const bool sx = negx(), sy = negy(), sz = negz();
            const float lnx = sx ? qx.y : qx.x, lfx = sx ? qx.x : qx.y, rnx = sx ? qx.w : qx.z, rfx = sx ? qx.z : qx.w;
            const float lny = sy ? qy.y : qy.x, lfy = sy ? qy.x : qy.y, rny = sy ? qy.w : qy.z, rfy = sy ? qy.z : qy.w;
            const float lnz = sz ? qz.y : qz.x, lfz = sz ? qz.x : qz.y, rnz = sz ? qz.w : qz.z, rfz = sz ? qz.z : qz.w;
            ln = fmaxf(fmaxf(fmaf_(lnx, inv.x, cn.x), fmaf_(lny, inv.y, cn.y)), fmaf_(lnz, inv.z, cn.z));
            rn = fmaxf(fmaxf(fmaf_(rnx, inv.x, cn.x), fmaf_(rny, inv.y, cn.y)), fmaf_(rnz, inv.z, cn.z));
            lf = fminf(fminf(fmaf_(lfx, inv.x, cf.x), fmaf_(lfy, inv.y, cf.y)), fmaf_(lfz, inv.z, cf.z)) * kWiden;
            rf = fminf(fminf(fmaf_(rfx, inv.x, cf.x), fmaf_(rfy, inv.y, cf.y)), fmaf_(rfz, inv.z, cf.z)) * kWiden;
#else
            // (plane - P) * inv with min/max picking the planes: ~2 ulp per distance, covered by the
            // 4-ulp widening (a NaN here is replaced by the other plane's finite distance, which is safe
            // only in this subtract-then-multiply form)
            const float lx0 = (qx.x - P.x) * inv.x, lx1 = (qx.y - P.x) * inv.x;
            const float rx0 = (qx.z - P.x) * inv.x, rx1 = (qx.w - P.x) * inv.x;
            const float ly0 = (qy.x - P.y) * inv.y, ly1 = (qy.y - P.y) * inv.y;
            const float ry0 = (qy.z - P.y) * inv.y, ry1 = (qy.w - P.y) * inv.y;
            const float lz0 = (qz.x - P.z) * inv.z, lz1 = (qz.y - P.z) * inv.z;
            const float rz0 = (qz.z - P.z) * inv.z, rz1 = (qz.w - P.z) * inv.z;
            ln = fmaxf(fmaxf(fminf(lx0, lx1), fminf(ly0, ly1)), fminf(lz0, lz1));
            lf = fminf(fminf(fmaxf(lx0, lx1), fmaxf(ly0, ly1)), fmaxf(lz0, lz1)) * kWiden;
            rn = fmaxf(fmaxf(fminf(rx0, rx1), fminf(ry0, ry1)), fminf(rz0, rz1));
            rf = fminf(fminf(fmaxf(rx0, rx1), fmaxf(ry0, ry1)), fmaxf(rz0, rz1)) * kWiden;
#endif
            li = __float_as_int(qr.x);
            ri = __float_as_int(qr.y);
        }
        const float lim = best_t * kWiden;
        const bool hl = (lf >= ln) && (lf >= 0.0f) && (ln <= lim);
        const bool hr = (rf >= rn) && (rf >= 0.0f) && (rn <= lim);
        const bool lfirst = ln <= rn;      // (its own statement: inside the expression below it comes back as a branch)
        const bool take_left = hl && (!hr || lfirst);
        const bool both = hl && hr, none = !(hl || hr);
        const int other = take_left ? ri : li;
        *reinterpret_cast<StackT*>(tos + stride) = (StackT)other;
        const int next = take_left ? li : ri;
        if (DEFER) {
            const bool cap = !none && is_leaf(next) && pend == 0;
            pend = cap ? next : pend;
            const bool usetop = none || (cap && !both);
            cur = usetop ? top : (cap ? other : next);
            tos += (both && !cap) ? stride : (usetop ? -stride : 0);
        } else {
            cur = none ? top : next;
            tos += both ? stride : (none ? -stride : 0);
        }
    }

    // One visit of a 4-wide node (Node4q, pt_internal.hpp; built by pt_wide.cpp): ONE 64-byte fetch decides what two
    // BVH2 levels decide.  The child planes are bytes on the node's own grid; a plane decodes as
    // fma((float)q, step, origin) -- the very expression the host checked to lie outside the child's box -- and then
    // goes through the same slab test as a BVH2 plane (entry / exit picked by the direction sign, here by selecting
    // the WORD that holds the four children's bytes: 6 selects for 24 planes).  The children that are hit are sorted
    // by entry distance (5 compare-exchanges); the nearest is visited next, the others are stored above the top of
    // the stack farthest first.  All three stores are unconditional (a child that was missed lands above the new
    // top, where it is never read), so the stack has room for top + 3 at every visit (wide_stack_entries()).
    // stack entry at address `at`, which may lie past the lane's LDS entries (kNodesWide only)
    PT_DEV int entry_of(const char* at) const {      // which stack entry an address is (stride = BLOCK x 4 B, a power of two)
        return (int)((unsigned)(at - reinterpret_cast<const char*>(pt_lds_raw)) >> (31 - __builtin_clz(stride)));
    }
    PT_DEV unsigned lane_of(const char* at) const {  // whose: the thread index inside the workgroup
        return ((unsigned)(at - reinterpret_cast<const char*>(pt_lds_raw)) & (unsigned)(stride - 1)) >> 2;
    }
    PT_DEV int stack_get(const SceneView& sv, const char* at) const {
        const int j = MODE == kNodesWide ? entry_of(at) - sv.lds_entries : -1;
        if (j < 0) return (int)*reinterpret_cast<const StackT*>(at);
        return (int)sv.ovf[(size_t)j * sv.ovf_stride + lane_of(at)];
    }
    PT_DEV void stack_put(const SceneView& sv, char* at, int v) const {
        const int j = MODE == kNodesWide ? entry_of(at) - sv.lds_entries : -1;
        if (j < 0) *reinterpret_cast<StackT*>(at) = (StackT)v;
        else sv.ovf[(size_t)j * sv.ovf_stride + lane_of(at)] = (unsigned)v;
    }

#if PT_WIDE_STEP_V1
    template <bool COUNT>
    PT_DEV void wide_step(const SceneView& sv, WorkCount* wc) {
        const float kWiden = 1.0000005f;
        // does any lane of the wave come within three entries of the end of its LDS part?  (rare: then every stack
        // access of this visit goes through the checked accessors)
        const bool tight = __ballot(entry_of(tos) + 3 >= sv.lds_entries) != 0;
        const int top = tight ? stack_get(sv, tos) : (int)*reinterpret_cast<const StackT*>(tos);
        if (COUNT) { wc->nodes++; if (first_active_lane()) wc->wnodes++; count_low(wc, 0); }
        const char* nb = reinterpret_cast<const char*>(sv.nodes);
        const unsigned off = (unsigned)cur << 6;
        const float4 h = *reinterpret_cast<const float4*>(nb + off);
        const uint4 qa = *reinterpret_cast<const uint4*>(nb + (off + 16u));
        const uint2 qb = *reinterpret_cast<const uint2*>(nb + (off + 32u));
        const int4 rf = *reinterpret_cast<const int4*>(nb + (off + 48u));
        const unsigned eb = (unsigned)__float_as_int(h.w);
        const float sx = __int_as_float((int)((eb & 0xffu) << 23)), sy = __int_as_float((int)(((eb >> 8) & 0xffu) << 23));
        const float sz = __int_as_float((int)(((eb >> 16) & 0xffu) << 23));
        const bool ngx = negx(), ngy = negy(), ngz = negz();
        const unsigned enx = ngx ? qa.y : qa.x, exx = ngx ? qa.x : qa.y;
        const unsigned eny = ngy ? qa.w : qa.z, exy = ngy ? qa.z : qa.w;
        const unsigned enz = ngz ? qb.y : qb.x, exz = ngz ? qb.x : qb.y;
        const float lim = best_t * kWiden;
        float key[4];
        int ref[4] = {rf.x, rf.y, rf.z, rf.w};
        int n = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float pnx = fmaf_((float)((enx >> (8 * k)) & 0xffu), sx, h.x), pfx = fmaf_((float)((exx >> (8 * k)) & 0xffu), sx, h.x);
            const float pny = fmaf_((float)((eny >> (8 * k)) & 0xffu), sy, h.y), pfy = fmaf_((float)((exy >> (8 * k)) & 0xffu), sy, h.y);
            const float pnz = fmaf_((float)((enz >> (8 * k)) & 0xffu), sz, h.z), pfz = fmaf_((float)((exz >> (8 * k)) & 0xffu), sz, h.z);
            const float tn = fmaxf(fmaxf(fmaf_(pnx, inv.x, cn.x), fmaf_(pny, inv.y, cn.y)), fmaf_(pnz, inv.z, cn.z));
            const float tf = fminf(fminf(fmaf_(pfx, inv.x, cf.x), fmaf_(pfy, inv.y, cf.y)), fmaf_(pfz, inv.z, cf.z)) * kWiden;
            // (no test for "no child": its box is inverted and its reference a harmless leaf -- pt_wide.cpp; a "hit" at
            // tn = +inf is a ray parallel to a slab it is outside of: sorting it among the misses drops it, rightly)
            const bool hit = (tf >= tn) && (tf >= 0.0f) && (tn <= lim);
            key[k] = hit ? tn : __builtin_inff();
            n += key[k] < __builtin_inff() ? 1 : 0;
        }
#define PT_CSWAP(a, b)                                                                        \
        {                                                                                     \
            const bool sw = key[a] > key[b];                                                  \
            const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];                 \
            const int ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];                   \
            key[a] = ka; key[b] = kb; ref[a] = ra; ref[b] = rb;                               \
        }
        PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#undef PT_CSWAP
        // the nearest child is a leaf and no leaf is pending: remember it (round() intersects it after the node phase)
        // and go on with the next child -- fewer node-phase / leaf-phase alternations, as in node_step
        const bool cap = n > 0 && is_leaf(ref[0]) && pend == 0;
        pend = cap ? ref[0] : pend;
        ref[0] = cap ? ref[1] : ref[0];
        ref[1] = cap ? ref[2] : ref[1];
        ref[2] = cap ? ref[3] : ref[2];
        n -= cap ? 1 : 0;
        // child j (1 <= j < n) goes to entry top + (n - j): popped nearest first; a missed child (j >= n) to top + j
        char* const a1 = tos + (1 < n ? n - 1 : 1) * stride;
        char* const a2 = tos + (2 < n ? n - 2 : 2) * stride;
        char* const a3 = tos + (3 < n ? n - 3 : 3) * stride;
        if (!tight) {
            *reinterpret_cast<StackT*>(a1) = (StackT)ref[1];
            *reinterpret_cast<StackT*>(a2) = (StackT)ref[2];
            *reinterpret_cast<StackT*>(a3) = (StackT)ref[3];
        } else {
            stack_put(sv, a1, ref[1]);
            stack_put(sv, a2, ref[2]);
            stack_put(sv, a3, ref[3]);
        }
        cur = n > 0 ? ref[0] : top;
        tos += (n > 0 ? n - 1 : -1) * stride;
    }

#else
    // Round 4: the visit re-counted in CLOCKS (tools/micro/exec_ops.hip: v_fma / v_mul / v_add 2.4-2.9 per wave
    // instruction, v_cvt / v_cmp / v_cndmask / v_min3 / v_max3 / v_lshlrev 4.2-4.4, a VOP2 v_cndmask on a VCC the SALU wrote 21):
    //  * decode and slab test are ONE fma per plane: t = q * (step * inv) + (origin * inv - P * inv).  step is a power of
    //    two, so step * inv is exact; the second term costs one rounding per axis and side, taken once per node (6 fma)
    //    instead of once per plane (24 fma saved).  The plane the host checked, fl(q * step + origin), no longer exists:
    //    the effective plane is the real number q * step + origin, at most half an ulp of the coordinate off it --
    //    1 / 170 of what padded_bounds() adds to every triangle's box (1e-5 of the coordinate), and the new rounding of
    //    (origin * inv + c) is one more half-ulp of P * inv out of the four that c is widened by (setup());
    //  * the three exponent bytes become step factors with one SDWA shift each (byte select in the instruction);
    //  * hit <=> max(tn, 0) <= min(tf, lim): one compare, whose VCC feeds the select of the key AND the count of hit
    //    children (three compares joined on the SALU ended in a 21-clock select);
    //  * the pushes are predicated stores at tos + (n - j): no address selects.
    PT_DEV static float exp_byte_step(unsigned eb, int which) {      // 2^(byte - 127): byte << 23
        unsigned r;
        if (which == 0) asm("v_lshlrev_b32_sdwa %0, 23, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(eb));
        else if (which == 1) asm("v_lshlrev_b32_sdwa %0, 23, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(eb));
        else asm("v_lshlrev_b32_sdwa %0, 23, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(eb));
        return __int_as_float((int)r);
    }
    template <bool COUNT>
    PT_DEV void wide_step(const SceneView& sv, WorkCount* wc) {
        const float kWiden = 1.0000005f;
        // does any lane of the wave come within three entries of the end of its LDS part?  (rare: then every stack
        // access of this visit goes through the checked accessors)
        const bool tight = __ballot(entry_of(tos) + 3 >= sv.lds_entries) != 0;
        const int top = tight ? stack_get(sv, tos) : (int)*reinterpret_cast<const StackT*>(tos);
        if (COUNT) { wc->nodes++; if (first_active_lane()) wc->wnodes++; count_low(wc, 0); }
        const char* nb = reinterpret_cast<const char*>(sv.nodes);
        const unsigned off = (unsigned)cur << 6;
        const float4 h = *reinterpret_cast<const float4*>(nb + off);
        const uint4 qa = *reinterpret_cast<const uint4*>(nb + (off + 16u));
        const uint2 qb = *reinterpret_cast<const uint2*>(nb + (off + 32u));
        const int4 rf = *reinterpret_cast<const int4*>(nb + (off + 48u));
        const unsigned eb = (unsigned)__float_as_int(h.w);
        const float ax = exp_byte_step(eb, 0) * inv.x, ay = exp_byte_step(eb, 1) * inv.y, az = exp_byte_step(eb, 2) * inv.z;
        const float bnx = fmaf_(h.x, inv.x, cn.x), bny = fmaf_(h.y, inv.y, cn.y), bnz = fmaf_(h.z, inv.z, cn.z);
        const float bfx = fmaf_(h.x, inv.x, cf.x), bfy = fmaf_(h.y, inv.y, cf.y), bfz = fmaf_(h.z, inv.z, cf.z);
        const bool ngx = negx(), ngy = negy(), ngz = negz();
        const unsigned enx = ngx ? qa.y : qa.x, exx = ngx ? qa.x : qa.y;
        const unsigned eny = ngy ? qa.w : qa.z, exy = ngy ? qa.z : qa.w;
        const unsigned enz = ngz ? qb.y : qb.x, exz = ngz ? qb.x : qb.y;
        const float lim = best_t * kWiden;
        float key[4];
        int ref[4] = {rf.x, rf.y, rf.z, rf.w};
        int n = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float tnx = fmaf_((float)((enx >> (8 * k)) & 0xffu), ax, bnx), tfx = fmaf_((float)((exx >> (8 * k)) & 0xffu), ax, bfx);
            const float tny = fmaf_((float)((eny >> (8 * k)) & 0xffu), ay, bny), tfy = fmaf_((float)((exy >> (8 * k)) & 0xffu), ay, bfy);
            const float tnz = fmaf_((float)((enz >> (8 * k)) & 0xffu), az, bnz), tfz = fmaf_((float)((exz >> (8 * k)) & 0xffu), az, bfz);
            // entry clamped to the ray's origin: then "exit >= entry" includes "exit >= 0", and lim >= 0 always
            const float tn = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), 0.0f);
            const float tf = fminf(fminf(fminf(tfx, tfy), tfz) * kWiden, lim);
            // (no test for "no child": its box is inverted and its reference a harmless leaf -- pt_wide.cpp.  A child
            // "hit" at tn = +inf -- a ray parallel to a slab it is outside of -- sorts among the misses; should it be
            // pushed, or a miss in its place, the visit that follows is wasted work, not a wrong answer.)
            const bool hit = tn <= tf;
            key[k] = hit ? tn : __builtin_inff();
            n += hit ? 1 : 0;
        }
#define PT_CSWAP(a, b)                                                                        \
        {                                                                                     \
            const bool sw = key[a] > key[b];                                                  \
            const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];                 \
            const int ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];                   \
            key[a] = ka; key[b] = kb; ref[a] = ra; ref[b] = rb;                               \
        }
        PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#undef PT_CSWAP
        // the nearest child is a leaf and no leaf is pending: remember it (round() intersects it after the node phase)
        // and go on with the next child -- fewer node-phase / leaf-phase alternations, as in node_step
        const bool cap = n > 0 && is_leaf(ref[0]) && pend == 0;
        pend = cap ? ref[0] : pend;
        ref[0] = cap ? ref[1] : ref[0];
        ref[1] = cap ? ref[2] : ref[1];
        ref[2] = cap ? ref[3] : ref[2];
        n -= cap ? 1 : 0;
#if PT_WIDE_PREDICATED_STORES
        // child j (1 <= j < n) goes to entry top + (n - j): popped nearest first; a missed child is not stored
        char* const a3 = tos + (n - 3) * stride;        // child 3's entry; children 2 and 1 one and two entries above it
        if (!tight) {
            if (1 < n) *reinterpret_cast<StackT*>(a3 + 2 * stride) = (StackT)ref[1];
            if (2 < n) *reinterpret_cast<StackT*>(a3 + stride) = (StackT)ref[2];
            if (3 < n) *reinterpret_cast<StackT*>(a3) = (StackT)ref[3];
        } else {
            if (1 < n) stack_put(sv, a3 + 2 * stride, ref[1]);
            if (2 < n) stack_put(sv, a3 + stride, ref[2]);
            if (3 < n) stack_put(sv, a3, ref[3]);
        }
#else
        // child j (1 <= j < n) goes to entry top + (n - j): popped nearest first; a missed child (j >= n) to top + j, above the
        // new top, where it is never read -- all three stores are unconditional (no branch in the visit)
        char* const a1 = tos + (1 < n ? n - 1 : 1) * stride;
        char* const a2 = tos + (2 < n ? n - 2 : 2) * stride;
        char* const a3 = tos + (3 < n ? n - 3 : 3) * stride;
        if (!tight) {
            *reinterpret_cast<StackT*>(a1) = (StackT)ref[1];
            *reinterpret_cast<StackT*>(a2) = (StackT)ref[2];
            *reinterpret_cast<StackT*>(a3) = (StackT)ref[3];
        } else {
            stack_put(sv, a1, ref[1]);
            stack_put(sv, a2, ref[2]);
            stack_put(sv, a3, ref[3]);
        }
#endif
        cur = n > 0 ? ref[0] : top;
        tos += (n > 0 ? n - 1 : -1) * stride;
    }

#endif
    // exact test of packed triangle ti against the ray, keeping the closest (ties: lower rank)
    template <bool COUNT>
    PT_DEV void tri_step(const SceneView& sv, int ti, WorkCount* wc) {
        const char* tb = reinterpret_cast<const char*>(sv.tris);
        const unsigned off = (unsigned)ti * 48u;
        const float4 a = *reinterpret_cast<const float4*>(tb + off), b = *reinterpret_cast<const float4*>(tb + (off + 16u)), c = *reinterpret_cast<const float4*>(tb + (off + 32u));
        tri_update<COUNT>(sv, a, b, c, ti, wc);
    }
    template <bool COUNT>
    PT_DEV void tri_update(const SceneView& sv, const float4 a, const float4 b, const float4 c, int ti, WorkCount* wc) {
        if (COUNT) { wc->tris++; if (first_active_lane()) wc->wtris++; count_low(wc, 1); }
        const float t = tri_test<MODE == kNodesLds>(a, b, c, P, D, best_t * 1.000002f, COUNT ? wc : nullptr);
        if (t > 0.0f) {
            bool better = t < best_t;
            if (t == best_t && best >= 0) better = sv.meta[ti].rank < sv.meta[best].rank;
            if (better) { best_t = t; best = ti; }
        }
    }

    // The big-triangle list (the host keeps walls, floors ... out of the tree: pt_builder.cpp build_and_pack), tested
    // at the start of every traversal; what it finds prunes the tree from its first node visit.  Two passes:
    //  1. every lane slab-tests the DISTINCT padded boxes of the listed triangles (the two halves of a wall share
    //     one: 6 boxes for the Cornell box's 12 triangles, grouped by the host) -- a wave-uniform loop at full lane
    //     utilisation, 14 VALU + 4 LDS reads per box, planes picked by address from the direction signs exactly as
    //     for a staged node -- and notes the triangles of the boxes it touches in a bit mask (the same
    //     conservative cull a leaf box performs in the tree);
    //  2. each lane runs the exact test only on ITS candidates (2-6 of the 12 in the Cornell box: a wall is two
    //     triangles with one box, and a ray may cross the planes of a few walls), packets read from the LDS copy
    //     at a per-lane address.
    // Same exact test and tie-break as in a leaf, so the closest hit is unchanged.  (Testing all 12 exactly on
    // every lane cost 984 VALU per pass, a third of the kernel's instructions; this form ~600.)
    template <bool COUNT>
    PT_DEV void flat_pass(const SceneView& sv, WorkCount* wc) {
        const float kWiden = 1.0000005f;
        unsigned mask = 0;
        const int fx = ox(), fy = oy(), fz = oz();
#pragma clang loop unroll(disable) vectorize(disable)      // (unrolled x8 it spills 40 registers around the loop)
        for (int i = 0; i < sv.n_fbox; ++i) {
            const char* bb = sv.lds_fbox + i * 48;
            const float2 x = *reinterpret_cast<const float2*>(bb + fx);      // (entry plane, exit plane) of the axis
            const float2 y = *reinterpret_cast<const float2*>(bb + fy);
            const float2 z = *reinterpret_cast<const float2*>(bb + fz);
            const float tn = fmaxf(fmaxf(fmaf_(x.x, inv.x, cn.x), fmaf_(y.x, inv.y, cn.y)), fmaf_(z.x, inv.z, cn.z));
            const float tf = fminf(fminf(fmaf_(x.y, inv.x, cf.x), fmaf_(y.y, inv.y, cf.y)), fmaf_(z.y, inv.z, cf.z)) * kWiden;
            mask |= ((tf >= tn) && (tf >= 0.0f)) ? sv.lds_fmask[i] : 0u;
        }
        // 1b. (VALU-bound LDS node path only: +1.9 % there, -1 % on the paths that wait for memory) of those, keep the
        //     triangles whose plane lies ahead: the first early-out of the exact test (same predicate, so nothing the
        //     exact test would accept is dropped), 11 VALU + 2 LDS reads per candidate.  A ray leaving a wall still
        //     touches that wall's box -- two candidates per lane that only lengthen pass 2, whose every iteration costs
        //     the wave a full exact test as long as ONE lane has a live candidate.
        unsigned keep = mask;
        if (MODE == kNodesLds) {
            keep = 0;
            const float limit = best_t * 1.000002f;
#pragma clang loop unroll(disable)
            while (mask != 0) {
                const int i = __ffs((int)mask) - 1;
                mask &= mask - 1;
                const float4* pk = sv.lds_flat + i * 3;
                const float4 a = pk[0], c = pk[2];
                const f3 N = mk(c.y, c.z, c.w);
                const float q = dot3(mk(a.x, a.y, a.z) - P, N) * __builtin_amdgcn_rcpf(dot3(D, N));
                keep |= (!(q < 0.0f) && !(q > limit)) ? (1u << i) : 0u;        // tri_test<true>'s early-out
            }
        }
        //  2. the exact test on what is left
#pragma clang loop unroll(disable)
        while (keep != 0) {
            const int i = __ffs((int)keep) - 1;
            keep &= keep - 1;
            const float4* pk = sv.lds_flat + i * 3;
            tri_update<COUNT>(sv, pk[0], pk[1], pk[2], i, wc);
        }
    }

    // One round: a node phase (every lane descends until it holds a leaf or has finished), then a leaf phase (the lanes holding
    // leaves intersect them and pop).  Round 4: a phase ENDS EARLY when few lanes are still in it and another lane is waiting for
    // the other phase -- the stragglers simply go on in the next round, next to the lanes that come back.  What a while-while
    // loop loses is its tails: half of all node-body executions ran for four lanes or fewer, and a VALU instruction with at most
    // 8 active lanes costs 3-4 times what it costs with 9 to 64 (tools/micro/exec_ops.hip).  MESH-100k 841 -> 980 Msamples/s,
    // MESH-1M 297 -> 353 at node_min_lanes = 8 (profiles/r04/).  The traversal state allows it as it is: round() can be entered
    // with a node, a leaf or a pending leaf in hand.
    template <bool COUNT>
    PT_DEV void step(const SceneView& sv, WorkCount* wc) {
        if (MODE == kNodesLds) node_step<COUNT, false, kVisitLds>(sv, wc);
        else if (MODE == kNodesGlobal) node_step<COUNT, true, kVisitGlobal>(sv, wc);
        else if (MODE == kNodesWide) wide_step<COUNT>(sv, wc);
        else node_step<COUNT, true, kVisitFlat>(sv, wc);
    }
    template <bool COUNT>
    PT_DEV void round(const SceneView& sv, WorkCount* wc) {
        // Every lane that enters a phase takes at least one step of it (so a round always makes progress, whatever the two
        // thresholds are); after that the phase ends for everybody as soon as at most `min_lanes` lanes are still in it AND some
        // lane has left it -- with a leaf to intersect, a node to visit, or a finished ray for the caller to replace.
        if (kPhaseSwitch) {
            const unsigned long long entered = __ballot(is_node(cur));
            if (is_node(cur)) {
                for (;;) {
                    step<COUNT>(sv, wc);
                    const bool live = is_node(cur);
                    if (!live) break;
                    const unsigned long long lm = __ballot(live);
                    if (lm != entered && __popcll(lm) <= sv.node_min_lanes) break;
                }
            }
        } else {
            while (is_node(cur)) step<COUNT>(sv, wc);
        }
        if (kDefer && pend != 0) {            // met first, so nearer: intersect it first
            const int v = leaf_bits(pend);
            const int first = v >> 3, count = (v & 7) + 1;
            for (int j = 0; j < count; ++j) tri_step<COUNT>(sv, first + j, wc);
            pend = 0;
        }
        if (kPhaseSwitch) {
            const unsigned long long entered = __ballot(is_leaf(cur));
            if (is_leaf(cur)) {
                for (;;) {
                    leaf_step<COUNT>(sv, wc);
                    const bool live = is_leaf(cur);
                    if (!live) break;
                    const unsigned long long lm = __ballot(live);
                    if (lm != entered && __popcll(lm) <= sv.leaf_min_lanes) break;
                }
            }
        } else {
            while (is_leaf(cur)) leaf_step<COUNT>(sv, wc);
        }
    }
    template <bool COUNT>
    PT_DEV void leaf_step(const SceneView& sv, WorkCount* wc) {
        const int popped = stack_get(sv, tos);     // in flight during the triangle tests
        const int v = leaf_bits(cur);
        const int first = v >> 3, count = (v & 7) + 1;
        for (int j = 0; j < count; ++j) tri_step<COUNT>(sv, first + j, wc);
        cur = popped;
        tos -= stride;
    }
};

template <int MODE, bool COUNT>
PT_DEV int closest_hit(const SceneView& sv, f3 P, f3 D, const LaneStack<typename StackOf<MODE>::type> stk, float* t_out, WorkCount* wc) {
    Trav<MODE> tr;
    tr.begin(P, D, stk);
    tr.template flat_pass<COUNT>(sv, wc);
    while (!tr.done()) tr.template round<COUNT>(sv, wc);
    *t_out = tr.best_t;
    return tr.best;
}

// ---------------------------------------------------------------------------- BSDF sampling
// prog.cl:186-218
template <bool SK>
PT_DEV f3 diffuse_direction(f3 N, float rnd1, float rnd2) {       // prog.cl:205-218 up to the normalisation (shade_hit does it)
    const float E = 0.001f;
    const bool yaxis = __builtin_fabsf(N.x) <= E && __builtin_fabsf(N.z) <= E;
    const float other = yaxis ? N.y : N.x;
    const float rl = 1.0f / __builtin_sqrtf(fmaf_(N.z, N.z, other * other));
    const f3 Z = yaxis ? mk(0.0f, -N.z * rl, N.y * rl) : mk(-N.z * rl, 0.0f, N.x * rl);
    const f3 X = cross3(N, Z);
    const float r = __builtin_sqrtf(rnd1);
    const float theta = (float)(6.283185307179586 * (double)rnd2);
    float sn, cs;
    spec_sincos<SK>(theta, &sn, &cs);
    const float x = r * cs, y = r * sn, z = __builtin_sqrtf(1.0f - rnd1);
    f3 d = X * x;
    d = madd(N, z, d);
    d = madd(Z, y, d);
    return d;
}

// prog.cl:219-222
PT_DEV f3 fresnel(f3 F0, f3 N, f3 D) {
    const float cosa = __builtin_fabsf(dot3(N, D));
    const float p5 = spec_pow5(1.0f - cosa);
    return mk(fmaf_(1.0f - F0.x, p5, F0.x), fmaf_(1.0f - F0.y, p5, F0.y), fmaf_(1.0f - F0.z, p5, F0.z));
}

// ---------------------------------------------------------------------------- path state + shading
// The path state of prog.cl:307-316: ray (P, D), LCG state and inside-glass flag are plain local variables of the
// caller; the four factors and the colour travel as one PathRegs.  (Round 3 also tried them in global memory behind the
// same accessors for the 72 / 80-VGPR kernel instances -- [field][lane of the grid], a segment moving only what its
// material touches: 1.5-3.6 % slower than letting the allocator spill, profiles/r03/c_*.)
struct PathRegs {
    f3 fL, fB, fS, fR, color;
    PT_DEV void reset() {
        fL = mk(1.f, 1.f, 1.f);
        fB = fL;
        fS = fL;
        fR = fL;
        color = mk(0.f, 0.f, 0.f);
    }
    PT_DEV f3 L() const { return fL; }
    PT_DEV f3 B() const { return fB; }
    PT_DEV f3 S() const { return fS; }
    PT_DEV f3 R() const { return fR; }
    PT_DEV f3 C() const { return color; }
    PT_DEV void setL(f3 v) { fL = v; }
    PT_DEV void setB(f3 v) { fB = v; }
    PT_DEV void setS(f3 v) { fS = v; }
    PT_DEV void setR(f3 v) { fR = v; }
    PT_DEV void setC(f3 v) { color = v; }
};

// one iteration body of prog.cl:317-366 for a ray that hit packed triangle `ti` at `t`
// (SK: double-precision constants pinned to scalar registers, see KC)
template <bool SK, class ST>
PT_DEV void shade_hit(f3& rP, f3& rD, ST& st, int& seed, bool& inside, const RenderParams& p, const float4* tris, const TriMeta* meta, int ti, float t) {
    const float4 c = tris[ti * 3 + 2];
    f3 N = mk(c.y, c.z, c.w);
    const f3 hp = madd(rD, t, rP);
    const pt_material* __restrict__ m = &p.mats[meta[ti].mati];
    const int type = m->type;
    if (p.iterations == 1) st.setC(ldf3(m->kd) + ldf3(m->emission));        // prog.cl:323-325
    if (dot3(rD, N) > 0.0f) N = -N;                                         // prog.cl:326-328
    // Every material that continues the path ends the same way: normalise the new direction, step off the surface
    // along +-N.  The two sampling branches below only produce the direction BEFORE normalisation and the side; the
    // tail is shared, so a wave that holds both kinds of hit runs one normalisation (IEEE sqrt + divide), not two.
    const bool lobe = type == 0 || type == 3, spec = type == 1 || type == 2;
    f3 dnew = rD;
    float side = 0.001f;
    float inten = 0.0f;
    if (lobe) {
        // diffuse (prog.cl:329-340) and emitter (prog.cl:358-366) both continue with a cosine-
        // sampled ray drawn from two LCG values; the emitter's cosine uses the OLD direction.
        inten = max0(dot3(-rD, N));
        const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
        dnew = diffuse_direction<SK>(N, rnd1, rnd2);
    } else if (spec) {
        // mirror (prog.cl:341-345) and dielectric (prog.cl:346-357, 228-245) share the Fresnel
        // term and the mirror direction; the dielectric may pick the refracted direction instead.
        const f3 oldD = rD;
        const f3 F0 = ldf3(m->F0);
        const f3 F = fresnel(F0, N, oldD);
        dnew = oldD - (N * dot3(N, oldD)) * 2.0f;
        if (type == 2) {
            float n = m->n;
            if (inside) n = 1.0f / n;
            const float rnd = lcg_rand(seed);
            const float cosa = dot3(-oldD, N);
            const float disc = 1.0f - (fmaf_(-cosa, cosa, 1.0f) / n) / n;
            const float prob = ((F.x + F.y) + F.z) / 3.0f;
            const bool refr = disc > 0.0f && rnd > prob;
            if (refr) {
                const f3 dn = mk(oldD.x / n, oldD.y / n, oldD.z / n);
                dnew = madd(N, cosa / n - __builtin_sqrtf(disc), dn);
                const float k = 1.0f / (1.0f - prob);
                st.setR((st.R() * mk(1.0f - F.x, 1.0f - F.y, 1.0f - F.z)) * k);
                inside = !inside;
                side = -0.001f;
            } else {
                const float k = 1.0f / prob;
                st.setR((st.R() * F) * k);
            }
        } else {
            st.setS(st.S() * F);
        }
    }
    if (lobe || spec) {
        rD = normalize3(dnew);
        rP = madd(N, side, hp);
    }
    if (type == 0) {
        const float idiff = max0(dot3(rD, N));
        st.setL(st.L() * (ldf3(m->kd) * idiff));
        // m->_pad = 1: ks is exactly 0 and shininess is finite >= 0, so ks * pow(...) is +-0 whatever the
        // (finite) power is -- skip the halfway vector (two normalisations) and the double-precision pow
        // (set by pt_upload_materials)
        float pw = 1.0f;
        if (!m->_pad) {
            const f3 view = normalize3(ldf3(p.cam.eye) - hp);
            const f3 halfway = normalize3(view + rD);
            const float ispec = max0(dot3(N, halfway));
            pw = spec_pow<SK>(ispec, m->shininess);
        }
        st.setB(st.B() * (ldf3(m->ks) * pw));
    } else if (type == 3) {
        const f3 e = ((ldf3(m->emission) * (st.L() + st.B())) * st.S()) * st.R();
        st.setC(madd(e, inten, st.C()));
    }
    // any other type: the ray is left unchanged and the loop hits the same surface again
}

PT_DEV f3 running_mean(f3 acc, f3 color, int s) {   // prog.cl:379
    const float cs = (float)s, cs1 = (float)(s + 1);
    return mk(fmaf_(acc.x, cs, color.x) / cs1, fmaf_(acc.y, cs, color.y) / cs1, fmaf_(acc.z, cs, color.z) / cs1);
}

// ---------------------------------------------------------------------------- LDS staging

// Nodes staged in LDS are re-laid out on the way in: the three box quads {L.lo, L.hi, R.lo, R.hi}
// become {L.lo, R.lo, L.hi, R.hi}, so that one 8-byte read at (quad + 0 | 8) returns the entry
// (or exit) planes of BOTH children for a ray whose direction sign on that axis is known
// (Trav::node_step, kVisitLds), and the child slots of the fourth quad are re-encoded to the 16-bit
// reference form.  The treelet of a large tree is staged verbatim (kVisitFlat reads both copies alike).
PT_DEV unsigned stage_ref(float slot) {
    const int r = __float_as_int(slot);
    return (unsigned)(r < 0 ? (0x8000 | ~r) : r) & 0xffffu;
}
template <int MODE>
PT_DEV void stage_nodes(const RenderParams& p, float4* lds_nodes) {
    if (MODE != kNodesLds) {                     // the treelet of a large tree: verbatim
        const int nn = p.treelet_nodes * 4;
        for (int i = threadIdx.x; i < nn; i += blockDim.x) lds_nodes[i] = p.nodes[i];
        return;
    }
    // whole tree: kLdsNodeBytes per node -- {L.lo, R.lo, L.hi, R.hi} per axis, then left | right << 16
    char* out = reinterpret_cast<char*>(lds_nodes);
    const int nn = p.n_nodes * 4;
    for (int i = threadIdx.x; i < nn; i += blockDim.x) {
        const float4 q = p.nodes[i];
        char* nb = out + (unsigned)(i >> 2) * (unsigned)kLdsNodeBytes;
        if ((i & 3) != 3) {
            float2* d = reinterpret_cast<float2*>(nb + (i & 3) * 16);
            d[0] = make_float2(q.x, q.z);
            d[1] = make_float2(q.y, q.w);
        } else {
            *reinterpret_cast<unsigned*>(nb + 48) = stage_ref(q.x) | (stage_ref(q.y) << 16);
        }
    }
}

// LDS layout of every traversal kernel: [per-lane stacks: stack_entries x BLOCK entries][staged nodes][flat packets]
template <int MODE, int BLOCK>
PT_DEV size_t traversal_stack_bytes_dev(const RenderParams& p) {
    return ((size_t)p.stack_entries * sizeof(typename StackOf<MODE>::type) * BLOCK + 15) & ~(size_t)15;
}
template <int MODE, int BLOCK>
PT_DEV size_t traversal_nodes_end_dev(const RenderParams& p) {
    return traversal_stack_bytes_dev<MODE, BLOCK>(p) + (MODE == kNodesLds ? (((size_t)p.n_nodes * kLdsNodeBytes + 15) & ~(size_t)15) : MODE == kNodesTreelet ? (size_t)p.treelet_nodes * 64 : 0);
}
template <int MODE, int BLOCK>
PT_DEV size_t traversal_lds_bytes_dev(const RenderParams& p) {      // == traversal_lds_bytes() on the host
    return traversal_nodes_end_dev<MODE, BLOCK>(p) + (size_t)p.n_flat * 100;    // packets + boxes + box-group masks of the big-triangle list
}
template <int MODE, int BLOCK>
PT_DEV void setup_traversal(const RenderParams& p, SceneView* sv, LaneStack<typename StackOf<MODE>::type>* stk) {
    typedef typename StackOf<MODE>::type StackT;
    stk->base = reinterpret_cast<StackT*>(pt_lds_raw) + threadIdx.x;          // [entry][lane]
    stk->stride = BLOCK;
    sv->nodes = p.nodes;
    sv->tris = p.tris;
    sv->meta = p.meta;
    sv->treelet = (unsigned)p.treelet_nodes;
    sv->n_flat = p.n_flat;
    sv->lds_nodes = nullptr;
    sv->lds_entries = p.stack_entries;
    sv->node_min_lanes = p.node_min_lanes;
    sv->leaf_min_lanes = p.leaf_min_lanes;
    sv->ovf = p.stack_ovf ? p.stack_ovf + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * BLOCK : nullptr;
    sv->ovf_stride = (unsigned)p.stack_ovf_lanes;
    float4* lds_flat = reinterpret_cast<float4*>(pt_lds_raw + traversal_nodes_end_dev<MODE, BLOCK>(p));
    for (int i = threadIdx.x; i < p.n_flat * 3; i += BLOCK) lds_flat[i] = p.tris[i];
    sv->lds_flat = lds_flat;
    // padded boxes of the listed triangles (what padded_bounds() gives a triangle in a leaf: 1e-5 of the largest
    // coordinate + 1e-6), per axis {lo, hi, hi, lo} so that (entry, exit) is one 8-byte read at + 0 or + 8
    float* lds_fbox = reinterpret_cast<float*>(lds_flat + p.n_flat * 3);
    unsigned* lds_fmask = reinterpret_cast<unsigned*>(lds_fbox + p.n_flat * 12);
    for (int i = threadIdx.x; i < p.n_fbox; i += BLOCK) {
        const int t = p.fbox_rep[i];
        lds_fmask[i] = p.fbox_mask[i];
        const float4 a = p.tris[t * 3], b = p.tris[t * 3 + 1], c = p.tris[t * 3 + 2];
        const float lox = fminf(fminf(a.x, a.w), b.z), hix = fmaxf(fmaxf(a.x, a.w), b.z);
        const float loy = fminf(fminf(a.y, b.x), b.w), hiy = fmaxf(fmaxf(a.y, b.x), b.w);
        const float loz = fminf(fminf(a.z, b.y), c.x), hiz = fmaxf(fmaxf(a.z, b.y), c.x);
        const float mx = fmaxf(__builtin_fabsf(lox), __builtin_fabsf(hix)), my = fmaxf(__builtin_fabsf(loy), __builtin_fabsf(hiy));
        const float mz = fmaxf(__builtin_fabsf(loz), __builtin_fabsf(hiz));
        const float pad = fmaxf(fmaxf(mx, my), mz) * 1e-5f + 1e-6f;
        float4* rec = reinterpret_cast<float4*>(lds_fbox + i * 12);
        rec[0] = make_float4(lox - pad, hix + pad, hix + pad, lox - pad);
        rec[1] = make_float4(loy - pad, hiy + pad, hiy + pad, loy - pad);
        rec[2] = make_float4(loz - pad, hiz + pad, hiz + pad, loz - pad);
    }
    sv->lds_fbox = reinterpret_cast<const char*>(lds_fbox);
    sv->lds_fmask = lds_fmask;
    sv->n_fbox = p.n_fbox;
    if (MODE == kNodesLds || MODE == kNodesTreelet) {
        float4* lds_nodes = reinterpret_cast<float4*>(pt_lds_raw + traversal_stack_bytes_dev<MODE, BLOCK>(p));
        stage_nodes<MODE>(p, lds_nodes);
        sv->lds_nodes = lds_nodes;
    }
    __syncthreads();
}

// statistics live in kStatRows rows of 8 counters; a block adds to the row picked by its index, so
// no single address sees more than a few dozen atomics per launch (one address saturates at
// ~88 atomics/us on MI355X, which cost a 32k-wave launch ~0.4 ms when every wave hit one word)
PT_DEV void stat_add(const RenderParams& p, int slot, unsigned long long v) {
    atomicAdd(&p.stats[(size_t)((blockIdx.x + blockIdx.y * 37u) % kStatRows) * kStatCols + slot], v);
}

PT_DEV unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

}  // namespace ptamd
