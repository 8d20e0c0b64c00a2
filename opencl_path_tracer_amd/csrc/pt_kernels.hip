// pt_kernels.hip -- gfx950 device code of libptamd.so.
//
// The hot path of the reference (prog.cl:292-389: gen_ray + trace_ray) as one persistent
// "render" kernel: every lane owns one pixel and runs ALL requested samples of it back to
// back with the whole path state in registers (the per-pixel LCG stream is sequential across
// samples, prog.cl:72-77 + main.cpp:382, so a pixel's samples can never run concurrently
// anyway).  A lane whose path ends regenerates its next camera ray in the same loop, so the
// wave stays converged on "traverse -> shade" instead of idling until the longest path of
// the wave is done.  The launch is persistent: waves pull (pass, tile) work items from a global
// counter (k_render); rnds/colors travel through HBM once per pass of 4-8 samples (~7 B/sample).
//
// Traversal: own BVH2 (64-B nodes holding both child boxes, 48-B triangle packets), near
// child first, far child pushed on a per-lane stack that lives in LDS ([entry][lane], bank =
// lane: conflict-free).  When the nodes fit (Cornell box: 941 nodes = 60 KB) every workgroup
// stages them in LDS, re-laid out so that the planes a ray needs are picked by address
// (stage_nodes, Trav::node_step); packets and larger trees are read through L1/L2.  The kernel is
// VALU-issue bound: the node visit is written for instruction count.
//
// Arithmetic: compiled with -ffp-contract=off; the only fused operations are the explicit
// fma calls, placed as DESIGN.md section 3 prescribes, so that results can be compared bit
// for bit with the CPU oracle (-fno-slp-vectorize: packed f32 ops cost what two scalar ones do,
// plus the shuffles).  '/' and sqrt are IEEE (hipcc default for HIP), sin/cos/pow are
// the double-precision polynomial routines below.  Box tests are NOT part of that contract:
// they are conservative (padded boxes, widened slabs) and only ever cull.
#include "pt_internal.hpp"

#include <algorithm>


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
PT_DEV void spec_sincos(float theta, float* s, float* c) {
    const double t = (double)theta;
    const int q = (int)fmad_(t, 0.63661977236758138, 0.5);
    const double qd = (double)q;
    double r = fmad_(qd, -1.5707963267948966, t);
    r = fmad_(qd, -6.123233995736766e-17, r);
    const double z = r * r;
    double ps = 1.6059043836821613e-10;
    ps = fmad_(ps, z, -2.505210838544172e-08);
    ps = fmad_(ps, z, 2.7557319223985893e-06);
    ps = fmad_(ps, z, -0.0001984126984126984);
    ps = fmad_(ps, z, 0.008333333333333333);
    ps = fmad_(ps, z, -0.16666666666666666);
    const double sr = fmad_(r * z, ps, r);
    double pc = -1.1470745597729725e-11;
    pc = fmad_(pc, z, 2.08767569878681e-09);
    pc = fmad_(pc, z, -2.755731922398589e-07);
    pc = fmad_(pc, z, 2.48015873015873e-05);
    pc = fmad_(pc, z, -0.001388888888888889);
    pc = fmad_(pc, z, 0.041666666666666664);
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
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double sdiv = f / (2.0 + f);
    const double z = sdiv * sdiv;
    double p = 0.10526315789473684;
    p = fmad_(p, z, 0.11764705882352941);
    p = fmad_(p, z, 0.13333333333333333);
    p = fmad_(p, z, 0.15384615384615385);
    p = fmad_(p, z, 0.18181818181818182);
    p = fmad_(p, z, 0.22222222222222221);
    p = fmad_(p, z, 0.2857142857142857);
    p = fmad_(p, z, 0.4);
    p = fmad_(p, z, 0.66666666666666663);
    p = fmad_(p, z, 2.0);
    const double lnm = sdiv * p;
    const double lg2 = fmad_(lnm, 1.4426950408889634, (double)e);
    const double w = (double)y * lg2;
    if (!(w > -126.0)) return 0.0f;
    if (w >= 128.0) return __builtin_inff();
    const double nd = __builtin_floor(w + 0.5);
    const double g = (w - nd) * 0.6931471805599453;
    double q = 2.08767569878681e-09;
    q = fmad_(q, g, 2.505210838544172e-08);
    q = fmad_(q, g, 2.755731922398589e-07);
    q = fmad_(q, g, 2.7557319223985893e-06);
    q = fmad_(q, g, 2.48015873015873e-05);
    q = fmad_(q, g, 0.0001984126984126984);
    q = fmad_(q, g, 0.001388888888888889);
    q = fmad_(q, g, 0.008333333333333333);
    q = fmad_(q, g, 0.041666666666666664);
    q = fmad_(q, g, 0.16666666666666666);
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
    if (p.pixel_map == 1) {
        // strided map: every wave gets pixels from all over the rank's tile set, so that the total
        // work per wave is nearly the same (matters when a rank has ~1 wave per SIMD slot: N >= 4)
        const int npix = p.width * p.local_rows;
        const int nw = (npix + 63) >> 6;
        const int li = lane * nw + wave;
        PixelId r;
        if (wave >= nw || li >= npix) {
            r.li = -1;
            r.gid = 0;
            return r;
        }
        const int lrow = li / p.width, x = li - lrow * p.width;
        const int grow = ((lrow / p.rows_per_block) * p.world + p.rank) * p.rows_per_block + (lrow % p.rows_per_block);
        r.li = li;
        r.gid = grow * p.width + x;
        return r;
    }
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
struct SceneView {
    const float4* nodes;   // global or LDS
    const float4* tris;    // global or LDS
    const TriMeta* meta;   // global
};

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
PT_DEV float tri_test(const float4 a, const float4 b, const float4 c, f3 P, f3 Vd, float limit) {
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
// Scenes whose nodes are staged in LDS use 16-bit entries (16 waves per CU fit next to the nodes).
template <class T>
struct LaneStack {
    T* base;        // already offset by the lane
    int stride;     // entries are `stride` elements apart
};

struct WorkCount {
    unsigned nodes, tris;     // per-lane visits / tests
    unsigned wnodes, wtris;   // wave-level executions of the two bodies (counted by the first active lane)
};
PT_DEV bool first_active_lane() {
    const unsigned long long m = __ballot(1);
    return (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1;
}

// Closest hit over the whole scene; ties in t go to the lower encounter rank (the triangle the
// reference's traversal, prog.cl:113-184, meets first).
// While-while traversal, one "round" at a time: every lane first descends through interior nodes
// until it holds a leaf (or has finished), then all lanes holding leaves intersect them.  The
// per-lane state survives between rounds so that a persistent kernel can hand a finished lane
// its next ray while the others keep going.
//
// Node references (`cur`, child slots, stack entries):
//   StackT = unsigned        nodes read from global memory as the host packed them:
//                            >= 0 interior index, < 0 leaf ~(first << 3 | count - 1), kDone finished
//   StackT = unsigned short  nodes staged in LDS by stage_nodes(), which also re-encodes the child
//                            slots to 16 bits: < 0x7fff interior index, 0x7fff finished,
//                            0x8000 | (first << 3 | count - 1) leaf -- nothing to encode or decode on
//                            a push or pop -- and swizzles the box quads (see node_step).
template <class StackT>
struct Trav {
    static constexpr bool kSel = sizeof(StackT) == 2;
    static constexpr int kDone = kSel ? 0x7fff : 0x7fffffff;
    f3 P, D, inv;
    float best_t;
    int best;      // packed triangle index of the closest hit so far, -1 none
    char* tos;     // top of this lane's stack (entry 0 = sentinel kDone), as a byte address
    int stride;    // bytes between entries
    int cur;
    int k;         // single-step schedules: next triangle of the current leaf
    int pend;      // round(): a leaf met during the node phase and not yet intersected (0: none)
    int onx, ony, onz;   // kSel: byte offset of the entry-plane pair of each axis inside a node
    f3 cn, cf;           // kSel: -(P * inv) widened down / up (entry / exit distance = fma(plane, inv, c))

    PT_DEV static bool is_node(int c) { return kSel ? c < 0x7fff : (unsigned)c < 0x7fffffffu; }
    PT_DEV static bool is_leaf(int c) { return kSel ? c > 0x7fff : c < 0; }
    PT_DEV static int leaf_bits(int c) { return kSel ? (c & 0x7fff) : ~c; }

    PT_DEV void begin(f3 P_, f3 D_, const LaneStack<StackT> stk) {
        P = P_;
        D = D_;
        inv = mk(__builtin_amdgcn_rcpf(D_.x), __builtin_amdgcn_rcpf(D_.y), __builtin_amdgcn_rcpf(D_.z));
        best_t = __builtin_inff();
        best = -1;
        tos = reinterpret_cast<char*>(stk.base);
        stride = stk.stride * (int)sizeof(StackT);
        *reinterpret_cast<StackT*>(tos) = (StackT)kDone;
        cur = 0;        // the root is always an interior node
        k = 0;
        pend = 0;
        onx = __float_as_int(inv.x) < 0 ? 8 : 0;
        ony = __float_as_int(inv.y) < 0 ? 24 : 16;
        onz = __float_as_int(inv.z) < 0 ? 40 : 32;
        if (kSel) {
            // Distance to a plane as ONE fma: plane * inv - P * inv.  The product P * inv is rounded
            // (half an ulp of |P * inv|, which can dwarf the distance itself), so the entry constant
            // is lowered and the exit constant raised by 4 such half-ulps: entry distances come out
            // too small, exit distances too large, never the other way round.  inf - inf = NaN
            // (direction component 0 or underflowing) is ignored by max3/min3: that slab counts as
            // entered, which only costs work.
            const float px = P.x * inv.x, py = P.y * inv.y, pz = P.z * inv.z;
            const float k = 2.3841858e-07f;    // 2^-22
            const float ex = __builtin_fabsf(px) * k, ey = __builtin_fabsf(py) * k, ez = __builtin_fabsf(pz) * k;
            cn = mk(-(px + ex), -(py + ey), -(pz + ez));
            cf = mk(-(px - ex), -(py - ey), -(pz - ez));
        }
    }
    PT_DEV void idle() { cur = kDone; }
    PT_DEV bool done() const { return cur == kDone; }

    // One interior-node visit: slab-test both children, descend into the nearer hit one, push the
    // other.  The visit is ONE basic block with one LDS round trip: the top of the stack is fetched
    // together with the node (it is the next node if neither child is hit), and the far child is
    // stored above the top unconditionally; only the stack pointer moves conditionally.
    //
    // DEFER (used by round() for nodes read from global memory): when the nearer child is a leaf and
    // no leaf is pending, the leaf is remembered and the descent goes on with the other child or
    // the stack; round() intersects it after the node phase.  A lane then goes through about half
    // as many node-phase / leaf-phase alternations, each of which the whole wave waits out -- at the
    // price of 3 % more node visits (the pending leaf cannot prune yet).  Measured: L1/L2 node path
    // +5..8 % (Cornell 1,042 -> 1,098, MESH-100k 436 -> 469 Msamples/s); LDS node path +-0, where
    // the 5 extra VALU instructions per visit cost what the saved alternations bring.
    template <bool COUNT, bool DEFER = false>
    PT_DEV void node_step(const SceneView& sv, WorkCount* wc) {
        const float kWiden = 1.0000005f;   // > 4 ulp: covers rcp + sub + mul (or the fma; begin() covers P * inv)
        const int top = (int)*reinterpret_cast<const StackT*>(tos);
        float ln, lf, rn, rf;
        int li, ri;
        if (COUNT) { wc->nodes++; if (first_active_lane()) wc->wnodes++; }
        if (kSel) {
            // Swizzled quads {L.lo, R.lo, L.hi, R.hi}: the entry / exit planes of both children are
            // picked by ADDRESS from the sign of the ray direction instead of by 12 v_min/v_max
            // (4-cycle ops on gfx950, tools/micro/exec_ops.hip; the adds that form the addresses
            // are 2-cycle ops).
            const char* nb = reinterpret_cast<const char*>(sv.nodes) + ((size_t)(unsigned)cur << 6);
            const float2 ex = *reinterpret_cast<const float2*>(nb + onx), xx = *reinterpret_cast<const float2*>(nb + (onx ^ 8));
            const float2 ey = *reinterpret_cast<const float2*>(nb + ony), xy = *reinterpret_cast<const float2*>(nb + (ony ^ 8));
            const float2 ez = *reinterpret_cast<const float2*>(nb + onz), xz = *reinterpret_cast<const float2*>(nb + (onz ^ 8));
            const int2 ch = *reinterpret_cast<const int2*>(nb + 48);
            ln = fmaxf(fmaxf(fmaf_(ex.x, inv.x, cn.x), fmaf_(ey.x, inv.y, cn.y)), fmaf_(ez.x, inv.z, cn.z));
            rn = fmaxf(fmaxf(fmaf_(ex.y, inv.x, cn.x), fmaf_(ey.y, inv.y, cn.y)), fmaf_(ez.y, inv.z, cn.z));
            lf = fminf(fminf(fmaf_(xx.x, inv.x, cf.x), fmaf_(xy.x, inv.y, cf.y)), fmaf_(xz.x, inv.z, cf.z)) * kWiden;
            rf = fminf(fminf(fmaf_(xx.y, inv.x, cf.x), fmaf_(xy.y, inv.y, cf.y)), fmaf_(xz.y, inv.z, cf.z)) * kWiden;
            li = ch.x;
            ri = ch.y;
        } else {
            // (unsigned 32-bit byte offsets: scalar base + vector offset addressing, no 64-bit address math)
            const char* nb = reinterpret_cast<const char*>(sv.nodes);
            const unsigned off = (unsigned)cur << 6;
            const float4 qx = *reinterpret_cast<const float4*>(nb + off);
            const float4 qy = *reinterpret_cast<const float4*>(nb + (off + 16u));
            const float4 qz = *reinterpret_cast<const float4*>(nb + (off + 32u));
            const float4 qr = *reinterpret_cast<const float4*>(nb + (off + 48u));
            // (plane - P) * inv: ~2 ulp per distance, covered by the 4-ulp widening.  The one-fma form
            // is NOT used here: with min/max picking the planes, a NaN from inf - inf (direction
            // component 0 or underflowing) would be replaced by the OTHER plane's distance and cull
            // real hits (tests/test_gpu_parity.py::test_closest_hit_adversarial_rays).
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

    // exact test of packed triangle ti against the ray, keeping the closest (ties: lower rank)
    template <bool COUNT>
    PT_DEV void tri_step(const SceneView& sv, int ti, WorkCount* wc) {
        const char* tb = reinterpret_cast<const char*>(sv.tris);
        const unsigned off = (unsigned)ti * 48u;
        const float4 a = *reinterpret_cast<const float4*>(tb + off), b = *reinterpret_cast<const float4*>(tb + (off + 16u)), c = *reinterpret_cast<const float4*>(tb + (off + 32u));
        if (COUNT) { wc->tris++; if (first_active_lane()) wc->wtris++; }
        const float t = tri_test<kSel>(a, b, c, P, D, best_t * 1.000002f);
        if (t > 0.0f) {
            bool better = t < best_t;
            if (t == best_t && best >= 0) better = sv.meta[ti].rank < sv.meta[best].rank;
            if (better) { best_t = t; best = ti; }
        }
    }

    PT_DEV void pop() {
        cur = (int)*reinterpret_cast<const StackT*>(tos);
        tos -= stride;
    }

    template <bool COUNT>
    PT_DEV void round(const SceneView& sv, WorkCount* wc) {
        constexpr bool kDefer = !kSel;
        while (is_node(cur)) node_step<COUNT, kDefer>(sv, wc);
        if (kDefer && pend != 0) {            // met first, so nearer: intersect it first
            const int v = leaf_bits(pend);
            const int first = v >> 3, count = (v & 7) + 1;
            for (int j = 0; j < count; ++j) tri_step<COUNT>(sv, first + j, wc);
            pend = 0;
        }
        while (is_leaf(cur)) {
            const int popped = (int)*reinterpret_cast<const StackT*>(tos);     // in flight during the triangle tests
            const int v = leaf_bits(cur);
            const int first = v >> 3, count = (v & 7) + 1;
            for (int j = 0; j < count; ++j) tri_step<COUNT>(sv, first + j, wc);
            cur = popped;
            tos -= stride;
        }
    }

    // one triangle of the current leaf (single-step schedules)
    template <bool COUNT>
    PT_DEV void leaf_step(const SceneView& sv, WorkCount* wc) {
        const int v = leaf_bits(cur);
        const int first = v >> 3, count = (v & 7) + 1;
        tri_step<COUNT>(sv, first + k, wc);
        if (++k >= count) {
            k = 0;
            pop();
        }
    }

    // Voting schedule: each step the wave runs ONE body -- a node visit or a single triangle test --
    // whichever more lanes are waiting for.  Lanes in the minority wait (and accumulate), so neither
    // body is ever executed for a thin tail of lanes as in the while-while loops.
    template <bool COUNT>
    PT_DEV void vote_step(const SceneView& sv, WorkCount* wc) {
        const bool want_node = is_node(cur);
        const bool want_tri = is_leaf(cur);
        const int nn = __popcll(__ballot(want_node)), nt = __popcll(__ballot(want_tri));
        if (nn >= nt) {
            if (want_node) node_step<COUNT>(sv, wc);
        } else if (want_tri) {
            leaf_step<COUNT>(sv, wc);
        }
    }
};

template <class StackT, bool COUNT, bool VOTE>
PT_DEV int closest_hit(const SceneView& sv, f3 P, f3 D, const LaneStack<StackT> stk, float* t_out, WorkCount* wc) {
    Trav<StackT> tr;
    tr.begin(P, D, stk);
    if (VOTE) {
        while (__ballot(!tr.done()) != 0) tr.template vote_step<COUNT>(sv, wc);
    } else {
        while (!tr.done()) tr.template round<COUNT>(sv, wc);
    }
    *t_out = tr.best_t;
    return tr.best;
}

// ---------------------------------------------------------------------------- BSDF sampling
struct RayPD {
    f3 P, D;
};

// prog.cl:186-218
PT_DEV RayPD new_ray_diffuse(f3 hp, f3 N, float rnd1, float rnd2) {
    const float E = 0.001f;
    const bool yaxis = __builtin_fabsf(N.x) <= E && __builtin_fabsf(N.z) <= E;
    const float other = yaxis ? N.y : N.x;
    const float rl = 1.0f / __builtin_sqrtf(fmaf_(N.z, N.z, other * other));
    const f3 Z = yaxis ? mk(0.0f, -N.z * rl, N.y * rl) : mk(-N.z * rl, 0.0f, N.x * rl);
    const f3 X = cross3(N, Z);
    const float r = __builtin_sqrtf(rnd1);
    const float theta = (float)(6.283185307179586 * (double)rnd2);
    float sn, cs;
    spec_sincos(theta, &sn, &cs);
    const float x = r * cs, y = r * sn, z = __builtin_sqrtf(1.0f - rnd1);
    f3 d = X * x;
    d = madd(N, z, d);
    d = madd(Z, y, d);
    RayPD o;
    o.P = madd(N, E, hp);
    o.D = normalize3(d);
    return o;
}

// prog.cl:219-222
PT_DEV f3 fresnel(f3 F0, f3 N, f3 D) {
    const float cosa = __builtin_fabsf(dot3(N, D));
    const float p5 = spec_pow5(1.0f - cosa);
    return mk(fmaf_(1.0f - F0.x, p5, F0.x), fmaf_(1.0f - F0.y, p5, F0.y), fmaf_(1.0f - F0.z, p5, F0.z));
}

// prog.cl:223-227
PT_DEV RayPD new_ray_specular(f3 hp, f3 N, f3 oldD) {
    const float cosa = dot3(N, oldD);
    RayPD o;
    o.D = normalize3(oldD - (N * cosa) * 2.0f);
    o.P = madd(N, 0.001f, hp);
    return o;
}

// prog.cl:228-245; *flipped = the path crossed the interface (in = !in)
PT_DEV RayPD new_ray_refractive(f3 hp, f3 N, f3 F0, float n, f3 oldD, bool in, float rnd, bool* flipped) {
    if (in) n = 1.0f / n;
    const float cosa = dot3(-oldD, N);
    const float disc = 1.0f - (fmaf_(-cosa, cosa, 1.0f) / n) / n;
    const f3 F = fresnel(F0, N, oldD);
    const float prob = ((F.x + F.y) + F.z) / 3.0f;
    const bool refr = disc > 0.0f && rnd > prob;
    *flipped = refr;
    // both candidate directions before normalisation; one normalize serves either branch
    const f3 dn = mk(oldD.x / n, oldD.y / n, oldD.z / n);
    const f3 dr = madd(N, cosa / n - __builtin_sqrtf(disc), dn);
    const f3 dm = oldD - (N * dot3(N, oldD)) * 2.0f;
    RayPD o;
    o.D = normalize3(refr ? dr : dm);
    o.P = madd(N, refr ? -0.001f : 0.001f, hp);
    return o;
}

// ---------------------------------------------------------------------------- path state + shading
// The path state of prog.cl:307-316 lives in plain local variables (registers), passed by
// reference: P, D, the four factors, the colour, the LCG state and the inside-glass flag.
#define PT_PATH_ARGS f3 &rP, f3 &rD, f3 &fL, f3 &fB, f3 &fS, f3 &fR, f3 &color, int &seed, bool &inside

// one iteration body of prog.cl:317-366 for a ray that hit packed triangle `ti` at `t`
PT_DEV void shade_hit(PT_PATH_ARGS, const RenderParams& p, const float4* tris, const TriMeta* meta, int ti, float t) {
    const float4 c = tris[ti * 3 + 2];
    f3 N = mk(c.y, c.z, c.w);
    const f3 hp = madd(rD, t, rP);
    const pt_material* __restrict__ m = &p.mats[meta[ti].mati];
    const int type = m->type;
    if (p.iterations == 1) color = ldf3(m->kd) + ldf3(m->emission);         // prog.cl:323-325
    if (dot3(rD, N) > 0.0f) N = -N;                                         // prog.cl:326-328
    if (type == 0 || type == 3) {
        // diffuse (prog.cl:329-340) and emitter (prog.cl:358-366) both continue with a cosine-
        // sampled ray drawn from two LCG values; the emitter's cosine uses the OLD direction.
        const float inten = max0(dot3(-rD, N));
        const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
        const RayPD nr = new_ray_diffuse(hp, N, rnd1, rnd2);
        if (type == 0) {
            const float idiff = max0(dot3(nr.D, N));
            fL = fL * (ldf3(m->kd) * idiff);
            const f3 view = normalize3(ldf3(p.cam.eye) - hp);
            const f3 halfway = normalize3(view + nr.D);
            const float ispec = max0(dot3(N, halfway));
            // m->_pad = 1: ks is exactly 0 and shininess is finite >= 0, so ks * pow(...) is +0 whatever the
            // (finite) power is -- skip the double-precision pow (set by pt_upload_materials)
            const float pw = m->_pad ? 1.0f : spec_pow(ispec, m->shininess);
            fB = fB * (ldf3(m->ks) * pw);
        } else {
            const f3 e = ((ldf3(m->emission) * (fL + fB)) * fS) * fR;
            color = madd(e, inten, color);
        }
        rP = nr.P;
        rD = nr.D;
    } else if (type == 1 || type == 2) {
        // mirror (prog.cl:341-345) and dielectric (prog.cl:346-357, 228-245) share the Fresnel
        // term and the mirror direction; the dielectric may pick the refracted direction instead.
        const f3 oldD = rD;
        const f3 F0 = ldf3(m->F0);
        const f3 F = fresnel(F0, N, oldD);
        f3 dsel = oldD - (N * dot3(N, oldD)) * 2.0f;
        bool refr = false;
        if (type == 2) {
            float n = m->n;
            if (inside) n = 1.0f / n;
            const float rnd = lcg_rand(seed);
            const float cosa = dot3(-oldD, N);
            const float disc = 1.0f - (fmaf_(-cosa, cosa, 1.0f) / n) / n;
            const float prob = ((F.x + F.y) + F.z) / 3.0f;
            refr = disc > 0.0f && rnd > prob;
            if (refr) {
                const f3 dn = mk(oldD.x / n, oldD.y / n, oldD.z / n);
                dsel = madd(N, cosa / n - __builtin_sqrtf(disc), dn);
                const float k = 1.0f / (1.0f - prob);
                fR = (fR * mk(1.0f - F.x, 1.0f - F.y, 1.0f - F.z)) * k;
                inside = !inside;
            } else {
                const float k = 1.0f / prob;
                fR = (fR * F) * k;
            }
        } else {
            fS = fS * F;
        }
        rD = normalize3(dsel);
        rP = madd(N, refr ? -0.001f : 0.001f, hp);
    }
    // any other type: the ray is left unchanged and the loop hits the same surface again
}

PT_DEV f3 running_mean(f3 acc, f3 color, int s) {   // prog.cl:379
    const float cs = (float)s, cs1 = (float)(s + 1);
    return mk(fmaf_(acc.x, cs, color.x) / cs1, fmaf_(acc.y, cs, color.y) / cs1, fmaf_(acc.z, cs, color.z) / cs1);
}

// ---------------------------------------------------------------------------- LDS staging
extern __shared__ __attribute__((aligned(16))) unsigned char pt_lds_raw[];

// Nodes staged in LDS are re-laid out on the way in: the three box quads {L.lo, L.hi, R.lo, R.hi}
// become {L.lo, R.lo, L.hi, R.hi}, so that one 8-byte read at (quad + 0 | 8) returns the entry
// (or exit) planes of BOTH children for a ray whose direction sign on that axis is known
// (Trav::node_step, kSel).  The child slots of the fourth quad are re-encoded to the 16-bit
// reference form of Trav<unsigned short>.
PT_DEV float stage_ref(float slot) {
    const int r = __float_as_int(slot);
    return __int_as_float(r < 0 ? (0x8000 | ~r) : r);
}
PT_DEV void stage_nodes(const RenderParams& p, float4* lds_nodes) {
    const int nn = p.n_nodes * 4;
    for (int i = threadIdx.x; i < nn; i += blockDim.x) {
        const float4 q = p.nodes[i];
        lds_nodes[i] = (i & 3) == 3 ? make_float4(stage_ref(q.x), stage_ref(q.y), q.z, q.w) : make_float4(q.x, q.z, q.y, q.w);
    }
}

PT_DEV void stage_scene(const RenderParams& p, float4* lds_nodes, float4* lds_tris) {
    const int nt = p.n_tris * 3;
    stage_nodes(p, lds_nodes);
    for (int i = threadIdx.x; i < nt; i += blockDim.x) lds_tris[i] = p.tris[i];
    __syncthreads();
}

// statistics live in kStatRows rows of 8 counters; a block adds to the row picked by its index, so
// no single address sees more than a few dozen atomics per launch (one address saturates at
// ~88 atomics/us on MI355X, which cost a 32k-wave launch ~0.4 ms when every wave hit one word)
PT_DEV void stat_add(const RenderParams& p, int slot, unsigned long long v) {
    atomicAdd(&p.stats[(size_t)((blockIdx.x + blockIdx.y * 37u) % kStatRows) * 8 + slot], v);
}

PT_DEV unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------- kernels
// gen_ray, prog.cl:384-389
__global__ void __launch_bounds__(256) k_gen_ray(RenderParams p) {
    const PixelId px = pixel_of_thread(p);
    if (px.li < 0) return;
    int seed = p.rnds[px.li];
    const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
    f3 P, D;
    camera_get_ray(px.gid, p.cam, rnd1, rnd2, &P, &D);
    p.rnds[px.li] = seed;
    float4* r = reinterpret_cast<float4*>(&p.rays[px.li]);
    r[0] = make_float4(P.x, P.y, P.z, 0.0f);
    r[1] = make_float4(D.x, D.y, D.z, 0.0f);
}

// All samples [s_begin, s_end) of ONE pixel, path state in registers: nsamples x (gen_ray + trace_ray).
// SPLIT: trace_ray alone (prog.cl:292-381) -- the ray comes from, and is left in, the rays buffer.
template <bool SPLIT, class StackT, bool COUNT, bool VOTE>
PT_DEV void render_pixel(const RenderParams& p, const SceneView& sv, const LaneStack<StackT> stk, const PixelId px,
                         int s_begin, int s_end, unsigned long long* segs, unsigned long long* samples, WorkCount* wc) {
    f3 rP = mk(0.f, 0.f, 0.f), rD = mk(0.f, 0.f, 1.f);
    f3 fL = mk(1.f, 1.f, 1.f), fB = fL, fS = fL, fR = fL, color = mk(0.f, 0.f, 0.f);
    bool inside = false;
    int seed = p.rnds[px.li];
    f3 acc = mk(0.0f, 0.0f, 0.0f);
    if (s_begin != 0) {                        // prog.cl:312-314: sample 0 starts from black
        const float4 c = p.colors[px.li];
        acc = mk(c.x, c.y, c.z);
    }
    int s = s_begin;
    int bounce = 0;
    bool fresh = true;
    const int camX = (int)p.cam.XM;
    const float pix_x = (float)(px.gid % camX), pix_y = (float)(px.gid / camX);      // prog.cl:84-85
    for (;;) {
        if (fresh) {                           // a lane whose path ended starts its next sample right here
            if (s == s_end) break;
            fL = mk(1.f, 1.f, 1.f);            // prog.cl:307-316
            fB = fL;
            fS = fL;
            fR = fL;
            color = mk(0.f, 0.f, 0.f);
            inside = false;
            if (SPLIT) {
                const float4* r = reinterpret_cast<const float4*>(&p.rays[px.li]);
                const float4 a = r[0], b = r[1];
                rP = mk(a.x, a.y, a.z);
                rD = mk(b.x, b.y, b.z);
            } else {
                const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
                camera_get_ray_xy(pix_x, pix_y, p.cam, rnd1, rnd2, &rP, &rD);
            }
            bounce = 0;
            fresh = false;
        }
        bool finished = true;
        if (bounce < p.iterations) {
            float t;
            const int ti = closest_hit<StackT, COUNT, VOTE>(sv, rP, rD, stk, &t, wc);
            ++*segs;
            if (ti >= 0) {
                shade_hit(rP, rD, fL, fB, fS, fR, color, seed, inside, p, sv.tris, sv.meta, ti, t);
                ++bounce;
                finished = (bounce >= p.iterations);
            }
        }
        if (finished) {
            acc = running_mean(acc, color, s);
            ++s;
            ++*samples;
            fresh = true;
        }
    }
    p.colors[px.li] = make_float4(acc.x, acc.y, acc.z, 0.0f);
    p.rnds[px.li] = seed;
    if (SPLIT) {
        float4* r = reinterpret_cast<float4*>(&p.rays[px.li]);
        r[0] = make_float4(rP.x, rP.y, rP.z, 0.0f);
        r[1] = make_float4(rD.x, rD.y, rD.z, 0.0f);
    }
}

// The render kernel.  One wave = one 8x8 pixel tile; each lane owns one pixel.
// LDS_SCENE: 0 = nodes and packets through L1/L2; 1 = both staged in LDS; 2 = nodes staged in LDS,
// packets through L1/L2 (the node loads are 3/4 of the traversal's gather instructions).
//
// Launch modes:
//  * plain (p.tile_counter == 0): wave w of the grid renders tile w, all samples.
//  * persistent (p.tile_counter != 0): the grid only fills the machine and every WAVE pulls its next
//    work item from a global counter as soon as it is done -- it never waits for the other waves of
//    its workgroup, whose tiles may take 30 % longer (tiles over the spheres vs. bare walls).
//  * persistent with chained passes (p.chunk_spp > 0): a work item is (pass, tile) = chunk_spp
//    samples of one tile, handed out pass-major.  The producer of a tile's previous pass was dequeued
//    n_tiles items earlier by a resident wave, so waiting for it cannot deadlock; its rnds/colors
//    reach this wave -- possibly on another XCD -- through an agent-scope release (producer: stores,
//    L2 write-back, tile_done[tile] = pass + 1) and acquire (consumer: poll, L1 invalidate, loads).
template <bool SPLIT, int LDS_SCENE, int BLOCK, class StackT, bool COUNT, int MINW, bool VOTE = false>
__global__ void __launch_bounds__(BLOCK, MINW) k_render(RenderParams p) {
    LaneStack<StackT> stk;
    stk.base = reinterpret_cast<StackT*>(pt_lds_raw) + threadIdx.x;          // [entry][lane]
    stk.stride = BLOCK;
    WorkCount wc;
    wc.nodes = 0;
    wc.tris = 0;
    wc.wnodes = 0;
    wc.wtris = 0;
    SceneView sv;
    sv.nodes = p.nodes;
    sv.tris = p.tris;
    sv.meta = p.meta;
    if (LDS_SCENE) {
        float4* lds_nodes = reinterpret_cast<float4*>(pt_lds_raw + (size_t)p.stack_entries * sizeof(StackT) * BLOCK);
        float4* lds_tris = lds_nodes + p.n_nodes * 4;
        if (LDS_SCENE == 1) {
            stage_scene(p, lds_nodes, lds_tris);
            sv.tris = lds_tris;
        } else {
            stage_nodes(p, lds_nodes);
            __syncthreads();
        }
        sv.nodes = lds_nodes;
    }
    const bool lane0 = (threadIdx.x & 63) == 0;
    const bool chained = p.tile_counter != nullptr && p.chunk_spp > 0;
    const int n_pass = chained ? (p.nsamples + p.chunk_spp - 1) / p.chunk_spp : 1;
    unsigned long long segs = 0, samples = 0, segs_before_item = 0, item_lane_steps = 0;
    int tile = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    int pass = 0;
    for (;;) {
        if (p.tile_counter) {                                    // ---- fetch the next work item
            int t = 0;
            if (lane0) t = (int)atomicAdd(p.tile_counter, 1u);
            t = __shfl(t, 0, 64);
            pass = chained ? t / p.n_tiles : 0;
            tile = t - pass * p.n_tiles;
            if (chained ? pass >= n_pass : tile >= p.n_tiles) break;
            if (pass > 0) {                                      // ---- acquire the tile's previous pass
                unsigned seen = 0;
                for (;;) {
                    if (lane0) seen = __hip_atomic_load(&p.tile_done[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    seen = __shfl(seen, 0, 64);
                    if (seen >= (unsigned)pass) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        const int s_begin = p.first_sample + (chained ? pass * p.chunk_spp : 0);
        const int s_end = chained ? min(s_begin + p.chunk_spp, p.first_sample + p.nsamples) : p.first_sample + p.nsamples;
        const PixelId px = pixel_of_wave(p, tile);
        if (px.li >= 0) render_pixel<SPLIT, StackT, COUNT, VOTE>(p, sv, stk, px, s_begin, s_end, &segs, &samples, &wc);
        if (COUNT) {      // segment-steps the wave executed for this item = 64 x the busiest lane's segments
            unsigned long long mx = segs - segs_before_item;
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned long long)__shfl_down(mx, off, 64));
            if (lane0) item_lane_steps += mx * 64ull;
            segs_before_item = segs;
        }
        if (chained) {                                           // ---- release this pass of the tile
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // keep the wait the compiler may drop (guide, G16)
            if (lane0) __hip_atomic_store(&p.tile_done[tile], (unsigned)(pass + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!p.tile_counter) break;
    }
    segs = wave_sum(segs);
    samples = wave_sum(samples);
    if (COUNT) {
        const unsigned long long wn = wave_sum((unsigned long long)wc.nodes), wt = wave_sum((unsigned long long)wc.tris);
        const unsigned long long wwn = wave_sum((unsigned long long)wc.wnodes), wwt = wave_sum((unsigned long long)wc.wtris);
        if (lane0 && p.stats) {
            stat_add(p, 2, wn);
            stat_add(p, 3, wt);
            stat_add(p, 4, wwn);
            stat_add(p, 5, wwt);
            stat_add(p, 6, item_lane_steps);
        }
    }
    if (lane0 && p.stats) {
        stat_add(p, 0, segs);
        stat_add(p, 1, samples);
    }
}

// Sliced variant of the render kernel (option traversal = 2).  In k_render a lane whose traversal
// ends early idles until the slowest ray of the wave is done (measured: the node body runs 57 times
// per wave and segment while the average lane needs 9.4).  Here a traversal is advanced by at most
// `slice` rounds per trip of the main loop; lanes that finished shade and start their next segment
// in the same trip, lanes with long traversals simply carry their Trav state into the next trip.
template <int BLOCK, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_render_sliced(RenderParams p, int slice) {
    LaneStack<unsigned> stk;
    stk.base = reinterpret_cast<unsigned*>(pt_lds_raw) + threadIdx.x;
    stk.stride = BLOCK;
    WorkCount wc;
    SceneView sv;
    sv.nodes = p.nodes;
    sv.tris = p.tris;
    sv.meta = p.meta;
    const PixelId px = pixel_of_thread(p);
    unsigned long long segs = 0, samples = 0;
    if (px.li >= 0) {
        f3 rP = mk(0.f, 0.f, 0.f), rD = mk(0.f, 0.f, 1.f);
        f3 fL = mk(1.f, 1.f, 1.f), fB = fL, fS = fL, fR = fL, color = mk(0.f, 0.f, 0.f);
        bool inside = false;
        int seed = p.rnds[px.li];
        f3 acc = mk(0.0f, 0.0f, 0.0f);
        if (p.first_sample != 0) {
            const float4 c = p.colors[px.li];
            acc = mk(c.x, c.y, c.z);
        }
        int s = p.first_sample;
        const int s_end = p.first_sample + p.nsamples;
        int bounce = 0;
        bool fresh = true, traversing = false;
        Trav<unsigned> tr;
        tr.begin(rP, rD, stk);
        tr.idle();
        for (;;) {
            if (!traversing) {
                if (fresh) {
                    if (s == s_end) break;
                    fL = mk(1.f, 1.f, 1.f);    // prog.cl:307-316
                    fB = fL;
                    fS = fL;
                    fR = fL;
                    color = mk(0.f, 0.f, 0.f);
                    inside = false;
                    const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
                    camera_get_ray(px.gid, p.cam, rnd1, rnd2, &rP, &rD);
                    bounce = 0;
                    fresh = false;
                }
                traversing = true;
                if (bounce < p.iterations) {
                    tr.begin(rP, rD, stk);
                    ++segs;
                } else {                        // iterations == 0: nothing to trace
                    tr.idle();
                    tr.best = -1;
                }
            }
            for (int r = 0; r < slice; ++r) {
                if (__ballot(!tr.done()) == 0) break;
                if (!tr.done()) tr.template round<false>(sv, &wc);
            }
            if (tr.done()) {
                traversing = false;
                bool finished = true;
                if (tr.best >= 0) {
                    shade_hit(rP, rD, fL, fB, fS, fR, color, seed, inside, p, sv.tris, sv.meta, tr.best, tr.best_t);
                    ++bounce;
                    finished = (bounce >= p.iterations);
                }
                if (finished) {
                    acc = running_mean(acc, color, s);
                    ++s;
                    ++samples;
                    fresh = true;
                }
            }
        }
        p.colors[px.li] = make_float4(acc.x, acc.y, acc.z, 0.0f);
        p.rnds[px.li] = seed;
    }
    segs = wave_sum(segs);
    samples = wave_sum(samples);
    if ((threadIdx.x & 63) == 0 && p.stats) {
        stat_add(p, 0, segs);
        stat_add(p, 1, samples);
    }
}

// ============================================================================ wavefront
// Stream-compacted formulation of the same path (BASELINE north_star): one pass = one sample of
// every local pixel.  generate -> for each bounce { intersect ; shade } with the path state SoA in
// HBM (WfParams) and index queues between the stages.
//   wf_generate : 2 LCG draws + camera ray per pixel (prog.cl:384-389), state init (prog.cl:307-316)
//   wf_intersect: each wave owns 256 consecutive entries of a ray queue and refills a lane as soon
//                 as its traversal ends (__ballot/__popcll rank inside the wave's range), so no
//                 lane idles while its neighbours finish long traversals.  At the end the block
//                 compacts its rays into three class queues by the material type they hit
//                 (order-preserving ballot scan through LDS, 3 global atomics per 1,024 rays).
//   wf_shade    : one block row per class -> waves are material-coherent.  Survivors go to the
//                 next bounce's ray queues; paths that end (miss / last bounce) fold their colour
//                 into the running mean (prog.cl:379) and store the LCG state.
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

PT_DEV void wf_finalize(const WfParams& w, int li, f3 color, int seed) {
    f3 acc = mk(0.0f, 0.0f, 0.0f);
    if (w.sample != 0) {
        const float4 c = w.rp.colors[li];
        acc = mk(c.x, c.y, c.z);
    }
    acc = running_mean(acc, color, w.sample);
    w.rp.colors[li] = make_float4(acc.x, acc.y, acc.z, 0.0f);
    w.rp.rnds[li] = seed;
}

__global__ void __launch_bounds__(256) wf_generate(WfParams w) {
    __shared__ unsigned s_scratch[2 * 5];
    const int li = blockIdx.x * 256 + threadIdx.x;
    const RenderParams& p = w.rp;
    // rows >= 1 (bounces >= 1) are cleared here; row 0 (bounce 0, filled by THIS launch) is cleared by
    // a memset the host enqueues in front of the kernel
    if (li >= kWfCounterStride && li < (p.iterations + 3) * kWfCounterStride) w.counters[li] = 0u;
    int cost = -1;
    f3 P = mk(0.f, 0.f, 0.f), D = mk(0.f, 0.f, 1.f);
    if (li < w.npix) {
        const int lrow = li / p.width, x = li - lrow * p.width;
        const int grow = ((lrow / p.rows_per_block) * p.world + p.rank) * p.rows_per_block + (lrow % p.rows_per_block);
        const int gid = grow * p.width + x;
        int seed = p.rnds[li];
        const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
        camera_get_ray(gid, p.cam, rnd1, rnd2, &P, &D);
        if (p.iterations <= 0) {
            wf_finalize(w, li, mk(0.0f, 0.0f, 0.0f), seed);
        } else {
            w.sC[li] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            w.sD[li] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            w.sE[li] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            w.sF[li] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(seed));
            cost = ray_cost_class(w, P, D);
        }
    }
    if ((threadIdx.x & 63) == 0 && li < w.npix && p.stats) stat_add(p, 1, (unsigned long long)min(64, w.npix - li));
    const unsigned pos = block_reserve<2, 256>(cost, w.counters + kWfGenRow * kWfCounterStride, s_scratch);
    if (cost >= 0) {
        w.rsA[0][cost][pos] = make_float4(P.x, P.y, P.z, D.x);
        w.rsB[0][cost][pos] = make_float4(D.y, D.z, __int_as_float(li), 0.0f);
    }
}

// Rays per wave: each wave owns a contiguous range of the bounce's ray stream (no global atomics
// on the fetch side; blocks that finish early are replaced by the dispatcher).
constexpr int kWfRaysPerWave = 256;

// Flat traversal loop: every iteration each lane performs at most one node visit and then at most
// one triangle test, and a lane whose ray is finished takes the next ray of the wave's range in the
// SAME iteration (the next ray's 32 B are prefetched one assignment ahead, so the switch costs no
// memory round trip).  No lane ever waits for another lane's traversal to end.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) wf_intersect(WfParams w, int bounce) {
    constexpr int WAVES = BLOCK / 64;
    constexpr int RPB = WAVES * kWfRaysPerWave;          // rays per block
    constexpr int CHUNKS = RPB / 64;
    const RenderParams& p = w.rp;
    const int cost = blockIdx.y;
    unsigned* ctr = w.counters + wf_row(bounce) * kWfCounterStride;
    const unsigned n = ctr[cost];
    const unsigned block_base = blockIdx.x * RPB;
    if (block_base >= n) return;                          // uniform for the whole block
    // dynamic LDS: [traversal stacks][class byte per ray of the block][CHUNKS x 3 counts][3 bases]
    LaneStack<unsigned> stk;
    stk.base = reinterpret_cast<unsigned*>(pt_lds_raw) + threadIdx.x;
    stk.stride = BLOCK;
    unsigned char* lds_cls = pt_lds_raw + (size_t)p.stack_entries * 4 * BLOCK;
    unsigned* lds_cnt = reinterpret_cast<unsigned*>(lds_cls + RPB);     // [CHUNKS][3]
    unsigned* lds_base = lds_cnt + CHUNKS * 3;                          // [3]
    SceneView sv;
    sv.nodes = p.nodes;
    sv.tris = p.tris;
    sv.meta = p.meta;
    const float4* __restrict__ rsA = w.rsA[bounce & 1][cost];
    const float4* __restrict__ rsB = w.rsB[bounce & 1][cost];
    float2* __restrict__ hits = w.hit[cost];
    const unsigned long long lt = lanemask_lt();
    const unsigned wave = threadIdx.x >> 6;
    unsigned cbase = block_base + wave * kWfRaysPerWave;  // uniform per wave: next unassigned ray
    const unsigned cend = min(cbase + (unsigned)kWfRaysPerWave, n);
    if (cbase > cend) cbase = cend;
    Trav<unsigned> tr;
    tr.begin(mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f), stk);
    tr.idle();
    unsigned pos = ~0u;          // stream position of the ray in flight (~0u: none)
    unsigned npos = ~0u;         // prefetched next ray (~0u: none)
    float4 nA = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 nB = make_float2(0.f, 1.f);
    WorkCount wc;
    for (;;) {
        // ---- lanes whose ray is finished switch to their prefetched ray
        if (tr.done() && npos != ~0u) {
            pos = npos;
            npos = ~0u;
            tr.begin(mk(nA.x, nA.y, nA.z), mk(nA.w, nB.x, nB.y), stk);
        }
        // ---- lanes without a prefetched ray reserve the next positions of the wave's range
        const unsigned long long want = __ballot(npos == ~0u);
        if (want != 0 && cbase < cend) {
            const unsigned my = cbase + (unsigned)__popcll(want & lt);
            if (npos == ~0u && my < cend) {
                npos = my;
                nA = rsA[my];
                nB = *reinterpret_cast<const float2*>(&rsB[my]);
            }
            cbase = min(cbase + (unsigned)__popcll(want), cend);
        }
        if (__ballot(!tr.done() || npos != ~0u) == 0) break;
        // ---- one node visit, then one triangle test
        if (tr.is_node(tr.cur)) tr.template node_step<false>(sv, &wc);
        if (tr.is_leaf(tr.cur)) tr.template leaf_step<false>(sv, &wc);
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
    // ---- order-preserving compaction of the block's rays into the three class queues
    __syncthreads();
    const unsigned nblock = min((unsigned)RPB, n - block_base);
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
}

constexpr int kWfShadeBlock = 1024;

__global__ void __launch_bounds__(kWfShadeBlock) wf_shade(WfParams w, int bounce) {
    __shared__ unsigned s_scratch[2 * (kWfShadeBlock / 64 + 1)];
    const RenderParams& p = w.rp;
    const int cls = blockIdx.y;
    unsigned* ctr = w.counters + wf_row(bounce) * kWfCounterStride;
    const unsigned n = ctr[2 + cls];
    if (blockIdx.x * kWfShadeBlock >= n) return;          // whole block idle
    const unsigned i = blockIdx.x * kWfShadeBlock + threadIdx.x;
    int li = 0;
    int cost = -1;                                        // >= 0: the path continues with a ray of that cost class
    f3 rP = mk(0.f, 0.f, 0.f), rD = mk(0.f, 0.f, 1.f);
    if (i < n) {
        const unsigned e = (unsigned)w.q_cls[cls][i];
        const int c_in = (int)(e >> 31);
        const unsigned pos = e & 0x7fffffffu;
        const float4 A = w.rsA[bounce & 1][c_in][pos], B = w.rsB[bounce & 1][c_in][pos];
        li = __float_as_int(B.z);
        const float4 F = w.sF[li];
        f3 color = mk(F.x, F.y, F.z);
        const int sbits = __float_as_int(F.w);
        int seed = sbits & 0x7fffffff;
        bool inside = sbits < 0;
        if (cls == 2) {                                   // miss: black environment, prog.cl:367-376
            wf_finalize(w, li, color, seed);
        } else {
            const float2 h = w.hit[c_in][pos];
            const float4 C = w.sC[li], Dq = w.sD[li], E = w.sE[li];
            rP = mk(A.x, A.y, A.z);
            rD = mk(A.w, B.x, B.y);
            f3 fL = mk(C.x, C.y, C.z), fB = mk(C.w, Dq.x, Dq.y), fS = mk(Dq.z, Dq.w, E.x), fR = mk(E.y, E.z, E.w);
            shade_hit(rP, rD, fL, fB, fS, fR, color, seed, inside, p, p.tris, p.meta, __float_as_int(h.y), h.x);
            if (bounce + 1 >= p.iterations) {
                wf_finalize(w, li, color, seed);
            } else {
                w.sC[li] = make_float4(fL.x, fL.y, fL.z, fB.x);
                w.sD[li] = make_float4(fB.y, fB.z, fS.x, fS.y);
                w.sE[li] = make_float4(fS.z, fR.x, fR.y, fR.z);
                w.sF[li] = make_float4(color.x, color.y, color.z, __int_as_float(seed | (inside ? (int)0x80000000 : 0)));
                cost = ray_cost_class(w, rP, rD);
            }
        }
    }
    const unsigned npos = block_reserve<2, kWfShadeBlock>(cost, w.counters + wf_row(bounce + 1) * kWfCounterStride, s_scratch);
    if (cost >= 0) {
        w.rsA[(bounce + 1) & 1][cost][npos] = make_float4(rP.x, rP.y, rP.z, rD.x);
        w.rsB[(bounce + 1) & 1][cost][npos] = make_float4(rD.y, rD.z, __int_as_float(li), 0.0f);
    }
}

// closest hit of arbitrary rays (test entry point): one ray per lane, packed triangle index out
__global__ void __launch_bounds__(256) k_debug_closest_hit(RenderParams p, const pt_ray* rays, long long n, float* out_t, int* out_tri) {
    LaneStack<unsigned> stk;
    stk.base = reinterpret_cast<unsigned*>(pt_lds_raw) + threadIdx.x;
    stk.stride = 256;
    SceneView sv;
    sv.nodes = p.nodes;
    sv.tris = p.tris;
    sv.meta = p.meta;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* r = reinterpret_cast<const float4*>(&rays[i]);
    const float4 a = r[0], b = r[1];
    WorkCount wc;
    float t;
    const int ti = closest_hit<unsigned, false, false>(sv, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), stk, &t, &wc);
    out_t[i] = ti >= 0 ? t : -1.0f;
    out_tri[i] = ti;
}

// ---- tone mapping, prog.cl:247-269 (value of write_imagef at prog.cl:380)
PT_DEV float srgb1(float a) {
    if (a <= 0.00304f) return 12.92f * a;
    return fmaf_(1.055f, spec_pow(a, 0.4167f), -0.055f);
}
__global__ void __launch_bounds__(256) k_resolve_reinhard(const float4* colors, float4* out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = colors[i];
    const float L = fmaf_(0.0722f, c.z, fmaf_(0.7152f, c.y, 0.2126f * c.x));
    const float L2 = L / (1.0f + L);
    out[i] = make_float4(srgb1(c.x * L2 / L), srgb1(c.y * L2 / L), srgb1(c.z * L2 / L), 1.0f);
}

// filt_im, prog.cl:391-427: 3x3 median by mean grey + filmic tone map; the reference's
// out-of-range reads at the right/top border (prog.cl:397-401) are not reproduced: border
// pixels are left untouched.
PT_DEV float filmic1(float cin) {   // prog.cl:259-263
    float c = cin - 0.004f;
    c = c > 0.0f ? c : 0.0f;
    return (c * fmaf_(c, 6.2f, 0.5f)) / fmaf_(c, fmaf_(c, 6.2f, 1.7f), 0.06f);
}
__global__ void __launch_bounds__(256) k_filt_im(const float4* colors, float4* out, int W, int H) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x <= 0 || y <= 0 || x >= W - 1 || y >= H - 1) return;
    float4 arr[9];
    float grey[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float4 c = colors[(size_t)(y - 1 + i) * W + (x - 1 + j)];
            arr[i * 3 + j] = c;
            grey[i * 3 + j] = ((c.x + c.y) + c.z) / 3.0f;
        }
#pragma unroll
    for (int jn = 9; jn > 1; --jn) {
        int maxi = 0;
#pragma unroll
        for (int i = 1; i < jn; ++i)
            if (grey[i] > grey[maxi]) maxi = i;
#pragma unroll
        for (int i = 0; i < 9; ++i) {       // swap without dynamic register indexing
            if (i == maxi) {
                const float tg = grey[jn - 1];
                const float4 tv = arr[jn - 1];
                grey[jn - 1] = grey[i];
                arr[jn - 1] = arr[i];
                grey[i] = tg;
                arr[i] = tv;
            }
        }
    }
    out[(size_t)y * W + x] = make_float4(filmic1(arr[4].x), filmic1(arr[4].y), filmic1(arr[4].z), 1.0f);
}

// ---------------------------------------------------------------------------- launchers
static inline int n_waves(const RenderParams& p) {
    const int tiles_x = (p.width + 7) >> 3, tiles_y = (p.local_rows + 7) >> 3;
    return tiles_x * tiles_y;
}

int mega_max_lds_scene_bytes() { return 160 * 1024; }

static inline bool stack16_ok(const RenderParams& p) { return p.lds_scene && p.n_nodes <= 32767 && p.n_tris <= 4096; }

size_t mega_lds_bytes(const RenderParams& p, int block) {
    size_t b = (size_t)p.stack_entries * (stack16_ok(p) ? 2 : 4) * (size_t)block;
    b = (b + 15) & ~(size_t)15;
    if (p.lds_scene) b += (size_t)p.n_nodes * 64 + (p.lds_scene == 1 ? (size_t)p.n_tris * 48 : 0);
    return b;
}

hipError_t launch_gen_ray(const RenderParams& p, const LaunchConfig&, hipStream_t stream) {
    const int waves = n_waves(p);
    if (waves == 0) return hipSuccess;
    const int blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(k_gen_ray, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <bool SPLIT, bool COUNT>
static hipError_t launch_render_t(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream) {
    const int waves = n_waves(p);
    if (waves == 0) return hipSuccess;
    const int wpb = lc.block / 64;
    int blocks = (waves + wpb - 1) / wpb;
    if (p.tile_counter) blocks = std::min(blocks, lc.persistent_blocks);
    const size_t lds = lc.lds_bytes;
#define PT_LAUNCH(LDS, B, ST) PT_LAUNCH_W(LDS, B, ST, 1)
#define PT_LAUNCH_W(LDS, B, ST, MW)                                                                \
    do {                                                                                           \
        auto kern = k_render<SPLIT, LDS, B, ST, COUNT, MW>;                                        \
        if (lds > 64 * 1024) {                                                                     \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                         \
        }                                                                                          \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(B), lds, stream, p);                           \
        return hipGetLastError();                                                                  \
    } while (0)
    if (p.lds_scene == 2) {                 // nodes in LDS: two 512-thread workgroups per CU, 4 waves/SIMD
        if (!stack16_ok(p) || lc.block != 512) return hipErrorInvalidValue;
        PT_LAUNCH_W(2, 512, unsigned short, 4);
    }
    if (p.lds_scene && stack16_ok(p)) {
        switch (lc.block) {
        case 64: PT_LAUNCH(1, 64, unsigned short);
        case 128: PT_LAUNCH(1, 128, unsigned short);
        case 256: PT_LAUNCH(1, 256, unsigned short);
        case 512: PT_LAUNCH(1, 512, unsigned short);
        case 1024: PT_LAUNCH(1, 1024, unsigned short);
        }
    } else if (p.lds_scene) {
        switch (lc.block) {
        case 64: PT_LAUNCH(1, 64, unsigned);
        case 128: PT_LAUNCH(1, 128, unsigned);
        case 256: PT_LAUNCH(1, 256, unsigned);
        case 512: PT_LAUNCH(1, 512, unsigned);
        case 1024: PT_LAUNCH(1, 1024, unsigned);
        }
    } else {
        if (!SPLIT && !COUNT && lc.block == 256 && lc.traversal >= 2) {     // sliced traversal, slice = traversal - 1 rounds
            hipLaunchKernelGGL((k_render_sliced<256, 4>), dim3(blocks), dim3(256), lds, stream, p, lc.traversal - 1);
            return hipGetLastError();
        }
        if (!SPLIT && lc.block == 256 && lc.traversal == 1) {               // voting schedule
            auto kern = k_render<SPLIT, 0, 256, unsigned, COUNT, 4, true>;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, p);
            return hipGetLastError();
        }
        if (!SPLIT && !COUNT && lc.block == 256 && lc.min_waves > 1) {      // occupancy experiments
            switch (lc.min_waves) {
            case 4: PT_LAUNCH_W(0, 256, unsigned, 4);
            case 5: PT_LAUNCH_W(0, 256, unsigned, 5);
            case 6: PT_LAUNCH_W(0, 256, unsigned, 6);
            case 8: PT_LAUNCH_W(0, 256, unsigned, 8);
            }
        }
        switch (lc.block) {
        case 64: PT_LAUNCH(0, 64, unsigned);
        case 128: PT_LAUNCH(0, 128, unsigned);
        case 256: PT_LAUNCH(0, 256, unsigned);
        case 512: PT_LAUNCH(0, 512, unsigned);
        case 1024: PT_LAUNCH(0, 1024, unsigned);
        }
    }
#undef PT_LAUNCH
#undef PT_LAUNCH_W
    return hipErrorInvalidValue;
}

hipError_t launch_trace_ray(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream) { return launch_render_t<true, false>(p, lc, stream); }
hipError_t launch_render_mega(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream) {
    return lc.count_work ? launch_render_t<false, true>(p, lc, stream) : launch_render_t<false, false>(p, lc, stream);
}

hipError_t launch_wf_generate(const WfParams& w, hipStream_t stream) {
    const int need = std::max(w.npix, (w.rp.iterations + 3) * kWfCounterStride);
    hipLaunchKernelGGL(wf_generate, dim3((need + 255) / 256), dim3(256), 0, stream, w);
    return hipGetLastError();
}

hipError_t launch_wf_intersect(const WfParams& w, int bounce, hipStream_t stream) {
    constexpr int BLOCK = 256, RPB = (BLOCK / 64) * kWfRaysPerWave;
    const size_t lds = (size_t)w.rp.stack_entries * 4 * BLOCK + RPB + (RPB / 64) * 3 * 4 + 32;
    hipLaunchKernelGGL(wf_intersect<BLOCK>, dim3((w.npix + RPB - 1) / RPB, 2), dim3(BLOCK), lds, stream, w, bounce);
    return hipGetLastError();
}

hipError_t launch_wf_shade(const WfParams& w, int bounce, hipStream_t stream) {
    hipLaunchKernelGGL(wf_shade, dim3((w.npix + kWfShadeBlock - 1) / kWfShadeBlock, 3), dim3(kWfShadeBlock), 0, stream, w, bounce);
    return hipGetLastError();
}

// experiment: persistent traversal-only kernel with ALL BVH nodes staged in LDS (triangles stay
// global); 512-thread blocks, 16-bit stack entries, grid-stride over the rays
template <bool LDS_NODES>
__global__ void __launch_bounds__(512) k_debug_closest_hit_persist(RenderParams p, const pt_ray* rays, long long n, float* out_t, int* out_tri) {
    using StackT = typename std::conditional<LDS_NODES, unsigned short, unsigned>::type;   // 16-bit refs only with staged nodes
    LaneStack<StackT> stk;
    stk.base = reinterpret_cast<StackT*>(pt_lds_raw) + threadIdx.x;
    stk.stride = 512;
    SceneView sv;
    sv.nodes = p.nodes;
    sv.tris = p.tris;
    sv.meta = p.meta;
    if (LDS_NODES) {
        float4* lds_nodes = reinterpret_cast<float4*>(pt_lds_raw + (((size_t)p.stack_entries * 2 * 512 + 15) & ~(size_t)15));
        stage_nodes(p, lds_nodes);
        __syncthreads();
        sv.nodes = lds_nodes;
    }
    WorkCount wc;
    for (long long i = (long long)blockIdx.x * 512 + threadIdx.x; i < n; i += (long long)gridDim.x * 512) {
        const float4* r = reinterpret_cast<const float4*>(&rays[i]);
        const float4 a = r[0], b = r[1];
        float t;
        const int ti = closest_hit<StackT, false, false>(sv, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), stk, &t, &wc);
        out_t[i] = ti >= 0 ? t : -1.0f;
        out_tri[i] = ti;
    }
}

hipError_t launch_debug_closest_hit(const RenderParams& p, const pt_ray* rays, int64_t n, float* out_t, int32_t* out_tri, hipStream_t stream, size_t lds_pad) {
    if (n == 0) return hipSuccess;
    if (lds_pad == 1 || lds_pad == 2) {      // 1: persistent, nodes from global; 2: persistent, nodes in LDS
        const bool in_lds = lds_pad == 2;
        size_t lds = (((size_t)p.stack_entries * (in_lds ? 2 : 4) * 512 + 15) & ~(size_t)15) + (in_lds ? (size_t)p.n_nodes * 64 : 0);
        if (p.n_nodes > 32767 || p.n_tris > 4096 || lds > 80 * 1024) return hipErrorInvalidValue;
        auto kern = in_lds ? k_debug_closest_hit_persist<true> : k_debug_closest_hit_persist<false>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(512), dim3(512), lds, stream, p, rays, (long long)n, out_t, out_tri);
        return hipGetLastError();
    }
    const size_t lds = (size_t)p.stack_entries * 4 * 256 + lds_pad;    // lds_pad: occupancy experiments
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_debug_closest_hit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_debug_closest_hit, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, stream, p, rays, (long long)n, out_t, out_tri);
    return hipGetLastError();
}

hipError_t launch_resolve_reinhard(const float4* colors, float4* out, int64_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve_reinhard, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, colors, out, (long long)n);
    return hipGetLastError();
}

hipError_t launch_filt_im(const float4* colors, float4* out, int32_t W, int32_t H, hipStream_t stream) {
    hipLaunchKernelGGL(k_filt_im, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, stream, colors, out, W, H);
    return hipGetLastError();
}

}  // namespace ptamd
