// CPU emulation of a 4-wide BVH traversal (collapsed from the product's BVH2 by tools/sim/bvh4_sim.py): per ray
// the number of wide-node visits, leaf visits and triangle tests.  Development tool.
// build: gcc -O2 -shared -fPIC -o tools/sim/libbvh4trace.so tools/sim/bvh4_trace.c -lm
#include <math.h>
#include <stdint.h>

typedef struct { float lo[4][3], hi[4][3]; int32_t ref[4]; int32_t n; int32_t pad[3]; } Node4;   // ref >= 0 node, < 0 leaf ~(first<<3|count-1)
typedef struct { float v[12]; } Tri;

static float tri_test(const Tri* T, const float P[3], const float D[3]) {
    const float* r1 = T->v; const float* r2 = T->v + 3; const float* r3 = T->v + 6; const float* N = T->v + 9;
    float num = (r1[0]-P[0])*N[0] + (r1[1]-P[1])*N[1] + (r1[2]-P[2])*N[2];
    float den = D[0]*N[0] + D[1]*N[1] + D[2]*N[2];
    float t = num / den;
    if (!(t > 0.0f)) return -1.0f;
    float p[3] = {P[0]+D[0]*t, P[1]+D[1]*t, P[2]+D[2]*t};
    const float* v[3] = {r1, r2, r3};
    for (int k = 0; k < 3; ++k) {
        const float* a = v[k]; const float* b = v[(k+1)%3];
        float e[3] = {b[0]-a[0], b[1]-a[1], b[2]-a[2]}, w[3] = {p[0]-a[0], p[1]-a[1], p[2]-a[2]};
        float c[3] = {e[1]*w[2]-e[2]*w[1], e[2]*w[0]-e[0]*w[2], e[0]*w[1]-e[1]*w[0]};
        if (!(c[0]*N[0]+c[1]*N[1]+c[2]*N[2] >= 0.0f)) return -1.0f;
    }
    return t;
}

void bvh4_trace(const Node4* nodes, const Tri* tris, int n_flat, const float* rays, int64_t n, int32_t* out /* n x 4: nodes, leaves, tris, max stack */) {
    for (int64_t i = 0; i < n; ++i) {
        const float* P = rays + 8*i; const float* D = rays + 8*i + 4;
        float inv[3] = {1.0f/D[0], 1.0f/D[1], 1.0f/D[2]};
        float best = INFINITY;
        for (int j = 0; j < n_flat; ++j) { float t = tri_test(&tris[j], P, D); if (t > 0 && t < best) best = t; }
        int32_t stack[128]; int sp = 0; int maxsp = 0;
        stack[sp++] = 0;
        int nn = 0, nl = 0, nt = 0;
        while (sp > 0) {
            int32_t cur = stack[--sp];
            if (cur < 0) {
                int v = ~cur, first = v >> 3, count = (v & 7) + 1;
                ++nl;
                for (int j = 0; j < count; ++j) { float t = tri_test(&tris[first+j], P, D); ++nt; if (t > 0 && t < best) best = t; }
                continue;
            }
            const Node4* nd = &nodes[cur];
            ++nn;
            float tn[4]; int32_t rf[4]; int h = 0;
            for (int c = 0; c < nd->n; ++c) {
                float a = -INFINITY, b = INFINITY;
                for (int k = 0; k < 3; ++k) {
                    float t0 = (nd->lo[c][k]-P[k])*inv[k], t1 = (nd->hi[c][k]-P[k])*inv[k];
                    a = fmaxf(a, fminf(t0,t1)); b = fminf(b, fmaxf(t0,t1));
                }
                b *= 1.0000005f;
                if (b >= a && b >= 0 && a <= best * 1.0000005f) { tn[h] = a; rf[h] = nd->ref[c]; ++h; }
            }
            // far to near onto the stack
            for (int x = 0; x < h; ++x) for (int y = x+1; y < h; ++y) if (tn[y] > tn[x]) { float tt = tn[x]; tn[x] = tn[y]; tn[y] = tt; int32_t r = rf[x]; rf[x] = rf[y]; rf[y] = r; }
            for (int x = 0; x < h; ++x) stack[sp++] = rf[x];
            if (sp > maxsp) maxsp = sp;
        }
        out[4*i] = nn; out[4*i+1] = nl; out[4*i+2] = nt; out[4*i+3] = maxsp;
    }
}
