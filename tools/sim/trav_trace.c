// CPU emulation of Trav<kNodesLds>::round (pt_device.hpp) on the product's own BVH, recording for every
// ray the phase structure of its while-while traversal: per round (node visits, triangle tests).
// Development tool for tools/sim/wave_sim.py (what would a different wave schedule execute?).
// build: gcc -O2 -shared -fPIC -o tools/sim/libtravtrace.so tools/sim/trav_trace.c -lm
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float q[3][4]; int32_t left, right, pad[2]; } Node64;
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

// rays: n x 8 floats (P.xyz pad D.xyz pad).  phases: n x max_rounds x 3 uint16 (node visits, tri tests, leaves per round),
// nrounds: n.  defer = the global path's postponed-leaf variant.
void trav_trace(const Node64* nodes, const Tri* tris, const float* rays, int64_t n, int max_rounds, int defer,
                uint16_t* phases, int32_t* nrounds, float* out_t) {
    for (int64_t i = 0; i < n; ++i) {
        const float* P = rays + 8*i; const float* D = rays + 8*i + 4;
        float inv[3] = {1.0f/D[0], 1.0f/D[1], 1.0f/D[2]};
        float best = INFINITY;
        int32_t stack[64]; int sp = 0; stack[0] = 0x7fffffff;
        int32_t cur = 0, pend = 0; int r = 0;
        uint16_t* ph = phases + (size_t)i * max_rounds * 3;
        memset(ph, 0, sizeof(uint16_t) * 3 * max_rounds);
        while (cur != 0x7fffffff && r < max_rounds) {
            int nn = 0, nt = 0, nl = 0;
            while (cur >= 0 && cur != 0x7fffffff) {
                const Node64* nd = &nodes[cur];
                ++nn;
                float ln = -INFINITY, lf = INFINITY, rn = -INFINITY, rf = INFINITY;
                for (int a = 0; a < 3; ++a) {
                    float l0 = (nd->q[a][0]-P[a])*inv[a], l1 = (nd->q[a][1]-P[a])*inv[a];
                    float r0 = (nd->q[a][2]-P[a])*inv[a], r1 = (nd->q[a][3]-P[a])*inv[a];
                    ln = fmaxf(ln, fminf(l0,l1)); lf = fminf(lf, fmaxf(l0,l1));
                    rn = fmaxf(rn, fminf(r0,r1)); rf = fminf(rf, fmaxf(r0,r1));
                }
                lf *= 1.0000005f; rf *= 1.0000005f;
                float lim = best * 1.0000005f;
                int hl = (lf >= ln) && (lf >= 0) && (ln <= lim), hr = (rf >= rn) && (rf >= 0) && (rn <= lim);
                int take_left = hl && (!hr || ln <= rn), both = hl && hr, none = !(hl || hr);
                int32_t other = take_left ? nd->right : nd->left, next = take_left ? nd->left : nd->right;
                int32_t top = stack[sp];
                stack[sp+1] = other;
                if (defer) {
                    int cap = !none && next < 0 && pend == 0;
                    if (cap) pend = next;
                    int usetop = none || (cap && !both);
                    cur = usetop ? top : (cap ? other : next);
                    sp += (both && !cap) ? 1 : (usetop ? -1 : 0);
                } else {
                    cur = none ? top : next;
                    sp += both ? 1 : (none ? -1 : 0);
                }
            }
            if (defer && pend != 0) {
                int v = ~pend, first = v >> 3, count = (v & 7) + 1;
                ++nl;
                for (int j = 0; j < count; ++j) { float t = tri_test(&tris[first+j], P, D); ++nt; if (t > 0 && t < best) best = t; }
                pend = 0;
            }
            while (cur < 0) {
                int v = ~cur, first = v >> 3, count = (v & 7) + 1;
                ++nl;
                for (int j = 0; j < count; ++j) { float t = tri_test(&tris[first+j], P, D); ++nt; if (t > 0 && t < best) best = t; }
                cur = stack[sp]; --sp;
            }
            ph[3*r] = (uint16_t)nn; ph[3*r+1] = (uint16_t)nt; ph[3*r+2] = (uint16_t)nl; ++r;
        }
        nrounds[i] = r;
        out_t[i] = best;
    }
}
