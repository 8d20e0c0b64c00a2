/*
 * pt_oracle.c -- CPU ORACLE (test infrastructure only; see pt_oracle.h for the rules,
 * the "PARITY UNPINNED" statement and the arithmetic contract).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -mfma -fPIC -shared (oracle/Makefile).
 *   -ffp-contract=off : the ONLY fused operations are the explicit fmaf()/fma() calls.
 *   -mfma             : those calls become one vfmadd instruction (same result as the
 *                       software routine, only faster).
 */
#define _GNU_SOURCE
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vector helpers */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 ld(const orc_f3* p) { return V(p->x, p->y, p->z); }
static inline void st(orc_f3* p, v3 v) { p->x = v.x; p->y = v.y; p->z = v.z; p->w = 0.0f; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* u*s + w as one fma per component */
static inline v3 madd(v3 u, float s, v3 w) { return V(fmaf(u.x, s, w.x), fmaf(u.y, s, w.y), fmaf(u.z, s, w.z)); }
static inline float dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 normalize(v3 a) { float s = 1.0f / sqrtf(dot(a, a)); return scale(a, s); }
static inline float max0(float c) { return c > 0.0f ? c : 0.0f; }

/* ------------------------------------------------------------------ layout table */
int orc_layout(int which) {
    switch (which) {
    case 0: return (int)sizeof(orc_material);
    case 1: return (int)offsetof(orc_material, ks);
    case 2: return (int)offsetof(orc_material, emission);
    case 3: return (int)offsetof(orc_material, F0);
    case 4: return (int)offsetof(orc_material, n);
    case 5: return (int)offsetof(orc_material, shininess);
    case 6: return (int)offsetof(orc_material, type);
    case 7: return (int)sizeof(orc_ray);
    case 8: return (int)offsetof(orc_ray, D);
    case 9: return (int)sizeof(orc_triangle);
    case 10: return (int)offsetof(orc_triangle, N);
    case 11: return (int)offsetof(orc_triangle, mati);
    case 12: return (int)sizeof(orc_node);
    case 13: return (int)offsetof(orc_node, bbox);
    case 14: return (int)sizeof(orc_camera);
    case 15: return (int)offsetof(orc_camera, XM);
    case 16: return (int)offsetof(orc_camera, YM);
    case 17: return (int)sizeof(orc_hit);
    case 18: return (int)offsetof(orc_hit, P);
    case 19: return (int)offsetof(orc_hit, mati);
    case 20: return (int)offsetof(orc_hit, mat);
    case 21: return (int)sizeof(orc_bbox);
    }
    return -1;
}

/* ------------------------------------------------------------------ RNG */
/* main.cpp:45 + 522-527: RNDS[i] = minstd_rand0() in pixel order; minstd_rand0 is
 * x <- x*16807 mod (2^31-1), default seed 1, and returns the NEW state.            */
void orc_seed_sequence(int32_t* out, int64_t n) {
    uint64_t x = 1;
    for (int64_t i = 0; i < n; ++i) {
        x = (x * 16807ull) % 2147483647ull;
        out[i] = (int32_t)x;
    }
}

/* prog.cl:72-77.  2147483647.0f rounds to 2^31, so the result lies in (0, 1]. */
float orc_rand(int32_t* seed) {
    uint64_t n = (uint64_t)(int64_t)(*seed);
    n = (n * 48271ull) % 2147483647ull;
    *seed = (int32_t)n;
    return (float)n / 2147483648.0f;
}

/* ------------------------------------------------------------------ spec math */
/* sin and cos of a float angle in [0, 8]: quadrant reduction and Taylor polynomials in
 * double (truncation < 1e-13), rounded once to float.                                */
void orc_spec_sincosf(float theta, float* s, float* c) {
    const double t = (double)theta;
    const int q = (int)fma(t, 0.63661977236758138, 0.5);
    const double qd = (double)q;
    double r = fma(qd, -1.5707963267948966, t);
    r = fma(qd, -6.123233995736766e-17, r);
    const double z = r * r;
    double ps = 1.6059043836821613e-10;           /*  1/13! */
    ps = fma(ps, z, -2.505210838544172e-08);      /* -1/11! */
    ps = fma(ps, z, 2.7557319223985893e-06);      /*  1/9!  */
    ps = fma(ps, z, -0.0001984126984126984);      /* -1/7!  */
    ps = fma(ps, z, 0.008333333333333333);        /*  1/5!  */
    ps = fma(ps, z, -0.16666666666666666);        /* -1/3!  */
    const double sr = fma(r * z, ps, r);
    double pc = -1.1470745597729725e-11;          /* -1/14! */
    pc = fma(pc, z, 2.08767569878681e-09);        /*  1/12! */
    pc = fma(pc, z, -2.755731922398589e-07);      /* -1/10! */
    pc = fma(pc, z, 2.48015873015873e-05);        /*  1/8!  */
    pc = fma(pc, z, -0.001388888888888889);       /* -1/6!  */
    pc = fma(pc, z, 0.041666666666666664);        /*  1/4!  */
    pc = fma(pc, z, -0.5);
    const double cr = fma(z, pc, 1.0);
    double sv, cv;
    switch (q & 3) {
    case 0: sv = sr; cv = cr; break;
    case 1: sv = cr; cv = -sr; break;
    case 2: sv = -sr; cv = -cr; break;
    default: sv = -cr; cv = sr; break;
    }
    *s = (float)sv;
    *c = (float)cv;
}

/* pow(x,5) of prog.cl:221 */
float orc_spec_pow5(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x4 * x; }

/* General pow for x >= 0: exp2(y*log2 x) in double, rounded once to float.  Results
 * below 2^-126 are flushed to +0 and above 2^128 go to +inf.                        */
float orc_spec_powf(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return NAN;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return y > 0.0f ? 0.0f : INFINITY;
    if (isinf(x)) return y > 0.0f ? INFINITY : 0.0f;
    double xd = (double)x;                 /* exact; float denormals are double normals */
    uint64_t bits; memcpy(&bits, &xd, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m; memcpy(&m, &bits, 8);        /* m in [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double sdiv = f / (2.0 + f);     /* ln m = 2 atanh(s) */
    const double z = sdiv * sdiv;
    double p = 0.10526315789473684;        /* 2/19 */
    p = fma(p, z, 0.11764705882352941);    /* 2/17 */
    p = fma(p, z, 0.13333333333333333);    /* 2/15 */
    p = fma(p, z, 0.15384615384615385);    /* 2/13 */
    p = fma(p, z, 0.18181818181818182);    /* 2/11 */
    p = fma(p, z, 0.22222222222222221);    /* 2/9  */
    p = fma(p, z, 0.2857142857142857);     /* 2/7  */
    p = fma(p, z, 0.4);                    /* 2/5  */
    p = fma(p, z, 0.66666666666666663);    /* 2/3  */
    p = fma(p, z, 2.0);
    const double lnm = sdiv * p;
    const double lg2 = fma(lnm, 1.4426950408889634, (double)e);
    const double w = (double)y * lg2;
    if (!(w > -126.0)) return 0.0f;
    if (w >= 128.0) return INFINITY;
    const double nd = floor(w + 0.5);
    const double g = (w - nd) * 0.6931471805599453;   /* |g| <= 0.3466 */
    double q = 2.08767569878681e-09;       /* 1/12! */
    q = fma(q, g, 2.505210838544172e-08);  /* 1/11! */
    q = fma(q, g, 2.755731922398589e-07);  /* 1/10! */
    q = fma(q, g, 2.7557319223985893e-06); /* 1/9!  */
    q = fma(q, g, 2.48015873015873e-05);   /* 1/8!  */
    q = fma(q, g, 0.0001984126984126984);  /* 1/7!  */
    q = fma(q, g, 0.001388888888888889);   /* 1/6!  */
    q = fma(q, g, 0.008333333333333333);   /* 1/5!  */
    q = fma(q, g, 0.041666666666666664);   /* 1/4!  */
    q = fma(q, g, 0.16666666666666666);    /* 1/3!  */
    q = fma(q, g, 0.5);
    q = fma(q, g, 1.0);
    q = fma(q, g, 1.0);
    uint64_t sb = (uint64_t)((int64_t)nd + 1023) << 52;
    double sc; memcpy(&sc, &sb, 8);
    return (float)(q * sc);
}

/* ------------------------------------------------------------------ host constructors */
/* main.cpp:101-111.  n = mean of N; F0_i = (k^2 + (n-1)^2) / (k^2 + (n+1)^2). */
void orc_material_make(orc_material* m, const float kd[3], const float ks[3], const float em[3],
                       const float N[3], const float K[3], float shininess, int type) {
    memset(m, 0, sizeof *m);
    m->kd.x = kd[0]; m->kd.y = kd[1]; m->kd.z = kd[2];
    m->ks.x = ks[0]; m->ks.y = ks[1]; m->ks.z = ks[2];
    m->emission.x = em[0]; m->emission.y = em[1]; m->emission.z = em[2];
    m->shininess = shininess;
    m->type = type;
    m->n = (N[0] + N[1] + N[2]) / 3.0f;
    float F0[3];
    for (int i = 0; i < 3; ++i) {
        float a = (N[i] - 1) * (N[i] - 1);
        float b = (N[i] + 1) * (N[i] + 1);
        F0[i] = (K[i] * K[i] + a) / (K[i] * K[i] + b);
    }
    m->F0.x = F0[0]; m->F0.y = F0[1]; m->F0.z = F0[2];
}

/* main.cpp:144-166: unit geometric normal = cross(r2-r1, r3-r1)/length, length via the
 * double sqrt narrowed to float (== correctly rounded float sqrt).                   */
void orc_triangle_make(orc_triangle* t, const float r1[3], const float r2[3], const float r3[3], int mati) {
    memset(t, 0, sizeof *t);
    t->r1.x = r1[0]; t->r1.y = r1[1]; t->r1.z = r1[2];
    t->r2.x = r2[0]; t->r2.y = r2[1]; t->r2.z = r2[2];
    t->r3.x = r3[0]; t->r3.y = r3[1]; t->r3.z = r3[2];
    t->mati = (uint16_t)mati;
    float v1[3], v2[3], n[3];
    for (int i = 0; i < 3; ++i) { v1[i] = r2[i] - r1[i]; v2[i] = r3[i] - r1[i]; }
    n[0] = v1[1] * v2[2] - v1[2] * v2[1];
    n[1] = v1[2] * v2[0] - v1[0] * v2[2];
    n[2] = v1[0] * v2[1] - v1[1] * v2[0];
    float length = (float)sqrt((double)(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]));
    for (int i = 0; i < 3; ++i) n[i] = n[i] / length;
    t->N.x = n[0]; t->N.y = n[1]; t->N.z = n[2];
}

/* main.cpp:47-70: rotations; the trig runs in double and each component is narrowed. */
static void rot_y(float v[3], float beta) {
    beta = beta / 180.0f * 3.141593f;
    double c = cos((double)beta), s = sin((double)beta);
    float r0 = (float)((double)v[0] * c + (double)v[2] * s);
    float r1 = v[1];
    float r2 = (float)(-(double)v[0] * s + (double)v[2] * c);
    v[0] = r0; v[1] = r1; v[2] = r2;
}
static void rot_x(float v[3], float gamma) {
    gamma = gamma / 180.0f * 3.141593f;
    double c = cos((double)gamma), s = sin((double)gamma);
    float r0 = v[0];
    float r1 = (float)((double)v[1] * c - (double)v[2] * s);
    float r2 = (float)((double)v[1] * s + (double)v[2] * c);
    v[0] = r0; v[1] = r1; v[2] = r2;
}

/* main.cpp:311-347 with the globals as parameters and no movement keys held
 * (global_forward/rightward/upward = 0, so global_shift is used unchanged). */
void orc_camera_make(orc_camera* c, float fov, float yaw, float pitch, const float shift[3], int width, int height) {
    memset(c, 0, sizeof *c);
    c->XM = (float)width;
    c->YM = (float)height;
    float up_length = c->YM / 2.0f;
    float right_length = c->XM / 2.0f;
    float ahead_length = (float)((double)right_length / tan((double)(fov / 2.0f / 180.0f * 3.141593f)));
    float up[3] = { 0.0f, 1.0f, 0.0f }, right[3] = { 1.0f, 0.0f, 0.0f }, ahead[3] = { 0.0f, 0.0f, 1.0f };
    rot_x(up, pitch); rot_y(up, yaw);
    rot_x(right, pitch); rot_y(right, yaw);
    rot_x(ahead, pitch); rot_y(ahead, yaw);
    for (int i = 0; i < 3; ++i) { up[i] = up[i] * up_length; right[i] = right[i] * right_length; ahead[i] = ahead[i] * ahead_length; }
    c->eye.x = 500.0f + shift[0]; c->eye.y = 500.0f + shift[1]; c->eye.z = -1299.037842f + shift[2];
    c->up.x = up[0]; c->up.y = up[1]; c->up.z = up[2];
    c->right.x = right[0]; c->right.y = right[1]; c->right.z = right[2];
    c->lookat.x = c->eye.x + ahead[0]; c->lookat.y = c->eye.y + ahead[1]; c->lookat.z = c->eye.z + ahead[2];
}

/* Camera()'s side effect, main.cpp:323-336: the unit axes rotated by pitch then yaw, and
 *   global_shift.s[i] = global_shift.s[i] + ahead.s[i]*global_forward + right.s[i]*global_rightward + up.s[i]*global_upward;
 * (float, left to right; x86-64 g++ emits no fma) -- applied to `shift` in place. */
void orc_camera_move(float shift[3], float yaw, float pitch, float forward, float rightward, float upward) {
    float up[3] = { 0.0f, 1.0f, 0.0f }, right[3] = { 1.0f, 0.0f, 0.0f }, ahead[3] = { 0.0f, 0.0f, 1.0f };
    rot_x(up, pitch); rot_y(up, yaw);
    rot_x(right, pitch); rot_y(right, yaw);
    rot_x(ahead, pitch); rot_y(ahead, yaw);
    for (int i = 0; i < 3; ++i) {
        volatile float t = ahead[i] * forward;          /* volatile: each product and sum rounded to float on its own */
        volatile float acc = shift[i] + t;
        t = right[i] * rightward; acc = acc + t;
        t = up[i] * upward; acc = acc + t;
        shift[i] = acc;
    }
}

/* Scene::add_Obj's per-vertex transform, main.cpp:598-606: negate x, rotate_x(pitch),
 * rotate_y(yaw), then v*scale+pos (float multiply then add: x86-64 g++, no fma). */
void orc_obj_vertex(float out[3], const float v[3], const float pos[3], const float scale[3], float pitch, float yaw) {
    float t[3] = { -v[0], v[1], v[2] };
    rot_x(t, pitch);
    rot_y(t, yaw);
    for (int i = 0; i < 3; ++i) out[i] = t[i] * scale[i] + pos[i];
}

/* ------------------------------------------------------------------ Scene + kd tree */
typedef struct hnode {            /* NodeOnHost, main.cpp:195-209 */
    struct hnode *left, *right;
    orc_bbox box;
    int leaf;
    int ntri;
    orc_triangle* tris;           /* leaf payload (copies, as the reference keeps) */
    int32_t* origs;               /* add-order index of each payload triangle */
    int from, to;                 /* global [from,to) assigned by convert */
} hnode;

struct orc_scene {
    orc_triangle* tris; int ntris, captris;       /* main.cpp:366 */
    int32_t* orig;                                /* add-order index of tris[i] (follows the reordering) */
    int tri_shift;                                /* main.cpp:368 */
    int32_t* shifts; int nshift, capshift;        /* kd_tree_shift, main.cpp:369 */
    orc_material* mats; int nmats, capmats;       /* main.cpp:370 */
    orc_node* nodes; int64_t nnodes, capnodes;    /* kd_tree, main.cpp:371 */
    int heap_ok;                                  /* 0 once a heap index would exceed ORC_HEAP_LIMIT */
    hnode** roots; int nroots;                    /* pointer trees, one per end_Obj */
    int32_t* rank; int nrank;                     /* encounter rank per add-order index */
    int next_rank;
};

#define ORC_HEAP_LIMIT (1 << 24)

orc_scene* orc_scene_create(void) {
    orc_scene* s = calloc(1, sizeof *s);
    s->heap_ok = 1;
    return s;
}

static void free_tree(hnode* n) {
    if (!n) return;
    free_tree(n->left); free_tree(n->right);
    free(n->tris); free(n->origs); free(n);
}

void orc_scene_destroy(orc_scene* s) {
    if (!s) return;
    for (int i = 0; i < s->nroots; ++i) free_tree(s->roots[i]);
    free(s->roots); free(s->tris); free(s->orig); free(s->shifts); free(s->mats); free(s->nodes); free(s->rank);
    free(s);
}

int orc_add_material(orc_scene* s, const orc_material* m) {           /* main.cpp:532-535 */
    if (s->nmats == s->capmats) { s->capmats = s->capmats ? 2 * s->capmats : 16; s->mats = realloc(s->mats, sizeof(orc_material) * s->capmats); }
    s->mats[s->nmats++] = *m;
    return s->nmats - 1;
}

void orc_add_triangle(orc_scene* s, const orc_triangle* t) {          /* main.cpp:529-531 */
    if (s->ntris == s->captris) {
        s->captris = s->captris ? 2 * s->captris : 64;
        s->tris = realloc(s->tris, sizeof(orc_triangle) * s->captris);
        s->orig = realloc(s->orig, sizeof(int32_t) * s->captris);
    }
    s->orig[s->ntris] = s->ntris;
    s->tris[s->ntris++] = *t;
}

void orc_add_triangles(orc_scene* s, const float* v, const uint16_t* mati, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        orc_triangle t;
        orc_triangle_make(&t, v + 9 * i, v + 9 * i + 3, v + 9 * i + 6, mati[i]);
        orc_add_triangle(s, &t);
    }
}

/* Triangle::bbox / midpoint / BBox::expand, main.cpp:131-136, 167-181 */
static orc_bbox tri_bbox(const orc_triangle* t) {
    orc_bbox b; memset(&b, 0, sizeof b);
    const float* a = &t->r1.x; const float* bb = &t->r2.x; const float* c = &t->r3.x;
    float* lo = &b.bl.x; float* hi = &b.tr.x;
    for (int i = 0; i < 3; ++i) {
        float m1 = bb[i] < a[i] ? bb[i] : a[i];          /* std::min(std::min(r1,r2),r3) */
        lo[i] = c[i] < m1 ? c[i] : m1;
        float M1 = a[i] < bb[i] ? bb[i] : a[i];          /* std::max(std::max(r1,r2),r3) */
        hi[i] = M1 < c[i] ? c[i] : M1;
    }
    return b;
}
static void bbox_expand(orc_bbox* b, const orc_bbox* o) {
    float* lo = &b->bl.x; float* hi = &b->tr.x; const float* olo = &o->bl.x; const float* ohi = &o->tr.x;
    for (int i = 0; i < 3; ++i) { if (olo[i] < lo[i]) lo[i] = olo[i]; if (ohi[i] > hi[i]) hi[i] = ohi[i]; }
}
static void tri_midpoint(const orc_triangle* t, float mp[3]) {
    const float* a = &t->r1.x; const float* b = &t->r2.x; const float* c = &t->r3.x;
    for (int i = 0; i < 3; ++i) mp[i] = (a[i] + b[i] + c[i]) / 3.0f;
}

typedef struct { orc_triangle t; int32_t orig; } tagged;

/* NodeOnHost::build, main.cpp:210-262.  *err is raised where the reference would spin
 * forever (more than 6 triangles that no axis separates: main.cpp:246-257).          */
static hnode* build_rec(const tagged* tr, int n, int depth, int* err) {
    hnode* node = calloc(1, sizeof *node);
    if (n <= 6) {                                             /* main.cpp:212-221 */
        node->leaf = 1; node->ntri = n;
        node->tris = malloc(sizeof(orc_triangle) * (size_t)(n ? n : 1));
        node->origs = malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
        for (int i = 0; i < n; ++i) { node->tris[i] = tr[i].t; node->origs[i] = tr[i].orig; }
        node->box = tri_bbox(&tr[0].t);
        for (int i = 1; i < n; ++i) { orc_bbox b = tri_bbox(&tr[i].t); bbox_expand(&node->box, &b); }
        return node;
    }
    node->box = tri_bbox(&tr[0].t);                           /* main.cpp:223-234 */
    float midpoint[3]; tri_midpoint(&tr[0].t, midpoint);
    for (int i = 1; i < n; ++i) {
        orc_bbox b = tri_bbox(&tr[i].t); bbox_expand(&node->box, &b);
        float mp[3]; tri_midpoint(&tr[i].t, mp);
        for (int j = 0; j < 3; ++j) midpoint[j] = midpoint[j] + mp[j];
    }
    for (int i = 0; i < 3; ++i) midpoint[i] = midpoint[i] / (float)(unsigned long)n;
    int axis = depth % 3;                                     /* main.cpp:236-257 */
    tagged* L = malloc(sizeof(tagged) * (size_t)n); tagged* R = malloc(sizeof(tagged) * (size_t)n);
    int nl = 0, nr = 0, tries = 0;
    for (;;) {
        nl = nr = 0;
        for (int i = 0; i < n; ++i) {
            float mp[3]; tri_midpoint(&tr[i].t, mp);
            if (midpoint[axis] >= mp[axis]) R[nr++] = tr[i]; else L[nl++] = tr[i];
        }
        if (nl != 0 && nr != 0) break;
        if (++tries >= 3) { *err = 1; break; }
        axis = (axis + 1) % 3;
    }
    if (*err) { free(L); free(R); node->leaf = 1; node->ntri = 0; return node; }
    node->left = build_rec(L, nl, depth + 1, err);            /* main.cpp:259-260 */
    node->right = build_rec(R, nr, depth + 1, err);
    free(L); free(R);
    return node;
}

static void heap_put(orc_scene* s, int64_t idx, int from, int to, const orc_bbox* box) {
    /* main.cpp:282-284 / 290-292: pad with filler nodes carrying the current box */
    while (s->nnodes <= idx) {
        if (s->nnodes == s->capnodes) { s->capnodes = s->capnodes ? 2 * s->capnodes : 256; s->nodes = realloc(s->nodes, sizeof(orc_node) * (size_t)s->capnodes); }
        orc_node f; memset(&f, 0, sizeof f); f.trii[0] = -1; f.trii[1] = -1; f.bbox = *box;
        s->nodes[s->nnodes++] = f;
    }
    orc_node nd; memset(&nd, 0, sizeof nd); nd.trii[0] = from; nd.trii[1] = to; nd.bbox = *box;
    s->nodes[idx] = nd;
}

/* depth-first, left-first: the order in which prog.cl:159-181 can reach the leaves */
static void rank_dfs(orc_scene* s, const hnode* n) {
    if (n->leaf) { for (int i = 0; i < n->ntri; ++i) s->rank[n->origs[i]] = s->next_rank++; return; }
    rank_dfs(s, n->left); rank_dfs(s, n->right);
}

/* Scene::end_Obj, main.cpp:536-551 (+ NodeOnHost::convert, main.cpp:263-303) */
int orc_end_obj(orc_scene* s) {
    int nobj = s->ntris - s->tri_shift;
    if (nobj <= 0) return -2;                       /* reference: tris[0] of an empty vector */
    int32_t shift = 0;
    if (s->nshift != 0) shift = (int32_t)(s->nnodes - 1);           /* main.cpp:537-540 */
    if (s->nshift == s->capshift) { s->capshift = s->capshift ? 2 * s->capshift : 8; s->shifts = realloc(s->shifts, sizeof(int32_t) * (size_t)s->capshift); }
    s->shifts[s->nshift++] = shift;

    tagged* in = malloc(sizeof(tagged) * (size_t)nobj);
    for (int i = 0; i < nobj; ++i) { in[i].t = s->tris[s->tri_shift + i]; in[i].orig = s->orig[s->tri_shift + i]; }
    int err = 0;
    hnode* root = build_rec(in, nobj, 0, &err);
    free(in);
    s->roots = realloc(s->roots, sizeof(hnode*) * (size_t)(s->nroots + 1));
    s->roots[s->nroots++] = root;
    if (err) return -1;

    /* convert: breadth-first, heap index ptr (children 2p, 2p+1), triangles re-emitted in
     * the order the leaves are dequeued (main.cpp:274-302, 548-549)                    */
    typedef struct { hnode* node; int64_t ptr; } qe;
    int64_t qcap = 1024, qh = 0, qt = 0;
    qe* q = malloc(sizeof(qe) * (size_t)qcap);
    q[qt].node = root; q[qt].ptr = 1; ++qt;
    int from = s->tri_shift, to = s->tri_shift, w = s->tri_shift;
    orc_triangle* neworder = malloc(sizeof(orc_triangle) * (size_t)nobj);
    int32_t* neworig = malloc(sizeof(int32_t) * (size_t)nobj);
    int nw = 0;
    while (qh < qt) {
        qe cur = q[qh++];
        if (cur.ptr >= ORC_HEAP_LIMIT) s->heap_ok = 0;
        if (cur.node->leaf) {
            to = to + cur.node->ntri;
            if (s->heap_ok) heap_put(s, cur.ptr + shift, from, to, &cur.node->box);
            cur.node->from = from; cur.node->to = to;
            for (int i = 0; i < cur.node->ntri; ++i) { neworder[nw] = cur.node->tris[i]; neworig[nw] = cur.node->origs[i]; ++nw; }
            from = from + cur.node->ntri;
        } else {
            if (s->heap_ok) heap_put(s, cur.ptr + shift, -1, -1, &cur.node->box);
            cur.node->from = cur.node->to = -1;
        }
        if (qt + 2 > qcap) { qcap *= 2; q = realloc(q, sizeof(qe) * (size_t)qcap); }
        if (cur.node->left) { q[qt].node = cur.node->left; q[qt].ptr = cur.ptr * 2; ++qt; }
        if (cur.node->right) { q[qt].node = cur.node->right; q[qt].ptr = cur.ptr * 2 + 1; ++qt; }
    }
    free(q);
    for (int i = 0; i < nobj; ++i) { s->tris[w + i] = neworder[i]; s->orig[w + i] = neworig[i]; }
    free(neworder); free(neworig);

    s->rank = realloc(s->rank, sizeof(int32_t) * (size_t)s->ntris);
    s->nrank = s->ntris;
    rank_dfs(s, root);
    s->tri_shift = s->ntris;
    if (!s->heap_ok) { free(s->nodes); s->nodes = NULL; s->nnodes = 0; s->capnodes = 0; }
    return 0;
}

int orc_scene_counts(const orc_scene* s, int* ntris, int* nnodes, int* nobj, int* nmats) {
    if (ntris) *ntris = s->ntris;
    if (nnodes) *nnodes = s->heap_ok ? (int)s->nnodes : -1;
    if (nobj) *nobj = s->nshift;
    if (nmats) *nmats = s->nmats;
    return 0;
}
const orc_triangle* orc_scene_tris(const orc_scene* s) { return s->tris; }
const orc_node* orc_scene_nodes(const orc_scene* s) { return s->heap_ok ? s->nodes : NULL; }
const int32_t* orc_scene_shifts(const orc_scene* s) { return s->shifts; }
const orc_material* orc_scene_mats(const orc_scene* s) { return s->mats; }
void orc_scene_encounter_rank(const orc_scene* s, int32_t* out) { memcpy(out, s->rank, sizeof(int32_t) * (size_t)s->nrank); }

/* ------------------------------------------------------------------ device functions */
typedef struct { float t; v3 P, N; int mati; } hit_t;           /* Hit without the material copy */
typedef struct { v3 P, D; } ray_t;

static inline hit_t init_hit(void) { hit_t h; h.t = -1.0f; h.P = V(0, 0, 0); h.N = V(0, 0, 0); h.mati = 0; return h; }  /* prog.cl:68-70 */

/* prog.cl:82-92 */
static inline ray_t camera_get_ray(int id, const orc_camera* cam, float rnd1, float rnd2) {
    int X = (int)cam->XM;
    int Y = (int)cam->YM;
    float x = (float)(id % X) + rnd1;
    float y = (float)(id / X) + rnd2;
    v3 right = scale(ld(&cam->right), (2.0f * x) / (float)X - 1.0f);
    v3 up = scale(ld(&cam->up), (2.0f * y) / (float)Y - 1.0f);
    v3 p = add(add(ld(&cam->lookat), right), up);
    ray_t r; r.P = ld(&cam->eye); r.D = normalize(sub(p, ld(&cam->eye)));
    return r;
}

/* prog.cl:94-112 */
static inline hit_t triangle_intersect(const orc_triangle* tri, const ray_t* ray) {
    hit_t hit = init_hit();
    v3 P = ray->P, Vd = ray->D, N = ld(&tri->N);
    v3 r1 = ld(&tri->r1), r2 = ld(&tri->r2), r3 = ld(&tri->r3);
    float t = dot(sub(r1, P), N) / dot(Vd, N);
    if (t < 0) return hit;
    v3 p = madd(Vd, t, P);
    if (dot(cross(sub(r2, r1), sub(p, r1)), N) >= 0)
        if (dot(cross(sub(r3, r2), sub(p, r2)), N) >= 0)
            if (dot(cross(sub(r1, r3), sub(p, r3)), N) >= 0) {
                hit.t = t; hit.P = p; hit.N = N; hit.mati = tri->mati;
            }
    return hit;
}

/* prog.cl:113-122 */
static inline hit_t first_intersect(const orc_triangle* tris, int from, int to, const ray_t* ray) {
    hit_t best = init_hit();
    for (int i = from; i < to; ++i) {
        hit_t h = triangle_intersect(&tris[i], ray);
        if (h.t > 0 && (best.t < 0 || h.t < best.t)) best = h;
    }
    return best;
}

/* prog.cl:123-143; fminf/fmaxf ignore a NaN operand like OpenCL fmin/fmax */
static inline int bbox_intersection(const orc_bbox* box, const ray_t* ray, float* tmin, float* tmax) {
    float tx1 = (box->bl.x - ray->P.x) / ray->D.x;
    float tx2 = (box->tr.x - ray->P.x) / ray->D.x;
    *tmin = fminf(tx1, tx2);
    *tmax = fmaxf(tx1, tx2);
    float ty1 = (box->bl.y - ray->P.y) / ray->D.y;
    float ty2 = (box->tr.y - ray->P.y) / ray->D.y;
    *tmin = fmaxf(*tmin, fminf(ty1, ty2));
    *tmax = fminf(*tmax, fmaxf(ty1, ty2));
    float tz1 = (box->bl.z - ray->P.z) / ray->D.z;
    float tz2 = (box->tr.z - ray->P.z) / ray->D.z;
    *tmin = fmaxf(*tmin, fminf(tz1, tz2));
    *tmax = fminf(*tmax, fmaxf(tz1, tz2));
    return *tmax >= *tmin;
}

/* prog.cl:271-290 */
static inline void stack_push(int* stack, int* ptr, int val) { if (*ptr < 300) { stack[*ptr] = val; *ptr = *ptr + 1; } }
static inline int stack_pop(int* stack, int* ptr) { if (*ptr > 0) { *ptr = *ptr - 1; return stack[*ptr]; } return stack[0]; }
static inline void stack_check(int* stack, int* sp, int* empty, int* ptr) { if (*sp == 0) *empty = 1; else *ptr = stack_pop(stack, sp); }

/* prog.cl:144-184, on the heap-indexed array */
static hit_t kd_intersect_heap(const orc_scene* s, const ray_t* ray) {
    const orc_node* kd = s->nodes; const orc_triangle* tris = s->tris;
    hit_t hit = init_hit(), best = init_hit();
    for (int i = 0; i < s->nshift; ++i) {
        const int sh = s->shifts[i];
        int ptr = 1 + sh;
        float tmin = 999999; float dist = 0; float tmax = -999999;
        int stack[300]; int sp = 0; int empty = 0;
        stack[0] = 0;
        while (!empty) {
            if (bbox_intersection(&kd[ptr].bbox, ray, &dist, &tmax)) {
                if (tmax >= 0) {
                    if (dist > tmin) {
                        stack_check(stack, &sp, &empty, &ptr);
                    } else if (kd[ptr].trii[0] < 0) {
                        stack_push(stack, &sp, 2 * (ptr - sh) + 1 + sh);
                        ptr = 2 * (ptr - sh) + sh;
                    } else {
                        hit = first_intersect(tris, kd[ptr].trii[0], kd[ptr].trii[1], ray);
                        if (hit.t > 0 && (best.t < 0 || hit.t < best.t)) { tmin = hit.t; best = hit; }
                        stack_check(stack, &sp, &empty, &ptr);
                    }
                } else stack_check(stack, &sp, &empty, &ptr);
            } else stack_check(stack, &sp, &empty, &ptr);
        }
    }
    return best;
}

/* the same traversal on the pointer tree (used when the heap array is infeasible, and to
 * cross-check the heap array).  The 300-entry push limit of prog.cl:272 is kept.      */
static hit_t kd_intersect_ptr(const orc_scene* s, const ray_t* ray) {
    const orc_triangle* tris = s->tris;
    hit_t hit = init_hit(), best = init_hit();
    for (int i = 0; i < s->nroots; ++i) {
        const hnode* ptr = s->roots[i];
        float tmin = 999999; float dist = 0; float tmax = -999999;
        const hnode* stack[300]; int sp = 0; int empty = 0;
        while (!empty) {
            int pop = 1;
            if (bbox_intersection(&ptr->box, ray, &dist, &tmax) && tmax >= 0) {
                if (dist > tmin) {
                } else if (!ptr->leaf) {
                    if (sp < 300) stack[sp++] = ptr->right;
                    ptr = ptr->left; pop = 0;
                } else {
                    hit = first_intersect(tris, ptr->from, ptr->to, ray);
                    if (hit.t > 0 && (best.t < 0 || hit.t < best.t)) { tmin = hit.t; best = hit; }
                }
            }
            if (pop) { if (sp == 0) empty = 1; else ptr = stack[--sp]; }
        }
    }
    return best;
}

/* exhaustive closest hit; equal t -> the triangle the reference traversal meets first */
static hit_t kd_intersect_brute(const orc_scene* s, const ray_t* ray) {
    hit_t best = init_hit(); int best_rank = 0x7fffffff;
    for (int i = 0; i < s->ntris; ++i) {
        hit_t h = triangle_intersect(&s->tris[i], ray);
        if (!(h.t > 0)) continue;
        int rk = s->rank[s->orig[i]];
        if (best.t < 0 || h.t < best.t || (h.t == best.t && rk < best_rank)) { best = h; best_rank = rk; }
    }
    return best;
}

static inline hit_t closest_hit(const orc_scene* s, const ray_t* ray, int mode) {
    if (mode == 2) return kd_intersect_brute(s, ray);
    if (mode == 0 && s->heap_ok && s->nodes) return kd_intersect_heap(s, ray);
    return kd_intersect_ptr(s, ray);
}

/* prog.cl:186-204 (V1 = N in, V2/V3 out) */
static inline void orthonormal_base(v3 v1, v3* V2, v3* V3) {
    const float E = 0.001f;
    v3 v2;
    if (fabsf(v1.x) <= E && fabsf(v1.z) <= E) {
        float rlength = 1.0f / sqrtf(fmaf(v1.z, v1.z, v1.y * v1.y));
        v2.x = 0; v2.y = -v1.z * rlength; v2.z = v1.y * rlength;
    } else {
        float rlength = 1.0f / sqrtf(fmaf(v1.z, v1.z, v1.x * v1.x));
        v2.x = -v1.z * rlength; v2.y = 0; v2.z = v1.x * rlength;
    }
    *V2 = v2; *V3 = cross(v1, v2);
}

/* prog.cl:205-218.  theta = 2*M_PI*rnd2 is a double product narrowed to float. */
static inline ray_t new_ray_diffuse(v3 P, v3 N, float rnd1, float rnd2) {
    const float E = 0.001f;
    v3 X, Y = N, Z;
    orthonormal_base(Y, &Z, &X);
    float r = sqrtf(rnd1);
    float theta = (float)(6.283185307179586 * (double)rnd2);
    float sn, cs; orc_spec_sincosf(theta, &sn, &cs);
    float x = r * cs;
    float y = r * sn;
    float z = sqrtf(1.0f - rnd1);
    v3 d = scale(X, x); d = madd(Y, z, d); d = madd(Z, y, d);
    ray_t o; o.P = madd(Y, E, P); o.D = normalize(d);
    return o;
}

/* prog.cl:219-222 */
static inline v3 fresnel(v3 F0, v3 N, v3 D) {
    float cosa = fabsf(dot(N, D));
    float p5 = orc_spec_pow5(1.0f - cosa);
    return V(fmaf(1.0f - F0.x, p5, F0.x), fmaf(1.0f - F0.y, p5, F0.y), fmaf(1.0f - F0.z, p5, F0.z));
}

/* prog.cl:223-227 */
static inline ray_t new_ray_specular(v3 P, v3 N, const ray_t* old) {
    float cosa = dot(N, old->D);
    v3 nd = normalize(sub(old->D, scale(scale(N, cosa), 2.0f)));
    ray_t o; o.P = madd(N, 0.001f, P); o.D = nd;
    return o;
}

/* prog.cl:228-245 */
static inline ray_t new_ray_refractive(v3 P, v3 N, v3 F0, float n, const ray_t* old, int* in, float rnd) {
    if (*in) n = 1.0f / n;
    float cosa = dot(neg(old->D), N);
    float disc = 1.0f - (fmaf(-cosa, cosa, 1.0f) / n) / n;
    v3 F = fresnel(F0, N, old->D);
    float prob = ((F.x + F.y) + F.z) / 3.0f;
    if (disc > 0 && rnd > prob) {
        *in = !*in;
        ray_t o;
        o.P = madd(N, -0.001f, P);
        v3 dn = V(old->D.x / n, old->D.y / n, old->D.z / n);
        o.D = normalize(madd(N, cosa / n - sqrtf(disc), dn));
        return o;
    }
    return new_ray_specular(P, N, old);
}

/* prog.cl:247-269 */
static inline void srgb(float a[3]) {
    for (int i = 0; i < 3; ++i) {
        if (a[i] <= 0.00304f) a[i] = 12.92f * a[i];
        else a[i] = fmaf(1.055f, orc_spec_powf(a[i], 0.4167f), -0.055f);
    }
}
void orc_reinhard_tone_map(float out[4], const float c[3]) {
    float L = fmaf(0.0722f, c[2], fmaf(0.7152f, c[1], 0.2126f * c[0]));
    float L2 = L / (1.0f + L);
    float a[3] = { c[0] * L2 / L, c[1] * L2 / L, c[2] * L2 / L };   /* L == 0 -> NaN, as the reference */
    srgb(a);
    out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = 1.0f;
}
void orc_filmic_tone_map(float out[4], const float cin[3]) {
    for (int i = 0; i < 3; ++i) {
        float c = cin[i] - 0.004f; c = c > 0.0f ? c : 0.0f;
        out[i] = (c * fmaf(c, 6.2f, 0.5f)) / fmaf(c, fmaf(c, 6.2f, 1.7f), 0.06f);
    }
    out[3] = 1.0f;
}

/* ---- exported unit-level wrappers */
static inline ray_t ray_in(const orc_ray* r) { ray_t o; o.P = ld(&r->P); o.D = ld(&r->D); return o; }
static inline void ray_out(orc_ray* o, const ray_t* r) { st(&o->P, r->P); st(&o->D, r->D); }
static void hit_out(orc_hit* o, const hit_t* h) {
    memset(o, 0, sizeof *o);
    o->t = h->t; st(&o->P, h->P); st(&o->N, h->N); o->mati = (uint16_t)h->mati;
}
void orc_camera_get_ray(orc_ray* out, int id, const orc_camera* cam, float rnd1, float rnd2) { ray_t r = camera_get_ray(id, cam, rnd1, rnd2); ray_out(out, &r); }
void orc_triangle_intersect(orc_hit* out, const orc_triangle* tri, const orc_ray* ray) { ray_t r = ray_in(ray); hit_t h = triangle_intersect(tri, &r); hit_out(out, &h); }
int orc_bbox_intersection(const orc_bbox* box, const orc_ray* ray, float* tmin, float* tmax) { ray_t r = ray_in(ray); return bbox_intersection(box, &r, tmin, tmax); }
void orc_kd_intersect(orc_hit* out, const orc_scene* s, const orc_ray* ray, int mode) {
    ray_t r = ray_in(ray); hit_t h = closest_hit(s, &r, mode); hit_out(out, &h);
    if (h.t > 0) out->mat = s->mats[h.mati];
}
void orc_new_ray_diffuse(orc_ray* out, const orc_f3* P, const orc_f3* N, float rnd1, float rnd2) { ray_t r = new_ray_diffuse(ld(P), ld(N), rnd1, rnd2); ray_out(out, &r); }
void orc_new_ray_specular(orc_ray* out, const orc_f3* P, const orc_f3* N, const orc_ray* old) { ray_t o = ray_in(old); ray_t r = new_ray_specular(ld(P), ld(N), &o); ray_out(out, &r); }
void orc_new_ray_refractive(orc_ray* out, const orc_f3* P, const orc_f3* N, const orc_f3* F0, float n, const orc_ray* old, int* in, float rnd) {
    ray_t o = ray_in(old); ray_t r = new_ray_refractive(ld(P), ld(N), ld(F0), n, &o, in, rnd); ray_out(out, &r);
}
void orc_fresnel(orc_f3* out, const orc_f3* F0, const orc_f3* N, const orc_f3* D) { st(out, fresnel(ld(F0), ld(N), ld(D))); }

/* ------------------------------------------------------------------ frame + kernels */
struct orc_frame {
    int W, H;
    orc_ray* rays; int32_t* rnds; orc_f3* colors; float* tex;      /* main.cpp:508-520 */
};

orc_frame* orc_frame_create(int W, int H) {
    orc_frame* f = calloc(1, sizeof *f);
    size_t n = (size_t)W * (size_t)H;
    f->W = W; f->H = H;
    f->rays = calloc(n, sizeof(orc_ray)); f->rnds = calloc(n, sizeof(int32_t));
    f->colors = calloc(n, sizeof(orc_f3)); f->tex = calloc(n * 4, sizeof(float));
    return f;
}
void orc_frame_destroy(orc_frame* f) { if (!f) return; free(f->rays); free(f->rnds); free(f->colors); free(f->tex); free(f); }
void orc_frame_seed_default(orc_frame* f) { orc_seed_sequence(f->rnds, (int64_t)f->W * f->H); }
int32_t* orc_frame_rnds(orc_frame* f) { return f->rnds; }
orc_ray* orc_frame_rays(orc_frame* f) { return f->rays; }
orc_f3* orc_frame_colors(orc_frame* f) { return f->colors; }
float* orc_frame_tex(orc_frame* f) { return f->tex; }

/* prog.cl:384-389 for one work-item.  rand(), rand() are evaluated left to right. */
static inline void gen_ray_item(orc_frame* f, const orc_camera* cam, int id) {
    float rnd1 = orc_rand(&f->rnds[id]);
    float rnd2 = orc_rand(&f->rnds[id]);
    ray_t r = camera_get_ray(id, cam, rnd1, rnd2);
    ray_out(&f->rays[id], &r);
}

/* prog.cl:292-381 for one work-item; returns the number of kd_intersect calls */
static inline int trace_ray_item(orc_frame* f, const orc_scene* s, const orc_camera* cam, int iterations, int current_sample, int mode, int id) {
    v3 factor_L = V(1, 1, 1), factor_B = V(1, 1, 1), factor_S = V(1, 1, 1), factor_R = V(1, 1, 1);
    v3 color = V(0, 0, 0);
    if (current_sample == 0) st(&f->colors[id], color);
    int in = 0;
    int segs = 0;
    ray_t ray = ray_in(&f->rays[id]);
    int32_t* seed = &f->rnds[id];
    for (int current = 0; current < iterations; ++current) {
        hit_t hit = closest_hit(s, &ray, mode);
        ++segs;
        if (hit.t > 0) {
            const orc_material* mat = &s->mats[hit.mati];
            if (iterations == 1) color = add(ld(&mat->kd), ld(&mat->emission));
            v3 N = hit.N;
            if (dot(ray.D, N) > 0) N = neg(N);
            if (mat->type == 0) {                                   /* prog.cl:329-340 */
                float rnd1 = orc_rand(seed); float rnd2 = orc_rand(seed);
                ray_t nr = new_ray_diffuse(hit.P, N, rnd1, rnd2);
                float intensity_diffuse = max0(dot(nr.D, N));
                factor_L = mul(factor_L, scale(ld(&mat->kd), intensity_diffuse));
                v3 view = normalize(sub(ld(&cam->eye), hit.P));
                v3 halfway = normalize(add(view, nr.D));
                float intensity_specular = max0(dot(N, halfway));
                factor_B = mul(factor_B, scale(ld(&mat->ks), orc_spec_powf(intensity_specular, mat->shininess)));
                ray = nr;
            }
            if (mat->type == 1) {                                   /* prog.cl:341-345 */
                ray_t old = ray;
                ray = new_ray_specular(hit.P, N, &old);
                factor_S = mul(factor_S, fresnel(ld(&mat->F0), N, old.D));
            }
            if (mat->type == 2) {                                   /* prog.cl:346-357 */
                ray_t old = ray;
                int before = in;
                float rnd = orc_rand(seed);
                ray = new_ray_refractive(hit.P, N, ld(&mat->F0), mat->n, &old, &in, rnd);
                v3 F = fresnel(ld(&mat->F0), N, old.D);
                float prob = ((F.x + F.y) + F.z) / 3.0f;
                if (before != in) {
                    float k = 1.0f / (1.0f - prob);
                    factor_R = scale(mul(factor_R, V(1.0f - F.x, 1.0f - F.y, 1.0f - F.z)), k);
                } else {
                    float k = 1.0f / prob;
                    factor_R = scale(mul(factor_R, F), k);
                }
            }
            if (mat->type == 3) {                                   /* prog.cl:358-366 */
                float intensity = max0(dot(neg(ray.D), N));
                float rnd1 = orc_rand(seed); float rnd2 = orc_rand(seed);
                ray_t nr = new_ray_diffuse(hit.P, N, rnd1, rnd2);
                v3 e = mul(mul(mul(ld(&mat->emission), add(factor_L, factor_B)), factor_S), factor_R);
                color = madd(e, intensity, color);
                ray = nr;
            }
        } else {
            break;                                                  /* prog.cl:367-376: black environment */
        }
    }
    ray_out(&f->rays[id], &ray);
    /* prog.cl:379 */
    v3 acc = ld(&f->colors[id]);
    float cs = (float)current_sample, cs1 = (float)(current_sample + 1);
    acc = V(fmaf(acc.x, cs, color.x) / cs1, fmaf(acc.y, cs, color.y) / cs1, fmaf(acc.z, cs, color.z) / cs1);
    st(&f->colors[id], acc);
    /* prog.cl:380 */
    float c3[3] = { acc.x, acc.y, acc.z };
    orc_reinhard_tone_map(&f->tex[(size_t)id * 4], c3);
    return segs;
}

/* ---- thread pool over rows: work-items are independent (each owns rays/rnds/colors[id]) */
typedef struct {
    orc_frame* f; const orc_scene* s; const orc_camera* cam;
    int iterations, first_sample, nsamples, mode, what;   /* what: 0 gen, 1 trace, 2 render, 3 filt */
    atomic_int next_row; atomic_llong segs;
    float* scratch;
} job_t;

static void filt_row(orc_frame* f, float* outtex, int y);

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    long long segs = 0;
    for (;;) {
        int y = atomic_fetch_add(&j->next_row, 1);
        if (y >= j->f->H) break;
        if (j->what == 3) { filt_row(j->f, j->scratch, y); continue; }
        for (int x = 0; x < j->f->W; ++x) {
            int id = y * j->f->W + x;
            if (j->what == 0) gen_ray_item(j->f, j->cam, id);
            else if (j->what == 1) segs += trace_ray_item(j->f, j->s, j->cam, j->iterations, j->first_sample, j->mode, id);
            else for (int k = 0; k < j->nsamples; ++k) {       /* Scene::render, main.cpp:683-687 */
                gen_ray_item(j->f, j->cam, id);
                segs += trace_ray_item(j->f, j->s, j->cam, j->iterations, j->first_sample + k, j->mode, id);
            }
        }
    }
    atomic_fetch_add(&j->segs, segs);
    return NULL;
}

static int64_t run_job(job_t* j, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    atomic_init(&j->next_row, 0); atomic_init(&j->segs, 0);
    pthread_t th[256];
    for (int i = 1; i < nthreads; ++i) pthread_create(&th[i], NULL, worker, j);
    worker(j);
    for (int i = 1; i < nthreads; ++i) pthread_join(th[i], NULL);
    return (int64_t)atomic_load(&j->segs);
}

void orc_gen_ray(orc_frame* f, const orc_camera* cam, int nthreads) {
    job_t j; memset(&j, 0, sizeof j); j.f = f; j.cam = cam; j.what = 0; run_job(&j, nthreads);
}
void orc_trace_ray(orc_frame* f, const orc_scene* s, const orc_camera* cam, int iterations, int current_sample, int mode, int nthreads) {
    job_t j; memset(&j, 0, sizeof j); j.f = f; j.s = s; j.cam = cam; j.iterations = iterations; j.first_sample = current_sample; j.mode = mode; j.what = 1;
    run_job(&j, nthreads);
}
/* Per pixel the samples are sequential and pixels are independent, so running all the
 * samples of one pixel back to back equals nsamples full-frame launch pairs.          */
int64_t orc_render(orc_frame* f, const orc_scene* s, const orc_camera* cam, int iterations, int first_sample, int nsamples, int mode, int nthreads) {
    job_t j; memset(&j, 0, sizeof j); j.f = f; j.s = s; j.cam = cam; j.iterations = iterations; j.first_sample = first_sample; j.nsamples = nsamples; j.mode = mode; j.what = 2;
    return run_job(&j, nthreads);
}

/* prog.cl:391-427: 3x3 median by mean-grey, filmic tone map.  The reference's guard
 * (x>0 && y>0 && x<width && y<height) lets x=width-1 / y=height-1 read outside the row /
 * buffer; here those border pixels are left untouched.                              */
static void filt_row(orc_frame* f, float* outtex, int y) {
    int W = f->W, H = f->H;
    if (y <= 0 || y >= H - 1) return;
    for (int x = 1; x < W - 1; ++x) {
        v3 arr[9]; float grey[9];
        for (int i = 0; i < 3; ++i) for (int jx = 0; jx < 3; ++jx) {
            int id = (y - 1 + i) * W + (x - 1 + jx);
            arr[i * 3 + jx] = ld(&f->colors[id]);
        }
        for (int i = 0; i < 9; ++i) grey[i] = ((arr[i].x + arr[i].y) + arr[i].z) / 3.0f;
        for (int jn = 9; jn > 1; jn--) {
            int maxi = 0;
            for (int i = 1; i < jn; ++i) if (grey[i] > grey[maxi]) maxi = i;
            float tg = grey[jn - 1]; grey[jn - 1] = grey[maxi]; grey[maxi] = tg;
            v3 tv = arr[jn - 1]; arr[jn - 1] = arr[maxi]; arr[maxi] = tv;
        }
        float med[3] = { arr[4].x, arr[4].y, arr[4].z };
        orc_filmic_tone_map(&outtex[((size_t)y * W + x) * 4], med);
    }
}
void orc_filt_im(orc_frame* f, int nthreads) {
    job_t j; memset(&j, 0, sizeof j); j.f = f; j.what = 3; j.scratch = f->tex; run_job(&j, nthreads);
}
