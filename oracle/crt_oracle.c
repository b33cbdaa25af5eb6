/*
 * crt_oracle.c — CPU oracle (TEST INFRASTRUCTURE, PARITY UNPINNED; see crt_oracle.h).
 *
 * Plain-C restatement of the reference's list-render path.  Structure mirrors the
 * reference on purpose (per-object Transform, rotate() with its sin/cos, the virtual
 * hit() wrapper applied at the world list AND at every triangle, double-typed sites),
 * so that every line can be checked against the file:line it cites.  Paths are relative
 * to /root/reference/CudaTest/src.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include "crt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ core/vec3.h */

static o_vec3 v3(float a, float b, float c) { o_vec3 r; r.e[0] = a; r.e[1] = b; r.e[2] = c; return r; } /* vec3.h:13-15 */
static o_vec3 v1(float a) { return v3(a, a, a); }                                                      /* vec3.h:10-12 */

static o_vec3 add(o_vec3 a, o_vec3 b) { return v3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); } /* vec3.h:73-75 */
static o_vec3 sub(o_vec3 a, o_vec3 b) { return v3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); } /* vec3.h:77-79 */
static o_vec3 mulv(o_vec3 a, o_vec3 b) { return v3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); } /* vec3.h:89-91 */
static o_vec3 muls(float t, o_vec3 v) { return v3(t * v.e[0], t * v.e[1], t * v.e[2]); }               /* vec3.h:93-99 (both orders compute t*v.e[i]) */
static o_vec3 divv(o_vec3 a, o_vec3 b) { return v3(a.e[0] / b.e[0], a.e[1] / b.e[1], a.e[2] / b.e[2]); } /* vec3.h:101-103 */
static o_vec3 divs(o_vec3 v, float t) { return v3(v.e[0] / t, v.e[1] / t, v.e[2] / t); }               /* vec3.h:105-107 */

static float dot(o_vec3 a, o_vec3 b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }   /* vec3.h:109-111 */
static o_vec3 cross(o_vec3 a, o_vec3 b) {                                                              /* vec3.h:113-117 */
    return v3((a.e[1] * b.e[2] - a.e[2] * b.e[1]),
              (-(a.e[0] * b.e[2] - a.e[2] * b.e[0])),
              (a.e[0] * b.e[1] - a.e[1] * b.e[0]));
}
static float length(o_vec3 v) { return sqrtf(v.e[0] * v.e[0] + v.e[1] * v.e[1] + v.e[2] * v.e[2]); }   /* vec3.h:38-40 (sqrt(float) -> float overload) */
static float squared_length(o_vec3 v) { return v.e[0] * v.e[0] + v.e[1] * v.e[1] + v.e[2] * v.e[2]; }  /* vec3.h:42-44 */
static float sum3(o_vec3 v) { return v.e[0] + v.e[1] + v.e[2]; }                                       /* vec3.h:68-71 */
static o_vec3 unit_vector(o_vec3 v) { return divs(v, length(v)); }                                     /* vec3.h:191-193 */
static o_vec3 lerp3(float t, o_vec3 from, o_vec3 to) { return add(from, muls(t, sub(to, from))); }      /* vec3.h:196-198 */

static float clip_single(float f, int min, int max) {                                                  /* vec3.h:119-123 */
    if (f > max) return max;
    else if (f < min) return min;
    return f;
}
static o_vec3 clip(o_vec3 v) {                                                                         /* vec3.h:125-131, min=0 max=1 */
    o_vec3 vr = v3(0, 0, 0);
    vr.e[0] = clip_single(v.e[0], 0, 1);
    vr.e[1] = clip_single(v.e[1], 0, 1);
    vr.e[2] = clip_single(v.e[2], 0, 1);
    return vr;
}
static o_vec3 div_assign_s(o_vec3 v, float t) {                                                        /* vec3.h:182-189 */
    float k = 1.0 / t; /* double site: (float)(1.0 / (double)t) */
    return v3(v.e[0] * k, v.e[1] * k, v.e[2] * k);
}

static o_vec3 rotate(o_vec3 origin, o_vec3 rotation) {                                                 /* vec3.h:200-217 */
    float radiansX = (M_PI / 180.) * rotation.e[0];
    float sin_X = sinf(radiansX);
    float cos_X = cosf(radiansX);
    float radiansY = (M_PI / 180.) * rotation.e[1];
    float sin_Y = sinf(radiansY);
    float cos_Y = cosf(radiansY);
    float radiansZ = -(M_PI / 180.) * rotation.e[2];
    float sin_Z = sinf(radiansZ);
    float cos_Z = cosf(radiansZ);

    o_vec3 rotate0 = v3(cos_Y * cos_Z, -cos_Y * sin_Z, sin_Y);
    o_vec3 rotate1 = v3(sin_X * sin_Y * cos_Z + cos_X * sin_Z, -sin_X * sin_Y * sin_Z + cos_X * cos_Z, -sin_X * cos_Y);
    o_vec3 rotate2 = v3(-cos_X * sin_Y * cos_Z + sin_X * sin_Z, cos_X * sin_Y * sin_Z + sin_X * cos_Z, cos_X * cos_Y);

    return v3(sum3(mulv(origin, rotate0)), sum3(mulv(origin, rotate1)), sum3(mulv(origin, rotate2)));
}

/* The vec3 restatements above, one operation at a time, for tests/test_vec3_ref.py: held bit for bit to the reference's
 * own core/vec3.h compiled for the host (oracle/ref_vec3.cpp -> oracle/_ref/libref_vec3.so).  Same op codes, same
 * layout: a, b, out are n x 3 floats, s is n floats; scalar results go to out[3i].  Returns 1 for an operation the
 * oracle does not restate (nothing on the path uses it). */
int o_vec3_op(int op, long long n, const float *a, const float *b, const float *s, float *out) {
    for (long long i = 0; i < n; i++) {
        o_vec3 A = v3(a[3 * i], a[3 * i + 1], a[3 * i + 2]), B = v3(b[3 * i], b[3 * i + 1], b[3 * i + 2]), R = v3(0, 0, 0);
        float t = s[i];
        switch (op) {
            case 0: R = add(A, B); break;
            case 1: R = sub(A, B); break;
            case 2: R = mulv(A, B); break;
            case 3: R = muls(t, A); break;
            case 4: R = divv(A, B); break;
            case 5: R = divs(A, t); break;
            case 6: R = v3(dot(A, B), 0, 0); break;
            case 7: R = cross(A, B); break;
            case 8: R = v3(length(A), 0, 0); break;
            case 9: R = v3(squared_length(A), 0, 0); break;
            case 10: R = unit_vector(A); break;
            case 11: R = lerp3(t, A, B); break;
            case 12: R = clip(A); break;
            case 13: R = div_assign_s(A, t); break;
            case 14: R = rotate(A, B); break;
            case 17: R = v3(sum3(A), 0, 0); break;
            default: return 1;
        }
        out[3 * i] = R.e[0]; out[3 * i + 1] = R.e[1]; out[3 * i + 2] = R.e[2];
    }
    return 0;
}

/* ------------------------------------------------------------------ core/ray.h */

typedef struct { float _time; o_vec3 _origin; o_vec3 _direction; } o_ray;                              /* ray.h:20-22 */
static o_ray make_ray(o_vec3 o, o_vec3 d, float t) { o_ray r; r._time = t; r._origin = o; r._direction = d; return r; } /* ray.h:11 */
static o_vec3 point_at_t(const o_ray *r, float t) { return add(r->_origin, muls(t, r->_direction)); }  /* ray.h:18 */

/* ------------------------------------------------------------------ cuRAND XORWOW
 * Third-party arithmetic that is NOT in /root/reference: cuRAND device API of CUDA
 * Toolkit 12.2 (CudaTest/CudaTest.vcxproj:33), curand_kernel.h.  Restated from the
 * published algorithm: Marsaglia's xorwow recurrence + Weyl sequence, cuRAND's seed
 * scrambling constants, and the 2^67-step subsequence skip expressed as powers of the
 * GF(2) transition matrix of the xorshift part.  The recurrence and both skips ARE held to an
 * independent XORWOW — rocRAND's xorwow_engine on raw states (oracle/ref_rocrand_xorwow.cpp,
 * tests/test_ref_xorwow.py).  The scrambling constants, the start words and curand_uniform's mapping
 * are cuRAND's alone and no known-answer vector exists offline: PARITY UNPINNED for those. */

unsigned int o_curand(o_curand_state *s) {                                /* curand_kernel.h: curand(curandStateXORWOW_t*) */
    unsigned int t;
    t = (s->v[0] ^ (s->v[0] >> 2));
    s->v[0] = s->v[1];
    s->v[1] = s->v[2];
    s->v[2] = s->v[3];
    s->v[3] = s->v[4];
    s->v[4] = (s->v[4] ^ (s->v[4] << 4)) ^ (t ^ (t << 1));
    s->d += 362437;
    return s->v[4] + s->d;
}

float o_curand_uniform(o_curand_state *s) {                               /* curand_kernel.h: _curand_uniform */
    unsigned int x = o_curand(s);
    const float CURAND_2POW32_INV = 2.3283064e-10f;
    return x * CURAND_2POW32_INV + (CURAND_2POW32_INV / 2.0f);
}

/* 160x160 GF(2) matrix, row-vector convention (cuRAND's __curand_matvec): row b is the
 * image of state bit b (word b/32, bit b%32). */
#define XW_N 5
#define XW_BITS (XW_N * 32)
typedef struct { unsigned int row[XW_BITS][XW_N]; } xw_matrix;

static void xw_step_words(unsigned int v[XW_N]) {      /* the xorshift part of curand() without the Weyl sequence */
    unsigned int t = (v[0] ^ (v[0] >> 2));
    v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

static void xw_matvec(const unsigned int vec[XW_N], const xw_matrix *m, unsigned int out[XW_N]) { /* __curand_matvec */
    unsigned int r[XW_N] = {0, 0, 0, 0, 0};
    for (int i = 0; i < XW_N; i++)
        for (int j = 0; j < 32; j++)
            if (vec[i] & (1u << j))
                for (int k = 0; k < XW_N; k++) r[k] ^= m->row[i * 32 + j][k];
    for (int k = 0; k < XW_N; k++) out[k] = r[k];
}

static void xw_matmat(xw_matrix *dst, const xw_matrix *a, const xw_matrix *b) { /* dst = a*b : apply a, then b */
    xw_matrix *tmp = (xw_matrix *)malloc(sizeof(xw_matrix));
    for (int r = 0; r < XW_BITS; r++) xw_matvec(a->row[r], b, tmp->row[r]);
    memcpy(dst, tmp, sizeof(xw_matrix));
    free(tmp);
}

static void xw_one_step_matrix(xw_matrix *m) {
    for (int b = 0; b < XW_BITS; b++) {
        unsigned int v[XW_N] = {0, 0, 0, 0, 0};
        v[b / 32] = 1u << (b % 32);
        xw_step_words(v);
        for (int k = 0; k < XW_N; k++) m->row[b][k] = v[k];
    }
}

/* Byte-sliced application table for one matrix (pure speed-up of xw_matvec). */
typedef struct { unsigned int t[XW_N * 4][256][XW_N]; } xw_table;

static void xw_build_table(const xw_matrix *m, xw_table *tb) {
    for (int byte = 0; byte < XW_N * 4; byte++)
        for (int val = 0; val < 256; val++) {
            unsigned int r[XW_N] = {0, 0, 0, 0, 0};
            for (int bit = 0; bit < 8; bit++)
                if (val & (1 << bit))
                    for (int k = 0; k < XW_N; k++) r[k] ^= m->row[byte * 8 + bit][k];
            for (int k = 0; k < XW_N; k++) tb->t[byte][val][k] = r[k];
        }
}

static void xw_apply_table(unsigned int v[XW_N], const xw_table *tb) {
    unsigned int r[XW_N] = {0, 0, 0, 0, 0};
    for (int w = 0; w < XW_N; w++)
        for (int b = 0; b < 4; b++) {
            const unsigned int *e = tb->t[w * 4 + b][(v[w] >> (8 * b)) & 0xff];
            for (int k = 0; k < XW_N; k++) r[k] ^= e[k];
        }
    for (int k = 0; k < XW_N; k++) v[k] = r[k];
}

/* cuRAND walks the skip count two bits at a time (PRECALC_BLOCK_SIZE 2): digit i of the
 * count applies M_i = base^(4^i) that many (0..3) times.  seq tables: base = T^(2^67)
 * (_skipahead_sequence_scratch); offset tables: base = T (_skipahead_scratch). */
#define XW_DIGITS 32
static xw_table *g_seq_tab[XW_DIGITS];
static xw_table *g_off_tab[XW_DIGITS];
static xw_matrix *g_seq_mat[XW_DIGITS];
static xw_matrix *g_off_mat[XW_DIGITS];

static void xw_ensure(int digit, int seq) {
    xw_matrix **mat = seq ? g_seq_mat : g_off_mat;
    xw_table **tab = seq ? g_seq_tab : g_off_tab;
    if (tab[digit]) return;
    if (!mat[0]) {
        mat[0] = (xw_matrix *)malloc(sizeof(xw_matrix));
        xw_one_step_matrix(mat[0]);
        if (seq)
            for (int i = 0; i < 67; i++) xw_matmat(mat[0], mat[0], mat[0]); /* T^(2^67) */
    }
    for (int i = 1; i <= digit; i++) {
        if (mat[i]) continue;
        mat[i] = (xw_matrix *)malloc(sizeof(xw_matrix));
        xw_matmat(mat[i], mat[i - 1], mat[i - 1]); /* ^2 */
        xw_matmat(mat[i], mat[i], mat[i]);         /* ^4 */
    }
    for (int i = 0; i <= digit; i++) {
        if (tab[i]) continue;
        tab[i] = (xw_table *)malloc(sizeof(xw_table));
        xw_build_table(mat[i], tab[i]);
    }
}

static void xw_skip(unsigned int v[XW_N], unsigned long long count, int seq) {
    unsigned long long p = count;
    int digit = 0;
    while (p) {
        unsigned int reps = (unsigned int)(p & 3);
        if (reps) {
            xw_ensure(digit, seq);
            for (unsigned int t = 0; t < reps; t++) xw_apply_table(v, (seq ? g_seq_tab : g_off_tab)[digit]);
        }
        p >>= 2;
        digit++;
    }
}

void o_curand_init(unsigned long long seed, unsigned long long subsequence,
                   unsigned long long offset, o_curand_state *state) {     /* curand_kernel.h: _curand_init_scratch */
    unsigned int s0 = ((unsigned int)seed) ^ 0xaad26b49U;
    unsigned int s1 = (unsigned int)(seed >> 32) ^ 0xf7dcefddU;
    unsigned int t0 = 1099087573U * s0;
    unsigned int t1 = 2591861531U * s1;
    state->d = 6615241 + t1 + t0;
    state->v[0] = 123456789U + t0;
    state->v[1] = 362436069U ^ t0;
    state->v[2] = 521288629U + t1;
    state->v[3] = 88675123U ^ t1;
    state->v[4] = 5783321U + t0;
    xw_skip(state->v, subsequence, 1); /* 2^67 steps per subsequence; d unchanged (2^67*362437 == 0 mod 2^32) */
    xw_skip(state->v, offset, 0);
    state->d += 362437 * (unsigned int)offset;
}

/* Test hook: the two skips of _curand_init_scratch on a RAW state (no seed scrambling), so that they can be held to an
 * independent XORWOW — rocRAND's engine, oracle/ref_rocrand_xorwow.cpp, tests/test_ref_xorwow.py. */
void o_xorwow_skip_raw(o_curand_state *state, unsigned long long subsequence, unsigned long long offset) {
    xw_skip(state->v, subsequence, 1);
    xw_skip(state->v, offset, 0);
    state->d += 362437 * (unsigned int)offset;
}

void o_set_curand_state_range(o_curand_state *state, long long first, long long count,
                              unsigned long long seed) {
    for (long long i = 0; i < count; i++)
        o_curand_init(seed, (unsigned long long)(first + i), 0, &state[i]);     /* deviceManage.h:127 */
}

void o_set_curand_state(o_curand_state *state, int nx, int ny, unsigned long long seed) { /* deviceManage.h:120-136 */
    o_set_curand_state_range(state, 0, (long long)nx * ny, seed);
}

/* ------------------------------------------------------------------ core/camera.h */

static o_vec3 random_in_unit_disk(o_curand_state *state) {                /* camera.h:6-13 */
    o_vec3 p;
    do {
        /* constructor-argument order fixed to source order (1st draw = x) */
        float a = o_curand_uniform(state);
        float b = o_curand_uniform(state);
        p = sub(muls(2.0f, v3(a, b, 0)), v3(1, 1, 0));
    } while (dot(p, p) >= 1.0f);
    return p;
}

void o_camera_init(o_camera *c, const float lookfrom_[3], const float lookat_[3],
                   const float vup_[3], float vfov, float aspect, float aperture,
                   float focus_dist) {                                    /* camera.h:18-38 */
    o_vec3 lookfrom = v3(lookfrom_[0], lookfrom_[1], lookfrom_[2]);
    o_vec3 lookat = v3(lookat_[0], lookat_[1], lookat_[2]);
    o_vec3 vup = v3(vup_[0], vup_[1], vup_[2]);
    c->lens_radius = aperture / 2.0f;
    float theta = vfov * M_PI / 180.0f; /* double product and quotient, rounded once to float */
    float half_height = tanf(theta / 2.0f);
    float half_width = half_height * aspect;

    c->origin = lookfrom;
    c->z = unit_vector(sub(lookfrom, lookat));
    c->x = unit_vector(cross(vup, c->z));
    c->y = cross(c->z, c->x);

    c->lower_left_corner = sub(sub(sub(c->origin, muls(half_width * focus_dist, c->x)),
                                   muls(half_height * focus_dist, c->y)),
                               muls(focus_dist, c->z));
    c->horizontal = muls(2.0f * half_width * focus_dist, c->x);
    c->vertical = muls(2.0f * half_height * focus_dist, c->y);
}

static o_ray get_ray(const o_camera *c, float s, float t, o_curand_state *state) { /* camera.h:40-44 */
    o_vec3 rd = muls(c->lens_radius, random_in_unit_disk(state));
    o_vec3 offset = add(muls(rd.e[0], c->x), muls(rd.e[1], c->y));
    return make_ray(add(c->origin, offset),
                    sub(sub(add(add(c->lower_left_corner, muls(s, c->horizontal)), muls(t, c->vertical)), c->origin), offset),
                    0.0f);
}

/* ------------------------------------------------------------------ hitable/transform.h */

static o_ray ScaleRay(const o_transform *tr, const o_ray *r) {            /* transform.h:50-54 */
    o_vec3 dir = divv(r->_direction, tr->scale);
    return make_ray(r->_origin, unit_vector(dir), r->_time * length(dir));
}
static o_ray RotateRay(const o_transform *tr, const o_ray *r) {           /* transform.h:43-48 */
    return make_ray(rotate(r->_origin, tr->rotation), rotate(r->_direction, tr->rotation), r->_time);
}
static o_ray TranslateRay(const o_transform *tr, const o_ray *r) {        /* transform.h:38-41 */
    return make_ray(sub(r->_origin, tr->position), r->_direction, r->_time);
}
static o_ray TransformRay(const o_transform *tr, const o_ray *r) {        /* transform.h:11-14 */
    o_ray a = ScaleRay(tr, r);
    o_ray b = RotateRay(tr, &a);
    return TranslateRay(tr, &b);
}

/* ------------------------------------------------------------------ hitable/hitable.h */

typedef struct { float t, u, v; o_vec3 p, normal; int mat_set; int tri; } o_hitrec; /* hitable.h:11-18; tri = index of the triangle mat_ptr belongs to */

/* shapes/triangle.h:57-100 */
static int triangle_collision_detection(const o_triangle *tri, const o_ray *r, float t_min, float t_max, o_hitrec *rec) {
    (void)t_min; (void)t_max; /* ignored by the reference too */
    if (dot(r->_direction, tri->normal) < 0) return 0;
    o_vec3 vertex0 = tri->vertices[0];
    o_vec3 vertex1 = tri->vertices[1];
    o_vec3 vertex2 = tri->vertices[2];

    o_vec3 edge1, edge2, h, s, q;
    float a, f, u, v;

    edge1 = sub(vertex1, vertex0);
    edge2 = sub(vertex2, vertex0);
    h = cross(r->_direction, edge2);
    a = dot(edge1, h);

    if (a < tri->EPSILON && tri->backCulling) return 0;
    if (a > -tri->EPSILON && a < tri->EPSILON) return 0;

    f = 1.0 / a; /* double site */
    s = sub(r->_origin, vertex0);
    u = f * dot(s, h);

    if (u < 0.0 || u > 1.0) return 0;

    q = cross(s, edge1);
    v = f * dot(r->_direction, q);
    if (v < 0.0 || u + v > 1.0) return 0;

    float t = f * dot(edge2, q);

    rec->t = t;
    rec->p = point_at_t(r, rec->t);
    rec->normal = tri->normal;
    rec->mat_set = 1;
    return 1;
}

static int triangle_hit(const o_triangle *tri, const o_ray *r, float t_min, float t_max, o_hitrec *rec) { /* hitable.h:29-38 */
    o_ray tr = TransformRay(&tri->transform, r);
    int flag = triangle_collision_detection(tri, &tr, t_min, t_max, rec);
    rec->normal = rotate(rec->normal, tri->transform.rotation); /* also on a miss */
    return flag;
}

static int list_collision_detection(const o_world *w, const o_ray *r, float t_min, float t_max, o_hitrec *rec, int *index) { /* hitable_list.h:76-91 */
    o_hitrec tmp_rec;
    memset(&tmp_rec, 0, sizeof tmp_rec); /* reference leaves it uninitialised; only read after a write on accepted hits */
    int hit_anything = 0;
    double closest_so_far = t_max;
    for (int i = 0; i < w->list_size; i++) {
        if (triangle_hit(&w->list[i], r, t_min, (float)closest_so_far, &tmp_rec) && tmp_rec.t < closest_so_far) {
            hit_anything = 1;
            closest_so_far = tmp_rec.t;
            *rec = tmp_rec;
            rec->tri = i;
            if (index) *index = i;
        }
    }
    return hit_anything;
}

static int world_hit(const o_world *w, const o_ray *r, float t_min, float t_max, o_hitrec *rec, int *index) { /* hitable.h:29-38 on the list */
    o_ray tr = TransformRay(&w->transform, r);
    int flag = list_collision_detection(w, &tr, t_min, t_max, rec, index);
    rec->normal = rotate(rec->normal, w->transform.rotation);
    return flag;
}

int o_world_hit(const o_world *world, const float origin[3], const float direction[3],
                float *t_out, int *index_out, float normal_out[3]) {
    o_ray r = make_ray(v3(origin[0], origin[1], origin[2]), v3(direction[0], direction[1], direction[2]), 0.0f);
    o_hitrec rec;
    memset(&rec, 0, sizeof rec);
    int idx = -1;
    int flag = world_hit(world, &r, 0.001f, FLT_MAX, &rec, &idx);
    if (t_out) *t_out = rec.t;
    if (index_out) *index_out = flag ? idx : -1;
    if (normal_out) { normal_out[0] = rec.normal.e[0]; normal_out[1] = rec.normal.e[1]; normal_out[2] = rec.normal.e[2]; }
    return flag;
}

/* ------------------------------------------------------------------ core/aabb.h */

static float ffmin(float a, float b) { return a < b ? a : b; }           /* aabb.h:7 */
static float ffmax(float a, float b) { return a > b ? a : b; }           /* aabb.h:8 */

static int aabb_hit(const o_aabb *box, const o_ray *r, float t_min, float t_max) { /* aabb.h:30-43 */
    for (int a = 0; a < 3; a++) {
        float t0 = ffmin((box->_min.e[a] - r->_origin.e[a]) / r->_direction.e[a],
                         (box->_max.e[a] - r->_origin.e[a]) / r->_direction.e[a]);
        float t1 = ffmax((box->_min.e[a] - r->_origin.e[a]) / r->_direction.e[a],
                         (box->_max.e[a] - r->_origin.e[a]) / r->_direction.e[a]);
        t_min = ffmax(t0, t_min);
        t_max = ffmin(t1, t_max);
        if (t_max <= t_min) return 0;
    }
    return 1;
}

static o_aabb surrounding_box(o_aabb box0, o_aabb box1) {                 /* aabb.h:58-66: fmin/fmax */
    o_aabb r;
    r._min = v3(fminf(box0._min.e[0], box1._min.e[0]), fminf(box0._min.e[1], box1._min.e[1]), fminf(box0._min.e[2], box1._min.e[2]));
    r._max = v3(fmaxf(box0._max.e[0], box1._max.e[0]), fmaxf(box0._max.e[1], box1._max.e[1]), fmaxf(box0._max.e[2], box1._max.e[2]));
    return r;
}

static void TransformAABB(const o_transform *tr, o_aabb *aabb) {          /* transform.h:22-31 */
    o_vec3 rotetedMin = rotate(aabb->_min, tr->rotation);
    o_vec3 rotetedMax = rotate(aabb->_max, tr->rotation);
    /* minVec3 / maxVec3, vec3.h:233-239: CUDA's min/max(float,float) = fminf/fmaxf */
    o_vec3 mn = v3(fminf(rotetedMin.e[0], rotetedMax.e[0]), fminf(rotetedMin.e[1], rotetedMax.e[1]), fminf(rotetedMin.e[2], rotetedMax.e[2]));
    o_vec3 mx = v3(fmaxf(rotetedMin.e[0], rotetedMax.e[0]), fmaxf(rotetedMin.e[1], rotetedMax.e[1]), fmaxf(rotetedMin.e[2], rotetedMax.e[2]));
    aabb->_min = add(mn, tr->position);
    aabb->_max = add(mx, tr->position);
}

static void triangle_bounding_box(const o_triangle *t, o_aabb *bbox) {    /* triangle.h:102-115 (min/max = fminf/fmaxf) */
    float minX = fminf(t->vertices[0].e[0], fminf(t->vertices[1].e[0], t->vertices[2].e[0]));
    float minY = fminf(t->vertices[0].e[1], fminf(t->vertices[1].e[1], t->vertices[2].e[1]));
    float minZ = fminf(t->vertices[0].e[2], fminf(t->vertices[1].e[2], t->vertices[2].e[2]));
    float maxX = fmaxf(t->vertices[0].e[0], fmaxf(t->vertices[1].e[0], t->vertices[2].e[0]));
    float maxY = fmaxf(t->vertices[0].e[1], fmaxf(t->vertices[1].e[1], t->vertices[2].e[1]));
    float maxZ = fmaxf(t->vertices[0].e[2], fmaxf(t->vertices[1].e[2], t->vertices[2].e[2]));
    bbox->_min = v3(minX, minY, minZ);
    bbox->_max = v3(maxX, maxY, maxZ);
}

static void triangle_GetBV(const o_triangle *t, o_aabb *box) {            /* hitable.h:45-50 */
    triangle_bounding_box(t, box);
    TransformAABB(&t->transform, box);
}

/* ------------------------------------------------------------------ hitable/bvh.h
 * BVHNode (bvh.h:48-190) over the triangles of a HitableList.  The reference sorts the list's
 * Hitable* array in place with thrust::sort inside ONE device thread (bvh.h:86-92) and a
 * comparator that is not a strict weak ordering (BoxCompare returns true when the keys are
 * equal, bvh.h:36-41), so the resulting order is whatever the algorithm does with it.
 * PARITY UNPINNED at this boundary — the rule adopted here: Thrust's sequential backend
 * (what thrust::sort resolves to inside a device thread), i.e. the iterative stable merge
 * sort of thrust/system/detail/sequential/stable_merge_sort.inl — insertion sort of every
 * 32-element partition (insertion_sort.h), then merges of adjacent partitions ping-ponging
 * with a temporary array (merge.inl) — restated below with the comparator fed literally. */

struct o_bvh {
    o_world *world;        /* the list whose triangles the leaves point at (not owned) */
    int *order;            /* the list's Hitable* array after all in-place sorts: order[slot] = triangle index */
    o_bvh_node *nodes;     /* preorder; nodes[0] is the root */
    int n_nodes, cap;
};

static int box_compare(const float *key, int a, int b) {                  /* BoxCompare::operator(), bvh.h:11-41 */
    float val1 = key[a], val2 = key[b];
    if (val1 - val2 < 0.0) return 0;
    else return 1;
}

static void thrust_insertion_sort(int *first, int *last, const float *key) { /* sequential/insertion_sort.h */
    if (first == last) return;
    for (int *i = first + 1; i != last; ++i) {
        int tmp = *i;
        if (box_compare(key, tmp, *first)) {
            memmove(first + 1, first, (size_t)(i - first) * sizeof(int)); /* copy_backward(first, i, i + 1) */
            *first = tmp;
        } else {
            int *j = i, *k = i - 1;
            while (box_compare(key, tmp, *k)) { *j = *k; j = k; --k; }
            *j = tmp;
        }
    }
}

static void thrust_merge(const int *first1, const int *last1, const int *first2, const int *last2, int *result, const float *key) { /* sequential/merge.inl */
    while (first1 != last1 && first2 != last2) {
        if (box_compare(key, *first2, *first1)) { *result = *first2; ++first2; }
        else { *result = *first1; ++first1; }
        ++result;
    }
    while (first1 != last1) *result++ = *first1++;
    while (first2 != last2) *result++ = *first2++;
}

static void thrust_seq_sort(int *first, int n, const float *key, int *temp) { /* iterative_stable_merge_sort */
    int partition_size = 32;
    for (int at = 0; at < n; at += partition_size)                           /* insertion_sort_each */
        thrust_insertion_sort(first + at, first + (at + partition_size < n ? at + partition_size : n), key);
    if (n <= partition_size) return;
    int ping = 1;
    for (; partition_size < n; partition_size *= 2, ping = !ping) {          /* merge_adjacent_partitions */
        const int *src = ping ? first : temp;
        int *dst = ping ? temp : first;
        for (int at = 0; at < n; at += 2 * partition_size) {
            int mid = at + partition_size < n ? at + partition_size : n;
            int end = mid + partition_size < n ? mid + partition_size : n;
            thrust_merge(src + at, src + mid, src + mid, src + end, dst + at, key);
        }
    }
    if (!ping) memcpy(first, temp, (size_t)n * sizeof(int));
}

/* the restatement above on a key array (tests/test_ref_thrust.py holds it to rocThrust's thrust::sort) */
int o_thrust_sort(int n, const float *key, int *order) {
    if (n < 0) return -1;
    if (n == 0) return 0;
    int *temp = (int *)malloc((size_t)n * sizeof(int));
    if (!temp) return -1;
    thrust_seq_sort(order, n, key, temp);
    free(temp);
    return 0;
}

static o_transform identity_transform(void) {                              /* Transform::ResetTransform, transform.h:16-20 */
    o_transform t; t.position = v1(0); t.rotation = v1(0); t.scale = v1(1); return t;
}

static void list_bounding_box(const o_bvh *b, const o_bvh_node *leaf, o_aabb *box) { /* HitableList::bounding_box, hitable_list.h:94-118 */
    o_aabb tmp_box;
    triangle_GetBV(&b->world->list[b->order[leaf->first]], &tmp_box);
    *box = tmp_box;
    for (int i = 1; i < leaf->count; i++) {
        triangle_GetBV(&b->world->list[b->order[leaf->first + i]], &tmp_box);
        *box = surrounding_box(*box, tmp_box);
    }
}

/* BVHNode::BVHNode(l, n, time0, time1, state), bvh.h:76-125.  [first, first+n) is the node's
 * span of the pointer array; keys[3][tri] = box.min() of the triangle after GetBV. */
static int bvh_construct(o_bvh *b, int first, int n, int depth, o_curand_state *state, float *const keys[3], int *temp) {
    int me = b->n_nodes++;
    o_bvh_node *node = &b->nodes[me];
    node->transform = identity_transform();                                /* transform->ResetTransform(), bvh.h:81 */
    node->depth = depth;
    int axis = (int)(3 * o_curand_uniform(state));                         /* bvh.h:84 */
    if (axis == 0) thrust_seq_sort(b->order + first, n, keys[0], temp);    /* BoxCompare(1): x */
    else if (axis == 1) thrust_seq_sort(b->order + first, n, keys[1], temp);
    else thrust_seq_sort(b->order + first, n, keys[2], temp);
    node->axis = axis;
    if (n == 1 || n == 2) {                                                /* bvh.h:95-109 */
        node->isLeaf = 1;
        node->first = first;
        node->count = n;
        node->left = node->right = -1;
        node->list_transform = identity_transform();                       /* childList->transform->ResetTransform() */
        list_bounding_box(b, node, &node->box);                            /* childList->GetBV(0,1,box) = bounding_box + */
        TransformAABB(&node->list_transform, &node->box);                  /* TransformAABB of the childList */
    } else {
        node->isLeaf = 0;
        node->first = first;
        node->count = n;
        int l = bvh_construct(b, first, n / 2, depth + 1, state, keys, temp);          /* bvh.h:111 */
        int r = bvh_construct(b, first + n / 2, n - n / 2, depth + 1, state, keys, temp); /* bvh.h:112 */
        node = &b->nodes[me];
        node->left = l;
        node->right = r;
        o_aabb box_left = b->nodes[l].box, box_right = b->nodes[r].box;    /* GetBV: bounding_box (= box) + TransformAABB */
        TransformAABB(&b->nodes[l].transform, &box_left);
        TransformAABB(&b->nodes[r].transform, &box_right);
        node->box = surrounding_box(box_left, box_right);                  /* bvh.h:121 */
    }
    b->nodes[me].skip = b->n_nodes;                                        /* preorder index after this subtree */
    return me;
}

o_bvh *o_bvh_build(o_world *world, o_curand_state *state) {                /* create_BVH, createScene.h:42-44 */
    int n = world->list_size;
    if (n < 1) return NULL;                                                /* the reference recurses without end on an empty list */
    o_bvh *b = (o_bvh *)calloc(1, sizeof(o_bvh));
    b->world = world;
    b->order = (int *)malloc((size_t)n * sizeof(int));
    b->cap = 2 * n;
    b->nodes = (o_bvh_node *)calloc((size_t)b->cap, sizeof(o_bvh_node));
    float *keys[3];
    for (int a = 0; a < 3; a++) keys[a] = (float *)malloc((size_t)n * sizeof(float));
    for (int i = 0; i < n; i++) {
        b->order[i] = i;
        o_aabb box;
        triangle_GetBV(&world->list[i], &box);                             /* ah->GetBV(0, 0, box_left), bvh.h:19 */
        for (int a = 0; a < 3; a++) keys[a][i] = box._min.e[a];
    }
    int *temp = (int *)malloc((size_t)n * sizeof(int));
    bvh_construct(b, 0, n, 0, state, keys, temp);
    free(temp);
    for (int a = 0; a < 3; a++) free(keys[a]);
    return b;
}

void o_bvh_free(o_bvh *b) {
    if (!b) return;
    free(b->order);
    free(b->nodes);
    free(b);
}

int o_bvh_node_count(const o_bvh *b) { return b->n_nodes; }
const o_bvh_node *o_bvh_nodes(const o_bvh *b) { return b->nodes; }
const int *o_bvh_order(const o_bvh *b) { return b->order; }

static void bvh_update(o_bvh *b, int me) {                                 /* BVHNode::UpdateBVH, bvh.h:135-158 */
    o_bvh_node *node = &b->nodes[me];
    if (node->isLeaf) {
        list_bounding_box(b, node, &node->box);                            /* childList->bounding_box(0, 1, box): no TransformAABB here */
    } else {
        bvh_update(b, node->left);
        bvh_update(b, node->right);
        o_aabb box_left = b->nodes[node->left].box, box_right = b->nodes[node->right].box;
        TransformAABB(&b->nodes[node->left].transform, &box_left);
        TransformAABB(&b->nodes[node->right].transform, &box_right);
        node->box = surrounding_box(box_right, box_left);                  /* bvh.h:155 */
    }
}

void o_bvh_update(o_bvh *b) { bvh_update(b, 0); }                          /* Update_BVH, createScene.h:46-58 */

static int leaf_list_hit(const o_bvh *b, const o_bvh_node *leaf, const o_ray *r, float t_min, float t_max, o_hitrec *rec, int *index) {
    /* childList->hit(...): Hitable::hit on the leaf's HitableList (hitable.h:29-38) + HitableList::collision_detection */
    o_ray tr = TransformRay(&leaf->list_transform, r);
    o_hitrec tmp_rec;
    memset(&tmp_rec, 0, sizeof tmp_rec);
    int hit_anything = 0;
    double closest_so_far = t_max;
    for (int i = 0; i < leaf->count; i++) {
        int tri = b->order[leaf->first + i];
        if (triangle_hit(&b->world->list[tri], &tr, t_min, (float)closest_so_far, &tmp_rec) && tmp_rec.t < closest_so_far) {
            hit_anything = 1;
            closest_so_far = tmp_rec.t;
            *rec = tmp_rec;
            rec->tri = tri;
            if (index) *index = tri;
        }
    }
    rec->normal = rotate(rec->normal, leaf->list_transform.rotation);
    return hit_anything;
}

static int bvh_node_hit(const o_bvh *b, int me, const o_ray *r, float t_min, float t_max, o_hitrec *rec, int *index);

static int bvh_collision_detection(const o_bvh *b, int me, const o_ray *r, float t_min, float t_max, o_hitrec *rec, int *index) { /* bvh.h:160-190 */
    const o_bvh_node *node = &b->nodes[me];
    if (!aabb_hit(&node->box, r, t_min, t_max)) return 0;
    if (node->isLeaf) return leaf_list_hit(b, node, r, t_min, t_max, rec, index);

    o_hitrec left_rec, right_rec;
    memset(&left_rec, 0, sizeof left_rec);
    memset(&right_rec, 0, sizeof right_rec);
    int li = -1, ri = -1;
    int hit_left = bvh_node_hit(b, node->left, r, t_min, t_max, &left_rec, &li);
    int hit_right = bvh_node_hit(b, node->right, r, t_min, t_max, &right_rec, &ri);
    if (hit_left && hit_right) {
        if (left_rec.t < right_rec.t) { *rec = left_rec; if (index) *index = li; }
        else { *rec = right_rec; if (index) *index = ri; }
        return 1;
    } else if (hit_left) {
        *rec = left_rec; if (index) *index = li;
        return 1;
    } else if (hit_right) {
        *rec = right_rec; if (index) *index = ri;
        return 1;
    }
    return 0;
}

static int bvh_node_hit(const o_bvh *b, int me, const o_ray *r, float t_min, float t_max, o_hitrec *rec, int *index) { /* Hitable::hit, hitable.h:29-38 */
    o_ray tr = TransformRay(&b->nodes[me].transform, r);
    int flag = bvh_collision_detection(b, me, &tr, t_min, t_max, rec, index);
    rec->normal = rotate(rec->normal, b->nodes[me].transform.rotation);
    return flag;
}

int o_bvh_hit(const o_bvh *bvh, const float origin[3], const float direction[3],
              float *t_out, int *index_out, float normal_out[3]) {
    o_ray r = make_ray(v3(origin[0], origin[1], origin[2]), v3(direction[0], direction[1], direction[2]), 0.0f);
    o_hitrec rec;
    memset(&rec, 0, sizeof rec);
    int idx = -1;
    int flag = bvh_node_hit(bvh, 0, &r, 0.001f, FLT_MAX, &rec, &idx);
    if (t_out) *t_out = rec.t;
    if (index_out) *index_out = flag ? idx : -1;
    if (normal_out) { normal_out[0] = rec.normal.e[0]; normal_out[1] = rec.normal.e[1]; normal_out[2] = rec.normal.e[2]; }
    return flag;
}

/* ------------------------------------------------------------------ batch forms for tests/test_ref_hit.py
 * (the reference's own headers answer the same questions on the GPU: oracle/ref_hit.hip) */

/* `levels` more HitableList(1) wrappers around the world list, each one more Hitable::hit level (hitable.h:29-38):
 * TransformRay on the way in (identity Transform: ScaleRay re-normalises), rotate(rec.normal, 0) on the way out; the
 * wrapper's own collision_detection accepts its one element's hit iff tmp_rec.t < closest_so_far (hitable_list.h:84) */
int o_world_hit_many(const o_world *world, int levels, long long n, const float *origin, const float *direction,
                     int *hit, float *t_out, int *index_out, float *normal_out, float *p_out) {
    const o_transform id = identity_transform();
    for (long long i = 0; i < n; i++) {
        o_ray r = make_ray(v3(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]),
                           v3(direction[3 * i], direction[3 * i + 1], direction[3 * i + 2]), 0.0f);
        for (int l = 0; l < levels; l++) r = TransformRay(&id, &r);
        o_hitrec rec;
        memset(&rec, 0, sizeof rec);
        int idx = -1;
        int flag = world_hit(world, &r, 0.001f, FLT_MAX, &rec, &idx);
        for (int l = 0; l < levels; l++) {
            if (flag && !(rec.t < (double)FLT_MAX)) flag = 0;
            rec.normal = rotate(rec.normal, id.rotation);
        }
        hit[i] = flag;
        t_out[i] = rec.t;
        index_out[i] = flag ? idx : -1;
        for (int c = 0; c < 3; c++) { normal_out[3 * i + c] = rec.normal.e[c]; p_out[3 * i + c] = rec.p.e[c]; }
    }
    return 0;
}

int o_bvh_hit_many(const o_bvh *bvh, long long n, const float *origin, const float *direction,
                   int *hit, float *t_out, int *index_out, float *normal_out, float *p_out) {
    for (long long i = 0; i < n; i++) {
        o_ray r = make_ray(v3(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]),
                           v3(direction[3 * i], direction[3 * i + 1], direction[3 * i + 2]), 0.0f);
        o_hitrec rec;
        memset(&rec, 0, sizeof rec);
        int idx = -1;
        int flag = bvh_node_hit(bvh, 0, &r, 0.001f, FLT_MAX, &rec, &idx);
        hit[i] = flag;
        t_out[i] = rec.t;
        index_out[i] = flag ? idx : -1;
        for (int c = 0; c < 3; c++) { normal_out[3 * i + c] = rec.normal.e[c]; p_out[3 * i + c] = rec.p.e[c]; }
    }
    return 0;
}

/* AABB::hit (aabb.h:30-43) on n (box, ray, t_min, t_max) rows */
int o_aabb_hit_many(long long n, const float *bmin, const float *bmax, const float *origin, const float *direction,
                    const float *t_min, const float *t_max, int *out) {
    for (long long i = 0; i < n; i++) {
        o_aabb box;
        box._min = v3(bmin[3 * i], bmin[3 * i + 1], bmin[3 * i + 2]);
        box._max = v3(bmax[3 * i], bmax[3 * i + 1], bmax[3 * i + 2]);
        o_ray r = make_ray(v3(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]),
                           v3(direction[3 * i], direction[3 * i + 1], direction[3 * i + 2]), 0.0f);
        out[i] = aabb_hit(&box, &r, t_min[i], t_max[i]);
    }
    return 0;
}

/* op 0: surrounding_box(a, b) (aabb.h:58-66); op 1: Transform(position = b.min, rotation = b.max, scale 1).TransformAABB(a)
 * (transform.h:22-31) */
int o_aabb_op_many(int op, long long n, const float *amin, const float *amax, const float *bmin, const float *bmax,
                   float *omin, float *omax) {
    if (op < 0 || op > 1) return -1;
    for (long long i = 0; i < n; i++) {
        o_aabb a, o;
        a._min = v3(amin[3 * i], amin[3 * i + 1], amin[3 * i + 2]);
        a._max = v3(amax[3 * i], amax[3 * i + 1], amax[3 * i + 2]);
        o_vec3 p = v3(bmin[3 * i], bmin[3 * i + 1], bmin[3 * i + 2]), q = v3(bmax[3 * i], bmax[3 * i + 1], bmax[3 * i + 2]);
        if (op == 0) {
            o_aabb b; b._min = p; b._max = q;
            o = surrounding_box(a, b);
        } else {
            o_transform tr; tr.position = p; tr.rotation = q; tr.scale = v1(1);
            o = a;
            TransformAABB(&tr, &o);
        }
        for (int c = 0; c < 3; c++) { omin[3 * i + c] = o._min.e[c]; omax[3 * i + c] = o._max.e[c]; }
    }
    return 0;
}

/* which 0: every triangle's GetBV (hitable.h:45-50 = Triangle::bounding_box + TransformAABB), n_tris x 3 + n_tris x 3;
 * which 1: the world list's GetBV (HitableList::bounding_box, hitable_list.h:94-118, + the list's TransformAABB), 3 + 3.
 * Returns the flag bounding_box gives (which 1: false for an empty list). */
int o_world_bounding_boxes(const o_world *w, int which, float *omin, float *omax) {
    if (which == 0) {
        for (int i = 0; i < w->list_size; i++) {
            o_aabb box;
            triangle_GetBV(&w->list[i], &box);
            for (int c = 0; c < 3; c++) { omin[3 * i + c] = box._min.e[c]; omax[3 * i + c] = box._max.e[c]; }
        }
        return 1;
    }
    if (w->list_size < 1) return 0;                                       /* hitable_list.h:97 */
    o_aabb box, tmp_box;
    triangle_GetBV(&w->list[0], &tmp_box);                                /* :100 (a Triangle's bounding_box is always true) */
    box = tmp_box;
    for (int i = 1; i < w->list_size; i++) {                              /* :109-116 */
        triangle_GetBV(&w->list[i], &tmp_box);
        box = surrounding_box(box, tmp_box);
    }
    TransformAABB(&w->transform, &box);                                   /* GetBV, hitable.h:48 */
    for (int c = 0; c < 3; c++) { omin[c] = box._min.e[c]; omax[c] = box._max.e[c]; }
    return 1;
}

/* ------------------------------------------------------------------ material/material.h */

static o_vec3 random_in_unit_sphere(o_curand_state *state) {              /* material.h:11-19 */
    o_vec3 p;
    do {
        /* constructor-argument order fixed to source order (x, y, z = 1st, 2nd, 3rd draw) */
        float a = o_curand_uniform(state);
        float b = o_curand_uniform(state);
        float c = o_curand_uniform(state);
        p = sub(muls(2.0, v3(a, b, c)), v3(1, 1, 1)); /* 2.0 converts to float 2.0f at the call (vec3.h:93) */
    } while (squared_length(p) >= 1.0f);
    return p;
}

static o_vec3 reflect(o_vec3 v, o_vec3 n) {                               /* material.h:6-8 */
    return sub(v, muls(2.0f * dot(v, n), n));
}

/* schlick, material.h:22-26.  pow((1.0f - cosine), 5.0f) is CUDA's powf in the reference, which
 * no host libm reproduces bit for bit (PARITY UNPINNED here).  Rule adopted by the oracle and the
 * kernels alike: the fifth power is formed in double (x*x, squared, times x) and rounded once to
 * float — within one ulp of any faithful powf. */
static float pow5f(float x) {
    double p = (double)x, p2 = p * p, p4 = p2 * p2;
    return (float)(p4 * p);
}
static float schlick(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5f(1.0f - cosine);
}

static int refract(o_vec3 v, o_vec3 n, float ni_over_nt, o_vec3 *refracted) { /* material.h:29-39 */
    o_vec3 uv = unit_vector(v);
    float dt = dot(uv, n);
    float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1 - dt * dt);
    if (discriminant > 0) {
        *refracted = sub(muls(ni_over_nt, sub(uv, muls(dt, n))), muls(sqrtf(discriminant), n));
        return 1;
    } else
        return 0;
}

static const o_material *material_of(const o_world *w, int tri) {
    return &w->materials[w->material_of ? w->material_of[tri] : 0];
}

static o_vec3 material_emitted(const o_material *m) {                     /* material.h:49-53,157-159 */
    if (m->kind == O_DIFFUSE_LIGHT) return m->albedo;
    return v3(0, 0, 0);
}

/* Material::scatter of the four classes (material.h:57-161) */
static int material_scatter(const o_material *m, const o_ray *r_in, const o_hitrec *rec,
                            o_vec3 *attenuation, o_ray *scattered, o_curand_state *state) {
    switch (m->kind) {
    case O_LAMBERTIAN: {                                                  /* material.h:61-71 */
        o_vec3 target = add(add(rec->p, rec->normal), random_in_unit_sphere(state));
        *scattered = make_ray(rec->p, sub(target, rec->p), r_in->_time);
        *attenuation = m->albedo;                                         /* ConstantTexture::value, texture.h:17-19 */
        return 1;
    }
    case O_METAL: {                                                       /* material.h:83-92 */
        float fuzz = m->param < 1 ? m->param : 1;                         /* ctor, material.h:79-81 */
        o_vec3 reflected = reflect(unit_vector(r_in->_direction), rec->normal);
        *scattered = make_ray(rec->p, add(reflected, muls(fuzz, random_in_unit_sphere(state))), 0.0f);
        *attenuation = m->albedo;
        return (dot(scattered->_direction, rec->normal) > 0.0f);
    }
    case O_DIELECTRIC: {                                                  /* material.h:103-141 */
        float ref_idx = m->param;
        o_vec3 outward_normal;
        o_vec3 reflected = reflect(r_in->_direction, rec->normal);
        float ni_over_nt;
        *attenuation = v3(1.0, 1.0, 1.0);
        o_vec3 refracted = v3(0, 0, 0);                                   /* uninitialised in the reference; only read after refract() wrote it ... */
        float reflect_prob;
        float cosine;
        if (dot(r_in->_direction, rec->normal) > 0.0f) {
            outward_normal = v3(-rec->normal.e[0], -rec->normal.e[1], -rec->normal.e[2]);
            ni_over_nt = ref_idx;
            cosine = dot(r_in->_direction, rec->normal) / length(r_in->_direction);
            cosine = sqrtf(1 - ref_idx * ref_idx * (1 - cosine * cosine));
        } else {
            outward_normal = rec->normal;
            ni_over_nt = 1.0f / ref_idx;
            cosine = -dot(r_in->_direction, rec->normal) / length(r_in->_direction);
        }
        if (refract(r_in->_direction, outward_normal, ni_over_nt, &refracted)) {
            reflect_prob = schlick(cosine, ref_idx);
        } else {
            reflect_prob = 1.0f;
        }
        /* ... except when reflect_prob is NaN (total internal reflection is 1.0f, so only a NaN
         * cosine gets here with refract() == false): curand_uniform < NaN is false and the
         * reference reads the uninitialised `refracted`; defined here as (0,0,0). */
        if (o_curand_uniform(state) < reflect_prob) {
            *scattered = make_ray(rec->p, reflected, 0.0f);
        } else {
            *scattered = make_ray(rec->p, refracted, 0.0f);
        }
        return 1;
    }
    default:                                                              /* DiffuseLight, material.h:147-156 */
        return 0;
    }
}

/* ------------------------------------------------------------------ core/render.h */

static o_vec3 backgroundSky(o_vec3 d) {                                   /* render.h:41-46 */
    o_vec3 v = unit_vector(d);
    float t = 0.5f * (v.e[1] + 1.0f);
    return lerp3(t, v1(1), v3(0.5f, 0.7f, 1.0f));
}

/* Hitable** world of the render kernel: the triangle list itself (renderListAnimation,
 * render.h:179) or the BVH built over it (renderBVHAnimation, render.h:222). */
typedef struct { const o_world *list; const o_bvh *bvh; } o_scene_ref;

static int scene_hit(const o_scene_ref *sc, const o_ray *r, float t_min, float t_max, o_hitrec *rec) {   /* (*world)->hit(...) */
    if (sc->bvh) return bvh_node_hit(sc->bvh, 0, r, t_min, t_max, rec, NULL);
    return world_hit(sc->list, r, t_min, t_max, rec, NULL);
}

static o_vec3 shade(const o_ray *r, const o_scene_ref *sc, int depth, o_curand_state *state, int frameIndex) { /* render.h:48-67 */
    o_hitrec rec;
    memset(&rec, 0, sizeof rec);
    if (scene_hit(sc, r, 0.001, FLT_MAX, &rec)) {
        const o_material *mat = material_of(sc->list, rec.tri);
        o_ray scattered;
        o_vec3 attenuation;
        o_vec3 emitted = material_emitted(mat);
        if (depth > 0 && material_scatter(mat, r, &rec, &attenuation, &scattered, state)) {
            return add(emitted, mulv(attenuation, shade(&scattered, sc, depth - 1, state, frameIndex)));
        } else {
            return add(emitted, v3(0.1, 0.1, 0.1));
        }
    } else {
        return backgroundSky(r->_direction);
    }
}

static o_vec3 LambertShade(const o_ray *r, const o_scene_ref *sc, int depth, o_curand_state *state, int frameIndex) { /* render.h:70-87 */
    (void)depth; (void)frameIndex;
    o_hitrec rec;
    memset(&rec, 0, sizeof rec);
    if (scene_hit(sc, r, 0.001, FLT_MAX, &rec)) {
        const o_material *mat = material_of(sc->list, rec.tri);
        o_ray scattered;
        o_vec3 attenuation = v3(0, 0, 0); /* DiffuseLight::scatter leaves it unwritten (uninitialised in the reference): defined as 0 */
        o_vec3 emitted = material_emitted(mat);
        material_scatter(mat, r, &rec, &attenuation, &scattered, state);
        float t = dot(r->_direction, rec.normal);
        if (t < 0) t = 0;
        return add(muls(0.2, mulv(muls(t, attenuation), backgroundSky(r->_direction))), emitted);
    } else {
        return backgroundSky(r->_direction);
    }
}

static o_vec3 shade_normal(const o_ray *r, const o_scene_ref *sc, int depth, o_curand_state *state, int frameIndex) { /* render.h:90-103 */
    (void)depth; (void)state; (void)frameIndex;
    o_hitrec rec;
    memset(&rec, 0, sizeof rec);
    if (scene_hit(sc, r, 0.001, FLT_MAX, &rec)) {
        return rec.normal;
    } else {
        return backgroundSky(r->_direction);
    }
}

static void render_window(float *colorBuffer, const o_scene_ref *world, int shader, const o_camera *camera,
                          o_curand_state *state, int nx, int ny, int samples, int max_depth,
                          int frameIndex, int row_begin, int row_end, int col_begin, int col_end) {
    /* render.h:105-129, one "thread" after another in pixel_index order */
    for (int y = row_begin; y < row_end; y++)
        for (int x = col_begin; x < col_end; x++) {
            if ((x >= nx) || (y >= ny)) continue;
            int pixel_index = y * nx + x;
            int ns = samples;
            o_vec3 col = v3(0, 0, 0);
            for (int i = 0; i < ns; i++) {
                float u = (float)(x + o_curand_uniform(&state[pixel_index])) / (float)nx;
                float v = (float)(y + o_curand_uniform(&state[pixel_index])) / (float)ny;
                o_ray r = get_ray(camera, u, v, state); /* base pointer: &state[0], render.h:118 */
                /* render.h:119-121: the reference picks one of these three by (un)commenting */
                if (shader == O_SHADER_SHADE) col = add(col, shade(&r, world, max_depth, &state[pixel_index], frameIndex));
                else if (shader == O_SHADER_NORMAL) col = add(col, shade_normal(&r, world, 0, &state[pixel_index], frameIndex));
                else col = add(col, LambertShade(&r, world, max_depth, &state[pixel_index], frameIndex));
            }
            col = div_assign_s(col, (float)ns);
            col.e[0] = sqrtf(col.e[0]);
            col.e[1] = sqrtf(col.e[1]);
            col.e[2] = sqrtf(col.e[2]);
            o_vec3 c = clip(col);
            colorBuffer[3 * (size_t)pixel_index + 0] = c.e[0];
            colorBuffer[3 * (size_t)pixel_index + 1] = c.e[1];
            colorBuffer[3 * (size_t)pixel_index + 2] = c.e[2];
        }
}

void o_render_window(float *colorBuffer, const o_world *world, const o_camera *camera,
                     o_curand_state *state, int nx, int ny, int samples, int max_depth,
                     int frameIndex, int row_begin, int row_end, int col_begin, int col_end) {
    o_scene_ref sc = {world, NULL};                                       /* renderListAnimation, render.h:179 */
    render_window(colorBuffer, &sc, O_SHADER_LAMBERT, camera, state, nx, ny, samples, max_depth, frameIndex, row_begin, row_end, col_begin, col_end);
}

void o_render_window_bvh(float *colorBuffer, const o_bvh *bvh, const o_camera *camera,
                         o_curand_state *state, int nx, int ny, int samples, int max_depth,
                         int frameIndex, int row_begin, int row_end, int col_begin, int col_end) {
    o_scene_ref sc = {bvh->world, bvh};                                   /* renderBVHAnimation, render.h:222 */
    render_window(colorBuffer, &sc, O_SHADER_LAMBERT, camera, state, nx, ny, samples, max_depth, frameIndex, row_begin, row_end, col_begin, col_end);
}

void o_render_window_ex(float *colorBuffer, const o_world *world, const o_bvh *bvh, int shader,
                        const o_camera *camera, o_curand_state *state, int nx, int ny, int samples,
                        int max_depth, int frameIndex, int row_begin, int row_end, int col_begin, int col_end) {
    o_scene_ref sc = {world, bvh};
    render_window(colorBuffer, &sc, shader, camera, state, nx, ny, samples, max_depth, frameIndex, row_begin, row_end, col_begin, col_end);
}

void o_render(float *colorBuffer, const o_world *world, const o_camera *camera,
              o_curand_state *state, int nx, int ny, int samples, int max_depth, int frameIndex) {
    o_render_window(colorBuffer, world, camera, state, nx, ny, samples, max_depth, frameIndex, 0, ny, 0, nx);
}

void o_to_rgba8(const float *colorBuffer, int nx, int ny, unsigned char *rgba) { /* render.h:135-146 */
    for (int i = 0; i < ny; i++)
        for (int j = 0; j < nx; j++) {
            size_t pixel_index = (size_t)(ny - 1 - i) * nx + j;
            for (int c = 0; c < 3; c++) {
                /* char(255.99 * c): double multiply, truncation toward zero, low 8 bits kept.
                 * NaN (the reference's cvttsd2si yields INT_MIN, low byte 0) maps to 0. */
                double d = 255.99 * colorBuffer[3 * pixel_index + c];
                int iv = (d != d) ? 0 : (int)d;
                rgba[4 * ((size_t)i * nx + j) + c] = (unsigned char)iv;
            }
            rgba[4 * ((size_t)i * nx + j) + 3] = 255;
        }
}

/* ------------------------------------------------------------------ createScene.h */

o_world *o_world_from_mesh(const float *points, int n_points, const float *idx3f, int n_tris,
                           const float *normals, int index_base, const float albedo[3]) { /* createScene.h:175-190 */
    o_world *w = (o_world *)calloc(1, sizeof(o_world));
    /* HitableList(int n) -> Hitable() -> new Transform(): position/rotation uninitialised in
     * the reference (transform.h:8); defined here as the identity. */
    w->transform.position = v1(0);
    w->transform.rotation = v1(0);
    w->transform.scale = v1(1);
    w->list_size = n_tris;
    w->list = (o_triangle *)calloc(n_tris > 0 ? n_tris : 1, sizeof(o_triangle));
    w->albedo = v3(albedo[0], albedo[1], albedo[2]);
    w->materials = (o_material *)calloc(1, sizeof(o_material));         /* new Lambertian(new ConstantTexture(...)), createScene.h:180 */
    w->materials[0].kind = O_LAMBERTIAN;
    w->materials[0].albedo = w->albedo;
    w->n_materials = 1;
    w->material_of = NULL;
    for (int i = 0; i < n_tris; i++) {
        int id[3];
        for (int k = 0; k < 3; k++) {
            id[k] = (int)idx3f[3 * i + k] - index_base;
            if (id[k] < 0 || id[k] >= n_points) { o_world_free(w); return NULL; }
        }
        o_triangle *tri = &w->list[i];
        /* reversed winding: {points[idx[2]], points[idx[1]], points[idx[0]]}, createScene.h:183 */
        for (int k = 0; k < 3; k++) {
            int p = id[2 - k];
            tri->vertices[k] = v3(points[3 * p], points[3 * p + 1], points[3 * p + 2]);
        }
        tri->transform.position = v1(0); /* new Transform(vec3(0), vec3(0), vec3(1)), createScene.h:184 */
        tri->transform.rotation = v1(0);
        tri->transform.scale = v1(1);
        tri->EPSILON = 0.000001; /* triangle.h:9,22 */
        tri->backCulling = 0;
        if (normals) {
            tri->normal = v3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]); /* triangle.h:21-29 */
        } else {
            o_vec3 edge1 = sub(tri->vertices[1], tri->vertices[0]);              /* triangle.h:14-17 */
            o_vec3 edge2 = sub(tri->vertices[2], tri->vertices[0]);
            tri->normal = unit_vector(cross(edge1, edge2));
        }
    }
    return w;
}

/* update_pose, createScene.h:99-110 (the device half of updateFBXObj, :126-140): every
 * triangle's vertices are re-read from the new point positions through the same float index
 * triplets and the same reversed winding; the stored normal is NOT touched (Triangle::SetVertices,
 * shapes/triangle.h:40-45). */
int o_world_update_pose(o_world *w, const float *new_points, int n_points, const float *idx3f, int index_base) {
    for (int i = 0; i < w->list_size; i++) {
        int id[3];
        for (int k = 0; k < 3; k++) {
            id[k] = (int)idx3f[3 * i + k] - index_base;
            if (id[k] < 0 || id[k] >= n_points) return -1;
        }
        for (int k = 0; k < 3; k++) {
            int p = id[2 - k];
            w->list[i].vertices[k] = v3(new_points[3 * p], new_points[3 * p + 1], new_points[3 * p + 2]);
        }
    }
    return 0;
}

int o_world_set_materials(o_world *w, const o_material *table, int n_materials, const int *triangle_material) {
    if (!w || !table || n_materials < 1) return -1;
    for (int k = 0; k < n_materials; k++)
        if (table[k].kind < O_LAMBERTIAN || table[k].kind > O_DIFFUSE_LIGHT) return -1;
    if (triangle_material)
        for (int i = 0; i < w->list_size; i++)
            if (triangle_material[i] < 0 || triangle_material[i] >= n_materials) return -1;
    free(w->materials);
    free(w->material_of);
    w->materials = (o_material *)malloc((size_t)n_materials * sizeof(o_material));
    memcpy(w->materials, table, (size_t)n_materials * sizeof(o_material));
    w->n_materials = n_materials;
    w->material_of = NULL;
    if (triangle_material) {
        w->material_of = (int *)malloc((size_t)(w->list_size > 0 ? w->list_size : 1) * sizeof(int));
        memcpy(w->material_of, triangle_material, (size_t)w->list_size * sizeof(int));
    }
    return 0;
}

void o_world_free(o_world *w) {
    if (!w) return;
    free(w->list);
    free(w->materials);
    free(w->material_of);
    free(w);
}

/* ------------------------------------------------------------------ Loader/obj_loader.h */

int o_parse_obj(const char *path, float *points, float *idx3f, int *n_points, int *n_tris) { /* obj_loader.h:20-111,120-133 */
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int np = 0, nt = 0;
    char line[1024];
    while (fgets(line, sizeof line, f)) {
        char label[16];
        int consumed = 0;
        if (sscanf(line, "%15s%n", label, &consumed) != 1) continue;
        if (strcmp(label, "v") == 0) {
            float p[3] = {0, 0, 0};
            sscanf(line + consumed, "%f %f %f", &p[0], &p[1], &p[2]);
            if (points) { points[3 * np] = p[0]; points[3 * np + 1] = p[1]; points[3 * np + 2] = p[2]; }
            np++;
        } else if (strcmp(label, "f") == 0) {
            float p[3] = {0, 0, 0};
            sscanf(line + consumed, "%f %f %f", &p[0], &p[1], &p[2]); /* indices read as floats, 1-based, kept */
            if (idx3f) { idx3f[3 * nt] = p[0]; idx3f[3 * nt + 1] = p[1]; idx3f[3 * nt + 2] = p[2]; }
            nt++;
        }
    }
    fclose(f);
    *n_points = np;
    *n_tris = nt;
    if (points) {
        o_vec3 mean = v3(0, 0, 0);                                        /* computeMean, obj_loader.h:120-126 */
        for (int i = 0; i < np; i++) mean = add(mean, v3(points[3 * i], points[3 * i + 1], points[3 * i + 2]));
        mean = divs(mean, (float)np);
        for (int i = 0; i < np; i++) {                                    /* centering, obj_loader.h:129-133 */
            points[3 * i] -= mean.e[0];
            points[3 * i + 1] -= mean.e[1];
            points[3 * i + 2] -= mean.e[2];
        }
    }
    return 0;
}
