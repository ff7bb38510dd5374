/*
 * orc_shapes.c -- ray/primitive tests of the CPU oracle (test infrastructure
 * only; see orc_math.h for the parity statement).
 *
 * Follows, with the corrections listed beside each function:
 *   shapes/src/sphere.rs:28-47,66-206 + quadric.rs:8-36 + geometry/src/float.rs:182-204
 *   shapes/src/triangle.rs:91-106,108-348
 *   geometry/src/transform.rs:332-516,567-587
 * Rectangles, boxes, moving spheres and constant media have no code in the
 * reference; their behaviour is specified by the scene files that use them
 * (test/cornell.toml, test/week.toml) and written down here.
 */
#include "orc_internal.h"

#define G2 orc_gamma(2)
#define G3 orc_gamma(3)
#define G5 orc_gamma(5)
#define G7 orc_gamma(7)
#define TWO_PI 6.28318548f
#define PI_F 3.14159274f

/* ---- affine helpers (row-major 3x4) -------------------------------------- */
static v3 xf_point(const float *m, v3 p) { /* transform.rs:335-341 */
    return v3_make(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3],
                   m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
                   m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
static v3 xf_vector(const float *m, v3 p) { /* transform.rs:435-442 */
    return v3_make(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
                   m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
/* normals use the transpose of the inverse, transform.rs:504-516 */
static v3 xf_normal(const float *inv, v3 n) {
    return v3_make(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z,
                   inv[1] * n.x + inv[5] * n.y + inv[9] * n.z,
                   inv[2] * n.x + inv[6] * n.y + inv[10] * n.z);
}
static v3 xf_abs_sum(const float *m, v3 p) { /* transform.rs:372-376 */
    return v3_make(fabsf(m[0] * p.x) + fabsf(m[1] * p.y) + fabsf(m[2] * p.z) + fabsf(m[3]),
                   fabsf(m[4] * p.x) + fabsf(m[5] * p.y) + fabsf(m[6] * p.z) + fabsf(m[7]),
                   fabsf(m[8] * p.x) + fabsf(m[9] * p.y) + fabsf(m[10] * p.z) + fabsf(m[11]));
}
/* point with incoming error -> point + outgoing error, transform.rs:390-429.
 * The incoming error is scaled by |M| without the translation column (the
 * reference adds |m[i][3]| to that sum too, which turns a translation of 400
 * units into a 400-unit error bound). */
static v3 xf_point_err(const float *m, v3 p, v3 e, v3 *err_out) {
    v3 es = v3_make(fabsf(m[0] * e.x) + fabsf(m[1] * e.y) + fabsf(m[2] * e.z),
                    fabsf(m[4] * e.x) + fabsf(m[5] * e.y) + fabsf(m[6] * e.z),
                    fabsf(m[8] * e.x) + fabsf(m[9] * e.y) + fabsf(m[10] * e.z));
    v3 as = xf_abs_sum(m, p);
    *err_out = v3_add(v3_scale(es, G3 + 1.0f), v3_scale(as, G3));
    return xf_point(m, p);
}
static aabb xf_bounds(const float *m, aabb b) { /* transform.rs:645-657 */
    aabb r = aabb_empty();
    for (int c = 0; c < 8; c++) {
        v3 p = v3_make((c & 1) ? b.hi.x : b.lo.x, (c & 2) ? b.hi.y : b.lo.y,
                       (c & 4) ? b.hi.z : b.lo.z);
        r = aabb_union_pt(r, xf_point(m, p));
    }
    return r;
}

static v3 ld3(const float *p) { return v3_make(p[0], p[1], p[2]); }

/* ---- quadratic, geometry/src/float.rs:182-204 in plain fp32 ---------------
 * The interval arithmetic of EFloat is not reproduced (its error term is
 * eps * (1/0) = inf in the reference, which rejects every hit). */
static int quadratic(float a, float b, float c, float *t0, float *t1) {
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return 0;
    float root = sqrtf(disc);
    float q = (b < 0.0f) ? -0.5f * (b - root) : -0.5f * (b + root);
    float r0 = q / a;
    float r1 = c / q;
    if (r0 > r1) { float tmp = r0; r0 = r1; r1 = tmp; }
    *t0 = r0;
    *t1 = r1;
    return 1;
}

/* ray vs sphere of radius |r| centred at the origin; o is relative to centre */
static int sphere_roots(v3 o, v3 d, float r, float *t0, float *t1) {
    float a = v3_dot(d, d);
    float b = 2.0f * v3_dot(d, o);
    float c = v3_dot(o, o) - r * r;
    return quadratic(a, b, c, t0, t1);
}

/* root selection of quadric.rs:24-35 on plain values */
static int sphere_pick(float t0, float t1, float t_max, float *t) {
    if (t0 > t_max || t1 <= 0.0f) return 0;
    float ts = t0;
    if (t0 <= 0.0f) {
        if (t1 > t_max) return 0;
        ts = t1;
    }
    if (!(ts > 0.0f && ts <= t_max)) return 0; /* rejects NaN */
    *t = ts;
    return 1;
}

static v3 msphere_centre(const orc_msphere *m, float time) {
    float dt = time - m->time0;
    return v3_make(m->c0[0] + dt * m->dcdt[0], m->c0[1] + dt * m->dcdt[1],
                   m->c0[2] + dt * m->dcdt[2]);
}

/* ---- slabs of an axis-aligned box ---------------------------------------- */
static int box_interval(v3 o, v3 d, const rt_box *b, float *tn_out, float *tf_out, int *fn_out,
                        int *ff_out) {
    float tn = -INFINITY, tf = INFINITY;
    int fn = 0, ff = 0;
    for (int i = 0; i < 3; i++) {
        float oi = v3_get(o, i), di = v3_get(d, i);
        if (di == 0.0f) {
            if (oi < b->lo[i] || oi > b->hi[i]) return 0;
            continue;
        }
        float inv = 1.0f / di;
        float t0 = (b->lo[i] - oi) * inv;
        float t1 = (b->hi[i] - oi) * inv;
        int f0 = 2 * i, f1 = 2 * i + 1;
        if (inv < 0.0f) {
            float tt = t0; t0 = t1; t1 = tt;
            int ft = f0; f0 = f1; f1 = ft;
        }
        if (t0 > tn) { tn = t0; fn = f0; }
        if (t1 < tf) { tf = t1; ff = f1; }
    }
    if (tn > tf) return 0;
    *tn_out = tn; *tf_out = tf; *fn_out = fn; *ff_out = ff;
    return 1;
}

static void rect_axes(uint32_t axis, int *A, int *B) {
    if (axis == 0) { *A = 1; *B = 2; }
    else if (axis == 1) { *A = 0; *B = 2; }
    else { *A = 0; *B = 1; }
}

/* ---- general sphere (reference Sphere) ----------------------------------- */
void orc_derive_gsphere(const rt_gsphere *in, orc_gsphere *g) {
    memcpy(g->o2w, in->o2w, sizeof g->o2w);
    memcpy(g->w2o, in->w2o, sizeof g->w2o);
    float r = in->radius;
    float zlo = in->z_min < in->z_max ? in->z_min : in->z_max;
    float zhi = in->z_min < in->z_max ? in->z_max : in->z_min;
    g->radius = r;
    g->z_min = orc_clamp(zlo, -r, r); /* sphere.rs:36-37 */
    g->z_max = orc_clamp(zhi, -r, r);
    g->theta_min = orc_acos(orc_clamp(g->z_min / r, -1.0f, 1.0f)); /* sphere.rs:43-44 */
    g->theta_max = orc_acos(orc_clamp(g->z_max / r, -1.0f, 1.0f));
    g->phi_max = orc_clamp(in->phi_max_deg, 0.0f, 360.0f) * 0.0174532924f; /* sphere.rs:45 */
    const float *m = in->o2w; /* swaps_handedness, transform.rs:259-266 */
    float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) +
                m[2] * (m[4] * m[9] - m[5] * m[8]);
    g->flip = (det < 0.0f) ? 1 : 0;
}

/* world ray -> object ray, transform.rs:567-587 (origin nudged along the
 * direction by the transform's rounding error; t_max unchanged) */
static void gsphere_ray(const orc_gsphere *g, const rt_ray *ray, v3 *o, v3 *d) {
    v3 wo = ld3(ray->o), wd = ld3(ray->d);
    v3 oo = xf_point(g->w2o, wo);
    v3 oerr = v3_scale(xf_abs_sum(g->w2o, wo), G3);
    v3 dd = xf_vector(g->w2o, wd);
    float len_sq = v3_dot(dd, dd);
    if (len_sq > 0.0f) {
        float dt = v3_dot(v3_abs(dd), oerr) / len_sq;
        oo = v3_add(oo, v3_scale(dd, dt));
    }
    *o = oo;
    *d = dd;
}

/* hit point refinement + phi, sphere.rs:70-79 (evaluated on the object-space
 * ray, which the reference's `ray.at` call misses) */
static v3 gsphere_point(const orc_gsphere *g, v3 o, v3 d, float t, float *phi) {
    v3 p = v3_add(o, v3_scale(d, t));
    p = v3_scale(p, g->radius / v3_len(p));
    if (p.x == 0.0f && p.y == 0.0f) p.x = 1e-5f * g->radius;
    float ph = orc_atan2(p.y, p.x);
    if (ph < 0.0f) ph += TWO_PI;
    *phi = ph;
    return p;
}
static int gsphere_clipped(const orc_gsphere *g, v3 p, float phi) { /* sphere.rs:82-84 */
    return (g->z_min > -g->radius && p.z < g->z_min) || (g->z_max < g->radius && p.z > g->z_max) ||
           phi > g->phi_max;
}
static int gsphere_test(const orc_gsphere *g, const rt_ray *ray, float *t_out) {
    v3 o, d;
    gsphere_ray(g, ray, &o, &d);
    float t0, t1, ts, phi;
    if (!sphere_roots(o, d, g->radius, &t0, &t1)) return 0;
    if (!sphere_pick(t0, t1, ray->t_max, &ts)) return 0;
    v3 p = gsphere_point(g, o, d, ts, &phi);
    if (gsphere_clipped(g, p, phi)) { /* sphere.rs:85-108 */
        if (ts == t1 || t1 > ray->t_max) return 0;
        ts = t1;
        if (!(ts > 0.0f)) return 0;
        p = gsphere_point(g, o, d, ts, &phi);
        if (gsphere_clipped(g, p, phi)) return 0;
    }
    *t_out = ts;
    return 1;
}

/* ---- triangle, shapes/src/triangle.rs:108-202 ------------------------------
 * Corrections: the axis permutation wraps with (k+1)%3 (the reference's
 * `if kz+1==3 {kz+1} else {0}` indexes out of range), and the same accept
 * test serves closest-hit and any-hit. */
static int tri_test(const float *v, const rt_ray *ray, float *t_out, float bary[3]) {
    v3 o = ld3(ray->o), d = ld3(ray->d);
    v3 p0 = v3_sub(ld3(v), o), p1 = v3_sub(ld3(v + 3), o), p2 = v3_sub(ld3(v + 6), o);
    /* max_dimension: last maximal index on ties, vector3.rs:155-162 */
    v3 ad = v3_abs(d);
    int kz = 0;
    float best = ad.x;
    if (ad.y >= best) { kz = 1; best = ad.y; }
    if (ad.z >= best) { kz = 2; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    v3 dp = v3_make(v3_get(d, kx), v3_get(d, ky), v3_get(d, kz));
    v3 a = v3_make(v3_get(p0, kx), v3_get(p0, ky), v3_get(p0, kz));
    v3 b = v3_make(v3_get(p1, kx), v3_get(p1, ky), v3_get(p1, kz));
    v3 c = v3_make(v3_get(p2, kx), v3_get(p2, ky), v3_get(p2, kz));
    float sx = -dp.x / dp.z, sy = -dp.y / dp.z, sz = 1.0f / dp.z;
    a.x += sx * a.z; a.y += sy * a.z;
    b.x += sx * b.z; b.y += sy * b.z;
    c.x += sx * c.z; c.y += sy * c.z;
    float e0 = b.x * c.y - b.y * c.x;
    float e1 = c.x * a.y - c.y * a.x;
    float e2 = a.x * b.y - a.y * b.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) { /* triangle.rs:145-155 */
        double ax = a.x, ay = a.y, bx = b.x, by = b.y, cx = c.x, cy = c.y;
        e0 = (float)(bx * cy - by * cx);
        e1 = (float)(cx * ay - cy * ax);
        e2 = (float)(ax * by - ay * bx);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return 0;
    float det = e0 + e1 + e2;
    if (det == 0.0f) return 0;
    a.z *= sz; b.z *= sz; c.z *= sz;
    float scaled = e0 * a.z + e1 * b.z + e2 * c.z;
    if (det < 0.0f && (scaled >= 0.0f || scaled < ray->t_max * det)) return 0;
    if (det > 0.0f && (scaled <= 0.0f || scaled > ray->t_max * det)) return 0;
    float inv_det = 1.0f / det;
    float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
    float t = scaled * inv_det;
    /* conservative t > delta_t, triangle.rs:186-202 */
    float max_zt = sel_max(sel_max(fabsf(a.z), fabsf(b.z)), fabsf(c.z));
    float delta_z = max_zt * G3;
    float max_xt = sel_max(sel_max(fabsf(a.x), fabsf(b.x)), fabsf(c.x));
    float max_yt = sel_max(sel_max(fabsf(a.y), fabsf(b.y)), fabsf(c.y));
    float delta_x = (max_xt + max_zt) * G5;
    float delta_y = (max_yt + max_zt) * G5;
    float delta_e = 2.0f * (max_xt * max_yt * G2 + delta_y * max_xt + delta_x * max_yt);
    float max_e = sel_max(sel_max(fabsf(e0), fabsf(e1)), fabsf(e2));
    float delta_t =
        3.0f * (max_e * max_zt * G3 + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
    if (t <= delta_t) return 0;
    if (!(t <= ray->t_max)) return 0; /* rejects NaN */
    *t_out = t;
    bary[0] = b0; bary[1] = b1; bary[2] = b2;
    return 1;
}

/* ---- medium: deterministic free-flight sample ----------------------------- */
static float medium_uniform(const rt_ray *ray, uint32_t prim) {
    const uint32_t k = 0x9E3779B1u; /* one multiply-xor step per word, one full mixing round */
    uint32_t h = (prim ^ f2u(ray->d[2])) * k;
    h = (h ^ f2u(ray->d[1])) * k;
    h = (h ^ f2u(ray->d[0])) * k;
    h = (h ^ f2u(ray->o[2])) * k;
    h = (h ^ f2u(ray->o[1])) * k;
    h = orc_hash32(h ^ f2u(ray->o[0]));
    return (float)(h >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;
}

static void to_object(const orc_scene *s, uint32_t xform, const rt_ray *ray, v3 *o, v3 *d) {
    *o = ld3(ray->o);
    *d = ld3(ray->d);
    if (xform) {
        const float *m = s->xforms[xform - 1].w2o;
        *o = xf_point(m, *o);
        *d = xf_vector(m, *d);
    }
}

static int medium_test(const orc_scene *s, uint32_t prim, const rt_medium *m, const rt_ray *ray,
                       float *t_out) {
    float t_in, t_out_b;
    v3 o, d;
    if (m->boundary_kind == RT_BOUNDARY_SPHERE) {
        const rt_sphere *sp = &s->spheres[m->boundary_index];
        o = v3_sub(ld3(ray->o), ld3(sp->c));
        d = ld3(ray->d);
        if (!sphere_roots(o, d, sp->r, &t_in, &t_out_b)) return 0;
    } else {
        int fn, ff;
        to_object(s, m->xform, ray, &o, &d);
        if (!box_interval(o, d, &s->boxes[m->boundary_index], &t_in, &t_out_b, &fn, &ff)) return 0;
    }
    if (t_in < 0.0f) t_in = 0.0f;
    if (t_out_b > ray->t_max) t_out_b = ray->t_max;
    if (!(t_in < t_out_b)) return 0;
    v3 wd = ld3(ray->d);
    float len = v3_len(wd);
    float inside = (t_out_b - t_in) * len;
    float neg_inv_density = -1.0f / m->density;
    float hit_dist = neg_inv_density * orc_log(medium_uniform(ray, prim));
    if (hit_dist > inside) return 0;
    float t = t_in + hit_dist / len;
    if (!(t > 0.0f && t <= ray->t_max)) return 0;
    *t_out = t;
    return 1;
}

/* ---- dispatch -------------------------------------------------------------- */
static aabb sphere_box(v3 c, float r) {
    aabb b = {v3_make(c.x - r, c.y - r, c.z - r), v3_make(c.x + r, c.y + r, c.z + r)};
    return b;
}
static aabb box_of(const rt_box *bx) {
    aabb b = {ld3(bx->lo), ld3(bx->hi)};
    return b;
}

aabb orc_prim_bounds(const orc_scene *s, uint32_t prim) {
    const rt_prim_ref *pr = &s->prims[prim];
    uint32_t kind = pr->kind & 0xffu;
    aabb b = aabb_empty();
    switch (kind) {
    case RT_SPHERE: {
        const rt_sphere *sp = &s->spheres[pr->index];
        b = sphere_box(ld3(sp->c), fabsf(sp->r));
        break;
    }
    case RT_MOVING_SPHERE: {
        const orc_msphere *m = &s->msph[pr->index];
        b = aabb_union(sphere_box(ld3(m->c0), fabsf(m->r)), sphere_box(ld3(m->c1), fabsf(m->r)));
        break;
    }
    case RT_TRIANGLE: { /* triangle.rs:100-106 */
        const float *v = s->tri_verts + 9 * (size_t)pr->index;
        b = aabb_union_pt(aabb_union_pt(aabb_union_pt(b, ld3(v)), ld3(v + 3)), ld3(v + 6));
        break;
    }
    case RT_RECT: {
        const rt_rect *r = &s->rects[pr->index];
        int A, B;
        rect_axes(r->axis, &A, &B);
        float lo[3], hi[3];
        lo[r->axis] = r->k - 1e-4f; hi[r->axis] = r->k + 1e-4f;
        lo[A] = r->a0; hi[A] = r->a1;
        lo[B] = r->b0; hi[B] = r->b1;
        b.lo = ld3(lo); b.hi = ld3(hi);
        if (pr->xform) b = xf_bounds(s->xforms[pr->xform - 1].o2w, b);
        break;
    }
    case RT_BOX: {
        b = box_of(&s->boxes[pr->index]);
        if (pr->xform) b = xf_bounds(s->xforms[pr->xform - 1].o2w, b);
        break;
    }
    case RT_GSPHERE: { /* sphere.rs:59-64 + shapes/src/lib.rs:51-53 */
        const orc_gsphere *g = &s->gsph[pr->index];
        aabb ob = {v3_make(-g->radius, -g->radius, g->z_min), v3_make(g->radius, g->radius, g->z_max)};
        b = xf_bounds(g->o2w, ob);
        break;
    }
    case RT_MEDIUM: {
        const rt_medium *m = &s->media[pr->index];
        if (m->boundary_kind == RT_BOUNDARY_SPHERE) {
            const rt_sphere *sp = &s->spheres[m->boundary_index];
            b = sphere_box(ld3(sp->c), fabsf(sp->r));
        } else {
            b = box_of(&s->boxes[m->boundary_index]);
            if (m->xform) b = xf_bounds(s->xforms[m->xform - 1].o2w, b);
        }
        break;
    }
    default: break;
    }
    return b;
}

int orc_prim_test(const orc_scene *s, uint32_t prim, const rt_ray *ray, float *t_out, float aux[4]) {
    const rt_prim_ref *pr = &s->prims[prim];
    uint32_t kind = pr->kind & 0xffu;
    float t0, t1;
    switch (kind) {
    case RT_SPHERE: {
        const rt_sphere *sp = &s->spheres[pr->index];
        v3 o = v3_sub(ld3(ray->o), ld3(sp->c));
        if (!sphere_roots(o, ld3(ray->d), sp->r, &t0, &t1)) return 0;
        return sphere_pick(t0, t1, ray->t_max, t_out);
    }
    case RT_MOVING_SPHERE: {
        const orc_msphere *m = &s->msph[pr->index];
        v3 o = v3_sub(ld3(ray->o), msphere_centre(m, ray->time));
        if (!sphere_roots(o, ld3(ray->d), m->r, &t0, &t1)) return 0;
        return sphere_pick(t0, t1, ray->t_max, t_out);
    }
    case RT_TRIANGLE:
        return tri_test(s->tri_verts + 9 * (size_t)pr->index, ray, t_out, aux);
    case RT_RECT: {
        const rt_rect *r = &s->rects[pr->index];
        v3 o, d;
        int A, B;
        to_object(s, pr->xform, ray, &o, &d);
        rect_axes(r->axis, &A, &B);
        float t = (r->k - v3_get(o, (int)r->axis)) / v3_get(d, (int)r->axis);
        if (!(t > 0.0f && t <= ray->t_max)) return 0;
        float a = v3_get(o, A) + t * v3_get(d, A);
        float b = v3_get(o, B) + t * v3_get(d, B);
        if (a < r->a0 || a > r->a1 || b < r->b0 || b > r->b1) return 0;
        *t_out = t;
        return 1;
    }
    case RT_BOX: {
        v3 o, d;
        int fn, ff;
        to_object(s, pr->xform, ray, &o, &d);
        if (!box_interval(o, d, &s->boxes[pr->index], &t0, &t1, &fn, &ff)) return 0;
        float t = t0;
        int face = fn;
        if (!(t0 > 0.0f)) { t = t1; face = ff; }
        if (!(t > 0.0f && t <= ray->t_max)) return 0;
        *t_out = t;
        aux[0] = (float)face;
        return 1;
    }
    case RT_GSPHERE:
        return gsphere_test(&s->gsph[pr->index], ray, t_out);
    case RT_MEDIUM:
        return medium_test(s, prim, &s->media[pr->index], ray, t_out);
    default: return 0;
    }
}

/* hit point, normal, (u,v), error bound of a full translated sphere.  Same
 * parameterisation as the general sphere with identity rotation
 * (sphere.rs:70-79,110-113), normal taken as p / r. */
static void sphere_surface(v3 oc, v3 d, v3 c, float r, float t, orc_surf *out) {
    float ar = fabsf(r);
    v3 p = v3_add(oc, v3_scale(d, t));
    p = v3_scale(p, ar / v3_len(p));
    if (p.x == 0.0f && p.y == 0.0f) p.x = 1e-5f * ar;
    float phi = orc_atan2(p.y, p.x);
    if (phi < 0.0f) phi += TWO_PI;
    float theta = orc_acos(orc_clamp(p.z / ar, -1.0f, 1.0f));
    out->u = phi / TWO_PI;
    out->v = (theta - PI_F) / (0.0f - PI_F);
    out->n = v3_div(p, r);
    out->p = v3_add(p, c);
    v3 ap = v3_abs(p);
    v3 e = v3_scale(ap, G5);
    out->perr = v3_add(v3_scale(e, G3 + 1.0f), v3_scale(v3_add(ap, v3_abs(c)), G3));
}

static void to_world(const orc_scene *s, uint32_t xform, v3 p, v3 e, v3 n, orc_surf *out) {
    if (xform) {
        const rt_xform *x = &s->xforms[xform - 1];
        out->p = xf_point_err(x->o2w, p, e, &out->perr);
        out->n = v3_norm(xf_normal(x->w2o, n));
    } else {
        out->p = p;
        out->perr = e;
        out->n = n;
    }
}

void orc_prim_finalize(const orc_scene *s, uint32_t prim, const rt_ray *ray, float t,
                       const float aux[4], orc_surf *out) {
    const rt_prim_ref *pr = &s->prims[prim];
    uint32_t kind = pr->kind & 0xffu;
    int reverse = (pr->kind & RT_PRIM_REVERSE) != 0;
    out->t = t;
    out->prim = prim;
    out->flags = kind;
    out->u = 0.0f;
    out->v = 0.0f;
    v3 o = ld3(ray->o), d = ld3(ray->d);
    switch (kind) {
    case RT_SPHERE: {
        const rt_sphere *sp = &s->spheres[pr->index];
        v3 c = ld3(sp->c);
        sphere_surface(v3_sub(o, c), d, c, sp->r, t, out);
        break;
    }
    case RT_MOVING_SPHERE: {
        const orc_msphere *m = &s->msph[pr->index];
        v3 c = msphere_centre(m, ray->time);
        sphere_surface(v3_sub(o, c), d, c, m->r, t, out);
        break;
    }
    case RT_TRIANGLE: { /* triangle.rs:204-223,253-255,275-345 */
        const float *v = s->tri_verts + 9 * (size_t)pr->index;
        v3 p0 = ld3(v), p1 = ld3(v + 3), p2 = ld3(v + 6);
        float b0 = aux[0], b1 = aux[1], b2 = aux[2];
        out->p = v3_make(p0.x * b0 + p1.x * b1 + p2.x * b2, p0.y * b0 + p1.y * b1 + p2.y * b2,
                         p0.z * b0 + p1.z * b1 + p2.z * b2);
        out->perr = v3_scale(v3_make(fabsf(p0.x * b0) + fabsf(p1.x * b1) + fabsf(p2.x * b2),
                                     fabsf(p0.y * b0) + fabsf(p1.y * b1) + fabsf(p2.y * b2),
                                     fabsf(p0.z * b0) + fabsf(p1.z * b1) + fabsf(p2.z * b2)),
                             G7);
        if (s->tri_uvs) {
            const float *uv = s->tri_uvs + 6 * (size_t)pr->index;
            out->u = uv[0] * b0 + uv[2] * b1 + uv[4] * b2;
            out->v = uv[1] * b0 + uv[3] * b1 + uv[5] * b2;
        } else { /* default (0,0) (1,0) (0,1), triangle.rs:72-78 */
            out->u = 0.0f * b0 + 1.0f * b1 + 0.0f * b2;
            out->v = 0.0f * b0 + 0.0f * b1 + 1.0f * b2;
        }
        v3 n = v3_norm(v3_cross(v3_sub(p0, p2), v3_sub(p1, p2)));
        if (s->tri_normals) {
            const float *nn = s->tri_normals + 9 * (size_t)pr->index;
            v3 ns = v3_norm(v3_make(nn[0] * b0 + nn[3] * b1 + nn[6] * b2,
                                    nn[1] * b0 + nn[4] * b1 + nn[7] * b2,
                                    nn[2] * b0 + nn[5] * b1 + nn[8] * b2));
            if (v3_dot(n, ns) < 0.0f) n = v3_neg(n);
        } else if (reverse) {
            n = v3_neg(n);
        }
        out->n = n;
        break;
    }
    case RT_RECT: {
        const rt_rect *r = &s->rects[pr->index];
        v3 oo, dd;
        int A, B;
        to_object(s, pr->xform, ray, &oo, &dd);
        rect_axes(r->axis, &A, &B);
        float p[3], e[3], n[3] = {0.0f, 0.0f, 0.0f};
        float ta = t * v3_get(dd, A), tb = t * v3_get(dd, B);
        p[A] = v3_get(oo, A) + ta;
        p[B] = v3_get(oo, B) + tb;
        p[r->axis] = r->k;
        e[A] = (fabsf(v3_get(oo, A)) + fabsf(ta)) * G5;
        e[B] = (fabsf(v3_get(oo, B)) + fabsf(tb)) * G5;
        e[r->axis] = fabsf(r->k) * G5;
        n[r->axis] = reverse ? -1.0f : 1.0f;
        out->u = (p[A] - r->a0) / (r->a1 - r->a0);
        out->v = (p[B] - r->b0) / (r->b1 - r->b0);
        out->flags |= r->axis << 8;
        to_world(s, pr->xform, ld3(p), ld3(e), ld3(n), out);
        break;
    }
    case RT_BOX: {
        const rt_box *bx = &s->boxes[pr->index];
        v3 oo, dd;
        to_object(s, pr->xform, ray, &oo, &dd);
        int face = (int)aux[0];
        int ax = face >> 1, A, B;
        rect_axes((uint32_t)ax, &A, &B);
        float p[3], e[3], n[3] = {0.0f, 0.0f, 0.0f};
        float ta = t * v3_get(dd, A), tb = t * v3_get(dd, B);
        p[A] = v3_get(oo, A) + ta;
        p[B] = v3_get(oo, B) + tb;
        p[ax] = (face & 1) ? bx->hi[ax] : bx->lo[ax];
        e[A] = (fabsf(v3_get(oo, A)) + fabsf(ta)) * G5;
        e[B] = (fabsf(v3_get(oo, B)) + fabsf(tb)) * G5;
        e[ax] = fabsf(p[ax]) * G5;
        n[ax] = (face & 1) ? 1.0f : -1.0f;
        out->u = (p[A] - bx->lo[A]) / (bx->hi[A] - bx->lo[A]);
        out->v = (p[B] - bx->lo[B]) / (bx->hi[B] - bx->lo[B]);
        out->flags |= (uint32_t)face << 8;
        to_world(s, pr->xform, ld3(p), ld3(e), ld3(n), out);
        break;
    }
    case RT_GSPHERE: { /* sphere.rs:110-154, interaction.rs:138-177,243-257 */
        const orc_gsphere *g = &s->gsph[pr->index];
        v3 go, gd;
        float phi;
        gsphere_ray(g, ray, &go, &gd);
        v3 p = gsphere_point(g, go, gd, t, &phi);
        out->u = phi / g->phi_max;
        float theta = orc_acos(orc_clamp(p.z / g->radius, -1.0f, 1.0f));
        out->v = (theta - g->theta_min) / (g->theta_max - g->theta_min);
        float z_radius = sqrtf(p.x * p.x + p.y * p.y);
        float inv_zr = 1.0f / z_radius;
        float cos_phi = p.x * inv_zr, sin_phi = p.y * inv_zr;
        v3 dpdu = v3_make(-g->phi_max * p.y, g->phi_max * p.x, 0.0f);
        v3 dpdv = v3_scale(v3_make(p.z * cos_phi, p.z * sin_phi, -g->radius * orc_sin(theta)),
                           g->theta_max - g->theta_min);
        v3 n = v3_norm(v3_cross(dpdu, dpdv));
        if (reverse ^ g->flip) n = v3_neg(n);
        v3 e = v3_scale(v3_abs(p), G5);
        out->p = xf_point_err(g->o2w, p, e, &out->perr);
        out->n = v3_norm(xf_normal(g->w2o, n));
        break;
    }
    case RT_MEDIUM:
    default:
        out->p = v3_add(o, v3_scale(d, t));
        out->perr = v3_make(0.0f, 0.0f, 0.0f);
        out->n = v3_make(1.0f, 0.0f, 0.0f);
        break;
    }
}
