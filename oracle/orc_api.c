/*
 * orc_api.c -- entry points of the CPU oracle (test infrastructure only; see
 * orc_math.h).  Batches are split over POSIX threads in contiguous blocks;
 * results do not depend on the thread count.
 */
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "orc_internal.h"

static __thread char g_err[256];
void orc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *orc_last_error(void) { return g_err; }

static void *dup_mem(const void *src, size_t bytes) {
    if (!src || bytes == 0) return NULL;
    void *p = malloc(bytes);
    if (p) memcpy(p, src, bytes);
    return p;
}

static int validate(const rt_scene_desc *d) {
    if (d->n_distant_lights && !d->distant_lights) return 0;
    for (uint32_t i = 0; i < d->n_distant_lights; i++) {
        const float *w = d->distant_lights[i].w;
        const float len2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
        if (!(len2 > 0.99f && len2 < 1.01f)) return 0;
    }
    if (d->n_prims && !d->prims) return 0;
    for (uint32_t i = 0; i < d->n_prims; i++) {
        const rt_prim_ref *p = &d->prims[i];
        uint32_t kind = p->kind & 0xffu, lim = 0;
        switch (kind) {
        case RT_SPHERE: lim = d->n_spheres; break;
        case RT_MOVING_SPHERE: lim = d->n_moving_spheres; break;
        case RT_TRIANGLE: lim = d->n_tris; break;
        case RT_RECT: lim = d->n_rects; break;
        case RT_BOX: lim = d->n_boxes; break;
        case RT_GSPHERE: lim = d->n_gspheres; break;
        case RT_MEDIUM: lim = d->n_media; break;
        default: return 0;
        }
        if (p->index >= lim) return 0;
        if (p->xform > d->n_xforms) return 0;
        if (d->n_materials && p->material >= d->n_materials) return 0;
        if (kind == RT_MEDIUM) {
            const rt_medium *m = &d->media[p->index];
            if (m->boundary_kind == RT_BOUNDARY_SPHERE ? m->boundary_index >= d->n_spheres
                                                       : m->boundary_index >= d->n_boxes)
                return 0;
            if (m->xform > d->n_xforms) return 0;
        }
    }
    for (uint32_t i = 0; i < d->n_materials; i++)
        if (d->n_textures && d->materials[i].texture >= d->n_textures) return 0;
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rt_texture *t = &d->textures[i];
        if ((t->kind == RT_TEX_CHECKER || t->kind == RT_TEX_CHECKER_UV) && (t->a >= d->n_textures || t->b >= d->n_textures)) return 0;
        if (t->kind == RT_TEX_IMAGE && t->a >= d->n_images) return 0;
        if (t->kind == RT_TEX_NOISE && (!d->perlin_vec || !d->perlin_perm)) return 0;
    }
    return 1;
}

int orc_scene_create(const rt_scene_desc *d, const rt_build_opts *opts, orc_scene **out) {
    if (!d || !out) { orc_set_error("null argument"); return RT_ERR_INVALID; }
    if (!validate(d)) { orc_set_error("malformed scene description"); return RT_ERR_INVALID; }
    orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
    if (!s) return RT_ERR_NOMEM;
    s->n_prims = d->n_prims; s->n_spheres = d->n_spheres; s->n_msph = d->n_moving_spheres;
    s->n_tris = d->n_tris; s->n_rects = d->n_rects; s->n_boxes = d->n_boxes;
    s->n_gsph = d->n_gspheres; s->n_xforms = d->n_xforms; s->n_media = d->n_media;
    s->n_materials = d->n_materials; s->n_textures = d->n_textures; s->n_images = d->n_images;
    s->prims = dup_mem(d->prims, sizeof(rt_prim_ref) * (size_t)d->n_prims);
    s->spheres = dup_mem(d->spheres, sizeof(rt_sphere) * (size_t)d->n_spheres);
    s->tri_verts = dup_mem(d->tri_verts, sizeof(float) * 9 * (size_t)d->n_tris);
    s->tri_uvs = dup_mem(d->tri_uvs, sizeof(float) * 6 * (size_t)d->n_tris);
    s->tri_normals = dup_mem(d->tri_normals, sizeof(float) * 9 * (size_t)d->n_tris);
    s->rects = dup_mem(d->rects, sizeof(rt_rect) * (size_t)d->n_rects);
    s->boxes = dup_mem(d->boxes, sizeof(rt_box) * (size_t)d->n_boxes);
    s->xforms = dup_mem(d->xforms, sizeof(rt_xform) * (size_t)d->n_xforms);
    s->media = dup_mem(d->media, sizeof(rt_medium) * (size_t)d->n_media);
    s->materials = dup_mem(d->materials, sizeof(rt_material) * (size_t)d->n_materials);
    s->textures = dup_mem(d->textures, sizeof(rt_texture) * (size_t)d->n_textures);
    s->perlin_vec = dup_mem(d->perlin_vec, sizeof(float) * 768);
    s->perlin_perm = dup_mem(d->perlin_perm, sizeof(int32_t) * 768);
    memcpy(s->background, d->background, sizeof s->background);
    if (d->n_moving_spheres) {
        s->msph = (orc_msphere *)calloc(d->n_moving_spheres, sizeof *s->msph);
        for (uint32_t i = 0; i < d->n_moving_spheres; i++) {
            const rt_moving_sphere *m = &d->moving_spheres[i];
            orc_msphere *o = &s->msph[i];
            float dt = m->time1 - m->time0;
            for (int k = 0; k < 3; k++) {
                o->c0[k] = m->c0[k];
                o->c1[k] = m->c1[k];
                o->dcdt[k] = (m->c1[k] - m->c0[k]) / dt;
            }
            o->r = m->r;
            o->time0 = m->time0;
            o->time1 = m->time1;
        }
    }
    if (d->n_gspheres) {
        s->gsph = (orc_gsphere *)calloc(d->n_gspheres, sizeof *s->gsph);
        for (uint32_t i = 0; i < d->n_gspheres; i++) orc_derive_gsphere(&d->gspheres[i], &s->gsph[i]);
    }
    if (d->n_images) {
        s->images = (rt_image *)calloc(d->n_images, sizeof *s->images);
        for (uint32_t i = 0; i < d->n_images; i++) {
            s->images[i] = d->images[i];
            s->images[i].rgb8 = dup_mem(d->images[i].rgb8, (size_t)d->images[i].w * d->images[i].h * 3);
        }
    }
    /* lights for direct lighting: plain rectangles with a solid-colour DiffuseLight */
    s->lights = (uint32_t *)calloc(d->n_prims ? d->n_prims : 1, sizeof *s->lights);
    s->is_light = (uint8_t *)calloc(d->n_prims ? d->n_prims : 1, 1);
    for (uint32_t i = 0; i < d->n_prims && d->n_materials && d->n_textures; i++) {
        const rt_prim_ref *p = &d->prims[i];
        if ((p->kind & 0xffu) != RT_RECT || p->xform) continue;
        const rt_material *m = &d->materials[p->material];
        if (m->kind == RT_MAT_DIFFUSE_LIGHT && d->textures[m->texture].kind == RT_TEX_SOLID) {
            s->lights[s->n_lights++] = i;
            s->is_light[i] = 1;
        }
    }
    s->n_distant = d->n_distant_lights;
    s->distant = dup_mem(d->distant_lights, sizeof(rt_distant_light) * (size_t)d->n_distant_lights);
    int rc = orc_build(s, opts);
    if (rc != RT_OK) { orc_scene_destroy(s); return rc; }
    if (s->n_distant && (s->n_nodes || s->n_enclosing)) {
        float wb[6];
        orc_scene_world_bound(s, wb);
        float hx = wb[3] - (wb[0] + wb[3]) * 0.5f, hy = wb[4] - (wb[1] + wb[4]) * 0.5f, hz = wb[5] - (wb[2] + wb[5]) * 0.5f;
        s->distant_reach = 2.0f * sqrtf((hx * hx + hy * hy) + hz * hz);
    }
    *out = s;
    return RT_OK;
}

void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    if (s->images)
        for (uint32_t i = 0; i < s->n_images; i++) free((void *)s->images[i].rgb8);
    free(s->prims); free(s->spheres); free(s->msph); free(s->tri_verts); free(s->tri_uvs);
    free(s->tri_normals); free(s->rects); free(s->boxes); free(s->gsph); free(s->xforms);
    free(s->media); free(s->materials); free(s->textures); free(s->images); free(s->perlin_vec);
    free(s->perlin_perm); free(s->nodes); free(s->order); free(s->lights); free(s->is_light); free(s->distant);
    free(s);
}

int orc_scene_world_bound(const orc_scene *s, float out[6]) { /* bvh.rs:728-734 */
    if (!s || !out) return RT_ERR_INVALID;
    aabb b = aabb_empty();
    if (s->n_nodes) {
        b.lo = v3_make(s->nodes[0].lo[0], s->nodes[0].lo[1], s->nodes[0].lo[2]);
        b.hi = v3_make(s->nodes[0].hi[0], s->nodes[0].hi[1], s->nodes[0].hi[2]);
    }
    for (uint32_t e = 0; e < s->n_enclosing; e++) /* the media kept out of the tree belong to the scene */
        b = aabb_union(b, orc_prim_bounds(s, s->enclosing[e]));
    out[0] = b.lo.x; out[1] = b.lo.y; out[2] = b.lo.z;
    out[3] = b.hi.x; out[4] = b.hi.y; out[5] = b.hi.z;
    return RT_OK;
}

int orc_scene_get_info(const orc_scene *s, rt_scene_info *out) {
    if (!s || !out) return RT_ERR_INVALID;
    memset(out, 0, sizeof *out);
    out->n_prims = s->n_prims;
    out->n_nodes = s->n_nodes;
    out->max_depth = s->max_depth;
    out->stack_slots = 64;
    out->build_ms = s->build_ms;
    out->n_enclosing = s->n_enclosing;
    for (uint32_t k = 0; k < s->n_enclosing; k++) out->enclosing[k] = s->enclosing[k];
    out->n_balls = s->n_balls;
    return RT_OK;
}

int64_t orc_scene_get_nodes(const orc_scene *s, rt_linear_node *out, size_t cap) {
    if (!s) return RT_ERR_INVALID;
    size_t n = s->n_nodes < cap ? s->n_nodes : cap;
    if (out && n) memcpy(out, s->nodes, n * sizeof *out);
    return (int64_t)s->n_nodes;
}
int64_t orc_scene_get_prim_order(const orc_scene *s, uint32_t *out, size_t cap) {
    if (!s) return RT_ERR_INVALID;
    size_t n = s->n_tree_prims < cap ? s->n_tree_prims : cap;
    if (out && n && s->order) memcpy(out, s->order, n * sizeof *out);
    return (int64_t)s->n_tree_prims;
}
int orc_scene_set_tree(orc_scene *s, const rt_linear_node *nodes, size_t n_nodes,
                       const uint32_t *order, size_t n_prims) {
    if (!s || n_prims != s->n_tree_prims) return RT_ERR_INVALID;
    free(s->nodes);
    free(s->order);
    s->nodes = dup_mem(nodes, n_nodes * sizeof *nodes);
    s->order = dup_mem(order, n_prims * sizeof *order);
    s->n_nodes = (uint32_t)n_nodes;
    s->max_depth = orc_tree_depth(s->nodes, s->n_nodes);
    return s->max_depth > 64 ? RT_ERR_DEPTH : RT_OK;
}

/* ---- threaded batches --------------------------------------------------------- */
typedef struct {
    int mode; /* 0 closest, 1 any, 2 brute, 3 render paths */
    const orc_scene *s;
    const rt_ray *rays;
    rt_hit *hits;
    uint8_t *occ;
    size_t begin, end;
    uint64_t counters[9];
    int want_counters;
    /* render */
    const rt_camera *cam;
    const rt_render_opts *opts;
    float *rad; /* 3 floats per path, indexed from first */
    uint32_t *segs;
    uint64_t first;
    uint64_t segments, occlusion;
} job;

static void surf_to_hit(const orc_surf *h, rt_hit *o) {
    o->t = h->t; o->prim = h->prim; o->u = h->u; o->v = h->v;
    o->n[0] = h->n.x; o->n[1] = h->n.y; o->n[2] = h->n.z;
    o->flags = h->flags;
}
static void miss_hit(rt_hit *o) {
    memset(o, 0, sizeof *o);
    o->prim = RT_MISS;
    o->t = INFINITY;
}

static void *run_job(void *arg) {
    job *j = (job *)arg;
    const orc_scene *s = j->s;
    for (size_t i = j->begin; i < j->end; i++) {
        if (j->mode == 0) {
            orc_surf h;
            if (orc_closest(s, &j->rays[i], &h, j->want_counters ? j->counters : NULL))
                surf_to_hit(&h, &j->hits[i]);
            else
                miss_hit(&j->hits[i]);
        } else if (j->mode == 1) {
            j->occ[i] = (uint8_t)orc_any(s, &j->rays[i]);
        } else if (j->mode == 2) {
            rt_ray ray = j->rays[i];
            int found = 0;
            uint32_t best = 0;
            float best_aux[4] = {0, 0, 0, 0};
            for (uint32_t p = 0; p < s->n_prims; p++) {
                float t, aux[4] = {0, 0, 0, 0};
                if (orc_prim_test(s, p, &ray, &t, aux)) {
                    ray.t_max = t; found = 1; best = p;
                    memcpy(best_aux, aux, sizeof aux);
                }
            }
            if (found) {
                orc_surf h;
                orc_prim_finalize(s, best, &j->rays[i], ray.t_max, best_aux, &h);
                surf_to_hit(&h, &j->hits[i]);
            } else {
                miss_hit(&j->hits[i]);
            }
        } else {
            uint32_t occ = 0;
            uint32_t n = orc_trace_path(s, j->cam, j->opts, j->first + i, j->rad + 3 * i,
                                        j->want_counters ? j->counters : NULL, &occ);
            if (j->segs) j->segs[i] = n + occ;
            j->segments += n;
            j->occlusion += occ;
        }
    }
    orc_thread_release();
    return NULL;
}

static int run_jobs(job *proto, size_t n, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    job *jobs = (job *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof *tid);
    if (!jobs || !tid) { free(jobs); free(tid); return RT_ERR_NOMEM; }
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; t++) {
        jobs[t] = *proto;
        jobs[t].begin = (size_t)t * per < n ? (size_t)t * per : n;
        jobs[t].end = ((size_t)t + 1) * per < n ? ((size_t)t + 1) * per : n;
    }
    for (int t = 1; t < threads; t++) pthread_create(&tid[t], NULL, run_job, &jobs[t]);
    run_job(&jobs[0]);
    for (int t = 1; t < threads; t++) pthread_join(tid[t], NULL);
    for (int t = 0; t < threads; t++) {
        for (int k = 0; k < 9; k++) proto->counters[k] += jobs[t].counters[k];
        proto->segments += jobs[t].segments;
        proto->occlusion += jobs[t].occlusion;
    }
    free(jobs);
    free(tid);
    return RT_OK;
}

int orc_intersect(const orc_scene *s, const rt_ray *rays, size_t n, rt_hit *hits, int threads,
                  uint64_t counters[9]) {
    if (!s || (n && (!rays || !hits))) return RT_ERR_INVALID;
    job j;
    memset(&j, 0, sizeof j);
    j.mode = 0; j.s = s; j.rays = rays; j.hits = hits; j.want_counters = counters != NULL;
    int rc = run_jobs(&j, n, threads);
    if (counters) memcpy(counters, j.counters, sizeof j.counters);
    return rc;
}
int orc_occluded(const orc_scene *s, const rt_ray *rays, size_t n, uint8_t *out, int threads) {
    if (!s || (n && (!rays || !out))) return RT_ERR_INVALID;
    job j;
    memset(&j, 0, sizeof j);
    j.mode = 1; j.s = s; j.rays = rays; j.occ = out;
    return run_jobs(&j, n, threads);
}
int orc_intersect_brute(const orc_scene *s, const rt_ray *rays, size_t n, rt_hit *hits, int threads) {
    if (!s || (n && (!rays || !hits))) return RT_ERR_INVALID;
    job j;
    memset(&j, 0, sizeof j);
    j.mode = 2; j.s = s; j.rays = rays; j.hits = hits;
    return run_jobs(&j, n, threads);
}

int orc_generate_rays(const rt_camera *cam, const rt_render_opts *opts, uint64_t first_path,
                      size_t n, rt_ray *rays) {
    if (!cam || !opts || !rays || !opts->spp || !opts->width || orc_sampler_check(opts)) return RT_ERR_INVALID;
    for (size_t i = 0; i < n; i++) {
        orc_rng rng;
        orc_path_setup(cam, opts, first_path + i, &rng, &rays[i]);
    }
    return RT_OK;
}

int orc_path_radiance(const orc_scene *s, const rt_camera *cam, const rt_render_opts *opts,
                      uint64_t first_path, size_t n, float *rgb, uint32_t *segments) {
    if (!s || !cam || !opts || !rgb || !opts->spp || orc_sampler_check(opts)) return RT_ERR_INVALID;
    job j;
    memset(&j, 0, sizeof j);
    j.mode = 3; j.s = s; j.cam = cam; j.opts = opts; j.rad = rgb; j.segs = segments;
    j.first = first_path;
    return run_jobs(&j, n, 1);
}

/* 32x32 tiles in row-major tile order; tile t belongs to shard t % count */
static int pixel_owned(const rt_render_opts *o, uint32_t px, uint32_t py) {
    uint32_t count = o->shard_count ? o->shard_count : 1;
    uint32_t tiles_x = (o->width + 31) / 32;
    uint32_t t = (py / 32) * tiles_x + px / 32;
    return t % count == o->shard_index;
}

/* ---- film: colour/src/film.rs + the filters of colour/src/filters ------------------------------
 * Deviations from the literal code, following pbrt-v3 which it transcribes: the table
 * index uses the table WIDTH (16), not its length (film.rs:296-309 multiply by and clamp
 * to 256, which indexes out of the 16 x 16 table); the pixel range of a sample is
 * [ceil(d - r), floor(d + r)] (film.rs:287-288 truncate toward zero instead); Mitchell's
 * outer cubic starts (-B - 6C) x^3 (mitchell.rs:33 multiplies B by 6C instead).  The
 * RGB -> XYZ -> RGB round trip of FilmTile merge and write_image is left out (spectra
 * are out of scope): sums are kept in RGB. */
static float filter_eval(const rt_filter *f, float x, float y) {
    float rx = f->radius[0], ry = f->radius[1];
    switch (f->kind) {
    case RT_FILTER_BOX: return 1.0f; /* box.rs:20-22 */
    case RT_FILTER_TRIANGLE: { /* triangle.rs:18-20 */
        float a = rx - fabsf(x), b = ry - fabsf(y);
        return (a > 0.0f ? a : 0.0f) * (b > 0.0f ? b : 0.0f);
    }
    case RT_FILTER_GAUSSIAN: { /* gaussian.rs:15-32 */
        float ex = expf(-f->a * rx * rx), ey = expf(-f->a * ry * ry);
        float gx = expf(-f->a * x * x) - ex, gy = expf(-f->a * y * y) - ey;
        return (gx > 0.0f ? gx : 0.0f) * (gy > 0.0f ? gy : 0.0f);
    }
    case RT_FILTER_MITCHELL: { /* mitchell.rs:17-50 */
        float B = f->a, Cc = f->b, v[2] = {x * (1.0f / rx), y * (1.0f / ry)}, out[2];
        for (int k = 0; k < 2; k++) {
            float t = fabsf(2.0f * v[k]);
            if (t > 1.0f)
                out[k] = ((-B - 6.0f * Cc) * t * t * t + (6.0f * B + 30.0f * Cc) * t * t +
                          (-12.0f * B - 48.0f * Cc) * t + (8.0f * B + 24.0f * Cc)) * (1.0f / 6.0f);
            else
                out[k] = ((12.0f - 9.0f * B - 6.0f * Cc) * t * t * t +
                          (-18.0f + 12.0f * B + 6.0f * Cc) * t * t + (6.0f - 2.0f * B)) * (1.0f / 6.0f);
        }
        return out[0] * out[1];
    }
    case RT_FILTER_LANCZOS: { /* lanczos_sinc.rs:17-38 */
        const float pi = 3.14159265358979323846f;
        float v[2] = {fabsf(x), fabsf(y)}, r[2] = {rx, ry}, out[2];
        for (int k = 0; k < 2; k++) {
            if (v[k] > r[k]) { out[k] = 0.0f; continue; }
            float a = v[k], b = v[k] / f->a;
            float sa = a < 1e-5f ? 1.0f : sinf(pi * a) / (pi * a);
            float sb = b < 1e-5f ? 1.0f : sinf(pi * b) / (pi * b);
            out[k] = sa * sb;
        }
        return out[0] * out[1];
    }
    default: return 0.0f;
    }
}

int orc_filter_table(const rt_filter *f, float out[256]) {
    if (!f || !out || f->kind < RT_FILTER_BOX || f->kind > RT_FILTER_LANCZOS) return RT_ERR_INVALID;
    if (!(f->radius[0] > 0.0f && f->radius[1] > 0.0f) || f->radius[0] > RT_FILTER_MAX_RADIUS ||
        f->radius[1] > RT_FILTER_MAX_RADIUS)
        return RT_ERR_INVALID;
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) /* film.rs:99-113 */
            out[y * 16 + x] = filter_eval(f, ((float)x + 0.5f) * f->radius[0] / 16.0f,
                                          ((float)y + 0.5f) * f->radius[1] / 16.0f);
    return RT_OK;
}

void orc_film_resolve(float *rgba, uint32_t width, uint32_t height) { /* film.rs:236-247 */
    for (size_t p = 0; p < (size_t)width * height; p++) {
        float w = rgba[4 * p + 3];
        if (w != 0.0f) {
            float inv = 1.0f / w;
            for (int c = 0; c < 3; c++) {
                float v = rgba[4 * p + c] * inv;
                rgba[4 * p + c] = v > 0.0f ? v : 0.0f;
            }
        }
        rgba[4 * p + 3] = 1.0f;
    }
}

/* FilmTile::add_sample (film.rs:281-328) in gather form: pixel (x, y) collects, for sample
 * index s = 0 .. spp-1 in turn, the samples s of the pixels around it row by row.  The
 * order of the additions is part of the definition (fp32 sums). */
static int render_filtered(const orc_scene *s, const rt_camera *cam, const rt_render_opts *opts, float *rgba,
                           rt_stats *stats, int threads) {
    const uint32_t W = opts->width, H = opts->height, spp = opts->spp;
    float table[256];
    int rc = orc_filter_table(&opts->filter, table);
    if (rc != RT_OK) return rc;
    size_t npix = (size_t)W * H;
    float *rad = (float *)calloc(npix * spp * 3, sizeof(float));
    if (!rad) return RT_ERR_NOMEM;
    uint64_t segments = 0, occlusion = 0, paths = 0, counters[9] = {0};
    int want = (opts->flags & RT_RENDER_COUNTERS) != 0;
    for (size_t p = 0; p < npix; p++) {
        if (!pixel_owned(opts, (uint32_t)(p % W), (uint32_t)(p / W))) continue;
        job j;
        memset(&j, 0, sizeof j);
        j.mode = 3; j.s = s; j.cam = cam; j.opts = opts; j.rad = rad + p * spp * 3;
        j.first = (uint64_t)p * spp; j.want_counters = want;
        run_jobs(&j, spp, 1);
        segments += j.segments;
        occlusion += j.occlusion;
        paths += spp;
        for (int k = 0; k < 9; k++) counters[k] += j.counters[k];
    }
    (void)threads;
    const float rx = opts->filter.radius[0], ry = opts->filter.radius[1];
    const float inv_rx = 1.0f / rx, inv_ry = 1.0f / ry;
    const int Rx = (int)ceilf(rx + 0.5f), Ry = (int)ceilf(ry + 0.5f);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (uint32_t k = 0; k < spp; k++)
                for (int qy = (int)y - Ry; qy <= (int)y + Ry; qy++)
                    for (int qx = (int)x - Rx; qx <= (int)x + Rx; qx++) {
                        if (qx < 0 || qy < 0 || qx >= (int)W || qy >= (int)H) continue;
                        if (!pixel_owned(opts, (uint32_t)qx, (uint32_t)qy)) continue;
                        uint64_t path = ((uint64_t)qy * W + (uint64_t)qx) * spp + k;
                        orc_rng rng;
                        float u[5];
                        orc_rng_seq(&rng, (opts->seed << 32) + path);
                        orc_camera_sample(opts, (uint32_t)qx, (uint32_t)qy, k, &rng, u);
                        float dx = ((float)qx + u[0]) - 0.5f, dy = ((float)qy + u[1]) - 0.5f;
                        float fxp = (float)x, fyp = (float)y;
                        if (fxp < ceilf(dx - rx) || fxp > floorf(dx + rx)) continue;
                        if (fyp < ceilf(dy - ry) || fyp > floorf(dy + ry)) continue;
                        int ix = (int)floorf(fabsf((fxp - dx) * inv_rx * 16.0f));
                        int iy = (int)floorf(fabsf((fyp - dy) * inv_ry * 16.0f));
                        if (ix > 15) ix = 15;
                        if (iy > 15) iy = 15;
                        float w = table[iy * 16 + ix];
                        const float *L = rad + path * 3;
                        acc[0] += L[0] * w; acc[1] += L[1] * w; acc[2] += L[2] * w;
                        acc[3] += w;
                    }
            float *o = rgba + 4 * ((size_t)y * W + x);
            o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
        }
    free(rad);
    if (!(opts->flags & RT_RENDER_RAW_FILM)) orc_film_resolve(rgba, W, H);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->paths = paths;
        stats->segments = segments + occlusion;
        stats->occlusion_rays = occlusion;
        stats->nodes_visited = counters[0];
        for (int k = 0; k < 8; k++) stats->prims_tested[k] = counters[1 + k];
    }
    return RT_OK;
}

int orc_render(const orc_scene *s, const rt_camera *cam, const rt_render_opts *opts,
               float *rgba, rt_stats *stats, int threads) {
    if (!s || !cam || !opts || !rgba || !opts->spp || !opts->width || !opts->height || orc_sampler_check(opts))
        return RT_ERR_INVALID;
    if (opts->filter.kind != RT_FILTER_NONE) return render_filtered(s, cam, opts, rgba, stats, threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    size_t npix = (size_t)opts->width * opts->height;
    memset(rgba, 0, npix * 4 * sizeof(float));
    /* rows of owned pixels are rendered in chunks to bound memory */
    size_t chunk_pix = 65536 / opts->spp ? 65536 / opts->spp : 1;
    float *rad = (float *)malloc(chunk_pix * opts->spp * 3 * sizeof(float));
    if (!rad) return RT_ERR_NOMEM;
    uint64_t segments = 0, occlusion = 0, paths = 0, counters[9] = {0};
    int want = (opts->flags & RT_RENDER_COUNTERS) != 0;
    for (size_t p0 = 0; p0 < npix; p0 += chunk_pix) {
        size_t p1 = p0 + chunk_pix < npix ? p0 + chunk_pix : npix;
        /* skip chunks with no owned pixel quickly */
        int any = 0;
        for (size_t p = p0; p < p1 && !any; p++)
            any = pixel_owned(opts, (uint32_t)(p % opts->width), (uint32_t)(p / opts->width));
        if (!any) continue;
        job j;
        memset(&j, 0, sizeof j);
        j.mode = 3; j.s = s; j.cam = cam; j.opts = opts; j.rad = rad;
        j.first = (uint64_t)p0 * opts->spp; j.want_counters = want;
        /* unowned pixels in the chunk are traced too only when sharding is off */
        size_t n = (p1 - p0) * opts->spp;
        if ((opts->shard_count ? opts->shard_count : 1) == 1) {
            run_jobs(&j, n, threads);
            segments += j.segments;
            occlusion += j.occlusion;
            paths += n;
            for (int k = 0; k < 9; k++) counters[k] += j.counters[k];
        } else {
            for (size_t p = p0; p < p1; p++) {
                if (!pixel_owned(opts, (uint32_t)(p % opts->width), (uint32_t)(p / opts->width)))
                    continue;
                job jj = j;
                jj.rad = rad + (p - p0) * opts->spp * 3;
                jj.first = (uint64_t)p * opts->spp;
                jj.segments = 0;
                jj.occlusion = 0;
                memset(jj.counters, 0, sizeof jj.counters);
                run_jobs(&jj, opts->spp, 1);
                segments += jj.segments;
                occlusion += jj.occlusion;
                paths += opts->spp;
                for (int k = 0; k < 9; k++) counters[k] += jj.counters[k];
            }
        }
        for (size_t p = p0; p < p1; p++) {
            if (!pixel_owned(opts, (uint32_t)(p % opts->width), (uint32_t)(p / opts->width)))
                continue;
            const float *r = rad + (p - p0) * opts->spp * 3;
            float acc[3] = {0.0f, 0.0f, 0.0f};
            for (uint32_t k = 0; k < opts->spp; k++) {
                acc[0] += r[3 * k]; acc[1] += r[3 * k + 1]; acc[2] += r[3 * k + 2];
            }
            float inv = (float)opts->spp;
            rgba[4 * p] = acc[0] / inv; rgba[4 * p + 1] = acc[1] / inv;
            rgba[4 * p + 2] = acc[2] / inv; rgba[4 * p + 3] = 1.0f;
        }
    }
    free(rad);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->paths = paths;
        stats->segments = segments + occlusion;
        stats->occlusion_rays = occlusion;
        stats->nodes_visited = counters[0];
        for (int k = 0; k < 8; k++) stats->prims_tested[k] = counters[1 + k];
        stats->total_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    }
    return RT_OK;
}

/* ---- scalar helpers for known-answer tests -------------------------------------- */
void orc_bounds_union(const float a[6], const float b[6], float out[6]) {
    aabb x = {{a[0], a[1], a[2]}, {a[3], a[4], a[5]}}, y = {{b[0], b[1], b[2]}, {b[3], b[4], b[5]}};
    aabb r = aabb_union(x, y);
    out[0] = r.lo.x; out[1] = r.lo.y; out[2] = r.lo.z; out[3] = r.hi.x; out[4] = r.hi.y; out[5] = r.hi.z;
}
float orc_bounds_distance_squared(const float b[6], const float p[3]) { /* bounds3.rs:173-177 */
    float d[3];
    for (int i = 0; i < 3; i++) {
        float lo = b[i] - p[i], hi = p[i] - b[3 + i];
        float m = lo > hi ? lo : hi;
        d[i] = m > 0.0f ? m : 0.0f;
    }
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}
float orc_bounds_surface_area(const float b[6]) {
    float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}
int orc_slab_test(const float box[6], const rt_ray *ray) {
    rt_linear_node n;
    memset(&n, 0, sizeof n);
    memcpy(n.lo, box, 12);
    memcpy(n.hi, box + 3, 12);
    float inv[3] = {1.0f / ray->d[0], 1.0f / ray->d[1], 1.0f / ray->d[2]};
    int neg[3] = {inv[0] < 0.0f, inv[1] < 0.0f, inv[2] < 0.0f};
    return orc_slab(&n, ray, inv, neg);
}
void orc_pcg32_stream(uint64_t seq, uint32_t *out, size_t n) {
    orc_rng r;
    orc_rng_seq(&r, seq);
    for (size_t i = 0; i < n; i++) out[i] = orc_rng_u32(&r);
}
void orc_pcg32_floats(uint64_t seq, float *out, size_t n) {
    orc_rng r;
    orc_rng_seq(&r, seq);
    for (size_t i = 0; i < n; i++) out[i] = orc_rng_f(&r);
}
float orc_math_eval(int fn, float a, float b) {
    switch (fn) {
    case 0: return orc_sin(a);
    case 1: return orc_cos(a);
    case 2: return orc_log(a);
    case 3: return orc_atan2(a, b);
    case 4: return orc_acos(a);
    case 5: return orc_gamma((int)a);
    default: return 0.0f;
    }
}
void orc_spawn_ray_to_test(const float p[3], const float err[3], const float n[3], const float target[3],
                           float time, rt_ray *out) {
    orc_spawn_ray_to(v3_make(p[0], p[1], p[2]), v3_make(err[0], err[1], err[2]), v3_make(n[0], n[1], n[2]),
                     v3_make(target[0], target[1], target[2]), time, out);
}
void orc_offset_ray_origin(const float p[3], const float err[3], const float n[3], const float w[3],
                           float out[3]) {
    v3 r = orc_offset_origin(v3_make(p[0], p[1], p[2]), v3_make(err[0], err[1], err[2]),
                             v3_make(n[0], n[1], n[2]), v3_make(w[0], w[1], w[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
