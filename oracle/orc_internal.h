/* orc_internal.h -- private declarations of the CPU oracle (test infrastructure only). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "orc.h"
#include "orc_math.h"

typedef struct { v3 lo, hi; } aabb;

#define ORC_MAX_ENCLOSING 4
#define ORC_MAX_BALLS 4

/* A medium bounded by a sphere that nothing else reaches into, with or without a sphere
 * primitive of exactly that geometry around it (see find_balls in orc_bvh.c). */
typedef struct {
    float c[3], r;
    float r_in2;      /* a ray starts "inside" when |o - c|^2 < r_in2 */
    uint32_t n;       /* primitives of the ball: 1 (the medium) or 2 */
    uint32_t prim[2]; /* in the order the tree's leaves hold them */
    int sphere;       /* which of them is the sphere primitive, or -1 */
} orc_ball;

/* Bounds3::ZERO is the inverted box [+MAX, -MAX], geometry/src/bounds3.rs:23-26 */
static inline aabb aabb_empty(void) {
    aabb b = {{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f},
              {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f}};
    return b;
}
static inline float fmin_sel(float a, float b) { return b < a ? b : a; }
static inline float fmax_sel(float a, float b) { return b > a ? b : a; }
/* union_box / union_point, bounds3.rs:62-75 */
static inline aabb aabb_union(aabb a, aabb b) {
    aabb r;
    r.lo = v3_make(fmin_sel(a.lo.x, b.lo.x), fmin_sel(a.lo.y, b.lo.y), fmin_sel(a.lo.z, b.lo.z));
    r.hi = v3_make(fmax_sel(a.hi.x, b.hi.x), fmax_sel(a.hi.y, b.hi.y), fmax_sel(a.hi.z, b.hi.z));
    return r;
}
static inline aabb aabb_union_pt(aabb a, v3 p) {
    aabb b = {p, p};
    return aabb_union(a, b);
}

/* derived, ready-to-intersect primitive records */
typedef struct {
    float o2w[12], w2o[12];
    float radius, z_min, z_max, theta_min, theta_max, phi_max;
    int flip; /* reverse_orientation ^ swaps_handedness */
} orc_gsphere;

typedef struct {
    float c0[3], r, dcdt[3], time0;
    float c1[3], time1; /* bounds only */
} orc_msphere;

struct orc_scene {
    /* copied description */
    uint32_t n_prims, n_spheres, n_msph, n_tris, n_rects, n_boxes, n_gsph, n_xforms, n_media;
    uint32_t n_materials, n_textures, n_images;
    rt_prim_ref *prims;
    rt_sphere *spheres;
    orc_msphere *msph;
    float *tri_verts, *tri_uvs, *tri_normals;
    rt_rect *rects;
    rt_box *boxes;
    orc_gsphere *gsph;
    rt_xform *xforms;
    rt_medium *media;
    rt_material *materials;
    rt_texture *textures;
    rt_image *images; /* pixel data copied */
    float *perlin_vec;
    int32_t *perlin_perm;
    float background[3];
    /* lights of RT_RENDER_DIRECT_LIGHT: rectangles without instance transform whose material is a
     * DiffuseLight with a solid colour, in primitive order */
    uint32_t n_lights, *lights;
    uint8_t *is_light; /* per primitive: in `lights` */
    /* distant lights (rt_distant_light) follow the rectangles in the uniform choice of a light;
     * distant_reach: twice the radius of the sphere around the scene's box (pbrt's distant light
     * aims its shadow ray at the point that far away in the light's direction) */
    uint32_t n_distant;
    rt_distant_light *distant;
    float distant_reach;
    /* Media that enclose the scene are kept out of the tree (see orc_build): every ray meets
     * them, so they are tested once per ray, before the traversal, in this order. */
    uint32_t n_enclosing, enclosing[ORC_MAX_ENCLOSING];
    uint32_t n_tree_prims; /* n_prims - n_enclosing */
    uint32_t n_balls;
    orc_ball balls[ORC_MAX_BALLS];
    /* tree over the other primitives */
    rt_linear_node *nodes;
    uint32_t n_nodes;
    uint32_t *order; /* ordered primitive ids (leaf ranges index this), n_tree_prims entries */
    uint32_t max_depth;
    double build_ms;
};

/* full description of an accepted hit, enough to shade and to spawn the next ray */
typedef struct {
    float t;
    uint32_t prim;
    float u, v;
    v3 n;    /* geometric normal as reported in rt_hit */
    v3 p;    /* world-space hit point */
    v3 perr; /* conservative absolute error of p */
    uint32_t flags;
} orc_surf;

void orc_set_error(const char *fmt, ...);

/* orc_shapes.c */
aabb orc_prim_bounds(const orc_scene *s, uint32_t prim);
/* Returns 1 and the accepted parametric distance when ray hits prim with
 * 0 < t <= ray->t_max.  `aux` receives per-kind data finalize needs. */
int orc_prim_test(const orc_scene *s, uint32_t prim, const rt_ray *ray, float *t_out, float aux[4]);
void orc_prim_finalize(const orc_scene *s, uint32_t prim, const rt_ray *ray, float t,
                       const float aux[4], orc_surf *out);
void orc_derive_gsphere(const rt_gsphere *in, orc_gsphere *out);

/* orc_bvh.c */
int orc_build(orc_scene *s, const rt_build_opts *opts);
int orc_slab(const rt_linear_node *n, const rt_ray *ray, const float inv[3], const int neg[3]);
int orc_closest(const orc_scene *s, const rt_ray *ray, orc_surf *out, uint64_t counters[9]);
int orc_any(const orc_scene *s, const rt_ray *ray);
uint32_t orc_tree_depth(const rt_linear_node *nodes, uint32_t n_nodes);

/* orc_render.c */
void orc_camera_ray(const rt_camera *cam, float fx, float fy, float lu, float lv, float tu,
                    rt_ray *out);
/* Camera sample of path (pixel, s): film offsets u[0], u[1], time u[2], lens u[3], u[4]. */
/* frees what the calling thread cached (per-pixel samples) */
void orc_thread_release(void);
void orc_camera_sample(const rt_render_opts *opts, uint32_t px, uint32_t py, uint32_t s, orc_rng *rng, float u[5]);
void orc_path_setup(const rt_camera *cam, const rt_render_opts *opts, uint64_t path, orc_rng *rng,
                    rt_ray *ray);
/* Returns the closest-hit segments of the path; *occlusion receives its shadow rays. */
uint32_t orc_trace_path(const orc_scene *s, const rt_camera *cam, const rt_render_opts *opts,
                        uint64_t path, float rgb[3], uint64_t counters[9], uint32_t *occlusion);
v3 orc_offset_origin(v3 p, v3 err, v3 n, v3 w);
void orc_spawn_ray_to(v3 p, v3 err, v3 n, v3 target, float time, rt_ray *out);

#endif
