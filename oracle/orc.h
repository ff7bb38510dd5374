/*
 * orc.h -- API of the CPU oracle (liborc.so).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker; never by the shipped library.
 * "Parity unpinned" for traversal/intersection (the reference has no vectors
 * for them); see orc_math.h for what is pinned.
 *
 * The data-layout structs (rays, hits, scene description, camera, options)
 * are the public ones of include/rt355.h so that one set of inputs feeds both
 * the oracle and the HIP library.
 */
#ifndef ORC_H
#define ORC_H

#include "../include/rt355.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

int orc_scene_create(const rt_scene_desc *desc, const rt_build_opts *opts, orc_scene **out);
void orc_scene_destroy(orc_scene *s);
int orc_scene_world_bound(const orc_scene *s, float out[6]);
int orc_scene_get_info(const orc_scene *s, rt_scene_info *out);
int64_t orc_scene_get_nodes(const orc_scene *s, rt_linear_node *out, size_t cap);
int64_t orc_scene_get_prim_order(const orc_scene *s, uint32_t *out, size_t cap);
/* Replace the tree by an externally built one (used to check the GPU
 * library's builder output with the oracle's traversal and vice versa). */
int orc_scene_set_tree(orc_scene *s, const rt_linear_node *nodes, size_t n_nodes,
                       const uint32_t *prim_order, size_t n_prims);

/* counters: [0] nodes visited, [1 + kind] primitives tested; may be NULL */
int orc_intersect(const orc_scene *s, const rt_ray *rays, size_t n, rt_hit *hits, int threads,
                  uint64_t counters[9]);
int orc_occluded(const orc_scene *s, const rt_ray *rays, size_t n, uint8_t *out, int threads);
/* exhaustive loop over the primitive list in caller order, no tree */
int orc_intersect_brute(const orc_scene *s, const rt_ray *rays, size_t n, rt_hit *hits, int threads);

int orc_generate_rays(const rt_camera *cam, const rt_render_opts *opts, uint64_t first_path,
                      size_t n, rt_ray *rays);
/* Film::new's filter table (colour/src/film.rs:99-113) and write_image's
 * normalisation (film.rs:236-247). */
void orc_camera_sample_test(const rt_render_opts *opts, uint32_t px, uint32_t py, uint32_t s, float u[5]);
int orc_filter_table(const rt_filter *filter, float out[256]);
/* the Sobol' tables the camera sampler uses for a width x height frame (data/src/sobol.rs) */
void orc_sobol_tables(uint32_t width, uint32_t height, uint32_t m32[5 * 52], uint64_t vdc[52], uint64_t inv[52]);
int orc_sampler_check(const rt_render_opts *opts);
void orc_film_resolve(float *rgba, uint32_t width, uint32_t height);
int orc_render(const orc_scene *s, const rt_camera *cam, const rt_render_opts *opts,
               float *rgba_out, rt_stats *stats, int threads);
/* radiance of individual paths (3 floats each) for path ids [first, first+n) */
int orc_path_radiance(const orc_scene *s, const rt_camera *cam, const rt_render_opts *opts,
                      uint64_t first_path, size_t n, float *rgb, uint32_t *segments);

/* scalar helpers exported for known-answer tests */
void orc_bounds_union(const float a[6], const float b[6], float out[6]);
float orc_bounds_distance_squared(const float b[6], const float p[3]);
float orc_bounds_surface_area(const float b[6]);
int orc_slab_test(const float box[6], const rt_ray *ray);
void orc_pcg32_stream(uint64_t seq, uint32_t *out, size_t n);
void orc_pcg32_floats(uint64_t seq, float *out, size_t n);
float orc_math_eval(int fn, float a, float b); /* 0 sin 1 cos 2 log 3 atan2 4 acos 5 gamma(a) */
void orc_offset_ray_origin(const float p[3], const float err[3], const float n[3], const float w[3],
                           float out[3]);
/* Interaction::spawn_ray_to / spawn_ray_intersect, shapes/src/interaction.rs:102-133 */
void orc_spawn_ray_to_test(const float p[3], const float err[3], const float n[3], const float target[3],
                           float time, rt_ray *out);
const char *orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
