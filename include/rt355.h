/*
 * rt355.h -- C ABI of librt355.so: BVH construction, closest-hit / any-hit
 * traversal, ray-primitive intersection and the wavefront path-tracing loop,
 * executed by hand-written HIP kernels on AMD MI355X (gfx950).
 *
 * The library is the drop-in for the `Primitive<T>` seam of the reference
 * (shapes/src/primitive.rs:8-27) as implemented by `BVH<T>`
 * (shapes/src/bvh.rs:106-121, 727-863).  A GPU call per ray is meaningless, so
 * every entry point is batch shaped; otherwise argument meaning follows the
 * reference: rays are `Ray<f32>` (geometry/src/ray.rs:7-12), a miss is
 * "None"/"false", the scene is immutable after construction (Send + Sync).
 *
 * All functions return RT_OK (0) or a negative error code and never throw or
 * abort across the boundary; rt_last_error() returns a thread-local message.
 * There is no CPU fallback: every compute entry point needs a HIP device.
 *
 * Threads: a scene may be used from several host threads at once, as the
 * reference's BVH is Send + Sync.  The trace entry points run concurrently (give
 * each caller its own stream for the device-pointer variants; callers sharing a
 * stream are served in call order); rt_render*, rt_intersect_count and the first
 * use of a device take a per-scene lock and run one at a time.  rt_scene_destroy
 * must not overlap any other call on that scene.  Every entry point makes
 * `device` the calling thread's current HIP device.
 */
#ifndef RT355_H
#define RT355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument / malformed scene description      */
#define RT_ERR_NO_DEVICE (-2) /* no usable HIP device, or device index invalid    */
#define RT_ERR_HIP (-3)       /* a HIP runtime call failed (message has details)  */
#define RT_ERR_DEPTH (-4)     /* tree needs more than 64 stack entries (bvh.rs:753)*/
#define RT_ERR_NOMEM (-5)
#define RT_ERR_LIMIT (-6)     /* size above a documented fixed maximum            */

#define RT_MISS 0xFFFFFFFFu

/* Ray<f32>: geometry/src/ray.rs:7-12 (origin, direction, t_max, time) = 32 B. */
typedef struct rt_ray {
    float o[3];
    float d[3];
    float t_max;
    float time;
} rt_ray;

/* Compact form of SurfaceInteraction (shapes/src/interaction.rs:52-60): the
 * parametric distance, the caller's primitive index, surface (u,v) and the
 * geometric normal (`interaction.normal`).  prim == RT_MISS means None. */
typedef struct rt_hit {
    float t;
    uint32_t prim;
    float u, v;
    float n[3];
    uint32_t flags; /* bits 0-7: rt_prim_kind; bits 8-15: face (rect axis / box face) */
} rt_hit;

typedef enum rt_prim_kind {
    RT_SPHERE = 0,        /* full sphere, translation only                        */
    RT_MOVING_SPHERE = 1, /* centre moves linearly with ray.time                  */
    RT_TRIANGLE = 2,      /* world-space triangle (shapes/src/triangle.rs)        */
    RT_RECT = 3,          /* axis-aligned rectangle, optional instance transform  */
    RT_BOX = 4,           /* axis-aligned box, optional instance transform        */
    RT_GSPHERE = 5,       /* reference Sphere: transform + z / phi clipping       */
    RT_MEDIUM = 6         /* constant-density medium inside a sphere or box       */
} rt_prim_kind;

#define RT_PRIM_REVERSE 0x100u /* reverse_orientation (shapes/src/lib.rs:68-70)   */

/* One entry of the caller's primitive list (the Vec<Arc<dyn Primitive>> handed
 * to BVH::new, bvh.rs:108-112).  Its position in the list is the primitive
 * index reported in rt_hit.prim. */
typedef struct rt_prim_ref {
    uint32_t kind;     /* rt_prim_kind | RT_PRIM_REVERSE                           */
    uint32_t index;    /* index into the array of that kind                       */
    uint32_t material; /* index into materials (rendering only)                   */
    uint32_t xform;    /* 0 = none, else 1-based index into xforms (RECT/BOX)      */
} rt_prim_ref;

typedef struct rt_sphere {
    float c[3];
    float r; /* negative radius flips the normal (hollow-glass trick)           */
} rt_sphere;

typedef struct rt_moving_sphere {
    float c0[3];
    float r;
    float c1[3];
    float time0;
    float time1;
    float _pad[3];
} rt_moving_sphere;

/* axis = index of the constant coordinate: 0 -> YzRect (x = k, a = y, b = z),
 * 1 -> XzRect (y = k, a = x, b = z), 2 -> XyRect (z = k, a = x, b = y). */
typedef struct rt_rect {
    uint32_t axis;
    float k;
    float a0, a1, b0, b1;
    uint32_t _pad[2];
} rt_rect;

typedef struct rt_box {
    float lo[3];
    float _p0;
    float hi[3];
    float _p1;
} rt_box;

/* Constructor arguments of the reference's Sphere::new
 * (shapes/src/sphere.rs:28-47).  Matrices are row-major 3x4 affine. */
typedef struct rt_gsphere {
    float o2w[12];
    float w2o[12];
    float radius;
    float z_min, z_max; /* unordered pair as given; clamped like sphere.rs:36-37 */
    float phi_max_deg;
    uint32_t _pad[4];
} rt_gsphere;

typedef struct rt_xform {
    float o2w[12];
    float w2o[12];
} rt_xform;

#define RT_BOUNDARY_SPHERE 0u
#define RT_BOUNDARY_BOX 1u
typedef struct rt_medium {
    uint32_t boundary_kind;  /* RT_BOUNDARY_*                                     */
    uint32_t boundary_index; /* index into spheres / boxes                        */
    uint32_t xform;          /* 0 = none, else 1-based (box boundary only)        */
    float density;
} rt_medium;

typedef enum rt_material_kind {
    RT_MAT_LAMBERTIAN = 0,
    RT_MAT_METALLIC = 1,
    RT_MAT_DIELECTRIC = 2,
    RT_MAT_DIFFUSE_LIGHT = 3,
    RT_MAT_ISOTROPIC = 4
} rt_material_kind;

typedef struct rt_material {
    uint32_t kind;
    uint32_t texture;
    float fuzz;
    float ior;
} rt_material;

typedef enum rt_texture_kind {
    RT_TEX_SOLID = 0,
    RT_TEX_CHECKER = 1, /* a = odd texture, b = even texture                      */
    RT_TEX_IMAGE = 2,   /* a = image index                                        */
    RT_TEX_NOISE = 3,   /* Perlin turbulence marble, `scale`                      */
    RT_TEX_CHECKER_UV = 4 /* (floor(u*scale) + floor(v*scale)) odd -> a, even -> b  */
} rt_texture_kind;

typedef struct rt_texture {
    uint32_t kind;
    float rgb[3];
    uint32_t a, b;
    float scale;
    uint32_t _pad;
} rt_texture;

typedef struct rt_image {
    const uint8_t *rgb8; /* w*h*3 bytes, row 0 = top                              */
    uint32_t w, h;
} rt_image;

/* A light infinitely far away that shines along one direction (test/test.pbrt:16-17,
 * LightSource "distant"; the reference's parser stops before the world section, so there is no
 * code for it there).  RT_RENDER_DIRECT_LIGHT samples it like the emissive rectangles: it is one
 * more light of the uniform choice, and its shadow ray is Interaction::spawn_ray_to
 * (shapes/src/interaction.rs:102-116) aimed at the point twice the scene's bounding radius away
 * in direction w, as pbrt's distant light does.  Without that flag it adds nothing (no path
 * can reach it by scattering). */
typedef struct rt_distant_light {
    float w[3];   /* unit vector from the scene towards the light */
    float rgb[3]; /* radiance arriving from it                    */
    uint32_t _pad[2];
} rt_distant_light;

/* Everything BVH::new + the shape constructors need.  The caller owns all
 * arrays; rt_scene_create copies what it keeps. */
typedef struct rt_scene_desc {
    const rt_prim_ref *prims;
    const rt_sphere *spheres;
    const rt_moving_sphere *moving_spheres;
    const float *tri_verts;   /* 9 floats per triangle, world space             */
    const float *tri_uvs;     /* 6 floats per triangle or NULL (triangle.rs:72-78)*/
    const float *tri_normals; /* 9 floats per triangle or NULL                   */
    const rt_rect *rects;
    const rt_box *boxes;
    const rt_gsphere *gspheres;
    const rt_xform *xforms;
    const rt_medium *media;
    const rt_material *materials;
    const rt_texture *textures;
    const rt_image *images;
    const float *perlin_vec;   /* 256*3 floats or NULL                          */
    const int32_t *perlin_perm; /* 3*256 ints (x, y, z tables) or NULL           */
    uint32_t n_prims, n_spheres, n_moving_spheres, n_tris, n_rects, n_boxes;
    uint32_t n_gspheres, n_xforms, n_media, n_materials, n_textures, n_images;
    float background[3];
    uint32_t n_distant_lights;
    const rt_distant_light *distant_lights; /* NULL when n_distant_lights == 0 */
} rt_scene_desc;

/* SplitMethod, shapes/src/bvh.rs:10-22 (same order). */
typedef enum rt_split_method {
    RT_SPLIT_SAH = 0,
    RT_SPLIT_HLBVH = 1,
    RT_SPLIT_MIDDLE = 2,
    RT_SPLIT_EQUAL_COUNTS = 3
} rt_split_method;

#define RT_BUILD_ON_DEVICE 1u /* HLBVH only: Morton codes, sort and treelets on the GPU */

typedef struct rt_build_opts {
    uint32_t max_prims_in_node; /* clamped to 255 (bvh.rs:114)                    */
    uint32_t split_method;
    uint32_t flags;             /* RT_BUILD_*                                     */
    uint32_t device;            /* HIP device of a RT_BUILD_ON_DEVICE build       */
} rt_build_opts;

/* LinearBVHNode, shapes/src/bvh.rs:79-101, exactly 32 bytes. */
typedef struct rt_linear_node {
    float lo[3];
    float hi[3];
    uint32_t child_offset;    /* leaf: first primitive; interior: second child    */
    uint16_t primitive_count; /* 0 = interior                                     */
    uint8_t axis;
    uint8_t _pad;
} rt_linear_node;

typedef struct rt_scene_info {
    uint32_t n_prims;
    uint32_t n_nodes;
    uint32_t max_depth;   /* maximum number of simultaneously stacked nodes + 1   */
    uint32_t stack_slots; /* stack entries a lane of the kernels can need (max_depth - 1) */
    uint64_t device_bytes;
    double build_ms;
    uint32_t n_enclosing; /* media enclosing the scene, kept out of the tree (see rt_scene_create) */
    uint32_t enclosing[4]; /* their primitive indices, ascending                                  */
    uint32_t n_balls;      /* sphere-bounded media nothing else reaches into (see rt_scene_create) */
} rt_scene_info;

/* Perspective camera: colour/src/camera/perspective.rs:16-78 with
 * projective.rs:28-52 already folded into raster_to_camera by the host. */
typedef struct rt_camera {
    float raster_to_camera[16]; /* row-major 4x4                                 */
    float camera_to_world[12];  /* row-major 3x4                                 */
    float lens_radius;
    float focal_distance;
    float shutter_open;
    float shutter_close;
} rt_camera;

#define RT_RENDER_COUNTERS 1u /* also count nodes visited / primitives tested    */
#define RT_RENDER_TIMING 2u   /* bracket every trace launch with HIP events       */
#define RT_RENDER_RAW_FILM 4u /* with a filter: leave {sum of w*L, sum of w} per pixel
                                 (shards are summed before rt_film_resolve_dev)   */
/* Direct lighting by shadow rays.  At every hit on a Lambertian surface one point q on one
 * light is sampled (lights: rectangles without instance transform whose material is a
 * DiffuseLight with a solid colour), the ray Interaction::spawn_ray_to(q)
 * (shapes/src/interaction.rs:102-116: origin offset by offset_ray_origin towards q, direction
 * q - origin un-normalised, t_max = 1 - 0.0001) goes through the any-hit traversal
 * (BVH::does_intersect, bvh.rs:801-863), and if it is not occluded
 * throughput * albedo / pi * emitted * cos cos' area n_lights / distance^2 is added to the
 * path's radiance; the light a path reaches by its next segment then adds nothing.  Such
 * frames use one launch per bounce plus one any-hit launch per bounce.  A different estimator
 * from the default one (the legacy scenes' scattering_mode = "Sphere"): images differ. */
#define RT_RENDER_DIRECT_LIGHT 8u

/* Reconstruction filters, colour/src/filters/{box,triangle,gaussian,mitchell,
 * lanczos_sinc}.rs.  RT_FILTER_NONE keeps the legacy scenes' behaviour: a pixel
 * is the plain mean of its own samples. */
typedef enum rt_filter_kind {
    RT_FILTER_NONE = 0,
    RT_FILTER_BOX = 1,
    RT_FILTER_TRIANGLE = 2,
    RT_FILTER_GAUSSIAN = 3, /* a = alpha                                           */
    RT_FILTER_MITCHELL = 4, /* a = B, b = C                                        */
    RT_FILTER_LANCZOS = 5   /* a = tau                                             */
} rt_filter_kind;

typedef struct rt_filter {
    uint32_t kind;
    float radius[2]; /* in pixels, each at most RT_FILTER_MAX_RADIUS               */
    float a, b;
    uint32_t _pad;
} rt_filter;

#define RT_FILTER_MAX_RADIUS 8.0f
#define RT_FILTER_TABLE_WIDTH 16 /* FILTER_TABLE_WIDTH, colour/src/film.rs:43      */

/* Where a path's camera sample (film x, film y, time, lens u, lens v) comes from.
 * Scattering always draws from the path's PCG32 stream (geometry/src/rng.rs). */
typedef enum rt_sampler_kind {
    RT_SAMPLER_RANDOM = 0,    /* five PCG32 draws                                  */
    RT_SAMPLER_HALTON = 1,    /* Halton::sample_dimension, colour/src/samplers/halton.rs:
                                 dimensions 0-4 of the Halton sequence, pixel-aligned */
    RT_SAMPLER_SOBOL = 2,     /* Sobol::get_index_for_sample / sample_dimension,
                                 colour/src/samplers/sobol.rs:31-95: dimensions 0-4 of
                                 the Sobol' sequence, pixel-aligned                   */
    RT_SAMPLER_ZERO_TWO = 3,  /* ZeroTwo::start_pixel, colour/src/samplers/zero_two.rs:
                                 scrambled, shuffled (0,2)-sequence points per pixel;
                                 spp must be a power of two (ZeroTwo::new rounds up)  */
    RT_SAMPLER_STRATIFIED = 4 /* Stratified::start_pixel, colour/src/samplers/
                                 stratified.rs: sampler_arg bits 0-15 = x strata,
                                 y strata = spp / x strata (must divide)              */
} rt_sampler_kind;

#define RT_STRATIFIED_NO_JITTER 0x80000000u /* samples at the stratum centres (jitter_samples = false) */

typedef struct rt_render_opts {
    uint32_t width, height;
    uint32_t spp;
    uint32_t max_depth;
    uint64_t seed;
    uint32_t shard_index; /* this call renders the 32x32 tiles t with             */
    uint32_t shard_count; /* t % shard_count == shard_index (0,1 = whole frame)   */
    uint32_t max_paths;   /* paths in flight per batch; 0 = library default       */
    uint32_t flags;
    rt_filter filter;     /* Film's filter (film.rs:20); kind 0 = none             */
    uint32_t sampler;     /* rt_sampler_kind                                      */
    uint32_t sampler_arg; /* RT_SAMPLER_STRATIFIED: x strata | RT_STRATIFIED_NO_JITTER */
} rt_render_opts;

typedef struct rt_stats {
    uint64_t paths;
    uint64_t segments;      /* ray segments traced: closest-hit + occlusion_rays   */
    uint64_t nodes_visited; /* with RT_RENDER_COUNTERS                           */
    uint64_t prims_tested[8]; /* per rt_prim_kind, with RT_RENDER_COUNTERS        */
    uint64_t trace_launches;
    double trace_ms; /* sum of trace-kernel durations (RT_RENDER_TIMING)        */
    double total_ms; /* first launch -> framebuffer resolved, device time        */
    uint32_t batches;
    uint32_t max_bounce;
    uint64_t bounce_rays[64]; /* closest-hit ray segments traced at bounce 0..63 */
    uint64_t occlusion_rays;  /* any-hit rays (RT_RENDER_DIRECT_LIGHT), part of `segments` */
} rt_stats;

typedef struct rt_scene rt_scene;

/* BVH::new (bvh.rs:108-121) + Sphere::new (sphere.rs:28-47) +
 * TriangleMesh::new / Triangle::from_mesh (triangle_mesh.rs:20-38,
 * triangle.rs:39-62): builds the tree on the host, flattens it
 * (bvh.rs:661-691).  Nothing is uploaded until a device is first used.
 * With RT_BUILD_ON_DEVICE (split method HLBVH) the Morton codes, the radix sort
 * and the treelets of hlbvh_build (bvh.rs:385-538) are computed on opts->device;
 * the tree is the same, node for node.
 * Media that enclose the scene (a Volume whose sphere contains the box around every
 * primitive that is not a sphere-bounded medium, e.g. the fog of test/week.toml:86-93) are
 * not given to the tree: every ray would meet them.  Up to four of them, in primitive order,
 * are tested once per ray before its traversal, and a hit shortens the ray like any other;
 * rt_scene_info lists them.  The reference has no media; its tree is built over exactly the
 * primitives it knows.
 * Balls: a medium bounded by a sphere (no instance transform) that no other primitive comes
 * near -- alone or inside one sphere primitive of exactly the same centre and radius, the glass
 * ball filled with fog of test/week.toml -- stays in the tree, but a ray that starts well inside
 * it (|o - c| < 0.998 r) is resolved by the ball's own primitives without a walk through the
 * tree whenever they decide its closest hit; the results are those of the walk, bit for bit,
 * and the traversal counters then take one test per primitive of the ball and no node.
 * rt_scene_info.n_balls says how many were found (at most four are used). */
int rt_scene_create(const rt_scene_desc *desc, const rt_build_opts *opts, rt_scene **out);

/* Drop for BVH. Frees host and device memory on every device used. */
void rt_scene_destroy(rt_scene *scene);

/* Primitive::world_bound (bvh.rs:728-734): root bounds {lo.xyz, hi.xyz} (joined with
 * the bounds of the enclosing media, which are part of the scene); an empty scene gives
 * the inverted box Bounds3::ZERO (bounds3.rs:23-26). */
int rt_scene_world_bound(const rt_scene *scene, float out[6]);

int rt_scene_get_info(const rt_scene *scene, rt_scene_info *out);

/* The flattened tree in the reference's own node format (bvh.rs:79-101) and
 * the primitive order the leaves index (BVH::primitives after construct,
 * bvh.rs:146).  `cap` = capacity of `out` in elements; returns the count (the
 * primitives of the tree: n_prims - n_enclosing). */
int64_t rt_scene_get_nodes(const rt_scene *scene, rt_linear_node *out, size_t cap);
int64_t rt_scene_get_prim_order(const rt_scene *scene, uint32_t *out, size_t cap);

/* Uploads the node array and primitive tables to `device` once (HBM resident
 * for the life of the scene).  Implicit on first use of a device. */
int rt_scene_upload(rt_scene *scene, int device);

/* <BVH as Primitive>::intersect (bvh.rs:736-799) over n rays.
 * Host pointers; the library stages through device memory. */
int rt_intersect(rt_scene *scene, const rt_ray *rays, size_t n, rt_hit *hits, int device);
/* Same, device pointers, asynchronous on `stream` (a hipStream_t; NULL = default). */
int rt_intersect_dev(rt_scene *scene, const rt_ray *d_rays, size_t n, rt_hit *d_hits, int device,
                     void *stream);

/* No counterpart in the reference.  rt_intersect_dev for callers that cannot allocate
 * device memory: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Calls no hipMalloc / hipFree or any other allocating routine: the scene must
 * already be resident on `device` (rt_scene_upload, else RT_ERR_INVALID), and the kernel's
 * scratch -- the 2 KiB of block cursors its waves claim ray blocks from -- lives in a
 * file-scope device array that each GPU holds for the life of the process.  At most
 * RT_NOALLOC_MAX_RAYS rays per call (RT_ERR_LIMIT above).  Every call on a GPU shares that
 * GPU's array: calls on one device must not overlap (same stream, or the caller orders
 * them); overlapping calls race on the cursors and may skip or repeat ray blocks. */
#define RT_NOALLOC_MAX_RAYS (1u << 28)
int rt_intersect_dev_noalloc(rt_scene *scene, const rt_ray *d_rays, size_t n, rt_hit *d_hits, int device,
                             void *stream);

/* Traversal counters of one rt_intersect_dev launch: out[0] = nodes visited,
 * out[1 + kind] = primitives tested per kind (9 values). Synchronous. */
int rt_intersect_count(rt_scene *scene, const rt_ray *d_rays, size_t n, rt_hit *d_hits, int device,
                       uint64_t out[9]);

/* <BVH as Primitive>::does_intersect (bvh.rs:801-863): 1 byte per ray. */
int rt_occluded(rt_scene *scene, const rt_ray *rays, size_t n, uint8_t *out, int device);
int rt_occluded_dev(rt_scene *scene, const rt_ray *d_rays, size_t n, uint8_t *d_out, int device,
                    void *stream);

/* Perspective::generate_ray (perspective.rs:54-78) for the paths
 * [first_path, first_path + n) of a frame; path p = pixel * spp + sample. */
int rt_generate_rays(const rt_camera *cam, const rt_render_opts *opts, uint64_t first_path,
                     size_t n, rt_ray *rays, int device);

/* The render loop the reference's main never got (main/src/main.rs:55-60):
 * generate -> [intersect -> shade/scatter -> compact] x depth -> resolve.
 * rgba_out: width*height*4 floats, mean linear radiance, alpha 1 on owned
 * pixels, all 0 on pixels of other shards.
 * With opts->filter.kind != 0 the samples go through the film instead
 * (FilmTile::add_sample, film.rs:281-328): a sample at raster position p adds
 * w * L and w to every pixel within the filter radius of p - 0.5, w from the
 * filter table; a shard adds only its own tiles' samples, so with several
 * shards pass RT_RENDER_RAW_FILM, sum the buffers and call rt_film_resolve_dev.
 * rt_render_dev with stats == NULL only enqueues the frame on `stream` and returns without waiting
 * (what follows on the stream, e.g. a collective, needs no host round trip); an error the kernels flag
 * in such a frame is reported by the next render call on that device.  With stats it waits for the
 * frame and fills them. */
int rt_render(rt_scene *scene, const rt_camera *cam, const rt_render_opts *opts, float *rgba_out,
              rt_stats *stats, int device);
int rt_render_dev(rt_scene *scene, const rt_camera *cam, const rt_render_opts *opts,
                  float *d_rgba_out, rt_stats *stats, int device, void *stream);

/* The 16 x 16 table of filter values Film::new precomputes (film.rs:99-113):
 * out[y * 16 + x] = eval((x + 0.5) * radius.x / 16, (y + 0.5) * radius.y / 16). */
int rt_filter_table(const rt_filter *filter, float out[256]);

/* The Sobol' tables RT_SAMPLER_SOBOL uses for a width x height frame, as data/src/sobol.rs
 * holds them: m32 = MATRICES_32[0 .. 5*52) (:68), vdc = MATRICES[m - 1] (:59978) and
 * inv = MATRICES_INV[m - 1] (:60151) for m = log2 of the frame's power-of-two resolution
 * (zeros when m = 0).  Computed from the Joe-Kuo direction numbers, not stored. */
int rt_sobol_tables(uint32_t width, uint32_t height, uint32_t m32[260], uint64_t vdc[52], uint64_t inv[52]);

/* Film::write_image's normalisation (film.rs:236-247) of a raw film
 * {sum of w*L (rgb), sum of w}: rgb = max(0, sum * (1 / weight)) where the weight
 * is not zero, alpha = 1.  In place, asynchronous on `stream`. */
int rt_film_resolve_dev(float *d_rgba, uint32_t width, uint32_t height, int device, void *stream);

/* Panic message analogue; valid until the next failing call on this thread. */
const char *rt_last_error(void);

/* sizeof() of every struct above in declaration order (layout check for FFI
 * bindings).  Returns the number of entries written (<= cap). */
int rt_abi_sizes(uint32_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RT355_H */
