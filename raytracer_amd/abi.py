"""ctypes mirror of ``include/rt355.h`` (struct layouts, enums, prototypes).

The same structures describe a scene to ``librt355.so`` and, in the test
suite, to the CPU oracle, so that both receive byte-identical input.
"""
import ctypes as C

import numpy as np

RT_OK = 0
RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_DEPTH, RT_ERR_NOMEM, RT_ERR_LIMIT = (
    -1, -2, -3, -4, -5, -6)
RT_MISS = 0xFFFFFFFF
RT_NOALLOC_MAX_RAYS = 1 << 28

# rt_prim_kind
RT_SPHERE, RT_MOVING_SPHERE, RT_TRIANGLE, RT_RECT, RT_BOX, RT_GSPHERE, RT_MEDIUM = range(7)
RT_PRIM_REVERSE = 0x100
RT_BOUNDARY_SPHERE, RT_BOUNDARY_BOX = 0, 1
# rt_material_kind
RT_MAT_LAMBERTIAN, RT_MAT_METALLIC, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC = range(5)
# rt_texture_kind
RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_IMAGE, RT_TEX_NOISE, RT_TEX_CHECKER_UV = range(5)
# rt_split_method == SplitMethod (shapes/src/bvh.rs:10-22)
RT_SPLIT_SAH, RT_SPLIT_HLBVH, RT_SPLIT_MIDDLE, RT_SPLIT_EQUAL_COUNTS = range(4)
RT_RENDER_COUNTERS, RT_RENDER_TIMING, RT_RENDER_RAW_FILM, RT_RENDER_DIRECT_LIGHT = 1, 2, 4, 8
(RT_FILTER_NONE, RT_FILTER_BOX, RT_FILTER_TRIANGLE, RT_FILTER_GAUSSIAN, RT_FILTER_MITCHELL,
 RT_FILTER_LANCZOS) = range(6)

# numpy record layouts of the plain-data structs
ray_dtype = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("t_max", "<f4"), ("time", "<f4")])
hit_dtype = np.dtype([("t", "<f4"), ("prim", "<u4"), ("u", "<f4"), ("v", "<f4"),
                      ("n", "<f4", 3), ("flags", "<u4")])
prim_ref_dtype = np.dtype([("kind", "<u4"), ("index", "<u4"), ("material", "<u4"), ("xform", "<u4")])
sphere_dtype = np.dtype([("c", "<f4", 3), ("r", "<f4")])
moving_sphere_dtype = np.dtype([("c0", "<f4", 3), ("r", "<f4"), ("c1", "<f4", 3), ("time0", "<f4"),
                                ("time1", "<f4"), ("_pad", "<f4", 3)])
rect_dtype = np.dtype([("axis", "<u4"), ("k", "<f4"), ("a0", "<f4"), ("a1", "<f4"), ("b0", "<f4"),
                       ("b1", "<f4"), ("_pad", "<u4", 2)])
box_dtype = np.dtype([("lo", "<f4", 3), ("_p0", "<f4"), ("hi", "<f4", 3), ("_p1", "<f4")])
gsphere_dtype = np.dtype([("o2w", "<f4", 12), ("w2o", "<f4", 12), ("radius", "<f4"),
                          ("z_min", "<f4"), ("z_max", "<f4"), ("phi_max_deg", "<f4"),
                          ("_pad", "<u4", 4)])
xform_dtype = np.dtype([("o2w", "<f4", 12), ("w2o", "<f4", 12)])
medium_dtype = np.dtype([("boundary_kind", "<u4"), ("boundary_index", "<u4"), ("xform", "<u4"),
                         ("density", "<f4")])
material_dtype = np.dtype([("kind", "<u4"), ("texture", "<u4"), ("fuzz", "<f4"), ("ior", "<f4")])
texture_dtype = np.dtype([("kind", "<u4"), ("rgb", "<f4", 3), ("a", "<u4"), ("b", "<u4"),
                          ("scale", "<f4"), ("_pad", "<u4")])
distant_light_dtype = np.dtype([("w", "<f4", 3), ("rgb", "<f4", 3), ("_pad", "<u4", 2)])
linear_node_dtype = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("child_offset", "<u4"),
                              ("primitive_count", "<u2"), ("axis", "u1"), ("_pad", "u1")])

assert ray_dtype.itemsize == 32 and hit_dtype.itemsize == 32 and linear_node_dtype.itemsize == 32


class rt_image(C.Structure):
    _fields_ = [("rgb8", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32)]


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("prims", C.c_void_p), ("spheres", C.c_void_p), ("moving_spheres", C.c_void_p),
        ("tri_verts", C.c_void_p), ("tri_uvs", C.c_void_p), ("tri_normals", C.c_void_p),
        ("rects", C.c_void_p), ("boxes", C.c_void_p), ("gspheres", C.c_void_p),
        ("xforms", C.c_void_p), ("media", C.c_void_p), ("materials", C.c_void_p),
        ("textures", C.c_void_p), ("images", C.c_void_p), ("perlin_vec", C.c_void_p),
        ("perlin_perm", C.c_void_p),
        ("n_prims", C.c_uint32), ("n_spheres", C.c_uint32), ("n_moving_spheres", C.c_uint32),
        ("n_tris", C.c_uint32), ("n_rects", C.c_uint32), ("n_boxes", C.c_uint32),
        ("n_gspheres", C.c_uint32), ("n_xforms", C.c_uint32), ("n_media", C.c_uint32),
        ("n_materials", C.c_uint32), ("n_textures", C.c_uint32), ("n_images", C.c_uint32),
        ("background", C.c_float * 3), ("n_distant_lights", C.c_uint32), ("distant_lights", C.c_void_p),
    ]


class rt_build_opts(C.Structure):
    _fields_ = [("max_prims_in_node", C.c_uint32), ("split_method", C.c_uint32), ("flags", C.c_uint32),
                ("device", C.c_uint32)]


RT_BUILD_ON_DEVICE = 1


class rt_scene_info(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("n_nodes", C.c_uint32), ("max_depth", C.c_uint32),
                ("stack_slots", C.c_uint32), ("device_bytes", C.c_uint64), ("build_ms", C.c_double),
                ("n_enclosing", C.c_uint32), ("enclosing", C.c_uint32 * 4), ("n_balls", C.c_uint32)]


class rt_camera(C.Structure):
    _fields_ = [("raster_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 12),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float),
                ("shutter_open", C.c_float), ("shutter_close", C.c_float)]


class rt_filter(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("radius", C.c_float * 2), ("a", C.c_float), ("b", C.c_float),
                ("_pad", C.c_uint32)]


class rt_render_opts(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32),
                ("max_depth", C.c_uint32), ("seed", C.c_uint64), ("shard_index", C.c_uint32),
                ("shard_count", C.c_uint32), ("max_paths", C.c_uint32), ("flags", C.c_uint32),
                ("filter", rt_filter), ("sampler", C.c_uint32), ("sampler_arg", C.c_uint32)]


RT_SAMPLER_RANDOM, RT_SAMPLER_HALTON, RT_SAMPLER_SOBOL, RT_SAMPLER_ZERO_TWO, RT_SAMPLER_STRATIFIED = range(5)
RT_STRATIFIED_NO_JITTER = 0x80000000


class rt_stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("prims_tested", C.c_uint64 * 8), ("trace_launches", C.c_uint64),
                ("trace_ms", C.c_double), ("total_ms", C.c_double), ("batches", C.c_uint32),
                ("max_bounce", C.c_uint32), ("bounce_rays", C.c_uint64 * 64), ("occlusion_rays", C.c_uint64)]


# sizeof() table in the order rt_abi_sizes() reports it
ABI_SIZES = [
    ("rt_ray", ray_dtype.itemsize), ("rt_hit", hit_dtype.itemsize),
    ("rt_prim_ref", prim_ref_dtype.itemsize), ("rt_sphere", sphere_dtype.itemsize),
    ("rt_moving_sphere", moving_sphere_dtype.itemsize), ("rt_rect", rect_dtype.itemsize),
    ("rt_box", box_dtype.itemsize), ("rt_gsphere", gsphere_dtype.itemsize),
    ("rt_xform", xform_dtype.itemsize), ("rt_medium", medium_dtype.itemsize),
    ("rt_material", material_dtype.itemsize), ("rt_texture", texture_dtype.itemsize),
    ("rt_image", C.sizeof(rt_image)), ("rt_scene_desc", C.sizeof(rt_scene_desc)),
    ("rt_build_opts", C.sizeof(rt_build_opts)), ("rt_linear_node", linear_node_dtype.itemsize),
    ("rt_scene_info", C.sizeof(rt_scene_info)), ("rt_camera", C.sizeof(rt_camera)),
    ("rt_filter", C.sizeof(rt_filter)),
    ("rt_render_opts", C.sizeof(rt_render_opts)), ("rt_stats", C.sizeof(rt_stats)),
    ("rt_distant_light", distant_light_dtype.itemsize),
]

# every symbol include/rt355.h declares
EXPORTS = [
    "rt_scene_create", "rt_scene_destroy", "rt_scene_world_bound", "rt_scene_get_info",
    "rt_scene_get_nodes", "rt_scene_get_prim_order", "rt_scene_upload", "rt_intersect",
    "rt_intersect_dev", "rt_intersect_dev_noalloc", "rt_intersect_count", "rt_occluded", "rt_occluded_dev",
    "rt_generate_rays", "rt_render", "rt_render_dev", "rt_filter_table", "rt_sobol_tables", "rt_film_resolve_dev",
    "rt_last_error", "rt_abi_sizes",
]


def declare(lib, prefix="rt_"):
    """Attach argtypes/restypes for the functions both libraries share."""
    vp, sz, i32, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
    P = C.POINTER

    def fn(name, restype, *argtypes):
        f = getattr(lib, prefix + name)
        f.restype = restype
        f.argtypes = list(argtypes)
        return f

    fn("scene_create", i32, P(rt_scene_desc), P(rt_build_opts), P(vp))
    fn("scene_destroy", None, vp)
    fn("scene_world_bound", i32, vp, P(C.c_float * 6))
    fn("scene_get_info", i32, vp, P(rt_scene_info))
    fn("scene_get_nodes", C.c_int64, vp, vp, sz)
    fn("scene_get_prim_order", C.c_int64, vp, vp, sz)
    fn("last_error", C.c_char_p)
    if prefix == "rt_":
        fn("scene_upload", i32, vp, i32)
        fn("intersect", i32, vp, vp, sz, vp, i32)
        fn("intersect_dev", i32, vp, vp, sz, vp, i32, vp)
        fn("intersect_dev_noalloc", i32, vp, vp, sz, vp, i32, vp)
        fn("intersect_count", i32, vp, vp, sz, vp, i32, P(u64 * 9))
        fn("occluded", i32, vp, vp, sz, vp, i32)
        fn("occluded_dev", i32, vp, vp, sz, vp, i32, vp)
        fn("generate_rays", i32, P(rt_camera), P(rt_render_opts), u64, sz, vp, i32)
        fn("filter_table", i32, P(rt_filter), vp)
        fn("render", i32, vp, P(rt_camera), P(rt_render_opts), vp, P(rt_stats), i32)
        fn("render_dev", i32, vp, P(rt_camera), P(rt_render_opts), vp, P(rt_stats), i32, vp)
        fn("film_resolve_dev", i32, vp, C.c_uint32, C.c_uint32, i32, vp)
        fn("abi_sizes", i32, P(C.c_uint32), sz)
    return lib
