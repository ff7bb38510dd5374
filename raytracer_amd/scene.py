"""Host-side scene assembly: flat arrays in the layout of ``rt_scene_desc``.

``SceneData`` collects primitives, transforms, materials and textures and
freezes them into the ctypes description that ``rt_scene_create`` takes.
Transforms and the perspective camera are evaluated in float64 on the host and
rounded to float32 once, mirroring geometry/src/transform.rs:100-290 and
colour/src/camera/projective.rs:28-52.
"""
import ctypes as C
import math

import numpy as np

from . import abi
from .pcg32 import Rng

F32 = np.float32


# ---- transforms (4x4 float64, row-major; reference Transform{mat, inv}) --------
def identity():
    return np.eye(4)


def translate(v):  # transform.rs Transform::translation
    m = np.eye(4)
    m[:3, 3] = v
    return m


def scale(v):
    return np.diag([v[0], v[1], v[2], 1.0])


def rotate(axis, angle_deg):  # transform.rs:204-235
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    s, c = math.sin(math.radians(angle_deg)), math.cos(math.radians(angle_deg))
    x, y, z = a
    m = np.eye(4)
    m[0, :3] = [x * x + (1 - x * x) * c, x * y * (1 - c) - z * s, x * z * (1 - c) + y * s]
    m[1, :3] = [x * y * (1 - c) + z * s, y * y + (1 - y * y) * c, y * z * (1 - c) - x * s]
    m[2, :3] = [x * z * (1 - c) - y * s, y * z * (1 - c) + x * s, z * z + (1 - z * z) * c]
    return m


def rotate_y(angle_deg):
    return rotate([0.0, 1.0, 0.0], angle_deg)


def look_at(pos, look, up, right_handed=False):
    """camera_to_world.  pbrt convention (transform.rs:240-256): right = up x dir.
    ``right_handed`` builds right = dir x up instead, which shows a y-up
    right-handed scene unmirrored (the legacy TOML scenes)."""
    pos, look, up = (np.asarray(v, dtype=np.float64) for v in (pos, look, up))
    d = look - pos
    d = d / np.linalg.norm(d)
    upn = up / np.linalg.norm(up)
    right = np.cross(d, upn) if right_handed else np.cross(upn, d)
    right = right / np.linalg.norm(right)
    new_up = np.cross(right, d) if right_handed else np.cross(d, right)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, new_up, d, pos
    return m


def mat34(m):
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64)[:3, :4].reshape(12), dtype=F32)


def perspective_camera(width, height, fov_deg, camera_to_world, lens_radius=0.0,
                       focal_distance=1e6, shutter=(0.0, 0.0), near=1e-2, far=1000.0):
    """rt_camera for Perspective::new (perspective.rs:18-50): fov spans the
    shorter image axis; raster (0,0) is the top-left corner."""
    aspect = width / height
    if aspect > 1.0:
        sx0, sx1, sy0, sy1 = -aspect, aspect, -1.0, 1.0
    else:
        sx0, sx1, sy0, sy1 = -1.0, 1.0, -1.0 / aspect, 1.0 / aspect
    persp = np.array([[1, 0, 0, 0], [0, 1, 0, 0],
                      [0, 0, far / (far - near), -far * near / (far - near)],
                      [0, 0, 1, 0]], dtype=np.float64)
    inv_tan = 1.0 / math.tan(math.radians(fov_deg) / 2.0)  # transform.rs:288 omits the 1/
    camera_to_screen = scale([inv_tan, inv_tan, 1.0]) @ persp
    screen_to_raster = (scale([width, height, 1.0])
                        @ scale([1.0 / (sx1 - sx0), 1.0 / (sy0 - sy1), 1.0])
                        @ translate([-sx0, -sy1, 0.0]))
    raster_to_camera = np.linalg.inv(camera_to_screen) @ np.linalg.inv(screen_to_raster)
    cam = abi.rt_camera()
    cam.raster_to_camera[:] = [float(F32(v)) for v in raster_to_camera.reshape(16)]
    cam.camera_to_world[:] = [float(v) for v in mat34(camera_to_world)]
    cam.lens_radius = lens_radius
    cam.focal_distance = focal_distance
    cam.shutter_open, cam.shutter_close = shutter
    return cam


# ---- scene arrays -----------------------------------------------------------------
class FrozenScene:
    """rt_scene_desc plus the numpy arrays it points into (kept alive here)."""

    def __init__(self, desc, keep):
        self.desc = desc
        self._keep = keep


class SceneData:
    def __init__(self, background=(0.0, 0.0, 0.0)):
        self.background = tuple(background)
        self.prims = []  # (kind, index, material, xform)
        self.spheres, self.moving, self.rects, self.boxes = [], [], [], []
        self.gspheres, self.xforms, self.media = [], [], []
        self.tri_verts, self.tri_uvs, self.tri_normals = [], [], []
        self.materials, self.textures, self.images = [], [], []
        self.perlin = None
        self.sphere_blocks = []  # (centres [n,3], radii [n]) appended after the single spheres

    # -- appearance --
    def add_texture(self, kind=abi.RT_TEX_SOLID, rgb=(0, 0, 0), a=0, b=0, scale=1.0):
        self.textures.append((kind, tuple(rgb), a, b, scale))
        return len(self.textures) - 1

    def solid(self, rgb):
        return self.add_texture(abi.RT_TEX_SOLID, rgb)

    def add_image(self, rgb8):
        img = np.ascontiguousarray(rgb8, dtype=np.uint8)
        assert img.ndim == 3 and img.shape[2] == 3
        self.images.append(img)
        return len(self.images) - 1

    def add_material(self, kind, texture=None, rgb=None, fuzz=0.0, ior=1.0):
        if texture is None:
            texture = self.solid(rgb if rgb is not None else (1.0, 1.0, 1.0))
        self.materials.append((kind, texture, fuzz, ior))
        return len(self.materials) - 1

    def lambertian(self, rgb=None, texture=None):
        return self.add_material(abi.RT_MAT_LAMBERTIAN, texture, rgb)

    def metallic(self, rgb, fuzz):
        return self.add_material(abi.RT_MAT_METALLIC, None, rgb, fuzz=fuzz)

    def dielectric(self, ior):
        return self.add_material(abi.RT_MAT_DIELECTRIC, None, (1, 1, 1), ior=ior)

    def diffuse_light(self, rgb):
        return self.add_material(abi.RT_MAT_DIFFUSE_LIGHT, None, rgb)

    def isotropic(self, rgb):
        return self.add_material(abi.RT_MAT_ISOTROPIC, None, rgb)

    def set_perlin(self, sequence=7):
        """256 random unit vectors and three permutations of 0..255 drawn from
        PCG32 (geometry/src/rng.rs) with a fixed sequence index."""
        rng = Rng(sequence)
        vec = np.empty((256, 3), dtype=np.float64)
        for i in range(256):
            v = np.array([2.0 * rng.uniform_float() - 1.0 for _ in range(3)])
            vec[i] = v / np.linalg.norm(v)
        perm = np.empty((3, 256), dtype=np.int32)
        for t in range(3):
            p = list(range(256))
            for i in range(255, 0, -1):
                j = rng.uniform_u32_limit(i + 1)
                p[i], p[j] = p[j], p[i]
            perm[t] = p
        self.perlin = (np.ascontiguousarray(vec, dtype=F32), perm)

    # -- geometry --
    def add_xform(self, object_to_world):
        m = np.asarray(object_to_world, dtype=np.float64)
        self.xforms.append((mat34(m), mat34(np.linalg.inv(m))))
        return len(self.xforms)  # 1-based

    def _prim(self, kind, index, material, xform=0, reverse=False):
        self.prims.append((kind | (abi.RT_PRIM_REVERSE if reverse else 0), index, material, xform))
        return len(self.prims) - 1

    def add_sphere(self, centre, radius, material=0):
        self.spheres.append((tuple(centre), radius))
        return self._prim(abi.RT_SPHERE, len(self.spheres) - 1, material)

    def add_sphere_block(self, centres, radii, material=0):
        centres = np.asarray(centres, dtype=F32).reshape(-1, 3)
        radii = np.broadcast_to(np.asarray(radii, dtype=F32), (len(centres),))
        first = sum(len(b[0]) for b in self.sphere_blocks)
        self.sphere_blocks.append((centres, radii))
        base = len(self.prims)
        self.prims.append(("sblock", abi.RT_SPHERE, first, len(centres), material))
        return base

    def add_moving_sphere(self, c0, c1, time0, time1, radius, material=0):
        self.moving.append((tuple(c0), radius, tuple(c1), time0, time1))
        return self._prim(abi.RT_MOVING_SPHERE, len(self.moving) - 1, material)

    def add_rect(self, axis, k, a0, a1, b0, b1, material=0, xform=0, reverse=False):
        self.rects.append((axis, k, a0, a1, b0, b1))
        return self._prim(abi.RT_RECT, len(self.rects) - 1, material, xform, reverse)

    def _box(self, lo, hi):
        self.boxes.append((tuple(lo), tuple(hi)))
        return len(self.boxes) - 1

    def add_box(self, lo, hi, material=0, xform=0):
        return self._prim(abi.RT_BOX, self._box(lo, hi), material, xform)

    def add_gsphere(self, object_to_world, radius, z=None, phi_max=360.0, material=0,
                    reverse=False, world_to_object=None):
        """Sphere::new (shapes/src/sphere.rs:28-47)."""
        o2w = np.asarray(object_to_world, dtype=np.float64)
        w2o = np.linalg.inv(o2w) if world_to_object is None else np.asarray(world_to_object)
        z = (-radius, radius) if z is None else z
        self.gspheres.append((mat34(o2w), mat34(w2o), radius, z[0], z[1], phi_max))
        return self._prim(abi.RT_GSPHERE, len(self.gspheres) - 1, material, 0, reverse)

    def add_triangles(self, verts, uvs=None, normals=None, material=0, reverse=False):
        """verts [n,3,3] world space; uvs [n,3,2]; normals [n,3,3].  One
        primitive per triangle (Triangle::from_mesh, triangle.rs:39-62)."""
        verts = np.asarray(verts, dtype=F32).reshape(-1, 3, 3)
        n = len(verts)
        first = sum(len(v) for v in self.tri_verts)
        self.tri_verts.append(verts)
        self.tri_uvs.append(None if uvs is None else np.asarray(uvs, dtype=F32).reshape(n, 3, 2))
        self.tri_normals.append(None if normals is None
                                else np.asarray(normals, dtype=F32).reshape(n, 3, 3))
        kind = abi.RT_TRIANGLE | (abi.RT_PRIM_REVERSE if reverse else 0)
        base = len(self.prims)
        self.prims.append(("block", kind, first, n, material))
        return base

    def add_mesh(self, object_to_world, positions, indices, uv=None, normal=None, material=0,
                 reverse=False):
        """TriangleMesh::new + with_uv/with_normal (triangle_mesh.rs:20-85):
        positions go to world space once, here (rounded to f32 after the f64
        product); one triangle per index triple."""
        m = np.asarray(object_to_world, dtype=np.float64)
        p = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        pw = (p @ m[:3, :3].T + m[:3, 3]).astype(F32)
        idx = np.asarray(indices, dtype=np.int64).reshape(-1, 3)
        uvs = None if uv is None else np.asarray(uv, dtype=F32).reshape(-1, 2)[idx]
        nrm = None
        if normal is not None:
            inv_t = np.linalg.inv(m[:3, :3]).T  # normals: inverse transpose
            nrm = (np.asarray(normal, dtype=np.float64).reshape(-1, 3) @ inv_t.T).astype(F32)[idx]
        return self.add_triangles(pw[idx], uvs, nrm, material, reverse)

    def add_medium(self, density, material, sphere=None, box=None, xform=0):
        if sphere is not None:
            self.spheres.append((tuple(sphere[0]), sphere[1]))
            self.media.append((abi.RT_BOUNDARY_SPHERE, len(self.spheres) - 1, 0, density))
        else:
            self.media.append((abi.RT_BOUNDARY_BOX, self._box(box[0], box[1]), xform, density))
        return self._prim(abi.RT_MEDIUM, len(self.media) - 1, material)

    def add_distant_light(self, towards, rgb):
        """A light infinitely far away in direction `towards` (from the scene; any length) with
        radiance rgb: LightSource "distant" of test/test.pbrt.  Sampled by RT_RENDER_DIRECT_LIGHT."""
        w = np.asarray(towards, dtype=np.float64)
        w = w / np.linalg.norm(w)
        if not hasattr(self, "distant"):
            self.distant = []
        self.distant.append((tuple(float(x) for x in w.astype(F32)), tuple(float(x) for x in rgb), (0, 0)))
        return len(self.distant) - 1

    # -- freeze --
    def n_prims(self):
        return sum(p[3] if isinstance(p[0], str) else 1 for p in self.prims)

    def freeze(self):
        keep = {}

        def rec(name, dtype, rows):
            arr = np.zeros(len(rows), dtype=dtype)
            for i, row in enumerate(rows):
                arr[i] = row
            keep[name] = arr
            return arr

        # primitive list (triangle blocks expand to one entry per triangle)
        n = self.n_prims()
        prims = np.zeros(n, dtype=abi.prim_ref_dtype)
        at = 0
        for p in self.prims:
            if isinstance(p[0], str):
                _, kind, first, cnt, mat = p
                if p[0] == "sblock":
                    first += len(self.spheres)
                prims["kind"][at:at + cnt] = kind
                prims["index"][at:at + cnt] = np.arange(first, first + cnt, dtype=np.uint32)
                prims["material"][at:at + cnt] = mat
                at += cnt
            else:
                prims[at] = p
                at += 1
        keep["prims"] = prims
        spheres = rec("spheres", abi.sphere_dtype, self.spheres)
        if self.sphere_blocks:
            extra = np.zeros(sum(len(b[0]) for b in self.sphere_blocks), dtype=abi.sphere_dtype)
            extra["c"] = np.concatenate([b[0] for b in self.sphere_blocks])
            extra["r"] = np.concatenate([b[1] for b in self.sphere_blocks])
            spheres = keep["spheres"] = np.concatenate([spheres, extra])
        moving = rec("moving", abi.moving_sphere_dtype,
                     [(c0, r, c1, t0, t1, (0, 0, 0)) for c0, r, c1, t0, t1 in self.moving])
        rects = rec("rects", abi.rect_dtype, [r + ((0, 0),) for r in self.rects])
        boxes = rec("boxes", abi.box_dtype, [(lo, 0.0, hi, 0.0) for lo, hi in self.boxes])
        gsph = rec("gspheres", abi.gsphere_dtype, [g + ((0, 0, 0, 0),) for g in self.gspheres])
        xforms = rec("xforms", abi.xform_dtype, self.xforms)
        media = rec("media", abi.medium_dtype, self.media)
        mats = rec("materials", abi.material_dtype, self.materials)
        texs = rec("textures", abi.texture_dtype, [t + (0,) for t in self.textures])
        n_tris = sum(len(v) for v in self.tri_verts)
        tv = tu = tn = None
        if n_tris:
            tv = keep["tri_verts"] = np.ascontiguousarray(np.concatenate(self.tri_verts), dtype=F32)
            if any(u is not None for u in self.tri_uvs):
                default = np.array([[0, 0], [1, 0], [0, 1]], dtype=F32)
                tu = keep["tri_uvs"] = np.ascontiguousarray(np.concatenate(
                    [u if u is not None else np.broadcast_to(default, (len(v), 3, 2))
                     for u, v in zip(self.tri_uvs, self.tri_verts)]), dtype=F32)
            if any(x is not None for x in self.tri_normals):
                parts = []
                for x, v in zip(self.tri_normals, self.tri_verts):
                    if x is None:  # flat: geometric normal on every vertex
                        g = np.cross(v[:, 0] - v[:, 2], v[:, 1] - v[:, 2])
                        x = np.repeat(g[:, None, :], 3, axis=1)
                    parts.append(x)
                tn = keep["tri_normals"] = np.ascontiguousarray(np.concatenate(parts), dtype=F32)
        images = (abi.rt_image * max(1, len(self.images)))()
        for i, img in enumerate(self.images):
            images[i].rgb8 = img.ctypes.data
            images[i].w, images[i].h = img.shape[1], img.shape[0]
        keep["images"] = (images, list(self.images))

        def ptr(a):
            return a.ctypes.data if a is not None and len(a) else None

        d = abi.rt_scene_desc()
        d.prims, d.spheres, d.moving_spheres = ptr(prims), ptr(spheres), ptr(moving)
        d.tri_verts, d.tri_uvs, d.tri_normals = ptr(tv), ptr(tu), ptr(tn)
        d.rects, d.boxes, d.gspheres, d.xforms = ptr(rects), ptr(boxes), ptr(gsph), ptr(xforms)
        d.media, d.materials, d.textures = ptr(media), ptr(mats), ptr(texs)
        d.images = C.cast(images, C.c_void_p) if self.images else None
        if self.perlin is not None:
            keep["perlin"] = self.perlin
            d.perlin_vec, d.perlin_perm = self.perlin[0].ctypes.data, self.perlin[1].ctypes.data
        (d.n_prims, d.n_spheres, d.n_moving_spheres, d.n_tris, d.n_rects, d.n_boxes) = (
            n, len(spheres), len(moving), n_tris, len(rects), len(boxes))
        (d.n_gspheres, d.n_xforms, d.n_media, d.n_materials, d.n_textures, d.n_images) = (
            len(gsph), len(xforms), len(media), len(mats), len(texs), len(self.images))
        d.background[:] = self.background
        lights = getattr(self, "distant", [])
        if lights:
            dl = keep["distant_lights"] = np.array(lights, dtype=abi.distant_light_dtype)
            d.distant_lights, d.n_distant_lights = dl.ctypes.data, len(dl)
        return FrozenScene(d, keep)
