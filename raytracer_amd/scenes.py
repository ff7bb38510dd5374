"""The benchmark scenes, built procedurally from the parameters of the
reference's scene files (test/random.toml, test/cornell.toml, test/week.toml,
test/test.pbrt) plus the synthetic stress scene.  Each builder returns
``(SceneData, camera_factory)`` where ``camera_factory(width, height)`` gives the
``rt_camera``.  Generators (SphereField, RandomBoxes, RandomCube) draw from
PCG32 (geometry/src/rng.rs) with fixed sequence indices.
"""
import math

import numpy as np

from . import abi, pcg32
from .scene import (SceneData, identity, look_at, perspective_camera, rotate, rotate_y, translate)


def _legacy_camera(origin, target, vfov, up=(0.0, 1.0, 0.0), aperture=0.0, focus=None,
                   shutter=(0.0, 1.0)):
    c2w = look_at(origin, target, up, right_handed=True)
    if focus is None:
        focus = float(np.linalg.norm(np.asarray(target, float) - np.asarray(origin, float)))

    def make(width, height):
        # vfov is the vertical angle; pbrt's fov spans the shorter axis
        fov = vfov
        if width < height:
            fov = math.degrees(2.0 * math.atan(math.tan(math.radians(vfov) / 2.0) * width / height))
        return perspective_camera(width, height, fov, c2w, lens_radius=aperture / 2.0,
                                  focal_distance=focus, shutter=shutter)

    return make


def random_spheres():
    """test/random.toml: ground + three large spheres + SphereField (22 x 22
    jittered small spheres; material chosen by a uniform draw)."""
    s = SceneData(background=(0.7, 0.8, 1.0))
    checker = s.add_texture(abi.RT_TEX_CHECKER, a=s.solid((0.2, 0.3, 0.1)),
                            b=s.solid((0.9, 0.9, 0.9)), scale=10.0)
    s.add_sphere((0.0, -1000.0, 0.0), 1000.0, s.lambertian(texture=checker))
    s.add_sphere((0.0, 1.0, 0.0), 1.0, s.dielectric(1.5))
    s.add_sphere((-4.0, 1.0, 0.0), 1.0, s.lambertian((0.4, 0.2, 0.1)))
    s.add_sphere((4.0, 1.0, 0.0), 1.0, s.metallic((0.7, 0.6, 0.5), 0.0))
    rng = pcg32.Rng(1)
    glass = s.dielectric(1.5)
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = rng.uniform_float()
            c = (a + 0.9 * rng.uniform_float(), 0.2, b + 0.9 * rng.uniform_float())
            if math.dist(c, (4.0, 0.2, 0.0)) <= 0.9:
                continue
            if choose < 0.8:
                rgb = [rng.uniform_float() * rng.uniform_float() for _ in range(3)]
                m = s.lambertian(rgb)
            elif choose < 0.95:
                rgb = [0.5 + 0.5 * rng.uniform_float() for _ in range(3)]
                m = s.metallic(rgb, 0.5 * rng.uniform_float())
            else:
                m = glass
            s.add_sphere(c, 0.2, m)
    cam = _legacy_camera((12.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, aperture=0.1, focus=10.0,
                         shutter=(0.0, 0.0))
    return s, cam


def cornell():
    """test/cornell.toml: five walls, a light and two smoke boxes."""
    s = SceneData()
    red, white = s.lambertian((0.65, 0.05, 0.05)), s.lambertian((0.73, 0.73, 0.73))
    green, light = s.lambertian((0.12, 0.45, 0.15)), s.diffuse_light((7.0, 7.0, 7.0))
    s.add_rect(0, 555.0, 0.0, 555.0, 0.0, 555.0, green)
    s.add_rect(0, 0.0, 0.0, 555.0, 0.0, 555.0, red)
    s.add_rect(1, 554.0, 113.0, 443.0, 127.0, 432.0, light)
    s.add_rect(1, 0.0, 0.0, 555.0, 0.0, 555.0, white)
    s.add_rect(1, 555.0, 0.0, 555.0, 0.0, 555.0, white)
    s.add_rect(2, 555.0, 0.0, 555.0, 0.0, 555.0, white)
    x1 = s.add_xform(translate((265.0, 0.0, 295.0)) @ rotate_y(15.0))
    s.add_medium(0.01, s.isotropic((0.0, 0.0, 0.0)), box=((0, 0, 0), (165.0, 330.0, 165.0)), xform=x1)
    x2 = s.add_xform(translate((130.0, 0.0, 65.0)) @ rotate_y(-18.0))
    s.add_medium(0.01, s.isotropic((1.0, 1.0, 1.0)), box=((0, 0, 0), (165.0, 165.0, 165.0)), xform=x2)
    cam = _legacy_camera((278.0, 287.0, -800.0), (278.0, 278.0, 0.0), 40.0, shutter=(0.0, 0.0))
    return s, cam


def earth_texture(width=1024, height=512):
    """Procedural stand-in for test/earthmap.jpg (same size): land / sea / ice
    from a few octaves of integer-hashed value noise, so every byte is
    reproducible without the asset."""
    yy, xx = np.mgrid[0:height, 0:width]

    def lattice(ix, iy, seed):
        h = (ix.astype(np.uint32) * np.uint32(374761393) + iy.astype(np.uint32) * np.uint32(668265263)
             + np.uint32(seed) * np.uint32(2246822519))
        h = (h ^ (h >> np.uint32(13))) * np.uint32(1274126177)
        h = h ^ (h >> np.uint32(16))
        return (h & np.uint32(0xFFFF)).astype(np.float64) / 65535.0

    def value_noise(cells, seed):
        fx, fy = xx * cells / width, yy * cells / height
        ix, iy = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
        tx, ty = fx - ix, fy - iy
        tx, ty = tx * tx * (3 - 2 * tx), ty * ty * (3 - 2 * ty)
        ixw = (ix + 1) % cells  # wrap in longitude
        a, b = lattice(ix, iy, seed), lattice(ixw, iy, seed)
        c, d = lattice(ix, iy + 1, seed), lattice(ixw, iy + 1, seed)
        return (a * (1 - tx) + b * tx) * (1 - ty) + (c * (1 - tx) + d * tx) * ty

    with np.errstate(over="ignore"):
        h = sum(value_noise(4 << o, 11 + o) * 0.5 ** o for o in range(5)) / 1.9375
    lat = np.abs(yy / (height - 1) - 0.5) * 2.0
    img = np.zeros((height, width, 3), dtype=np.float64)
    sea, land, ice = np.array([20, 60, 140.0]), np.array([60, 130, 50.0]), np.array([235, 240, 245.0])
    img[:] = sea
    img[h > 0.52] = land
    img[(h > 0.62)] = np.array([120, 110, 70.0])
    img[lat > 0.86] = ice
    return img.astype(np.uint8)


def earth_image(file_name="earthmap.jpg"):
    """The Image texture the legacy scenes name (test/week.toml:101, test/globe.toml): the
    reference's test/earthmap.jpg, decoded once to RGB8 and kept as a fixture
    (tests/golden/earthmap_rgb8.npz, written by tests/golden/make_reference_assets.py).  A tree
    without the fixture falls back to the procedural stand-in of the same size."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "earthmap_rgb8.npz")
    if os.path.basename(file_name) == "earthmap.jpg" and os.path.exists(path):
        return np.load(path)["rgb8"]
    return earth_texture()


def next_week(boxes_per_side=20, cube_count=1000):
    """test/week.toml: RandomBoxes ground, light, moving sphere, glass, metal,
    a glass ball filled with blue fog, a thin global fog, an image-textured
    globe, a Perlin-noise sphere and a rotated, translated cube of 1000 small
    spheres."""
    s = SceneData()
    s.set_perlin()
    ground = s.lambertian((0.48, 0.83, 0.53))
    h = pcg32.streams(1000, boxes_per_side * boxes_per_side, 1)[:, 0]
    for i in range(boxes_per_side):
        for j in range(boxes_per_side):
            w = 100.0
            x0, z0 = -1000.0 + i * w, -1000.0 + j * w
            y1 = 1.0 + 100.0 * float(h[i * boxes_per_side + j])
            s.add_box((x0, 0.0, z0), (x0 + w, y1, z0 + w), ground)
    s.add_rect(1, 554.0, 123.0, 423.0, 147.0, 412.0, s.diffuse_light((7.0, 7.0, 7.0)))
    s.add_moving_sphere((400.0, 400.0, 200.0), (430.0, 400.0, 200.0), 0.0, 1.0, 50.0,
                        s.lambertian((0.7, 0.3, 0.1)))
    glass = s.dielectric(1.5)
    s.add_sphere((260.0, 150.0, 45.0), 50.0, glass)
    s.add_sphere((0.0, 150.0, 145.0), 50.0, s.metallic((0.8, 0.8, 0.9), 1.0))
    s.add_sphere((360.0, 150.0, 145.0), 70.0, glass)
    s.add_medium(0.2, s.isotropic((0.2, 0.4, 0.9)), sphere=((360.0, 150.0, 145.0), 70.0))
    s.add_medium(0.0001, s.isotropic((1.0, 1.0, 1.0)), sphere=((0.0, 0.0, 0.0), 5000.0))
    earth = s.add_texture(abi.RT_TEX_IMAGE, a=s.add_image(earth_image()))
    # the reference sphere is z-polar; the scene is y-up: rotate the pole onto +y
    s.add_gsphere(translate((400.0, 200.0, 400.0)) @ rotate([1.0, 0.0, 0.0], -90.0), 100.0,
                  material=s.lambertian(texture=earth))
    noise = s.add_texture(abi.RT_TEX_NOISE, scale=0.1)
    s.add_sphere((220.0, 280.0, 300.0), 80.0, s.lambertian(texture=noise))
    white = s.lambertian((0.73, 0.73, 0.73))
    c = pcg32.streams(2000, cube_count, 3).astype(np.float64) * 165.0
    m = translate((-100.0, 270.0, 395.0)) @ rotate_y(15.0)
    cw = c @ m[:3, :3].T + m[:3, 3]
    for p in cw:
        s.add_sphere(tuple(float(np.float32(v)) for v in p), 10.0, white)
    cam = _legacy_camera((478.0, 278.0, -600.0), (278.0, 278.0, 0.0), 40.0, shutter=(0.0, 1.0))
    return s, cam


def pbrt_test():
    """test/test.pbrt: unit glass sphere at the origin and a 40 x 40 checkered
    quad (two triangles, st coordinates) at z = -1 under a uniform sky and the file's distant
    light (a 3000 K black body scaled by 1.5, shining from (-30, 40, 100) towards pbrt's default
    target (0, 0, 1)), which frames rendered with RT_RENDER_DIRECT_LIGHT sample."""
    from .frontends import blackbody_rgb
    s = SceneData(background=(0.4, 0.45, 0.5))
    s.add_distant_light((-30.0, 40.0, 99.0), blackbody_rgb(3000.0, 1.5))
    s.add_gsphere(identity(), 1.0, material=s.dielectric(1.5))
    # pbrt's checkerboard: cells with even floor(u) + floor(v) take tex1 (0.1), odd ones tex2 (0.8)
    tex1, tex2 = s.solid((0.1, 0.1, 0.1)), s.solid((0.8, 0.8, 0.8))
    checks = s.add_texture(abi.RT_TEX_CHECKER_UV, a=tex2, b=tex1, scale=8.0)
    s.add_mesh(translate((0.0, 0.0, -1.0)),
               [[-20, -20, 0], [20, -20, 0], [20, 20, 0], [-20, 20, 0]], [0, 1, 2, 0, 2, 3],
               uv=[[0, 0], [1, 0], [1, 1], [0, 1]], material=s.lambertian(texture=checks))
    c2w = look_at((3.0, 4.0, 1.5), (0.5, 0.5, 0.0), (0.0, 0.0, 1.0))

    def cam(width, height):
        return perspective_camera(width, height, 45.0, c2w)

    return s, cam


def synthetic(grid=708, n_spheres=10_000_000, extent=2000.0, height=200.0):
    """Displaced triangle grid (2 * grid^2 triangles, heights from PCG32
    sequences 1<<32 + vertex) under a cloud of small spheres (centres and radii
    from sequences 2<<32 + sphere)."""
    s = SceneData(background=(0.6, 0.7, 0.9))
    grey = s.lambertian((0.6, 0.6, 0.6))
    nv = grid + 1
    hgt = pcg32.streams(1 << 32, nv * nv, 1)[:, 0].reshape(nv, nv) * np.float32(0.02 * extent)
    xs = np.linspace(-extent / 2, extent / 2, nv, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    P = np.stack([X, hgt, Z], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    tris = np.concatenate([np.stack([a, c, b], axis=2).reshape(-1, 3, 3),
                           np.stack([a, d, c], axis=2).reshape(-1, 3, 3)])
    s.add_triangles(tris, material=grey)
    if n_spheres:
        r = pcg32.streams(2 << 32, n_spheres, 4)
        centres = np.empty((n_spheres, 3), dtype=np.float32)
        centres[:, 0] = (r[:, 0] - 0.5) * extent
        centres[:, 1] = 0.02 * extent + 1.0 + r[:, 1] * height
        centres[:, 2] = (r[:, 2] - 0.5) * extent
        radii = (0.2 + 0.8 * r[:, 3]).astype(np.float32)
        s.add_sphere_block(centres, radii, s.lambertian((0.7, 0.5, 0.4)))
    c2w = look_at((0.0, 0.35 * extent, -0.75 * extent), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0),
                  right_handed=True)

    def cam(width, height):
        return perspective_camera(width, height, 40.0, c2w)

    return s, cam


BUILDERS = {"random": random_spheres, "cornell": cornell, "week": next_week, "pbrt": pbrt_test,
            "synthetic": synthetic}
