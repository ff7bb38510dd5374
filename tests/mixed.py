"""A seeded scene that contains every primitive kind the path supports, with
instance transforms, clipped spheres, per-vertex normals and both medium
boundaries, for parity tests."""
import numpy as np

from raytracer_amd import abi
from raytracer_amd.scene import SceneData, rotate, rotate_y, translate


def mixed_scene(seed=1, n_each=40, with_normals=True, with_media=True):
    rng = np.random.default_rng(seed)
    s = SceneData(background=(0.5, 0.6, 0.8))
    s.set_perlin()
    mats = [s.lambertian((0.7, 0.6, 0.5)), s.metallic((0.8, 0.8, 0.9), 0.3), s.dielectric(1.5),
            s.diffuse_light((4.0, 4.0, 4.0)),
            s.lambertian(texture=s.add_texture(abi.RT_TEX_NOISE, scale=0.7)),
            s.lambertian(texture=s.add_texture(abi.RT_TEX_CHECKER, a=s.solid((0.1, 0.2, 0.1)),
                                               b=s.solid((0.9, 0.9, 0.9)), scale=2.0))]
    img = (rng.random((16, 32, 3)) * 255).astype(np.uint8)
    mats.append(s.lambertian(texture=s.add_texture(abi.RT_TEX_IMAGE, a=s.add_image(img))))
    iso = s.isotropic((0.3, 0.5, 0.8))

    def mat():
        return mats[int(rng.integers(len(mats)))]

    def pos(scale=20.0):
        return (rng.random(3) - 0.5) * scale

    for _ in range(n_each):
        r = float(rng.random() * 1.5 + 0.2)
        s.add_sphere(pos(), r if rng.random() > 0.1 else -r, mat())
    for _ in range(max(2, n_each // 4)):
        c0 = pos()
        s.add_moving_sphere(c0, c0 + (rng.random(3) - 0.5) * 3.0, 0.0, 1.0, float(rng.random() + 0.3), mat())
    tri = np.stack([pos() for _ in range(n_each)])[:, None, :] + (rng.random((n_each, 3, 3)) - 0.5) * 4.0
    uvs = rng.random((n_each, 3, 2))
    nrm = None
    if with_normals:
        g = np.cross(tri[:, 0] - tri[:, 2], tri[:, 1] - tri[:, 2])
        nrm = g[:, None, :] + (rng.random((n_each, 3, 3)) - 0.5) * 0.5 * np.linalg.norm(g, axis=1)[:, None, None]
    s.add_triangles(tri, uvs, nrm, mat())
    # a watertight pair sharing an edge, hit along the shared edge by tests
    quad = np.array([[[-3, -3, -9], [3, -3, -9], [3, 3, -9]], [[-3, -3, -9], [3, 3, -9], [-3, 3, -9]]], float)
    s.add_triangles(quad, None, None if not with_normals else np.tile([0, 0, 1.0], (2, 3, 1)), mats[0])
    for i in range(max(3, n_each // 4)):
        a, b = sorted(rng.random(2) * 8 - 4), sorted(rng.random(2) * 8 - 4)
        xf = 0
        if i % 2:
            xf = s.add_xform(translate(pos()) @ rotate(rng.random(3) + 0.1, float(rng.random() * 360)))
        s.add_rect(int(rng.integers(3)), float(rng.random() * 10 - 5), a[0], a[1] + 0.5, b[0], b[1] + 0.5,
                   mat(), xform=xf, reverse=bool(i % 3 == 0))
    for i in range(max(3, n_each // 4)):
        lo = pos()
        xf = 0 if i % 2 == 0 else s.add_xform(translate(pos(5.0)) @ rotate_y(float(rng.random() * 90)))
        s.add_box(lo, lo + rng.random(3) * 3 + 0.2, mat(), xform=xf)
    for i in range(max(3, n_each // 4)):
        r = float(rng.random() * 1.5 + 0.5)
        m = translate(pos()) @ rotate(rng.random(3) + 0.1, float(rng.random() * 360))
        if i % 3 == 0:
            m = m @ np.diag([1.0, -1.0, 1.0, 1.0])  # swaps handedness
        z = None if i % 2 == 0 else (-0.6 * r, 0.8 * r)
        phi = 360.0 if i % 4 < 2 else 200.0
        s.add_gsphere(m, r, z=z, phi_max=phi, material=mat(), reverse=bool(i % 5 == 0))
    if not with_media:
        return s
    s.add_medium(0.4, iso, sphere=(pos(), 3.0))
    s.add_medium(0.2, iso, box=((-2.0, -2.0, -2.0), (2.0, 3.0, 2.5)),
                 xform=s.add_xform(translate((4.0, 0.0, -3.0)) @ rotate_y(30.0)))
    s.add_medium(0.01, iso, sphere=((0.0, 0.0, 0.0), 40.0))
    return s
