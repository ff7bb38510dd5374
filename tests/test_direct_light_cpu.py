"""Shadow rays (Interaction::spawn_ray_to / spawn_ray_intersect, shapes/src/interaction.rs:
102-133) and the direct-lighting option that sends them through the any-hit traversal:
known answers for the ray, and the oracle's estimator against the closed-form irradiance of
a rectangular light."""
import numpy as np

from raytracer_amd import abi
from raytracer_amd.scene import SceneData, look_at, perspective_camera
from tests import orc


def test_spawn_ray_to_known_answers():
    p, n = np.float32([1.0, 2.0, 3.0]), np.float32([0.0, 1.0, 0.0])
    err = np.float32([1e-4, 2e-4, 3e-4])
    up, down = np.float32([1.5, 7.0, 3.0]), np.float32([1.5, -7.0, 3.0])
    for target, side in ((up, 1.0), (down, -1.0)):
        r = orc.spawn_ray_to(p, err, n, target, time=0.25)
        # offset_ray_origin (geometry/src/lib.rs:129-151): |n| . err along the normal, towards the
        # target's side, then one ulp further
        off = np.float32(2e-4) * side
        want_y = np.nextafter(np.float32(p[1] + off), np.float32(np.inf * side), dtype=np.float32)
        assert r["o"][0] == p[0] and r["o"][2] == p[2] and r["o"][1] == want_y
        # direction = target - origin, not normalised; the ray ends just short of the target
        assert (r["d"] == target - r["o"]).all()
        assert r["t_max"] == np.float32(1.0) - np.float32(0.0001) and r["time"] == np.float32(0.25)
        end = r["o"] + r["t_max"] * r["d"]
        assert np.abs(end - target).max() < 1e-3 and np.linalg.norm(end - r["o"]) < np.linalg.norm(target - r["o"])


def floor_under_light(albedo=0.6, emit=5.0):
    s = SceneData()
    s.add_rect(1, 0.0, -50.0, 50.0, -50.0, 50.0, s.lambertian((albedo, albedo, albedo)))
    s.add_rect(1, 3.0, -1.0, 1.0, -0.5, 0.5, s.diffuse_light((emit, emit, emit)))
    cam = perspective_camera(32, 32, 20.0, look_at((0.0, 2.0, -6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    return s, cam


def test_direct_light_matches_the_form_factor_of_a_rectangle():
    """A Lambertian floor under a rectangular light, black sky, paths of two segments: a
    pixel's radiance is the direct light albedo / pi * Le * integral of cos cos' / d^2 over the
    light at the floor point it sees (the second segment adds nothing: a light reached right
    after its sampling is skipped, everything else is black)."""
    albedo, emit = 0.6, 5.0
    s, cam = floor_under_light(albedo, emit)
    fr = s.freeze()
    o = orc.OracleScene(fr)
    w = h = 32
    spp = 256
    img, st = o.render(cam, orc.render_opts(w, h, spp, 2, seed=1, flags=abi.RT_RENDER_DIRECT_LIGHT))
    assert st.occlusion_rays > 0 and st.segments > st.occlusion_rays
    # float64 quadrature at the floor points the pixel centres see
    rays = orc.generate_rays(cam, orc.render_opts(w, h, 1, sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=1 | abi.RT_STRATIFIED_NO_JITTER), 0, w * h)
    hits = o.intersect(rays)
    on_floor = hits["prim"] == 0
    assert on_floor.mean() > 0.8
    P = rays["o"].astype(np.float64) + hits["t"][:, None].astype(np.float64) * rays["d"].astype(np.float64)
    gx, gz = np.meshgrid(np.linspace(-1, 1, 64, endpoint=False) + 1 / 64, np.linspace(-0.5, 0.5, 32, endpoint=False) + 1 / 64)
    Q = np.stack([gx.ravel(), np.full(gx.size, 3.0), gz.ravel()], axis=1)
    d = Q[None, :, :] - P[:, None, :]
    d2 = (d * d).sum(-1)
    cos_floor, cos_light = d[..., 1] / np.sqrt(d2), d[..., 1] / np.sqrt(d2)
    E = (cos_floor * cos_light / d2).mean(axis=1) * 2.0  # light area 2 x 1
    want = albedo / np.pi * emit * E
    got = img[..., 0].reshape(-1)
    m = on_floor & (want > 0.02 * want.max())
    # 256 samples per pixel: compare in aggregate and per pixel with a noise allowance
    assert abs(got[m].sum() / want[m].sum() - 1.0) < 0.01
    assert np.abs(got[m] / want[m] - 1.0).max() < 0.25
    # paths of one segment: the shadow rays of the camera vertex are all there is
    img1, st1 = o.render(cam, orc.render_opts(w, h, spp, 1, seed=1, flags=abi.RT_RENDER_DIRECT_LIGHT))
    assert img1.tobytes() == img.tobytes() and st1.occlusion_rays == st.occlusion_rays
    # without lights the flag changes nothing
    s2 = SceneData(background=(0.5, 0.7, 1.0))
    s2.add_sphere((0, 0, 0), 1.0, s2.lambertian((0.5, 0.5, 0.5)))
    o2 = orc.OracleScene(s2.freeze())
    a, sa = o2.render(cam, orc.render_opts(w, h, 2, 8, seed=3))
    b, sb = o2.render(cam, orc.render_opts(w, h, 2, 8, seed=3, flags=abi.RT_RENDER_DIRECT_LIGHT))
    assert a.tobytes() == b.tobytes() and sa.segments == sb.segments and sb.occlusion_rays == 0


def test_direct_light_converges_faster_than_the_default_estimator(scene_cache):
    """Cornell box: same scene, same sample count -- sampling the light gives a far less noisy
    image (the default estimator only gains light when a path happens to hit the small lamp)."""
    _, cam, fr = scene_cache("cornell")
    o = orc.OracleScene(fr)
    w = h = 48
    c = cam(w, h)
    ref, _ = o.render(c, orc.render_opts(w, h, 64, 12, seed=9, flags=abi.RT_RENDER_DIRECT_LIGHT))
    a, sa = o.render(c, orc.render_opts(w, h, 8, 12, seed=1, flags=abi.RT_RENDER_DIRECT_LIGHT))
    b, sb = o.render(c, orc.render_opts(w, h, 8, 12, seed=1))
    assert sa.occlusion_rays > 0 and sb.occlusion_rays == 0
    err_direct = np.sqrt(((a[..., :3] - ref[..., :3]) ** 2).mean())
    plain_ref, _ = o.render(c, orc.render_opts(w, h, 64, 12, seed=9))
    err_plain = np.sqrt(((b[..., :3] - plain_ref[..., :3]) ** 2).mean())
    assert err_direct < 0.6 * err_plain, (err_direct, err_plain)  # about half; 64-spp references are noisy themselves


def sphere_and_rect_lights(rect_emit=4.0, sphere_emit=(3.0, 2.0, 1.0)):
    """A Lambertian floor under a rectangular light (sampled by the flag) and an emissive
    sphere (reached by scattering only), black sky."""
    s = SceneData()
    s.add_rect(1, 0.0, -50.0, 50.0, -50.0, 50.0, s.lambertian((0.7, 0.7, 0.7)))
    s.add_rect(1, 3.0, -1.0, 1.0, -0.5, 0.5, s.diffuse_light((rect_emit, rect_emit, rect_emit)))
    s.add_sphere((1.5, 1.0, 0.5), 0.6, s.diffuse_light(sphere_emit))
    cam = perspective_camera(24, 24, 25.0, look_at((0.0, 2.0, -6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    return s, cam


def test_emitters_outside_the_light_list_keep_their_emission():
    """Only the lights the flag samples may be skipped at the next vertex.  An emissive sphere is
    not in the list: what it adds to the floor with the flag (frame with the sphere lit minus the
    same frame, same draws, with the sphere black) must be what it adds without the flag (the
    default estimator with the rectangle black), up to noise.  Before the rule was narrowed that
    share was lost with the flag."""
    w = h = 24
    flag = abi.RT_RENDER_DIRECT_LIGHT

    def render(rect_emit, sphere_emit, spp, seed, flags):
        s, cam = sphere_and_rect_lights(rect_emit, sphere_emit)
        return orc.OracleScene(s.freeze()).render(cam, orc.render_opts(w, h, spp, 3, seed=seed, flags=flags))

    lit, st = render(4.0, (3.0, 2.0, 1.0), 1024, 5, flag)
    dark, _ = render(4.0, (0.0, 0.0, 0.0), 1024, 5, flag)
    alone, sa = render(0.0, (3.0, 2.0, 1.0), 1024, 6, 0)
    assert st.occlusion_rays > 0 and sa.occlusion_rays == 0
    floor = lit[..., :3].sum(-1) < 2.0  # not the lights themselves
    share = (lit - dark)[floor][:, :3].mean(0)
    want = alone[floor][:, :3].mean(0)
    assert want.min() > 0.003 and np.abs(share / want - 1.0).max() < 0.05, (share, want)


def floor_under_sun(albedo=(0.6, 0.5, 0.4), sun=(2.0, 1.5, 1.0), towards=(1.0, 2.0, -0.5), ball=True, lamp=False,
                    lamp_emit=5.0):
    """A Lambertian floor (y = 0) under a distant light, black sky; optionally a ball that casts
    a shadow and a rectangular lamp (then the flag chooses between two lights)."""
    s = SceneData()
    s.add_rect(1, 0.0, -50.0, 50.0, -50.0, 50.0, s.lambertian(albedo))
    if ball:
        s.add_sphere((0.0, 1.0, 0.0), 0.5, s.lambertian((0.5, 0.5, 0.5)))
    if lamp:
        s.add_rect(1, 3.0, -1.0, 1.0, -0.5, 0.5, s.diffuse_light((lamp_emit, lamp_emit, lamp_emit)))
    s.add_distant_light(towards, sun)
    cam = perspective_camera(32, 32, 30.0, look_at((0.0, 4.0, -8.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    return s, cam


def test_distant_light_closed_form():
    """LightSource "distant" (test/test.pbrt:16-17) under RT_RENDER_DIRECT_LIGHT, paths of one
    segment, black sky: a floor pixel in the sun holds albedo / pi * radiance * cos(angle between
    the floor's normal and the light) for every sample, exactly the same number; a pixel in the
    ball's shadow holds nothing.  Where the shadow falls follows from the geometry: the floor point
    p is shaded when the line p + s w passes the ball's centre closer than its radius."""
    albedo, sun, towards = (0.6, 0.5, 0.4), (2.0, 1.5, 1.0), np.array([1.0, 2.0, -0.5])
    s, cam = floor_under_sun(albedo, sun, towards)
    o = orc.OracleScene(s.freeze())
    w = h = 32
    opts = dict(sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=1 | abi.RT_STRATIFIED_NO_JITTER)
    img, st = o.render(cam, orc.render_opts(w, h, 1, 1, seed=1, flags=abi.RT_RENDER_DIRECT_LIGHT, **opts))
    rays = orc.generate_rays(cam, orc.render_opts(w, h, 1, **opts), 0, w * h)
    hits = o.intersect(rays)
    floor = hits["prim"] == 0
    assert floor.sum() > 700 and st.occlusion_rays >= floor.sum()
    P = rays["o"].astype(np.float64) + hits["t"][:, None].astype(np.float64) * rays["d"].astype(np.float64)
    wdir = towards / np.linalg.norm(towards)
    oc = P - np.array([0.0, 1.0, 0.0])
    along = oc @ wdir
    dist = np.sqrt(np.maximum((oc * oc).sum(1) - along ** 2, 0.0))
    in_shadow = floor & (dist < 0.5 - 1e-3) & (along < 0)
    in_sun = floor & (dist > 0.5 + 1e-3)
    assert in_shadow.sum() > 10 and in_sun.sum() > 600
    got = img.reshape(-1, 4)[:, :3]
    want = np.array(albedo) / np.pi * np.array(sun) * wdir[1]
    assert np.abs(got[in_sun] / want - 1.0).max() < 1e-5
    assert (got[in_shadow] == 0).all()
    # without the flag the light adds nothing: no path reaches it by scattering
    plain, pst = o.render(cam, orc.render_opts(w, h, 1, 4, seed=1, **opts))
    assert pst.occlusion_rays == 0 and (plain[..., :3] == 0).all()
    # the far side of a surface gets nothing: a light below the floor
    s2, cam2 = floor_under_sun(albedo, sun, (1.0, -2.0, -0.5), ball=False)
    below, bst = orc.OracleScene(s2.freeze()).render(cam2, orc.render_opts(w, h, 1, 1, seed=1, flags=abi.RT_RENDER_DIRECT_LIGHT, **opts))
    assert bst.occlusion_rays == 0 and (below[..., :3] == 0).all()


def test_distant_light_shares_the_choice_with_the_lamps():
    """One light is chosen uniformly among rectangles and distant lights and weighted by their
    number: over many samples the floor receives lamp + sun, each as when the other is black
    (the lamp stays in place when it is black: it also casts a shadow)."""
    w = h = 32
    flag = abi.RT_RENDER_DIRECT_LIGHT

    def render(**kw):
        s, cam = floor_under_sun(ball=False, **kw)
        return orc.OracleScene(s.freeze()).render(cam, orc.render_opts(w, h, 256, 1, seed=2, flags=flag))[0][..., :3]

    both = render(lamp=True)
    sun_only = render(lamp=True, lamp_emit=0.0)
    lamp_only = render(lamp=True, sun=(0.0, 0.0, 0.0))
    m = (lamp_only.sum(-1) > 0.05) & (lamp_only.sum(-1) < 2.0)  # floor pixels the lamp lights, not the lamp itself
    assert m.sum() > 50
    assert abs(both[m].sum() / (sun_only[m].sum() + lamp_only[m].sum()) - 1.0) < 0.02
