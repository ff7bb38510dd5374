"""Scene front-ends against the reference's own scene files (tests/golden/scenes, copied
from its test/ directory): each file loads, and the three benchmark scenes come out as
the same primitives, materials, textures and camera as the hand-built ones in
raytracer_amd.scenes, so the BVH and every image are the same too."""
import os

import numpy as np
import pytest

from raytracer_amd import abi, frontends, lib, scenes
from tests import orc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def canonical(sd):
    """Primitive list with materials and textures resolved to values (index-free)."""
    fr = sd.freeze()
    k = fr._keep

    def tex(i):
        t = k["textures"][i]
        kind = int(t["kind"])
        if kind in (abi.RT_TEX_CHECKER, abi.RT_TEX_CHECKER_UV):
            return (kind, float(t["scale"]), tex(int(t["a"])), tex(int(t["b"])))
        if kind == abi.RT_TEX_IMAGE:
            img = k["images"][1][int(t["a"])]
            return (kind, img.shape, int(img.astype(np.uint64).sum()))
        if kind == abi.RT_TEX_NOISE:
            return (kind, float(t["scale"]))
        return (kind, tuple(float(v) for v in t["rgb"]))

    def mat(i):
        m = k["materials"][i]
        return (int(m["kind"]), float(m["fuzz"]), float(m["ior"]), tex(int(m["texture"])))

    tables = {abi.RT_SPHERE: "spheres", abi.RT_MOVING_SPHERE: "moving", abi.RT_RECT: "rects",
              abi.RT_BOX: "boxes", abi.RT_GSPHERE: "gspheres", abi.RT_MEDIUM: "media"}
    out = []
    for p in k["prims"]:
        kind = int(p["kind"]) & 0xff
        if kind == abi.RT_TRIANGLE:
            geom = k["tri_verts"][int(p["index"])].tobytes()
            if "tri_uvs" in k:
                geom += k["tri_uvs"][int(p["index"])].tobytes()
        elif kind == abi.RT_MEDIUM:
            m = k["media"][int(p["index"])]
            src = k["spheres"] if int(m["boundary_kind"]) == abi.RT_BOUNDARY_SPHERE else k["boxes"]
            xf = k["xforms"][int(m["xform"]) - 1].tobytes() if int(m["xform"]) else b""
            geom = src[int(m["boundary_index"])].tobytes() + xf + np.float32(m["density"]).tobytes()
        else:
            geom = k[tables[kind]][int(p["index"])].tobytes()
        xf = k["xforms"][int(p["xform"]) - 1].tobytes() if int(p["xform"]) else b""
        out.append((int(p["kind"]), geom, xf, mat(int(p["material"]))))
    lights = k["distant_lights"].tobytes() if "distant_lights" in k else b""
    return out, (tuple(fr.desc.background), lights)


@pytest.mark.parametrize("name", ["week", "cornell", "random", "globe", "perlin", "scene", "camera"])
def test_toml_files_load(name):
    sd, cam, st = frontends.load(os.path.join(HERE, name + ".toml"))
    assert sd.n_prims() > 0 and st["width"] > 0
    o = orc.OracleScene(sd.freeze())
    assert o.info().n_nodes >= 1
    img, stats = o.render(cam(24, 16), orc.render_opts(24, 16, 1, 6, seed=1))
    assert np.isfinite(img).all() and stats.segments >= 24 * 16


@pytest.mark.parametrize("name,builder", [("week.toml", "week"), ("cornell.toml", "cornell"),
                                          ("random.toml", "random"), ("test.pbrt", "pbrt")])
def test_loaded_scene_equals_hand_built(name, builder):
    sd, cam, st = frontends.load(os.path.join(HERE, name))
    ref_sd, ref_cam = scenes.BUILDERS[builder]()
    got, want = canonical(sd), canonical(ref_sd)
    assert len(got[0]) == len(want[0])
    for i, (a, b) in enumerate(zip(got[0], want[0])):
        assert a == b, (name, i, a[0], a[3], b[3])
    assert got[1] == want[1]
    for w, h in ((64, 48), (48, 64)):
        a = np.frombuffer(bytes(cam(w, h)), dtype=np.float32)
        b = np.frombuffer(bytes(ref_cam(w, h)), dtype=np.float32)
        # pbrt: the camera matrix is the inverse of the accumulated world-to-camera transform
        assert np.allclose(a, b, rtol=1e-6, atol=1e-6) if name.endswith(".pbrt") else a.tobytes() == b.tobytes()
    # same primitives in the same order -> the same tree
    g, r = lib.Scene(sd.freeze()), lib.Scene(ref_sd.freeze())
    assert g.nodes().tobytes() == r.nodes().tobytes()


def test_settings_follow_the_files():
    _, _, st = frontends.load(os.path.join(HERE, "week.toml"))
    assert (st["width"], st["height"], st["spp"], st["max_depth"]) == (800, 800, 10000, 50)
    _, _, st = frontends.load(os.path.join(HERE, "test.pbrt"))
    assert (st["width"], st["height"], st["spp"], st["sampler"]) == (400, 400, 128, "halton")
    assert st["skipped"] == []  # every directive of the file is modelled
    sd, _, _ = frontends.load(os.path.join(HERE, "test.pbrt"))
    (w, rgb, _), = sd.distant  # LightSource "distant" "point from" [-30 40 100] "blackbody L" [3000 1.5]
    towards = np.array([-30.0, 40.0, 100.0]) - np.array([0.0, 0.0, 1.0])  # "to" defaults to (0, 0, 1) in pbrt
    assert np.allclose(w, towards / np.linalg.norm(towards), atol=1e-7)
    assert rgb[0] > rgb[1] > rgb[2] > 0 and 0.9 < rgb[0] < 1.3  # a warm white, 1.5 x a 3000 K black body


def test_blackbody_colour():
    """The conversion behind "blackbody L": chromaticity of Planck's law against published CIE 1931
    values (x, y) = (0.4369, 0.4041) at 3000 K and (0.3135, 0.3237) at 6500 K, and the scale."""
    m = np.linalg.inv(np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556],
                                [0.055648, -0.204043, 1.057311]]))
    for kelvin, xy in ((3000.0, (0.4369, 0.4041)), (6500.0, (0.3135, 0.3237))):
        X, Y, Z = m @ np.array(frontends.blackbody_rgb(kelvin), dtype=np.float64)
        # negative primaries are clipped at 3000 K, which moves the point a little
        assert abs(X / (X + Y + Z) - xy[0]) < 0.012 and abs(Y / (X + Y + Z) - xy[1]) < 0.012, (kelvin, X, Y, Z)
    a, b = frontends.blackbody_rgb(3000.0, 1.0), frontends.blackbody_rgb(3000.0, 1.5)
    assert np.allclose(np.array(a) * 1.5, b, rtol=1e-6)


def test_errors_name_the_problem():
    with pytest.raises(frontends.SceneError, match="unknown shape"):
        frontends.load_toml('[[world]]\nshape = "Torus"\n')
    with pytest.raises(frontends.SceneError, match="unknown material"):
        frontends.load_toml('[[world]]\nshape = "Sphere"\ncentre=[0,0,0]\nradius=1\nmaterial="nope"\n')
    with pytest.raises(frontends.SceneError, match="not supported"):
        frontends.load_pbrt('WorldBegin\nShape "cone"\nWorldEnd\n')


def test_pbrt_sampler_directives():
    """Sampler names of parsers/src/pbrt/mod.rs:688-696 -> render options."""
    from raytracer_amd import abi
    base = 'Camera "perspective" "float fov" 45\n%s\nWorldBegin\nShape "sphere" "float radius" 1\nWorldEnd\n'
    cases = [
        ('Sampler "halton" "integer pixelsamples" 128', dict(spp=128, sampler=abi.RT_SAMPLER_HALTON)),
        ('Sampler "sobol" "integer pixelsamples" 48', dict(spp=48, sampler=abi.RT_SAMPLER_SOBOL)),
        ('Sampler "02sequence" "integer pixelsamples" 48', dict(spp=64, sampler=abi.RT_SAMPLER_ZERO_TWO)),
        ('Sampler "lowdiscrepancy" "integer pixelsamples" 64', dict(spp=64, sampler=abi.RT_SAMPLER_ZERO_TWO)),
        ('Sampler "stratified" "integer xsamples" 8 "integer ysamples" 4',
         dict(spp=32, sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=8)),
        ('Sampler "stratified" "bool jitter" "false" "integer xsamples" [2] "integer ysamples" [3]',
         dict(spp=6, sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=2 | abi.RT_STRATIFIED_NO_JITTER)),
        ('Sampler "random" "integer pixelsamples" 5', dict(spp=5, sampler=abi.RT_SAMPLER_RANDOM)),
    ]
    for text, want in cases:
        _, _, st = frontends.load_pbrt(base % text)
        assert frontends.sampler_options(st) == want
    _, _, st = frontends.load_pbrt(base % 'Sampler "maxmindist" "integer pixelsamples" 16')
    with pytest.raises(frontends.SceneError):
        frontends.sampler_options(st)
