"""Scene front-ends: the legacy TOML schema of the reference's test scenes
(test/*.toml: settings, named materials and textures, [[world]] shapes with
Translate / RotateY wrappers, Volume, and the RandomBoxes / RandomCube /
SphereField generators) and the subset of the pbrt format that test/test.pbrt
uses (directive list: parsers/src/pbrt/mod.rs:600-640; transform stack:
parsers/src/pbrt/transform_set.rs:52-128).  Both produce a SceneData, a camera
factory and the render settings; nothing here touches the GPU.

The reference has no loader for the TOML files at this commit and its pbrt
parser stops at the camera block, so the meaning of each key is the evident one
(the scenes are those of the "Ray Tracing in One Weekend" books).  Generators
draw from PCG32 (geometry/src/rng.rs) with the fixed sequence numbers that
raytracer_amd.scenes uses, so a loaded file and the hand-built scene agree.
"""
import math
import re

import numpy as np

from . import abi, pcg32
from .scene import SceneData, identity, look_at, perspective_camera, rotate, rotate_y, scale, translate
from .scenes import _legacy_camera, earth_image

try:  # Python >= 3.11
    import tomllib as _toml
except ImportError:  # the image ships tomli
    import tomli as _toml


class SceneError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------
# TOML
# ---------------------------------------------------------------------------------------------
class _TomlBuilder:
    def __init__(self, doc):
        self.doc = doc
        st = doc.get("settings", {})
        self.settings = st
        self.s = SceneData(background=tuple(float(v) for v in st.get("background_colour", (0.0, 0.0, 0.0))))
        self.named_materials = {}
        self.shutter = None
        self.uses_noise = False

    # -- textures: RGB triple, name, or {kind=...} ------------------------------------------
    def texture(self, spec):
        s = self.s
        if isinstance(spec, str):
            table = self.doc.get("textures", {})
            if spec not in table:
                raise SceneError("unknown texture %r" % spec)
            return self.texture(table[spec])
        if isinstance(spec, (list, tuple)):
            return s.solid(tuple(float(v) for v in spec))
        kind = spec.get("kind")
        if kind == "Solid":
            return s.solid(tuple(float(v) for v in spec["albedo"]))
        if kind == "Checker":
            odd, even = self.texture(spec["odd"]), self.texture(spec["even"])
            return s.add_texture(abi.RT_TEX_CHECKER, a=odd, b=even, scale=float(spec.get("scale", 10.0)))
        if kind == "Image":
            # the only image the scenes name is the earth map (a decoded fixture; see scenes.earth_image)
            return s.add_texture(abi.RT_TEX_IMAGE, a=s.add_image(earth_image(spec.get("file_name", "earthmap.jpg"))))
        if kind == "Noise":
            self.uses_noise = True
            return s.add_texture(abi.RT_TEX_NOISE, scale=float(spec["scale"]))
        raise SceneError("unknown texture kind %r" % kind)

    @staticmethod
    def _is_rgb(spec):
        return isinstance(spec, (list, tuple))

    def material(self, spec):
        s = self.s
        if isinstance(spec, str):
            if spec not in self.named_materials:
                table = self.doc.get("materials", {})
                if spec not in table:
                    raise SceneError("unknown material %r" % spec)
                self.named_materials[spec] = self.material(table[spec])
            return self.named_materials[spec]
        kind = spec.get("kind")
        if kind == "Lambertian":
            a = spec["albedo"]
            return s.lambertian(tuple(map(float, a))) if self._is_rgb(a) else s.lambertian(texture=self.texture(a))
        if kind == "Metallic":
            return s.metallic(tuple(map(float, spec["albedo"])), float(spec.get("fuzz", 0.0)))
        if kind == "Dielectric":
            return s.dielectric(float(spec["refractive_index"]))
        if kind == "DiffuseLight":
            return s.diffuse_light(tuple(map(float, spec["albedo"])))
        raise SceneError("unknown material kind %r" % kind)

    def _has_image(self, mat_index):
        def walk(t):
            kind, _, a, b, _ = self.s.textures[t]
            if kind == abi.RT_TEX_IMAGE:
                return True
            return kind == abi.RT_TEX_CHECKER and (walk(a) or walk(b))
        return walk(self.s.materials[mat_index][1])

    # -- shapes -----------------------------------------------------------------------------
    def shape(self, node, m, material=None):
        """Adds `node` under the object-to-world matrix m (4x4 float64)."""
        s = self.s
        kind = node.get("shape")
        if "material" in node:
            material = self.material(node["material"])
        plain = np.array_equal(m, identity())
        if kind == "Translate":
            return self.shape(node["child"], m @ translate(tuple(map(float, node["offset"]))), material)
        if kind == "RotateY":
            return self.shape(node["child"], m @ rotate_y(float(node["angle"])), material)
        if kind == "Sphere":
            c = np.asarray(node["centre"], dtype=np.float64)
            r = float(node["radius"])
            cw = c @ m[:3, :3].T + m[:3, 3]
            if material is not None and self._has_image(material):
                # surface (u, v) matter: the reference sphere is z-polar, these scenes are y-up
                return s.add_gsphere(m @ translate(tuple(c)) @ rotate([1.0, 0.0, 0.0], -90.0), r, material=material)
            return s.add_sphere(tuple(float(np.float32(v)) for v in cw), r, material or 0)
        if kind == "MovingSphere":
            t0, t1 = float(node["time0"]), float(node["time1"])
            self.shutter = (t0, t1) if self.shutter is None else (min(self.shutter[0], t0), max(self.shutter[1], t1))
            c0 = np.asarray(node["centre0"], float) @ m[:3, :3].T + m[:3, 3]
            c1 = np.asarray(node["centre1"], float) @ m[:3, :3].T + m[:3, 3]
            return s.add_moving_sphere(tuple(c0), tuple(c1), t0, t1, float(node["radius"]), material or 0)
        if kind in ("YzRect", "XzRect", "XyRect"):
            axis = {"YzRect": 0, "XzRect": 1, "XyRect": 2}[kind]
            xf = 0 if plain else s.add_xform(m)
            return s.add_rect(axis, float(node["k"]), float(node["a0"]), float(node["a1"]), float(node["b0"]),
                              float(node["b1"]), material or 0, xform=xf)
        if kind == "Cube":
            xf = 0 if plain else s.add_xform(m)
            return s.add_box(tuple(map(float, node["minimum"])), tuple(map(float, node["maximum"])), material or 0,
                             xform=xf)
        if kind == "Volume":
            phase = s.isotropic(tuple(float(v) for v in node["texture"]))
            return self.volume(node["boundary"], m, float(node["density"]), phase)
        if kind == "RandomBoxes":
            n, w = int(node["boxes_per_side"]), float(node["box_size"])
            lo_h, hi_h = float(node["min_height"]), float(node["max_height"])
            h = pcg32.streams(1000, n * n, 1)[:, 0]
            xf = 0 if plain else s.add_xform(m)
            for i in range(n):
                for j in range(n):
                    x0, z0 = -0.5 * n * w + i * w, -0.5 * n * w + j * w
                    y1 = lo_h + hi_h * float(h[i * n + j])
                    s.add_box((x0, 0.0, z0), (x0 + w, y1, z0 + w), material or 0, xform=xf)
            return None
        if kind == "RandomCube":
            obj = node["object"]
            if obj.get("shape") != "Sphere":
                raise SceneError("RandomCube of %r is not supported" % obj.get("shape"))
            mat = self.material(obj["material"]) if "material" in obj else (material or 0)
            c = pcg32.streams(2000, int(node["count"]), 3).astype(np.float64) * float(node["size"])
            c = c + np.asarray(obj.get("centre", (0, 0, 0)), dtype=np.float64)
            cw = c @ m[:3, :3].T + m[:3, 3]
            for p in cw:
                s.add_sphere(tuple(float(np.float32(v)) for v in p), float(obj["radius"]), mat)
            return None
        if kind == "SphereField":
            return self.sphere_field()
        raise SceneError("unknown shape %r" % kind)

    def volume(self, boundary, m, density, phase):
        kind = boundary.get("shape")
        if kind == "Translate":
            return self.volume(boundary["child"], m @ translate(tuple(map(float, boundary["offset"]))), density, phase)
        if kind == "RotateY":
            return self.volume(boundary["child"], m @ rotate_y(float(boundary["angle"])), density, phase)
        if kind == "Sphere":
            c = np.asarray(boundary["centre"], float) @ m[:3, :3].T + m[:3, 3]
            return self.s.add_medium(density, phase, sphere=(tuple(c), float(boundary["radius"])))
        if kind == "Cube":
            xf = 0 if np.array_equal(m, identity()) else self.s.add_xform(m)
            return self.s.add_medium(density, phase, box=(tuple(map(float, boundary["minimum"])),
                                                         tuple(map(float, boundary["maximum"]))), xform=xf)
        raise SceneError("a volume boundary cannot be %r" % kind)

    def sphere_field(self):
        """22 x 22 jittered small spheres; the material of each is chosen by a uniform draw."""
        s = self.s
        rng = pcg32.Rng(1)
        glass = s.dielectric(1.5)
        for a in range(-11, 11):
            for b in range(-11, 11):
                choose = rng.uniform_float()
                c = (a + 0.9 * rng.uniform_float(), 0.2, b + 0.9 * rng.uniform_float())
                if math.dist(c, (4.0, 0.2, 0.0)) <= 0.9:
                    continue
                if choose < 0.8:
                    m = s.lambertian([rng.uniform_float() * rng.uniform_float() for _ in range(3)])
                elif choose < 0.95:
                    rgb = [0.5 + 0.5 * rng.uniform_float() for _ in range(3)]
                    m = s.metallic(rgb, 0.5 * rng.uniform_float())
                else:
                    m = glass
                s.add_sphere(c, 0.2, m)

    def build(self):
        for node in self.doc.get("world", []):
            self.shape(node, identity())
        if self.uses_noise:
            self.s.set_perlin()
        st = self.settings
        origin = tuple(map(float, st.get("origin", (0, 0, 0))))
        target = tuple(map(float, st.get("look_at", (0, 0, -1))))
        cam = _legacy_camera(origin, target, float(st.get("vfov", 90.0)),
                             up=tuple(map(float, st.get("up_direction", (0, 1, 0)))),
                             aperture=float(st.get("aperture", 0.0)),
                             focus=float(st["focus_distance"]) if "focus_distance" in st else None,
                             shutter=self.shutter or (0.0, 0.0))
        settings = {"width": int(st.get("width", 400)), "height": int(st.get("height", 225)),
                    "spp": int(st.get("samples_per_pixel", 100)), "max_depth": int(st.get("recursive_depth", 50)),
                    "gamma": float(st.get("gamma", 2.0))}
        return self.s, cam, settings


def load_toml(text):
    """Scene in the legacy TOML schema -> (SceneData, camera_factory(width, height), settings)."""
    return _TomlBuilder(_toml.loads(text)).build()


# ---------------------------------------------------------------------------------------------
# pbrt (the directives test/test.pbrt uses)
# ---------------------------------------------------------------------------------------------
_TOKEN = re.compile(r'"[^"]*"|\[|\]|[^\s\[\]"]+')


def _tokens(text):
    out = []
    for line in text.splitlines():
        line = line.split("#", 1)[0]
        out.extend(_TOKEN.findall(line))
    return out


def _is_directive(tok):
    return tok[0].isalpha()


def _params(toks, at):
    """"type name" value... pairs following a directive -> ({name: (type, values)}, next index)."""
    out = {}
    while at < len(toks) and toks[at].startswith('"') and " " in toks[at].strip('"'):
        typ, name = toks[at].strip('"').split()
        at += 1
        vals = []
        if toks[at] == "[":
            at += 1
            while toks[at] != "]":
                vals.append(toks[at])
                at += 1
            at += 1
        else:
            vals.append(toks[at])
            at += 1
        vals = [v.strip('"') if v.startswith('"') else float(v) for v in vals]
        out[name] = (typ, vals)
    return out, at


def blackbody_rgb(kelvin, scale=1.0):
    """RGB of pbrt's "blackbody L" [T scale]: Planck's law sampled every nanometre over 360-830 nm,
    normalised by its value at the Wien peak (which lies in the infrared for 3000 K), weighted
    with the CIE 1931 matching functions, Y normalised by the integral of y-bar, then XYZ -> linear
    sRGB primaries.  The matching functions are the multi-lobe Gaussian fit of Wyman, Sloan and
    Shirley (2013), not the reference's tables (data/src/lib.rs), so the last digits differ from
    a table-driven conversion by about a per cent."""
    lam = np.arange(360.0, 831.0)

    def planck(nm):
        m = nm * 1e-9
        h, c, kb = 6.62606957e-34, 299792458.0, 1.3806488e-23
        return (2.0 * h * c * c) / (m ** 5 * (np.exp((h * c) / (m * kb * kelvin)) - 1.0))

    v = planck(lam) / planck(2.8977721e-3 / kelvin * 1e9)

    def lobe(mu, s1, s2):
        return np.exp(-0.5 * ((lam - mu) / np.where(lam < mu, s1, s2)) ** 2)

    xb = 1.056 * lobe(599.8, 37.9, 31.0) + 0.362 * lobe(442.0, 16.0, 26.7) - 0.065 * lobe(501.1, 20.4, 26.2)
    yb = 0.821 * lobe(568.8, 46.9, 40.5) + 0.286 * lobe(530.9, 16.3, 31.1)
    zb = 1.217 * lobe(437.0, 11.8, 36.0) + 0.681 * lobe(459.0, 26.0, 13.8)
    x, y, z = ((v * b).sum() / yb.sum() for b in (xb, yb, zb))
    rgb = (3.240479 * x - 1.537150 * y - 0.498535 * z, -0.969256 * x + 1.875991 * y + 0.041556 * z,
           0.055648 * x - 0.204043 * y + 1.057311 * z)
    return tuple(float(np.float32(max(0.0, c) * scale)) for c in rgb)


def load_pbrt(text):
    """The subset of pbrt-v3 that test/test.pbrt needs -> (SceneData, camera_factory, settings).
    LightSource "infinite" becomes the background radiance; "distant" becomes a distant light
    (SceneData.add_distant_light) that frames rendered with RT_RENDER_DIRECT_LIGHT sample."""
    toks = _tokens(text)
    s = SceneData()
    ctm = identity()
    stack = []
    named_textures = {}
    material = [None]
    camera = {"c2w": identity(), "fov": 90.0}
    settings = {"width": 400, "height": 400, "spp": 16, "max_depth": 5, "gamma": 2.2}
    skipped = []

    def graphics_state():
        return (ctm.copy(), material[0])

    def current_material():
        if material[0] is None:
            material[0] = s.lambertian((0.5, 0.5, 0.5))  # pbrt's default matte
        return material[0]

    at = 0
    while at < len(toks):
        d = toks[at]
        at += 1
        if not _is_directive(d):
            raise SceneError("unexpected token %r" % d)
        if d == "LookAt":
            v = [float(x) for x in toks[at:at + 9]]
            at += 9
            ctm = ctm @ np.linalg.inv(look_at(v[0:3], v[3:6], v[6:9]))
        elif d == "Translate":
            v = [float(x) for x in toks[at:at + 3]]
            at += 3
            ctm = ctm @ translate(tuple(v))
        elif d == "Scale":
            v = [float(x) for x in toks[at:at + 3]]
            at += 3
            ctm = ctm @ scale(tuple(v))
        elif d == "Rotate":
            v = [float(x) for x in toks[at:at + 4]]
            at += 4
            ctm = ctm @ rotate(v[1:4], v[0])
        elif d in ("WorldBegin",):
            ctm = identity()
        elif d in ("WorldEnd",):
            pass
        elif d == "AttributeBegin":
            stack.append(graphics_state())
        elif d == "AttributeEnd":
            ctm, material[0] = stack.pop()
        else:
            kind = toks[at].strip('"')
            at += 1
            if d == "Texture":  # Texture "name" "spectrum" "checkerboard" ...
                name, cls = kind, toks[at + 1].strip('"')
                at += 2
                p, at = _params(toks, at)
                if cls != "checkerboard":
                    raise SceneError("texture class %r is not supported" % cls)
                t1 = s.solid(tuple(p.get("tex1", ("rgb", [1.0, 1.0, 1.0]))[1]))
                t2 = s.solid(tuple(p.get("tex2", ("rgb", [0.0, 0.0, 0.0]))[1]))
                us = p.get("uscale", ("float", [1.0]))[1][0]
                # cell (floor(u*s) + floor(v*s)) even -> tex1
                named_textures[name] = s.add_texture(abi.RT_TEX_CHECKER_UV, a=t2, b=t1, scale=float(us))
                continue
            p, at = _params(toks, at)
            if d == "Camera":
                if kind != "perspective":
                    raise SceneError("camera %r is not supported" % kind)
                camera["c2w"] = np.linalg.inv(ctm)
                camera["fov"] = float(p.get("fov", ("float", [90.0]))[1][0])
                ctm = identity()
            elif d == "Sampler":  # parsers/src/pbrt/mod.rs:486-520,688-696
                settings["sampler"] = kind
                if kind == "stratified":
                    settings["strata"] = (int(p.get("xsamples", ("integer", [4]))[1][0]),
                                          int(p.get("ysamples", ("integer", [4]))[1][0]))
                    jitter = p.get("jitter", ("bool", [True]))[1][0]
                    settings["jitter"] = jitter in (True, "true", '"true"')
                    settings["spp"] = settings["strata"][0] * settings["strata"][1]
                else:
                    settings["spp"] = int(p.get("pixelsamples", ("integer", [16]))[1][0])
            elif d == "Film":
                settings["width"] = int(p.get("xresolution", ("integer", [640]))[1][0])
                settings["height"] = int(p.get("yresolution", ("integer", [480]))[1][0])
            elif d == "Integrator":
                settings["max_depth"] = int(p.get("maxdepth", ("integer", [5]))[1][0])
            elif d == "LightSource":
                if kind == "infinite":
                    s.background = tuple(float(v) for v in p.get("L", ("rgb", [1.0, 1.0, 1.0]))[1])
                elif kind == "distant":
                    # the light shines from `from` towards `to`, both in the current transform
                    src = np.asarray(p.get("from", ("point", [0.0, 0.0, 0.0]))[1], dtype=np.float64)
                    dst = np.asarray(p.get("to", ("point", [0.0, 0.0, 1.0]))[1], dtype=np.float64)
                    towards = ctm[:3, :3] @ (src - dst)
                    typ, val = p.get("L", ("rgb", [1.0, 1.0, 1.0]))
                    rgb = blackbody_rgb(float(val[0]), float(val[1])) if typ == "blackbody" else tuple(float(v) for v in val)
                    sc = p.get("scale", ("rgb", [1.0, 1.0, 1.0]))[1]
                    sc = [float(v) for v in sc] * (3 if len(sc) == 1 else 1)
                    s.add_distant_light(towards, tuple(float(np.float32(a * b)) for a, b in zip(rgb, sc)))
                else:
                    skipped.append("LightSource " + kind)
            elif d == "Material":
                if kind == "glass":
                    material[0] = s.dielectric(float(p.get("eta", p.get("index", ("float", [1.5])))[1][0]))
                elif kind == "matte":
                    kd = p.get("Kd", ("rgb", [0.5, 0.5, 0.5]))
                    if kd[0] == "texture":
                        material[0] = s.lambertian(texture=named_textures[kd[1][0]])
                    else:
                        material[0] = s.lambertian(tuple(kd[1]))
                else:
                    raise SceneError("material %r is not supported" % kind)
            elif d == "Shape":
                if kind == "sphere":
                    r = float(p.get("radius", ("float", [1.0]))[1][0])
                    z = (float(p["zmin"][1][0]) if "zmin" in p else -r, float(p["zmax"][1][0]) if "zmax" in p else r)
                    s.add_gsphere(ctm, r, z=z, phi_max=float(p.get("phimax", ("float", [360.0]))[1][0]),
                                  material=current_material())
                elif kind == "trianglemesh":
                    pos = np.asarray(p["P"][1], dtype=np.float64).reshape(-1, 3)
                    idx = [int(v) for v in p["indices"][1]]
                    uv = p.get("st", p.get("uv"))
                    s.add_mesh(ctm, pos, idx, uv=None if uv is None else np.asarray(uv[1]).reshape(-1, 2),
                               normal=None if "N" not in p else np.asarray(p["N"][1]).reshape(-1, 3),
                               material=current_material())
                else:
                    raise SceneError("shape %r is not supported" % kind)
            else:
                skipped.append(d)
    settings["skipped"] = skipped
    c2w, fov = camera["c2w"], camera["fov"]

    def cam(width, height):
        return perspective_camera(width, height, fov, c2w)

    return s, cam, settings


def sampler_options(settings):
    """Keyword arguments for lib.render_opts (spp, sampler, sampler_arg) from a loaded file's
    Sampler directive (names as parsers/src/pbrt/mod.rs:688-696)."""
    kind = settings.get("sampler", "random")
    spp = int(settings.get("spp", 16))
    if kind == "halton":
        return dict(spp=spp, sampler=abi.RT_SAMPLER_HALTON)
    if kind == "sobol":
        return dict(spp=spp, sampler=abi.RT_SAMPLER_SOBOL)
    if kind in ("02sequence", "lowdiscrepancy"):
        return dict(spp=1 << max(0, spp - 1).bit_length(), sampler=abi.RT_SAMPLER_ZERO_TWO)  # round_up_pow_2
    if kind == "stratified":
        nx, ny = settings.get("strata", (4, 4))
        return dict(spp=nx * ny, sampler=abi.RT_SAMPLER_STRATIFIED,
                    sampler_arg=nx | (0 if settings.get("jitter", True) else abi.RT_STRATIFIED_NO_JITTER))
    if kind == "random":
        return dict(spp=spp, sampler=abi.RT_SAMPLER_RANDOM)
    raise SceneError("sampler %r is not supported" % kind)


def load(path):
    with open(path, "rb") as f:
        text = f.read().decode("utf-8")
    return load_pbrt(text) if str(path).endswith(".pbrt") else load_toml(text)
