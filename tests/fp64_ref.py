"""An independent float64 ray/primitive intersector (numpy), written from the geometry of
each shape and not from the oracle's or the kernels' code: closest hit by brute force over
all primitives, with the hit distance, the caller's primitive index, the geometric normal
and the surface (u, v) under the conventions of include/rt355.h.

Used by the CPU tests to check the oracle and by the GPU tests to check the HIP path's
output directly (that comparison does not pass through oracle/ at all).

Media (a constant density inside a sphere or a box; no reference code, the scene files' Volume):
the free-flight variate is a hash of the ray's 32-bit words and the primitive's index -- integer
arithmetic, restated here bit for bit in numpy uint32 -- turned into a number u in (0, 1); the
ray scatters after travelling -ln(u) / density inside the boundary, if it is still inside by
then.  Everything after the hash is float64 here: the boundary's interval from the closed-form
sphere roots or the slabs of the box, the logarithm, the distance along the ray.

Tolerances.  Planar primitives reproduce float64 to within 1e-5 (t relative, normal and
(u, v) absolute).  A sphere does not, for any fp32 quadratic: the discriminant b^2 - 4ac
cancels when the origin is far from a small sphere or the ray grazes it, and the error of t
(hence of the hit point along the ray, of the normal and of (u, v)) grows like
eps * kappa with kappa = |o - c|^2 / (r^2 cos(incidence)).  The reference's own sphere has the
same conditioning (it only bounds the error, float.rs:182-204, and re-projects the point onto
the sphere, sphere.rs:72).  So each sphere hit carries its kappa and the bound it is checked
against is  1e-5 + 16 eps kappa  (eps = 2^-23; measured worst case 5.3 eps kappa); hits with
kappa < 32 are additionally held to the plain 1e-5.  The bound on t also carries the
resolution of the ray's own fp32 coordinates, 2 eps max|o| / |d| (an origin 0.005 in front of a
plane at coordinate 14 cannot know t to 1e-5 relative), divided by the cosine of incidence
and taken in object space for a transformed primitive.

Conventions (the reference's, shapes/src/sphere.rs:110-136, triangle.rs:204-345):
  sphere         z-polar: phi = atan2(y, x) in [0, 2 pi), u = phi / phi_max,
                 theta = acos(z / r), v = (theta - theta_min) / (theta_max - theta_min),
                 theta_min = acos(z_min / r), theta_max = acos(z_max / r); a full sphere has
                 theta_min = pi, theta_max = 0.  Normal = normalised dp/du x dp/dv, negated when
                 reverse_orientation differs from "the transform swaps handedness"; for the
                 translation-only spheres n = (p - c) / r with the radius' sign.
  triangle       barycentric blend of the vertex uvs (default (0,0), (1,0), (0,1)); geometric
                 normal (p0 - p2) x (p1 - p2), turned to the side of the interpolated vertex
                 normal when the mesh has normals, else negated by reverse_orientation.
  rect / box     u, v = position along the two in-plane axes relative to the extent; normal =
                 +axis (rect; negated by reverse) or the outward face normal (box: entry face,
                 or exit face when the origin is inside).
"""
import numpy as np

from raytracer_amd import abi


EPS = 2.0 ** -23  # f32::EPSILON, the reference's machine epsilon (geometry/src/number.rs:205)


def ray_words(rays):
    """The rays' origin and direction as the 32-bit words the medium's hash reads: (n, 6) uint32."""
    o = np.ascontiguousarray(rays["o"], dtype=np.float32).view(np.uint32)
    d = np.ascontiguousarray(rays["d"], dtype=np.float32).view(np.uint32)
    return np.concatenate([o, d], axis=1)


def medium_variate(words, prim):
    """u in (0, 1) for each ray: multiply-xor over direction z, y, x and origin z, y of the ray
    (constant 0x9E3779B1, starting from the primitive index), then one permuted-congruential
    mixing round over the origin's x; the top 24 bits, centred in their cell."""
    k = np.uint32(0x9E3779B1)
    ox, oy, oz, dx, dy, dz = (words[:, i] for i in range(6))
    with np.errstate(over="ignore"):
        h = (np.uint32(prim) ^ dz) * k
        for w in (dy, dx, oz, oy):
            h = (h ^ w) * k
        v = h ^ ox
        st = v * np.uint32(747796405) + np.uint32(2891336453)
        w = ((st >> ((st >> np.uint32(28)) + np.uint32(4))) ^ st) * np.uint32(277803737)
        h = (w >> np.uint32(22)) ^ w
    return (h >> np.uint32(8)).astype(np.float64) * 2.0 ** -24 + 2.0 ** -25


def _m34(a):
    return np.asarray(a, dtype=np.float64).reshape(3, 4)


def _apply_point(m, p):
    return p @ m[:, :3].T + m[:, 3]


def _apply_vector(m, v):
    return v @ m[:, :3].T


def _apply_normal(inv, n):  # inverse transpose
    return n @ inv[:, :3]


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


class Fp64Scene:
    """Reads the frozen description (the very arrays handed to rt_scene_create)."""

    def __init__(self, frozen):
        k = frozen._keep
        self.k = k
        self.prims = k["prims"]
        self.n_media = int((self.prims["kind"] & 0xFF == abi.RT_MEDIUM).sum())
        self._object_coord = None  # largest object-space coordinate of the primitive just tested

    def closest(self, rays, prim_mask=None):
        """-> dict(t, prim, n, u, v) ; prim = -1 for a miss.  rays: abi.ray_dtype array."""
        O = rays["o"].astype(np.float64)
        D = rays["d"].astype(np.float64)
        tmax = rays["t_max"].astype(np.float64)
        time = rays["time"].astype(np.float64)
        return self.closest_od(O, D, tmax, time, ray_words(rays))

    def closest_od(self, O, D, tmax, time, words=None):
        """words: the 32-bit words of the rays as the caller holds them (ray_words); a medium's
        variate is a function of those, also when O and D are perturbed copies."""
        n = len(O)
        best = dict(t=np.full(n, np.inf), prim=np.full(n, -1, dtype=np.int64), n=np.zeros((n, 3)),
                    u=np.zeros(n), v=np.zeros(n), kappa=np.ones(n), radius=np.zeros(n),
                    t_second=np.full(n, np.inf))
        dlen = np.linalg.norm(D, axis=1)
        best["t_resolution"] = np.zeros(n)
        world_coord = np.abs(O).max(axis=1)
        k = self.k
        with np.errstate(all="ignore"):
            for pid, pr in enumerate(self.prims):
                kind = int(pr["kind"]) & 0xFF
                reverse = bool(int(pr["kind"]) & abi.RT_PRIM_REVERSE)
                idx, xf = int(pr["index"]), int(pr["xform"])
                if kind == abi.RT_SPHERE:
                    s = k["spheres"][idx]
                    res = self._sphere(O, D, np.float64(s["c"]), float(s["r"]))
                elif kind == abi.RT_MOVING_SPHERE:
                    m = k["moving"][idx]
                    c0, c1 = np.float64(m["c0"]), np.float64(m["c1"])
                    c = c0 + ((time - float(m["time0"])) / (float(m["time1"]) - float(m["time0"])))[:, None] * (c1 - c0)
                    res = self._sphere(O, D, c, float(m["r"]))
                elif kind == abi.RT_TRIANGLE:
                    res = self._triangle(O, D, idx, reverse)
                elif kind == abi.RT_RECT:
                    res = self._rect(O, D, k["rects"][idx], xf, reverse)
                elif kind == abi.RT_BOX:
                    res = self._box(O, D, k["boxes"][idx], xf)
                elif kind == abi.RT_GSPHERE:
                    res = self._gsphere(O, D, k["gspheres"][idx], reverse)
                else:
                    assert words is not None, "media need the rays' words (ray_words)"
                    res = self._medium(O, D, tmax, k["media"][idx], medium_variate(words, pid))
                t, ok, nrm, u, v = res[:5]
                valid = ok & (t > 0) & (t <= tmax)
                ok = valid & (t < best["t"])
                # the runner-up: the old winner where this one wins, else this one if it is valid
                best["t_second"] = np.where(ok, best["t"], np.where(valid, np.minimum(best["t_second"], t),
                                                                   best["t_second"]))
                if len(res) > 7:  # a medium: conditioning and incidence of its boundary where the ray enters
                    kappa, radius, cos_entry = res[5], res[6], res[7]
                elif len(res) > 5:  # a sphere: conditioning of its quadratic for this ray (see module text)
                    dist_sq, radius = res[5], res[6]
                    cos_i = np.abs((nrm * D).sum(1)) / dlen
                    kappa = np.maximum(1.0, dist_sq / (radius * radius * np.maximum(cos_i, 1e-12)))
                else:
                    kappa, radius = 1.0, 0.0
                # what fp32 coordinates of this size can resolve along the ray: eps * coordinate /
                # (|d| cos(incidence)); in object space for a transformed primitive
                coord = np.maximum(world_coord, self._object_coord) if self._object_coord is not None else world_coord
                cos_any = np.maximum(np.abs((nrm * D).sum(1)) / (dlen * np.linalg.norm(nrm, axis=1)), 1e-3)
                if len(res) > 7:
                    cos_any = np.maximum(cos_entry, 1e-3)
                best["t_resolution"] = np.where(ok, EPS * coord / (dlen * cos_any), best["t_resolution"])
                self._object_coord = None
                best["kappa"] = np.where(ok, kappa, best["kappa"])
                best["radius"] = np.where(ok, radius, best["radius"])
                best["t"] = np.where(ok, t, best["t"])
                best["prim"] = np.where(ok, pid, best["prim"])
                best["n"] = np.where(ok[:, None], nrm, best["n"])
                best["u"] = np.where(ok, u, best["u"])
                best["v"] = np.where(ok, v, best["v"])
        return best

    # -- shapes ------------------------------------------------------------------------------
    @staticmethod
    def _sphere_t(oc, D, r, tmax=None):
        a = (D * D).sum(1)
        hb = (oc * D).sum(1)
        c = (oc * oc).sum(1) - r * r
        disc = hb * hb - a * c
        sq = np.sqrt(np.maximum(disc, 0.0))
        # the numerically stable pair of roots
        q = -(hb + np.where(hb >= 0, sq, -sq))
        ra, rb = q / a, c / q
        t0, t1 = np.minimum(ra, rb), np.maximum(ra, rb)
        return t0, t1, disc >= 0

    def _sphere(self, O, D, c, r):
        oc = O - c
        t0, t1, ok = self._sphere_t(oc, D, abs(r))
        t = np.where(t0 > 0, t0, t1)
        p = oc + D * t[:, None]
        p = p * (abs(r) / np.linalg.norm(p, axis=1))[:, None]
        phi = np.arctan2(p[:, 1], p[:, 0])
        phi = np.where(phi < 0, phi + 2 * np.pi, phi)
        theta = np.arccos(np.clip(p[:, 2] / abs(r), -1, 1))
        return t, ok, p / r, phi / (2 * np.pi), 1.0 - theta / np.pi, (oc * oc).sum(1), abs(r)

    def _triangle(self, O, D, idx, reverse):
        v = np.float64(self.k["tri_verts"][idx])
        e1, e2 = v[1] - v[0], v[2] - v[0]
        pv = np.cross(D, e2)
        det = pv @ e1
        tv = O - v[0]
        b1 = (tv * pv).sum(1) / det
        qv = np.cross(tv, e1)
        b2 = (D * qv).sum(1) / det
        t = (qv @ e2) / det
        b0 = 1.0 - b1 - b2
        ok = (b0 >= 0) & (b1 >= 0) & (b2 >= 0) & (det != 0)
        uv = np.float64(self.k["tri_uvs"][idx]) if "tri_uvs" in self.k else np.array([[0, 0], [1, 0], [0, 1.0]])
        u = b0 * uv[0, 0] + b1 * uv[1, 0] + b2 * uv[2, 0]
        w = b0 * uv[0, 1] + b1 * uv[1, 1] + b2 * uv[2, 1]
        ng = np.cross(v[0] - v[2], v[1] - v[2])
        ng = ng / np.linalg.norm(ng)
        nrm = np.broadcast_to(ng, (len(O), 3)).copy()
        if "tri_normals" in self.k:
            vn = np.float64(self.k["tri_normals"][idx])
            ns = b0[:, None] * vn[0] + b1[:, None] * vn[1] + b2[:, None] * vn[2]
            nrm = np.where(((ns * ng).sum(1) < 0)[:, None], -nrm, nrm)
        elif reverse:
            nrm = -nrm
        return t, ok, nrm, u, w

    def _object_ray(self, O, D, xf):
        if not xf:
            return O, D, None
        x = self.k["xforms"][xf - 1]
        w2o = _m34(x["w2o"])
        o = _apply_point(w2o, O)
        self._object_coord = np.abs(O) @ np.abs(w2o[:, :3]).T.max(axis=1) + np.abs(w2o[:, 3]).max()
        return o, _apply_vector(w2o, D), w2o

    def _rect(self, O, D, r, xf, reverse):
        o, d, w2o = self._object_ray(O, D, xf)
        axis = int(r["axis"])
        A, B = [(1, 2), (0, 2), (0, 1)][axis]
        kk, a0, a1, b0, b1 = (float(r[f]) for f in ("k", "a0", "a1", "b0", "b1"))
        t = (kk - o[:, axis]) / d[:, axis]
        a, b = o[:, A] + t * d[:, A], o[:, B] + t * d[:, B]
        ok = (a >= a0) & (a <= a1) & (b >= b0) & (b <= b1)
        nrm = np.zeros((len(O), 3))
        nrm[:, axis] = -1.0 if reverse else 1.0
        if w2o is not None:
            nrm = _unit(_apply_normal(w2o, nrm))
        return t, ok, nrm, (a - a0) / (a1 - a0), (b - b0) / (b1 - b0)

    def _box(self, O, D, bx, xf):
        o, d, w2o = self._object_ray(O, D, xf)
        lo, hi = np.float64(bx["lo"]), np.float64(bx["hi"])
        ta, tb = (lo - o) / d, (hi - o) / d
        tn_axes, tf_axes = np.minimum(ta, tb), np.maximum(ta, tb)
        # an axis the ray runs along constrains nothing when the origin lies between its planes
        along = d == 0
        inside = (o >= lo) & (o <= hi)
        tn_axes = np.where(along, -np.inf, tn_axes)
        tf_axes = np.where(along, np.inf, tf_axes)
        tn, tf = tn_axes.max(1), tf_axes.min(1)
        ok = (tn <= tf) & ~(along & ~inside).any(1)
        entering = tn > 0
        t = np.where(entering, tn, tf)
        ax = np.where(entering, tn_axes.argmax(1), tf_axes.argmin(1))
        p = o + d * t[:, None]
        rows = np.arange(len(O))
        # outward normal of the face: the low face when the point sits on lo[ax]
        on_low = np.abs(p[rows, ax] - lo[ax]) < np.abs(p[rows, ax] - hi[ax])
        nrm = np.zeros((len(O), 3))
        nrm[rows, ax] = np.where(on_low, -1.0, 1.0)
        A = np.where(ax == 0, 1, 0)
        B = np.where(ax == 2, 1, 2)
        u = (p[rows, A] - lo[A]) / (hi[A] - lo[A])
        v = (p[rows, B] - lo[B]) / (hi[B] - lo[B])
        if w2o is not None:
            nrm = _unit(_apply_normal(w2o, nrm))
        return t, ok, nrm, u, v

    def _medium(self, O, D, tmax, m, u):
        """Scatter point of a constant medium: the ray is inside the boundary from t_in (0 if it
        starts inside) to t_out (or its own end); it scatters -ln(u) / density along its path
        from t_in, if that is still inside.  The record carries no surface: normal (1, 0, 0),
        u = v = 0."""
        n = len(O)
        dlen = np.linalg.norm(D, axis=1)
        if int(m["boundary_kind"]) == abi.RT_BOUNDARY_SPHERE:
            sp = self.k["spheres"][int(m["boundary_index"])]
            r = abs(float(sp["r"]))
            oc = O - np.float64(sp["c"])
            t0, t1, ok = self._sphere_t(oc, D, r)
            entry = oc + D * t0[:, None]
            cos_entry = np.where(t0 > 0, np.abs((entry * D).sum(1)) / (r * dlen), 1.0)
            kappa = np.where(t0 > 0, np.maximum(1.0, (oc * oc).sum(1) / (r * r * np.maximum(cos_entry, 1e-12))), 1.0)
            radius = r
        else:
            o, d, w2o = self._object_ray(O, D, int(m["xform"]))
            bx = self.k["boxes"][int(m["boundary_index"])]
            lo, hi = np.float64(bx["lo"]), np.float64(bx["hi"])
            ta, tb = (lo - o) / d, (hi - o) / d
            tn_axes, tf_axes = np.minimum(ta, tb), np.maximum(ta, tb)
            along = d == 0
            inside = (o >= lo) & (o <= hi)
            tn_axes = np.where(along, -np.inf, tn_axes)
            tf_axes = np.where(along, np.inf, tf_axes)
            t0, t1 = tn_axes.max(1), tf_axes.min(1)
            ok = (t0 <= t1) & ~(along & ~inside).any(1)
            ax = tn_axes.argmax(1)
            cos_entry = np.where(t0 > 0, np.abs(d[np.arange(n), ax]) / np.linalg.norm(d, axis=1), 1.0)
            kappa, radius = np.ones(n), 0.0
        t_in = np.maximum(t0, 0.0)
        t_out = np.minimum(t1, tmax)
        t = t_in + (-np.log(u) / float(m["density"])) / dlen
        ok = ok & (t_in < t_out) & (t <= t_out)
        nrm = np.zeros((n, 3))
        nrm[:, 0] = 1.0
        return t, ok, nrm, np.zeros(n), np.zeros(n), kappa, radius, cos_entry

    def _gsphere(self, O, D, g, reverse):
        o2w, w2o = _m34(g["o2w"]), _m34(g["w2o"])
        r = float(g["radius"])
        z_lo, z_hi = sorted((float(g["z_min"]), float(g["z_max"])))
        z_lo, z_hi = np.clip(z_lo, -r, r), np.clip(z_hi, -r, r)
        theta_min, theta_max = np.arccos(np.clip(z_lo / r, -1, 1)), np.arccos(np.clip(z_hi / r, -1, 1))
        phi_max = np.radians(np.clip(float(g["phi_max_deg"]), 0, 360))
        o, d = _apply_point(w2o, O), _apply_vector(w2o, D)
        # The reference moves a transformed ray's origin forward along the ray by the bound of its
        # rounding error (Transform::apply_err, geometry/src/transform.rs:567-587) and reports t
        # from there: dt = |d| . err / |d|^2, err = gamma(3) (|M| |o| + |translation|).
        g3 = 3 * EPS / (1 - 3 * EPS)
        err = g3 * (np.abs(O) @ np.abs(w2o[:, :3]).T + np.abs(w2o[:, 3]))
        shift = (np.abs(d) * err).sum(1) / (d * d).sum(1)
        o = o + d * shift[:, None]
        t0, t1, ok = self._sphere_t(o, d, r)

        def at(t):
            p = o + d * t[:, None]
            p = p * (r / np.linalg.norm(p, axis=1))[:, None]
            phi = np.arctan2(p[:, 1], p[:, 0])
            phi = np.where(phi < 0, phi + 2 * np.pi, phi)
            clipped = ((z_lo > -r) & (p[:, 2] < z_lo)) | ((z_hi < r) & (p[:, 2] > z_hi)) | (phi > phi_max)
            return p, phi, clipped

        first = np.where(t0 > 0, t0, t1)
        p, phi, clipped = at(first)
        use_far = clipped & (first == t0)  # the near point is cut away: the far one may be there
        p2, phi2, clipped2 = at(t1)
        t = np.where(use_far, t1, first)
        ok = ok & ~np.where(use_far, clipped2, clipped)
        p = np.where(use_far[:, None], p2, p)
        phi = np.where(use_far, phi2, phi)
        theta = np.arccos(np.clip(p[:, 2] / r, -1, 1))
        u = phi / phi_max
        v = (theta - theta_min) / (theta_max - theta_min)
        # p(u, v) = r (sin th cos ph, sin th sin ph, cos th), ph = u phi_max, th = theta_min + v (theta_max - theta_min)
        dpdu = phi_max * np.stack([-p[:, 1], p[:, 0], np.zeros(len(O))], axis=1)
        dpdv = (theta_max - theta_min) * np.stack([p[:, 2] * np.cos(phi), p[:, 2] * np.sin(phi),
                                                   -r * np.sin(theta)], axis=1)
        nrm = _unit(np.cross(dpdu, dpdv))
        flips = np.linalg.det(o2w[:, :3]) < 0
        if reverse != flips:
            nrm = -nrm
        nrm = _unit(_apply_normal(w2o, nrm))
        # conditioning in object space, with the direction's length taken out
        return t, ok, nrm, u, v, (o * o).sum(1), r


def clear_rays(scene64, rays, best, rounds=4, seed=1, eps_o=1e-4, eps_d=1e-5):
    """True where small perturbations of the ray keep the same winner (not a grazing ray)."""
    O, D = rays["o"].astype(np.float64), rays["d"].astype(np.float64)
    tmax, time = rays["t_max"].astype(np.float64), rays["time"].astype(np.float64)
    words = ray_words(rays)  # a medium's variate stays that of the ray itself under the perturbations
    rng = np.random.default_rng(seed)
    # two primitives at the same distance (coplanar faces of neighbouring boxes): which of them
    # is reported is a tie-break, not a geometric fact, and no perturbation separates them
    with np.errstate(invalid="ignore"):
        gap = np.where(np.isfinite(best["t_second"]), best["t_second"] - best["t"], np.inf)
    clear = ~((best["prim"] >= 0) & (gap <= 1e-5 * best["t"] + 16 * best["t_resolution"]))
    for _ in range(rounds):
        other = scene64.closest_od(O + rng.normal(size=O.shape) * eps_o, D + rng.normal(size=D.shape) * eps_d,
                                   tmax, time, words)
        clear &= other["prim"] == best["prim"]
        both = (other["prim"] >= 0) & (best["prim"] >= 0)
        # also away from t_max and from a change of face / clipping branch: the normal must not jump
        clear &= ~both | ((other["n"] * best["n"]).sum(1) > 0.999)
    return clear


def compare(hits, best, clear, rel_t=1e-5, abs_n=1e-5, abs_uv=1e-5, kappa_factor=16.0, well_conditioned=32.0):
    """Checks hit records (abi.hit_dtype) against the float64 answer on the clear rays.
    Returns a dict of the largest deviations seen (relative to their bounds where a bound
    depends on the hit)."""
    got_prim = hits["prim"].astype(np.int64)
    got_prim[hits["prim"] == abi.RT_MISS] = -1
    bad = np.nonzero(clear & (got_prim != best["prim"]))[0]
    assert len(bad) == 0, ("hit / primitive differs from float64", bad[:5], got_prim[bad[:5]], best["prim"][bad[:5]])
    h = clear & (best["prim"] >= 0)
    slack = kappa_factor * EPS * np.where(best["radius"][h] > 0, best["kappa"][h], 0.0)  # spheres only
    dt = np.abs(hits["t"][h].astype(np.float64) - best["t"][h])
    dn = np.abs(hits["n"][h].astype(np.float64) - best["n"][h]).max(axis=1)
    du = np.abs(hits["u"][h].astype(np.float64) - best["u"][h])
    du = np.minimum(du, np.abs(1.0 - du))  # phi = 0 and phi = 2 pi are the same meridian
    dv = np.abs(hits["v"][h].astype(np.float64) - best["v"][h])
    duv = np.maximum(du, dv)
    t_bound = rel_t * best["t"][h] + 2 * best["t_resolution"][h] + slack * best["radius"][h]
    out = dict(n_hits=int(h.sum()), t=float((dt / t_bound).max()), n=float((dn / (abs_n + slack)).max()),
               uv=float((duv / (abs_uv + slack)).max()))
    assert out["t"] <= 1.0 and out["n"] <= 1.0 and out["uv"] <= 1.0, out
    # hits whose quadratic is well conditioned, and every planar hit: the plain bounds
    w = best["kappa"][h] < well_conditioned
    out["n_well"] = int(w.sum())
    out["n_plain"] = float(dn[w].max()) if w.any() else 0.0
    out["uv_plain"] = float(duv[w].max()) if w.any() else 0.0
    assert out["n_plain"] <= abs_n and out["uv_plain"] <= abs_uv, out
    return out
