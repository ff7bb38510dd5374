"""Halton camera samples (colour/src/samplers/halton.rs) of the oracle against exact
rational arithmetic: the sample's first two dimensions are radical inverses in bases 2
and 3 of ONE global index per (pixel, sample), that index's Halton point lies in the
pixel it was asked for, scrambled dimensions use a permutation of the digits, and
successive samples of a pixel are `stride` apart."""
from fractions import Fraction

import numpy as np
import pytest

from raytracer_amd import abi, scenes
from tests import orc


def radical_inverse(base, n):
    f, scale = Fraction(0), Fraction(1, base)
    while n:
        f += (n % base) * scale
        n //= base
        scale /= base
    return f


def digits_reversed(base, n, count):
    """Integer whose base-`base` digits are the first `count` digits of n, reversed."""
    out = 0
    for _ in range(count):
        out = out * base + n % base
        n //= base
    return out


@pytest.mark.parametrize("w,h", [(64, 48), (200, 150), (400, 300)])
def test_halton_points_land_in_their_pixel(w, h):
    L = orc.lib()
    import ctypes as C
    L.orc_camera_sample_test.restype = None
    L.orc_camera_sample_test.argtypes = [C.POINTER(abi.rt_render_opts), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    spp = 5
    opts = orc.render_opts(w, h, spp, sampler=abi.RT_SAMPLER_HALTON)
    sx, ex = 1, 0
    while sx < min(w, 128):
        sx, ex = sx * 2, ex + 1
    sy, ey = 1, 0
    while sy < min(h, 128):
        sy, ey = sy * 3, ey + 1
    stride = sx * sy
    # cell of the Halton point i in the sx x sy grid: its first digits reversed
    cell_to_index = {(digits_reversed(2, i, ex), digits_reversed(3, i, ey)): i for i in range(stride)}
    assert len(cell_to_index) == stride  # one index per cell in every block of `stride` points
    rng = np.random.default_rng(1)
    for _ in range(40):
        px, py = int(rng.integers(0, w)), int(rng.integers(0, h))
        base_index = cell_to_index[((px % 128) % sx, (py % 128) % sy)]
        for k in range(spp):
            u = np.zeros(5, np.float32)
            L.orc_camera_sample_test(C.byref(opts), px, py, k, u.ctypes.data)
            assert ((u >= 0) & (u < 1)).all()
            i = base_index + k * stride
            fx, fy = radical_inverse(2, i) * sx, radical_inverse(3, i) * sy
            assert int(fx) == (px % 128) % sx and int(fy) == (py % 128) % sy  # the point lies in the pixel
            assert abs(float(fx - int(fx)) - u[0]) < 2e-6 and abs(float(fy - int(fy)) - u[1]) < 2e-6


def test_halton_low_discrepancy_and_scrambling():
    L = orc.lib()
    import ctypes as C
    L.orc_camera_sample_test.restype = None
    L.orc_camera_sample_test.argtypes = [C.POINTER(abi.rt_render_opts), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    w = h = 1  # one pixel: stride 1, index = sample number
    n = 5 ** 3 * 7
    opts = orc.render_opts(w, h, n, sampler=abi.RT_SAMPLER_HALTON)
    u = np.zeros((n, 5), np.float32)
    for s in range(n):
        L.orc_camera_sample_test(C.byref(opts), 0, 0, s, u[s].ctypes.data)
    # dimension 2 is base 5 with permuted digits: each of the 125 cells of width 1/125 gets 7 points
    cells = np.floor(u[:, 2] * 125).astype(int)
    assert (np.bincount(cells, minlength=125) == 7).all()
    # unscrambled dimensions are exact radical inverses
    for s in (1, 2, 3, 10, 100, 873):
        assert abs(float(radical_inverse(2, s)) - u[s, 0]) < 1e-6
        assert abs(float(radical_inverse(3, s)) - u[s, 1]) < 1e-6
    # first digit of dimension 2 is a permutation of 0..4 applied to s % 5
    perm = [int(u[d, 2] * 5) for d in range(5)]
    assert sorted(perm) == [0, 1, 2, 3, 4]
    assert all(int(u[s, 2] * 5) == perm[s % 5] for s in range(n))


def test_render_with_halton(scene_cache):
    _, cam, fr = scene_cache("pbrt")
    o = orc.OracleScene(fr)
    a, sa = o.render(cam(48, 32), orc.render_opts(48, 32, 4, 8, seed=2, sampler=abi.RT_SAMPLER_HALTON))
    b, sb = o.render(cam(48, 32), orc.render_opts(48, 32, 4, 8, seed=2))
    assert np.isfinite(a).all() and a.tobytes() != b.tobytes()
    assert abs(a[..., :3].mean() - b[..., :3].mean()) < 0.05


def _camera_samples(opts, px, py, n):
    import ctypes as C
    L = orc.lib()
    L.orc_camera_sample_test.restype = None
    L.orc_camera_sample_test.argtypes = [C.POINTER(abi.rt_render_opts), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    u = np.zeros((n, 5), np.float32)
    for s in range(n):
        L.orc_camera_sample_test(C.byref(opts), px, py, s, u[s].ctypes.data)
    return u


# --- Sobol' (colour/src/samplers/sobol.rs) ----------------------------------------------------
# Generator matrices of dimensions 0-4, MATRICES_32 of data/src/sobol.rs (first 5 x 52
# entries), regenerated here from the Joe-Kuo parameters with Python integers.
JOE_KUO = [(0, 0, []), (1, 0, [1]), (2, 1, [1, 3]), (3, 1, [1, 3, 1]), (3, 2, [1, 1, 1])]


def sobol_columns(s, a, m, bits=52):
    if s == 0:
        return [1 << (bits - 1 - j) for j in range(bits)]
    m = list(m)
    for k in range(s, bits):
        v = m[k - s] ^ (m[k - s] << s)
        for i in range(1, s):
            if (a >> (s - 1 - i)) & 1:
                v ^= m[k - i] << i
        m.append(v)
    return [m[j] << (bits - 1 - j) for j in range(bits)]


def sobol_u32(index, dim):
    cols = sobol_columns(*JOE_KUO[dim])
    v, j = 0, 0
    while index:
        if index & 1:
            v ^= cols[j] >> 20
        index >>= 1
        j += 1
    return v


def test_sobol_tables_known_entries():
    """A few entries of the three tables as data/src/sobol.rs prints them (MATRICES_32 rows
    of dimensions 2-4, MATRICES[10], MATRICES_INV[10]: the 2048-pixel resolution of a
    1920x1080 frame)."""
    import ctypes as C
    L = orc.lib()
    m32 = np.zeros(260, np.uint32)
    vdc = np.zeros(52, np.uint64)
    inv = np.zeros(52, np.uint64)
    L.orc_sobol_tables(C.c_uint32(1920), C.c_uint32(1080), C.c_void_p(m32.ctypes.data), C.c_void_p(vdc.ctypes.data),
                       C.c_void_p(inv.ctypes.data))
    m32 = m32.reshape(5, 52)
    assert [hex(v) for v in m32[2, :8]] == ["0x80000000", "0xc0000000", "0x60000000", "0x90000000", "0xe8000000",
                                            "0x5c000000", "0x8e000000", "0xc5000000"]
    assert [hex(v) for v in m32[3, :5]] == ["0x80000000", "0xc0000000", "0x20000000", "0x50000000", "0xf8000000"]
    assert [hex(v) for v in m32[4, :4]] == ["0x80000000", "0x40000000", "0x20000000", "0xb0000000"]
    assert hex(m32[1, 51]) == "0xf000f000" and m32[0, 32] == 0
    for d in range(5):
        assert [int(v) for v in m32[d]] == [c >> 20 for c in sobol_columns(*JOE_KUO[d])]
    # T maps an index to its pixel: T(inv[c]) = 1 << c, and vdc is T of the higher index bits
    m = 11

    def pixel_bits(index):
        return ((sobol_u32(index, 0) >> (32 - m)) << m) | (sobol_u32(index, 1) >> (32 - m))
    for c in range(2 * m):
        assert pixel_bits(int(inv[c])) == 1 << c and int(inv[c]) < 1 << (2 * m)
    for c in range(52 - 2 * m):
        assert int(vdc[c]) == pixel_bits(1 << (2 * m + c))
    assert not inv[2 * m:].any() and not vdc[52 - 2 * m:].any()


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (64, 48), (400, 300), (1920, 1080)])
def test_sobol_points_land_in_their_pixel(w, h):
    spp = 6
    opts = orc.render_opts(w, h, spp, sampler=abi.RT_SAMPLER_SOBOL)
    m = 0
    while (1 << m) < max(w, h):
        m += 1
    rng = np.random.default_rng(3)
    seen = set()
    for _ in range(25):
        px, py = int(rng.integers(0, w)), int(rng.integers(0, h))
        u = _camera_samples(opts, px, py, spp)
        assert ((u >= 0) & (u < 1)).all()
        for s in range(spp):
            # the index of sample s of this pixel: the only one in [s << 2m, (s + 1) << 2m) whose
            # first two dimensions fall into the pixel
            cands = [i for i in range(s << (2 * m), (s + 1) << (2 * m))
                     if (sobol_u32(i, 0) >> (32 - m) if m else 0) == px and (sobol_u32(i, 1) >> (32 - m) if m else 0) == py] \
                if m <= 3 else None
            if cands is not None:
                assert len(cands) == 1
                index = cands[0]
                want = [np.float32(sobol_u32(index, d)) * np.float32(2.0 ** -32) for d in range(5)]
                assert u[s, 2] == want[2] and u[s, 3] == want[3] and u[s, 4] == want[4]
                assert u[s, 0] == np.float32(want[0] * np.float32(1 << m) - np.float32(px))
                seen.add(index)
            # dimensions 0 and 1 scaled by the resolution lie in the pixel for every frame size
            assert 0.0 <= u[s, 0] < 1.0 and 0.0 <= u[s, 1] < 1.0
    # the time dimension of successive samples of a pixel is stratified: two samples, two halves
    u = _camera_samples(orc.render_opts(w, h, 64, sampler=abi.RT_SAMPLER_SOBOL), w // 2, h // 2, 64)
    assert len(np.unique(u[:, 2])) == 64


def test_sobol_is_the_sobol_sequence_on_one_pixel():
    n = 256
    u = _camera_samples(orc.render_opts(1, 1, n, sampler=abi.RT_SAMPLER_SOBOL), 0, 0, n)
    for s in (0, 1, 2, 3, 100, 255):
        for d in range(5):
            assert u[s, d] == min(np.float32(sobol_u32(s, d)) * np.float32(2.0 ** -32), np.float32(1 - 2.0 ** -23))
    # (0,2)-net in base 2 over dimensions 0 and 1: every 16 x 16 cell holds exactly one of 256 points
    cells = np.floor(u[:, 0] * 16).astype(int) * 16 + np.floor(u[:, 1] * 16).astype(int)
    assert (np.bincount(cells, minlength=256) == 1).all()


# --- (0,2)-sequence and stratified pixel samplers --------------------------------------------
@pytest.mark.parametrize("n", [1, 4, 16, 64])
def test_zero_two_points_are_a_net(n):
    opts = orc.render_opts(20, 10, n, seed=5, sampler=abi.RT_SAMPLER_ZERO_TWO)
    for px, py in [(0, 0), (7, 3), (19, 9)]:
        u = _camera_samples(opts, px, py, n)
        assert ((u >= 0) & (u < 1)).all()
        k = n.bit_length() - 1
        for pair in ((0, 1), (3, 4)):  # film and lens points: one point in every elementary interval
            for a in range(k + 1):
                ca, cb = 1 << a, 1 << (k - a)
                cells = np.floor(u[:, pair[0]] * ca).astype(int) * cb + np.floor(u[:, pair[1]] * cb).astype(int)
                assert (np.bincount(cells, minlength=n) == 1).all()
        assert (np.bincount(np.floor(u[:, 2] * n).astype(int), minlength=n) == 1).all()  # time: one per 1/n
    a = _camera_samples(opts, 1, 1, n)
    b = _camera_samples(opts, 2, 1, n)
    assert n == 1 or a.tobytes() != b.tobytes()  # pixels are scrambled independently
    other_seed = _camera_samples(orc.render_opts(20, 10, n, seed=6, sampler=abi.RT_SAMPLER_ZERO_TWO), 1, 1, n)
    assert a.tobytes() != other_seed.tobytes()


@pytest.mark.parametrize("nx,ny,jitter", [(4, 4, True), (8, 2, True), (3, 5, True), (4, 2, False), (1, 1, True)])
def test_stratified_points_fill_their_strata(nx, ny, jitter):
    n = nx * ny
    arg = nx | (0 if jitter else abi.RT_STRATIFIED_NO_JITTER)
    opts = orc.render_opts(20, 10, n, seed=1, sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=arg)
    for px, py in [(0, 0), (11, 4)]:
        u = _camera_samples(opts, px, py, n)
        assert ((u >= 0) & (u < 1)).all()
        for pair in ((0, 1), (3, 4)):
            cells = np.floor(u[:, pair[0]] * nx).astype(int) * ny + np.floor(u[:, pair[1]] * ny).astype(int)
            assert (np.bincount(cells, minlength=n) == 1).all()
        assert (np.bincount(np.floor(u[:, 2] * n).astype(int), minlength=n) == 1).all()
        if not jitter:  # stratum centres
            assert np.allclose(np.sort(np.unique(u[:, 0])), (np.arange(nx) + 0.5) / nx)
            assert np.allclose(np.sort(np.unique(u[:, 1])), (np.arange(ny) + 0.5) / ny)
            assert np.allclose(np.sort(u[:, 2]), (np.arange(n) + 0.5) / n)
    if jitter and n > 1:
        assert _camera_samples(opts, 3, 3, n).tobytes() != _camera_samples(opts, 4, 3, n).tobytes()


def test_sampler_arguments_are_checked(scene_cache):
    _, cam, fr = scene_cache("pbrt")
    o = orc.OracleScene(fr)
    for kw in (dict(spp=6, sampler=abi.RT_SAMPLER_ZERO_TWO),
               dict(spp=6, sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=4),
               dict(spp=6, sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=0),
               dict(spp=4, sampler=9)):
        spp = kw.pop("spp")
        with pytest.raises(RuntimeError):
            o.render(cam(16, 8), orc.render_opts(16, 8, spp, 4, **kw))


@pytest.mark.parametrize("kw", [dict(sampler=abi.RT_SAMPLER_SOBOL), dict(sampler=abi.RT_SAMPLER_ZERO_TWO),
                                dict(sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=2)])
def test_render_with_each_sampler(scene_cache, kw):
    _, cam, fr = scene_cache("pbrt")
    o = orc.OracleScene(fr)
    a, _ = o.render(cam(48, 32), orc.render_opts(48, 32, 4, 8, seed=2, **kw))
    b, _ = o.render(cam(48, 32), orc.render_opts(48, 32, 4, 8, seed=2))
    assert np.isfinite(a).all() and a.tobytes() != b.tobytes()
    assert abs(a[..., :3].mean() - b[..., :3].mean()) < 0.05
    # threads and sharding do not change the image
    c, _ = o.render(cam(48, 32), orc.render_opts(48, 32, 4, 8, seed=2, **kw), threads=1)
    assert a.tobytes() == c.tobytes()


def test_library_sobol_tables_match_the_oracle():
    """Host code of the product (no GPU needed): the tables it uploads for a frame."""
    import ctypes as C
    from raytracer_amd import lib
    L = orc.lib()
    for w, h in [(1, 1), (2, 1), (40, 24), (1920, 1080), (3840, 2160), (70000, 3)]:
        m32 = np.zeros(260, np.uint32)
        vdc = np.zeros(52, np.uint64)
        inv = np.zeros(52, np.uint64)
        L.orc_sobol_tables(C.c_uint32(w), C.c_uint32(h), C.c_void_p(m32.ctypes.data), C.c_void_p(vdc.ctypes.data),
                           C.c_void_p(inv.ctypes.data))
        a, b, c = lib.sobol_tables(w, h)
        assert a.tobytes() == m32.tobytes() and b.tobytes() == vdc.tobytes() and c.tobytes() == inv.tobytes()
