"""Loads librt355.so (built in-tree by raytracer_amd/build.py) and wraps its C ABI.

There is no fallback: if the shared library is missing or a call fails, an
exception is raised.  Device memory handed to the ``*_dev`` entry points comes
from the caller (raw pointers, e.g. ``torch.Tensor.data_ptr()``).
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT355_LIB") or os.path.join(_HERE, "librt355.so")
_LIB = None


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("librt355 error %d: %s" % (code, message))
        self.code = code


def load():
    global _LIB
    if _LIB is None:
        if os.environ.get("RT355_NO_TORCH") != "1":
            # PyTorch bundles its own HIP runtime under the same SONAME as the
            # system one.  Loading it first makes the dynamic loader hand that
            # single copy to librt355 too; two runtimes in one process cannot
            # both open the GPU.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -m raytracer_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        _LIB = abi.declare(C.CDLL(LIB_PATH), prefix="rt_")
    return _LIB


def check(rc):
    if rc < 0:
        raise RtError(rc, load().rt_last_error().decode(errors="replace"))
    return rc


def make_filter(kind=abi.RT_FILTER_NONE, radius=(0.5, 0.5), a=0.0, b=0.0):
    """rt_filter: Box / Triangle / Gaussian(alpha) / Mitchell(B, C) / LanczosSinc(tau) ::new."""
    f = abi.rt_filter()
    f.kind = kind
    f.radius[:] = radius
    f.a, f.b = a, b
    return f


def render_opts(width, height, spp, max_depth=50, seed=0, shard=(0, 1), max_paths=0, flags=0, filter=None,
                sampler=0, sampler_arg=0):
    return abi.rt_render_opts(width, height, spp, max_depth, seed, shard[0], shard[1], max_paths,
                              flags, filter if filter is not None else abi.rt_filter(), sampler, sampler_arg)


def filter_table(filt):
    """Film::new's 16 x 16 table of filter values (film.rs:99-113)."""
    out = np.zeros(256, dtype=np.float32)
    check(load().rt_filter_table(C.byref(filt), out.ctypes.data))
    return out.reshape(16, 16)


def sobol_tables(width, height):
    """(MATRICES_32[:5*52], MATRICES[m-1], MATRICES_INV[m-1]) of data/src/sobol.rs for a frame."""
    m32 = np.zeros(260, dtype=np.uint32)
    vdc = np.zeros(52, dtype=np.uint64)
    inv = np.zeros(52, dtype=np.uint64)
    check(load().rt_sobol_tables(C.c_uint32(width), C.c_uint32(height), C.c_void_p(m32.ctypes.data),
                                 C.c_void_p(vdc.ctypes.data), C.c_void_p(inv.ctypes.data)))
    return m32.reshape(5, 52), vdc, inv


def film_resolve_dev(rgba_ptr, width, height, device=0, stream=None):
    check(load().rt_film_resolve_dev(rgba_ptr, width, height, device, stream))


class Scene:
    """An immutable scene resident in HBM: the aggregate behind
    ``Primitive::{world_bound, intersect, does_intersect}``
    (shapes/src/primitive.rs:8-27, shapes/src/bvh.rs:727-863)."""

    def __init__(self, frozen, max_prims_in_node=4, split_method=abi.RT_SPLIT_SAH, device=0,
                 build_on_device=False):
        self._lib = load()
        self._frozen = frozen  # keeps the description's arrays alive during create
        self.device = device
        h = C.c_void_p()
        opts = abi.rt_build_opts(max_prims_in_node, split_method,
                                 abi.RT_BUILD_ON_DEVICE if build_on_device else 0, device)
        check(self._lib.rt_scene_create(C.byref(frozen.desc), C.byref(opts), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_scene_destroy(self._h)
            self._h = None

    __del__ = close

    # -- host-side queries (no GPU needed) --
    def world_bound(self):
        out = (C.c_float * 6)()
        check(self._lib.rt_scene_world_bound(self._h, C.byref(out)))
        return np.array(out[:], dtype=np.float32)

    def info(self):
        i = abi.rt_scene_info()
        check(self._lib.rt_scene_get_info(self._h, C.byref(i)))
        return i

    def nodes(self):
        n = check(self._lib.rt_scene_get_nodes(self._h, None, 0))
        out = np.zeros(n, dtype=abi.linear_node_dtype)
        check(self._lib.rt_scene_get_nodes(self._h, out.ctypes.data, n))
        return out

    def prim_order(self):
        n = check(self._lib.rt_scene_get_prim_order(self._h, None, 0))
        out = np.zeros(n, dtype=np.uint32)
        check(self._lib.rt_scene_get_prim_order(self._h, out.ctypes.data, n))
        return out

    # -- GPU work --
    def upload(self):
        check(self._lib.rt_scene_upload(self._h, self.device))

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=abi.ray_dtype)
        hits = np.zeros(len(rays), dtype=abi.hit_dtype)
        check(self._lib.rt_intersect(self._h, rays.ctypes.data, len(rays), hits.ctypes.data, self.device))
        return hits

    def does_intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=abi.ray_dtype)
        out = np.zeros(len(rays), dtype=np.uint8)
        check(self._lib.rt_occluded(self._h, rays.ctypes.data, len(rays), out.ctypes.data, self.device))
        return out

    def intersect_dev(self, rays_ptr, n, hits_ptr, stream=None):
        check(self._lib.rt_intersect_dev(self._h, rays_ptr, n, hits_ptr, self.device, stream))

    def intersect_dev_noalloc(self, rays_ptr, n, hits_ptr, stream=None):
        """rt_intersect_dev without any device allocation (the scene must be uploaded)."""
        check(self._lib.rt_intersect_dev_noalloc(self._h, rays_ptr, n, hits_ptr, self.device, stream))

    def occluded_dev(self, rays_ptr, n, out_ptr, stream=None):
        check(self._lib.rt_occluded_dev(self._h, rays_ptr, n, out_ptr, self.device, stream))

    def intersect_count(self, rays_ptr, n, hits_ptr):
        out = (C.c_uint64 * 9)()
        check(self._lib.rt_intersect_count(self._h, rays_ptr, n, hits_ptr, self.device, C.byref(out)))
        return np.array(out[:], dtype=np.uint64)

    def render(self, cam, opts):
        img = np.zeros((opts.height, opts.width, 4), dtype=np.float32)
        st = abi.rt_stats()
        check(self._lib.rt_render(self._h, C.byref(cam), C.byref(opts), img.ctypes.data, C.byref(st),
                                  self.device))
        return img, st

    def render_dev(self, cam, opts, rgba_ptr, stream=None, want_stats=True):
        """want_stats=False: the frame is enqueued on `stream` and not waited for (rt_render_dev
        with stats == NULL); returns None."""
        if not want_stats:
            check(self._lib.rt_render_dev(self._h, C.byref(cam), C.byref(opts), rgba_ptr, None, self.device, stream))
            return None
        st = abi.rt_stats()
        check(self._lib.rt_render_dev(self._h, C.byref(cam), C.byref(opts), rgba_ptr, C.byref(st),
                                      self.device, stream))
        return st


def generate_rays(cam, opts, first, n, device=0):
    rays = np.zeros(n, dtype=abi.ray_dtype)
    check(load().rt_generate_rays(C.byref(cam), C.byref(opts), first, n, rays.ctypes.data, device))
    return rays
