"""raytracer_rs_amd — host-side mirror of raytracer-rs's `raytracer_lib` API over libmi355rt.so.

The names, argument meaning and error behaviour follow the reference's public surface
(/root/reference/raytracer_lib/src):

    create_raytracer(collada_doc, triangles_per_leaf, width, height)        lib.rs:15-20
    create_raytracer_from_file(collada_filename, triangles_per_leaf, w, h)   lib.rs:22-27
    RayTracer.trace_frame_additive() -> int                                  raytracer/mod.rs:80-117
    RayTracer.get_tonemapped_pixels() -> uint32[w*h] (0xAARRGGBB)            raytracer/mod.rs:120-128
    RayTracer.camera.move_rel / add_x_angle / add_y_angle                    scene/camera.rs:63-78
    RayTracer.film.clear / get_pixels / get_estimated_variances              raytracer/film.rs:37-67
    stats.Stats                                                              stats.rs:3-40
    DEFAULT_TRIANGLES_PER_LEAF = 70                                          lib.rs:7

Everything that computes runs in the HIP library through its C ABI (include/mi355rt.h).  There is
no CPU fallback here: if the shared library is missing, importing `lib()` raises, and without a GPU
`create_*` raises RuntimeError carrying the library's error text (the reference's
`Result<_, String>` error becomes the exception message).
"""
import ctypes as C
import os
import sys
import time
import weakref

import numpy as np

DEFAULT_TRIANGLES_PER_LEAF = 70

FLAG_FIX_ROW_INDEX = 1
FLAG_COUNT_STEPS = 2
FLAG_TIME_KERNELS = 4
FLAG_OCTREE_SEMANTICS = 8
FLAG_GROUP_SHARES_DEVICE = 16
FLAG_TRUE_CLOSEST_HIT = 32
FLAG_DEVICE_LBVH = 64

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355RT_LIB") or os.path.join(_HERE, "libmi355rt.so")     # MI355RT_LIB: an A/B build of the library
_lib = None


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rgb", C.c_float * 3), ("tex_id", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("color", C.c_float * 3)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_float))]


class SceneDesc(C.Structure):
    _fields_ = [
        ("tri_verts", C.POINTER(C.c_float)), ("tri_geom", C.POINTER(C.c_uint32)), ("ntri", C.c_uint32),
        ("materials", C.POINTER(Material)), ("nmaterials", C.c_uint32),
        ("lights", C.POINTER(Light)), ("nlights", C.c_uint32),
        ("textures", C.POINTER(Texture)), ("ntextures", C.c_uint32),
        ("camera_orientation", C.c_float * 16), ("camera_fov_deg", C.c_float),
    ]


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("triangles_per_leaf", C.c_uint32),
        ("recursions", C.c_uint32), ("spread", C.c_uint32), ("flags", C.c_uint32),
        ("seed", C.c_uint64), ("device", C.c_int32),
        ("stripe_rows", C.c_uint32), ("stripe_rank", C.c_uint32), ("stripe_world", C.c_uint32),
        ("samples_per_pass", C.c_uint32), ("device_count", C.c_uint32),
    ]


class RayCounts(C.Structure):
    _fields_ = [
        ("primary", C.c_uint64), ("bounce", C.c_uint64), ("shadow", C.c_uint64), ("primary_hits", C.c_uint64),
        ("primary_culled", C.c_uint64),
        ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("trace_launches", C.c_uint64),
        ("inner_execs", C.c_uint64), ("leaf_execs", C.c_uint64),
        ("trace_ms", C.c_double), ("total_ms", C.c_double),
        ("trace_secondary_ms", C.c_double), ("trace_secondary_launches", C.c_uint64), ("shader_clock_mhz", C.c_double),
        ("shadow_skipped", C.c_uint64),
    ]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["total_rays"] = self.primary + self.bounce + self.shadow
        d["traced_rays"] = d["total_rays"] - self.primary_culled - self.shadow_skipped
        return d


# every symbol include/mi355rt.h declares: (name, restype, argtypes)
_H = C.c_void_p
_F = C.POINTER(C.c_float)
_U = C.POINTER(C.c_uint32)
ABI = [
    ("mi355rt_default_config", None, [C.POINTER(Config)]),
    ("mi355rt_create", C.c_int, [C.POINTER(SceneDesc), C.POINTER(Config), C.POINTER(_H)]),
    ("mi355rt_create_from_collada_str", C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(Config), C.POINTER(_H)]),
    ("mi355rt_create_from_collada_file", C.c_int, [C.c_char_p, C.POINTER(Config), C.POINTER(_H)]),
    ("mi355rt_create_from_scene_file", C.c_int, [C.c_char_p, C.POINTER(Config), C.POINTER(_H)]),
    ("mi355rt_destroy", None, [_H]),
    ("mi355rt_last_error", C.c_char_p, [_H]),
    ("mi355rt_trace_frame_additive", C.c_uint32, [_H]),
    ("mi355rt_render", C.c_int, [_H, C.c_uint32, C.POINTER(RayCounts)]),
    ("mi355rt_render_async", C.c_int, [_H, C.c_uint32]),
    ("mi355rt_last_counts", C.c_int, [_H, C.POINTER(RayCounts)]),
    ("mi355rt_get_tonemapped_pixels", C.c_int, [_H, _U, C.c_size_t]),
    ("mi355rt_tonemap_owned_rows_device", C.c_int, [_H, C.c_void_p, C.c_size_t]),
    ("mi355rt_tonemap_owned_rows_device_on_stream", C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("mi355rt_owned_rows", C.c_uint32, [_H]),
    ("mi355rt_owned_row_list", C.c_int, [_H, _U, C.c_size_t]),
    ("mi355rt_film_get", C.c_int, [_H, _F, _F, _U]),
    ("mi355rt_film_clear", C.c_int, [_H]),
    ("mi355rt_film_get_pixels", C.c_int, [_H, _F]),
    ("mi355rt_film_get_estimated_variances", C.c_int, [_H, _F]),
    ("mi355rt_camera_move_rel", C.c_int, [_H, C.c_float, C.c_float, C.c_float]),
    ("mi355rt_camera_add_x_angle", C.c_int, [_H, C.c_float]),
    ("mi355rt_camera_add_y_angle", C.c_int, [_H, C.c_float]),
    ("mi355rt_camera_get", C.c_int, [_H, _F, _F, _F]),
    ("mi355rt_camera_get_ray", C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_float, C.c_float, _F]),
    ("mi355rt_set_seed", C.c_int, [_H, C.c_uint64]),
    ("mi355rt_set_flags", C.c_int, [_H, C.c_uint32]),
    ("mi355rt_set_slices", C.c_int, [_H, C.c_uint32]),
    ("mi355rt_get_slices", C.c_uint32, [_H]),
    ("mi355rt_intersect_rays", C.c_int, [_H, _F, C.c_size_t, _F, _U]),
    ("mi355rt_occluded_rays", C.c_int, [_H, _F, C.c_size_t, C.POINTER(C.c_uint8)]),
    ("mi355rt_get_sample_table", C.c_int, [_H, _F]),
    ("mi355rt_debug_sample", C.c_int, [_H, C.c_uint32, C.c_uint32, _F, _F, C.c_size_t]),
    ("mi355rt_debug_numerics", C.c_int, [_H, _F, _F, C.c_size_t, _F, _F, _F]),
    ("mi355rt_debug_slab", C.c_int, [_H, _F, _F, C.c_size_t, C.POINTER(C.c_uint8), _F]),
    ("mi355rt_tree_nodes", C.c_uint32, [_H]),
    ("mi355rt_debug_speculation", C.c_int, [_H, C.POINTER(C.c_uint64)]),
    ("mi355rt_debug_light_map", C.c_int, [_F, C.c_uint32, _F, C.c_double, C.c_uint32, _F, C.POINTER(C.c_double)]),
    ("mi355rt_debug_wide_bvh", C.c_int, [_F, C.c_uint32, _U]),
    ("mi355rt_accel_stats", C.c_int, [_H, _U]),
    ("mi355rt_octree_stats", C.c_int, [_H, _U]),
    ("mi355rt_bvh_build_info", C.c_int, [_H, _U]),
    ("mi355rt_device_count", C.c_uint32, [_H]),
    ("mi355rt_synchronize", C.c_int, [_H]),
    ("mi355rt_comm_unique_id", C.c_int, [C.POINTER(C.c_uint8)]),
    ("mi355rt_comm_available", C.c_int, [_H]),
    ("mi355rt_comm_init", C.c_int, [_H, C.POINTER(C.c_uint8)]),
    ("mi355rt_comm_gather_frame", C.c_int, [_H, C.c_uint32, _U, C.c_size_t]),
    ("mi355rt_comm_destroy", C.c_int, [_H]),
    ("mi355rt_comm_ranks", C.c_uint32, [_H]),
    ("mi355rt_hbm_allocated_bytes", C.c_uint64, [_H]),
    ("mi355rt_debug_check_guards", C.c_int64, [_H]),
    ("mi355rt_debug_gather_rate", C.c_int, [_H, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    ("mi355rt_width", C.c_uint32, [_H]),
    ("mi355rt_height", C.c_uint32, [_H]),
    ("mi355rt_triangle_count", C.c_uint32, [_H]),
    ("mi355rt_current_row", C.c_uint32, [_H]),
]


def lib():
    """Load libmi355rt.so (built by `make -C raytracer-rs_amd` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libmi355rt.so is not built (%s); run __graft_entry__.build() — there is no CPU fallback" % LIB_PATH)
        # PyTorch ships its own copy of the HIP runtime.  A process that uses both (tests, bench.py: torch buffers and streams are
        # handed to the library) must have torch's copy loaded FIRST, so that libmi355rt.so's libamdhip64 dependency resolves to
        # the runtime already in the process instead of bringing /opt/rocm's as a second one.  Done here, once, for whoever has
        # torch imported already; a host without torch (the C++ CLI, a Rust binary) never meets the question.
        if "torch" in sys.modules:
            try:
                sys.modules["torch"].cuda.is_available()
            except Exception:       # noqa: BLE001 — a CPU-only torch build
                pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(_F)


def _up(a):
    return a.ctypes.data_as(_U)


def default_config(width=1024, height=768, **kw):
    cfg = Config()
    lib().mi355rt_default_config(C.byref(cfg))
    cfg.width, cfg.height = int(width), int(height)
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError("unknown config field %r" % k)
        setattr(cfg, k, v)
    return cfg


class Camera:
    """scene/camera.rs — the `pub camera` field of RayTracer (mod.rs:38)."""

    def __init__(self, rt):
        self._rt = rt

    def move_rel(self, x, y, z):
        self._rt._check(lib().mi355rt_camera_move_rel(self._rt._h, x, y, z))

    def add_x_angle(self, radians):
        self._rt._check(lib().mi355rt_camera_add_x_angle(self._rt._h, radians))

    def add_y_angle(self, radians):
        self._rt._check(lib().mi355rt_camera_add_y_angle(self._rt._h, radians))

    def matrices(self):
        rot = np.zeros(16, np.float32); orient = np.zeros(16, np.float32); mx = np.zeros(2, np.float32)
        self._rt._check(lib().mi355rt_camera_get(self._rt._h, _fp(rot), _fp(orient), _fp(mx)))
        return rot, orient, mx

    def get_ray(self, u, v, xi1, xi2):
        ray = np.zeros(6, np.float32)
        self._rt._check(lib().mi355rt_camera_get_ray(self._rt._h, u, v, xi1, xi2, _fp(ray)))
        return ray


class Film:
    """raytracer/film.rs — the `pub film` field of RayTracer (mod.rs:41)."""

    def __init__(self, rt):
        self._rt = rt

    def clear(self):
        self._rt._check(lib().mi355rt_film_clear(self._rt._h))

    def pixel_datas(self):
        """(pixel_sum[n,3], pixel_sum_squared[n,3], num_samples[n]) — film.rs:3-8"""
        n = self._rt.width * self._rt.height
        s = np.zeros((n, 3), np.float32); q = np.zeros((n, 3), np.float32); c = np.zeros(n, np.uint32)
        self._rt._check(lib().mi355rt_film_get(self._rt._h, _fp(s), _fp(q), _up(c)))
        return s, q, c

    def get_pixels(self):
        out = np.zeros((self._rt.width * self._rt.height, 3), np.float32)
        self._rt._check(lib().mi355rt_film_get_pixels(self._rt._h, _fp(out)))
        return out

    def get_estimated_variances(self):
        out = np.zeros((self._rt.width * self._rt.height, 3), np.float32)
        self._rt._check(lib().mi355rt_film_get_estimated_variances(self._rt._h, _fp(out)))
        return out


class RayTracer:
    """raytracer/mod.rs:32-47.  Construct through create_raytracer* below."""

    def __init__(self, handle, keepalive=None):
        self._h = C.c_void_p(handle)
        self._keep = keepalive
        L = lib()
        self.width = L.mi355rt_width(self._h)
        self.height = L.mi355rt_height(self._h)
        # weak back-references: no reference cycle, so dropping the last reference to a RayTracer destroys the handle (and
        # frees its tens of GB of pass buffers) at once instead of whenever the cycle collector runs
        self.camera = Camera(weakref.proxy(self))
        self.film = Film(weakref.proxy(self))

    def close(self):
        if self._h:
            lib().mi355rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, code):
        if code != 0:
            raise RuntimeError((lib().mi355rt_last_error(self._h) or b"").decode() or "mi355rt error %d" % code)

    # --- the reference's methods
    def trace_frame_additive(self):
        n = lib().mi355rt_trace_frame_additive(self._h)
        if n == 0:
            self._check(-4)
        return n

    def get_tonemapped_pixels(self, out=None):
        """width*height u32 0xAARRGGBB; `out` (uint32[width*height]) is reused when given"""
        if out is None:
            out = np.empty(self.width * self.height, np.uint32)
        self._check(lib().mi355rt_get_tonemapped_pixels(self._h, _up(out), out.size))
        return out

    # --- additions (no reference counterpart)
    def render(self, spp, wait=True):
        """whole frame (owned stripes) x spp; wait=False only queues it (mi355rt_render_async) and returns None: last_counts() waits"""
        if not wait:
            self._check(lib().mi355rt_render_async(self._h, int(spp)))
            return None
        rc = RayCounts()
        self._check(lib().mi355rt_render(self._h, int(spp), C.byref(rc)))
        return rc

    def last_counts(self):
        rc = RayCounts()
        self._check(lib().mi355rt_last_counts(self._h, C.byref(rc)))
        return rc

    def set_seed(self, seed):
        self._check(lib().mi355rt_set_seed(self._h, seed))

    def set_flags(self, flags):
        self._check(lib().mi355rt_set_flags(self._h, flags))

    def set_slices(self, slices):
        """Concurrent frame slices of render() (1..8); results do not depend on it."""
        self._check(lib().mi355rt_set_slices(self._h, slices))

    def get_slices(self):
        return int(lib().mi355rt_get_slices(self._h))

    def synchronize(self):
        self._check(lib().mi355rt_synchronize(self._h))

    @property
    def device_count(self):
        return int(lib().mi355rt_device_count(self._h))

    # --- one process per GPU: RCCL gather inside the library (include/mi355rt.h, mi355rt_comm_*)
    def comm_available(self):
        """(ok, error text): the local pre-check of comm_init; no communication"""
        rc = lib().mi355rt_comm_available(self._h)
        return rc == 0, "" if rc == 0 else (lib().mi355rt_last_error(self._h) or b"").decode()

    def comm_init(self, id128):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(id128))
        self._check(lib().mi355rt_comm_init(self._h, buf))

    def comm_gather_frame(self, root=0, out=None):
        """collective; `out` (uint32[width*height], root only) receives the frame, None only queues the transfers"""
        if out is None:
            self._check(lib().mi355rt_comm_gather_frame(self._h, root, None, 0))
        else:
            self._check(lib().mi355rt_comm_gather_frame(self._h, root, _up(out), out.size))
        return out

    def comm_destroy(self):
        self._check(lib().mi355rt_comm_destroy(self._h))

    def comm_ranks(self):
        """ranks of the live RCCL communicator as RCCL counts them (ncclCommCount); 0 without one"""
        return int(lib().mi355rt_comm_ranks(self._h))

    def debug_gather_rate(self, table_nodes=48000, steps=2000):
        """the device's divergent-gather rate (kernels.hip, gather_rate_kernel): dict(line_accesses_per_s, ms, node_fetches_per_s)"""
        out = (C.c_double * 3)()
        self._check(lib().mi355rt_debug_gather_rate(self._h, table_nodes, steps, out))
        return {"line_accesses_per_s": out[0], "ms": out[1], "node_fetches_per_s": out[2]}

    def debug_speculation(self):
        """(50-row frames launched ahead of their call, how many the next call took over)"""
        out = (C.c_uint64 * 2)()
        self._check(lib().mi355rt_debug_speculation(self._h, out))
        return int(out[0]), int(out[1])

    def debug_check_guards(self):
        """MI355RT_DEBUG_GUARD: overwritten guard bytes behind the pass buffers (0 = clean)"""
        return int(lib().mi355rt_debug_check_guards(self._h))

    def hbm_allocated_bytes(self):
        """device memory the handle holds right now (scene, film, pass buffers, gather slots)"""
        return int(lib().mi355rt_hbm_allocated_bytes(self._h))

    def owned_rows(self):
        n = lib().mi355rt_owned_rows(self._h)
        rows = np.zeros(n, np.uint32)
        if n:
            self._check(lib().mi355rt_owned_row_list(self._h, _up(rows), n))
        return rows

    def tonemap_owned_rows_device(self, device_ptr, n, stream=None):
        """Packed owned rows into device memory.  stream (a hipStream_t as int, e.g.
        torch.cuda.current_stream().cuda_stream): asynchronous, ordered on that stream; None: synchronous."""
        if stream:
            self._check(lib().mi355rt_tonemap_owned_rows_device_on_stream(self._h, C.c_void_p(device_ptr), n, C.c_void_p(stream)))
        else:
            self._check(lib().mi355rt_tonemap_owned_rows_device(self._h, C.c_void_p(device_ptr), n))

    def intersect_rays(self, rays6):
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        n = rays6.shape[0]
        tuv = np.zeros((n, 3), np.float32); prim = np.zeros(n, np.uint32)
        self._check(lib().mi355rt_intersect_rays(self._h, _fp(rays6), n, _fp(tuv), _up(prim)))
        return tuv, prim

    def occluded_rays(self, rays6):
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        n = rays6.shape[0]
        out = np.zeros(n, np.uint8)
        self._check(lib().mi355rt_occluded_rays(self._h, _fp(rays6), n, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def sample_table(self):
        out = np.zeros((65536, 3), np.float32)
        self._check(lib().mi355rt_get_sample_table(self._h, _fp(out)))
        return out

    def debug_sample(self, pixel, sampleno):
        nodes = lib().mi355rt_tree_nodes(self._h)
        color = np.zeros(3, np.float32); node_l = np.zeros((nodes, 3), np.float32)
        self._check(lib().mi355rt_debug_sample(self._h, pixel, sampleno, _fp(color), _fp(node_l), nodes))
        return color, node_l

    def debug_numerics(self, a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        q = np.zeros_like(a); r = np.zeros_like(a); p = np.zeros_like(a)
        self._check(lib().mi355rt_debug_numerics(self._h, _fp(a), _fp(b), a.size, _fp(q), _fp(r), _fp(p)))
        return q, r, p

    def debug_slab(self, inv_rays6, cubes6):
        """intersect_cube_inverse_ray on the device: (hit[n] bool, tmin[n])"""
        r = np.ascontiguousarray(inv_rays6, np.float32).reshape(-1, 6); c = np.ascontiguousarray(cubes6, np.float32).reshape(-1, 6)
        assert r.shape == c.shape
        hit = np.zeros(r.shape[0], np.uint8); tmin = np.zeros(r.shape[0], np.float32)
        self._check(lib().mi355rt_debug_slab(self._h, _fp(r), _fp(c), r.shape[0], hit.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(tmin)))
        return hit.astype(bool), tmin

    def accel_stats(self):
        out = np.zeros(8, np.uint32)
        self._check(lib().mi355rt_accel_stats(self._h, _up(out)))
        return dict(nodes=int(out[0]), leaves=int(out[1]), max_depth=int(out[2]), max_leaf=int(out[3]),
                    node_bytes=int(out[4]), tri_bytes=int(out[5]), bvh_build_ms=int(out[6]) / 1000.0, octree_build_ms=int(out[7]) / 1000.0)

    def bvh_build_info(self):
        out = np.zeros(2, np.uint32)
        self._check(lib().mi355rt_bvh_build_info(self._h, _up(out)))
        return dict(on_device=bool(out[0]), device_ms=int(out[1]) / 1000.0)

    def octree_stats(self):
        out = np.zeros(8, np.uint32)
        self._check(lib().mi355rt_octree_stats(self._h, _up(out)))
        return dict(nodes=int(out[0]), inner=int(out[1]), leaves=int(out[2]), empty=int(out[3]), depth=int(out[4]), tri_refs=int(out[5]))

    @property
    def triangle_count(self):
        return lib().mi355rt_triangle_count(self._h)

    @property
    def current_row(self):
        return lib().mi355rt_current_row(self._h)


def _finish(code, handle, keep=None):
    if code != 0:
        raise RuntimeError((lib().mi355rt_last_error(None) or b"").decode() or "mi355rt error %d" % code)
    return RayTracer(handle.value, keep)


def create_raytracer(collada_doc, triangles_per_leaf, width, height, data_dir=None, **cfg_kw):
    """lib.rs:15-20.  Raises RuntimeError(message) where the reference returns Err(String)."""
    cfg = default_config(width, height, triangles_per_leaf=triangles_per_leaf, **cfg_kw)
    doc = collada_doc.encode() if isinstance(collada_doc, str) else bytes(collada_doc)
    h = C.c_void_p()
    code = lib().mi355rt_create_from_collada_str(doc, len(doc), data_dir.encode() if data_dir else None, C.byref(cfg), C.byref(h))
    return _finish(code, h)


def create_raytracer_from_file(collada_filename, triangles_per_leaf, width, height, **cfg_kw):
    """lib.rs:22-27"""
    cfg = default_config(width, height, triangles_per_leaf=triangles_per_leaf, **cfg_kw)
    h = C.c_void_p()
    code = lib().mi355rt_create_from_collada_file(str(collada_filename).encode(), C.byref(cfg), C.byref(h))
    return _finish(code, h)


def create_raytracer_from_scene_file(scene_filename, triangles_per_leaf, width, height, **cfg_kw):
    cfg = default_config(width, height, triangles_per_leaf=triangles_per_leaf, **cfg_kw)
    h = C.c_void_p()
    code = lib().mi355rt_create_from_scene_file(str(scene_filename).encode(), C.byref(cfg), C.byref(h))
    return _finish(code, h)


def create_raytracer_from_arrays(scene, triangles_per_leaf, width, height, **cfg_kw):
    """build_raytracer (lib.rs:29-44) from parsed arrays — what a Rust shim would marshal.
    `scene` is the dict produced by scene_io.load_scene_file()."""
    cfg = default_config(width, height, triangles_per_leaf=triangles_per_leaf, **cfg_kw)
    verts = np.ascontiguousarray(scene["tri_verts"], np.float32).reshape(-1)
    geom = np.ascontiguousarray(scene["tri_geom"], np.uint32)
    nm = len(scene["mat_kind"])
    mats = (Material * max(nm, 1))()
    for i in range(nm):
        mats[i].kind = int(scene["mat_kind"][i]); mats[i].tex_id = int(scene["mat_tex"][i])
        for c in range(3):
            mats[i].rgb[c] = float(scene["mat_rgb"][i][c])
    nl = len(scene["lights"])
    lights = (Light * max(nl, 1))()
    for i in range(nl):
        for c in range(3):
            lights[i].pos[c] = float(scene["lights"][i][c]); lights[i].color[c] = float(scene["lights"][i][3 + c])
    nt = len(scene["textures"])
    texs = (Texture * max(nt, 1))()
    keep = [verts, geom, mats, lights, texs]
    for i, t in enumerate(scene["textures"]):
        arr = np.ascontiguousarray(t, np.float32)
        keep.append(arr)
        texs[i].height, texs[i].width = arr.shape[0], arr.shape[1]
        texs[i].rgb = _fp(arr)
    sd = SceneDesc()
    sd.tri_verts = _fp(verts); sd.tri_geom = _up(geom); sd.ntri = geom.size
    sd.materials = mats; sd.nmaterials = nm; sd.lights = lights; sd.nlights = nl
    sd.textures = texs; sd.ntextures = nt
    for i in range(16):
        sd.camera_orientation[i] = float(scene["camera_matrix"][i])
    sd.camera_fov_deg = float(scene["camera_fov"])
    h = C.c_void_p()
    code = lib().mi355rt_create(C.byref(sd), C.byref(cfg), C.byref(h))
    return _finish(code, h, keep)


def debug_light_map(tri_verts, light, pad, res):
    """(dist2[6, res, res], nearest): the depth cube map the library builds around a point light (host code; include/mi355rt.h)"""
    v = np.ascontiguousarray(tri_verts, np.float32).reshape(-1, 9)
    l = np.ascontiguousarray(light, np.float32).reshape(3)
    out = np.zeros((6, res, res), np.float32); nearest = C.c_double(0.0)
    code = lib().mi355rt_debug_light_map(_fp(v), v.shape[0], _fp(l), float(pad), int(res), _fp(out), C.byref(nearest))
    if code != 0:
        raise RuntimeError("mi355rt_debug_light_map failed: %d" % code)
    return out, nearest.value


def debug_wide_bvh(tri_verts):
    """facts about the 4-wide tree of 48-byte nodes the MI355RT_WIDE experiment builds walk, checked by a host walk (include/mi355rt.h)"""
    v = np.ascontiguousarray(tri_verts, np.float32).reshape(-1, 9)
    out = (C.c_uint32 * 8)()
    code = lib().mi355rt_debug_wide_bvh(_fp(v), v.shape[0], out)
    if code != 0:
        raise RuntimeError("mi355rt_debug_wide_bvh failed: %d" % code)
    keys = ("wide_nodes", "binary_nodes", "stack_need", "binary_depth", "children", "tris_once_wide", "bad_boxes", "tris_once_binary")
    return dict(zip(keys, (int(x) for x in out)))


def comm_unique_id():
    """128-byte RCCL id (rank 0 creates it, the host application hands it to the other ranks)"""
    buf = (C.c_uint8 * 128)()
    if lib().mi355rt_comm_unique_id(buf) != 0:
        raise RuntimeError((lib().mi355rt_last_error(None) or b"").decode() or "mi355rt_comm_unique_id failed")
    return bytes(buf)


class Stats:
    """stats.rs:3-40 — fps and primary rays/s per call and running mean."""

    def __init__(self):
        self.last_iteration = time.perf_counter()
        self.fps_sum = 0.0
        self.primrays_per_sec_sum = 0.0
        self.num_measurements = 0

    def stats(self, num_primary_rays):
        now = time.perf_counter()
        dt = now - self.last_iteration
        self.last_iteration = now
        fps = 1.0 / dt
        self.fps_sum += fps
        prs = num_primary_rays / dt
        self.primrays_per_sec_sum += prs
        self.num_measurements += 1
        return "fps: %s  primary rays/s: %d" % (fps, int(prs))

    def mean_stats(self):
        return "mean fps: %s  mean primary rays/s: %s" % (
            self.fps_sum / self.num_measurements, self.primrays_per_sec_sum / self.num_measurements)


class stats:  # namespace alias: raytracer_lib::stats::Stats
    Stats = Stats
