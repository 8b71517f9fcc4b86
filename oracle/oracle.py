"""ctypes wrapper of liboracle.so — TEST INFRASTRUCTURE (see the header of oracle.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
FLAG_FIX_ROW_INDEX = 1
FLAG_BRUTE_FORCE = 2
_lib = None
_F = C.POINTER(C.c_float)
_U = C.POINTER(C.c_uint32)
_U64 = C.POINTER(C.c_uint64)


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [_F, _U, C.c_uint32, _U, _F, _U, C.c_uint32, _F, C.c_uint32, _U, _F, C.c_uint32, _F, C.c_float,
                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_set_flags.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_trace_frame_additive.restype = C.c_uint32
        L.oracle_trace_frame_additive.argtypes = [C.c_void_p]
        L.oracle_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _U64]
        L.oracle_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _U64]
        L.oracle_sample_debug.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _F, _F, C.POINTER(C.c_uint8)]
        L.oracle_primary_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _F]
        L.oracle_get_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, _F]
        L.oracle_intersect_mt.argtypes = [C.c_void_p, _F, C.c_uint32, C.c_int, _F, _U, C.c_uint32]
        L.oracle_film_clear.argtypes = [C.c_void_p]
        L.oracle_film_get.argtypes = [C.c_void_p, _F, _F, _U]
        L.oracle_get_pixels.argtypes = [C.c_void_p, _F]
        L.oracle_get_estimated_variances.argtypes = [C.c_void_p, _F]
        L.oracle_get_tonemapped.argtypes = [C.c_void_p, _U]
        L.oracle_tonemap_pack.restype = C.c_uint32
        L.oracle_tonemap_pack.argtypes = [C.c_float, C.c_float, C.c_float]
        L.oracle_camera_move_rel.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.oracle_camera_add_x_angle.argtypes = [C.c_void_p, C.c_float]
        L.oracle_camera_add_y_angle.argtypes = [C.c_void_p, C.c_float]
        L.oracle_camera_get.argtypes = [C.c_void_p, _F, _F, _F]
        L.oracle_sample_table.argtypes = [C.c_void_p, _F]
        L.oracle_current_row.restype = C.c_uint32
        L.oracle_current_row.argtypes = [C.c_void_p]
        L.oracle_octree_stats.argtypes = [C.c_void_p, _U]
        L.oracle_counters.argtypes = [C.c_void_p, _U64]
        L.oracle_tree_nodes.restype = C.c_uint32
        L.oracle_tree_nodes.argtypes = [C.c_void_p]
        L.oracle_slab.restype = C.c_int
        L.oracle_slab.argtypes = [_F, _F, _F]
        L.oracle_mt.restype = C.c_int
        L.oracle_mt.argtypes = [_F, _F, _F]
        L.oracle_mat_mul.argtypes = [_F, _F, _F]
        L.oracle_mat_vec.argtypes = [_F, _F, _F]
        L.oracle_collada_matrix_to_vecmath.argtypes = [_F, _F]
        L.oracle_pow32.restype = C.c_float
        L.oracle_pow32.argtypes = [C.c_float]
        L.oracle_pcg4d.argtypes = [_U, _U]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(_F)


def _up(a):
    return a.ctypes.data_as(_U)


class Oracle:
    """One reference RayTracer (raytracer/mod.rs:32-47) on the CPU."""

    def __init__(self, scene, width, height, tris_per_leaf=70, recursions=2, spread=1, seed=1, flags=0):
        L = lib()
        verts = np.ascontiguousarray(scene["tri_verts"], np.float32).reshape(-1)
        geom = np.ascontiguousarray(scene["tri_geom"], np.uint32)
        kind = np.ascontiguousarray(scene["mat_kind"], np.uint32)
        rgb = np.ascontiguousarray(scene["mat_rgb"], np.float32).reshape(-1)
        tex = np.ascontiguousarray(scene["mat_tex"], np.uint32)
        lights = np.ascontiguousarray(scene["lights"], np.float32).reshape(-1)
        dims = np.zeros(max(2 * len(scene["textures"]), 2), np.uint32)
        chunks = []
        for i, t in enumerate(scene["textures"]):
            dims[2 * i], dims[2 * i + 1] = t.shape[1], t.shape[0]
            chunks.append(np.ascontiguousarray(t, np.float32).reshape(-1))
        texdata = np.concatenate(chunks) if chunks else np.zeros(3, np.float32)
        cam = np.ascontiguousarray(scene["camera_matrix"], np.float32)
        self.width, self.height = width, height
        self.ntri = geom.size
        self._h = L.oracle_create(_fp(verts), _up(geom), geom.size, _up(kind), _fp(rgb), _up(tex), kind.size,
                                  _fp(lights), len(scene["lights"]), _up(dims), _fp(texdata), len(scene["textures"]),
                                  _fp(cam), float(scene["camera_fov"]), width, height, tris_per_leaf, recursions, spread, seed, flags)

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_seed(self, s): lib().oracle_set_seed(self._h, s)
    def set_flags(self, f): lib().oracle_set_flags(self._h, f)
    def trace_frame_additive(self): return lib().oracle_trace_frame_additive(self._h)

    def render(self, spp, nthreads=1, rows=None):
        c = np.zeros(8, np.uint64)
        if rows is None:
            lib().oracle_render(self._h, spp, nthreads, c.ctypes.data_as(_U64))
        else:
            lib().oracle_render_rows(self._h, rows[0], rows[1], spp, nthreads, c.ctypes.data_as(_U64))
        return dict(primary=int(c[0]), bounce=int(c[1]), shadow=int(c[2]), nodes=int(c[3]), cubes=int(c[4]), tris=int(c[5]), primary_hits=int(c[6]))

    def sample_debug(self, pixel, sampleno):
        n = lib().oracle_tree_nodes(self._h)
        color = np.zeros(3, np.float32); node_l = np.zeros((n, 3), np.float32); hit = np.zeros(n, np.uint8)
        lib().oracle_sample_debug(self._h, pixel, sampleno, _fp(color), _fp(node_l), hit.ctypes.data_as(C.POINTER(C.c_uint8)))
        return color, node_l, hit

    def primary_ray(self, pixel, sampleno):
        r = np.zeros(6, np.float32); lib().oracle_primary_ray(self._h, pixel, sampleno, _fp(r)); return r

    def get_ray(self, u, v, xi1, xi2):
        r = np.zeros(6, np.float32); lib().oracle_get_ray(self._h, u, v, xi1, xi2, _fp(r)); return r

    def intersect(self, rays6, brute=False, nthreads=8):
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        n = rays6.shape[0]
        tuv = np.zeros((n, 3), np.float32); prim = np.zeros(n, np.uint32)
        lib().oracle_intersect_mt(self._h, _fp(rays6), n, 1 if brute else 0, _fp(tuv), _up(prim), nthreads)
        return tuv, prim

    def film_clear(self): lib().oracle_film_clear(self._h)

    def film(self):
        n = self.width * self.height
        s = np.zeros((n, 3), np.float32); q = np.zeros((n, 3), np.float32); c = np.zeros(n, np.uint32)
        lib().oracle_film_get(self._h, _fp(s), _fp(q), _up(c))
        return s, q, c

    def get_pixels(self):
        out = np.zeros((self.width * self.height, 3), np.float32); lib().oracle_get_pixels(self._h, _fp(out)); return out

    def get_estimated_variances(self):
        out = np.zeros((self.width * self.height, 3), np.float32); lib().oracle_get_estimated_variances(self._h, _fp(out)); return out

    def get_tonemapped_pixels(self):
        out = np.zeros(self.width * self.height, np.uint32); lib().oracle_get_tonemapped(self._h, _up(out)); return out

    def camera_move_rel(self, x, y, z): lib().oracle_camera_move_rel(self._h, x, y, z)
    def camera_add_x_angle(self, r): lib().oracle_camera_add_x_angle(self._h, r)
    def camera_add_y_angle(self, r): lib().oracle_camera_add_y_angle(self._h, r)

    def camera_matrices(self):
        rot = np.zeros(16, np.float32); orient = np.zeros(16, np.float32); mx = np.zeros(2, np.float32)
        lib().oracle_camera_get(self._h, _fp(rot), _fp(orient), _fp(mx))
        return rot, orient, mx

    def sample_table(self):
        out = np.zeros((65536, 3), np.float32); lib().oracle_sample_table(self._h, _fp(out)); return out

    def octree_stats(self):
        s = np.zeros(8, np.uint32); lib().oracle_octree_stats(self._h, _up(s))
        return dict(nodes=int(s[0]), inner=int(s[1]), leaves=int(s[2]), empty=int(s[3]), depth=int(s[4]), tri_refs=int(s[5]), max_leaf=int(s[6]))

    @property
    def current_row(self): return lib().oracle_current_row(self._h)

    def counters(self):
        c = np.zeros(8, np.uint64); lib().oracle_counters(self._h, c.ctypes.data_as(_U64))
        return dict(primary=int(c[0]), bounce=int(c[1]), shadow=int(c[2]), nodes=int(c[3]), cubes=int(c[4]), tris=int(c[5]), primary_hits=int(c[6]))


def slab(inv_ray6, cube6):
    t = C.c_float(0)
    hit = lib().oracle_slab(_fp(np.asarray(inv_ray6, np.float32)), _fp(np.asarray(cube6, np.float32)), C.byref(t))
    return (hit != 0), t.value


def moller_trumbore(ray6, tri9):
    tuv = np.zeros(3, np.float32)
    hit = lib().oracle_mt(_fp(np.asarray(ray6, np.float32)), _fp(np.asarray(tri9, np.float32)), _fp(tuv))
    return (hit != 0), tuv


def mat_mul(a, b):
    out = np.zeros(16, np.float32); lib().oracle_mat_mul(_fp(np.asarray(a, np.float32)), _fp(np.asarray(b, np.float32)), _fp(out)); return out


def mat_vec(m, v):
    out = np.zeros(4, np.float32); lib().oracle_mat_vec(_fp(np.asarray(m, np.float32)), _fp(np.asarray(v, np.float32)), _fp(out)); return out


def collada_matrix_to_vecmath(c16):
    out = np.zeros(16, np.float32); lib().oracle_collada_matrix_to_vecmath(_fp(np.asarray(c16, np.float32)), _fp(out)); return out


def pow32(x): return lib().oracle_pow32(float(x))
def tonemap_pack(r, g, b): return lib().oracle_tonemap_pack(float(r), float(g), float(b))


def pcg4d(v4):
    out = np.zeros(4, np.uint32); lib().oracle_pcg4d(_up(np.asarray(v4, np.uint32)), _up(out)); return out
