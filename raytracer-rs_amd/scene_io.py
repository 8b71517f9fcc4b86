"""Reader for the flat scene container written by csrc/collada.cpp:write_scene_file
(tools/dae2scene).  Pure data handling: lets hosts without the .dae files (the GPU box) feed the
same parsed arrays to the C ABI (mi355rt_create) and, in tests, to the CPU oracle."""
import struct

import numpy as np

MAGIC = b"M355SCN1"


def load_scene_file(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != MAGIC:
        raise ValueError("not a scene file: %s" % path)
    ntri, nmat, nlight, ntex, ncam = struct.unpack_from("<5I", data, 8)
    off = 28
    tri_verts = np.frombuffer(data, np.float32, ntri * 9, off).reshape(ntri, 9).copy(); off += ntri * 36
    tri_geom = np.frombuffer(data, np.uint32, ntri, off).copy(); off += ntri * 4
    mat_kind = np.zeros(nmat, np.uint32); mat_rgb = np.zeros((nmat, 3), np.float32); mat_tex = np.zeros(nmat, np.uint32)
    for i in range(nmat):
        mat_kind[i] = struct.unpack_from("<I", data, off)[0]
        mat_rgb[i] = np.frombuffer(data, np.float32, 3, off + 4)
        mat_tex[i] = struct.unpack_from("<I", data, off + 16)[0]
        off += 20
    lights = np.frombuffer(data, np.float32, nlight * 6, off).reshape(nlight, 6).copy(); off += nlight * 24
    cams = []
    for _ in range(ncam):
        m = np.frombuffer(data, np.float32, 16, off).copy()
        fov = struct.unpack_from("<f", data, off + 64)[0]
        cams.append((m, fov)); off += 68
    textures = []
    for _ in range(ntex):
        w, h, as_bytes = struct.unpack_from("<3I", data, off); off += 12
        if as_bytes:
            t = np.frombuffer(data, np.uint8, w * h * 3, off).astype(np.float32) / np.float32(256.0); off += w * h * 3
        else:
            t = np.frombuffer(data, np.float32, w * h * 3, off).copy(); off += w * h * 12
        textures.append(t.reshape(h, w, 3))
    if not cams:
        raise ValueError("scene has no camera")
    return dict(tri_verts=tri_verts, tri_geom=tri_geom, mat_kind=mat_kind, mat_rgb=mat_rgb, mat_tex=mat_tex,
                lights=lights, textures=textures, camera_matrix=cams[0][0], camera_fov=np.float32(cams[0][1]),
                cameras=cams)
