"""ctypes front-end of oracle/mc_lewiner.c (CPU ORACLE -- test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmc_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "mc_lewiner.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.run(["make", "-C", HERE, "-s"], check=True)
        _lib = C.CDLL(LIB)
        _lib.mc_lewiner.restype = C.c_int
        _lib.mc_lewiner.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        _lib.mc_free.argtypes = [C.c_void_p]
    return _lib


def marching_cubes(volume, level):
    """Same return convention as skimage.measure.marching_cubes(volume, level):
    (verts (V,3) f32, faces (F,3) i32, normals (V,3) f32, values (V,) f32); RuntimeError when empty."""
    lib = load()
    vol = np.ascontiguousarray(volume, dtype=np.float32)
    level = float(level)
    if level < vol.min() or level > vol.max():
        raise ValueError("Surface level must be within volume data range.")
    pv, pf, pn, pval = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    nv, nf = C.c_int(), C.c_int()
    rc = lib.mc_lewiner(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2], level, C.byref(pv), C.byref(nv),
                        C.byref(pf), C.byref(nf), C.byref(pn), C.byref(pval))

    def take(ptr, n, dtype, cols):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float if dtype == np.float32 else C.c_int)),
                                  shape=(max(n * cols, 1),))[:n * cols].astype(dtype).copy()
        lib.mc_free(ptr)
        return a.reshape(n, cols) if cols > 1 else a

    verts = take(pv, nv.value, np.float32, 3)
    faces = take(pf, nf.value, np.int32, 3)
    normals = take(pn, nv.value, np.float32, 3)
    values = take(pval, nv.value, np.float32, 1)
    if rc != 0:
        raise RuntimeError("No surface found at the given iso value.")
    return verts, faces, normals, values
