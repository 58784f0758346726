"""ctypes binding of libnrhip.so (include/nrhip.h).  No fallback: if the HIP library is missing or a
call fails, an exception is raised -- the product never silently computes on the CPU."""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', os.environ.get('NRHIP_LIB_NAME', 'libnrhip.so'))   # NRHIP_LIB_NAME: a build variant (build.sh)

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)


class NrhipError(RuntimeError):
    pass


_lib = None


def _sig(lib, name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


def load():
    """Load libnrhip.so (built by build.sh / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NrhipError(f"{LIB_PATH} not found: run ./build.sh (hipcc --offload-arch=gfx950) first")
    lib = ctypes.CDLL(LIB_PATH)
    i32, i64, dbl, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
    _sig(lib, 'nrhip_last_error', ctypes.c_char_p, [])
    _sig(lib, 'nrhip_device_count', ctypes.c_int, [])
    _sig(lib, 'nrhip_ctx_create', ctypes.c_int, [ctypes.c_int, dbl, dbl, dbl, ctypes.c_int, c_void_pp])
    _sig(lib, 'nrhip_ctx_destroy', None, [vp])
    _sig(lib, 'nrhip_synchronize', ctypes.c_int, [vp])
    _sig(lib, 'nrhip_malloc', ctypes.c_int, [vp, ctypes.c_uint64, c_void_pp])
    _sig(lib, 'nrhip_free', ctypes.c_int, [vp, vp])
    _sig(lib, 'nrhip_memcpy_h2d', ctypes.c_int, [vp, vp, vp, ctypes.c_uint64])
    _sig(lib, 'nrhip_memcpy_d2h', ctypes.c_int, [vp, vp, vp, ctypes.c_uint64])
    _sig(lib, 'nrhip_find_solutions_batch', ctypes.c_int,
         [vp, i64, c_double_p, c_double_p, i32, c_int32_p, c_int32_p] + [c_double_p] * 7)
    _sig(lib, 'nrhip_ctx_set_ray_finder', ctypes.c_int, [vp, i32])
    _sig(lib, 'nrhip_ctx_set_gl3_table', ctypes.c_int, [vp, i32, c_double_p, c_double_p, c_double_p])
    _sig(lib, 'nrhip_ray_records_batch', ctypes.c_int,
         [vp, i64, c_double_p, c_double_p, i32, c_double_p, c_int32_p, c_int32_p] + [c_double_p] * 7)
    _sig(lib, 'nrhip_attenuation_batch', ctypes.c_int,
         [vp, i64, c_double_p, c_double_p, c_double_p, i32, c_double_p, c_double_p, c_int32_p])
    _sig(lib, 'nrhip_attenuation_last_overflow', ctypes.c_int64, [vp])
    _sig(lib, 'nrhip_attenuation_length', ctypes.c_int, [vp, i64, c_double_p, c_double_p, c_double_p])
    refl_sig = [vp, i64, c_double_p, c_double_p, i32, i32, ctypes.c_double, c_int32_p, c_int32_p, c_double_p, c_double_p,
                c_int32_p, c_int32_p] + [c_double_p] * 5 + [c_int32_p, c_int32_p]
    _sig(lib, 'nrhip_find_solutions_reflections_batch', ctypes.c_int, refl_sig)
    _sig(lib, 'nrhip_ray_records_reflections_batch', ctypes.c_int, refl_sig)
    _sig(lib, 'nrhip_attenuation_reflections_batch', ctypes.c_int,
         [vp, i64, c_double_p, c_double_p, c_double_p, c_int32_p, c_int32_p, ctypes.c_double, i32, c_double_p, c_double_p,
          c_double_p])
    _sig(lib, 'nrhip_arz_time_trace_batch', ctypes.c_int,
         [vp, i64, c_double_p, c_double_p, c_double_p, c_int32_p, c_double_p, c_int32_p, c_double_p, i32, i32, c_double_p,
          c_double_p, c_double_p, i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, i32, ctypes.c_double, c_double_p,
          c_double_p])
    _sig(lib, 'nrhip_birefringence_batch', ctypes.c_int,
         [vp, i64, c_double_p, c_double_p, c_double_p, c_double_p, c_int32_p, c_double_p, c_double_p, ctypes.c_double,
          ctypes.c_double, i32, ctypes.c_double, c_double_p, c_double_p])
    _sig(lib, 'nrhip_earth_weights_batch', ctypes.c_int,
         [vp, i64, c_double_p, c_double_p, c_int32_p, c_double_p, c_double_p, i32, i32, ctypes.c_void_p, ctypes.c_double,
          ctypes.c_double, c_double_p, c_double_p])
    for name, sig in _OPTIONAL.items():
        if hasattr(lib, name):
            _sig(lib, name, *sig)
    _lib = lib
    return lib


_OPTIONAL = {}


def check(status):
    if status != 0:
        raise NrhipError(load().nrhip_last_error().decode())


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    return a.ctypes.data_as(c_int32_p)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a
