"""ctypes front-end of oracle/libct_oracle.so (the plain-C oracle, ct_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of ct_oracle.c.  numpy in, numpy out.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

WORKFLOWS = {"metallic": 0, "specular": 1, "converted": 2}
LIGHT_TYPES = {"directional": 0, "point": 1}


class _Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "batch", "height", "width", "height_total", "y_offset", "light_type", "n_lights",
        "workflow", "albedo_is_srgb", "specular_is_srgb", "return_srgb", "has_normal")] + [
        ("view", ctypes.c_double * 3), ("light_size", ctypes.c_double),
        ("lights", ctypes.POINTER(ctypes.c_double)), ("intensities", ctypes.POINTER(ctypes.c_double))]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libct_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _prep(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def render(albedo, normal, roughness, metallic=None, specular=None, *, view, lights, intensities,
           light_type="point", light_size=None, workflow="metallic", albedo_is_srgb=True,
           specular_is_srgb=True, return_srgb=True, y_offset=0, H_total=None, dtype=np.float32,
           threads=None):
    """[B,C,H,W] (or [C,H,W]) planar maps -> [B,3,H,W] (or [3,H,W]).  lights/intensities: [L,3] or [3]."""
    dt = np.dtype(dtype)
    suf = "f32" if dt == np.float32 else "f64"
    squeeze = albedo.ndim == 3
    A, N, R, M, S = [None if t is None else _prep(t[None] if squeeze else t, dt)
                     for t in (albedo, normal, roughness, metallic, specular)]
    B, _, H, W = A.shape
    lights = np.ascontiguousarray(np.atleast_2d(np.asarray(lights, dtype=np.float64)))
    intens = np.ascontiguousarray(np.atleast_2d(np.asarray(intensities, dtype=np.float64)))
    assert lights.shape == intens.shape and lights.shape[1] == 3
    p = _Params()
    p.batch, p.height, p.width = B, H, W
    p.height_total = H if H_total is None else H_total
    p.y_offset = y_offset
    p.light_type = LIGHT_TYPES[light_type]
    p.n_lights = lights.shape[0]
    p.workflow = WORKFLOWS[workflow]
    p.albedo_is_srgb, p.specular_is_srgb, p.return_srgb = int(albedo_is_srgb), int(specular_is_srgb), int(return_srgb)
    p.has_normal = int(N is not None)
    v = np.asarray(view, dtype=np.float64)
    p.view[0], p.view[1], p.view[2] = v
    p.light_size = float(light_size or 1.0)
    p.lights = lights.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    p.intensities = intens.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    out = np.empty((B, 3, H, W), dtype=dt)
    if threads is not None:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    fn = getattr(lib(), "ct_oracle_render_" + suf)
    fn.restype = ctypes.c_int
    rc = fn(ctypes.byref(p), _ptr(A), _ptr(N), _ptr(R), _ptr(M), _ptr(S), _ptr(out))
    if rc != 0:
        raise ValueError("ct_oracle_render failed: %d" % rc)
    return out[0] if squeeze else out


def _elementwise(name, x, dtype):
    dt = np.dtype(dtype)
    a = _prep(x, dt)
    out = np.empty_like(a)
    fn = getattr(lib(), f"ct_oracle_{name}_" + ("f32" if dt == np.float32 else "f64"))
    fn.restype = None
    fn(_ptr(a), _ptr(out), ctypes.c_size_t(a.size))
    return out


def srgb_to_linear(x, dtype=np.float32):
    return _elementwise("srgb_to_linear", x, dtype)


def linear_to_srgb(x, dtype=np.float32):
    return _elementwise("linear_to_srgb", x, dtype)


def metallic_to_specular(albedo_lin, metallic, dtype=np.float32):
    dt = np.dtype(dtype)
    a, m = _prep(albedo_lin, dt), _prep(metallic, dt)
    d, s = np.empty_like(a), np.empty_like(a)
    P = a.shape[-1] * a.shape[-2]
    fn = getattr(lib(), "ct_oracle_metallic_to_specular_" + ("f32" if dt == np.float32 else "f64"))
    fn.restype = None
    fn(_ptr(a), _ptr(m), _ptr(d), _ptr(s), ctypes.c_size_t(P))
    return d, s


def specular_to_metallic(diffuse_lin, specular_raw, dtype=np.float32):
    dt = np.dtype(dtype)
    d, s = _prep(diffuse_lin, dt), _prep(specular_raw, dt)
    b, m = np.empty_like(d), np.empty_like(d)
    P = d.shape[-1] * d.shape[-2]
    fn = getattr(lib(), "ct_oracle_specular_to_metallic_" + ("f32" if dt == np.float32 else "f64"))
    fn.restype = None
    fn(_ptr(d), _ptr(s), _ptr(b), _ptr(m), ctypes.c_size_t(P))
    return b, m


def decode_normal(n, dtype=np.float32):
    dt = np.dtype(dtype)
    a = _prep(n, dt)
    C, H, W = a.shape
    out = np.empty((3, H, W), dtype=dt)
    fn = getattr(lib(), "ct_oracle_decode_normal_" + ("f32" if dt == np.float32 else "f64"))
    fn.restype = ctypes.c_int
    if fn(_ptr(a), _ptr(out), ctypes.c_int(C), ctypes.c_size_t(H * W)) != 0:
        raise ValueError("Normal map must have 2 or 3 channels.")
    return out
