"""ORACLE SUPPORT (test infrastructure only): ctypes access to oracle/_ref/libmpc_ref_math.so = the reference's utils/math_utils.h compiled from where it lies under
/root/reference (oracle/ref_math.cpp; `make -C oracle ref`).  It is the only part of the reference that builds in this image (everything else needs Eigen / corbo /
ROS / teb headers, which are absent, and no stand-ins are written for them).  Where the reference tree is absent (the GPU box) and the library was not shipped, the tests
use the recorded vectors tests/golden/ref_math_utils.npz (generator: tests/golden/make_ref_math_vectors.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "libmpc_ref_math.so")
REFERENCE_INCLUDE = "/root/reference/mpc_local_planner/include"
_lib = None


def build() -> bool:
    """compiles oracle/_ref when the reference tree is present; True if the library exists afterwards"""
    if os.path.isdir(REFERENCE_INCLUDE):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)
    return os.path.exists(LIB)


def load():
    global _lib
    if _lib is None:
        if not build():
            return None
        _lib = C.CDLL(LIB)
        _lib.ref_average_angles.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def normalize_theta(theta):
    t = np.ascontiguousarray(theta, float).ravel(); out = np.empty_like(t)
    load().ref_normalize_theta(C.c_int(t.size), _p(t), _p(out))
    return out.reshape(np.shape(theta))


def interpolate_angle(a1, a2, factor):
    a1, a2, f = (np.ascontiguousarray(np.broadcast_to(v, np.broadcast(a1, a2, factor).shape), float).ravel() for v in (a1, a2, factor))
    out = np.empty_like(a1)
    load().ref_interpolate_angle(C.c_int(a1.size), _p(a1), _p(a2), _p(f), _p(out))
    return out


def average_angles(angles) -> float:
    a = np.ascontiguousarray(angles, float).ravel()
    return float(load().ref_average_angles(C.c_int(a.size), _p(a)))


def distance_points2d(p1, p2):
    """(templated overload, scalar overload) of inc/utils/math_utils.h:57-64 on rows of p1, p2 (N, 2)"""
    p1 = np.ascontiguousarray(p1, float).reshape(-1, 2); p2 = np.ascontiguousarray(p2, float).reshape(-1, 2)
    a = np.empty(p1.shape[0]); b = np.empty(p1.shape[0])
    load().ref_distance_points2d(C.c_int(p1.shape[0]), _p(p1), _p(p2), _p(a), _p(b))
    return a, b


def cross2d(v1, v2):
    v1 = np.ascontiguousarray(v1, float).reshape(-1, 2); v2 = np.ascontiguousarray(v2, float).reshape(-1, 2)
    out = np.empty(v1.shape[0])
    load().ref_cross2d(C.c_int(v1.shape[0]), _p(v1), _p(v2), _p(out))
    return out
