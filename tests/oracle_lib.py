"""ctypes binding of the CPU oracle (oracle/libpovu_oracle.so) -- tests only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libpovu_oracle.so")


class Forest(C.Structure):
    _fields_ = [("n_comp", C.c_uint32), ("comp_nv", C.POINTER(C.c_uint32)), ("comp_ne", C.POINTER(C.c_uint32)),
                ("text", C.POINTER(C.c_void_p)), ("text_len", C.POINTER(C.c_size_t)),
                ("n_pvst", C.POINTER(C.c_uint32)), ("total_flubbles", C.c_uint64),
                ("t_componetize", C.c_double), ("t_tree", C.c_double), ("t_classes", C.c_double),
                ("t_stack", C.c_double), ("t_pvst", C.c_double), ("t_wall_components", C.c_double),
                ("threads", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_ROOT, "oracle", "povu_oracle.c")
        if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s"])
        _lib = C.CDLL(_SO)
        _lib.orc_decompose_arrays.restype = C.POINTER(Forest)
        _lib.orc_decompose_arrays.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_decompose_arrays_mt.restype = C.POINTER(Forest)
        _lib.orc_decompose_arrays_mt.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.orc_forest_free.argtypes = [C.POINTER(Forest)]
        _lib.orc_decompose_gfa.restype = C.c_int
        _lib.orc_decompose_gfa.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        _lib.orc_set_faithful_rescan.argtypes = [C.c_int]
        _lib.orc_set_leaf_subflubbles.argtypes = [C.c_int]
    return _lib


def decompose(links, tips=None, want_text=True, timings=False, threads=1, lpt=False, leaf=False):
    """Run the oracle on a workloads.Links. Returns {component_id: pvst_text}
    (component ids are 1-based, skipped components absent).  threads > 1 runs the per-component part
    on that many threads: the reference's contiguous chunks (decompose.cpp:78-92,116-157) or, with
    lpt, components bin-packed by size.  leaf=True: the two relabelling passes of `-s` (find_tiny, find_parallel)
    run on every PVST, so the text carries T / O lines."""
    l = lib()
    vid = np.ascontiguousarray(links.vid, dtype=np.uint32)
    v1 = np.ascontiguousarray(links.v1, dtype=np.uint32)
    v2 = np.ascontiguousarray(links.v2, dtype=np.uint32)
    s1 = np.ascontiguousarray(links.s1, dtype=np.uint8)
    s2 = np.ascontiguousarray(links.s2, dtype=np.uint8)
    tp = None
    if tips is not None:
        tips = np.ascontiguousarray(tips, dtype=np.uint8)
        tp = tips.ctypes.data
    l.orc_set_leaf_subflubbles(int(leaf))  # (True = 1: the two relabelling passes; 2: all five passes of -s)
    try:
        f = l.orc_decompose_arrays_mt(len(vid), vid.ctypes.data, len(v1), v1.ctypes.data, s1.ctypes.data,
                                      v2.ctypes.data, s2.ctypes.data, tp, 1 if want_text else 0, int(threads),
                                      1 if lpt else 0)
    finally:
        l.orc_set_leaf_subflubbles(0)
    fo = f.contents
    out = {}
    for c in range(fo.n_comp):
        if fo.text[c]:
            out[c + 1] = C.string_at(fo.text[c], fo.text_len[c]).decode()
    info = dict(n_comp=fo.n_comp, comp_nv=[fo.comp_nv[c] for c in range(min(fo.n_comp, 100000))],
                n_pvst=[fo.n_pvst[c] for c in range(min(fo.n_comp, 100000))],
                total_flubbles=fo.total_flubbles, t_componetize=fo.t_componetize, t_tree=fo.t_tree,
                t_classes=fo.t_classes, t_stack=fo.t_stack, t_pvst=fo.t_pvst,
                t_wall_components=fo.t_wall_components, threads=fo.threads)
    l.orc_forest_free(f)
    return (out, info) if timings else out


def decompose_gfa(path: str, outdir: str) -> int:
    err = C.create_string_buffer(1024)
    n = lib().orc_decompose_gfa(path.encode(), outdir.encode(), err, 1024)
    if n < 0:
        raise RuntimeError(err.value.decode())
    return n
