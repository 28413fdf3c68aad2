"""ctypes mirror of include/povu_hip.h.

Fails loudly when the HIP library is missing or no GPU is visible: there is no CPU
fallback on the product path (the CPU oracle lives under oracle/ and is test-only).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class HipUnavailable(RuntimeError):
    pass


def lib_path() -> str:
    return os.path.join(_HERE, "lib", "libpovu_hip.so")


class _Opts(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("flags", C.c_uint32)]


class _SubTree(C.Structure):
    _fields_ = [("n_total", C.c_uint32), ("n_flubble_like", C.c_uint32), ("n_concealed", C.c_uint32), ("n_midi", C.c_uint32),
                ("n_smothered", C.c_uint32), ("fam", C.POINTER(C.c_uint8)), ("or1", C.POINTER(C.c_uint8)),
                ("or2", C.POINTER(C.c_uint8)), ("route", C.POINTER(C.c_uint8)), ("id1", C.POINTER(C.c_uint32)),
                ("id2", C.POINTER(C.c_uint32)), ("child_off", C.POINTER(C.c_uint32)), ("child", C.POINTER(C.c_uint32))]


class _Tree(C.Structure):
    _fields_ = [("component_id", C.c_uint32), ("n_vtx", C.c_uint32), ("n_links", C.c_uint32),
                ("n_pvst", C.c_uint32), ("a_id", C.POINTER(C.c_uint32)), ("z_id", C.POINTER(C.c_uint32)),
                ("a_or", C.POINTER(C.c_uint8)), ("z_or", C.POINTER(C.c_uint8)), ("parent", C.POINTER(C.c_uint32)),
                ("n_hairpins", C.c_uint32), ("hairpins", C.POINTER(C.c_uint64))]


class _ShardInfo(C.Structure):
    _fields_ = [("n_vtx", C.c_uint32), ("n_links", C.c_uint32), ("n_components", C.c_uint32), ("weight", C.c_uint64),
                ("bytes", C.c_size_t), ("device_ptr", C.c_void_p)]


class _MultiRank(C.Structure):
    _fields_ = [("device", C.c_int), ("n_vtx", C.c_uint32), ("n_links", C.c_uint32), ("n_components", C.c_uint32),
                ("shard_bytes", C.c_uint64), ("recv_ms", C.c_double), ("csr_ms", C.c_double), ("decompose_ms", C.c_double),
                ("sink_ms", C.c_double), ("h2d", C.c_uint64), ("d2h", C.c_uint64), ("peer_out", C.c_uint64),
                ("peer_in", C.c_uint64)]


class _StageTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_double), ("launches", C.c_uint32)]


F_HAIRPINS = 1
F_SEQUENTIAL = 2
F_SEQ_TREE = 4
F_FORCE_REDO = 8
F_SORTED_ADJ = 16
F_NO_STAGE_TIMES = 32
F_BIG_CLASS_DFS = 64
F_SPARSE_SPLITTERS = 128
F_REDO_ODD = 256
F_ALL_VERTEX_CLASSES = 512
F_CHECK_LAMINAR = 1024
F_LEAF_SUBFLUBBLES = 2048
F_ASYNC = 4096
F_SUBFLUBBLES = 8192  # all five passes of -s (implies F_LEAF_SUBFLUBBLES): Forest.texts() then carry C / M / S lines

_lib = None


def load_lib():
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise HipUnavailable(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback)")
    l = C.CDLL(p)
    l.povu_hip_device_count.restype = C.c_int
    l.povu_hip_create.restype = C.c_void_p
    l.povu_hip_create.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    l.povu_hip_destroy.argtypes = [C.c_void_p]
    l.povu_hip_graph_upload.restype = C.c_int
    l.povu_hip_graph_upload.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    l.povu_hip_last_upload_times.restype = C.c_int
    l.povu_hip_last_upload_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    l.povu_hip_decompose.restype = C.c_void_p
    l.povu_hip_decompose.argtypes = [C.c_void_p, C.POINTER(_Opts), C.c_char_p, C.c_size_t]
    l.povu_hip_forest_total_components.restype = C.c_uint32
    l.povu_hip_forest_total_components.argtypes = [C.c_void_p]
    l.povu_hip_forest_tree_count.restype = C.c_uint32
    l.povu_hip_forest_tree_count.argtypes = [C.c_void_p]
    l.povu_hip_forest_get.restype = C.c_int
    l.povu_hip_forest_get.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_Tree)]
    l.povu_hip_forest_free.argtypes = [C.c_void_p]
    l.povu_hip_forest_wait.restype = C.c_int
    l.povu_hip_forest_wait.argtypes = [C.c_void_p]
    l.povu_hip_forest_pass_ms.restype = C.c_double
    l.povu_hip_forest_pass_ms.argtypes = [C.c_void_p]
    l.povu_hip_forest_span_ms.restype = C.c_double
    l.povu_hip_forest_span_ms.argtypes = [C.c_void_p, C.c_void_p]
    l.povu_hip_forest_get_subtree.restype = C.c_int
    l.povu_hip_forest_get_subtree.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    l.povu_hip_forest_get_sub.restype = C.c_int
    l.povu_hip_forest_get_sub.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(C.c_uint32)),
                                          C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint8))]
    l.povu_hip_forest_raw.restype = C.c_int
    l.povu_hip_forest_raw.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64)]
    l.povu_hip_forest_first.restype = C.c_uint64
    l.povu_hip_forest_first.argtypes = [C.c_void_p, C.c_uint32]
    l.povu_hip_forest_pvst_text.restype = C.c_void_p
    l.povu_hip_forest_pvst_text.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t)]
    l.povu_hip_buffer_free.argtypes = [C.c_void_p]
    l.povu_hip_last_stage_times.restype = C.c_int
    l.povu_hip_last_stage_times.argtypes = [C.c_void_p, C.POINTER(_StageTime), C.c_int]
    l.povu_hip_last_seq_redo.restype = C.c_uint32
    l.povu_hip_last_seq_redo.argtypes = [C.c_void_p]
    l.povu_hip_last_links_processed.restype = C.c_uint64
    l.povu_hip_last_links_processed.argtypes = [C.c_void_p]
    l.povu_hip_debug_components.restype = C.c_int
    l.povu_hip_debug_components.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    l.povu_hip_debug_tree.restype = C.c_int
    l.povu_hip_debug_tree.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
    l.povu_hip_last_black_only_classes.restype = C.c_int
    l.povu_hip_last_black_only_classes.argtypes = [C.c_void_p]
    l.povu_hip_last_crossings.restype = C.c_int
    l.povu_hip_last_crossings.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    l.povu_hip_last_laminar_check_ran.restype = C.c_int
    l.povu_hip_last_laminar_check_ran.argtypes = [C.c_void_p]
    l.povu_hip_debug_edge_ids.restype = C.c_int
    l.povu_hip_debug_edge_ids.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
    l.povu_hip_debug_stack.restype = C.c_int
    l.povu_hip_debug_stack.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p,
                                       C.c_void_p]
    l.povu_hip_version.restype = C.c_char_p
    l.povu_hip_debug_scan.restype = C.c_int
    l.povu_hip_debug_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.c_size_t]
    l.povu_hip_workspace_estimate.restype = C.c_uint64
    l.povu_hip_workspace_estimate.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    l.povu_hip_prewarm.restype = C.c_int
    l.povu_hip_prewarm.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    l.povu_hip_leaf_workspace_estimate.restype = C.c_uint64
    l.povu_hip_leaf_workspace_estimate.argtypes = [C.c_uint32, C.c_uint32]
    # ---- multi-GPU sharding
    l.povu_hip_lpt_assign.restype = C.c_int
    l.povu_hip_lpt_assign.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    l.povu_hip_shard_partition.restype = C.c_void_p
    l.povu_hip_shard_partition.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
    l.povu_hip_shards_world.restype = C.c_uint32
    l.povu_hip_shards_world.argtypes = [C.c_void_p]
    l.povu_hip_shards_total_components.restype = C.c_uint32
    l.povu_hip_shards_total_components.argtypes = [C.c_void_p]
    l.povu_hip_shards_get.restype = C.c_int
    l.povu_hip_shards_get.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_ShardInfo)]
    l.povu_hip_shards_times.restype = C.c_int
    l.povu_hip_shards_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    l.povu_hip_shards_export.restype = C.c_int
    l.povu_hip_shards_export.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    l.povu_hip_shards_free.argtypes = [C.c_void_p]
    l.povu_hip_graph_upload_shard.restype = C.c_int
    l.povu_hip_graph_upload_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
    l.povu_hip_shard_total_components.restype = C.c_uint32
    l.povu_hip_shard_total_components.argtypes = [C.c_void_p]
    l.povu_hip_forest_globalize.restype = C.c_int
    l.povu_hip_forest_globalize.argtypes = [C.c_void_p, C.c_void_p]
    l.povu_hip_forest_pack_size.restype = C.c_size_t
    l.povu_hip_forest_pack_size.argtypes = [C.c_void_p]
    l.povu_hip_forest_pack.restype = C.c_int
    l.povu_hip_forest_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    l.povu_hip_forest_merge.restype = C.c_void_p
    l.povu_hip_forest_merge.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_char_p,
                                        C.c_size_t]
    l.povu_hip_comm_unique_id.restype = C.c_int
    l.povu_hip_comm_unique_id.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    l.povu_hip_comm_create.restype = C.c_void_p
    l.povu_hip_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    l.povu_hip_comm_destroy.argtypes = [C.c_void_p]
    l.povu_hip_comm_scatter.restype = C.c_int
    l.povu_hip_comm_scatter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    l.povu_hip_comm_gather.restype = C.c_void_p
    l.povu_hip_comm_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    l.povu_hip_comm_times.restype = C.c_int
    l.povu_hip_comm_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    l.povu_hip_share_results.restype = C.c_int
    l.povu_hip_share_results.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    l.povu_hip_forest_share.restype = C.c_int
    l.povu_hip_forest_share.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    l.povu_hip_forest_attach.restype = C.c_void_p
    l.povu_hip_forest_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint32, C.c_char_p,
                                         C.c_size_t]
    l.povu_hip_transfer_bytes.restype = C.c_int
    l.povu_hip_transfer_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    # ---- one process, N GPUs
    l.povu_hip_multi_create.restype = C.c_void_p
    l.povu_hip_multi_create.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.c_char_p, C.c_size_t]
    l.povu_hip_multi_destroy.argtypes = [C.c_void_p]
    l.povu_hip_multi_world.restype = C.c_uint32
    l.povu_hip_multi_world.argtypes = [C.c_void_p]
    l.povu_hip_multi_upload.restype = C.c_int
    l.povu_hip_multi_upload.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    l.povu_hip_multi_scatter.restype = C.c_int
    l.povu_hip_multi_scatter.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    l.povu_hip_multi_decompose.restype = C.c_void_p
    l.povu_hip_multi_decompose.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    l.povu_hip_multi_rank.restype = C.c_int
    l.povu_hip_multi_rank.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_MultiRank)]
    l.povu_hip_multi_times.restype = C.c_int
    l.povu_hip_multi_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    l.povu_hip_multi_transport.restype = C.c_char_p
    l.povu_hip_multi_transport.argtypes = [C.c_void_p]
    l.povu_hip_gfa_write.restype = C.c_int
    l.povu_hip_gfa_write.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_char_p, C.c_size_t]
    _lib = l
    return l


def write_gfa(links, path: str) -> None:
    """GFA v1 text of a workloads.Links through the host library's writer (host only, no GPU needed)."""
    l = load_lib()
    vid = np.ascontiguousarray(links.vid, dtype=np.uint32)
    v1 = np.ascontiguousarray(links.v1, dtype=np.uint32)
    v2 = np.ascontiguousarray(links.v2, dtype=np.uint32)
    s1 = np.ascontiguousarray(links.s1, dtype=np.uint8)
    s2 = np.ascontiguousarray(links.s2, dtype=np.uint8)
    err = C.create_string_buffer(512)
    if l.povu_hip_gfa_write(path.encode(), len(vid), vid.ctypes.data, len(v1), v1.ctypes.data, s1.ctypes.data,
                            v2.ctypes.data, s2.ctypes.data, err, 512) != 0:
        raise RuntimeError(err.value.decode())


def lpt_assign(weights, world: int) -> np.ndarray:
    """The library's LPT rule (host only): heaviest first (stable), least loaded rank, +1 per item."""
    w = np.ascontiguousarray(weights, dtype=np.uint64)
    out = np.zeros(len(w), dtype=np.uint32)
    if load_lib().povu_hip_lpt_assign(w.ctypes.data, len(w), world, out.ctypes.data) != 0:
        raise ValueError("bad LPT arguments")
    return out


@dataclass
class PvstTree:
    component_id: int
    n_vtx: int
    n_links: int
    a_id: np.ndarray
    z_id: np.ndarray
    a_or: np.ndarray
    z_or: np.ndarray
    parent: np.ndarray
    hairpins: np.ndarray
    text: Optional[str] = None


class Forest:
    """Result of one decompose call (host memory)."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.povu_hip_forest_free(self._h)
            self._h = None

    def wait(self) -> "Forest":
        """Completes a forest of a decompose with F_ASYNC (its arrays may still be on their way to the host)."""
        self._lib.povu_hip_forest_wait(self._h)
        return self

    def pass_ms(self) -> float:
        """HIP-event time of the pass that produced this forest: first kernel to last byte on the host (waits first)."""
        return float(self._lib.povu_hip_forest_pass_ms(self._h))

    def span_ms(self, last: "Forest") -> float:
        """HIP-event time from this forest's first kernel to the last byte of `last` (a later pass of the same context)."""
        return float(self._lib.povu_hip_forest_span_ms(self._h, last._h))

    @property
    def total_components(self) -> int:
        return self._lib.povu_hip_forest_total_components(self._h)

    def __len__(self) -> int:
        return self._lib.povu_hip_forest_tree_count(self._h)

    def pvst_sizes(self) -> List[int]:
        """PVST vertices of every tree (no array is copied)."""
        out = []
        t = _Tree()
        for i in range(len(self)):
            if self._lib.povu_hip_forest_get(self._h, i, C.byref(t)) != 0:
                raise IndexError(i)
            out.append(int(t.n_pvst))
        return out

    def tree(self, i: int, with_text: bool = False) -> PvstTree:
        t = _Tree()
        if self._lib.povu_hip_forest_get(self._h, i, C.byref(t)) != 0:
            raise IndexError(i)
        n = t.n_pvst
        arr = lambda p, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True)  # noqa: E731
        hp = (np.ctypeslib.as_array(t.hairpins, shape=(2 * t.n_hairpins,)).reshape(-1, 2).copy()
              if t.n_hairpins else np.zeros((0, 2), dtype=np.uint64))
        out = PvstTree(t.component_id, t.n_vtx, t.n_links, arr(t.a_id, np.uint32), arr(t.z_id, np.uint32),
                       arr(t.a_or, np.uint8), arr(t.z_or, np.uint8), arr(t.parent, np.uint32), hp)
        if with_text:
            out.text = self.text(i)
        return out

    def sub(self, i: int):
        """(ai, zi, line letters) of tree i after a decompose with F_LEAF_SUBFLUBBLES: compute_ai_zi's spanning-tree
        vertices (flubbles.cpp:264-290) and 'D' / 'F' / 'T' / 'O' per PVST vertex."""
        t = _Tree()
        if self._lib.povu_hip_forest_get(self._h, i, C.byref(t)) != 0:
            raise IndexError(i)
        ai, zi, fam = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint8)()
        rc = self._lib.povu_hip_forest_get_sub(self._h, i, C.byref(ai), C.byref(zi), C.byref(fam))
        if rc != 0:
            raise RuntimeError(f"forest carries no subflubble labels (rc {rc})")
        n = t.n_pvst
        return (np.ctypeslib.as_array(ai, shape=(n,)).copy(), np.ctypeslib.as_array(zi, shape=(n,)).copy(),
                np.ctypeslib.as_array(fam, shape=(n,)).copy())

    def subtree(self, i: int):
        """Tree i after all five passes of -s (a decompose with F_SUBFLUBBLES): dict of n_total, n_flubble_like, n_concealed,
        n_midi, n_smothered, fam / or1 / or2 / route (uint8) and id1 / id2 (uint32) per vertex, child_off (uint32, relative) and
        child (uint32): children of vertex v = child[child_off[v]:child_off[v + 1]]."""
        st = _SubTree()
        rc = self._lib.povu_hip_forest_get_subtree(self._h, i, C.byref(st))
        if rc != 0:
            raise RuntimeError(f"forest carries no subflubble trees (rc {rc})")
        n = st.n_total
        a8 = lambda p: np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint8)
        a32 = lambda p: np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        off = np.ctypeslib.as_array(st.child_off, shape=(n + 1,)).copy()
        lo, hi = int(off[0]), int(off[-1])
        child = np.ctypeslib.as_array(st.child, shape=(max(hi, 1),))[lo:hi].copy()
        return dict(n_total=n, n_flubble_like=st.n_flubble_like, n_concealed=st.n_concealed, n_midi=st.n_midi,
                    n_smothered=st.n_smothered, fam=a8(st.fam), or1=a8(st.or1), or2=a8(st.or2), route=a8(st.route),
                    id1=a32(st.id1), id2=a32(st.id2), child_off=off - lo, child=child)

    def raw(self):
        """Zero-copy view of the whole result: (uint8 block over the pinned host memory, total entries,
        byte offsets of a_id/z_id/parent/a_or/z_or, header int64 [n_trees, 3] = (component id, n_pvst, first))."""
        blk, nb, tot = C.c_void_p(), C.c_size_t(0), C.c_uint64(0)
        offs = (C.c_uint64 * 5)()
        if self._lib.povu_hip_forest_raw(self._h, C.byref(blk), C.byref(nb), C.byref(tot), offs) != 0:
            raise RuntimeError("forest has no raw block")
        n = len(self)
        hdr = np.zeros((n, 3), dtype=np.int64)
        t = _Tree()
        for i in range(n):
            self._lib.povu_hip_forest_get(self._h, i, C.byref(t))
            hdr[i] = (t.component_id, t.n_pvst, self._lib.povu_hip_forest_first(self._h, i))
        if nb.value == 0 or not blk.value:
            block = np.zeros(0, dtype=np.uint8)
        else:
            block = np.ctypeslib.as_array(C.cast(blk, C.POINTER(C.c_uint8)), shape=(nb.value,))
        return block, int(tot.value), [int(x) for x in offs], hdr

    def pack(self) -> np.ndarray:
        """The forest in its wire format (host bytes): what a transport other than RCCL ships."""
        n = self._lib.povu_hip_forest_pack_size(self._h)
        buf = np.zeros(n, dtype=np.uint8)
        if self._lib.povu_hip_forest_pack(self._h, buf.ctypes.data, n) != 0:
            raise RuntimeError("forest pack failed")
        return buf

    def share(self, rank: int) -> np.ndarray:
        """Descriptor (8 x uint64, word 7 = `rank`) of this forest's block in shared memory for the root of a multi-process
        job (povu_hip_forest_share; the context must have been put into shared-results mode)."""
        d = (C.c_uint64 * 8)()
        rc = self._lib.povu_hip_forest_share(self._h, d)
        if rc != 0:
            raise RuntimeError({2: "the forest's block is no shared segment: call HipDecomposer.share_results first",
                                4: "a MERGED forest with hairpin boundaries / subflubble labels cannot be shared: share its parts"}.get(rc, f"forest share failed ({rc})"))
        out = np.array(list(d), dtype=np.uint64)
        out[7] = rank
        return out

    def component_ids(self) -> List[int]:
        t = _Tree()
        out = []
        for i in range(len(self)):
            self._lib.povu_hip_forest_get(self._h, i, C.byref(t))
            out.append(int(t.component_id))
        return out

    def text(self, i: int) -> str:
        ln = C.c_size_t(0)
        p = self._lib.povu_hip_forest_pvst_text(self._h, i, C.byref(ln))
        if not p:
            raise IndexError(i)
        s = C.string_at(p, ln.value).decode()
        self._lib.povu_hip_buffer_free(p)
        return s

    def texts(self) -> Dict[int, str]:
        """{component_id: pvst text} -- what `povu decompose` writes as <id>.pvst."""
        out = {}
        for i in range(len(self)):
            t = _Tree()
            self._lib.povu_hip_forest_get(self._h, i, C.byref(t))
            out[t.component_id] = self.text(i)
        return out


class Shards:
    """Device-resident partition of a resident graph into per-rank packed shards (povu_hip_shard_partition)."""

    def __init__(self, lib, handle, owner):
        self._lib, self._h, self._owner = lib, handle, owner

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.povu_hip_shards_free(self._h)
            self._h = None

    @property
    def world(self) -> int:
        return self._lib.povu_hip_shards_world(self._h)

    @property
    def total_components(self) -> int:
        return self._lib.povu_hip_shards_total_components(self._h)

    def info(self, rank: int) -> dict:
        i = _ShardInfo()
        if self._lib.povu_hip_shards_get(self._h, rank, C.byref(i)) != 0:
            raise IndexError(rank)
        return dict(n_vtx=i.n_vtx, n_links=i.n_links, n_components=i.n_components, weight=int(i.weight), bytes=int(i.bytes),
                    device_ptr=i.device_ptr)

    def times(self) -> dict:
        t = (C.c_double * 3)()
        self._lib.povu_hip_shards_times(self._h, t)
        return dict(label_ms=t[0], lpt_ms=t[1], partition_ms=t[2])

    def export(self, rank: int) -> np.ndarray:
        """Packed shard `rank` as host bytes."""
        buf = np.zeros(self.info(rank)["bytes"], dtype=np.uint8)
        if self._lib.povu_hip_shards_export(self._h, self._owner._ctx, rank, buf.ctypes.data) != 0:
            raise RuntimeError("shard export failed")
        return buf


class HipDecomposer:
    """One context = one GPU, one stream, one workspace arena."""

    def __init__(self, device: int = 0):
        self._lib = load_lib()
        if self._lib.povu_hip_device_count() <= 0:
            raise HipUnavailable("no HIP device visible: the decompose path has no CPU fallback")
        err = C.create_string_buffer(512)
        self._ctx = self._lib.povu_hip_create(device, err, 512)
        if not self._ctx:
            raise HipUnavailable(err.value.decode())
        self._keep = None

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.povu_hip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()

    def upload(self, links, tips=None):
        """links: povu_amd.workloads.Links (vertex idx based link arrays)."""
        vid = np.ascontiguousarray(links.vid, dtype=np.uint32)
        v1 = np.ascontiguousarray(links.v1, dtype=np.uint32)
        v2 = np.ascontiguousarray(links.v2, dtype=np.uint32)
        s1 = np.ascontiguousarray(links.s1, dtype=np.uint8)
        s2 = np.ascontiguousarray(links.s2, dtype=np.uint8)
        tp = None
        if tips is not None:
            tips = np.ascontiguousarray(tips, dtype=np.uint8)
            tp = tips.ctypes.data
        err = C.create_string_buffer(512)
        rc = self._lib.povu_hip_graph_upload(self._ctx, len(vid), vid.ctypes.data, len(v1), v1.ctypes.data,
                                             s1.ctypes.data, v2.ctypes.data, s2.ctypes.data, tp, err, 512)
        if rc != 0:
            raise RuntimeError(err.value.decode())

    def prewarm(self, n_vtx: int, n_links: int) -> None:
        """Reserve the device memory a graph of this size will need (povu_hip_prewarm; only on a context that holds nothing)."""
        err = C.create_string_buffer(512)
        if self._lib.povu_hip_prewarm(self._ctx, n_vtx, n_links, err, 512) != 0:
            raise RuntimeError(err.value.decode())

    def upload_times(self) -> dict:
        """Device time of the last upload (HIP events, ms): host-to-device copies, CSR build, reverse-slot table."""
        t = (C.c_double * 3)()
        if self._lib.povu_hip_last_upload_times(self._ctx, t) != 0:
            raise RuntimeError("no graph resident")
        return dict(h2d_ms=t[0], csr_ms=t[1], twin_ms=t[2])

    # ---- multi-GPU sharding
    def partition(self, world: int) -> Shards:
        """Labels the resident graph's components, bin-packs them over `world` ranks and partitions the links on the
        device (povu_hip_shard_partition)."""
        err = C.create_string_buffer(512)
        h = self._lib.povu_hip_shard_partition(self._ctx, world, err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return Shards(self._lib, h, self)

    def upload_shard(self, packed, nbytes: Optional[int] = None, on_device: bool = False):
        """Makes a packed shard the resident graph: host bytes (numpy uint8) or a device pointer (int) + size."""
        err = C.create_string_buffer(512)
        if on_device:
            ptr, n = int(packed), int(nbytes)
        else:
            packed = np.ascontiguousarray(packed, dtype=np.uint8)
            ptr, n = packed.ctypes.data, packed.size
        if self._lib.povu_hip_graph_upload_shard(self._ctx, ptr, n, 1 if on_device else 0, err, 512) != 0:
            raise RuntimeError(err.value.decode())

    def share_results(self, tag: str) -> None:
        """From now on the PVST blocks of this context's forests are shared-memory segments "/povu.<tag>.<k>" (the
        multi-process convention: tag = "<job>.<rank>"), see povu_hip_share_results."""
        err = C.create_string_buffer(512)
        if self._lib.povu_hip_share_results(self._ctx, tag.encode(), err, 512) != 0:
            raise RuntimeError(err.value.decode())

    def attach_forests(self, own: Optional["Forest"], own_rank: int, job_tag: str, descs) -> "Forest":
        """Root of a multi-process job: the merged forest of every rank's descriptor (Forest.share) and of its own forest,
        which it takes over.  No PVST array is copied: the other ranks' blocks are mapped where their GPUs put them."""
        d = np.ascontiguousarray(np.asarray(descs, dtype=np.uint64).reshape(-1, 8))
        err = C.create_string_buffer(512)
        h = self._lib.povu_hip_forest_attach(self._ctx, own._h if own is not None else None, own_rank, job_tag.encode(),
                                             d.ctypes.data, d.shape[0], err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return Forest(self._lib, h)

    def transfer_bytes(self) -> dict:
        """Bytes this context moved since it was created: over PCIe in either direction, to and from other GPUs."""
        b = (C.c_uint64 * 4)()
        self._lib.povu_hip_transfer_bytes(self._ctx, b)
        return dict(h2d=int(b[0]), d2h=int(b[1]), peer_out=int(b[2]), peer_in=int(b[3]))

    def shard_total_components(self) -> int:
        return int(self._lib.povu_hip_shard_total_components(self._ctx))

    def merge_forests(self, packed_list) -> Forest:
        """Merges packed forests (Forest.pack() bytes) into one forest ordered by component id."""
        bufs = [np.ascontiguousarray(b, dtype=np.uint8) for b in packed_list]
        n = len(bufs)
        ptrs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
        sizes = (C.c_size_t * max(n, 1))(*[b.size for b in bufs])
        err = C.create_string_buffer(512)
        h = self._lib.povu_hip_forest_merge(self._ctx, ptrs, sizes, n, err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return Forest(self._lib, h)

    def decompose_shard(self, flags: int = 0) -> Forest:
        """Decomposes the resident shard and rewrites the component ids to those of the whole graph; a rank that
        owns no component gets an empty forest."""
        if self.shard_total_components() == 0:
            raise RuntimeError("the resident graph is not a shard")
        err = C.create_string_buffer(512)
        o = _Opts(0, 1, flags)
        h = self._lib.povu_hip_decompose(self._ctx, C.byref(o), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        f = Forest(self._lib, h)
        if self._lib.povu_hip_forest_globalize(h, self._ctx) != 0:
            raise RuntimeError("forest globalize failed")
        return f

    def decompose(self, rank: int = 0, world: int = 1, flags: int = 0) -> Forest:
        o = _Opts(rank, world, flags)
        err = C.create_string_buffer(512)
        h = self._lib.povu_hip_decompose(self._ctx, C.byref(o), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return Forest(self._lib, h)

    def stage_times(self) -> List[dict]:
        buf = (_StageTime * 64)()
        n = self._lib.povu_hip_last_stage_times(self._ctx, buf, 64)
        return [dict(name=buf[i].name.decode(), ms=buf[i].ms, launches=buf[i].launches) for i in range(min(n, 64))]

    def seq_redo_count(self) -> int:
        return int(self._lib.povu_hip_last_seq_redo(self._ctx))

    def links_processed(self) -> int:
        return int(self._lib.povu_hip_last_links_processed(self._ctx))

    # ---- parity hooks
    def debug_scan(self, a, op: int = 0, b=None):
        """Unit-test hook: exclusive scan of `a` on the device (op 0 sum, 1 running max); with `b`, an
        independent sum scan of `b` in the same launch."""
        a = np.ascontiguousarray(a, dtype=np.uint32)
        out = np.empty_like(a)
        if b is None:
            rc = self._lib.povu_hip_debug_scan(self._ctx, op, a.ctypes.data, out.ctypes.data, a.size, None, None, 0)
            if rc:
                raise RuntimeError(f"debug_scan failed ({rc})")
            return out
        b = np.ascontiguousarray(b, dtype=np.uint32)
        out2 = np.empty_like(b)
        rc = self._lib.povu_hip_debug_scan(self._ctx, op, a.ctypes.data, out.ctypes.data, a.size, b.ctypes.data,
                                           out2.ctypes.data, b.size)
        if rc:
            raise RuntimeError(f"debug_scan failed ({rc})")
        return out, out2

    def debug_components(self, n_vtx: int):
        comp = np.zeros(n_vtx, dtype=np.uint32)
        loc = np.zeros(n_vtx, dtype=np.uint32)
        if self._lib.povu_hip_debug_components(self._ctx, comp.ctypes.data, loc.ctypes.data) != 0:
            raise RuntimeError("no decompose state")
        return comp, loc

    def debug_tree(self, comp: int):
        n = C.c_uint32(0)
        if self._lib.povu_hip_debug_tree(self._ctx, comp, C.byref(n), None, None, None, None) != 0:
            raise RuntimeError("no decompose state")
        gid = np.zeros(n.value, dtype=np.uint32)
        typ = np.zeros(n.value, dtype=np.uint8)
        par = np.zeros(n.value, dtype=np.uint32)
        cls = np.zeros(n.value, dtype=np.uint32)
        self._lib.povu_hip_debug_tree(self._ctx, comp, C.byref(n), gid.ctypes.data, typ.ctypes.data, par.ctypes.data,
                                      cls.ctypes.data)
        return dict(gid=gid, typ=typ & 3, black=(typ >> 2) & 1, par=par, cls=cls)

    def last_black_only_classes(self) -> bool:
        """True when the last pass numbered the cycle classes of the black tree edges only (the fast path)."""
        return bool(self._lib.povu_hip_last_black_only_classes(self._ctx))

    def last_crossings(self):
        """(flagged, crossed): candidate-stack entries whose interval is crossed, and those of them whose class was no longer
        open (povu_hip_last_crossings); (0, 0) when the laminarity check did not run."""
        o = (C.c_uint32 * 2)()
        if self._lib.povu_hip_last_crossings(self._ctx, o) != 0:
            raise RuntimeError("povu_hip_last_crossings failed")
        return int(o[0]), int(o[1])

    def last_laminar_check_ran(self) -> bool:
        """True when the last pass ran the laminarity check (the literal hi_2 rule deviated somewhere, or it was forced)."""
        return bool(self._lib.povu_hip_last_laminar_check_ran(self._ctx))

    def debug_edge_ids(self, comp: int):
        """Id of the tree edge into every tree vertex ([0] = 0xFFFFFFFF), Tree::add_tree_edge's shared counter."""
        n = C.c_uint32(0)
        rc = self._lib.povu_hip_debug_edge_ids(self._ctx, comp, C.byref(n), None)
        if rc != 0:
            raise RuntimeError("no parallel-tree state" if rc == 3 else "no decompose state")
        ids = np.zeros(n.value, dtype=np.uint32)
        if self._lib.povu_hip_debug_edge_ids(self._ctx, comp, C.byref(n), ids.ctypes.data) != 0:
            raise RuntimeError("povu_hip_debug_edge_ids failed")
        return ids

    def debug_stack(self, comp: int):
        n = C.c_uint32(0)
        if self._lib.povu_hip_debug_stack(self._ctx, comp, C.byref(n), None, None, None) != 0:
            raise RuntimeError("no decompose state")
        vtx = np.zeros(n.value, dtype=np.uint32)
        cls = np.zeros(n.value, dtype=np.uint32)
        ns = np.zeros(n.value, dtype=np.uint32)
        self._lib.povu_hip_debug_stack(self._ctx, comp, C.byref(n), vtx.ctypes.data, cls.ctypes.data, ns.ctypes.data)
        return dict(tree_vtx=vtx, cls=cls, next_seen=ns)


class MultiDecomposer:
    """One process, N GPUs (povu_hip_multi_*): one context and one host thread per device; the root device partitions, the
    shards travel over xGMI (RCCL), every GPU lands its PVST block in host memory over its own PCIe link."""

    def __init__(self, devices):
        self._lib = load_lib()
        if self._lib.povu_hip_device_count() <= 0:
            raise HipUnavailable("no HIP device visible: the decompose path has no CPU fallback")
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        err = C.create_string_buffer(512)
        self._h = self._lib.povu_hip_multi_create(devs, len(devices), err, 512)
        if not self._h:
            raise HipUnavailable(err.value.decode())
        self.world = len(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.povu_hip_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def transport(self) -> str:
        return self._lib.povu_hip_multi_transport(self._h).decode()

    def upload(self, links, tips=None):
        vid = np.ascontiguousarray(links.vid, dtype=np.uint32)
        v1 = np.ascontiguousarray(links.v1, dtype=np.uint32)
        v2 = np.ascontiguousarray(links.v2, dtype=np.uint32)
        s1 = np.ascontiguousarray(links.s1, dtype=np.uint8)
        s2 = np.ascontiguousarray(links.s2, dtype=np.uint8)
        tp = None
        if tips is not None:
            tips = np.ascontiguousarray(tips, dtype=np.uint8)
            tp = tips.ctypes.data
        err = C.create_string_buffer(512)
        if self._lib.povu_hip_multi_upload(self._h, len(vid), vid.ctypes.data, len(v1), v1.ctypes.data, s1.ctypes.data,
                                           v2.ctypes.data, s2.ctypes.data, tp, err, 512) != 0:
            raise RuntimeError(err.value.decode())

    def scatter(self, keep_graph: bool = True):
        err = C.create_string_buffer(512)
        if self._lib.povu_hip_multi_scatter(self._h, 1 if keep_graph else 0, err, 512) != 0:
            raise RuntimeError(err.value.decode())

    def decompose(self, flags: int = 0) -> Forest:
        err = C.create_string_buffer(512)
        h = self._lib.povu_hip_multi_decompose(self._h, flags, None, None, err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return Forest(self._lib, h)

    def rank_info(self, rank: int) -> dict:
        r = _MultiRank()
        if self._lib.povu_hip_multi_rank(self._h, rank, C.byref(r)) != 0:
            raise IndexError(rank)
        return {k: getattr(r, k) for k, _ in _MultiRank._fields_}

    def times(self) -> dict:
        t = (C.c_double * 6)()
        self._lib.povu_hip_multi_times(self._h, t)
        return dict(label_ms=t[0], lpt_ms=t[1], partition_ms=t[2], scatter_wall_ms=t[3], decompose_wall_ms=t[4], merge_ms=t[5])
