"""Python face of the host runtime pieces the solver drives (include/gvs.h): partition, schedule, samplers."""
import ctypes as C

import numpy as np

from . import _lib


def partition(weights, num_partition):
    """-> (part int32[N], local uint32[N], sizes uint32[P]); SolverMixin::partition, solver.h:873-887."""
    w = np.ascontiguousarray(weights, np.float32)
    part = np.empty(w.size, np.int32)
    local = np.empty(w.size, np.uint32)
    sizes = np.zeros(num_partition, np.uint32)
    rc = _lib.lib().gvs_partition(w.ctypes.data, w.size, num_partition, part.ctypes.data, local.ctypes.data,
                                  sizes.ctypes.data)
    _lib.check(rc, "gvs_partition")
    return part, local, sizes


def schedule(num_partition, num_worker):
    """-> int32 [steps, workers, 2] of (head partition, tail partition); SolverMixin::get_schedule."""
    nw = 1 if num_partition == 1 else num_worker
    cap = max(2, (num_partition // max(nw, 1)) ** 2 * nw * nw * 2)
    out = np.zeros(cap, np.int32)
    steps = _lib.lib().gvs_schedule(num_partition, num_worker, out.ctypes.data, out.size)
    if steps < 0:
        _lib.check(steps, "gvs_schedule")
    return out[:steps * nw * 2].reshape(steps, nw, 2).copy()


def negative_weights(vertex_weights, ids, exponent):
    """powf(vertex_weights[ids], exponent) as the reference's host code computes it (solver.h:1263-1278)."""
    w = np.ascontiguousarray(vertex_weights, np.float32)
    ids = np.ascontiguousarray(ids, np.uint32)
    out = np.zeros(ids.size, np.float32)
    _lib.check(_lib.lib().gvs_negative_weights(w.ctypes.data, ids.ctypes.data, ids.size, float(exponent),
                                               out.ctypes.data), "gvs_negative_weights")
    return out


def host_uniforms(seed, stream, first, n):
    out = np.empty(n, np.float64)
    _lib.lib().gvs_host_uniforms(seed, stream, first, n, out.ctypes.data)
    return out


class Sampler(object):
    """The multi-threaded CPU positive sampler (edge / random walk / node2vec) of one worker."""

    MODES = {"edge": _lib.MODE_EDGE, "walk": _lib.MODE_WALK, "biased_walk": _lib.MODE_BIASED_WALK,
             "biased_reject": _lib.MODE_BIASED_REJECT}

    def __init__(self, graph, part, local, num_partition, seed):
        self._lib = _lib.lib()
        self.graph = graph  # keeps the native graph alive
        self.num_partition = num_partition
        part = np.ascontiguousarray(part, np.int32)
        local = np.ascontiguousarray(local, np.uint32)
        if part.size != graph.num_vertex or local.size != graph.num_vertex:
            raise ValueError("part / local must have one entry per vertex")
        self._handle = self._lib.gvs_sampler_create(graph._handle, part.ctypes.data, local.ctypes.data,
                                                    num_partition, seed)
        if not self._handle:
            raise ValueError("gvs_sampler_create failed: %s" % self._lib.gvk_last_error().decode())

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.gvs_sampler_destroy(h)

    def prepare(self, mode, p=1.0, q=1.0, num_thread=1):
        _lib.check(self._lib.gvs_sampler_prepare(self._handle, self.MODES[mode], p, q, num_thread),
                   "gvs_sampler_prepare")

    def prepare_column(self, tail_partition, num_thread=1):
        _lib.check(self._lib.gvs_sampler_prepare_column(self._handle, tail_partition, num_thread),
                   "gvs_sampler_prepare_column")

    def fill(self, pools, pool_size, mode, num_thread, sample_batch_size=4000, walk_length=40, walk_batch=100,
             augmentation_step=1, shuffle_base=1, tail_partition=-1, os_threads=0, cpu_offset=-1):
        """pools: dict {(hp, tp): uint32 array/tensor-backed buffer of >= pool_size*2 elements} or a P*P list."""
        P = self.num_partition
        ptrs = (C.c_void_p * (P * P))()
        for hp in range(P):
            for tp in range(P):
                buf = pools.get((hp, tp)) if isinstance(pools, dict) else pools[hp * P + tp]
                if buf is None:
                    continue
                ptr = buf.data_ptr() if hasattr(buf, "data_ptr") else buf.ctypes.data
                n = buf.numel() if hasattr(buf, "numel") else buf.size
                if n < pool_size * 2:
                    raise ValueError("pool (%d, %d) holds %d values, needs %d" % (hp, tp, n, pool_size * 2))
                ptrs[hp * P + tp] = ptr
        cfg = _lib.FillConfig(self.MODES[mode], num_thread, sample_batch_size, walk_length, walk_batch,
                              augmentation_step, shuffle_base, tail_partition, os_threads, cpu_offset)
        _lib.check(self._lib.gvs_sampler_fill(self._handle, ptrs, pool_size, C.byref(cfg)), "gvs_sampler_fill")

    def stream_position(self, thread):
        return self._lib.gvs_sampler_stream_position(self._handle, thread)

    def set_stream_position(self, thread, position):
        _lib.check(self._lib.gvs_sampler_set_stream_position(self._handle, thread, position))

    def _array(self, name, count, ctype, dtype):
        ptr = getattr(self._lib, "gvs_sampler_" + name)(self._handle)
        if not ptr or not count:
            return np.zeros(0, dtype)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,))

    @property
    def edge_prob(self):
        return self._array("edge_prob", self.graph.num_directed_edge, C.c_float, np.float32)

    @property
    def edge_alias(self):
        return self._array("edge_alias", self.graph.num_directed_edge, C.c_uint64, np.uint64)

    @property
    def edge_edge_offsets(self):
        return self._array("edge_edge_offsets", self.graph.num_directed_edge + 1, C.c_uint64, np.uint64)

    def column(self, tail_partition):
        """(edge_ids u64[n], prob f32[n], alias u64[n]) of the EDGE-mode column table of a tail partition."""
        count, ids, prob, alias = C.c_uint64(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self._lib.gvs_sampler_column(self._handle, tail_partition, C.byref(count), C.byref(ids),
                                                C.byref(prob), C.byref(alias)), "gvs_sampler_column")
        n = count.value
        return (np.ctypeslib.as_array(C.cast(ids, C.POINTER(C.c_uint64)), shape=(n,)),
                np.ctypeslib.as_array(C.cast(prob, C.POINTER(C.c_float)), shape=(n,)),
                np.ctypeslib.as_array(C.cast(alias, C.POINTER(C.c_uint64)), shape=(n,)))

    def neighbor_tables(self, count):
        return (self._array("neighbor_prob", count, C.c_float, np.float32),
                self._array("neighbor_alias", count, C.c_uint32, np.uint32))
