"""graphvite_amd.graph — `Graph`, the drop-in for graphvite.graph.Graph (pyGraph, include/bind.h:109-187) over
the native graph store (include/gvs.h, graphvite_amd/csrc/gvs_host.cpp)."""
import ctypes as C
from collections.abc import Mapping, Sequence

import numpy as np

from . import _lib
from .base import dtype, io, logger


class _Name2Id(Mapping):
    def __init__(self, graph):
        self._g = graph

    def __getitem__(self, name):
        i = _lib.lib().gvs_graph_name2id(self._g._handle, str(name).encode())
        if i < 0:
            raise KeyError(name)
        return i

    def __contains__(self, name):
        return _lib.lib().gvs_graph_name2id(self._g._handle, str(name).encode()) >= 0

    def __iter__(self):
        return iter(self._g.id2name)

    def __len__(self):
        return self._g.num_vertex


class _Id2Name(Sequence):
    def __init__(self, graph):
        self._g = graph

    def __len__(self):
        return self._g.num_vertex

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        buf = C.create_string_buffer(64)
        n = _lib.lib().gvs_graph_id2name(self._g._handle, i, buf, len(buf))
        if n >= len(buf):
            buf = C.create_string_buffer(n + 1)
            _lib.lib().gvs_graph_id2name(self._g._handle, i, buf, len(buf))
        return buf.value.decode()


def _view(ptr, count, ctype, np_dtype):
    if not ptr or not count:
        return np.zeros(0, np_dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,))


class Graph(object):
    """
    Graph(index_type=dtype.uint32)
    Normal graphs without attributes.

    Parameters:
        index_type (dtype): type of node indexes (only uint32 is instantiated, as in the reference)
    """

    def __init__(self, index_type=dtype.uint32):
        if index_type != dtype.uint32:
            raise AttributeError("Can't find an instantiation of Graph with index_type=%s" % index_type)
        self.index_type = index_type
        self._lib = _lib.lib()
        self._handle = self._lib.gvs_graph_create()
        if not self._handle:
            raise MemoryError("gvs_graph_create failed")
        self.name2id = _Name2Id(self)
        self.id2name = _Id2Name(self)

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.gvs_graph_destroy(h)

    # ---- loading (3 overloads, include/bind.h:134-168) ----
    def load(self, *args, **kwargs):
        """
        load(file_name, as_undirected=True, normalization=False, delimiters=' \\t\\r\\n', comment='#')
        load(edge_list, as_undirected=True, normalization=False)
        load(weighted_edge_list, as_undirected=True, normalization=False)

        `edge_list` may also be an integer ndarray of shape (n, 2) (node names are then the decimal labels).
        """
        names = ["source", "as_undirected", "normalization", "delimiters", "comment"]
        params = dict(zip(names, args))
        for key in ("file_name", "edge_list", "weighted_edge_list"):
            if key in kwargs:
                if "source" in params:
                    raise TypeError("load() got multiple graph sources")
                params["source"] = kwargs.pop(key)
                params["_kind"] = key
        params.update(kwargs)
        if "source" not in params:
            raise TypeError("load() needs file_name, edge_list or weighted_edge_list")
        src = params["source"]
        und = bool(params.get("as_undirected", True))
        norm = bool(params.get("normalization", False))
        if isinstance(src, (str, bytes)):
            delimiters = params.get("delimiters", " \t\r\n")
            comment = params.get("comment", "#")
            logger.info("loading graph from %s", src)
            rc = self._lib.gvs_graph_load_file(self._handle, src.encode() if isinstance(src, str) else src, und, norm,
                                               delimiters.encode(), comment.encode())
            _lib.check(rc, "Graph.load")
        elif isinstance(src, np.ndarray):
            if src.ndim != 2 or src.shape[1] not in (2, 3):
                raise ValueError("edge array must have shape (n, 2) or (n, 3)")
            u = np.ascontiguousarray(src[:, 0], np.uint32)
            v = np.ascontiguousarray(src[:, 1], np.uint32)
            w = np.ascontiguousarray(src[:, 2], np.float32) if src.shape[1] == 3 else None
            rc = self._lib.gvs_graph_load_labels(self._handle, u.ctypes.data, v.ctypes.data,
                                                 None if w is None else w.ctypes.data, len(u), und, norm)
            _lib.check(rc, "Graph.load")
        else:
            edges = list(src)
            n = len(edges)
            weighted = n > 0 and len(edges[0]) == 3
            us = (C.c_char_p * n)(*[str(e[0]).encode() for e in edges])
            vs = (C.c_char_p * n)(*[str(e[1]).encode() for e in edges])
            w = np.ascontiguousarray([e[2] for e in edges], np.float32) if weighted else None
            rc = self._lib.gvs_graph_load_names(self._handle, us, vs, None if w is None else w.ctypes.data, n, und,
                                                norm)
            _lib.check(rc, "Graph.load")
        logger.warning(io.block(repr(self)))

    def save(self, file_name, weighted=True, anonymous=False):
        """Save the graph in edge-list format."""
        logger.info("Saving weighted graph to %s", file_name)
        _lib.check(self._lib.gvs_graph_save(self._handle, file_name.encode(), weighted, anonymous), "Graph.save")

    # ---- read-only attributes ----
    @property
    def num_vertex(self):
        return self._lib.gvs_graph_num_vertex(self._handle)

    @property
    def num_edge(self):
        return self._lib.gvs_graph_num_edge(self._handle)

    @property
    def as_undirected(self):
        return bool(self._lib.gvs_graph_as_undirected(self._handle))

    @property
    def normalization(self):
        return bool(self._lib.gvs_graph_normalization(self._handle))

    # ---- flattened views (borrowed from the native store; valid until the next load) ----
    @property
    def num_directed_edge(self):
        return self._lib.gvs_graph_num_directed_edge(self._handle)

    @property
    def edges(self):
        """uint32 [D, 2] directed edges {u, v} in vertex order (GraphMixin::flatten, core/graph.h:87-101)."""
        D = self.num_directed_edge
        return _view(self._lib.gvs_graph_edges(self._handle), 2 * D, C.c_uint32, np.uint32).reshape(D, 2)

    @property
    def edge_weights(self):
        return _view(self._lib.gvs_graph_edge_weights(self._handle), self.num_directed_edge, C.c_float, np.float32)

    @property
    def flat_offsets(self):
        return _view(self._lib.gvs_graph_flat_offsets(self._handle), self.num_vertex + 1, C.c_uint64, np.uint64)

    @property
    def vertex_weights(self):
        return _view(self._lib.gvs_graph_vertex_weights(self._handle), self.num_vertex, C.c_float, np.float32)

    def info(self):
        return "Graph<%s>\n%s\n#vertex: %d, #edge: %d\nas undirected: %s, normalization: %s" % (
            "uint32", io.header("Graph"), self.num_vertex, self.num_edge, io.yes_no(self.as_undirected),
            io.yes_no(self.normalization))

    __repr__ = info


class WordGraph(Graph):
    """
    WordGraph(index_type=dtype.uint32)
    Normal graphs of word co-occurrences (include/bind.h:189-234, include/instance/word_graph.cuh).

    Parameters:
        index_type (dtype): type of node indexes
    """

    def load(self, file_name, window=5, min_count=5, normalization=False, delimiters=" \t\r\n", comment="#"):
        """
        load(file_name, window=5, min_count=5, normalization=False, delimiters=' \\t\\r\\n', comment='#')
        Load a word graph from a corpus file (one sentence per line).

        Parameters:
            file_name (str): file name
            window (int, optional): word pairs with distance <= window are counted as edges
            min_count (int, optional): words with occurrence < min_count are discarded
            normalization (bool, optional): normalize the adjacency matrix or not
            delimiters (str, optional): string of delimiter characters
            comment (str, optional): prefix of comment strings
        """
        logger.info("generating graph from corpus %s", file_name)
        rc = self._lib.gvs_graph_load_corpus(self._handle, file_name.encode() if isinstance(file_name, str) else file_name,
                                             int(window), int(min_count), bool(normalization), delimiters.encode(),
                                             comment.encode())
        _lib.check(rc, "WordGraph.load")
        logger.warning(io.block(repr(self)))

    def info(self):
        return "WordGraph<%s>\n%s\n#vertex: %d, #edge: %d\nas undirected: %s, normalization: %s" % (
            "uint32", io.header("Graph"), self.num_vertex, self.num_edge, io.yes_no(self.as_undirected),
            io.yes_no(self.normalization))

    __repr__ = info
