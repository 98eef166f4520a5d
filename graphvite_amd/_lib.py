"""Loader of the native library (libgvk.so = HIP kernels + C++ host runtime, C ABI in include/gvk.h, gvs.h).

The product path has no CPU fallback: if the library is missing this module raises, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GVK_LIBRARY is a TEST switch: tests/ and scripts/experiments/ load the A/B baselines (build/ab/libgvk_ab.so, make -C
# graphvite_amd/csrc ab) or the host build of the engine over the oracle's kernels (tests/hostdev) through it.  It is honoured
# only together with GVK_ALLOW_TEST_LIBRARY=1, which those set themselves; on its own it is an error, so that nothing can point
# the product at an oracle-backed build by accident.
PRODUCT_LIBRARY = os.path.join(_HERE, "libgvk.so")
LIB_PATH = PRODUCT_LIBRARY
if os.environ.get("GVK_LIBRARY"):
    if os.environ.get("GVK_ALLOW_TEST_LIBRARY") != "1":
        raise ImportError("GVK_LIBRARY=%s: another kernel library is a test switch (tests/, scripts/experiments/); it needs "
                          "GVK_ALLOW_TEST_LIBRARY=1 as well" % os.environ["GVK_LIBRARY"])
    LIB_PATH = os.environ["GVK_LIBRARY"]

GVK_OK, GVK_EINVAL, GVK_EDIM, GVK_EHIP, GVK_ENOMEM = 0, -1, -2, -3, -4
SGD, MOMENTUM, ADAGRAD, RMSPROP, ADAM = range(5)
TUNE_LANES_PER_PAIR = 1
TUNE_VARIANT = 2
TUNE_RUN_CAP = 3
TUNE_GENERATION = 4
TUNE_SEGMENT_STEPS = 5
TUNE_SKIP_LOSS = 6
TUNE_SPLIT_HITS = 7
TUNE_CHAIN_CAP = 8


class AliasEntry(C.Structure):
    _fields_ = [("prob", C.c_float), ("alias", C.c_uint32)]


class Optimizer(C.Structure):
    _fields_ = [("type", C.c_int32), ("lr", C.c_float), ("weight_decay", C.c_float), ("hp0", C.c_float),
                ("hp1", C.c_float), ("epsilon", C.c_float)]


class Tables(C.Structure):
    _fields_ = [("vertex", C.c_void_p), ("context", C.c_void_p), ("vertex_moment1", C.c_void_p),
                ("context_moment1", C.c_void_p), ("vertex_moment2", C.c_void_p), ("context_moment2", C.c_void_p),
                ("n_vertex", C.c_uint32), ("n_context", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class NegativeSource(C.Structure):
    _fields_ = [("negatives", C.c_void_p), ("table", C.c_void_p), ("count", C.c_uint32), ("seed", C.c_uint64),
                ("classes", C.c_void_p), ("class_count", C.c_uint32)]


class FillConfig(C.Structure):
    _fields_ = [("mode", C.c_int), ("num_thread", C.c_int), ("sample_batch_size", C.c_int), ("walk_length", C.c_int),
                ("walk_batch", C.c_int), ("augmentation_step", C.c_int), ("shuffle_base", C.c_int),
                ("tail_partition", C.c_int), ("os_threads", C.c_int), ("cpu_offset", C.c_int)]


MODE_EDGE, MODE_WALK, MODE_BIASED_WALK, MODE_BIASED_REJECT = 0, 1, 2, 3


class WalkGraph(C.Structure):
    _fields_ = [("flat_offsets", C.c_void_p), ("edges_uv", C.c_void_p), ("edge_table", C.c_void_p),
                ("neighbor_table", C.c_void_p), ("sorted_neighbors", C.c_void_p), ("local", C.c_void_p),
                ("num_vertex", C.c_uint32), ("num_edge_entries", C.c_uint32), ("biased", C.c_int32),
                ("p", C.c_float), ("q", C.c_float)]


# ---- include/gvx.h ---------------------------------------------------------------------------------------------------
GVX_AUTO = 0
GVX_DEVICE_SAMPLING, GVX_PAIR_ORDER, GVX_SEED, GVX_NEGATIVE_TABLE, GVX_NODE2VEC_TABLE_LIMIT, GVX_HUB_ROWS, GVX_HUB_PARTS, GVX_FIDELITY = 1, 2, 3, 4, 5, 6, 7, 8
GVX_HUB_LERP, GVX_HUB_CHAIN_CAP, GVX_HUB_ROUNDS, GVX_HUB_EXECUTOR, GVX_HUB_PAIR_LAUNCHES, GVX_HUB_GROUP = 9, 10, 11, 12, 13, 14
GVX_UNIQUE_ID_BYTES = 256
SCHEDULE_FUNCTION = C.CFUNCTYPE(C.c_float, C.c_int, C.c_int, C.c_void_p)
TRANSPORT_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
TRANSPORT_ALL_TO_ALL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class SolverOptimizer(C.Structure):  # gvx_optimizer
    _fields_ = [("type", C.c_int32), ("lr", C.c_float), ("weight_decay", C.c_float), ("hp0", C.c_float),
                ("hp1", C.c_float), ("epsilon", C.c_float), ("schedule", C.c_int32),
                ("schedule_function", SCHEDULE_FUNCTION), ("user", C.c_void_p)]


class TrainConfig(C.Structure):  # gvx_train_config
    _fields_ = [("model", C.c_char_p), ("num_epoch", C.c_int), ("resume", C.c_int), ("augmentation_step", C.c_int),
                ("random_walk_length", C.c_int), ("random_walk_batch_size", C.c_int), ("shuffle_base", C.c_int),
                ("p", C.c_float), ("q", C.c_float), ("positive_reuse", C.c_int), ("negative_sample_exponent", C.c_float),
                ("negative_weight", C.c_float), ("log_frequency", C.c_int)]


class SolverMembers(C.Structure):  # gvx_solver_members
    _fields_ = [("dim", C.c_int), ("num_partition", C.c_int), ("num_negative", C.c_int), ("num_epoch", C.c_int),
                ("resume", C.c_int), ("episode_size", C.c_int), ("batch_size", C.c_int), ("augmentation_step", C.c_int),
                ("random_walk_length", C.c_int), ("random_walk_batch_size", C.c_int), ("shuffle_base", C.c_int),
                ("positive_reuse", C.c_int), ("log_frequency", C.c_int), ("num_worker", C.c_int), ("num_sampler", C.c_int),
                ("negative_sample_exponent", C.c_float), ("negative_weight", C.c_float), ("p", C.c_float), ("q", C.c_float),
                ("gpu_memory_limit", C.c_size_t), ("gpu_memory_cost", C.c_size_t), ("model", C.c_char_p),
                ("optimizer", SolverOptimizer), ("batch_id", C.c_uint64), ("num_batch", C.c_uint64),
                ("train_seconds", C.c_double), ("rank", C.c_int), ("num_local_worker", C.c_int), ("pair_order", C.c_int),
                ("sampler_mode", C.c_int), ("device_sampling", C.c_int), ("partition_rows", C.c_uint32),
                ("transport", C.c_char_p), ("hub_rows", C.c_uint32), ("hub_parts", C.c_int32), ("hub_lerp", C.c_int32), ("hub_rounds", C.c_int32),
                ("lists_prefetched", C.c_uint32)]


class Transport(C.Structure):  # gvx_transport
    _fields_ = [("all_gather", TRANSPORT_ALL_GATHER), ("all_to_all", TRANSPORT_ALL_TO_ALL), ("user", C.c_void_p)]


class NativeLibraryError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library. torch is imported first so that libgvk.so binds to the HIP runtime torch uses
    (streams and device pointers cross the boundary, so there must be exactly one runtime in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C graphvite_amd/csrc`. graphvite_amd has no CPU fallback for the training path." % LIB_PATH)
    import torch  # noqa: F401  (loads libamdhip64 first)
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    if LIB_PATH == PRODUCT_LIBRARY and hasattr(l, "gvh_is_host_build"):
        raise NativeLibraryError("%s is a host build of the engine (oracle kernels): not a product library" % LIB_PATH)
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float
    P = C.POINTER
    l.gvk_train.restype = i32
    l.gvk_train.argtypes = [vp, i32, P(Optimizer), P(Tables), vp, P(NegativeSource), u32, vp, i32, i32, f32]
    l.gvk_train_episode.restype = i32
    l.gvk_train_episode.argtypes = [vp, i32, P(Optimizer), i32, P(Tables), vp, P(NegativeSource), u32, u32, u32,
                                    i32, vp, i32, i32, f32]
    l.gvk_hot_plan.restype = i32
    l.gvk_hot_plan.argtypes = [i32, i32, i32, u32, u32, i32, i32, i32, P(C.c_size_t)]
    l.gvk_hot_build.restype = i32
    l.gvk_hot_build.argtypes = [vp, i32, vp, C.c_size_t, vp, i32, i32, i32, P(NegativeSource), u32, u32, u32, u32, i32, i32]
    l.gvk_hot_build_sliced.restype = i32
    l.gvk_hot_build_sliced.argtypes = [vp, i32, vp, C.c_size_t, vp, i32, i32, i32, P(NegativeSource), u32, u32, u32, u32, i32, i32, i32]
    l.gvk_train_episode_hot.restype = i32
    l.gvk_train_episode_hot.argtypes = [vp, i32, P(Optimizer), i32, P(Tables), vp, P(NegativeSource), u32, u32, u32, i32, vp,
                                        i32, i32, f32, vp, C.c_size_t, u32, u32, i32, i32, i32, i32]
    l.gvk_ahead_plan.restype = i32
    l.gvk_ahead_plan.argtypes = [i32, i32, i32, u32, u32, i32, i32, i32, P(C.c_size_t)]
    l.gvk_ahead_build.restype = i32
    l.gvk_ahead_build.argtypes = [vp, i32, vp, C.c_size_t, vp, i32, i32, i32, P(NegativeSource), u32, u32, u32, u32, i32, i32, i32]
    l.gvk_train_episode_ahead.restype = i32
    l.gvk_train_episode_ahead.argtypes = [vp, vp, i32, P(Optimizer), i32, P(Tables), vp, P(NegativeSource), u32, u32, u32, i32, vp,
                                          i32, i32, f32, vp, C.c_size_t, u32, u32, i32, i32, i32, i32, i32, i32]
    l.gvk_predict.restype = i32
    l.gvk_predict.argtypes = [vp, i32, vp, vp, vp, vp, i32]
    l.gvk_probe_row_traffic.restype = i32
    l.gvk_probe_row_traffic.argtypes = [vp, i32, vp, vp, vp, vp, C.c_float, i32]
    l.gvk_alias_sample.restype = i32
    l.gvk_alias_sample.argtypes = [vp, vp, u32, vp, vp, i32]
    l.gvk_negative_draw.restype = i32
    l.gvk_negative_draw.argtypes = [vp, vp, u32, u64, u32, vp, i32, i32]
    l.gvk_class_table_build.restype = i32
    l.gvk_class_table_build.argtypes = [vp, C.c_size_t, vp, vp]
    l.gvk_negative_draw_classes.restype = i32
    l.gvk_negative_draw_classes.argtypes = [vp, vp, u32, u64, u32, vp, i32, i32]
    l.gvk_sample_pairs.restype = i32
    l.gvk_sample_pairs.argtypes = [vp, vp, vp, u32, u64, u64, vp, C.c_size_t]
    l.gvk_sample_edges.restype = i32
    l.gvk_sample_edges.argtypes = [vp, vp, u32, u64, u64, vp, C.c_size_t]
    l.gvk_sample_walks.restype = i32
    l.gvk_sample_walks.argtypes = [vp, P(WalkGraph), u64, u64, vp, C.c_size_t, i32, i32, i32]
    l.gvk_sample_walks_blocks.restype = i32
    l.gvk_sample_walks_blocks.argtypes = [vp, P(WalkGraph), vp, i32, u64, u64, u64, vp, vp, vp, u32, i32, i32, i32, i32]
    l.gvk_sample_walks_blocks_thinned.restype = i32
    l.gvk_sample_walks_blocks_thinned.argtypes = [vp, P(WalkGraph), vp, i32, u64, u64, u64, vp, vp, vp, u32, i32, i32, i32, i32, vp]
    l.gvk_spread_pairs.restype = i32
    l.gvk_spread_pairs.argtypes = [vp, vp, vp, C.c_size_t, i32]
    l.gvk_group_pairs.restype = i32
    l.gvk_group_pairs.argtypes = [vp, vp, vp, vp, P(C.c_size_t), i32, i32, i32]
    l.gvk_alias_build.restype = i32
    l.gvk_alias_build.argtypes = [vp, C.c_size_t, vp, vp, i32, vp]
    l.gvk_set_tuning.restype = i32
    l.gvk_set_tuning.argtypes = [i32, i32]
    l.gvk_train_launches.restype = i32
    l.gvk_train_launches.argtypes = [i32, u32]
    l.gvk_has_ab_builds.restype = i32
    l.gvk_has_ab_builds.argtypes = []
    l.gvk_describe_train.restype = i32
    l.gvk_describe_train.argtypes = [i32, i32, i32, i32, i32, u32, C.c_char_p, C.c_size_t]
    l.gvk_range_push.restype = None
    l.gvk_range_push.argtypes = [C.c_char_p]
    l.gvk_range_pop.restype = None
    l.gvk_last_error.restype = C.c_char_p
    l.gvk_version.restype = C.c_char_p
    # ---- host runtime (include/gvs.h) ----
    sz, i64, f = C.c_size_t, C.c_int64, C.c_float
    cp = C.c_char_p
    l.gvs_graph_create.restype = vp
    l.gvs_graph_destroy.argtypes = [vp]
    l.gvs_graph_destroy.restype = None
    l.gvs_graph_load_file.restype = i32
    l.gvs_graph_load_file.argtypes = [vp, cp, i32, i32, cp, cp]
    l.gvs_graph_load_corpus.restype = i32
    l.gvs_graph_load_corpus.argtypes = [vp, cp, i32, i32, i32, cp, cp]
    l.gvs_graph_load_names.restype = i32
    l.gvs_graph_load_names.argtypes = [vp, P(cp), P(cp), vp, sz, i32, i32]
    l.gvs_graph_load_labels.restype = i32
    l.gvs_graph_load_labels.argtypes = [vp, vp, vp, vp, sz, i32, i32]
    l.gvs_graph_save.restype = i32
    l.gvs_graph_save.argtypes = [vp, cp, i32, i32]
    l.gvs_negative_weights.restype = i32
    l.gvs_negative_weights.argtypes = [vp, vp, u64, f32, vp]
    l.gvs_graph_neighbor_tables.restype = i32
    l.gvs_graph_neighbor_tables.argtypes = [vp, i32, vp]
    l.gvs_graph_num_vertex.restype = u32
    l.gvs_graph_num_vertex.argtypes = [vp]
    l.gvs_graph_num_edge.restype = u64
    l.gvs_graph_num_edge.argtypes = [vp]
    l.gvs_graph_num_directed_edge.restype = u64
    l.gvs_graph_num_directed_edge.argtypes = [vp]
    l.gvs_graph_as_undirected.restype = i32
    l.gvs_graph_as_undirected.argtypes = [vp]
    l.gvs_graph_normalization.restype = i32
    l.gvs_graph_normalization.argtypes = [vp]
    l.gvs_graph_name2id.restype = i64
    l.gvs_graph_name2id.argtypes = [vp, cp]
    l.gvs_graph_id2name.restype = i64
    l.gvs_graph_id2name.argtypes = [vp, u32, vp, sz]
    for name in ("edges", "edge_weights", "flat_offsets", "vertex_weights"):
        fn = getattr(l, "gvs_graph_" + name)
        fn.restype = vp
        fn.argtypes = [vp]
    l.gvs_partition.restype = i32
    l.gvs_partition.argtypes = [vp, u32, i32, vp, vp, vp]
    l.gvs_schedule.restype = i32
    l.gvs_schedule.argtypes = [i32, i32, vp, sz]
    l.gvs_sampler_create.restype = vp
    l.gvs_sampler_create.argtypes = [vp, vp, vp, i32, u64]
    l.gvs_sampler_destroy.restype = None
    l.gvs_sampler_destroy.argtypes = [vp]
    l.gvs_sampler_prepare.restype = i32
    l.gvs_sampler_prepare.argtypes = [vp, i32, f, f, i32]
    l.gvs_sampler_prepare_column.restype = i32
    l.gvs_sampler_prepare_column.argtypes = [vp, i32, i32]
    l.gvs_sampler_fill.restype = i32
    l.gvs_sampler_fill.argtypes = [vp, P(vp), u64, P(FillConfig)]
    l.gvs_sampler_stream_position.restype = u64
    l.gvs_sampler_stream_position.argtypes = [vp, i32]
    l.gvs_sampler_set_stream_position.restype = i32
    l.gvs_sampler_set_stream_position.argtypes = [vp, i32, u64]
    for name in ("edge_prob", "edge_alias", "neighbor_prob", "neighbor_alias", "edge_edge_offsets"):
        fn = getattr(l, "gvs_sampler_" + name)
        fn.restype = vp
        fn.argtypes = [vp]
    l.gvs_sampler_column.restype = i32
    l.gvs_sampler_column.argtypes = [vp, i32, P(u64), P(vp), P(vp), P(vp)]
    l.gvs_host_uniforms.restype = None
    l.gvs_host_uniforms.argtypes = [u64, u32, u64, sz, vp]
    # the solver engine (include/gvx.h)
    l.gvx_solver_create.restype = vp
    l.gvx_solver_create.argtypes = [i32, P(i32), i32, i32, sz]
    l.gvx_solver_create_distributed.restype = vp
    l.gvx_solver_create_distributed.argtypes = [i32, i32, i32, i32, vp, sz, P(Transport), i32, sz]
    l.gvx_unique_id.restype = i32
    l.gvx_unique_id.argtypes = [vp, sz]
    l.gvx_solver_destroy.restype = None
    l.gvx_solver_destroy.argtypes = [vp]
    l.gvx_solver_set.restype = i32
    l.gvx_solver_set.argtypes = [vp, i32, C.c_int64]
    l.gvx_solver_build.restype = i32
    l.gvx_solver_build.argtypes = [vp, vp, P(SolverOptimizer), i32, i32, i32, i32]
    l.gvx_solver_train.restype = i32
    l.gvx_solver_train.argtypes = [vp, P(TrainConfig)]
    l.gvx_solver_predict.restype = i32
    l.gvx_solver_predict.argtypes = [vp, vp, sz, vp]
    l.gvx_solver_clear.restype = i32
    l.gvx_solver_clear.argtypes = [vp]
    l.gvx_solver_embeddings.restype = vp
    l.gvx_solver_embeddings.argtypes = [vp, i32, P(u64)]
    l.gvx_solver_get.restype = i32
    l.gvx_solver_get.argtypes = [vp, P(SolverMembers)]
    l.gvx_solver_info.restype = sz
    l.gvx_solver_info.argtypes = [vp, C.c_char_p, sz]
    l.gvx_solver_save_embeddings.restype = i32
    l.gvx_solver_save_embeddings.argtypes = [vp, C.c_char_p]
    l.gvx_rccl_selftest.restype = i32
    l.gvx_rccl_selftest.argtypes = [i32]
    l.gvx_session_open.restype = i32
    l.gvx_session_open.argtypes = [vp, P(TrainConfig), i32]
    l.gvx_session_steps.restype = i32
    l.gvx_session_steps.argtypes = [vp]
    l.gvx_session_block.restype = i32
    l.gvx_session_block.argtypes = [vp, i32, i32, P(i32), P(i32)]
    l.gvx_session_fill.restype = i32
    l.gvx_session_fill.argtypes = [vp, i32]
    l.gvx_session_stage.restype = i32
    l.gvx_session_stage.argtypes = [vp, i32, i32, i32]
    l.gvx_session_train.restype = i32
    l.gvx_session_train.argtypes = [vp, i32, i32, i32, i32, i32]
    l.gvx_session_exchange.restype = i32
    l.gvx_session_exchange.argtypes = [vp, i32]
    for name in ("wait", "synchronize", "close"):
        fn = getattr(l, "gvx_session_" + name)
        fn.restype = i32
        fn.argtypes = [vp]
    l.gvx_session_stream.restype = vp
    l.gvx_session_stream.argtypes = [vp, i32]
    l.gvx_session_loss.restype = i32
    l.gvx_session_loss.argtypes = [vp, i32, P(f32)]
    l.gvx_session_probe.restype = i32
    l.gvx_session_probe.argtypes = [vp, i32, i32, i32, i32, P(f32)]
    l.gvx_session_exchange_stats.restype = i32
    l.gvx_session_exchange_stats.argtypes = [vp, P(u64), P(u64)]
    _lib = l
    return l


class profiler_range(object):
    """with profiler_range("Train Batch"): ... — a roctx range around a host phase (gvk_range_push / _pop), visible in
    `rocprofv3 --marker-trace` next to the kernels it issued.  Costs two C calls; a no-op without a profiler."""

    def __init__(self, name):
        self.name = name.encode()

    def __enter__(self):
        lib().gvk_range_push(self.name)

    def __exit__(self, *exc):
        lib().gvk_range_pop()
        return False


def check(rc, what="gvk"):
    if rc == GVK_OK:
        return
    msg = lib().gvk_last_error().decode("utf-8", "replace")
    if rc in (GVK_EINVAL, GVK_EDIM):
        raise ValueError("%s failed (%d): %s" % (what, rc, msg))
    if rc == GVK_ENOMEM:
        raise MemoryError("%s failed (%d): %s" % (what, rc, msg))
    raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))
