"""The solver ENGINE's host logic without a GPU.  The product has no CPU training path (constructing a GraphSolver without
a GPU raises — checked here with the product library); everything else in this file starts a process that loads the HOST
BUILD of the engine instead (tests/hostdev: the engine's own sources over a host stand-in for the HIP runtime, its kernels
being the CPU oracle) and runs one scenario of tests/host_scenarios.py in it: partitioning, sampling, pool handling,
batch-id / learning-rate accounting, the slot claim + all-gather exchange between several workers of one process and
between processes (gloo through the engine's transport hook), the routing of walk pools, write-back, the application
layer."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIBRARY = os.path.join(ROOT, "tests", "hostdev", "build", "libgvk_host.so")


@pytest.fixture(scope="session")
def host_build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "liboracle.so")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostdev")])
    return HOST_LIBRARY


def scenario(host_build, name, *arguments, timeout=1200):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["GVK_LIBRARY"] = host_build
    env["GVK_ALLOW_TEST_LIBRARY"] = "1"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_scenarios.py"), name, json.dumps(list(arguments))],
                         cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    print(run.stdout[-3000:])
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-6000:]
    return run.stdout


def test_no_cpu_training_path():
    import torch
    import graphvite_amd as gv
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="No GPU"):
        gv.solver.GraphSolver(128)
    with pytest.raises(AttributeError):
        gv.solver.GraphSolver(100)
    with pytest.raises(AttributeError):
        gv.solver.GraphSolver(128, float_type=gv.float64)


def test_build_defaults_follow_the_reference(host_build):
    scenario(host_build, "build_defaults")


def test_hub_row_rules(host_build):
    scenario(host_build, "hub_row_rules")


def test_work_lists_built_ahead_train_the_same_tables(host_build):
    scenario(host_build, "lists_prefetch")


def test_executor_simulator_forms(host_build):
    scenario(host_build, "executor_simulator")


def test_train_accounting_and_determinism(host_build):
    scenario(host_build, "accounting_and_determinism")


def test_grouped_pair_order_permutes_inside_batches_only(host_build):
    scenario(host_build, "grouped_pair_order")


@pytest.mark.parametrize("model,aug", [("LINE", 1), ("DeepWalk", 3), ("node2vec", 2)])
def test_models_and_samplers_run(host_build, model, aug):
    scenario(host_build, "models_and_samplers", model, aug)


def test_device_sampling_trains_pairs_of_the_block(host_build):
    scenario(host_build, "device_sampling")


def test_training_session_steps_equal_train(host_build):
    scenario(host_build, "session_equals_train")


def test_custom_schedule_and_optimizers(host_build):
    scenario(host_build, "custom_schedule_and_optimizers")


@pytest.mark.parametrize("workers,partitions,model,aug,device_sampling,order", [
    (2, 2, "LINE", 1, False, "sampled"), (2, 4, "LINE", 1, False, "grouped"), (4, 8, "LINE", 1, False, "grouped"),
    (2, 2, "DeepWalk", 2, False, "sampled"), (2, 4, "LINE", 1, True, "grouped"), (2, 4, "node2vec", 2, True, "sampled"),
    (2, 4, "LINE", 2, True, "sampled"), (2, 4, "LINE", 2, False, "sampled")])
def test_workers_of_one_process(host_build, workers, partitions, model, aug, device_sampling, order):
    """device_ids = [0] * W: slot claims, one in-place all-gather per step (the copies carrier), interleaved head groups,
    pinned context shards, routed walk pools — every pair trained in the block it belongs to, every batch id once."""
    scenario(host_build, "workers_in_one_process", workers, partitions, model, aug, device_sampling, order)


def test_walk_pairs_binned_per_block_keep_a_walk_s_pairs_apart(host_build):
    """The host restatement of gvk_sample_walks_blocks follows the rule the device kernel is pinned to (tests/test_kernel_gpu.py): per
    block and stripe the oracle's pairs, the pseudo shuffle's part chosen by the pair's index in its walk (DESIGN.md section 7.11 a)."""
    scenario(host_build, "walk_blocks_layout")


def test_partitions_travel_through_host_memory_when_the_model_does_not_fit(host_build):
    scenario(host_build, "streamed_partitions")


def test_auto_build_rules_match_the_reference_solver(host_build):
    scenario(host_build, "auto_build_rules")


def test_predict_and_link_prediction_pipeline(host_build, tmp_path):
    scenario(host_build, "link_prediction_pipeline", str(tmp_path))


def test_node_classification_cli_and_embedding_file(host_build, tmp_path):
    scenario(host_build, "node_classification_and_cli", str(tmp_path))


def test_word_graph_application_trains_a_corpus(host_build, tmp_path):
    scenario(host_build, "word_graph_application", str(tmp_path))


# ---- one process per worker over gloo (world_size 2 and 4) ----------------------------------------------------------------

@pytest.mark.parametrize("world,model,aug,partitions,order,device_sampling", [
    (2, "LINE", 1, 0, "sampled", False), (2, "DeepWalk", 2, 0, "sampled", False), (2, "node2vec", 2, 0, "sampled", False),
    (2, "LINE", 1, 4, "grouped", False), (2, "DeepWalk", 2, 4, "sampled", False), (4, "LINE", 1, 8, "grouped", False),
    (2, "LINE", 1, 4, "grouped", True), (2, "DeepWalk", 2, 2, "sampled", True), (2, "node2vec", 2, 4, "sampled", True),
    (2, "LINE", 2, 4, "sampled", True)])
def test_processes_over_gloo(host_build, tmp_path, world, model, aug, partitions, order, device_sampling):
    """gvx_solver_create_distributed with the collectives carried by gloo (the transport hook): after write-back every
    process holds the same, complete tables; batch ids interleave, every id exactly once; context shards pinned per
    process; walk pools sampled in slices by every process and routed by one all-to-all (CPU samplers or device-side
    sampling) — every trained pair is a (walk) pair of the block being trained."""
    scenario(host_build, "processes_over_gloo", str(tmp_path), world, model, aug, partitions, order, device_sampling)


def test_hub_rows_over_gloo(host_build, tmp_path):
    scenario(host_build, "hub_rows_over_gloo", str(tmp_path))


def test_two_processes_learn_what_the_reference_learns(host_build, tmp_path):
    scenario(host_build, "learning_quality_over_gloo", str(tmp_path), timeout=3000)
