"""Register budgets of the training kernels, read from the compiler (hipcc -Rpass-analysis=kernel-resource-usage; no GPU).
A training kernel that spills pays for it twice over in memory time — the moment optimizers at dim 256 / 512 ran at 0.35 of
the HBM peak for three rounds with 76–456 bytes of scratch per lane, at 0.75–0.81 without — so the budgets the
kernels are built for (`train_waves`, the hot kernel's four wavefronts per SIMD) are pinned here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "graphvite_amd", "csrc")


@pytest.fixture(scope="module")
def resources(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("isa")
    names, rows = [], []
    for source in ("gvk_pairs.hip", "gvk_chains.hip"):  # the training kernels' translation units
        run = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                              "-c", os.path.join(CSRC, source), "-o", str(tmp / (source + ".o")), "-Rpass-analysis=kernel-resource-usage"],
                             capture_output=True, text=True, cwd=str(tmp))
        assert run.returncode == 0, run.stderr[-3000:]
        for block in re.split(r"Function Name: ", run.stderr)[1:]:
            def field(key):
                m = re.search(re.escape(key) + r": (\d+)", block)
                return int(m.group(1)) if m else -1
            names.append(block.split()[0])
            rows.append(dict(vgpr=field("VGPRs"), scratch=field("ScratchSize [bytes/lane]"), occupancy=field("Occupancy [waves/SIMD]")))
    plain = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {p.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""): r for p, r in zip(plain, rows)}


def test_training_kernels_do_not_spill(resources):
    training = {k: r for k, r in resources.items() if k.startswith(("train_kernel<", "train_runs_kernel<", "train_hot_kernel<"))}
    assert len(training) > 100  # six dims x five optimizers x the builds of each
    spilling = {k: r["scratch"] for k, r in training.items() if r["scratch"] > 0}
    # what is left: 12 bytes in the RMSprop builds of the runs kernel at 16 floats per lane (three wavefronts per SIMD)
    assert set(spilling) <= {"train_runs_kernel<256, 16, 3, 0, -1, 3>", "train_runs_kernel<512, 32, 3, 0, -1, 3>"}, spilling
    assert all(v <= 16 for v in spilling.values()), spilling


def test_occupancy_the_kernels_are_built_for(resources):
    # the shipped per-pair kernel (dim 128, SGD, one negative drawn in the kernel): eight wavefronts per SIMD
    assert resources["train_kernel<128, 16, 0, 1, 1, 4>"]["occupancy"] == 8
    # hub rows by chains + the pairs of a unit in one launch: four (the short chains keep seven partner rows in flight)
    for hot in ("train_hot_kernel<128, 16, 1, 1, 0>", "train_hot_kernel<128, 16, 1, 1, 1>", "train_hot_kernel<128, 16, 1, 2, 0>", "train_hot_kernel<64, 16, 1, 1, 0>", "train_hot_kernel<32, 8, 1, 1, 1>"):
        assert resources[hot]["occupancy"] >= 4, (hot, resources[hot])
    # moment optimizers: waves per SIMD by the rows a lane group holds (train_waves)
    assert resources["train_kernel<256, 16, 4, 0, -1, 2>"]["occupancy"] == 2  # Adam, 16 floats per lane
    assert resources["train_kernel<256, 16, 1, 0, -1, 3>"]["occupancy"] == 3  # Momentum
    assert resources["train_kernel<128, 16, 1, 0, -1, 4>"]["occupancy"] >= 4
