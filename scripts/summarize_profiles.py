"""Copies the judged summaries of a scripts/gpu_job.sh run from gpurun_out/ (scratch) into profiles/<round>/.

    python scripts/summarize_profiles.py r2
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r3"
SRC, DST = "gpurun_out", os.path.join("profiles", ROUND)
os.makedirs(DST, exist_ok=True)


def find(directory, suffix):
    hits = sorted(glob.glob(os.path.join(SRC, directory, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None  # gpurun_out/ accumulates the runs of a round: the newest


for name in ("bench_n1.json", "bench_n1_steps20.json", "configs_2_4.jsonl", "configs_4.jsonl", "friendster_shard.json", "engine_e2e.jsonl", "access_pattern_probe.jsonl", "dim_sweep.jsonl", "shard_sweep.jsonl", "parity_auc.log", "c2_parity.log", "pytest_gpu_full.log", "smoke.log", "bench_n1_steps20_throughput.json", "bench_n1_steps20_with_traffic.json", "bench_by_partitions.jsonl", "pytest_gpu_refresh.log", "parity_auc_refresh.log"):
    if os.path.exists(os.path.join(SRC, name)) and os.path.getsize(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, name))

# ---- kernel trace -------------------------------------------------------------------------------------------------
# algorithmic bytes of one launch of the training kernel: what the bench line of the same job says (a batch trained as parts is
# several launches of train_hot_kernel)
PER_LAUNCH, LAUNCHES = 308800000, 1
for name in ("bench_n1_steps20.json", "bench_n1.json"):
    path = os.path.join(SRC, name)
    if os.path.exists(path) and os.path.getsize(path):
        line = [l for l in open(path) if l.startswith("{")]
        if line:
            roofline = json.loads(line[-1])["roofline"]
            PER_LAUNCH, LAUNCHES = roofline["algorithmic_bytes_per_launch"], roofline.get("launches_per_step", 1)
            break
stats, trace = find("prof_kernel", "kernel_stats.csv"), find("prof_kernel", "kernel_trace.csv")
if stats:
    shutil.copy(stats, os.path.join(DST, "kernel_stats_bench_n1.csv"))
if trace:
    every = list(csv.DictReader(open(trace)))
    rows = [r for r in every if "train_" in r["Kernel_Name"] and (LAUNCHES == 1 or int(r["Grid_Size_X"]) > 250000)]  # with parts: the launches that hold pairs (the first of a call holds chains only)
    probe = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in every if "probe_rows_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    gap = sorted(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1))
    by_name = collections.Counter(r["Kernel_Name"] for r in rows)
    summary = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                          "--no-end-to-end (the driver's command, end-to-end legs off)", "kernels": dict(by_name), "launches": len(rows), "mean_ns": sum(dur) / len(dur),
               "min_ns": min(dur), "max_ns": max(dur), "median_gap_ns": gap[len(gap) // 2], "grid": rows[0]["Grid_Size_X"],
               "workgroup": rows[0]["Workgroup_Size_X"], "algorithmic_bytes_per_launch": PER_LAUNCH, "launches_per_batch": LAUNCHES,
               "achieved_GBps": PER_LAUNCH / (sum(dur) / len(dur)), "fraction_of_hbm_peak": PER_LAUNCH / (sum(dur) / len(dur)) / 8000.0}
    if probe:  # roofline.access_pattern: the rows of a batch read and written back, nothing else
        summary["access_pattern_probe"] = {"kernel": "probe_rows_kernel", "launches": len(probe),
                                           "mean_ns": sum(probe) / len(probe),
                                           "fraction_of_hbm_peak": 308800000 / (sum(probe) / len(probe)) / 8000.0}
    json.dump(summary, open(os.path.join(DST, "kernel_trace_summary_bench_n1.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))

# ---- PMC passes, per dim ---------------------------------------------------------------------------------------------
by_dim = {"command": "rocprofv3 --pmc <counter> -- python bench.py --dim <d> --steps 20 --warmup 5 --no-cpu-baseline "
                     "--no-end-to-end (one pass per counter: FETCH_SIZE | WRITE_SIZE; dim 128 also TCC_HIT_sum TCC_MISS_sum)",
          "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE is doubled per MI355X_MICROARCH.md §HBM "
                   "(gfx950 tallies the 128-B requests of 16 B/lane loads at 64 B)"}


def counters(directory):
    path = find(directory, "counter_collection.csv")
    acc, kernel = collections.defaultdict(list), None
    if path:
        for r in csv.DictReader(open(path)):
            if "train_" in r["Kernel_Name"] and (LAUNCHES == 1 or int(r.get("Grid_Size", r.get("Grid_Size_X", 1 << 30))) > 250000):
                kernel = r["Kernel_Name"]
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in acc.items()}, kernel


for dim in (32, 64, 96, 128, 256, 512):
    fetch, kernel = counters("pmc_FETCH_SIZE_%d" % dim)
    write, _ = counters("pmc_WRITE_SIZE_%d" % dim)
    if "FETCH_SIZE" not in fetch or "WRITE_SIZE" not in write:
        continue
    f, w = fetch["FETCH_SIZE"]["mean"] * 1024, write["WRITE_SIZE"]["mean"] * 1024
    algorithmic = (8 * dim * 3 + 16) * 100000 // (LAUNCHES if dim == 128 else 1)
    entry = {"kernel": kernel, "FETCH_SIZE": fetch["FETCH_SIZE"], "WRITE_SIZE": write["WRITE_SIZE"],
             "hbm_read_bytes_per_launch_corrected": 2 * f, "hbm_write_bytes_per_launch": w,
             "traffic_bytes_per_launch": 2 * f + w, "algorithmic_bytes_per_launch": algorithmic,
             "launches_per_batch": LAUNCHES if dim == 128 else 1, "traffic_over_algorithmic": (2 * f + w) / algorithmic}
    if dim == 128:
        l2, _ = counters("pmc_L2_128")
        if "TCC_HIT_sum" in l2:
            entry.update(l2)
            entry["l2_hit_rate"] = l2["TCC_HIT_sum"]["mean"] / (l2["TCC_HIT_sum"]["mean"] + l2["TCC_MISS_sum"]["mean"])
        out = dict(by_dim, **entry)
        json.dump(out, open(os.path.join(DST, "pmc_summary_bench_n1.json"), "w"), indent=1)
    by_dim["dim_%d" % dim] = entry
    print("dim %d: traffic / algorithmic = %.3f" % (dim, entry["traffic_over_algorithmic"]))
# ---- traffic by role: the serialized form's launches are a unit's chains (long + short + idle-row blocks) or its pairs ----------------
def by_role(directory):
    path = find(directory, "counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if path:
        for r in csv.DictReader(open(path)):
            if "train_hot_kernel" in r["Kernel_Name"]:
                grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
                acc["pairs" if grid < 250000 else "chains"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {role: {k: sum(v) / len(v) for k, v in c.items()} for role, c in acc.items()}


roles = {}
for directory, names in (("pmc_FETCH_SIZE_roles", ("FETCH_SIZE",)), ("pmc_WRITE_SIZE_roles", ("WRITE_SIZE",)), ("pmc_L2_roles", ("TCC_HIT_sum", "TCC_MISS_sum"))):
    for role, values in by_role(directory).items():
        roles.setdefault(role, {}).update(values)
if roles:
    for role, v in roles.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["traffic_bytes_per_launch"] = 2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in v:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    roles["note"] = ("bench.py --tune 9=1 (GVK_TUNE_HOT_SERIALIZED): per unit one launch of its chains, then one of its pairs; means per launch; "
                     "FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE doubled in traffic_bytes_per_launch, MI355X_MICROARCH.md); the pairs' algorithmic "
                     "bytes per launch: %d" % PER_LAUNCH)
    json.dump(roles, open(os.path.join(DST, "pmc_summary_by_role.json"), "w"), indent=1)
    print("traffic by role:", json.dumps({k: v.get("traffic_bytes_per_launch") for k, v in roles.items() if isinstance(v, dict)}))

# ---- the shard size of an 8-GPU run: kernel trace -------------------------------------------------------------------------
trace8 = find("prof_kernel_p8", "kernel_trace.csv")
if trace8:
    rows8 = [r for r in csv.DictReader(open(trace8)) if "train_hot_kernel" in r["Kernel_Name"]]
    dur8 = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows8]
    if dur8:
        json.dump({"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --partitions 8 ...", "kernel": rows8[0]["Kernel_Name"],
                   "launches": len(dur8), "mean_ns": sum(dur8) / len(dur8), "min_ns": min(dur8), "max_ns": max(dur8)},
                  open(os.path.join(DST, "kernel_trace_summary_bench_p8.json"), "w"), indent=1)
    stats8 = find("prof_kernel_p8", "kernel_stats.csv")
    if stats8:
        shutil.copy(stats8, os.path.join(DST, "kernel_stats_bench_p8.csv"))

fetch, kernel = counters("pmc_FETCH_SIZE_shard96")
write, _ = counters("pmc_WRITE_SIZE_shard96")
if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:  # configs[4]'s kernel: dim 96 on one Friendster shard (8.2M rows)
    f, w = fetch["FETCH_SIZE"]["mean"] * 1024, write["WRITE_SIZE"]["mean"] * 1024
    algorithmic = (8 * 96 * 3 + 16) * 100000
    by_dim["dim_96_friendster_shard_8200000_rows"] = {
        "kernel": kernel, "FETCH_SIZE": fetch["FETCH_SIZE"], "WRITE_SIZE": write["WRITE_SIZE"],
        "traffic_bytes_per_launch": 2 * f + w, "algorithmic_bytes_per_launch": algorithmic,
        "traffic_over_algorithmic": (2 * f + w) / algorithmic}
    print("dim 96, Friendster shard: traffic / algorithmic = %.3f" % ((2 * f + w) / algorithmic))
if len(by_dim) > 2:
    json.dump(by_dim, open(os.path.join(DST, "pmc_summary_by_dim.json"), "w"), indent=1)

# ---- marker traces of end-to-end runs: roctx ranges next to the kernels ---------------------------------------------------
def marker_summary(directory, command, target):
    marker, kernels = find(directory, "marker_api_trace.csv"), find(directory, "kernel_trace.csv")
    if not (marker and kernels):
        return
    ranges = [r for r in csv.DictReader(open(marker))]
    kernel_rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kernels)))
    t0 = min([int(r["Start_Timestamp"]) for r in ranges] + [k[0] for k in kernel_rows])
    per_name = collections.defaultdict(lambda: {"count": 0, "total_ms": 0.0})
    timeline = []
    for r in sorted(ranges, key=lambda r: int(r["Start_Timestamp"])):
        name = r.get("Function") or r.get("Message") or r.get("Name") or "?"
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        per_name[name]["count"] += 1
        per_name[name]["total_ms"] += (b - a) / 1e6
        if len(timeline) < 80:
            timeline.append({"range": name, "thread": r.get("Thread_Id"), "start_ms": (a - t0) / 1e6, "end_ms": (b - t0) / 1e6})
    busy = collections.Counter()
    for a, b, name in kernel_rows:
        busy[name.split("(")[0][:60]] += (b - a) / 1e6
    out = {"command": command, "ranges": per_name, "kernel_busy_ms": dict(busy.most_common(8)),
           "first_kernel_ms": (kernel_rows[0][0] - t0) / 1e6, "last_kernel_ms": (kernel_rows[-1][1] - t0) / 1e6,
           "timeline_first_80_ranges": timeline}
    json.dump(out, open(os.path.join(DST, target), "w"), indent=1)
    print(target, "marker ranges:", {k: v["count"] for k, v in per_name.items()})


marker_summary("prof_marker", "rocprofv3 --marker-trace --kernel-trace -- python scripts/quick_start.py (configs[0] end to end)",
               "marker_trace_summary_e2e.json")
marker_summary("prof_marker_engine", "rocprofv3 --marker-trace --kernel-trace -- python scripts/measure_engine.py --quick "
               "(the native engine through the libgraphvite module: LINE on a Youtube-sized graph, CPU samplers then device "
               "sampling)", "marker_trace_summary_engine.json")
