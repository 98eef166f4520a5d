"""Timeline of a rocprofv3 kernel trace of bench.py under the chain-stream executor: per kernel name count / mean / min / max duration,
and for the chain kernels the gap between consecutive launches; the wall time of a steady batch.  Usage: trace_timeline.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    by[name].append(r)
print("%-60s %6s %9s %9s %9s" % ("kernel", "n", "mean us", "min", "max"))
for name, rs in sorted(by.items(), key=lambda kv: -sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kv[1])):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rs]
    print("%-60s %6d %9.2f %9.2f %9.2f" % (name[:60], len(d), sum(d) / len(d), min(d), max(d)))
for key in ("chain_kernel", "hot_pairs_kernel", "train_hot_kernel"):
    rs = [r for r in rows if key in r["Kernel_Name"]]
    if len(rs) < 10:
        continue
    rs = rs[len(rs) // 2:]  # the timed region is the later half
    gaps = [(int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["End_Timestamp"])) / 1e3 for i in range(len(rs) - 1)]
    period = [(int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["Start_Timestamp"])) / 1e3 for i in range(len(rs) - 1)]
    gaps_sorted, period_sorted = sorted(gaps), sorted(period)
    print("%s: gap to the next launch median %.2f us (10%% %.2f, 90%% %.2f); start-to-start median %.2f us, mean %.2f" % (
        key, gaps_sorted[len(gaps) // 2], gaps_sorted[len(gaps) // 10], gaps_sorted[9 * len(gaps) // 10], period_sorted[len(period) // 2], sum(period) / len(period)))
    print("   a stretch of it (start us, duration us, queue):")
    t0 = int(rs[0]["Start_Timestamp"])
    for r in rs[:24]:
        print("     %9.2f %7.2f  q%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?")))
both = rows
if len(both) > 100:
    both = both[len(both) * 2 // 3:][:110]
    t0 = int(both[0]["Start_Timestamp"])
    print("merged timeline (start us, end us, kernel, queue):")
    for r in both:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
        print("   %9.2f %9.2f  %-28s q%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, name, r.get("Queue_Id", "?")))
