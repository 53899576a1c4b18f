"""rocprofv3 --kernel-trace --stats summary (kernel_stats.csv) of a bench.py run -> a small JSON that bench.py can
quote next to its own isolated per-op timing: per kernel instance the calls and the average duration INSIDE the timed
loop (lanes overlapping, kernels queueing behind each other).  Carries the same stamp as traffic_report.py (hash of
the kernel sources), so bench.py withholds the figures when the sources have changed since.
  python tools/kernel_stats_report.py <kernel_stats.csv> <out.json> "<workload>" """
import csv, datetime, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha16():
    h = hashlib.sha256()
    base = os.path.join(ROOT, "iip_uavsal_saliency_amd", "csrc")
    for name in sorted(os.listdir(base)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(base, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "uavsal_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def clean(name):
    k = re.sub(r"\(anonymous namespace\)::", "", name)
    k = re.sub(r"\((uavsal_gemm::)?ConvK\)|\(DwK\)|\(.*\)$", "", k).replace("void ", "").strip()
    return k


out = {}
total = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    k = clean(r["Name"])
    out[k] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
              "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2),
              "percent": round(float(r["Percentage"]), 2)}
out["__stamp__"] = {"kernel_sources_sha16": sources_sha16(), "workload": sys.argv[3] if len(sys.argv) > 3 else "",
                    "collected": datetime.date.today().isoformat(),
                    "method": "rocprofv3 --kernel-trace --stats -- python3 bench.py ... (all launches of the process: warm-up, timed windows, per-op timing)"}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("percent", 0) if kv[0] != "__stamp__" else 1)[:14]:
    if k != "__stamp__":
        print("%-64s calls %5d avg %8.1f us  %5.1f %%" % (k[:64], v["calls"], v["avg_us"], v["percent"]))
