"""Per-launch HBM traffic of each kernel from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE),
with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte requests as 64 B for
wide coalesced streaming reads: doubled; units are KiB)."""
import collections, csv, json, re, sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\((uavsal_gemm::)?ConvK\)|\(DwK\)|\(.*\)$", "", k).replace("void ", "").strip()
        agg[k][0] += float(r["Counter_Value"])
        agg[k][1] += 1
    return agg


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, [0, 1])
    w, nw = write.get(k, [0, 1])
    out[k] = {"launches": nf, "fetch_mb_per_launch_corrected": round(2 * f * 1024 / nf / 1e6, 3),
              "write_mb_per_launch": round(w * 1024 / max(nw, 1) / 1e6, 3)}
    out[k]["hbm_mb_per_launch"] = round(out[k]["fetch_mb_per_launch_corrected"] + out[k]["write_mb_per_launch"], 3)
# stamp: what these counters were collected on (bench.py withholds the figure when the kernel sources changed)
import hashlib, os, datetime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
base = os.path.join(ROOT, "iip_uavsal_saliency_amd", "csrc")
for name in sorted(os.listdir(base)):
    if name.endswith((".hip", ".h")):
        h.update(name.encode())
        h.update(open(os.path.join(base, name), "rb").read())
h.update(open(os.path.join(ROOT, "include", "uavsal_hip.h"), "rb").read())
out["__stamp__"] = {"kernel_sources_sha16": h.hexdigest()[:16],
                    "workload": sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --no-cpu-baseline --no-extra (configs[1]: 360x640, 1 clip x 8 frames, f32)",
                    "collected": datetime.date.today().isoformat()}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in out.items():
    if k != "__stamp__" and ("conv_gemm" in k or "dw3x3" in k or "dwproj" in k or "fused_ir" in k):
        print("%-60s n=%4d fetch %9.2f MB write %9.2f MB" % (k[:60], v["launches"], v["fetch_mb_per_launch_corrected"], v["write_mb_per_launch"]))
