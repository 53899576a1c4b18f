"""One timed step out of a `rocprofv3 --kernel-trace` of bench.py: per stream (queue) the busy time and the gaps between
consecutive kernels, the longest gaps, and the union of all kernels' busy intervals (device not idle).
  python tools/step_timeline.py <kernel_trace.csv> [step_index_from_the_end] [--list]      (--list: every launch of the step,
  start offset, duration and queue, in start order)"""
import csv, sys, collections, re

LIST = "--list" in sys.argv
if LIST:
    sys.argv.remove("--list")
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
# a step = from one stem_kernel to the next
starts = [i for i, r in enumerate(rows) if "stem_kernel" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-k - 1], starts[-k]
step = rows[a:b]
t0, t1 = step[0][0], rows[b][0]
print("step of %d launches, %.1f us from stem to stem" % (len(step), (t1 - t0) / 1e3))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*$", "", n).replace("void ", "")
    return n[:58]
byq = collections.defaultdict(list)
for s, e, n, q in step:
    byq[q].append((s, e, n))
if LIST:
    qid = {q: i for i, q in enumerate(sorted(byq, key=lambda q: -len(byq[q])))}
    for s, e, n, q in step:
        print("%8.1f us  +%7.1f us  lane %d %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, qid[q], "        " * qid[q], short(n)))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in ks)
    gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print("queue %s: %d kernels, busy %.1f us, positive gaps %d summing %.1f us (median %.2f us)" % (
        q, len(ks), busy / 1e3, len(pos), sum(g[0] for g in pos) / 1e3, sorted(g[0] for g in pos)[len(pos) // 2] / 1e3 if pos else 0))
    for g in sorted(pos, reverse=True)[:6]:
        print("     gap %7.1f us between %s -> %s" % (g[0] / 1e3, short(g[1]), short(g[2])))
# union of busy intervals
iv = sorted((s, e) for s, e, _, _ in step)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("device busy (union of all kernels) %.1f us of %.1f us: idle %.1f us" % (busy / 1e3, (t1 - t0) / 1e3, (t1 - t0 - busy) / 1e3))
