"""Per-op table out of `UAVSAL_BENCH_OPS=1 python bench.py ... 2> ops.err` (one line per launch of the plan: isolated
device time, algorithmic GB/s and TFLOP/s).  python tools/op_table.py ops.err [name-prefix ...] [--vs other.err]"""
import re
import sys


def load(f):
    rows = []
    for l in open(f):
        m = re.match(r"\[op\s+(\d+)\] (\S+)\s+(\S.*?)\s+([\d.]+) us\s+([\d.]+) GB/s\s+([\d.]+) TFLOP/s", l)
        if m:
            rows.append((int(m.group(1)), m.group(2), m.group(3), float(m.group(4)), float(m.group(5)), float(m.group(6))))
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    other = None
    if "--vs" in args:
        i = args.index("--vs")
        other = {r[1]: r for r in load(args[i + 1])}
        args = args[:i] + args[i + 2:]
    rows = load(args[0])
    pref = tuple(args[1:])
    sel = [r for r in rows if not pref or r[1].startswith(pref)]
    print("total %.1f us over %d ops (selected %.1f us over %d)" % (sum(r[3] for r in rows), len(rows), sum(r[3] for r in sel), len(sel)))
    for r in sel:
        line = "%3d %-26s %-24s %8.1f us %7.1f GB/s %7.2f TF" % r
        if other is not None and r[1] in other:
            line += "   | was %8.1f us" % other[r[1]][3]
        print(line)
