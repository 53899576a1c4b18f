"""Sum the counters of a `rocprofv3 --pmc ... --output-format csv` run per kernel name.
  python tools/pmc_by_kernel.py <counter_collection.csv> [name-substring]"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*$", "", n).replace("void ", "")
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), n)
    if key not in seen:
        seen.add(key)
        calls[n] += 1
for n in sorted(acc):
    print("%-64s calls %4d  %s" % (n[:64], calls[n], "  ".join("%s %.4g" % (k, v / calls[n]) for k, v in sorted(acc[n].items()))))
