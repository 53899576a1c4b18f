import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
seen = collections.defaultdict(list)
for r in rows:
    if 'conv_gemm' not in r['Kernel_Name'] or r['Counter_Name'] != 'GRBM_GUI_ACTIVE':
        continue
    dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    seen[(r['Kernel_Name'][40:80], r['Grid_Size'])].append((dur, float(r['Counter_Value']) / 8 / dur))
for k, v in seen.items():
    v = v[3:]
    if v:
        print(k, 'n=%d avg dur %.1f us  clk %.3f GHz' % (len(v), sum(d for d, _ in v) / len(v) / 1e3, sum(c for _, c in v) / len(v)))
