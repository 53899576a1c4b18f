# Same-box A/B of bench.py under environment knobs (how profiles/r5_experiments.md was measured; run through gpurun):
#   bash tools/ab.sh <outdir> "<extra bench args>" <label=ENV1=v ENV2=v ...> ...
# Every variant runs twice, alternating; prints the median window (ms per step) and the three windows of each run.
O=$1; shift
ARGS=$1; shift
mkdir -p $O
for rep in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  if [ "$envs" = "$spec" ]; then envs=""; fi
  env $envs python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --windows 3 $ARGS > $O/$label.$rep.json 2> $O/$label.$rep.err || echo "FAILED $label"
  python - <<PY
import json
d=json.load(open("$O/$label.$rep.json"))
print("$label", $rep, d["ms_per_step"], d["windows_ms_per_step"])
PY
done
done
