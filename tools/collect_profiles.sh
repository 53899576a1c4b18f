# How the profiles/r2_* set was produced on the GPU box (run through gpurun from the repo root, then copy
# gpurun_out/r2final/{kt/kt_kernel_stats.csv, r2_hbm_traffic_f32_c1.json, traffic.txt, bench_*.json/.err} into profiles/).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r2final && mkdir -p $R/gpurun_out/r2final
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r2final/f -o f --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 5 --windows 1 > /dev/null 2> $R/gpurun_out/r2final/f.err
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r2final/w -o w --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 5 --windows 1 > /dev/null 2> $R/gpurun_out/r2final/w.err
cd $R
python3 tools/traffic_report.py gpurun_out/r2final/f/f_counter_collection.csv gpurun_out/r2final/w/w_counter_collection.csv profiles/r2_hbm_traffic_f32_c1.json > gpurun_out/r2final/traffic.txt
cp profiles/r2_hbm_traffic_f32_c1.json gpurun_out/r2final/
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2final/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra > $R/gpurun_out/r2final/kt_bench.json 2> $R/gpurun_out/r2final/kt.err
cd $R
python3 bench.py > gpurun_out/r2final/bench_default.json 2> gpurun_out/r2final/bench_default.err
python3 bench.py --prec f16x3 --no-extra --no-cpu-baseline > gpurun_out/r2final/bench_f16x3_c1.json 2> gpurun_out/r2final/bench_f16x3_c1.err
python3 bench.py --prec f16x3 --clips 8 --no-extra --no-cpu-baseline > gpurun_out/r2final/bench_f16x3_c8.json 2> gpurun_out/r2final/bench_f16x3_c8.err
python3 bench.py --clips 8 --no-extra --no-cpu-baseline > gpurun_out/r2final/bench_f32_c8.json 2> gpurun_out/r2final/bench_f32_c8.err
