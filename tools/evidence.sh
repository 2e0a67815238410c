#!/bin/bash
# usage (repo root, GPU box): bash tools/evidence.sh  -- the bench lines and profiles a round's evidence set is made of
O=gpurun_out
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/ev_bench.json 2> $O/ev_bench.err; echo "bench rc=$?"
python bench.py --workload dexta --steps 5 --warmup 1 > $O/ev_bench_dexta.json 2> $O/ev_bench_dexta.err; echo "dexta rc=$?"
python bench.py --workload dexar --steps 5 --warmup 1 > $O/ev_bench_dexar.json 2> $O/ev_bench_dexar.err; echo "dexar rc=$?"
python bench.py --pipeline --no-cpu-baseline --only-main --steps 20 --warmup 5 > $O/ev_bench_pipeline.json 2> $O/ev_bench_pipeline.err; echo "pipeline rc=$?"
python bench.py --entries 2500000 --steps 3 --warmup 1 --only-main --no-cpu-baseline > $O/ev_bench_config4_slice.json 2> $O/ev_bench_config4_slice.err; echo "slice rc=$?"
python bench.py --entries 2500000 --steps 3 --warmup 1 --only-main --no-cpu-baseline --scratch-budget-gb 64 > $O/ev_bench_config4_slice_budget64.json 2> $O/ev_bench_config4_slice_budget64.err; echo "slice64 rc=$?"
bash profiles/tools/profile_all.sh > $O/profile_all.log 2>&1; tail -2 $O/profile_all.log
bash profiles/tools/timeline.sh > $O/timeline.log 2>&1
python tools/microbench/hbm_rates.py > $O/ev_hbm_rates.txt 2>&1
