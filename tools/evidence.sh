#!/bin/bash
# usage (repo root, GPU box): bash tools/evidence.sh  -- the bench lines and profiles a round's evidence set is made of, ALL of them
# from the tree this script runs in (here, afterwards: python tools/evidence_collect.py <tag> copies the set into profiles/ with a
# manifest of the tree it was made from; tests/test_profiles.py fails when the kernels have changed since, or when a rocprof
# average disagrees with the bench line of the same set)
O=gpurun_out
mkdir -p $O
rm -f $O/ev_*.json $O/ev_*.txt
python bench.py --steps 20 --warmup 5 > $O/ev_bench.json 2> $O/ev_bench.err; echo "bench rc=$?"
python bench.py --workload dexta --steps 5 --warmup 1 > $O/ev_bench_dexta.json 2> $O/ev_bench_dexta.err; echo "dexta rc=$?"
python bench.py --workload dexar --steps 5 --warmup 1 > $O/ev_bench_dexar.json 2> $O/ev_bench_dexar.err; echo "dexar rc=$?"
python bench.py --pipeline --no-cpu-baseline --only-main --steps 20 --warmup 5 > $O/ev_bench_pipeline.json 2> $O/ev_bench_pipeline.err; echo "pipeline rc=$?"
bash profiles/tools/profile_all.sh > $O/profile_all.log 2>&1; tail -2 $O/profile_all.log
bash profiles/tools/timeline.sh > $O/timeline.log 2>&1
./tools/microbench/copy_rate > $O/ev_copy_rate.txt 2>&1
python tools/microbench/hbm_rates.py > $O/ev_hbm_rates.txt 2>&1
ls $O | head -40
