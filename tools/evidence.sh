#!/bin/bash
# usage (repo root, GPU box): bash tools/evidence.sh  -- the bench lines and profiles a round's evidence set is made of, ALL of them
# from the tree this script runs in (here, afterwards: python tools/evidence_collect.py <tag> copies the set into profiles/ with a
# manifest of the tree it was made from; tests/test_profiles.py fails when the kernels have changed since, or when a rocprof
# average disagrees with the bench line of the same set)
# bash tools/evidence.sh a | b | c: the set in three parts (a GPU call is 20 minutes at most): a = the bench lines, b = one rank of
# eight + the tools at scale, c = the profiles
O=gpurun_out
S=${1:-all}
mkdir -p $O
if [ $S = a ] || [ $S = all ]; then
rm -f $O/ev_*.json $O/ev_*.txt $O/ev_*.err
SECONDS=0; python bench.py --steps 20 --warmup 5 > $O/ev_bench.json 2> $O/ev_bench.err; echo "bench rc=$? in $SECONDS s"
python bench.py --workload dexta --steps 5 --warmup 1 > $O/ev_bench_dexta.json 2> $O/ev_bench_dexta.err; echo "dexta rc=$?"
python bench.py --workload dexar --steps 5 --warmup 1 > $O/ev_bench_dexar.json 2> $O/ev_bench_dexar.err; echo "dexar rc=$?"
python bench.py --pipeline --no-cpu-baseline --only-main --steps 20 --warmup 5 > $O/ev_bench_pipeline.json 2> $O/ev_bench_pipeline.err; echo "pipeline rc=$?"
fi
if [ $S = b ] || [ $S = all ]; then
# one rank of the 8-GPU job as the driver launches it (BASELINE configs[4]: 2.5 M entries, 125 GB, the scratch budget of a rank, the
# decode from the encoder's index on) -- alone on its GPU: what a scaling efficiency is to be read against (profiles/per_gpu_reference.json)
python bench.py --entries 2500000 --scratch-budget-gb 64 --no-cpu-baseline --only-main --steps 3 --warmup 1 > $O/ev_bench_rank_of_8.json 2> $O/ev_bench_rank_of_8.err; echo "rank of 8 rc=$?"
# the tools end to end on a 20 GB .quiva and its 4 GB .fasta, tmpfs to tmpfs, the reference's undexqv / undexta reading the files back
timeout -k 10 900 python tools/cli_scale.py 20 --ref > $O/ev_cli_scale.json 2> $O/ev_cli_scale.err; echo "cli scale rc=$?"
fi
if [ $S = c ] || [ $S = all ]; then
bash profiles/tools/profile_all.sh > $O/profile_all.log 2>&1; tail -2 $O/profile_all.log
bash profiles/tools/timeline.sh > $O/timeline.log 2>&1
./tools/microbench/copy_rate > $O/ev_copy_rate.txt 2>&1
python tools/microbench/hbm_rates.py > $O/ev_hbm_rates.txt 2>&1
fi
ls $O | head -40
