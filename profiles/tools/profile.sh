#!/bin/bash
# usage (repo root, GPU box): bash profiles/tools/profile.sh ; then, here: python profiles/summarize.py <tag>
# full-size profile set for profiles/summarize.py: kernel stats + FETCH_SIZE + WRITE_SIZE passes
R=$PWD
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -f csv -- python3 $R/bench.py --only-main --no-cpu-baseline --steps 5 --warmup 1 > $R/gpurun_out/prof_stats.json 2> $R/gpurun_out/prof_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -f csv -- python3 $R/bench.py --only-main --no-cpu-baseline --no-verify --no-check --steps 2 --warmup 1 > $R/gpurun_out/prof_fetch.json 2> $R/gpurun_out/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -f csv -- python3 $R/bench.py --only-main --no-cpu-baseline --no-verify --no-check --steps 2 --warmup 1 > $R/gpurun_out/prof_write.json 2> $R/gpurun_out/prof_write.err
# keep only the small summaries (the traces are large)
find $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof_stats -name "*_stats.csv" | head
du -sh $R/gpurun_out/prof_*
tail -c 400 $R/gpurun_out/prof_stats.json
