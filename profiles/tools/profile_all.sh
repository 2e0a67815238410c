#!/bin/bash
# usage (repo root, GPU box): bash profiles/tools/profile_all.sh ; then, here: python profiles/summarize.py <tag>
# Full evidence set: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel trace only)
# for the dexqv bench workload (with the on-device decode, so that the decode kernels are covered) and for the
# dexta / dexar workloads; SQ counters of the dexqv kernels on a 200 k-entry batch.
R=$PWD
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/prof_*
cd /tmp && export TMPDIR=/tmp
Q="--only-main --no-cpu-baseline --steps 3 --warmup 1"
timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof_stats -f csv -- python3 $R/bench.py --only-main --no-cpu-baseline --steps 5 --warmup 1 > $O/prof_stats.json 2> $O/prof_stats.err
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -f csv -- python3 $R/bench.py $Q > $O/prof_fetch.json 2> $O/prof_fetch.err
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -f csv -- python3 $R/bench.py $Q > $O/prof_write.json 2> $O/prof_write.err
for w in dexta dexar; do
  timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof_stats_$w -f csv -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 1 > $O/prof_stats_$w.json 2> $O/prof_stats_$w.err
  timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch_$w -f csv -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > $O/prof_fetch_$w.json 2> $O/prof_fetch_$w.err
  timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write_$w -f csv -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > $O/prof_write_$w.json 2> $O/prof_write_$w.err
done
# SQ counters (200 k entries, decode included)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 420 rocprofv3 --kernel-trace --pmc $set -d $O/prof_sq$i -f csv -- python3 $R/bench.py --only-main --no-cpu-baseline --entries 200000 --steps 2 --warmup 1 > $O/prof_sq$i.json 2> $O/prof_sq$i.err
done
# keep only the small summaries (the traces are large)
find $O -name "*kernel_trace.csv" -delete
du -sh $O/prof_* | tail -20
tail -c 300 $O/prof_stats.json
