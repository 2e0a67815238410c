#!/bin/bash
# round 6, GPU session 1: dx_qv_scan tested and timed against the two calls; a timeline of the new step; counters nobody has looked at yet
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "scan or golden or config4 or onepass" > $O/s1_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/s1_tests.log
B="python bench.py --only-main --no-cpu-baseline --no-walk-index --no-verify --steps 10 --warmup 3"
run() { tag=$1; shift; env "$@" $B > $O/s1_$tag.json 2> $O/s1_$tag.err; python - $tag <<P
import json,sys
d=json.loads([l for l in open("gpurun_out/s1_%s.json"%sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], {k: round(v["ms_avg"],3) for k,v in d["kernels"].items()})
P
}
for rep in 1 2; do
  run new_$rep X=1
  run noguess_$rep DEXGPU_TEST=no_scan_guess
  run onegroup_$rep DEXGPU_TEST=onepass_groups=1
done
bash profiles/tools/timeline.sh --no-walk-index --no-verify > $O/s1_timeline.log 2>&1; cp $O/timeline.txt $O/s1_timeline.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/s1_counters_avail.txt 2>&1
cd $GRAFT_REPO_ROOT
Q="--entries 200000 --no-cpu-baseline --only-main --steps 1 --warmup 1 --no-walk-index --no-verify"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_WAIT_IFETCH SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf $O/s1pmc$i
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/$O/s1pmc$i -f csv -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $GRAFT_REPO_ROOT/$O/s1pmc$i.json 2> $GRAFT_REPO_ROOT/$O/s1pmc$i.err)
  python tools/sqsum.py $O/s1pmc$i --kernel k_qv_ > $O/s1pmc$i.txt 2>&1
  find $O/s1pmc$i -name "*kernel_trace.csv" -delete; find $O/s1pmc$i -name "*.csv" -size +2M -delete
done
cat $O/s1pmc?.txt | cut -c1-600
