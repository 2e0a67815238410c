# tools/dev/short_pmc.sh [kernel substr]: SQ counters + HBM traffic of the lane-per-entry kernels on the bench's short-entry shape (4 M x 300)
R=$PWD; O=$R/gpurun_out; K=${1:-k_qs_}; cd /tmp && export TMPDIR=/tmp
Q="--entries 4000000 --mean 300 --no-cpu-baseline --only-main --steps 1 --warmup 1 --no-walk-index --no-verify"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $O/spf$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set -d $O/spf$i -f csv -- python3 $R/bench.py $Q > $O/spf$i.json 2> $O/spf$i.err
done
cd $R; python tools/sqsum.py gpurun_out/spf1 gpurun_out/spf2 gpurun_out/spf3 gpurun_out/spf4 --kernel $K
find $O/spf? -name "*kernel_trace.csv" -delete
