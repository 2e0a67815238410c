# tools/dev/walk_pmc_full.sh [kernel substr]: SQ counters + HBM traffic of the walk kernels at the product's configuration (the
# bench's 1 M x 10 kb record stream, one lane per 36.9 KB piece) -- tools/dev/walk_pmc.sh's 100 k entries fill a ninth of the chip
R=$PWD; O=$R/gpurun_out; K=${1:-walk_pieces}; cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --only-main --steps 1 --warmup 1 --no-walk-index"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $O/wpf$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set -d $O/wpf$i -f csv -- python3 $R/bench.py $Q > $O/wpf$i.json 2> $O/wpf$i.err
done
cd $R; python tools/sqsum.py gpurun_out/wpf1 gpurun_out/wpf2 gpurun_out/wpf3 gpurun_out/wpf4 --kernel $K
find $O/wpf? -name "*kernel_trace.csv" -delete
