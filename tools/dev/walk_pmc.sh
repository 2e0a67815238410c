R=$PWD; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  rm -rf $O/wpmc$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/wpmc$i -f csv -- python3 $R/tools/dev/walk_try.py ${1:-100000} 10000 > $O/wpmc$i.log 2>&1
  find $O/wpmc$i -name "*kernel_trace.csv" -delete
done
cd $R; python tools/sqsum.py gpurun_out/wpmc1 gpurun_out/wpmc2 --kernel walk_pieces
