# tools/dev/r6_entry_pmc.sh: what an ENTRY costs k_qv_encode_fast / k_qv_hist beside its symbols: SQ counters at three shapes of the same symbol count
R=$PWD; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
i=0
for shape in "--entries 100000 --mean 4096" "--entries 200000 --mean 2048" "--entries 400000 --mean 1024"; do
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
    i=$((i+1))
    rm -rf $O/epf$i
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $O/epf$i -f csv -- python3 $R/bench.py $shape --no-cpu-baseline --only-main --steps 1 --warmup 1 --no-walk-index --no-verify > $O/epf$i.json 2> $O/epf$i.err
    echo "== $shape" >> $O/epf_sum.txt
    (cd $R; python tools/sqsum.py gpurun_out/epf$i --kernel k_qv_encode_fast; python tools/sqsum.py gpurun_out/epf$i --kernel k_qv_hist) >> $O/epf_sum.txt
    rm -rf $O/epf$i
  done
done
cat $O/epf_sum.txt
