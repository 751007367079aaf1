cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in qkv ffn1 ffn2 out; do
  for pass in 1 2; do
    if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; else C="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; fi
    rm -rf /tmp/p_$s$pass
    rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$s$pass -- python3 $R/tools/gemm_pmc.py run $s > /dev/null 2>&1
    echo "== $s pass $pass"; python3 $R/tools/gemm_pmc.py sum /tmp/p_$s$pass
  done
done
