set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ap1 /tmp/ap2 /tmp/ap3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ap1 -- python3 $R/tools/attn_pmc.py > /tmp/ap1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d /tmp/ap2 -- python3 $R/tools/attn_pmc.py > /tmp/ap2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/ap3 -- python3 $R/tools/attn_pmc.py > /tmp/ap3.log 2>&1
python3 $R/tools/attn_pmc_report.py /tmp/ap1 /tmp/ap2 /tmp/ap3 > $R/gpurun_out/r06_attention_pmc_raw.txt 2>&1
tail -3 /tmp/ap1.log /tmp/ap3.log >> $R/gpurun_out/r06_attention_pmc_raw.txt
cd $R
VS_HIPBLASLT=1 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r06_vs_hipblaslt_raw.txt
