#!/bin/bash
# Step time of the default bench under HIP / ROCclr runtime switches (one box, through gpurun).  Result of round 4:
#   HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory): 11.73-11.76 -> 11.37-11.39 ms/step (-3.1 %): the package
#   sets it by default (w2v2_speaker_amd/__init__.py);  AMD_OPT_FLUSH=0 (system-scope fences): +3 %;
#   ROC_SYSTEM_SCOPE_SIGNAL=0, ROC_USE_FGS_KERNARG=0, DEBUG_HIP_KERNARG_COPY_OPT=0, DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1,
#   ROC_ACTIVE_WAIT_TIMEOUT=1000, GPU_MAX_HW_QUEUES=1, ROC_AQL_QUEUE_SIZE=65536: within noise (11.34-11.43).
run() { python bench.py --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do
  echo -n "package default (HIP_FORCE_DEV_KERNARG=1) "; run
  for kv in HIP_FORCE_DEV_KERNARG=0 ROC_SYSTEM_SCOPE_SIGNAL=0 AMD_OPT_FLUSH=0 ROC_USE_FGS_KERNARG=0 DEBUG_HIP_KERNARG_COPY_OPT=0 \
            DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 ROC_ACTIVE_WAIT_TIMEOUT=1000 GPU_MAX_HW_QUEUES=1 ROC_AQL_QUEUE_SIZE=65536; do
    echo -n "$kv "; env $kv bash -c "$(declare -f run); run"
  done
done
