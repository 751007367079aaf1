#!/bin/bash
# First thing to run on an N-GPU node: weak-scaling efficiency of the training step over
# RCCL / xGMI for the knobs that were never measured on hardware --
#   W2V2_RESERVE_CUS   CUs kept out of the persistent GEMM grids (room for the RCCL channels' workgroups)
#   NCCL_MAX_NCHANNELS RCCL channel cap (each channel's workgroup displaces a 144 KiB-LDS GEMM workgroup)
# Usage: bash tools/scale_sweep.sh [max_gpus] [steps]      -> gpurun_out/scale_sweep.tsv + a printed table
MAXG=${1:-8}; STEPS=${2:-20}
OUT=${OUT:-gpurun_out}; mkdir -p $OUT
NG=$(python3 -c "import torch; print(torch.cuda.device_count())")
[ "$NG" -lt "$MAXG" ] && MAXG=$NG
echo -e "gpus\treserve_cus\tnchannels\tutt_per_s\tms_per_step\tefficiency" > $OUT/scale_sweep.tsv
BASE=""
for R in 0 8 16; do
  for CH in 8 16 32; do
    for N in 1 2 4 8; do
      [ "$N" -gt "$MAXG" ] && continue
      [ "$N" = 1 ] && [ "$CH" != 8 ] && continue          # the channel cap is irrelevant without a collective
      LINE=$(W2V2_RESERVE_CUS=$R NCCL_MAX_NCHANNELS=$CH python3 bench.py --gpus $N --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1)
      V=$(echo "$LINE" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>/dev/null) || continue
      U=${V% *}; MS=${V#* }
      [ "$N" = 1 ] && BASE1[$R]=$U
      EFF=$(python3 -c "print(round($U / ($N * ${BASE1[$R]:-$U}), 4))")
      echo -e "$N\t$R\t$CH\t$U\t$MS\t$EFF" | tee -a $OUT/scale_sweep.tsv
    done
  done
done
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/scale_sweep.tsv"), delimiter="\t"))
best = {}
for r in rows:
    n = int(r["gpus"])
    if n > 1 and (n not in best or float(r["efficiency"]) > float(best[n]["efficiency"])):
        best[n] = r
for n in sorted(best):
    b = best[n]
    print(f"best at {n} GPUs: W2V2_RESERVE_CUS={b['reserve_cus']} NCCL_MAX_NCHANNELS={b['nchannels']}: "
          f"{b['utt_per_s']} utt/s, efficiency {b['efficiency']}")
PY
