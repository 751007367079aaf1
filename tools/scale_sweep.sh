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
# Step 0 (self-check): two ranks through BOTH reducers -- C-ABI all-reduce bitwise against
# torch.distributed's, identical replicas, host counters broadcast -- and N distinct devices; nothing is swept on a node
# where this fails
if [ "$MAXG" -ge 2 ]; then
  python3 tools/ddp_selfcheck.py --gpus 2 | tail -1 | tee $OUT/ddp_selfcheck_2.json
  [ "${PIPESTATUS[0]}" = 0 ] || { echo "ddp_selfcheck failed: not sweeping"; exit 1; }
  if [ "$MAXG" -gt 2 ]; then
    python3 tools/ddp_selfcheck.py --gpus $MAXG | tail -1 | tee $OUT/ddp_selfcheck_$MAXG.json
    [ "${PIPESTATUS[0]}" = 0 ] || { echo "ddp_selfcheck at $MAXG ranks failed: not sweeping"; exit 1; }
  fi
fi
echo -e "gpus\treserve_cus\tnchannels\tutt_per_s\tms_per_step\tefficiency\tratio_vs_solo\trank_spread_ms" > $OUT/scale_sweep.tsv
BASE=""
for R in 0 8 16; do
  for CH in 8 16 32; do
    for N in 1 2 4 8; do
      [ "$N" -gt "$MAXG" ] && continue
      [ "$N" = 1 ] && [ "$CH" != 8 ] && continue          # the channel cap is irrelevant without a collective
      LINE=$(W2V2_RESERVE_CUS=$R NCCL_MAX_NCHANNELS=$CH python3 bench.py --gpus $N --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1)
      # a line whose rccl.ranks does not list N answering ranks (or a failed all-reduce check) stops the sweep
      V=$(echo "$LINE" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
n = d['n_gpus']
if n > 1:
    r = d.get('rccl', {}).get('ranks', [])
    assert len(r) == n and all(x['allreduce_ok'] for x in r), 'rccl.ranks incomplete'
dd = d.get('ddp') or {}
print(d['value'], d['ms_per_step'], dd.get('step_time_ratio_vs_solo', 1.0), dd.get('rank_spread_ms', 0.0))") || { echo "bad line at N=$N: $LINE"; exit 1; }
      set -- $V; U=$1; MS=$2; RS=$3; SP=$4
      [ "$N" = 1 ] && BASE1[$R]=$U
      EFF=$(python3 -c "print(round($U / ($N * ${BASE1[$R]:-$U}), 4))")
      echo -e "$N\t$R\t$CH\t$U\t$MS\t$EFF\t$RS\t$SP" | tee -a $OUT/scale_sweep.tsv
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
