#!/bin/bash
# Registers / spills / LDS of every kernel in one .hip file (device-only compile with the resource-usage remarks).
#   bash tools/kernel_resources.sh w2v2_speaker_amd/csrc/gemm_f32_dma.hip
src=$(realpath "$1")
root=$(dirname "$(dirname "$(realpath "$0")")")
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$root/include -I$root/w2v2_speaker_amd/csrc \
  --cuda-device-only -c "$src" -o /tmp/kres.co -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, sys
name, d = None, {}
for line in sys.stdin:
    if "error" in line: print(line.rstrip())
    m = re.search(r"remark: .*?:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if name: print(name, d)
        name, d = t.split(":", 1)[1].strip(), {}
    elif ":" in t:
        k, v = t.split(":", 1)
        if k.strip() in ("VGPRs", "AGPRs", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            d[k.strip().split(" ")[0] if k.strip() != "VGPRs Spill" else "spill"] = v.strip()
if name: print(name, d)
'
