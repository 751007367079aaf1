#!/bin/bash
# Compiler-inserted `s_waitcnt vmcnt(n)` (i.e. those NOT inside an inline-assembly block) of the kernels matching <regex> in one
# .hip file, with the loop / block they sit in -- a wait at the header of a K loop is executed in every iteration and also waits for
# every LDS-DMA piece in flight.   bash tools/loop_waits.sh gemm_ring.hip "IDF16_DF16_Lb0E"
f=$1; pat=${2:-.}
root=$(dirname "$(dirname "$(realpath "$0")")")
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$root/include -I$root/w2v2_speaker_amd/csrc --cuda-device-only -S $root/w2v2_speaker_amd/csrc/$f -o /tmp/lw.s 2>/dev/null
python3 - "$pat" <<'P'
import re,sys
pat=sys.argv[1]
lines=open('/tmp/lw.s').read().split('\n')
cur=None; inasm=False; out={}
for i,l in enumerate(lines):
    m=re.match(r'^(_Z\w+):',l)
    if m: cur=m.group(1); continue
    if 's_endpgm' in l: cur=None
    if cur is None or not re.search(pat,cur): continue
    if 'ASMSTART' in l: inasm=True
    if 'ASMEND' in l: inasm=False
    if 's_waitcnt' in l and 'vmcnt' in l and not inasm:
        # find whether in loop: look back for nearest label comment 'in Loop' / 'Loop Header'
        ctx=''
        for j in range(i,max(i-400,0),-1):
            if lines[j].startswith('.LBB') or 'Loop Header' in lines[j]:
                ctx=lines[j].strip()+' '+(lines[j+1].strip() if 'Loop' in lines[j+1] else ''); break
        out.setdefault(cur,[]).append((i,l.strip(),ctx[:110]))
for k,v in out.items():
    print(k)
    for x in v: print('   ',x)
P
