#!/usr/bin/env python3
"""Does the grouped weight-gradient launch get slower per tile when every CU has one?  Times the plain 256x256-tile
launch at the encoder's token count with 27 x n tiles for n = 4..10 (the two-block group is n = 8: 216 tiles on 256
CUs); one tile = 308 K steps on every CU that has one, so a flat time up to 256 tiles means the spare CUs are free
capacity, a rising one means the chip-wide rate is capped."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops

dev, lp = "cuda", torch.float16
M = 66 * 149
Mp = (M + 63) // 64 * 64
H = 768


def mk(c):
    t = torch.zeros(Mp, c, dtype=lp, device=dev)
    t[:M] = (torch.randn(M, c, device=dev) * 0.5).to(lp)
    return t


for tiles_target in (108, 162, 216, 234, 252, 256, 270, 324):
    # problems of 768 x 768 (9 tiles) and one of 256*r x 256 to hit the target exactly
    probs, tiles = [], 0
    while tiles + 9 <= tiles_target:
        probs.append((mk(H), mk(H))); tiles += 9
    if tiles < tiles_target:
        r = tiles_target - tiles
        probs.append((mk(256 * r), mk(256))); tiles += r
    outs = [(torch.zeros(dy.shape[1], x.shape[1], device=dev), torch.zeros(dy.shape[1], device=dev)) for dy, x in probs]
    wg = ops.WgradGroup([(dy, x, dw, db) for (dy, x), (dw, db) in zip(probs, outs)], M, Mp)
    for _ in range(3):
        wg()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        wg()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"tiles={tiles:4d} problems={len(probs):3d}  {us:8.1f} us  {wg.flops / us / 1e6:8.1f} TFLOP/s  "
          f"{wg.flops / us / 1e6 / min(tiles, 256):6.2f} per busy CU")
