#!/usr/bin/env python3
"""Per-kernel hardware counters from three separate rocprofv3 PMC passes of the SAME bench command (rocpd sqlite):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dirF> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dirW> -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d <dirM> -- python3 bench.py ...
    python tools/pmc_counters.py <dirF> <dirW> <dirM> > profiles/rNN_pmc_counters.json

HBM traffic (MI355X_MICROARCH.md, HBM section; calibrated on adam_kernel whose traffic is known exactly): FETCH_SIZE
counts 64 B per 128-B request on gfx950 -> read bytes = 2 * FETCH_SIZE KiB; WRITE_SIZE (KiB) is exact.
Matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES counts the cycles a SIMD's matrix pipe is busy (16-17 per
v_mfma_f32_16x16x32: checked against the launches' MFMA counts), summed over the chip's 256 CUs x 4 SIMDs;
GRBM_GUI_ACTIVE is the dispatch's duration in shader clocks, reported as the SUM over the 8 XCD instances (one row per
dispatch in the rocpd database; GUI_ACTIVE / 8 / duration = 2.2 GHz, the clock under profiling).
mfma_busy = MFMA_BUSY / (1024 SIMDs * GUI_ACTIVE / 8).
gui_over_wall_ghz = GUI_ACTIVE / 8 / the dispatch's wall time in the same (profiled) pass.  NOT a clock reading at these
kernel lengths: the counter window includes a per-dispatch overhead (10-50 us kernels come out at 2.5-4.8 "GHz"); only
for the 300 us+ launches (weight gradients 2.28, layer-0 conv 2.27) does it approach the shader clock.  What the matrix
pipe sustains under the power cap is measured directly by tools/probes/mfma_sustained_probe."""
import glob
import json
import re
import sqlite3
import subprocess
import sys

N_SIMD = 4 * 256
N_XCD = 8


_DEMANGLED = {}


def demangle(name):
    """rocprofv3 leaves names with the _Float16 builtin (DF16_) mangled; binutils' c++filt does not know that code
    either, so it is swapped for the other half type (Dh; builtin types are not substitution candidates, the rest of
    the name is unaffected) and the printed `half` is renamed back."""
    if not name.startswith("_Z"):
        return name
    if name not in _DEMANGLED:
        try:
            txt = subprocess.run(["c++filt", name.replace("DF16_", "Dh")], capture_output=True, text=True,
                                 check=True).stdout.strip()
            _DEMANGLED[name] = re.sub(r"\bhalf\b", "_Float16", txt) if txt and not txt.startswith("_Z") else name
        except Exception:
            _DEMANGLED[name] = name
    return _DEMANGLED[name]


def per_kernel(d, counter, how="sum"):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    # one row per (dispatch, counter instance): fold the instances of a dispatch first, then the dispatches of a kernel
    q = ("select k.name, c.dispatch_id, sum(c.value), avg(c.value), count(*) from counters_collection c join kernels k "
         "on k.dispatch_id = c.dispatch_id where c.counter_name = ? group by k.name, c.dispatch_id")
    out = {}
    for name, _, s, a, _n in cur.execute(q, (counter,)):
        name = re.sub(r"\(.*", "", demangle(name)).replace("void ", "")
        e = out.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += s if how == "sum" else a
    return out


def durations(d):
    """kernel name -> summed wall time (ns) of its dispatches in this pass"""
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, ns in cur.execute("select name, sum(end - start) from kernels group by name"):
        name = re.sub(r"\(.*", "", demangle(name)).replace("void ", "")
        out[name] = out.get(name, 0) + ns
    return out


def main():
    dF, dW, dM = sys.argv[1:4]
    dV = sys.argv[4] if len(sys.argv) > 4 else None          # optional 4th pass: --pmc SQ_INSTS_VALU SQ_WAVES
    f, w = per_kernel(dF, "FETCH_SIZE"), per_kernel(dW, "WRITE_SIZE")
    mf = per_kernel(dM, "SQ_VALU_MFMA_BUSY_CYCLES")
    sb = per_kernel(dM, "SQ_BUSY_CYCLES")
    ga = per_kernel(dM, "GRBM_GUI_ACTIVE")
    dur = durations(dM)
    out = {"note": "rocprofv3 PMC passes of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` (B=66, fp16 mode), "
                   "one counter group per pass; read bytes = 2 * FETCH_SIZE KiB (gfx950 correction), write bytes = "
                   "WRITE_SIZE KiB; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCD instances).",
           "kernels": {}}
    # which kernel sources these counters were taken on: bench.py recomputes the hash and marks the counters stale
    # when the kernels have changed since (w2v2_speaker_amd/_build.py: source_hash)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from w2v2_speaker_amd._build import source_hash
    out["source_hash"] = source_hash()
    for name in sorted(f, key=lambda n: -(2 * f[n][1] + w.get(n, (0, 0))[1])):
        n, fk = f[name]
        wk = w.get(name, (n, 0.0))[1]
        rec = {"launches": n, "fetch_kib_raw": round(fk / n, 1), "write_kib": round(wk / n, 1),
               "hbm_bytes_per_launch": int((2 * fk + wk) / n * 1024)}
        if name in mf and name in ga and ga[name][1] > 0:
            rec["mfma_busy_cycles_per_launch"] = round(mf[name][1] / mf[name][0], 1)
            rec["gui_active_cycles_per_launch"] = round(ga[name][1] / ga[name][0] / N_XCD, 1)
            rec["mfma_busy"] = round(mf[name][1] / (N_SIMD * ga[name][1] / N_XCD), 4)
            if dur.get(name, 0) > 0:
                rec["gui_over_wall_ghz"] = round(ga[name][1] / N_XCD / dur[name], 3)
            if name in sb and sb[name][1] > 0:
                rec["sq_busy_cycles_per_launch"] = round(sb[name][1] / sb[name][0], 1)
        out["kernels"][name] = rec
    if dV is not None:
        # VALU wave-instructions per launch (SQ_INSTS_VALU, summed over the chip); for the attention kernels also per
        # attention score: a launch evaluates B * heads * T^2 scores (w2v2-base bench: 66 * 12 * 149^2), each wave
        # instruction covers 64 lanes -> instructions per score = 64 * SQ_INSTS_VALU / scores
        va = per_kernel(dV, "SQ_INSTS_VALU")
        scores = 66 * 12 * 149 * 149
        for name, (n, tot) in va.items():
            rec = out["kernels"].setdefault(name, {"launches": n})
            rec["valu_wave_insts_per_launch"] = round(tot / n, 1)
            if name.startswith("attn_"):
                rec["valu_lane_insts_per_score"] = round(64.0 * tot / n / scores, 2)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
