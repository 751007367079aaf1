"""Digest a rocprofv3 kernel_trace.csv: per distinct (kernel, grid) the launch geometry, LDS, registers, mean us."""
import csv, re, sys, collections
rows = collections.OrderedDict()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if len(sys.argv) > 2 and not re.search(sys.argv[2], name):
            continue
        mt = re.search(r"MT\d+x\d+x\d+", name)
        short = (name[:40] + " " + mt.group(0) + " " + " ".join(re.findall(r"(?:LDSB\d|MIWT\d+_\d+|PGR\d|PLR\d|SK\d|WG\d+_\d+_\d+|DTL\d|DTV[AB]\d|1LDSB\d|WGM\d+|SU\d+|GSU\d+|TLDS\d)", name))) if mt else name[:90]
        key = (short, r["Grid_Size_X"], r["Workgroup_Size_X"])
        d = rows.setdefault(key, {"n": 0, "ns": 0, "lds": r.get("LDS_Block_Size"), "vgpr": r.get("VGPR_Count"),
                                  "agpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count"), "scr": r.get("Scratch_Size")})
        d["n"] += 1
        d["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for (short, gx, wx), d in rows.items():
    print(f"{d['ns'] / d['n'] / 1e3:9.1f} us x{d['n']:3d} grid {gx:>8} wg {wx:>4} lds {d['lds']:>6} vgpr {d['vgpr']} agpr {d['agpr']} scratch {d['scr']}  {short}")
