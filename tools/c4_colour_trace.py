#!/usr/bin/env python3
"""Reads the kernel trace (and counters) tools/c4_colour_trace.sh collected and prints the LAST complete Gauss-Seidel
trip launch by launch: a trip runs from one fused pass over A (spmv_panel_kernel<6,...> or whatever carries EPI_GS_FUSED)
to the next."""
import csv, glob, sys, collections
d = sys.argv[1]
kt = glob.glob(d + "/kt/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(kt))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
def short(nm):
    nm = nm.replace("void its::", "").replace("its::", "")
    return nm.split("(")[0][:44]
# the fused pass: the longest-running kernel name that repeats once per trip
fused = [i for i, nm in enumerate(names) if "spmv_panel_kernel<6" in nm or "panel_kernel<6" in nm]
if len(fused) < 3:
    fused = [i for i, nm in enumerate(names) if "reduce_finalize" in nm]
a, b = fused[-2], fused[-1]
pm = collections.defaultdict(dict)
for f in glob.glob(d + "/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        pm[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
# counters come from another run: match by position in the launch sequence (same program, same launches)
pm_seq = [pm[k] for k in sorted(pm)]
t0 = int(rows[a]["Start_Timestamp"])
print("# one trip = launches %d..%d of %d; times in us from the trip's first launch" % (a, b - 1, len(rows)))
print("%-44s %9s %9s %7s %8s %10s %12s %12s" % ("kernel", "start", "dur", "gap", "grid", "SQ_WAVES", "SQ_BUSY_CYC", "GRBM_ACTIVE"))
tot = 0.0; gaps = 0.0
for i in range(a, b):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nxt = int(rows[i + 1]["Start_Timestamp"])
    c = pm_seq[i] if len(pm_seq) == len(rows) else {}
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
    print("%-44s %9.1f %9.1f %7.1f %8d %10.0f %12.0f %12.0f" % (short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, (nxt - e) / 1e3, grid,
          c.get("SQ_WAVES", -1), c.get("SQ_BUSY_CYCLES", -1), c.get("GRBM_GUI_ACTIVE", -1)))
    tot += (e - s) / 1e3; gaps += (nxt - e) / 1e3
print("# kernel time %.1f us, gaps %.1f us, trip %.1f us" % (tot, gaps, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
