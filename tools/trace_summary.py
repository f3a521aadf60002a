"""Diagnostics: from a `rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d DIR -- python3 tools/dynamic_tick_breakdown.py`
run, the medians of the last twenty ticks: duration of the neighbour search kernel and of the resident schedule launch, the host's
wait for the search, the tick period.  usage: python tools/trace_summary.py DIR"""
import csv, glob, sys
d = sys.argv[1]
k = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
a = list(csv.DictReader(open(glob.glob(d + "/*/*hip_api_trace.csv")[0])))
pr = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in k if "k_pairs_rows" in r["Kernel_Name"] or "k_grid_rows" in r["Kernel_Name"]][-20:]
rs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in k if "k_robot_sweep" in r["Kernel_Name"]][-20:]
sy = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in a if r["Function"] == "hipStreamSynchronize"][-20:]
med = lambda v: sorted(v)[len(v) // 2] / 1e3
print("search kernel duration (median of last 20): %.1f us; resident launch %.1f us; hipStreamSynchronize %.1f us" % (med([e - s for s, e in pr]), med([e - s for s, e in rs]), med([e - s for s, e in sy])))
if len(rs) > 2: print("tick period: %.1f us" % med([rs[i + 1][0] - rs[i][0] for i in range(len(rs) - 1)]))
