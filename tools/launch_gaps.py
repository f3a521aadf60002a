"""Diagnostics: from a `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/quick_ir_bench.py` run: duration of the
resident schedule launches and the gaps between consecutive ones (end of one dispatch to the start of the next), medians over the
timed part.  usage: python tools/launch_gaps.py DIR"""
import csv, glob, sys
d = sys.argv[1]
k = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
rs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in k if "k_robot_sweep" in r["Kernel_Name"])[-150:]
med = lambda v: sorted(v)[len(v) // 2] / 1e3
dur = [e - s for s, e in rs]
gap = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
per = [rs[i + 1][0] - rs[i][0] for i in range(len(rs) - 1)]
print("resident launch: duration %.2f us, gap to the next %.2f us (p10 %.2f, p90 %.2f), period %.2f us  [medians of the last %d dispatches]"
      % (med(dur), med(gap), sorted(gap)[len(gap) // 10] / 1e3, sorted(gap)[9 * len(gap) // 10] / 1e3, med(per), len(rs)))
