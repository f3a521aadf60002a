"""Summarise rocprofv3 output (kernel trace + PMC passes) of bench.py into profiles/<tag>_summary.md.

usage: python tools/summarize_profile.py <tag> <steps> <warmup> <kernel_trace_dir> [<pmc_steps> <pmc_warmup> <pmc_fetch_dir> <pmc_write_dir>]
The sweep kernel serves both bench workloads in two instantiations: k_robot_sweep<K, 0> (no inter-robot
edges: configs[1], one dispatch per 10-step tick; the warm-up and timed dispatches come first in the trace,
the whole-tick measurements after them) and k_robot_sweep<K, 2> (inter-robot messages staged: configs[2])."""
import collections
import csv
import glob
import os
import sys


def rows(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def main():
    tag, steps, warm, kdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    pmc_steps, pmc_warm = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 0)
    pmc = sys.argv[7:9]

    def split(rows_, value, steps, warm):
        n_w, n_t = -(-warm // 10), -(-steps // 10)
        prim = [value(r) for r in rows_ if ", 0>" in r["Kernel_Name"]][n_w:n_w + n_t]   # timed primary dispatches
        sec = [value(r) for r in rows_ if ", 2>" in r["Kernel_Name"]]
        sec = sec[len(sec) // 5:]                                                        # drop the secondary warm-up share
        return {"configs[1] dyn+obs, 10 iterations per dispatch": prim, "configs[2] +inter-robot, one iteration per dispatch": sec}

    out = [f"# rocprofv3 summary — {tag}", "", "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-dynamic ...`", ""]
    tr = [r for r in rows(kdir, "*kernel_trace.csv") if "k_robot_sweep" in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    out += [f"bench.py --steps {steps} --warmup {warm}", "", "| workload | dispatches | avg us | min us | max us | grid x block |", "|---|---|---|---|---|---|"]
    for name, d in split(tr, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, steps, warm).items():
        if d:
            out.append(f"| {name} | {len(d)} | {sum(d) / len(d):.2f} | {min(d):.2f} | {max(d):.2f} | {tr[0]['Grid_Size_X']} x {tr[0]['Workgroup_Size_X']} |")
    st = rows(kdir, "*kernel_stats.csv")
    if st:
        out += ["", "Kernel stats (all kernels of the run):", "", "| kernel | calls | total ns | avg ns | % |", "|---|---|---|---|---|"]
        for r in st[:6]:
            out.append(f"| {r['Name'][:60]} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {float(r['Percentage']):.2f} |")
    for d, name in zip(pmc, ("FETCH_SIZE", "WRITE_SIZE")):
        cr = [r for r in rows(d, "*counter_collection.csv") if "k_robot_sweep" in r["Kernel_Name"] and r["Counter_Name"] == name]
        cr.sort(key=lambda r: int(r["Start_Timestamp"]))
        out += ["", f"{name} per dispatch (KiB as reported by rocprofv3; own --pmc pass, bench.py --steps {pmc_steps} --warmup {pmc_warm}):", ""]
        for wl, v in split(cr, lambda r: float(r["Counter_Value"]), pmc_steps, pmc_warm).items():
            if v:
                out.append(f"* {wl}: mean {sum(v) / len(v):.1f} KiB over {len(v)} dispatches")
    os.makedirs("profiles", exist_ok=True)
    path = os.path.join("profiles", f"{tag}_summary.md")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
