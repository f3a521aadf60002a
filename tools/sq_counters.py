"""Aggregate rocprofv3 --pmc passes of bench.py per sweep-kernel instantiation: mean counter value per dispatch.

usage: python tools/sq_counters.py <pmc_dir> [<pmc_dir> ...]      (one directory per --pmc pass)
Prints a markdown table; the steady-state share of the dispatches is used (the first fifth is dropped)."""
import collections
import csv
import glob
import os
import sys


def main():
    acc = collections.defaultdict(list)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_robot_sweep" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            for r in rows:
                inst = r["Kernel_Name"].split("k_robot_sweep")[1].split(">")[0] + ">"
                acc[(inst, r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("| instantiation | counter | dispatches | mean per dispatch |")
    print("|---|---|---|---|")
    for (inst, name), v in sorted(acc.items()):
        v = v[len(v) // 5:]
        print(f"| k_robot_sweep{inst} | {name} | {len(v)} | {sum(v) / len(v):.4g} |")


if __name__ == "__main__":
    main()
