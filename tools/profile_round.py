"""gpurun_out/<tag>_* (tools/profile_round.sh) -> profiles/<tag>_kernel_stats.csv, <tag>_configs_kernel_stats.csv,
<tag>_summary.md and traffic_<tag>.json (what bench.py's roofline block quotes as measured HBM traffic / VALU issue).

usage: python tools/profile_round.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def per_kernel(d, counter):
    """mean counter value per dispatch of every k_robot_sweep instantiation (steady state: first fifth dropped)"""
    acc = collections.defaultdict(list)
    rs = [r for r in rows(d, "*counter_collection.csv") if "k_robot_sweep" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rs.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rs:
        acc[r["Kernel_Name"].split("k_robot_sweep")[1].split(">")[0] + ">"].append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 5:]) / len(v[len(v) // 5:]) for k, v in acc.items()}


def pass_durations(d):
    """mean dispatch duration (us) per instantiation as the COUNTER pass itself saw it (first fifth dropped)"""
    acc = collections.defaultdict(dict)
    rs = [r for r in rows(d, "*counter_collection.csv") if "k_robot_sweep" in r["Kernel_Name"] and "Start_Timestamp" in r and "End_Timestamp" in r]
    for r in rs:
        acc[r["Kernel_Name"].split("k_robot_sweep")[1].split(">")[0] + ">"][r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    out = {}
    for k, dd in acc.items():
        v = [(e - s0) / 1e3 for s0, e in sorted(dd.values())]
        v = v[len(v) // 5:]
        out[k] = sum(v) / len(v)
    return out


def durations(d):
    acc = collections.defaultdict(list)
    tr = [r for r in rows(d, "*kernel_trace.csv") if "k_robot_sweep" in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in tr:
        acc[r["Kernel_Name"].split("k_robot_sweep")[1].split(">")[0] + ">"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # (the MEDIAN beside the mean: under the driver's command one dispatch of the resident kernel is a whole `sustained` block —
    # seconds of ticks posted into one lingering launch)
    return {k: (sum(v[len(v) // 5:]) / len(v[len(v) // 5:]), len(v), sorted(v)[len(v) // 2]) for k, v in acc.items()}


def kernel_digest():
    sys.path.insert(0, ROOT)
    import bench
    return bench.kernel_digest()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    g = os.path.join(ROOT, "gpurun_out", tag)
    # (on the GPU box: write beside the raw traces — tools/profile_round.sh then drops those, which are far beyond what comes
    # back from a gpurun call — and copy gpurun_out/<tag>_profiles/* into profiles/ afterwards)
    prof = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    out = [f"# rocprofv3 summary — {tag}", "",
           "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 20 --repeats 60 --preheat-ms 20 "
           "--no-cpu-baseline --no-extras --sustained-seconds 0` (tools/profile_round.sh; MI355X) — the driver's block shape, repeated.  Every tick "
           "is its own mgx_iterate call: `<16, 2, true, false>`: configs[2], a timed block is ONE resident dispatch of 20 iterations (the first tick "
           "launches it, the second is posted into the lingering launch, the closing synchronisation ends it); `<16, 0, false, false>`: configs[1] "
           "(ticks bracketed two at a time, 20 iterations per dispatch); `<16, 2, false, false>` (second run, `MGX_PERSISTENT=0`): configs[2], one "
           "iteration per dispatch (template arguments: horizon, inter-robot message mode, resident, sharded).", ""]
    for sub, name in (("_kt", "kernel_stats"), ("_kt_np", "kernel_stats_launch_per_iteration"), ("_cfg", "configs_kernel_stats")):
        f = glob.glob(os.path.join(g + sub, "**", "*kernel_stats.csv"), recursive=True)
        if f:
            shutil.copy(f[0], os.path.join(prof, f"{tag}_{name}.csv"))
            st = list(csv.DictReader(open(f[0])))
            out += [f"## {name} (`profiles/{tag}_{name}.csv`)", "", "| kernel | calls | total ns | avg ns | % |", "|---|---|---|---|---|"]
            for r in st[:8]:
                out.append(f"| {r['Name'][:70]} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {float(r['Percentage']):.2f} |")
            out.append("")
    dur = durations(g + "_kt")
    dur.update({k + " (MGX_PERSISTENT=0)": v for k, v in durations(g + "_kt_np").items()})
    dur.update({k + " (the driver's command: --gpus 1 --steps 20 --warmup 5)": v for k, v in durations(g + "_kt_driver").items()})
    out += ["## steady-state dispatch durations (kernel trace)", "", "| instantiation | dispatches | avg us | median us |", "|---|---|---|---|"]
    for k, (avg, n, med) in sorted(dur.items()):
        out.append(f"| k_robot_sweep{k} | {n} | {avg:.2f} | {med:.2f} |")
    traffic = {"_kernel_digest": kernel_digest(), "_source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes, tools/profile_round.sh) on `python3 bench.py "
                          f"--steps 20 --warmup 20 --repeats 60 --no-cpu-baseline --no-extras`, MI355X, round {tag}.  KiB per dispatch as reported; per "
                          "MI355X_MICROARCH.md (HBM section) gfx950 FETCH_SIZE counts half the bytes of wide 16-B-per-lane streaming reads, so "
                          "bench.py doubles the reads (an upper bound here: part of the staging uses 8-B loads)."}
    keys = {"<16, 0, false, false>": ("config1", ""), "<16, 2, true, false>": ("config2_resident", ""), "<16, 2, false, false>": ("config2", "_np")}
    out += ["", "## PMC per dispatch (separate passes)", "",
            "GRBM_GUI_ACTIVE is summed over the 8 XCDs and its window opens a few microseconds before a dispatch's start stamp and closes "
            "after its end stamp, so the clock THIS box ran at is the SLOPE between two instantiations of one pass, "
            "(cycles_A - cycles_B) / 8 / (duration_A - duration_B): see the line under the table (durations of other boxes, e.g. the "
            "driver's, relate to these counters through it).", "",
            "| instantiation | FETCH_SIZE KiB | WRITE_SIZE KiB | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES | SQ_WAIT_ANY / SQ_WAVE_CYCLES | kernel trace avg us | GRBM_GUI_ACTIVE / 8, k cycles | us in the counter passes (FETCH / WRITE / SQ / GRBM) |", "|---|---|---|---|---|---|---|---|---|"]
    clock_pts = []
    f64_rows = []
    # iterations per dispatch: what the profiled bench run itself reported (its JSON line)
    iters = {"config2": 1}
    try:
        ln = json.loads([x for x in open(g + "_kt.json").read().splitlines() if x.startswith("{")][-1])
        iters["config2_resident"] = ln["roofline"]["iterations_per_launch"]
        iters["config1"] = ln["configs1"]["roofline"]["iterations_per_launch"]
    except Exception as e:  # noqa: BLE001
        print("iterations per dispatch: not found in", g + "_kt.json", e)
    for inst, (key, suf) in keys.items():
        f = per_kernel(g + "_pmc_FETCH_SIZE" + suf, "FETCH_SIZE").get(inst)
        w = per_kernel(g + "_pmc_WRITE_SIZE" + suf, "WRITE_SIZE").get(inst)
        sq = {c: per_kernel(g + "_pmc_SQ_INSTS_VALU" + suf, c).get(inst) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY")}
        f64 = {c: per_kernel(g + "_pmc_SQ_INSTS_VALU_ADD_F64" + suf, c).get(inst)
               for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_THREAD_CYCLES_VALU",
                         "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU")}
        d_us = (dur.get(inst + (" (MGX_PERSISTENT=0)" if suf else "")) or (None, 0))[0]
        gui = per_kernel(g + "_pmc_GRBM_GUI_ACTIVE" + suf, "GRBM_GUI_ACTIVE").get(inst)
        pd = [pass_durations(g + "_pmc_" + nm + suf).get(inst) for nm in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")]
        pd_f64 = pass_durations(g + "_pmc_SQ_INSTS_VALU_ADD_F64" + suf).get(inst)
        clock = None
        if gui and pd[3]:
            clock_pts.append((suf, pd[3], gui / 8.0))  # (pass, us, cycles per XCD)
        if f is None or w is None:
            continue
        ent = {"fetch_kib": round(f, 1), "write_kib": round(w, 1), "source": f"profiles/{tag}_summary.md (rocprofv3 --pmc, separate passes)",
               "iterations_per_dispatch": iters.get(key)}
        busy = None
        if sq["SQ_INSTS_VALU"]:
            ent["valu_wave_instr"] = round(sq["SQ_INSTS_VALU"])
            if sq["SQ_BUSY_CYCLES"]:
                # SQ_ACTIVE_INST_VALU: quad-cycles summed over the SIMDs; SQ_BUSY_CYCLES: quad-cycles summed over the shader engines' SQs
                busy = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"] * 100.0
                ent["valu_busy_pct"] = round(busy, 1)
            if sq["SQ_WAIT_ANY"] and sq["SQ_WAVE_CYCLES"]:
                ent["wait_pct"] = round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"] * 100.0, 1)
        if f64.get("SQ_INSTS_VALU") and f64.get("SQ_INSTS_VALU_ADD_F64") is not None:
            n64 = sum(f64[c] or 0.0 for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
            flops_w = (f64["SQ_INSTS_VALU_ADD_F64"] or 0.0) + (f64["SQ_INSTS_VALU_MUL_F64"] or 0.0) + 2.0 * (f64["SQ_INSTS_VALU_FMA_F64"] or 0.0) + \
                      (f64["SQ_INSTS_VALU_TRANS_F64"] or 0.0)
            ent["f64_wave_instr"] = round(n64)
            ent["f64_by_kind"] = {k.replace("SQ_INSTS_VALU_", "").lower(): round(f64[k] or 0.0) for k in
                                  ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")}
            ent["f64_flops_per_wave_instr"] = round(flops_w / n64, 4) if n64 else 1.0  # (per lane: an FMA counts 2)
            ent["valu_wave_instr_f64_pass"] = round(f64["SQ_INSTS_VALU"])
            ent["salu_instr"] = round(f64["SQ_INSTS_SALU"] or 0.0)
            if f64.get("SQ_THREAD_CYCLES_VALU") and f64.get("SQ_ACTIVE_INST_VALU"):
                # rocprofiler's VALUUtilization: the share of a wave's 64 lanes active in the VALU instructions it issues
                ent["lane_occupancy"] = round(f64["SQ_THREAD_CYCLES_VALU"] / (f64["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
            f64_rows.append(f"| k_robot_sweep{inst} | {f64['SQ_INSTS_VALU']:.4g} | {n64:.4g} ({n64 / f64['SQ_INSTS_VALU'] * 100:.1f} %) | "
                            f"{f64['SQ_INSTS_VALU_ADD_F64'] or 0:.4g} / {f64['SQ_INSTS_VALU_MUL_F64'] or 0:.4g} / {f64['SQ_INSTS_VALU_FMA_F64'] or 0:.4g} / "
                            f"{f64['SQ_INSTS_VALU_TRANS_F64'] or 0:.4g} | {ent.get('lane_occupancy', 0) * 100:.1f} % | {f64['SQ_INSTS_SALU'] or 0:.4g} |")
        if d_us:
            ent["kernel_trace_avg_us"] = round(d_us, 2)
        if gui:
            ent["gui_active_cycles_per_xcd"] = round(gui / 8.0)
        ent["counter_pass_avg_us"] = [round(x, 2) if x else None for x in pd]
        traffic[key] = ent
        out.append(f"| k_robot_sweep{inst} | {f:.1f} | {w:.1f} | {sq['SQ_INSTS_VALU'] or 0:.4g} | VALU active {busy or 0:.1f} % of wave-cycles | "
                   f"{(sq['SQ_WAIT_ANY'] or 0) / (sq['SQ_WAVE_CYCLES'] or 1) * 100:.0f} % waiting | {d_us or 0:.2f} | {(gui or 0) / 8e3:.1f} | "
                   + " / ".join(f"{x:.1f}" if x else "-" for x in pd) + " |")
    if f64_rows:
        out += ["", "### what the VALU instructions are (pass `SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 SQ_THREAD_CYCLES_VALU ...`)", "",
                "SQ_INSTS_VALU counts EVERY VALU wave-instruction; the f64 arithmetic among them by kind; lane occupancy = SQ_THREAD_CYCLES_VALU / "
                "(SQ_ACTIVE_INST_VALU x 64), the share of a wave's 64 lanes its VALU instructions keep busy (rocprofiler's VALUUtilization).", "",
                "| instantiation | SQ_INSTS_VALU | of which f64 | add / mul / fma / trans | lane occupancy | SQ_INSTS_SALU |", "|---|---|---|---|---|---|"] + f64_rows
    same = [pt for pt in clock_pts if pt[0] == ""]
    if len(same) >= 2:
        a, b = max(same, key=lambda t: t[1]), min(same, key=lambda t: t[1])
        ghz = (a[2] - b[2]) / ((a[1] - b[1]) * 1e3)
        window = b[2] / (ghz * 1e3) - b[1]
        for key in ("config1", "config2_resident"):
            if key in traffic:
                traffic[key]["shader_clock_ghz"] = round(ghz, 3)
        out += ["", f"Shader clock of this box during the sweep kernels: **{ghz:.2f} GHz** (slope between the {a[1]:.1f} us and the {b[1]:.1f} us "
                    f"instantiation of the GRBM pass; the counter's window is {window:.1f} us wider than a dispatch's start / end stamps)."]
    # the K = 32 instantiations (tools/bench_configs.py under the same three counter passes)
    cdur = durations(g + "_cfg")
    if cdur:
        out += ["", "## K = 32 and the other BASELINE shapes: PMC per dispatch (tools/bench_configs.py, separate passes)", "",
                "| instantiation | dispatches | avg us | FETCH_SIZE KiB | WRITE_SIZE KiB | SQ_INSTS_VALU | VALU active % of wave-cycles | waiting % |", "|---|---|---|---|---|---|---|---|"]
        cf, cw = per_kernel(g + "_cfg_pmc_FETCH_SIZE", "FETCH_SIZE"), per_kernel(g + "_cfg_pmc_WRITE_SIZE", "WRITE_SIZE")
        csq = {c: per_kernel(g + "_cfg_pmc_SQ_INSTS_VALU", c) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY")}
        for inst, (avg, n, _med) in sorted(cdur.items()):
            wc = csq["SQ_WAVE_CYCLES"].get(inst) or 0
            out.append(f"| k_robot_sweep{inst} | {n} | {avg:.2f} | {cf.get(inst, 0):.1f} | {cw.get(inst, 0):.1f} | {csq['SQ_INSTS_VALU'].get(inst, 0):.4g} | "
                       f"{(csq['SQ_ACTIVE_INST_VALU'].get(inst, 0) / wc * 100) if wc else 0:.1f} | {(csq['SQ_WAIT_ANY'].get(inst, 0) / wc * 100) if wc else 0:.0f} |")
            traffic["configs" + inst] = {"fetch_kib": round(cf.get(inst, 0), 1), "write_kib": round(cw.get(inst, 0), 1),
                                         "valu_wave_instr": round(csq["SQ_INSTS_VALU"].get(inst, 0)), "kernel_trace_avg_us": round(avg, 2),
                                         "note": "mean over every dispatch of this instantiation in tools/bench_configs.py (several shapes share one)"}
    json.dump(traffic, open(os.path.join(prof, f"traffic_{tag}.json"), "w"), indent=1)
    for f in glob.glob(g + "_kt.json"):
        shutil.copy(f, os.path.join(prof, f"{tag}_bench_under_rocprof.json"))
    for f in glob.glob(g + "_kt_driver.json"):  # the bench line of the driver's command from the same box, beside its dispatch durations
        shutil.copy(f, os.path.join(prof, f"{tag}_driver_command_bench_line.json"))
    f = glob.glob(os.path.join(g + "_kt_driver", "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(prof, f"{tag}_driver_command_kernel_stats.csv"))
    cfg = g + "_cfg.log"
    if os.path.exists(cfg):
        out += ["", "## every BASELINE config on one MI355X (`tools/bench_configs.py` under rocprofv3 --kernel-trace --stats)", ""]
        out += ["* " + ln.strip() for ln in open(cfg) if "us / iteration" in ln]
    open(os.path.join(prof, f"{tag}_summary.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
