"""Extracts the known-answer vectors held by the reference's OWN unit tests for the hot path
into tests/golden/reference_vectors.json (data only: inputs and expected outputs).

Run in the build container (needs /root/reference):  python tests/golden/make_reference_vectors.py

Sources:
  crates/gbp_schedule/src/schedules/{centered,soon_as_possible,late_as_possible,
      interleave_evenly,half_beginning_half_end}.rs  #[cfg(test)] tables
  crates/magics/src/utils.rs:96-133                     get_variable_timesteps vectors
  crates/magics/src/factorgraph/factor/marginalise_factor_distance.rs:140-233
"""
import json
import os
import re

REF = "/root/reference"
KINDS = {"centered": 0, "soon_as_possible": 1, "late_as_possible": 2, "interleave_evenly": 3,
         "half_beginning_half_end": 4}


def schedule_cases(path):
    src = open(path).read()
    tests = src[src.index("#[cfg(test)]"):]
    cases, cur, internal = [], None, None
    for line in tests.splitlines():
        s = line.strip()
        if s.startswith("//"):
            continue
        m = re.search(r"internal:\s*(\d+)", s)
        if m:
            internal = int(m.group(1))
        m = re.search(r"external:\s*(\d+)", s)
        if m and internal is not None:
            cur = {"internal": internal, "external": int(m.group(1)), "steps": []}
            internal = None
        m = re.search(r"Some\(ts\((true|false),\s*(true|false)\)\)", s)
        if m and cur is not None:
            cur["steps"].append([m.group(1) == "true", m.group(2) == "true"])
        if re.search(r"schedule\.next\(\),\s*None", s) and cur is not None:
            cases.append(cur)
            cur = None
    return cases


def timestep_cases(path):
    src = open(path).read()
    tests = src[src.index("fn test_get_variable_timesteps"):]
    hs = [int(x) for x in re.findall(r"let lookahead_horizon = (\d+);", tests)]
    ms = [int(x) for x in re.findall(r"let lookahead_multiple = (\d+);", tests)]
    vs = [[int(y) for y in x.split(",") if y.strip()] for x in re.findall(r"vec!\[([0-9,\s]+)\]", tests)]
    assert len(hs) == len(ms) == len(vs)
    return [{"horizon": h, "multiple": m, "timesteps": v} for h, m, v in zip(hs, ms, vs)]


def main():
    out = {"schedules": {}, "timesteps": [], "marginalise": {}}
    for name, kind in KINDS.items():
        cases = schedule_cases(os.path.join(REF, "crates/gbp_schedule/src/schedules", name + ".rs"))
        out["schedules"][name] = {"kind": kind, "cases": cases}
    out["timesteps"] = timestep_cases(os.path.join(REF, "crates/magics/src/utils.rs"))
    # marginalise_factor_distance.rs:212-233: a 4-dim potential passes through unchanged
    out["marginalise"]["passthrough"] = {
        "eta": [0.0, 1.0, 2.0, 3.0],
        "lam": [[5.0, 0.2, 0.0, 0.0], [0.2, 5.0, 0.0, 0.0], [0.0, 0.0, 5.0, 0.3], [0.0, 0.0, 0.3, 5.0]],
        "marg_idx": 0,
    }
    # :140-210: block layout of the 8x8 (1..64 row-major): marg_idx 0 -> aa = upper-left,
    # ab = upper-right, ba = lower-left, bb = lower-right; marg_idx 4 -> the mirror image
    out["marginalise"]["blocks_8x8"] = {"matrix": "1..64 row-major",
                                        "0": {"aa": "ul", "ab": "ur", "ba": "ll", "bb": "lr"},
                                        "4": {"aa": "lr", "ab": "ll", "ba": "ur", "bb": "ul"}}
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "reference_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    n = sum(len(v["cases"]) for v in out["schedules"].values())
    print(f"{n} schedule cases, {len(out['timesteps'])} timestep cases")


if __name__ == "__main__":
    main()
