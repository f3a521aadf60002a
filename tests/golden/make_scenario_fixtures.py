"""Parses the reference's scenario directories (config/scenarios/*/{config.toml, environment.yaml,
formation.yaml}: data files, inputs of the path) with the product's readers and writes what they
yield — only the sections the hot path consumes — as canonical JSON to
tests/golden/scenarios.json, so that the GPU box (which has no /root/reference) can run the same
scenarios, and so that tests/test_scenario_readers.py can tell when a reader changes its output.

Run in the build container:  python tests/golden/make_scenario_fixtures.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from magics_amd import config  # noqa: E402

REF = "/root/reference/config/scenarios"
KEEP = ("environment_image", "gbp", "robot", "simulation")


def slim(sc):
    return {"name": sc["name"], "config": {k: sc["config"][k] for k in KEEP}, "environment": sc["environment"],
            "formation": sc["formation"]}


def main():
    out = {name: slim(config.load_scenario(os.path.join(REF, name))) for name in sorted(os.listdir(REF))}
    with open(os.path.join(ROOT, "tests", "golden", "scenarios.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=1, sort_keys=True)
    print(len(out), "scenarios")


if __name__ == "__main__":
    main()
