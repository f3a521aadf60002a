"""Diagnostics: registers, spills and scratch of every k_robot_sweep instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [extra hipcc flags ...]"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "magics_amd", "csrc", "mgx_sweep_inst.hip")


def remarks(fk):
    f, k = fk
    return subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-ffp-contract=off", f"-DMGX_FLAVOR={f}", f"-DMGX_KSET={k}",
                           "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:], src, "-o", f"/tmp/_kernel_resources_{f}{k}.o"],
                          capture_output=True, text=True).stderr


with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
    out = "\n".join(ex.map(remarks, [(f, k) for f in range(3) for k in range(3)]))
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'v-spill':>8s} {'s-spill':>8s} {'scratch':>8s} {'occ':>4s}")
table = []
for name, r in rows.items():
    m = re.match(r"_ZN3mgx13k_robot_sweepILi(n?\d+)ELi(\d)ELb(\d)ELb(\d)EE", name)
    if not m:
        continue
    k = int(m.group(1).replace("n", "-"))
    table.append((k, int(m.group(2)), int(m.group(3)), int(m.group(4)), r))
for k, irm, per, sh, r in sorted(table, key=lambda t: (t[0] <= 0, t[0], t[1], t[2], t[3])):
    label = f"k_robot_sweep<{k:>3d}, {irm}, {'true ' if per else 'false'}{', shard' if sh else ''}>"
    print(f"{label:44s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} "
          f"{r.get('VGPRs Spill', 0):8d} {r.get('SGPRs Spill', 0):8d} {r.get('ScratchSize', 0):8d} {r.get('Occupancy', 0):4d}")
