"""Diagnostics: registers, spills and scratch of every k_robot_sweep instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [extra hipcc flags ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "magics_amd", "csrc", "mgx_kernels.hip")
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage",
                      *sys.argv[1:], src, "-o", "/tmp/_kernel_resources.o"], capture_output=True, text=True).stderr
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
for name, r in rows.items():
    m = re.match(r"_ZN3mgx13k_robot_sweepILi(n?\d+)ELi(\d)ELb(\d)EE", name)
    if not m:
        continue
    k = m.group(1).replace("n", "-")
    print(f"k_robot_sweep<{k:>3s}, {m.group(2)}, {'true ' if m.group(3) == '1' else 'false'}>{'':18s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} "
          f"{r.get('VGPRs Spill', 0):8d} {r.get('SGPRs Spill', 0):8d} {r.get('ScratchSize', 0):8d} {r.get('Occupancy', 0):4d}")
