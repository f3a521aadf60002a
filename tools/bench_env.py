"""Environment rasteriser on one MI355X: wall time of env_to_sdf_image through the C ABI (device
rasterise + blur + the download of the image) for the reference's Junction Twoway tile at growing
grid sizes, next to the CPU restatement.  Kernel times come from running this script under
`rocprofv3 --kernel-trace --stats` (profiles/); algorithmic bytes per pixel: 1 written by the
rasteriser, 1 read + 4 written by the vertical blur pass, 4 read + 1 written by the horizontal one."""
import json
import sys
import time

sys.path.insert(0, ".")
from magics_amd import environment as ENV, scenarios as S  # noqa: E402

cpu = "--no-cpu" not in sys.argv
out = []
for tiles in (1, 5, 10, 20):
    env = S.junction_environment(tiles)
    ENV.env_to_sdf_image(env)  # first call: allocations, code load
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        img = ENV.env_to_sdf_image(env)
    dev = (time.perf_counter() - t0) / n
    row = {"tiles": f"{tiles}x{tiles}", "pixels": img.shape[0] * img.shape[1], "device_call_ms": round(dev * 1e3, 3),
           "Mpixel_per_s": round(img.shape[0] * img.shape[1] / dev / 1e6, 1)}
    if cpu and tiles <= 5:
        from oracle import env as E
        t0 = time.perf_counter()
        ref = E.env_to_sdf_image(env)
        row["cpu_numpy_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        row["identical"] = bool((ref == img[:, :, 0]).all())
    out.append(row)
    print(json.dumps(row), flush=True)
