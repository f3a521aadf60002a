"""Host-side cost of one driver tick through the Python binding vs the device time (diagnostic)."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402,F401  (HIP runtime preload)
from magics_amd import World, scenarios as S  # noqa: E402


def main():
    sc = S.grid_scenario(1000, 16, interrobot=False)
    w = World(sc["params"])
    S.populate(w, sc)
    tk = S.tick_inputs(sc)
    steps = sc["steps"]
    for _ in range(20):
        w.update_priors(**tk)
        w.iterate(steps)
    w.synchronize()
    n = 300

    def timeit(fn, sync_each):
        w.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
            if sync_each:
                w.synchronize()
        w.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    print("update_priors  async %.1f us  sync-each %.1f us" % (timeit(lambda: w.update_priors(**tk), False), timeit(lambda: w.update_priors(**tk), True)))
    print("iterate(10)    async %.1f us  sync-each %.1f us" % (timeit(lambda: w.iterate(steps), False), timeit(lambda: w.iterate(steps), True)))

    def tick():
        w.update_priors(**tk)
        w.iterate(steps)
    print("tick           async %.1f us  sync-each %.1f us" % (timeit(tick, False), timeit(tick, True)))
    # host-only cost: time to enqueue without waiting (queue depth limited to keep it honest)
    w.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        w.update_priors(**tk)
    t1 = time.perf_counter()
    w.synchronize()
    print("enqueue cost of update_priors: %.1f us/call" % ((t1 - t0) / 50 * 1e6))
    t0 = time.perf_counter()
    for _ in range(50):
        w.iterate(steps)
    t1 = time.perf_counter()
    w.synchronize()
    print("enqueue cost of iterate: %.1f us/call" % ((t1 - t0) / 50 * 1e6))


if __name__ == "__main__":
    main()
