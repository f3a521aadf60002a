"""Diagnostics: host time per call in the dynamic-tick loop of bench.py (no synchronisation inside the loop)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa
from magics_amd import World, scenarios as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = S.grid_scenario(N, 16, interrobot=True, seed=805)
sc["ir"] = []
_st = torch.cuda.Stream()  # (a stream of its own, as in bench.py: launches on the null stream pay for its implicit synchronisation)
w = World(sc["params"], stream=_st.cuda_stream); S.populate(w, sc)
rng = np.random.default_rng(805)
base = np.array([[rb["pos"][0], 0.5, rb["pos"][1]] for rb in sc["robots"]], dtype=np.float32)
tk = S.tick_inputs(sc)
nxt, _, _ = w.update_topology(base, 8.0, 1)
w.iterate(sc["steps"]); w.synchronize()
poss = [base + rng.normal(0, 0.15, size=base.shape).astype(np.float32) for _ in range(60)]
for pos in poss * 8:  # (a host core that has just woken up runs the bookkeeping a quarter slower: half a second of ticks first)
    nxt, c, d = w.update_topology(pos, 8.0, nxt)
    w.tick(steps=sc["steps"], **tk)
w.synchronize()
for label, do_topo in (("tick only", False), ("topology + tick", True)):
    T = {"topo": 0.0, "tick": 0.0}
    t_all = time.perf_counter()
    for pos in poss:
        t0 = time.perf_counter()
        if do_topo:
            nxt, c, d = w.update_topology(pos, 8.0, nxt)
        t1 = time.perf_counter()
        w.tick(steps=sc["steps"], **tk)
        t2 = time.perf_counter()
        T["topo"] += t1 - t0; T["tick"] += t2 - t1
    w.synchronize()
    tot = time.perf_counter() - t_all
    print(label, {k: round(v / len(poss) * 1e6, 1) for k, v in T.items()}, "us per call;", round(tot / len(poss) * 1e6, 1), "us per tick overall")
