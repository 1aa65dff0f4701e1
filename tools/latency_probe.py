import os, sys, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
from mpmcxx_amd import gen_box
from mpmcxx_amd import energy, pqr
# usage: python tools/latency_probe.py [key=value ...]      (keys of mpmc_debug_configure, e.g. side_stream=0 single_launch=0)
for kv in sys.argv[1:]:
    k, _, v = kv.partition("=")
    energy.configure(k, float(v))
wd = tempfile.mkdtemp()
for name in ("lj1000", "ion1000_polar", "ion216_polar"):
    inp, _ = gen_box.materialize(name, wd)
    atoms, basis, opts = pqr.load_case(inp)
    S = energy.System(atoms, basis, opts)
    S.energy()
    t0 = time.perf_counter()
    for _ in range(300):
        S.energy()
    print(name, " ".join(sys.argv[1:]), f"{(time.perf_counter()-t0)/300*1e6:.1f} us")
    S.close()
