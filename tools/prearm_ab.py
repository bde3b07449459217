import sys, time, os
sys.path.insert(0, os.getcwd())
import bench, cafe_amd
w = bench.synthetic_workload("cfg2", 0, 1, "weak", None)
eng = cafe_amd.Engine(0)
leg = bench.Leg(w, 0, None, shared_engine=eng)
leg.prepare_rates(600)
leg.prime()
for rep in range(4):
    for pa in (0, 1):
        eng.set_option("prearm", pa)
        for s in range(50): leg.step(s)
        s0 = eng.prearm_stats()
        t0 = time.perf_counter()
        for s in range(500): leg.step(50 + s)
        dt = time.perf_counter() - t0
        s1 = eng.prearm_stats()
        print("prearm=%d  %.2f us per step   used +%d let go +%d expired +%d" % (pa, 1e6 * dt / 500, s1["used"] - s0["used"], s1["let_go"] - s0["let_go"], s1["expired"] - s0["expired"]), flush=True)
eng.close()
