import sys, time, numpy as np
sys.path.insert(0,"minimap2-fpga_amd"); sys.path.insert(0,"tests")
import mm2chain, oracle_binding as ob
from mm2chain import params, synth
mm2chain.init()
P = params.map_ont(); c = mm2chain.split_model("map-ont")
rng = np.random.default_rng(987)
def timed(fn):
    t0=time.perf_counter(); fn(); return (time.perf_counter()-t0)*1e3
tot = dict(model=0., best=0., cpu=0., gpu=0.)
for prof in ("mixed", "dense", "colinear", "sparse"):
    row = dict(model=0., best=0., cpu=0., gpu=0.); ag = 0; ng = 0
    for n in rng.integers(60, 9000, 12):
        t = synth.make_stream(prof, 1, int(n), seed=int(rng.integers(1 << 30)))[1].numpy().view(np.uint64)
        mm2chain.chain_task(P, t, 0.15)
        _, ts, tt = ob.predict(t, P.max_dist_x)
        hw = min(timed(lambda: mm2chain.chain_task(P, t, 0.15)) for _ in range(3)); sw = min(timed(lambda: ob.chain_fpv(P, t, 0.15)) for _ in range(2))
        pg = c["K1_HW"]*t.shape[0] + c["K2_HW"]*ts + c["C_HW"] < c["K_SW"]*tt + c["C_SW"]
        ag += int(pg == (hw < sw)); ng += int(pg)
        for k, v in (("model", hw if pg else sw), ("best", min(hw, sw)), ("cpu", sw), ("gpu", hw)): row[k] += v; tot[k] += v
    print(prof, {k: round(v, 1) for k, v in row.items()}, "agree", ag, "model->gpu", ng)
print("total", {k: round(v, 1) for k, v in tot.items()})
