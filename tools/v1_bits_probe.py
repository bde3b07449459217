import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import cafe_amd
from tests import _orc as O
newick = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"
sp, ids, counts = O.load_family_table("tests/golden/example_data.tab")
t = O.PyTree(newick); counts = O.reorder_to_tree(sp, counts, t)
rng = O.range_from_max(int(counts.max())); prior = O.prior_poisson(1000, rng.root_min, 9.442907)
for lamv, muv in ((0.0017, -1.0), (0.002, 0.0015)):
    lam = np.full(t.n_nodes, lamv); mu = np.full(t.n_nodes, muv)
    so, fzo, mlo, amo, mpo = O.eval_posterior(t, counts, rng, lam, mu, prior)
    for k2 in ("v1ref", "v1", "auto"):
        eng = cafe_amd.Engine(0); eng.set_option("k1", "exact"); eng.set_option("k2", k2); eng.set_option("compress", 0)
        eng.set_tree(t.parent, t.left, t.right, t.branchlength)
        eng.set_families(counts, cafe_amd.FamilySizeRange(rng.min, rng.max, rng.root_min, rng.root_max))
        s, fz, ml, am, mp = eng.get_posterior(lam, mu, prior, per_family=True)
        d = np.abs(ml.view(np.int64) - mlo.view(np.int64)); dp = np.abs(mp.view(np.int64) - mpo.view(np.int64))
        print("lam %g mu %g k2=%s: max_lik bit-identical %d of %d (worst %d ulp); max_post identical %d (worst %d ulp); score equal %s" % (lamv, muv, k2, (d==0).sum(), len(d), d.max(), (dp==0).sum(), dp.max(), s == so))
        eng.close()
