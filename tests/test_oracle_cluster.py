"""k-cluster model on the oracle (cafe/cafe_main.c:165-253, cafe/cafe_tree.c:704-850).  The reference holds no
numeric transcript of a `lambda -k` run (its tests cover the argument parsing and the weight copy only), so the
restatement is pinned by the reference's weight-copy vectors and by identities that tie it to the single-model
posterior, which IS pinned by transcripts."""
import os

import numpy as np

from tests import _orc as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEWICK = "(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"


def _example():
    sp, ids, counts = O.load_family_table(os.path.join(ROOT, "tests", "golden", "example_data.tab"))
    t = O.PyTree(NEWICK)
    counts = O.reorder_to_tree(sp, counts, t)
    rng = O.range_from_max(int(counts.max()))
    prior = O.prior_poisson(1000, rng.root_min, 9.442907)
    return t, counts, rng, prior


def test_copy_weights_known_answers():
    # /root/reference/tests/test.cpp:1600-1625 (FirstTestGroup, initialize_k_weights)
    params = np.arange(100) / 100.0
    w = O.copy_weights(params, 2, 5)
    assert np.allclose(w, [.02, .03, .04, .05, .86], atol=1e-4)
    w = O.copy_weights(params, 15, 6)
    assert np.allclose(w, [.15, .16, .17, .18, .19, .15], atol=1e-4)


def test_one_cluster_is_the_plain_posterior():
    t, counts, rng, prior = _example()
    lam = np.full((1, t.n_nodes), 0.0017)
    mu = np.full((1, t.n_nodes), -1.0)
    s1, fz1, ml, am, mp = O.eval_posterior(t, counts, rng, lam[0], mu[0], prior)
    s, fz, MAP, pz, neww = O.clustered_posterior(t, counts, rng, lam, mu, [1.0], prior)
    # MAP_0 = max_post * 1, membership 1, MAP = 1 * MAP_0
    assert np.array_equal(MAP, mp) and np.all(pz == 1.0) and neww[0] == 1.0 and fz == fz1 == -1
    assert abs(s - s1) <= 1e-12 * abs(s1)


def test_two_clusters_follow_the_mixture_formulas():
    t, counts, rng, prior = _example()
    lams = [0.0017, 0.009]
    lam = np.array([np.full(t.n_nodes, x) for x in lams])
    mu = np.full_like(lam, -1.0)
    w = np.array([0.3, 0.7])
    mp = np.stack([O.eval_posterior(t, counts, rng, lam[k], mu[k], prior)[4] for k in range(2)], axis=1)
    s, fz, MAP, pz, neww = O.clustered_posterior(t, counts, rng, lam, mu, w, prior)
    mapk = mp * w
    tot = mapk[:, 0] + mapk[:, 1]
    assert np.array_equal(pz, mapk / tot[:, None])
    assert np.array_equal(MAP, pz[:, 0] * mapk[:, 0] + pz[:, 1] * mapk[:, 1])
    assert abs(s - np.log(MAP).sum()) <= 1e-12 * abs(s)
    assert np.allclose(neww, pz.sum(axis=0) / len(counts), rtol=1e-13) and abs(neww.sum() - 1) < 1e-12
    # duplicate rows copy their reference row's values (cafe_main.c:220-229)
    c2 = np.vstack([counts, counts[:3]])
    ref = np.arange(len(c2), dtype=np.int32)
    ref[-3:] = [0, 1, 2]
    s2, fz2, MAP2, pz2, neww2 = O.clustered_posterior(t, c2, rng, lam, mu, w, prior, ref=ref)
    assert np.array_equal(MAP2[-3:], MAP[:3]) and np.array_equal(pz2[-3:], pz[:3])
