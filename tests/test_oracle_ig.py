"""Pin the oracle's information-gain primitives against vectors produced by the reference's own
Map / edfMap / targetMap / ig_mcts objects (tests/golden/ig_primitives.npz)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ig_primitives.npz")
WORLDS = ["corridor", "rects"]


@pytest.fixture(scope="module")
def z():
    orc.build()
    return np.load(GOLD)


@pytest.mark.parametrize("w", WORLDS)
def test_edt_exact(z, w):
    occ = orc.rasterize(z[w + "__obstacles"])
    # the reference's EDF is indexed [row from y] WITHOUT the y-flip of Map: same raster array
    edf, d2 = orc.edt(occ)
    assert np.array_equal(edf, z[w + "__edf"])
    assert np.array_equal(np.sqrt(d2.astype(np.float64)) * 0.1, edf)


def test_edf_known_answers(z):
    edf = z["corridor__edf"]
    f = lambda x, y: edf[int(np.floor((y + 15) / .1)), int(np.floor((x + 15) / .1))]
    assert abs(f(-5, 0) - 2.0) < 1e-12 and abs(f(0, 0) - 2.8284271247) < 1e-9
    assert abs(f(1.9, 1.0) - 1.0198039027) < 1e-9 and abs(edf.max() - 7.0710678119) < 1e-9


@pytest.mark.parametrize("w", WORLDS)
def test_visible_cells_bit_exact(z, w):
    edf = np.ascontiguousarray(z[w + "__edf"])
    for p, m in zip(z[w + "__vis_poses"], z[w + "__vis_masks"]):
        got = orc.visible_cells(edf, p)
        assert np.array_equal(got, m), p
    if w == "corridor":  # SURVEY 8(c) known answers: 52 / 50 / 29 visible cells
        cnt = [int(sum(bin(int(x)).count("1") for x in m)) for m in z[w + "__vis_masks"][:3]]
        assert cnt == [52, 50, 29]


@pytest.mark.parametrize("w", WORLDS)
def test_check_visibility(z, w):
    edf = np.ascontiguousarray(z[w + "__edf"])
    got = np.array([orc.check_visibility(edf, a, b) for a, b in zip(z[w + "__cv_a"], z[w + "__cv_b"])])
    assert np.array_equal(got, z[w + "__cv_visible"])
    assert got.any() and (~got).any()


@pytest.mark.parametrize("w", WORLDS)
def test_belief_update_and_reward(z, w):
    edf = np.ascontiguousarray(z[w + "__edf"])
    bel = np.ones((60, 60))
    for t in range(z[w + "__upd_poses"].shape[0]):
        obs = orc.update_belief(bel, edf, z[w + "__upd_poses"][t], z[w + "__upd_dets"][t], z[w + "__upd_ndet"][t])
        assert np.array_equal(obs, z[w + "__upd_observed"][t])
        assert np.array_equal(bel, z[w + "__upd_belief"][t])  # same multiplication order -> bit-exact
        r = orc.mi_reward(bel, obs)
        assert abs(r - z[w + "__upd_reward"][t]) <= 1e-12 * max(1, abs(r))
    for m, r in zip(z[w + "__vis_masks"], z[w + "__mi_reward"]):
        assert abs(orc.mi_reward(bel, m) - r) <= 1e-12 * max(1, abs(r))


def test_mi_known_answer():
    bel = np.ones((60, 60))
    m = np.zeros(60, dtype=np.uint64)
    m[3] = 1 << 7
    assert abs(orc.mi_reward(bel, m) - 0.020654806323) < 1e-11


@pytest.mark.parametrize("w", WORLDS)
def test_next_pose(z, w):
    edf = np.ascontiguousarray(z[w + "__edf"])
    acts = [np.array([v, ww]) for v in (0.0, 2.0, 4.0) for ww in (-0.5 * np.pi, 0, 0.5 * np.pi)]
    nxt, feas = z[w + "__np_next"], z[w + "__np_feasible"]
    for q, p in enumerate(z[w + "__vis_poses"]):
        for k, a in enumerate(acts):
            r = orc.next_pose(edf, p, a)
            assert (r is not None) == bool(feas[q, k]), (q, k)
            if r is not None:
                assert np.abs(r - nxt[q, k]).max() <= 1e-12
    assert feas.any() and (~feas).any()
