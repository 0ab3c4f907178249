"""oracle/libmtm_cpu.so (the C++ CPU port bench.py times as cpu_baseline) computes what the reference pipeline computes:
its pre-NMS hits equal the float64 oracle's - same boxes, scores within the float32-DFT noise cv2 itself has."""
import numpy as np
import pytest

import mtm_oracle as O
import synth
from helpers import canon, coin_templates, load_coins


@pytest.fixture(scope="module")
def cpu():
    import build_oracle
    build_oracle.build()
    import mtm_cpu
    return mtm_cpu


@pytest.mark.parametrize("method,thr", [(5, 0.5), (3, 0.95), (1, 0.2)])
def test_cpu_port_equals_oracle_on_coins(cpu, method, thr):
    coins = load_coins()
    small, big = coin_templates(coins)
    lt = [("small", small), ("big", big)]
    got, info = cpu.find_matches(lt, coins, method, thr, n_threads=2)
    exp = O.find_matches(lt, coins, method, float("inf"), thr)
    a, b = canon(got), canon(exp)
    assert len(a) == len(b) > 0
    for g, e in zip(sorted(a, key=lambda h: (h[0], h[1])), sorted(b, key=lambda h: (h[0], h[1]))):
        assert g[0] == e[0] and g[1] == e[1] and abs(g[2] - e[2]) <= 2e-5, (g, e)
    assert info["total_s"] > 0 and info["threads"] == 2


def test_cpu_port_blocks_and_threads(cpu):
    """an image larger than one 512 x 512 DFT block, templates of different sizes, more threads than templates"""
    img, units, plants = synth.make_workload(seed=8, image_hw=(700, 1100), n_base=5, templ=48, noisy_per_unit=2)
    units.append(("wide", np.ascontiguousarray(img[300:330, 500:620])))
    got, _ = cpu.find_matches(units, img, 5, 0.5, n_threads=8)
    exp = O.find_matches(units, img, 5, float("inf"), 0.5)
    assert sorted((h[0], h[1]) for h in got) == sorted((h[0], h[1]) for h in exp)
    assert {(p[0], p[1]) for p in plants} <= {(h[0], h[1]) for h in got}
    ga, ea = canon(got), canon(exp)
    assert max(abs(g[2] - e[2]) for g, e in zip(sorted(ga, key=lambda h: (h[0], h[1])), sorted(ea, key=lambda h: (h[0], h[1])))) <= 2e-5
