"""Pin the oracle to the committed golden digests (generated from the real reference by tests/golden/make_golden.py).
Runs anywhere: needs neither /root/reference nor oracle/_ref."""
import json
import os

import pytest

from backends import Orc
from cases import digest

HERE = os.path.dirname(os.path.abspath(__file__))
import sys
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

with open(os.path.join(HERE, "golden", "primitives_golden.json")) as f:
    GOLD = json.load(f)["golden"]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_primitives_match_golden(depth):
    got = make_golden.prim_digests(Orc, depth)
    want = GOLD[str(depth)]["prims"]
    assert set(got) == set(want)
    bad = [k for k in want if got[k] != want[k]]
    assert not bad, bad[:10]
    assert len(want) > 2000


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_motion_estimate_matches_golden(depth):
    assert make_golden.me_digests(Orc, depth) == GOLD[str(depth)]["me"]
    assert make_golden.umh_results(Orc, depth) == GOLD[str(depth)]["umh"]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_chroma_motion_estimate_matches_golden(depth):
    assert make_golden.chroma_me_results(Orc, depth) == GOLD[str(depth)]["chroma_me"]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_loop_filter_primitives_match_golden(depth):
    got, want = make_golden.loop_digests(Orc, depth), GOLD[str(depth)]["loop"]
    assert len(want) >= 380 and got == want, [k for k in want if got.get(k) != want[k]][:8]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_weightp_analysis_matches_golden(depth):
    got = {k: digest(v) for k, v in make_golden.weightp_results(Orc, depth).items()}
    assert len(got) == 8 and got == GOLD[str(depth)]["weightp"]
    got = {k: digest(v) for k, v in make_golden.lookahead_weightp_results(Orc, depth).items()}
    assert len(got) == 8 and got == GOLD[str(depth)]["lookahead_weightp"]
    got = {k: digest(v) for k, v in make_golden.aq_results(Orc, depth).items()}
    assert len(got) == 32 and got == GOLD[str(depth)]["aq"]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_sea_search_matches_golden(depth):
    """--me sea: the window-sum planes and the search results of the real reference, committed as golden (runs without oracle/_ref)."""
    got, want = make_golden.sea_results(Orc, depth), GOLD[str(depth)]["sea"]
    assert got["planes"] == want["planes"]
    assert len(want["cases"]) == 60 and got["cases"] == want["cases"]


def test_oracle_cutree_matches_golden():
    got = {k: [digest(v[0]), digest(v[1]), digest(v[2])] for k, v in make_golden.cutree_results(Orc).items()}
    assert len(got) == 20 and got == GOLD["cutree"]


def test_oracle_coefficient_scan_primitives_match_golden():
    got = make_golden.coef_digests(Orc)
    want = GOLD["coef"]
    assert len(want) >= 600 and got == want, [k for k in want if got.get(k) != want[k]][:8]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_motion_compensation_matches_golden(depth):
    assert {k: digest(v) for k, v in make_golden.mc_results(Orc, depth).items()} == GOLD[str(depth)]["mc"]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_bipred_matches_golden(depth):
    assert {k: digest(v) for k, v in make_golden.bipred_results(Orc, depth).items()} == GOLD[str(depth)]["bipred"]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_lowres_pass_matches_golden(depth):
    assert make_golden.lowres_digests(Orc, depth) == GOLD[str(depth)]["lowres"]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_lookahead_cost_matches_golden(depth):
    assert make_golden.lookahead_digests(Orc, depth) == GOLD[str(depth)]["lookahead"]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_lookahead_b_cost_matches_golden(depth):
    assert {k: digest(v) for k, v in make_golden.lookahead_b_results(Orc, depth).items()} == GOLD[str(depth)]["lookahead_b"]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_mvcost_matches_golden(depth):
    for qp, d in GOLD[str(depth)]["mvcost"].items():
        assert digest(Orc(depth).mvcost_table(int(qp))) == d


def test_framepass_config_golden_is_reproducible_on_the_cpu_tier():
    """tests/golden/framepass_configs_golden.json (the digests the GPU tier compares BASELINE configs[2..4] against) is what the oracle produces:
    the small case is re-derived here, and the file holds every case of the generator with matching parameters."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_framepass_golden as mg
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "framepass_configs_golden.json")))["cases"]
    assert sorted(gold) == sorted(mg.CASES)
    for name, case in gold.items():
        assert tuple(case["params"][k] for k in ("width", "height", "depth", "qp", "method", "merange", "subme", "seed", "pass")) == mg.CASES[name]
        assert case["info"]["nonzero_mvs"] > 0 and case["info"]["numSig"] > 0
    small = [n for n in mg.CASES if n.startswith("small")][0]
    _, dg, _ = mg.run_oracle(small)
    assert dg == gold[small]["digests"]
