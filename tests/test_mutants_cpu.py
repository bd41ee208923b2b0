"""Mutation sensitivity of the parity suite, CPU half (no GPU): does the suite BITE?

The reference ships no vectors for the hot path, so "the HIP path equals the oracle" only means something if the suite's
assertions would fail for an oracle that restated a quirk of SURVEY.md §8.1 wrongly. oracle/mutants.h + tests/search_restatement.py
can "fix" each quirk one at a time; here every such mutant must
  (1) change the outputs of the committed quirk fixtures (tests/golden/quirks.npz) by more than the suite's tolerances, and
  (2) be caught at the fixture case and by the assertion that tests/mutant_runs.py::TABLE names (the table of DESIGN.md §5).
tests/test_mutants_gpu.py then holds the HIP path to the faithful oracle on the same fixtures and away from every mutant.
"""
import numpy as np
import pytest

from tests import helpers, mutant_runs as M, oracle_lib, quirk_fixtures as Q


@pytest.fixture(scope="module")
def faithful():
    return M.cpu_outputs()


@pytest.fixture(scope="module")
def golden():
    return np.load(helpers.golden_path("quirks.npz"))


def test_committed_fixtures_are_the_generators_inputs(golden):
    """tests/golden/quirks.npz holds exactly what tests/quirk_fixtures.py builds (the generator is the documentation of the bytes)."""
    from tests.golden import make_golden_quirks as G
    d = G.collect()
    assert set(d) == set(golden.files)
    for k in d:
        assert np.array_equal(np.asarray(d[k]), golden[k], equal_nan=True), k


def test_faithful_oracle_reproduces_the_committed_outputs(faithful, golden):
    for name in ("main", "dark", "behind", "minfts", "away"):
        o = faithful["sparse:" + name]
        assert np.array_equal(o["T"], golden[f"{name}_out_T"], equal_nan=True) and o["n"] == int(golden[f"{name}_out_n"])
        for k in ("iters", "exit_code", "n_ref", "n_vis"):
            assert o[k] == list(golden[f"{name}_out_{k}"]), (name, k)
    assert np.array_equal(faithful["align2d"][0], golden["a2d_out_conv"]) and np.array_equal(faithful["align2d"][1], golden["a2d_out_px"], equal_nan=True)
    for i, k in enumerate(("warp_out_affine", "warp_out_level", "warp_out_border", "warp_out_patch")):
        assert np.array_equal(faithful["warp"][i], golden[k]), k
    for name in Q.SEARCH_WORLDS:
        lst, mask = faithful["search:" + name]
        assert np.array_equal(np.array(lst, np.float64).reshape(-1, 5), golden[f"search_{name}_matches"])
        assert np.array_equal(np.packbits(mask == 255, axis=1), golden[f"search_{name}_mask_rows"])


def test_the_fixtures_reach_what_they_are_for(faithful):
    """The inputs really exercise the quirks: border features drop out between levels, the dark frame runs into exact chi2 ties
    (cap reached without a revert), the flipped seed tracks points BEHIND the camera, Min_fts counts uninitialised features,
    Align2D has failures and last-column windows, the warp prelude has all three search levels."""
    m = faithful["sparse:main"]
    assert m["n_ref"][0] > m["n_ref"][1] > m["n_ref"][2] and m["n_vis"][0] < m["n_ref"][0] and m["n"] == m["n_vis"][0]
    d = faithful["sparse:dark"]
    assert d["iters"][2] == 4 and d["exit_code"][2] == 0            # four iterations of identical chi2: `>` never reverts
    assert faithful["sparse:behind"]["n"] > 50
    assert faithful["sparse:minfts"]["n"] == 8
    aw = faithful["sparse:away"]
    assert aw["n"] == 0 and aw["n_vis"][:3] == [0, 0, 0] and aw["exit_code"][:3] == [2, 2, 2] and all(np.isnan(aw["chi2"][:3]))
    conv, px = faithful["align2d"]
    assert 4 <= (~conv).sum() <= 10 and conv[30:].any()
    assert not conv[28:30].any() and np.isnan(px[28:30]).all()       # singular H: NaN written back (A2, A4)
    assert set(faithful["warp"][1]) == {0, 1, 2}
    assert len(faithful["search:dense"][0]) == 200 and len(faithful["search:std"][0]) < 200


def test_mutation_build_with_no_switch_is_the_faithful_oracle(faithful):
    """-DORACLE_MUTANTS alone changes nothing: every output bit for bit (so a difference below is the switch, not the build)."""
    with oracle_lib.mutant("MUT_NONE") as lib:
        assert lib.oracle_get_mutant() == 0
        o = M.cpu_outputs(domains=("sparse", "align2d", "warp", "pose_opt", "detector"))
    for case, a in o.items():
        b = faithful[case]
        if case == "detector":
            assert M.first_difference(case, a, b) is None
        elif case == "pose_opt":
            for (Ta, ra, sa), (Tb, rb, sb) in zip(a, b):
                assert np.array_equal(Ta, Tb) and np.array_equal(ra, rb)
                assert all(np.array_equal(sa[k], sb[k]) for k in sa)
        elif case.startswith("sparse:"):
            assert np.array_equal(a["T"], b["T"], equal_nan=True) and {k: a[k] for k in a if k != "T" and k != "chi2"} == {k: b[k] for k in b if k != "T" and k != "chi2"}
            assert np.array_equal(a["chi2"], b["chi2"], equal_nan=True)
        else:
            assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b)), case


def test_every_quirk_of_the_survey_has_a_mutant():
    quirks = {row[2] for row in M.TABLE.values()}
    assert {"Q1", "Q3", "Q4", "Q5", "Q6", "Q8", "Q9", "Q10", "Q11", "A1", "A2", "A3", "A4", "W1", "W2", "W3", "S1"} <= quirks
    assert {m for m in M.TABLE if m.startswith("MUT_")} == set(oracle_lib.MUTANTS) - {"MUT_NONE"}
    assert sum(1 for m in M.TABLE if M.domain(m) == "pose_opt") >= 6          # Optimizer::PoseOptimization too
    from tests import search_restatement as SR
    assert {m for m in M.TABLE if not m.startswith("MUT_")} == set(SR.SEARCH_MUTANTS)


@pytest.mark.parametrize("mutant", M.ALL_MUTANTS)
def test_mutant_differs_from_the_faithful_oracle(faithful, mutant):
    case, check, quirk, cite = M.TABLE[mutant]
    dom = M.domain(mutant)
    out = M.cpu_outputs(mutant, domains=(dom,), search_worlds=[case.split(":")[1]] if dom == "search" else None)
    diffs = M.first_differences(out, faithful)
    assert diffs[case] == check, (mutant, quirk, cite, diffs)
