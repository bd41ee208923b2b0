"""Runs every mutant of the oracle (oracle/mutants.h, tests/search_restatement.py) over the quirk fixtures.

`cpu_outputs(mutant)` = what the CPU oracle with that quirk "fixed" returns for every fixture case; the same call with
mutant None = the faithful restatement. `first_differences(a, b)` = per case, the first assertion of the parity suite
that `b` fails against `a` (None: passes all). TABLE = the committed answer: for each mutant the case and the assertion
that catches it — tests/test_mutants_cpu.py checks the table against the CPU oracle, tests/test_mutants_gpu.py checks the
HIP path against the faithful oracle and against every row."""
from __future__ import annotations

import contextlib

from tests import oracle_lib, quirk_fixtures as Q
from tests import search_restatement as SR

# mutant -> (fixture case that catches it, the first assertion of the suite it fails there, the quirk of SURVEY.md §8.1, cite)
TABLE = {
    "MUT_Q1_FX":       ("sparse:main", "pose", "Q1", "src/Sprase_ImageAlign.cpp:70,160"),
    "MUT_Q3_TIGHT":    ("sparse:main", "n_tracked", "Q3", "src/Sprase_ImageAlign.cpp:245,262"),
    "MUT_Q3_ZTEST":    ("sparse:behind", "n_tracked", "Q3", "src/Sprase_ImageAlign.cpp:254-262"),
    "MUT_Q3_INITIAL":  ("sparse:main", "n_tracked", "Q3", "src/Sprase_ImageAlign.cpp:86"),
    "MUT_Q3_ZERO":     ("sparse:main", "n_tracked", "Q3", "src/Sprase_ImageAlign.cpp:93-100"),
    "MUT_Q4_SHIFT":    ("sparse:main", "iters", "Q4", "src/Sprase_ImageAlign.cpp:143,278"),
    "MUT_Q5_RAWGRAD":  ("sparse:main", "iters", "Q5", "src/Sprase_ImageAlign.cpp:150-158"),
    "MUT_Q6_TREF":     ("sparse:main", "iters", "Q6", "src/Sprase_ImageAlign.cpp:117-119"),
    "MUT_Q8_LEFT":     ("sparse:main", "pose", "Q8", "src/Sprase_ImageAlign.cpp:335"),
    "MUT_Q9_GE":       ("sparse:dark", "iters", "Q9", "src/Sprase_ImageAlign.cpp:328"),
    "MUT_Q9_SUM":      ("sparse:main", "chi2", "Q9", "src/Sprase_ImageAlign.cpp:298"),
    "MUT_Q9_EPS":      ("sparse:main", "iters", "Q9", "src/Sprase_ImageAlign.cpp:305,341"),
    "MUT_Q10_COARSE":  ("sparse:main", "n_tracked", "Q10", "src/Sprase_ImageAlign.cpp:59"),
    "MUT_Q10_MINFTS":  ("sparse:minfts", "n_tracked", "Q10", "src/Sprase_ImageAlign.cpp:34"),
    "MUT_Q11_ZERO":    ("sparse:away", "chi2", "Q11", "src/Sprase_ImageAlign.cpp:298"),
    "MUT_A2_CHECK":    ("align2d", "px", "A2", "src/Feature_alignment.cpp:345,367-369"),
    "MUT_A1_DOUBLE":   ("align2d", "px", "A1", "src/Feature_alignment.cpp:330-398"),
    "MUT_A1_ROWSUMS":  ("align2d", "px", "A1", "src/Feature_alignment.cpp:386-392"),
    "MUT_A3_STRICT":   ("align2d", "converged", "A3", "src/Feature_alignment.cpp:367-368"),
    "MUT_A4_NOWRITE":  ("align2d", "px", "A4", "src/Feature_alignment.cpp:414"),
    "MUT_A4_NOMEAN":   ("align2d", "px", "A4", "src/Feature_alignment.cpp:386"),
    "MUT_W1_FLOATDIV": ("warp", "patch_border bytes", "W1", "src/Feature_alignment.cpp:231"),
    "MUT_W2_ROUND":    ("warp", "patch_border bytes", "W2", "src/Feature_alignment.cpp:254"),
    "MUT_W2_REFLEVEL": ("warp", "patch_border bytes", "W2", "src/Feature_alignment.cpp:215-216"),
    "MUT_A13_DET":     ("warp", "search_level", "a13", "src/Feature_alignment.cpp:198"),
    "MUT_P1_JSCALE":   ("pose_opt", "iterations", "f3", "include/Optimizer.h:162,176-189"),
    "MUT_P2_NOLOSS":   ("pose_opt", "pose", "f3", "src/Optimizer.cpp:33"),
    "MUT_P3_PLUS":     ("pose_opt", "iterations", "f3", "include/Optimizer.h:222-236"),
    "MUT_P4_ALLFEAT":  ("pose_opt", "n_residual_blocks", "f3", "src/Optimizer.cpp:47-65"),
    "MUT_P5_PIXELS":   ("pose_opt", "pose", "f3", "include/Optimizer.h:160"),
    "MUT_P6_ITERS":    ("pose_opt", "iterations", "f3", "src/Optimizer.cpp:70"),
    "MUT_PD_ROUND":    ("detector", "pyramid bytes", "a9", "src/Frame.cpp:79 (cv::pyrDown)"),
    "MUT_PD_BORDER":   ("detector", "pyramid bytes", "a9", "src/Frame.cpp:79 (cv::pyrDown)"),
    "MUT_D_NMS_TIE":   ("detector", "fast score map / survivors", "f4", "Thirdparty/fast/src/nonmax_3x3.cpp:47-106"),
    "MUT_D_SCORE":     ("detector", "fast score map / survivors", "f4", "Thirdparty/fast/src/fast_10_score.cpp"),
    "MUT_D_CELLMAX":   ("detector", "cell x", "f4", "src/Feature_detection.cpp:104"),
    "MUT_D_BOX":       ("detector", "cell score", "f4", "src/Feature_detection.cpp:173-185"),
    "S1_SHUFFLE":      ("search:std", "match count", "S1", "src/Feature_alignment.cpp:38-43,75"),
    "S1_NOCAP":        ("search:dense", "match count", "S1", "src/Feature_alignment.cpp:80"),
    "S1_NOSORT":       ("search:std", "match count", "S1", "src/Feature_alignment.cpp:88,123-126"),
    "S1_NOMASK":       ("search:std", "match count", "S1", "src/Feature_alignment.cpp:96"),
    "S1_NOBAD":        ("search:std", "match count", "S1", "src/Feature_alignment.cpp:93"),
    "S1_ALLCANDS":     ("search:dense", "cells", "S1", "src/Feature_alignment.cpp:115"),
    "W3_BORDER":       ("search:dense_border", "cells", "W3", "src/Feature_alignment.cpp:138-140"),
    "W3_FIRSTOBS":     ("search:std", "match count", "W3", "src/MapPoint.cpp:148-171"),
}

ALL_MUTANTS = list(TABLE)


def _oracle_run(sc, *prm, min_fts=15, T_seed=None):
    return oracle_lib.sparse_align(sc, *prm, min_fts=min_fts, T_seed=T_seed)


_FIXTURES = {}


def fixtures():
    if not _FIXTURES:
        _FIXTURES.update(sparse=Q.sparse_cases(), align2d=Q.align2d_cases(), warp=Q.warp_cases(), pose_opt=Q.pose_problems())
    return _FIXTURES


def domain(mutant):
    return TABLE[mutant][0].split(":")[0] if mutant else None


def cpu_outputs(mutant=None, domains=("sparse", "align2d", "warp", "search", "pose_opt", "detector"), search_worlds=None):
    """{case: outputs} from the CPU restatement with `mutant` switched (None: the faithful one), for the given domains."""
    fx = fixtures()
    c_mut = mutant if (mutant and mutant.startswith("MUT_")) else None
    s_mut = mutant if (mutant and not mutant.startswith("MUT_")) else None
    out = {}
    with (oracle_lib.mutant(c_mut) if c_mut else contextlib.nullcontext()):
        if "sparse" in domains:
            for name, case in fx["sparse"].items():
                out["sparse:" + name] = Q.sparse_outputs(_oracle_run, case)
        if "align2d" in domains:
            a = fx["align2d"]
            out["align2d"] = oracle_lib.align2d_batch(a["pyr"], a["patch_border"], a["patch"], a["level"], a["px0"], 10)
        if "warp" in domains:
            out["warp"] = Q.warp_outputs(oracle_lib.warp_patches, fx["warp"])
        if "detector" in domains:
            out["detector"] = Q.detector_outputs(oracle_lib.pyrdown, oracle_lib.fast10_list, oracle_lib.detect_cells)
        if "pose_opt" in domains:
            out["pose_opt"] = [oracle_lib.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed, linear_solver=0)
                               for P in fx["pose_opt"]]
        if "search" in domains:
            for name in (search_worlds or Q.SEARCH_WORLDS):
                out["search:" + name] = Q.search_restated(name, mutant=s_mut)
    return out


def first_difference(case, a, b):
    if case.startswith("sparse:"):
        return Q.sparse_first_difference(a, b)
    if case == "align2d":
        return Q.align2d_first_difference(a, b)
    if case == "warp":
        return Q.warp_first_difference(a, b)
    if case == "pose_opt":
        return Q.pose_first_difference(a, b)
    if case == "detector":
        return Q.detector_first_difference(a, b)
    return Q.search_first_difference(a, b)


def first_differences(a, b):
    return {case: first_difference(case, a[case], b[case]) for case in a if case in b}
