#!/usr/bin/env python3
"""Which closed-form input of the 256 x 256 training-parity case (tests/golden/cases.py: t256_md64) keeps every discontinuity of the graph
(ReLUs, the two hard picks) on the float64 oracle's side with margin: runs the test body for a few seeds and prints the median / worst
relative gradient difference.  Test tooling (imports tests/, oracle/).  usage: python tools/pick_train_seed.py [seed ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import semstereo_amd as sa  # noqa: E402
from golden import cases  # noqa: E402
import test_parity_gpu as T  # noqa: E402

for seed in [int(a) for a in sys.argv[1:]] or [848, 852, 856, 860, 864, 868, 872, 876]:
    cases._SEGMENT_SEED["t256_md64"] = seed
    try:
        T.test_hot_segment_training_step_runs_on_the_hip_stack(sa, "t256_md64")
        verdict = "ok"
    except AssertionError as e:
        verdict = "FAIL " + str(e)[:120].replace("\n", " ")
    print(seed, "median %.2e worst %.2e same_picks %s other-candidate pixels %s %s" % (
        T.REPORT.get("segment_train/median_relative_grad_diff_vs_oracle_f64", float("nan")), T.REPORT.get("segment_train/worst_relative_grad_diff_vs_oracle_f64", float("nan")),
        T.REPORT.get("segment_train/same_picks_as_the_oracle"), T.REPORT.get("segment_train/t256_md64/pixels_with_other_candidates"), verdict), flush=True)
