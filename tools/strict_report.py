#!/usr/bin/env python3
"""The strict full-size parity report (tests/strict.py: HIP attention branch -> reference picks restored where they differ ->
HIP matching branch, against the reference's fixture AND its float64 truth) for one library build, as one JSON line per
fixture.  A development aid for accuracy work on the conv engines: SS_TOOL_LIB=tools/_build/lib_<variant>.so selects an
experimental build (tools/build_variant.sh), SS_CONV_ENGINE the engine.
usage: tools/strict_report.py [fixture ...]      (default: f1024_md128_cal; r05: also the plain-run EPE of every fixture and the means)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from semstereo_amd import _lib  # noqa: E402

if os.environ.get("SS_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
import semstereo_amd as sa  # noqa: E402
import strict  # noqa: E402
from golden import cases  # noqa: E402

names = sys.argv[1:] or ["f1024_md128_cal"]
g = np.load(os.path.join(ROOT, "tests", "golden", "segment_full.npz"))
KEYS = ("pixels_with_other_candidates", "epe_vs_reference_px", "epe_vs_reference_fullres_px", "epe_vs_reference_off_ties_px", "max_err_off_ties_px",
        "pixels_beyond_1e-3", "hip_vs_truth_epe_off_ties_px", "reference_vs_truth_epe_off_ties_px",
        "hip_vs_truth_max_off_ties_px", "reference_vs_truth_max_off_ties_px", "hip_vs_truth_epe_px", "reference_vs_truth_epe_px")
ALL = []
for name in names:
    B, H, W, maxdisp = cases.segment_shape(name)
    seg = sa.HotSegment(maxdisp)
    seg.load_state_dict(cases.segment_params(name, g), strict=False)
    seg = seg.cuda().eval()
    rep, v, pred, differs, unexplained = strict.run_strict(seg, g, name)
    fl4, fr4, fl8, fr8, _ = cases.segment_inputs(name)
    with torch.no_grad():
        plain = seg(fl4.cuda(), fr4.cuda(), fl8.cuda(), fr8.cuda())["pred"].cpu().squeeze(1)
    out = {"fixture": name, "plain_run_epe_vs_reference_fullres_px": 4.0 * float((plain - torch.as_tensor(g[f"{name}/pred_map"])).abs().mean()), "lib": os.path.basename(_lib.LIB_PATH), "engine": sa.modules.CONV_ENGINE}
    out.update({k: rep[k] for k in KEYS})
    out["mean_ratio_to_reference"] = rep["hip_vs_truth_epe_off_ties_px"] / rep["reference_vs_truth_epe_off_ties_px"]
    out["max_ratio_to_reference"] = rep["hip_vs_truth_max_off_ties_px"] / rep["reference_vs_truth_max_off_ties_px"]
    out["unexplained"] = int(unexplained.sum())
    print(json.dumps(out), flush=True)
    ALL.append(out)
    del seg
    torch.cuda.empty_cache()
if len(ALL) > 1:
    same = [o for o in ALL if o["fixture"].startswith("f1024") and "_cal" in o["fixture"]]      # (the calibrated records; the uncalibrated one has its own line)
    if len(same) > 1:
        print(json.dumps({"mean_over": [o["fixture"] for o in same],
                          "plain_run_epe_vs_reference_fullres_px": sum(o["plain_run_epe_vs_reference_fullres_px"] for o in same) / len(same),
                          "picks_restored_epe_vs_reference_fullres_px": sum(o["epe_vs_reference_fullres_px"] for o in same) / len(same)}))
