"""bench.py's line without a GPU: the roofline object built from the committed counter files and the
fields the driver's contract names.  No compute here."""
import json
import os

import pytest

from common import ROOT
import bench


def _newest_pmc(workload):
    for tag in bench.PROFILE_TAGS:
        p = os.path.join(ROOT, "profiles", "pmc_%s_%s.json" % (tag, workload))
        if os.path.exists(p):
            return p
    return None


@pytest.mark.parametrize("kernel,workload,ms,alg", [("k_ao_rays", "c3t", 4.83, 47.8e9), ("k_ppll_raster_prism", "c4", 0.289, 0.2e9),
                                                     ("k_render_rt", "c2", 0.158, 0.7e9)])
def test_roofline_object_from_the_committed_counters(kernel, workload, ms, alg):
    assert _newest_pmc(workload), "profiles/ holds no counter file for %s" % workload
    r = bench.roofline(kernel, workload, ms, alg, 1)
    # the contract's keys
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-4)
    # achieved = counter traffic per launch / the launch time handed in, with the gfx950 correction of the guide
    pmc = json.load(open(_newest_pmc(workload)))
    c = max((v for k, v in pmc["kernels"].items() if k.startswith(kernel + "<") or k == kernel), key=lambda d: d.get("SQ_INSTS_VALU", 0.0))
    traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    assert r["traffic"] == int(traffic)
    assert r["achieved"] == pytest.approx(traffic / (ms * 1e6), rel=1e-3)
    # hardware-counter figures that need no calibration of the build's
    assert 0.0 < r["valu_busy"] < 1.25 and 1.5 < r["clock_ghz_live"] < 3.0 and 0.0 < r["lane_utilisation"] <= 1.0
    assert r["valu_busy"] == pytest.approx(4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8.0), abs=1e-3)
    assert isinstance(r["pmc_matches_build"], bool) and r["source_sha"]


def test_roofline_degrades_without_counters():
    r = bench.roofline("k_ao_rays", "no_such_workload", 4.8, 1e9, 1)
    assert r["frac"] is None and r["traffic"] is None and r["pmc_matches_build"] is False and "note" in r
    r = bench.roofline("k_ao_rays", "c3t", 4.8, 1e9, 8)     # counter-backed figures are a 1-GPU statement
    assert r["frac"] is None


def test_workload_table_names_the_baseline_configs():
    names = {k: w["name"] for k, w in bench.WORKLOADS.items()}
    assert names["c3"].startswith("C3: 1M-segment") and "64 spp" in names["c3"] and "1920x1080" in names["c3"]
    assert names["c2"].startswith("C2: 100k-segment") and names["c4"].startswith("C4: 1M-segment") and "3840x2160" in names["c5"]
