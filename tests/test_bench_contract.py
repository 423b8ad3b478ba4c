"""bench.py's measurement contract, the parts that need no GPU: the roofline's algorithmic bytes are SURVEY.md 8(d)'s rows and
nothing else, and a counter file is reported only for the build it was recorded on."""
import importlib
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_frame_kernel_bytes_are_the_survey_rows(bench):
    cfg = types.SimpleNamespace(cols=1241, rows=376)
    stats = {"N": 2061.1, "P": 684.0, "M": 237.0, "I": 17.9, "R": 100.0, "fused": True}
    ab, extra = bench.algorithmic_bytes(cfg, 160, stats)
    N, P, M, I = stats["N"], stats["P"], stats["M"], stats["I"]
    per_frame = P * (24 + 64 + 8) + P * 32 * 2 + I * M * 64 + M * (8 + 1) + 2 * N * 32 + 96      # SURVEY.md 8(d), frame path
    assert ab["k_frame"] == pytest.approx(per_frame * 160)
    assert ab["k_fast_box"] == 2 * 1241 * 376 * 160                                               # each image byte once
    assert extra["k_frame"] > 0 and "k_fast_box" not in extra                                     # recovery taps etc.: separate, never in frac
    assert extra["k_frame_hbm_only"] == pytest.approx((per_frame - I * M * 64) * 160)            # 8(d)'s second figure: without the aligner re-reads
    ab2, extra2 = bench.algorithmic_bytes(cfg, 160, dict(stats, fused=False))
    assert ab2["k_frame"] == ab["k_frame"] and set(extra2) == {"k_frame_hbm_only"}


def test_counter_file_is_tied_to_the_build(bench, monkeypatch, tmp_path):
    from vslam_pose_estimation_framework_amd import buildinfo
    here = buildinfo.source_sha16()
    assert len(here) == 16 and here == buildinfo.source_sha16()
    prof = tmp_path / "profiles"
    prof.mkdir()
    rec = {"streams": 160, "source_sha16": here, "per_launch_KB": {"k_frame": {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 24.0}}}
    (prof / bench.PMC_SUMMARY).write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("k_frame", 160) == (1024 * 1024, None)
    t, note = bench.pmc_traffic("k_frame", 144)                       # another stream count
    assert t is None and "144" in note
    t, note = bench.pmc_traffic("k_brief", 160)                       # a kernel the file does not hold
    assert t is None and "k_brief" in note
    rec["source_sha16"] = "0" * 16                                    # recorded on another build
    (prof / bench.PMC_SUMMARY).write_text(json.dumps(rec))
    t, note = bench.pmc_traffic("k_frame", 160)
    assert t is None and "recorded on source" in note and here in note
    (prof / bench.PMC_SUMMARY).unlink()
    t, note = bench.pmc_traffic("k_frame", 160)
    assert t is None and "no counter file" in note


def test_source_hash_follows_the_sources(tmp_path, monkeypatch):
    from vslam_pose_estimation_framework_amd import buildinfo
    files = buildinfo.source_files()
    assert any(f.endswith("vslam_hip.hip") for f in files) and any(f.endswith("vslam_hip.h") for f in files)
    a = buildinfo.source_sha16()
    extra = tmp_path / "x.h"
    extra.write_text("// changed")
    monkeypatch.setattr(buildinfo, "source_files", lambda: files + [str(extra)])
    assert buildinfo.source_sha16() != a


def test_sq_utilisation_formula_and_build_tie(bench, monkeypatch, tmp_path):
    from vslam_pose_estimation_framework_amd import buildinfo
    prof = tmp_path / "profiles"
    prof.mkdir()
    # k_fast_box of profiles/r03l_sq_counters.json: 142.7 M quad-cycles of VALU issue, 5.18 M GRBM cycles summed over 8 XCDs
    rec = {"source_sha16": buildinfo.source_sha16(), "per_kernel": {"k_fast_box": {"SQ_ACTIVE_INST_VALU": 142686400.2, "SQ_INSTS_SALU": 86842098.2,
                                                                                  "SQ_ACTIVE_INST_LDS": 24865027.2, "GRBM_GUI_ACTIVE": 5177150.3}}}
    (prof / bench.SQ_SUMMARY).write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    util, note = bench.sq_utilisation()
    assert note is None
    # 4 x 142.7 M busy SIMD-cycles over (5.18 M / 8) x 1024 available: 86 %
    assert util["k_fast_box"]["valu_util"] == pytest.approx(4 * 142686400.2 / (5177150.3 / 8 * 1024), rel=1e-3)
    assert 0.85 < util["k_fast_box"]["valu_util"] < 0.87 and 0.5 < util["k_fast_box"]["salu_util"] < 0.55
    rec["source_sha16"] = "0" * 16
    (prof / bench.SQ_SUMMARY).write_text(json.dumps(rec))
    util, note = bench.sq_utilisation()
    assert util == {} and "recorded on source" in note
