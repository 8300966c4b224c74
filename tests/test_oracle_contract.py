"""(CPU) How far a stock GCC build of the reference (g++ -O3 -march=native: -ffp-contract=fast) can move from the parity oracle
(-ffp-contract=off, like the HIP kernels): tools/fma_sensitivity.py runs the restated path compiled both ways on the same VGA inputs.
The numbers go to DESIGN.md section 2 / profiles/r03_fma_sensitivity.md; this test keeps the tool alive and pins the qualitative result:
integer stages are untouched, float stages move by rounding only (no systematic divergence)."""
import importlib.util
import os
import subprocess
import numpy as np
from conftest import ROOT


def _tool():
    spec = importlib.util.spec_from_file_location("fma_sensitivity", os.path.join(ROOT, "tools", "fma_sensitivity.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_contract_variant_really_contracts():
    from oracle.pyoracle import build
    dis = subprocess.run(["objdump", "-d", build(variant="contract")], capture_output=True, text=True).stdout
    if "fma" not in open("/proc/cpuinfo").read():
        import pytest
        pytest.skip("host CPU has no FMA: -march=native cannot contract")
    assert dis.count("vfmadd") + dis.count("vfmsub") + dis.count("vfnmadd") > 100
    assert subprocess.run(["objdump", "-d", build()], capture_output=True, text=True).stdout.count("vfmadd") == 0


def test_fma_contraction_moves_float_stages_by_rounding_only():
    m = _tool()
    R = m.measure(3)
    print(m.table(R))
    assert R["frames"] == 3 and R["keypoints"] > 2000
    # FAST / NMS / pyramid / Hamming are integer; the grid selection compares float Shi-Tomasi scores computed from exact integer sums
    assert R["keypoints_differ"] <= 0.002 * R["keypoints"]
    assert R["angles_max_abs_diff_deg"] < 1e-3
    assert R["descriptor_bits_flipped"] <= 1e-3 * R["descriptor_bits_total"]
    assert R["match_idx_differ"] <= 0.01 * R["match_pairs"]
    assert R["klt_status_differ"] <= 0.01 * R["klt_points"] and R["klt_track_max_abs_diff_px"] < 0.05
    assert R["fdp_flag_differ"] <= 0.01 * R["fdp_candidates"] and R["fdp_px_max_abs_diff"] < 0.05
    assert R["sparse_align_pose_max_abs_diff"] < 1e-6
    assert max(R["ba_err_max_rel_diff"], R["ba_Hpp_max_rel_diff"], R["ba_Hpl_max_rel_diff"]) < 1e-12       # far inside the 1e-5 bar of north_star
    assert R["g2o_lm_10x2000_chi2_final_rel_diff"] < 1e-9
