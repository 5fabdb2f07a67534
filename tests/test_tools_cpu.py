"""The profiling helpers that turn rocprofv3 counter files into the numbers quoted in DESIGN.md / bench.py (pure Python, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _csv(path, rows):
    with open(path, "w") as f:
        f.write('"Kernel_Name","Counter_Name","Counter_Value"\n')
        for k, c, v in rows:
            f.write('"%s","%s",%s\n' % (k, c, v))


def test_summarize_pmc_applies_the_calibrated_factors(tmp_path):
    """FETCH_SIZE (KiB) is doubled for every kernel (all of them are fed by vector loads since round 3: the trellis kernels fetch the packed soft
    stream with 16-bit vector loads); WRITE_SIZE is taken as reported (profiles/r02_k_calibration.json)."""
    f, w = str(tmp_path / "f.csv"), str(tmp_path / "w.csv")
    _csv(f, [("sora::k_frame(sora::RxArgs)", "FETCH_SIZE", 1000), ("sora::k_viterbi(sora::VitJob const*)", "FETCH_SIZE", 2000), ("other_kernel()", "FETCH_SIZE", 5)])
    _csv(w, [("sora::k_frame(sora::RxArgs)", "WRITE_SIZE", 300), ("sora::k_viterbi(sora::VitJob const*)", "WRITE_SIZE", 10)])
    out = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), f, w, "4096", "", "k_viterbi"]))   # (the trellis kernel of the call being summed)
    k = out["kernels"]
    assert k["k_frame"]["fetch_bytes"] == 1000 * 1024 * 2 and k["k_frame"]["write_bytes"] == 300 * 1024
    assert k["k_viterbi"]["fetch_bytes"] == 2000 * 1024 * 2 and k["k_viterbi"]["write_bytes"] == 10 * 1024
    assert "other_kernel" not in k
    assert out["total_hbm_bytes_per_call"] == k["k_frame"]["hbm_bytes"] + k["k_viterbi"]["hbm_bytes"]
    assert out["algorithmic_bytes_per_call"] == round(4096 * 4880 * 4.3375)


def test_summarize_calib_reports_reported_over_moved(tmp_path):
    f, w = str(tmp_path / "f.csv"), str(tmp_path / "w.csv")
    gib_kib = (1 << 30) / 1024
    _csv(f, [("calib_read16(uint4 const*, unsigned long, unsigned int*)", "FETCH_SIZE", gib_kib / 2), ("calib_read_scalar(unsigned int const*)", "FETCH_SIZE", gib_kib)])
    _csv(w, [("calib_write4(unsigned int*, unsigned long)", "WRITE_SIZE", gib_kib)])
    out = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "calib", "summarize_calib.py"), f, w]))
    r = out["reported_over_moved"]
    assert r["calib_read16"]["FETCH_SIZE"] == 0.5 and r["calib_read_scalar"]["FETCH_SIZE"] == 1.0 and r["calib_write4"]["WRITE_SIZE"] == 1.0


def test_committed_calibration_matches_the_factors_in_use():
    cal = json.load(open(os.path.join(ROOT, "profiles", "r02_k_calibration.json")))["reported_over_moved"]
    assert abs(cal["calib_read16"]["FETCH_SIZE"] - 0.5) < 0.01 and abs(cal["calib_read4"]["FETCH_SIZE"] - 0.5) < 0.01
    assert abs(cal["calib_read_scalar"]["FETCH_SIZE"] - 1.0) < 0.01
    for k in ("calib_write16", "calib_write4", "calib_write1"):
        assert abs(cal[k]["WRITE_SIZE"] - 1.0) < 0.02
