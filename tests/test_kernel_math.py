"""The few-flake kernel's arithmetic (csrc/sg_few.h, k_power_few) against the general per-lane path (csrc/sg_beam.h) on the host.

Both are the device functions the kernels run, compiled for the host by hipcc (--cuda-host-only): 3 x 300 000 random beams with one
to three flakes -- wrapped wedges, shared endpoints, identical intervals, equal ranges, overlapping windows, near-equal amplitudes --
must give the same number of scatterers, the same maximum (bit for bit) and the same first-maximum bin.  No oracle, no GPU."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
def test_few_flake_path_equals_the_general_path_bit_for_bit(tmp_path):
    exe = tmp_path / "few_vs_general"
    src = ROOT / "tests" / "host_harness" / "few_vs_general.cpp"
    cmd = [HIPCC, "--cuda-host-only", "-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off", "-w",
           "-I", str(ROOT / "lidar_snow_sim_amd" / "csrc"), "-I", str(ROOT / "include"), str(src), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), "300000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("few<")]
    assert len(lines) == 3 and all(" 0 mismatches" in ln for ln in lines), r.stdout


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
def test_device_occlusion_dict_equals_the_oracle_bit_for_bit(tmp_path):
    """sg_beam_dict (csrc/sg_beam.h: phase 2 of every per-beam kernel), compiled for the host, against the oracle's
    compute_occlusion_dict (oracle/snow_oracle.c, pinned to the reference's L2 golden vectors) on random interval lists for the list
    capacities 4, 8, 16 and 63: same entries, same ranges, same ratios to the last bit."""
    so = tmp_path / "snow_oracle.o"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-c", str(ROOT / "oracle" / "snow_oracle.c"), "-o", str(so)])
    obj, exe = tmp_path / "dict_vs_oracle.o", tmp_path / "dict_vs_oracle"
    r = subprocess.run([HIPCC, "--cuda-host-only", "-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off", "-w",
                        "-I", str(ROOT / "lidar_snow_sim_amd" / "csrc"), "-I", str(ROOT / "include"),
                        str(ROOT / "tests" / "host_harness" / "dict_vs_oracle.cpp"), "-c", "-o", str(obj)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    subprocess.check_call([HIPCC, str(obj), str(so), "-o", str(exe), "-lm", "-lpthread"])
    r = subprocess.run([str(exe), "60000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("dict<")]
    assert len(lines) == 4 and all(" 0 mismatches" in ln for ln in lines), r.stdout


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
def test_device_per_beam_chain_equals_the_oracle_byte_for_byte(tmp_path):
    """The whole per-beam chain of the kernels on the host: a random table filed by the product's host filing (csrc/sg_table_host.h),
    random float32 and float64 beams through sg_beam (geometry, candidate scan, occlusion dict, amplitudes), sg_lane_power (received power with the
    exact pruning, first maximum), sg_beam_decide and sg_scatter_scale -- against the oracle's process_single_channel
    (oracle/snow_oracle.c, pinned to the reference's golden vectors): identical output rows and intensity-difference sums, in the
    kernels' default arithmetic (own sine / tangent polynomials, computed range grid) and in the exact-math mode (libm)."""
    so = tmp_path / "snow_oracle.o"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-c", str(ROOT / "oracle" / "snow_oracle.c"), "-o", str(so)])
    obj, exe = tmp_path / "beam_vs_oracle.o", tmp_path / "beam_vs_oracle"
    r = subprocess.run([HIPCC, "--cuda-host-only", "-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off", "-w",
                        "-I", str(ROOT / "lidar_snow_sim_amd" / "csrc"), "-I", str(ROOT / "include"),
                        str(ROOT / "tests" / "host_harness" / "beam_vs_oracle.cpp"), "-c", "-o", str(obj)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    subprocess.check_call([HIPCC, str(obj), str(so), "-o", str(exe), "-lm", "-lpthread"])
    r = subprocess.run([str(exe), "10000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("beams<")]
    assert len(lines) == 5 and all(" 0 mismatches" in ln for ln in lines), r.stdout


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
def test_fast_limit_ray_test_decides_as_the_reference_expression(tmp_path):
    """sg_near_ray (csrc/sg_beam.h: |y cos - x sin| < r, the reference's tangent / root / quotient only inside a band of
    1e-12 (|x| + |y|) around equality) against the reference's expression (geometry.py:94-106, :131-135) on 3 x 1.6 million flakes placed
    at r (1 +- eps) from a limit ray, eps down to 1e-17, rays at and next to the quadrant boundaries included: no decision differs,
    the two distances agree to 1e-14 (|x| + |y|) (measured: 7e-16), and on ordinary input the band decides fewer than 1e-5 of the tests."""
    exe = tmp_path / "near_ray"
    src = ROOT / "tests" / "host_harness" / "near_ray_vs_reference.cpp"
    cmd = [HIPCC, "--cuda-host-only", "-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off", "-w",
           "-I", str(ROOT / "lidar_snow_sim_amd" / "csrc"), "-I", str(ROOT / "include"), str(src), "-o", str(exe), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), "100000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("near<")]
    assert len(lines) == 3 and all(" 0 mismatches" in ln for ln in lines), r.stdout


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
def test_wave_scan_with_one_word_lists_equals_the_per_lane_scan(tmp_path):
    """sg_wave_scan (csrc/sg_beam.h: the candidate scan of the pass over all rows -- counts per bin through the coarse range index and the search,
    pair numbering, the records' first half for the decision and the second for the list, ONE word per listed flake and its resolution, the sort
    by (range, scan order), counting on beyond a full list, deferred distance tests) run as a wave of one lane on the host, against sg_beam_scan
    (one beam per lane, held to the oracle by the test above): 175 000 random float32 / float64 beams, 3 and 30 mrad wide, tables with and without the
    coarse index -- the same number of intersecting flakes and the same lists, bit for bit."""
    exe = tmp_path / "wave_vs_lane"
    src = ROOT / "tests" / "host_harness" / "wave_vs_lane.cpp"
    cmd = [HIPCC, "--cuda-host-only", "-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off", "-w",
           "-I", str(ROOT / "lidar_snow_sim_amd" / "csrc"), "-I", str(ROOT / "include"), str(src), "-o", str(exe), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), "50000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("wave<")]
    assert len(lines) == 5 and all(" 0 mismatches" in ln for ln in lines), r.stdout
