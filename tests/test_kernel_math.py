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
