"""Print registers / scratch / occupancy of every gfx950 kernel in one csrc file (hipcc resource-usage remarks).

usage: python scripts/kernel_resources.py [snowgpu_kernels.hip] [filter]
"""
import re, subprocess, sys, tempfile, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "snowgpu_kernels.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-I" + root + "/include", "-I" + root + "/lidar_snow_sim_amd/csrc", "-c",
                        root + "/lidar_snow_sim_amd/csrc/" + src, "-o", td + "/k.o",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for c in rows:
    if flt in c["name"]:
        print("%-70s VGPR %3d AGPR %3d SGPR %3d scratch %4d occ %d LDS %d  vspill %d sspill %d" % (
            c["name"][:70], c.get("VGPRs", -1), c.get("AGPRs", -1), c.get("SGPRs", -1), c.get("ScratchSize", -1),
            c.get("Occupancy", -1), c.get("LDS Size", -1), c.get("VGPRs Spill", -1), c.get("SGPRs Spill", -1)))
