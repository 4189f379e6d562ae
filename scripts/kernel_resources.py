"""Prints VGPR / spill / scratch / occupancy per kernel of a .hip file (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "--offload-device-only", "-S", "-o", "/tmp/_kr.s", src,
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:70]
        print(f"{name:70s} VGPR {v.get('VGPRs',0):3d} AGPR {v.get('AGPRs',0):3d} spill {v.get('VGPRs Spill',0):3d} "
              f"scratch {v.get('ScratchSize',0):4d} occ {v.get('Occupancy',0)} LDS {v.get('LDS Size',0)}")
