"""Looks for one miscompilation pattern in the device ISA of a .hip file: an s_cselect whose SCC comes from
an exec-mask instruction (s_*_saveexec, s_or_b64 exec, ...) with a v_cmp in between, i.e. a select on a vector
comparison whose result never reached SCC.  amdclang 22 (ROCm 7.2) emitted this for a select between two
integer constants on an fp64 comparison of values loaded from LDS at a lane-independent address, inside a
single-thread region of the refit kernel.  Usage: python scripts/scan_scc_select.py sfm_amd/csrc/*.hip"""
import re, subprocess, sys
SCC = re.compile(r'^\s*(s_cmp|s_cmpk|s_and|s_or|s_xor|s_add|s_sub|s_lshl|s_lshr|s_ashr|s_bitcmp|s_andn2|s_orn2|s_nand|s_nor|'
                 r's_xnor|s_not|s_abs|s_min|s_max|s_bfe|s_absdiff|s_addc|s_subb|s_wqm|s_quadmask|s_bcnt|s_ff|s_flbit)')
bad = 0
for src in sys.argv[1:]:
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                          "-fhip-fp32-correctly-rounded-divide-sqrt", "--offload-device-only", "-S", "-o", "-", src],
                         capture_output=True, text=True).stdout.split("\n")
    kern = None
    for i, l in enumerate(asm):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            kern = m.group(1)
        if re.match(r'^\s*s_cselect', l):
            saw_vcmp = False
            for j in range(i - 1, max(i - 200, 0), -1):
                t = asm[j]
                if re.match(r'^\s*v_cmp', t):
                    saw_vcmp = True
                if SCC.match(t):
                    if saw_vcmp and ("saveexec" in t or re.match(r'^\s*s_\w+\s+exec', t)):
                        print(f"{src}: {kern}: line {i + 1}: {l.strip()}  (SCC from: {t.strip()})")
                        bad += 1
                    break
                if re.match(r'^_Z\w+:', t):
                    break
print("suspicious selects:", bad)
sys.exit(1 if bad else 0)
