#!/usr/bin/env python3
"""profiles/rNN_valu.json: what bench.py needs for `roofline.valu_frac` of the walker kernel -- its dynamic VALU instruction count
per launch (SQ_INSTS_VALU, tools/pmc.sh summary) and the SIMD cycles one VALU instruction of its mix occupies (tools/isa_budget.py
with the per-class rates of tools/ubench/simd_rate.hip).  valu_frac = instructions x cycles per instruction / (launch duration x
shader clock x SIMDs): the share of the chip's VALU issue capacity the launch used.  Tagged with the library version like the
traffic file: bench.py quotes it only for the build it was taken from (KITTI workload).

  python tools/valu.py profiles/r05_pmc_summary.txt profiles/r05_isa_budget.txt profiles/r05_valu.json"""
import json, re, sys
sys.path.insert(0, ".")
pmc, isa, dst = sys.argv[1:4]
insts = None
walk = False
for line in open(pmc):
    if not line.startswith(" "):
        walk = "k_v5_walk" in line
    elif walk and "SQ_INSTS_VALU " in line:
        insts = float(line.split()[2])
cyc = float(re.search(r"SIMD cycles per VALU instruction on this path: ([0-9.]+)", open(isa).read()).group(1))
import stereo_matching_cuda_amd as smx
out = {"library": smx.lib().smx_version().decode(), "workload": "kitti", "walker_valu_instructions_per_launch": insts,
       "simd_cycles_per_valu_instruction": cyc, "simds": 1024, "shader_clock_ghz": 2.4,
       "source": f"{pmc} (SQ_INSTS_VALU of k_v5_walk per dispatch), {isa} (instruction mix of the interior path x the whole-launch "
                 "rates of profiles/r05_simd_rate.txt: plain f32 3.0, packed f32 4.7, DPP / mixed / packed-half 4.4 cycles per wave64 "
                 "instruction and SIMD)"}
json.dump(out, open(dst, "w"), indent=1)
print(out)
