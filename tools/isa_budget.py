#!/usr/bin/env python3
"""Per-role instruction budget of the comb walker's interior path, counted in the ISA (dev tool, runs on the CPU:
hipcc cross-compiles).

  python tools/isa_budget.py [profiles/rNN_isa_budget.txt]

Compiles stereo_matching_cuda_amd/csrc/smx_agg_v5.hip with -DSMX_V5_MARK (comments in the instruction stream around the
regions below), and counts the instructions between the first begin/end pair of each region by class.  The regions are
what ONE wave of a role executes per band (= 10 image rows x 152 output columns of a strip, 1520 cells) on the
interior path; `per 64 cells` = (instructions x waves of the role) / 1520 x 64."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "stereo_matching_cuda_amd", "csrc", "smx_agg_v5.hip")
FLAGS = ("-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math "
         "-fno-slp-vectorize -fvisibility=hidden -DSMX_V5_MARK").split()
REGIONS = [  # name, waves of the role per workgroup, what
    ("cost", 1, "cost wave: the band's 430 quads (7 per lane): 14 loads, packed-half costs, 14 tile stores"),
    ("scan", 1, "row-scan wave: sequential row prefix of stage 1 (p, I p) and stage 2 (a, b), 43 groups of 4 columns"),
    ("s1rows", 3, "stage-1 comb wave: 10 rows: column sum, box (DPP left taps), division, a_k b_k -> tile 2"),
    ("s2head", 3, "stage-2 comb wave: a/b band out of tile 2 into registers, hand-off record out"),
    ("s2rows", 3, "stage-2 comb wave: 10 rows: column sum, box, division, exactness vote, q -> HBM"),
    ("s1handin", 3, "stage-1 comb wave: left neighbour's record into tile 2 / carries"),
]
CELLS = 10 * 152


# SIMD cycles a wave64 VALU instruction occupies at four waves per SIMD, measured over whole launches with
# tools/ubench/simd_rate.hip (profiles/r05_simd_rate.txt): plain f32 VOP2 / VOP3 (add, mul, fma) 3.0 (2.6 at eight waves);
# everything packed, DPP, mixed-precision, packed-half or with source modifiers 4.3-4.8.  A v_pk_*_f32 is ONE pass of the
# pipeline, not two (round 4 assumed two from the single-wave issue rate).
def simd_cycles(ins):
    op = ins.split()[0]
    if not op.startswith("v_") or op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return 0.0
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return 4.7
    if "_dpp" in op or " row_shr" in ins or "wave_shr" in ins:
        return 4.4
    if "mix" in op or op.endswith("_f16") or "|" in ins or "sdwa" in op:
        return 4.4
    return 3.0


def classify(ins):
    op = ins.split()[0]
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "scratch_")): return "vmem"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_pk_") and op.endswith("_f32"): return "valu_pk"
    if "_dpp" in op or " row_shr" in ins or "wave_shr" in ins: return "valu_dpp"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "valu_lane"
    if op.startswith("v_"): return "valu"
    return "other"


def main(dst=None):
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(SRC),
                               "--save-temps", "-c", SRC, "-o", os.path.join(td, "x.o")], cwd=td,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(td, "smx_agg_v5-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    # the exact kernel (template argument 0); the FAST instantiation comes first in the file
    first = next(i for i, l in enumerate(asm) if l.startswith("_ZN3smx2v59k_v5_walkILi0ELi1E"))
    asm = asm[first:]
    out = []
    cols = ["valu", "valu_pk", "valu_dpp", "valu_lane", "salu", "s_nop", "s_waitcnt", "lds", "vmem"]
    out.append("k_v5_walk, interior path, instructions one wave executes per band (10 rows x 152 output columns = 1520 cells)")
    out.append("(`cycles` = SIMD cycles of the region's VALU instructions at the measured per-class rates: plain f32 3.0, packed f32 4.7,")
    out.append(" DPP / mixed / packed-half / modifiers 4.4 -- profiles/r05_simd_rate.txt; `cyc/64c` = the same per 64 cells)")
    out.append("")
    out.append(f"{'region':10s} {'waves':>5s} " + " ".join(f"{c:>9s}" for c in cols) + f" {'cycles':>7s} {'VALU/64c':>9s} {'SALU/64c':>9s} {'cyc/64c':>8s}")
    tot_v = tot_s = tot_p = 0.0
    for name, waves, what in REGIONS:
        try:
            a = next(i for i, l in enumerate(asm) if f"; MARK {name} begin" in l)
            b = next(i for i, l in enumerate(asm) if i > a and f"; MARK {name} end" in l)
        except StopIteration:
            out.append(f"{name:10s} (markers not found)")
            continue
        # basic blocks of the region; the exact-division slow path (a wave-uniform branch that is taken when a window
        # sum is tiny: never on the bench images) is not part of the budget
        blocks, cur = [], []
        for l in asm[a:b]:
            t = l.strip()
            if re.match(r"\.LBB\d+_\d+:", t):
                blocks.append(cur); cur = []
                continue
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            cur.append(t)
            if t.startswith(("s_cbranch", "s_branch")):
                blocks.append(cur); cur = []
        blocks.append(cur)
        cnt = collections.Counter()
        cyc = 0.0
        for blk in blocks:
            if any(t.startswith(("v_div_scale", "v_div_fixup")) for t in blk):
                continue
            for t in blk:
                cnt[classify(t)] += 1
                cyc += simd_cycles(t)
        valu = cnt["valu"] + cnt["valu_pk"] + cnt["valu_dpp"] + cnt["valu_lane"]
        passes = cyc
        v64 = valu * waves / CELLS * 64
        s64 = (cnt["salu"] + cnt["s_nop"] + cnt["s_waitcnt"]) * waves / CELLS * 64
        tot_v += v64; tot_s += s64; tot_p += passes * waves / CELLS * 64
        out.append(f"{name:10s} {waves:5d} " + " ".join(f"{cnt[c]:9d}" for c in cols) + f" {passes:7.0f} {v64:9.1f} {s64:9.1f} {passes * waves / CELLS * 64:8.1f}")
    out.append(f"{'sum':10s} {'':5s} " + " ".join(f"{'':9s}" for _ in cols) + f" {'':7s} {tot_v:9.1f} {tot_s:9.1f}   (SIMD cycles per 64 cells: {tot_p:.1f}; a CU has 4 SIMD-cycles per cycle)")
    out.append("")
    for name, waves, what in REGIONS:
        out.append(f"  {name:9s} {what}")
    out.append("")
    out.append("Arithmetic minimum of this decomposition (no instruction for addressing, waits, hazards or votes), VALU instructions:")
    out.append("  comb lane and row, stage 1: column sum 1 pk, box 2 dpp + 1 pk + 2 dpp, division 3 pk, a_k b_k 5          = 14")
    out.append("  comb lane and row, stage 2: column sum 1 pk, box 5, division 3 pk, q 2, guidance cvt 0.5, vote 1.5         = 13")
    out.append("  152 of the 192 comb lanes of a stage produce an output: (14 + 13) x 192/152                      = 34.1 per 64 cells")
    out.append("  cost: 7 per cell, 172 columns evaluated per 152 outputs                                         =  7.9 per 64 cells")
    out.append("  row scans: 4 columns x 43 groups per band, 40 (row, component) lanes of 64 at a time            =  7.2 per 64 cells")
    out.append("  floor                                                                                            = 49.2 per 64 cells")
    out.append(f"SIMD cycles per VALU instruction on this path: {tot_p / tot_v:.2f} (used by bench.py for roofline.valu_frac together with the")
    out.append("dynamic SQ_INSTS_VALU of the whole kernel, profiles/r05_pmc_summary.txt)")
    txt = "\n".join(out)
    print(txt)
    if dst:
        open(dst, "w").write(txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:2])
