#!/usr/bin/env python3
"""Lint of the comb walker's ISA for the one hazard the compiler cannot see (dev tool + CPU test, no GPU needed):

  a VGPR written by a VALU instruction may not be read as the DPP source (src0 of a *_dpp instruction) for two wait
  states (gfx9 / CDNA data hazard "VALU writes VGPR -> DPP reads that VGPR").

The DPP taps of smx_agg_v5.hip are inline assembly (box_bottom / box_top), and LLVM's hazard recogniser does not look
inside inline assembly: box_bottom brings its own `s_nop 1`, box_top relies on its DPP source being a ring slot written
rows earlier.  A different register allocation (compiler bump, another -D variant) could put a v_mov / VALU write of
that source right in front of the asm; results would then be silently stale.  This scan fails on any such site.

  python tools/dpp_hazard_check.py [extra -D flags ...]      exit status 1 + a list of sites on a finding
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def regs(tok):
    m = REG.match(tok.strip().strip(","))
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def wait_states(ins):
    op = ins.split()[0]
    if op == "s_nop":
        return int(ins.split()[1], 0) + 1
    return 1


def check(asm_lines):
    findings = []
    window = []       # (wait states this instruction provides, vgprs it writes if VALU, text, line)
    for no, raw in enumerate(asm_lines, 1):
        line = raw.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        op = line.split()[0]
        toks = line[len(op):].split(",")
        if "_dpp" in op or " row_shr" in line or " wave_shr" in line or " quad_perm" in line or " row_bcast" in line:
            src0 = regs(toks[1].split()[0]) if len(toks) > 1 else set()
            ws = 0
            for w_states, written, text, lno in reversed(window):
                if ws >= 2:
                    break
                if written & src0:
                    findings.append((no, line, lno, text))
                    break
                ws += w_states
        written = set()
        if op.startswith("v_") and not op.startswith(("v_cmp", "v_cmpx")) and toks:
            written = regs(toks[0].split()[0])
        window.append((wait_states(line), written, line, no))
        if len(window) > 8:
            window.pop(0)
    return findings


def main(extra):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "v5.s")
        subprocess.check_call([os.path.join(ROOT, "tools", "v5_asm.sh"), out, *extra])
        lines = open(out).read().split("\n")
    ndpp = sum(1 for l in lines if "_dpp" in l.split(";")[0])
    f = check(lines)
    for no, line, lno, text in f:
        print(f"HAZARD line {no}: `{line}` reads a DPP source written {no - lno} lines earlier by `{text}` (line {lno})")
    print(f"{ndpp} DPP instructions scanned, {len(f)} hazards")
    return 1 if f or ndpp == 0 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
