#!/usr/bin/env python3
"""Instruction census of one kernel of an ISA listing, by region and by class (VERDICT r5 #2a: where do the rough-terrain kernel's moves,
compares, selects and scalar instructions come from?).

    hipcc ... -DODK_MARK --cuda-device-only -S -o mark.s odk_engine.hip
    python tools/isa_census.py mark.s [kernel-substring ...]          (default: step_kernel<ShapeB, 32, 1> -- the height-field instantiation)

Regions are cut at the markers the -DODK_MARK build leaves in the listing (`; ODK_PHASE_END n` between the phases of forward_env, `; HF_LOOP_BEGIN /
END` around one iteration of the height-field pair loop, `; SAT_MARK n` inside sat_prism_row).  Classes: float arithmetic (fma / mul / add / min /
max / rcp / rsq / sqrt ...), integer arithmetic, moves + selects (v_mov, v_cndmask, v_readlane / readfirstlane / writelane, DPP moves, permlane,
bpermute is LDS), compares (v_cmp*), scalar ALU, LDS, global / scalar loads, branches + waits.  Static counts: inner loops count once."""
import collections
import re
import sys


def classify(op, line):
    if op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_setpc", "s_swappc", "s_getpc", "s_sleep", "s_setprio")):
        return "branch_wait"
    if op.startswith("s_load") or op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "mem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if op.startswith(("v_mov", "v_cndmask", "v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_swap", "v_accvgpr")):
        return "move_select"
    if re.match(r"v_(fma|fmac|mac|mad|mul|add|sub|subrev|min|max|rcp|rsq|sqrt|exp|log|sin|cos|floor|ceil|trunc|rndne|fract|ldexp|frexp|div|med3|cvt|pk_fma|pk_mul|pk_add|dot)\w*_f(32|16|64)", op) or op.startswith("v_cvt_"):
        return "float"
    if op.startswith("v_"):
        return "int"
    return "other"


def main():
    path = sys.argv[1]
    keys = sys.argv[2:] or ["step_kernel", "ILi31E", "Li32ELi1E", "ELb0ELin1ELb0E"]
    lines = open(path, errors="replace").read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l) and all(k in l.split(":")[0] for k in keys))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    print(lines[start][:160])
    region = "prologue"
    stats = collections.OrderedDict()
    dpp = collections.Counter()
    for l in lines[start:end]:
        m = re.search(r"; (ODK_PHASE_(?:END|BEGIN)\s*\d*|HF_LOOP_(?:BEGIN|END)|SAT_MARK \d+)", l)
        if m:
            region = "after " + m.group(1).replace("ODK_PHASE_", "phase ").replace("HF_LOOP_", "hf loop ").replace("SAT_MARK", "sat mark")
            continue
        if not l.startswith("\t") or l.startswith("\t.") or l.startswith("\t;"):
            continue
        op = l.split()[0]
        c = classify(op, l)
        d = stats.setdefault(region, collections.Counter())
        d[c] += 1; d["all"] += 1
        if "dpp" in l or "row_" in l or "quad_perm" in l:
            d["dpp"] += 1
    cols = ["all", "float", "int", "move_select", "cmp", "salu", "lds", "mem", "branch_wait", "dpp"]
    print(f"{'region':28s} " + " ".join(f"{c:>11s}" for c in cols))
    tot = collections.Counter()
    for k, d in stats.items():
        if d["all"] < 40:
            tot.update(d); continue
        print(f"{k:28s} " + " ".join(f"{d[c]:11d}" for c in cols))
        tot.update(d)
    print(f"{'total (static)':28s} " + " ".join(f"{tot[c]:11d}" for c in cols))


if __name__ == "__main__":
    main()
