#!/usr/bin/env python3
"""Control flow per region of one kernel of a `-DODK_MARK` listing: branches, if / else pairs (`s_andn2_saveexec` / `s_or_saveexec`: the "else" half of a
structured if), exec-mask saves, per region between the phase / loop markers.
    hipcc ... $(ENGINE_FLAGS) -DODK_DEV_HF -DODK_MARK --cuda-device-only -S -o mark.s odk_engine.hip
    python tools/isa_branch_census.py mark.s [kernel-name substring ...]
Why: nested `a ? x : (b ? y : z)` chains and `p || (q && r)` conditions can come out of the compiler as basic blocks with exec-mask bookkeeping -- ~40
instructions and four branches for one five-way select (round 6: two thirds of the rough-terrain kernel's gain was rewriting such lines as one select per
statement, packed-table shifts or bitwise compares).  A region whose if / else count is not explained by its data-dependent branches is the place to look."""
import re
import sys


def main():
    path, keys = sys.argv[1], sys.argv[2:] or ["step_kernel", "ELi32ELi1EEv5KArgs"]
    name, rows, cur = None, [], None
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            name = m.group(1) if all(k in m.group(1) for k in keys) else None
            if name:
                print(name[:150]); cur = ["prologue", 0, 0, 0, 0]; rows = [cur]
            continue
        if name is None:
            continue
        if re.match(r"^\.Lfunc_end", line):
            break
        m = re.search(r";\s*(ODK_PHASE_\w+(?: \d+)?|HF_LOOP_\w+|SAT_MARK \d+)\s*$", line)
        if m:
            cur = [m.group(1), 0, 0, 0, 0]; rows.append(cur); continue
        t = line.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        cur[4] += 1
        if t[0].startswith("s_cbranch") or t[0] == "s_branch":
            cur[1] += 1
        elif t[0].startswith(("s_andn2_saveexec", "s_or_saveexec")):
            cur[2] += 1
        elif t[0].startswith("s_and_saveexec"):
            cur[3] += 1
    print(f"{'region (from this marker on)':32s} {'instructions':>12s} {'branches':>9s} {'if/else pairs':>14s} {'exec saves':>11s}")
    for r in rows:
        print(f"{r[0]:32s} {r[4]:12d} {r[1]:9d} {r[2]:14d} {r[3]:11d}")


if __name__ == "__main__":
    main()
