#!/usr/bin/env python3
"""Are two builds' instruction streams the same, function by function?

    hipcc ... --cuda-device-only -S -o old.s odk_engine.hip      (before a change; `make -C open_duck_playground_amd/csrc engine.s`)
    hipcc ... --cuda-device-only -S -o new.s odk_engine.hip      (after)
    python tools/isa_compare.py old.s new.s [name-filter ...]

For every function (kernel or out-of-line device function) present in both listings: the instruction lines with comments, labels and
blank lines dropped and local labels renumbered in order of appearance.  Prints IDENTICAL / the number of differing lines (and, for small
differences, a unified diff) plus each kernel's resource footer (VGPRs, SGPRs, scratch, LDS, occupancy).  Used to show that a change meant
for one model shape leaves the other shapes' kernels the instruction streams they were (round 6: the NU-generic env logic vs the duck's
kernels)."""
import difflib
import re
import sys


def functions(path):
    out, cur, name = {}, None, None
    meta = {}
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z[\w$.]+):\s*(;.*)?$", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or re.match(r"^\.Lfunc_end\d+:", line):
            cur = None
            continue
        m = re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte|NumAgprs): (\S+)", line)
        if m:
            meta.setdefault(name, {})[m.group(1)] = m.group(2)
        cur.append(line)
    # resource lines follow .Lfunc_end: second pass keyed by the last function seen
    last = None
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z[\w$.]+):\s*(;.*)?$", line)
        if m:
            last = m.group(1)
        m = re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte|NumAgprs): (\S+)", line)
        if m and last:
            meta.setdefault(last, {})[m.group(1)] = m.group(2)
    return out, meta


def normalise(lines):
    labels = {}
    res = []
    for l in lines:
        l = l.split(";")[0].rstrip()
        if not l.strip() or l.strip().startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", l.strip()):
                labels.setdefault(l.strip()[:-1], f"L{len(labels)}")
                res.append(labels[l.strip()[:-1]] + ":")
            continue
        res.append(l.strip())
    txt = "\n".join(res)
    for k in sorted(labels, key=len, reverse=True):
        txt = txt.replace(k, labels[k])
    # forward references to labels not yet defined when first seen are handled by the global replace above; leftovers (other functions'
    # labels) are normalised by number only
    txt = re.sub(r"\.LBB\d+_(\d+)", r"LX\1", txt)
    return txt.split("\n")


def short(name):
    m = re.search(r"(reset_kernel|step_kernel|physics_kernel)IN3odk5ShapeI([^E]*)EE?ELi(\d+)ELi(\d+)", name)
    if m:
        dims = m.group(2).replace("ELi", ",").replace("Li", "").replace("ELb", ",b").replace("n1", "-1")
        return f"{m.group(1)}<Shape<{dims}>,{m.group(3)},{m.group(4)}>"
    return name[:100]


def main():
    a, am = functions(sys.argv[1])
    b, bm = functions(sys.argv[2])
    filt = sys.argv[3:]
    same = diff = 0
    for name in sorted(set(a) & set(b)):
        if filt and not any(f in name or f in short(name) for f in filt):
            continue
        na, nb = normalise(a[name]), normalise(b[name])
        res = " ".join(f"{k}={v}" for k, v in sorted(bm.get(name, {}).items()))
        if na == nb:
            same += 1
            print(f"IDENTICAL  {len(na):6d} lines  {short(name)}  [{res}]")
        else:
            diff += 1
            d = [l for l in difflib.unified_diff(na, nb, lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---")]
            print(f"DIFFERENT  {len(na):6d} -> {len(nb):6d} lines, {len(d)} changed  {short(name)}  [{res}] (was: {' '.join(f'{k}={v}' for k, v in sorted(am.get(name, {}).items()))})")
            if len(d) <= 40:
                for l in d:
                    print("    " + l)
    only_a, only_b = sorted(set(a) - set(b)), sorted(set(b) - set(a))
    for n in only_a:
        if not filt or any(f in n or f in short(n) for f in filt):
            print(f"GONE       {short(n)}")
    for n in only_b:
        if not filt or any(f in n or f in short(n) for f in filt):
            res = " ".join(f"{k}={v}" for k, v in sorted(bm.get(n, {}).items()))
            print(f"NEW        {len(normalise(b[n])):6d} lines  {short(n)}  [{res}]")
    print(f"{same} identical, {diff} different, {len(only_a)} gone, {len(only_b)} new")
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
