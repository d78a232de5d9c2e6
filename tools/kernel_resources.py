#!/usr/bin/env python3
"""Table of every env / physics kernel's registers, scratch and occupancy from the compiler's own remarks:
    make -C open_duck_playground_amd/csrc resource 2>&1 | python tools/kernel_resources.py > profiles/rN/kernel_resources.txt"""
import re
import sys

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
print("# make -C open_duck_playground_amd/csrc resource (hipcc -Rpass-analysis=kernel-resource-usage): kernel<Shape dims, lanes per env, floor> VGPRs AGPRs "
      "scratch(B/lane) waves/SIMD SGPRs (SGPRs kept in VGPR lanes)")
for r in rows:
    m = re.search(r"(reset_kernel|step_kernel|physics_kernel)IN3odk5ShapeI(.*?)EEELi(\d+)ELi(\d+)", r["name"])
    if not m:
        continue
    dims = m.group(2).replace("ELi", ",").replace("Li", "").replace("ELb", ",b").replace("n1", "-1")
    print(f"{m.group(1):15s} Shape<{dims}> G={m.group(3)} HF={m.group(4)}  vgpr {r.get('VGPRs', '?'):>3s} agpr {r.get('AGPRs', '?'):>3s} scratch {r.get('ScratchSize', '?'):>3s} "
          f"occupancy {r.get('Occupancy', '?')} sgpr {r.get('TotalSGPRs', '?')} (spilled to lanes: {r.get('SGPRs Spill', '?')})")
