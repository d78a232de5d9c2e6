"""Static per-phase instruction statistics of the fused step kernel (spill hunting):
    python tools/isa_phase_stats.py [A|B]
Compiles csrc/odk_engine.hip with -DODK_MARK (phase-end comments in the ISA) and counts, between consecutive
markers of step_kernel<Shape, 32>, VALU / LDS / scratch (spill) / global instructions."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shape = sys.argv[1] if len(sys.argv) > 1 else "A"
key = "ILi21E" if shape == "A" else "ILi31E"
out = os.path.join(tempfile.gettempdir(), "odk_mark.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-load-store-vectorizer=0", "-DODK_MARK", *sys.argv[2:], "-S", "--cuda-device-only", "-o", out,
                       os.path.join(ROOT, "open_duck_playground_amd/csrc/odk_engine.hip")], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z11step_kernel") and key in l and "Li32ELi0E" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
phase = "pre"
stats = collections.OrderedDict()
for l in lines[start:end]:
    m = re.search(r"; ODK_PHASE_(END|BEGIN)\s*(\d*)", l)
    if m:
        phase = "after " + (m.group(2) or "begin")
        continue
    if not l.startswith("\t") or l.startswith("\t.") or l.startswith("\t;"):
        continue
    op = l.split()[0]
    d = stats.setdefault(phase, collections.Counter())
    d["all"] += 1
    if op.startswith("scratch_store"): d["sp_st"] += 1
    elif op.startswith("scratch_load"): d["sp_ld"] += 1
    elif op.startswith("ds_"): d["lds"] += 1
    elif op.startswith("global_") or op.startswith("s_load") or op.startswith("buffer_"): d["glb"] += 1
    elif op.startswith("v_"): d["valu"] += 1
    elif op.startswith("s_"): d["salu"] += 1
print(f"{'region':12s} {'all':>6s} {'valu':>6s} {'salu':>6s} {'lds':>5s} {'glb':>5s} {'sp_ld':>6s} {'sp_st':>6s}")
for k, d in stats.items():
    print(f"{k:12s} {d['all']:6d} {d['valu']:6d} {d['salu']:6d} {d['lds']:5d} {d['glb']:5d} {d['sp_ld']:6d} {d['sp_st']:6d}")
t = collections.Counter()
for d in stats.values():
    t.update(d)
print(f"{'total':12s} {t['all']:6d} {t['valu']:6d} {t['salu']:6d} {t['lds']:5d} {t['glb']:5d} {t['sp_ld']:6d} {t['sp_st']:6d}")
