"""Where and when each single-wave workgroup of the weight-gradient launch ran (odk_dw_set_profile):
    python tools/gpu_dw_profile.py     -> wave durations, launch span, waves per SIMD / CU / XCD"""
import collections, ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_duck_playground_amd import engine
import numpy as np

L = engine.load_library()
L.odk_dw_set_profile.argtypes = [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
specs = ((5120, 101, 28), (5376, 212, 1))
tot, entries = 0, []
for n, n_in, n_out in specs:
    widths = (n_in,) + engine.MLP_HIDDEN + (n_out,)
    for l in range(4):
        entries.append((tot, widths[l + 1], widths[l])); tot += widths[l + 1] * widths[l] + widths[l + 1]
flat_g = torch.zeros(tot, device="cuda")
layers, k = [], 0
for n, n_in, n_out in specs:
    widths = (n_in,) + engine.MLP_HIDDEN + (n_out,)
    for l in range(4):
        dz = torch.randn(n // 4, widths[l + 1], 4, device="cuda", generator=g) * 1e-2
        h = torch.randn(n // 4, widths[l], 4, device="cuda", generator=g)
        layers.append((dz, h, widths[l + 1], widths[l], entries[k][0])); k += 1
KS = int(os.environ.get("ODK_DW_KS", "8"))
ws = torch.empty(KS * engine.DwGemm.workspace_stride(tot), device="cuda")
dw = engine.DwGemm(layers, flat_g, ws, KS)
for _ in range(20): dw()
torch.cuda.synchronize()
prof = torch.zeros(4 * 4096, dtype=torch.int64, device="cuda")
L.odk_dw_set_profile(prof.data_ptr())
dw(); torch.cuda.synchronize()
L.odk_dw_set_profile(None)
p = prof.cpu().numpy().reshape(-1, 4)
p = p[: int((p[:, 1] > 0).sum())]
t0 = p[:, 0].min()
start, end = (p[:, 0] - t0) / 100.0, (p[:, 1] - t0) / 100.0     # us (100 MHz)
dur = end - start
print(f"waves {len(p)}  launch span {end.max():.1f} us   wave duration: min {dur.min():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f} us")
print(f"start times: median {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f} us")
hw, xcc = p[:, 2] & 0xFFFFFFFF, (p[:, 2] >> 32) & 0xF
cyc = p[:, 3]
print(f"shader clock during the launch: median {np.median(cyc / dur):.0f} MHz (cycles / wall time per wave; min {np.min(cyc / dur):.0f} max {np.max(cyc / dur):.0f})")
slot = np.arange(len(p)) >> 3
print("median duration by slot range: 0-119", round(float(np.median(dur[slot < 120])), 1), " 120-127", round(float(np.median(dur[slot >= 120])), 1))
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
per_simd = collections.Counter(zip(xcc, se, sh, cu, simd)); per_cu = collections.Counter(zip(xcc, se, sh, cu)); per_x = collections.Counter(xcc)
print("SIMDs used", len(per_simd), "waves per SIMD histogram", sorted(collections.Counter(per_simd.values()).items()))
print("CUs used", len(per_cu), "waves per CU histogram", sorted(collections.Counter(per_cu.values()).items()))
print("waves per XCC", sorted(per_x.items()))
# duration vs co-residency
for nres in sorted(set(per_simd.values())):
    sel = np.array([per_simd[(a, b, c, d, e)] == nres for a, b, c, d, e in zip(xcc, se, sh, cu, simd)])
    print(f"  waves on a SIMD with {nres} wave(s): n {sel.sum()} median duration {np.median(dur[sel]):.1f} us, end median {np.median(end[sel]):.1f} max {end[sel].max():.1f}")
order = np.argsort(dur)
print("longest 5 waves: dur/start/end", [(round(dur[i], 1), round(start[i], 1), round(end[i], 1)) for i in order[-5:]])
print("shortest 5 waves:", [(round(dur[i], 1), round(start[i], 1), round(end[i], 1)) for i in order[:5]])
