"""Per-workgroup timeline of the fused network launches (odk_mlp_set_wg_profile): how long a 16-sample tile takes, by how many
tiles share its CU, and how the launch's span compares:   python tools/gpu_mlp_wg_profile.py [fwd|bwd]"""
import collections, ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open_duck_playground_amd import engine

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
L = engine.load_library()
L.odk_mlp_set_wg_profile.argtypes = [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
specs = ((5120, 101, 28), (5376, 212, 1))
tot, entries = 0, []
for n, n_in, n_out in specs:
    widths = (n_in,) + engine.MLP_HIDDEN + (n_out,)
    for l in range(4):
        entries.append((tot, widths[l + 1], widths[l], l > 0)); tot += widths[l + 1] * widths[l] + widths[l + 1]
table = engine.WeightTable(entries)
flat = torch.randn(tot, device="cuda", generator=g) * 0.05
pf, pb = torch.zeros(table.fwd_size, device="cuda"), torch.zeros(table.bwd_size, device="cuda")
engine.pack_weights(flat, pf, pb, table)
nets, k = [], 0
for n, n_in, n_out in specs:
    widths = (n_in,) + engine.MLP_HIDDEN + (n_out,)
    x = torch.randn(n, n_in, device="cuda", generator=g)
    b = [flat[entries[k + l][0] + widths[l + 1] * widths[l]:][:widths[l + 1]] for l in range(4)]
    nets.append(dict(x=x, wf=[table.fwd_view(pf, k + l) for l in range(4)], wb=[table.bwd_view(pb, k + l) for l in range(4)], b=b, out=torch.empty(n, n_out, device="cuda"),
                     dout=torch.randn(n, n_out, device="cuda", generator=g) * 1e-3, **engine.FusedMLP.train_buffers(n, n_in, n_out, "cuda")))
    k += 4
L.odk_mlp_set_diag.argtypes = [C.c_int]
L.odk_mlp_set_diag(int(os.environ.get("ODK_MLP_DIAG", "0")))
op = engine.FusedMLP(nets)
fn = op.forward if which == "fwd" else op.backward
op.forward()
for _ in range(20): fn()
torch.cuda.synchronize()
prof = torch.zeros(4 * 2048, dtype=torch.int64, device="cuda")
L.odk_mlp_set_wg_profile(prof.data_ptr())
fn(); torch.cuda.synchronize()
L.odk_mlp_set_wg_profile(None)
p = prof.cpu().numpy().reshape(-1, 4)
nwg = int((p[:, 1] > 0).sum()); p = p[:nwg]
ntp = specs[0][0] // 16
t0 = p[:, 0].min()
start, end = (p[:, 0] - t0) / 100.0, (p[:, 1] - t0) / 100.0
dur = end - start
hw, xcc = p[:, 2] & 0xFFFFFFFF, (p[:, 2] >> 32) & 0xF
cu_key = list(zip(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF))
per_cu = collections.Counter(cu_key)
print(f"{which}: workgroups {nwg} (policy {ntp}, value {nwg - ntp}; tile deal {'off' if os.environ.get('ODK_MLP_NO_DEAL') else 'on'})  launch span {end.max():.1f} us  clock {np.median(p[:, 3] / dur):.0f} MHz")
print("CUs used", len(per_cu), "tiles per CU histogram", sorted(collections.Counter(per_cu.values()).items()))
print(f"start: median {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f} us")
kind = np.array(["V" if (v >> 40) & 1 else "P" for v in p[:, 2]])      # which network the block worked on (the kernels' tile deal decides)
mix = collections.defaultdict(list)
for i, key in enumerate(cu_key):
    mix[key].append(i)
for ncu in sorted(set(per_cu.values())):
    sel = np.array([per_cu[key] == ncu for key in cu_key])
    for kd in "PV":
        s2 = sel & (kind == kd)
        if s2.any():
            print(f"  CU with {ncu} tiles, {kd} tile: n {s2.sum():4d}  duration median {np.median(dur[s2]):.1f} p90 {np.percentile(dur[s2], 90):.1f} max {dur[s2].max():.1f}   end median {np.median(end[s2]):.1f} max {end[s2].max():.1f}")
combos = collections.Counter("".join(sorted(kind[i] for i in v)) for v in mix.values())
print("CU tile mixes", sorted(combos.items()))
cu_end = {key: max(end[i] for i in v) for key, v in mix.items()}
for combo in sorted(combos):
    e = [cu_end[key] for key, v in mix.items() if "".join(sorted(kind[i] for i in v)) == combo]
    print(f"  {combo}: CUs {len(e)}  last tile ends: median {np.median(e):.1f} max {max(e):.1f} us")
