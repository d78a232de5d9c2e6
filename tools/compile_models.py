"""Compiles the reference's MJCF scenes and reference-motion pickle into the assets shipped
with the package (run in the build container, where /root/reference is mounted):

    python tools/compile_models.py [/root/reference]

Outputs (data only, no reference code):
  open_duck_playground_amd/assets/<task>.npz            compiled model arrays (mjcf.compile_mjcf)
  open_duck_playground_amd/assets/prm_table.npz         [6,4,10,40,16] polynomial table + grids
"""
import hashlib
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from open_duck_playground_amd import mjcf  # noqa: E402
from open_duck_playground_amd.model import Model  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
XML = os.path.join(REF, "playground/open_duck_mini_v2/xmls")
OUT = os.path.join(os.path.dirname(__file__), "..", "open_duck_playground_amd", "assets")

# reference constants.py:18-34 (rough_terrain's XML does not exist in the reference)
TASKS = {
    "flat_terrain": "scene_flat_terrain.xml",
    "flat_terrain_backlash": "scene_flat_terrain_backlash.xml",
    "rough_terrain_backlash": "scene_rough_terrain_backlash.xml",
}


def convert_prm(pkl_path):
    """Restates PolyReferenceMotion.process (reference poly_reference_motion.py:74-144):
    sorted grids, coefficients flipped to highest-power-first (polyval order)."""
    raw = open(pkl_path, "rb").read()
    data = pickle.loads(raw)
    dxs, dys, dths = set(), set(), set()
    for name in data:
        a, b, c = (float(t) for t in name.split("_"))
        dxs.add(a); dys.add(b); dths.add(c)
    dxs, dys, dths = sorted(dxs), sorted(dys), sorted(dths)
    first = next(iter(data.values()))
    table = np.zeros((len(dxs), len(dys), len(dths), 40, 16))
    for name, e in data.items():
        a, b, c = (float(t) for t in name.split("_"))
        coeffs = [np.flip(np.asarray(v, dtype=np.float64)) for v in e["coefficients"].values()]
        table[dxs.index(a), dys.index(b), dths.index(c)] = np.array(coeffs)
    # ranges start from [0,0] and are widened (poly_reference_motion.py:58-60,101-106)
    rng = lambda v: [min(0.0, min(v)), max(0.0, max(v))]
    return dict(table=table.astype(np.float32), table64=table, dxs=np.array(dxs), dys=np.array(dys), dthetas=np.array(dths),
                dx_range=np.array(rng(dxs)), dy_range=np.array(rng(dys)), dtheta_range=np.array(rng(dths)),
                nb_steps_in_period=np.array([int(first["period"] * first["fps"])]),
                period=np.array([first["period"]]), fps=np.array([first["fps"]]),
                source_sha256=np.array(hashlib.sha256(raw).hexdigest()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for task, xml in TASKS.items():
        m = Model.from_xml(os.path.join(XML, xml), sim_dt=0.002)
        m.save(os.path.join(OUT, f"{task}.npz"))
        print(task, "nq/nv/nu", m.nq, m.nv, m.nu)
    prm = convert_prm(os.path.join(REF, "playground/open_duck_mini_v2/data/polynomial_coefficients.pkl"))
    np.savez_compressed(os.path.join(OUT, "prm_table.npz"), **prm)
    print("prm", prm["table"].shape, prm["dxs"], prm["dys"], prm["dthetas"], prm["nb_steps_in_period"])
