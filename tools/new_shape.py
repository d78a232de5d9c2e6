#!/usr/bin/env python3
"""What does it take to run YOUR robot through the kernels?  (reference README.md:74-85 "Adding a new robot")

    python tools/new_shape.py path/to/robot.xml [--add]

Compiles the MJCF with the build's compiler (open_duck_playground_amd/mjcf.py), builds the kernels' topology tables (tables.py) and asks
the loader (`odk_model_load`, host-only: no GPU needed) whether a compiled kernel shape takes the model.
 * yes: prints the shape it matched and the env's sizes -- `python -m open_duck_playground_amd.runner --xml robot.xml` trains it.
 * no compiled shape: prints the TWO lines to add to open_duck_playground_amd/csrc/odk_engine.hip -- the `using ShapeX = Shape<...>` line
   (model dimensions are template parameters: every loop of the fused step kernel is unrolled over them) and the entry of `ODK_SHAPES`, the
   list every per-shape dispatch of the host code goes through -- then `make -C open_duck_playground_amd/csrc` (~90 s).
 * `--add`: does it for you -- appends the robot's shape to open_duck_playground_amd/csrc/odk_shapes_user.h (which odk_engine.hip includes when it
   exists: `using ShapeU<k> = Shape<...>;` lines + `#define ODK_USER_SHAPES(X) ...`), checks that the kernels' static_asserts take the shape
   (`hipcc -fsyntax-only`, seconds) and rebuilds the library; `--add --no-build` stops before the rebuild.
 * anything else the kernels do not model (a tree that is not a floating base + <= 3 serial chains of <= 6 dofs, tendons, more than two
   foot colliders, ...): the loader's own message, by name.
What the XML must carry (the names reference constants.py / base.py look up): sites `imu`, `left_foot`, `right_foot`; geoms
`left_foot_bottom_tpu`, `right_foot_bottom_tpu`, `floor`; the 15 sensors of open_duck_mini_v2.xml:26-42; keyframe `home` with qpos and ctrl;
gear-1 position actuators."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shape_line(model, name="ShapeX"):
    """the `using` line for a model: Shape<nq, nv, nbody, nu, njnt, nM, nH, nrow, depth, virtual depth, cone-only, chain length, optional constraint code>"""
    import numpy as np
    from open_duck_playground_amd.tables import build_kernel_tables
    t = build_kernel_tables(model.a)
    nM, nH = int(t["k_nM"][0]), int(t["k_nH"][0])
    nrow = len(t["k_fl_dof"]) + len(t["k_lim_jnt"]) + 48
    dt, dv = int(np.max(t["k_dof_depth"])), int(np.max(t["k_vdof_depth"]))
    chains = [int(c) for c in np.asarray(t["k_chain_len"])[: int(t["k_nchain"][0])]]
    cl = max(chains) if chains else 0
    dims = (model.nq, model.nv, model.nbody, model.nu, model.njnt, nM, nH, nrow, dt, dv)
    line = f"using {name} = Shape<{', '.join(str(d) for d in dims)}, false, {cl}, true>;"
    return line, dims, chains


USER_HEADER = os.path.join(ROOT, "open_duck_playground_amd", "csrc", "odk_shapes_user.h")
FIRST_USER_INDEX = 4      # ODK_SHAPES' own entries: 0 .. 3


def add_user_shape(line_args: str, header: str = USER_HEADER) -> str:
    """appends `Shape<line_args>` to the user header (idempotent); returns the alias name"""
    shapes = []
    if os.path.exists(header):
        shapes = re.findall(r"using ShapeU\d+ = Shape<([^>]*)>;", open(header).read())
    if line_args not in shapes:
        shapes.append(line_args)
    with open(header, "w") as f:
        f.write("// written by tools/new_shape.py --add: robots added without editing odk_engine.hip (which includes this file when it exists)\n#pragma once\n")
        for k, a in enumerate(shapes):
            f.write(f"using ShapeU{k} = Shape<{a}>;\n")
        f.write("#define ODK_USER_SHAPES(X) " + " ".join(f"X({FIRST_USER_INDEX + k}, ShapeU{k})" for k in range(len(shapes))) + "\n")
    return f"ShapeU{shapes.index(line_args)}"


def syntax_check(include_dir: str = None):
    """do the kernels' static_asserts take every listed shape?  (hipcc -fsyntax-only: template instantiation without code generation, seconds)"""
    import subprocess
    csrc = os.path.join(ROOT, "open_duck_playground_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fsyntax-only"] + (["-I", include_dir] if include_dir else []) + [os.path.join(csrc, "odk_engine.hip")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    return p.returncode == 0, "\n".join(l for l in p.stderr.splitlines() if "error" in l or "static assertion" in l)[:2000]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    if len(args) != 1:
        print(__doc__)
        return 2
    from open_duck_playground_amd import constants, engine
    from open_duck_playground_amd.model import Model
    xml = args[0]
    model = Model.from_xml(xml, sim_dt=0.002)
    try:
        line, dims, chains = shape_line(model)
    except (ValueError, KeyError, IndexError) as e:
        print(f"{xml}: the kernels' tables cannot be built -- a name the reference's constants.py / base.py look up is missing ({e}); see this file's docstring")
        return 1
    robot = constants.robot_of(model)
    print(f"{xml}: nq {model.nq}, nv {model.nv}, {model.nbody} bodies, {model.nu} actuators, {model.njnt} joints; serial chains of {chains} dofs")
    print(f"  leg joints (JOINTS_ORDER_NO_HEAD): {robot.joints_order_no_head}")
    try:
        red = engine.model_reduction(model)
    except engine.OdkError as e:
        msg = str(e)
        if "has no compiled kernel" not in msg:
            print(f"  the loader refuses this model: {msg}")
            return 1
        src = open(os.path.join(ROOT, "open_duck_playground_amd", "csrc", "odk_engine.hip")).read()
        m = re.search(r"#define ODK_SHAPES\(X\) (.*)", src)
        entries = m.group(1).strip() if m else "..."
        n = len(re.findall(r"X\(", entries))
        print("  no compiled kernel shape takes it.  Add to open_duck_playground_amd/csrc/odk_engine.hip (next to ShapeD):")
        print(f"    {line}")
        print("  and extend the list every per-shape dispatch goes through:")
        print(f"    #define ODK_SHAPES(X) {entries} X({n}, ShapeX)")
        print("  then: make -C open_duck_playground_amd/csrc   (hipcc, ~90 s; a twin-dof (backlash) model must have nv = 30 like the duck's: Shape::PAIRED)")
        print("  or let this tool do it: python tools/new_shape.py " + xml + " --add")
        if "--add" in flags:
            alias = add_user_shape(re.search(r"Shape<([^>]*)>", line).group(1))
            ok, err = syntax_check()
            print(f"  --add: {alias} written to {os.path.relpath(USER_HEADER, ROOT)}; the kernels' static_asserts " + ("take it" if ok else "REFUSE it:\n" + err))
            if not ok:
                return 1
            if "--no-build" not in flags:
                print("  rebuilding csrc/libodk.so ...")
                engine.build_library(force=True)
                import subprocess      # (a fresh process: this one holds the old library image)
                return subprocess.call([sys.executable, os.path.abspath(__file__), xml])
        if max(chains or [0]) > 6 or len(chains) > 3:
            print("  NOTE: the chain solve handles a floating base with <= 3 serial chains of <= 6 dofs; this tree will be refused at load")
        if dims[7] < 71:
            print(f"  NOTE: {dims[7]} constraint rows: the foot-foot routine borrows 282 floats from the four row arrays (4 nrow >= 282, i.e. a robot with >= 12 "
                  "limited hinges + friction-loss dofs in total); this shape will stop at a static_assert")
        if model.nu > 16 or model.nv > 32 or model.nbody > 32:
            print("  NOTE: one lane per dof / body / actuator at 32 lanes per env: nv, nbody <= 32, nu <= 16")
        return 1
    nobs, npriv = engine.model_obs_sizes(model, 0)
    print(f"  a compiled kernel shape takes it ({'twin dofs merged: ' if red['paired'] else ''}{red['nvr']} reduced dofs, {red['env_lds_floats']} floats of LDS per env)")
    print(f"  env: action {model.nu}, observation {nobs}, privileged observation {npriv}")
    print(f"  train: python -m open_duck_playground_amd.runner --xml {xml}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
