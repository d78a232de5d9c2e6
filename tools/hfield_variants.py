"""Height-field hypothesis sweep.  The reference's rough-terrain task does not train on the prism algorithm as
this build recalls it; nothing here can say what MJX really does, but it CAN say which reading of `hfield_convex` makes the
shipped task behave.  Readings (oracle `hfield_mode`; 3 is a run-time OPT-IN of the shipped library (odk_env_config.hfield_up_normals_only, default off), 4 a separate kernel build
`make -C open_duck_playground_amd/csrc libodk_hfv4.so`):
  0  prisms, full SAT over all prism faces / edges, the four deepest contacts of all prisms (what the kernels run)
  1  round 2's rule: the plane of ONE triangle under the foot centre
  2  prisms, but only a prism's TOP triangle collides (side / bottom faces and vertical edges give no axis)       [oracle only]
  3  mode 0, contacts kept only when the normal points up (n_z > 0.5)
  4  mode 0, one contact per prism (its deepest), the four deepest kept (MuJoCo-C gives one contact per prism)
  5  mode 0's candidates, but the four contacts chosen by the plane-convex manifold heuristic (first active, farthest, farthest from the line, farthest
     from the triangle) over all prisms' active candidates with their mean normal, instead of the four deepest  [oracle only; round 6, VERDICT r5 #6]

    python tools/hfield_variants.py oracle           CPU: zero-action topple rate + contact statistics per mode (slow oracle: 48 envs x 40 steps)
    python tools/hfield_variants.py gpu [steps]      GPU box: modes 0 / 3 / 4 through the kernels -- zero-action topple rate at 4096 envs, then the
                                                     reference's command line (--task rough_terrain_backlash) for `steps` (default 40 M) env steps
Writes gpurun_out/hfield_variants_<part>.json (copied to profiles/r4/hfield_variants.json)."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
TASK = "rough_terrain_backlash"
MODES = tuple(int(x) for x in os.environ.get("ODK_HFIELD_MODES", "0,1,2,3,4,5").split(","))


def oracle_part():
    import oracle as O
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    O.build()
    model = load_task_model(TASK)
    prm = O.OraclePRM(engine.load_prm(), f32=True)
    res = {}
    for mode in MODES:
        om = O.OracleModel(model.blob(), f32=True); om.set_int("hfield_mode", mode)
        n, steps = 48, 40
        envs = [O.OracleEnv(om, prm) for _ in range(n)]
        for i, e in enumerate(envs):
            e.cfg["noise_level"][0] = 0; e.cfg["push_enable"][0] = 0
            e.reset(1, i)
        done = 0; ncon = 0; nhoriz = 0; depth = []; feet = 0
        t0 = time.time()
        for t in range(steps):
            for e in envs:
                e.step(np.zeros(14))
                done += int(e["done"][0] != 0)
                cd = np.array(e.data["contact_dist"][:8]); fr = np.array(e.data["contact_frame"][:72]).reshape(8, 9)
                live = cd < 0
                ncon += int(live.sum()); nhoriz += int((live & (np.abs(fr[:, 2]) < 0.5)).sum()); depth += list(-cd[live])
                feet += int(live[:4].any()) + int(live[4:].any())
        res[str(mode)] = dict(envs=n, steps=steps, zero_action_done_rate_per_step=done / (n * steps), active_floor_contacts_per_env_step=ncon / (n * steps),
                              horizontal_normal_fraction=nhoriz / max(ncon, 1), median_depth_mm=1e3 * float(np.median(depth)) if depth else None,
                              feet_in_contact_fraction=feet / (2 * n * steps), seconds=round(time.time() - t0, 1))
        print(mode, res[str(mode)], flush=True)
    json.dump(res, open(os.path.join(OUT, "hfield_variants_oracle.json"), "w"), indent=1)
    return res


def gpu_part(steps):
    res = {}
    csrc = os.path.join(ROOT, "open_duck_playground_amd", "csrc")
    for mode, lib in ((0, "libodk.so"), (3, "libodk.so"), (4, "libodk_hfv4.so")):
        path = os.path.join(csrc, lib)
        if not os.path.exists(path):
            print("missing", path); continue
        env = dict(os.environ, ODK_LIB=path)
        code = ("import torch, json\nfrom open_duck_playground_amd import engine\nfrom open_duck_playground_amd.model import load_task_model\n"
                f"cfg = engine.default_config(); cfg.hfield_up_normals_only = {int(mode == 3)}\n"
                f"b = engine.Batch(load_task_model('{TASK}'), 4096, cfg)\nb.reset(seed=1)\nz = torch.zeros(4096, 14, device='cuda')\nd = []; r = []\n"
                "for t in range(100):\n    b.step(z); d.append(float(b.done.mean())); r.append(float(b.reward.mean()))\n"
                "print(json.dumps(dict(done=sum(d) / len(d), done_last20=sum(d[-20:]) / 20, reward=sum(r) / len(r))))\n")
        o = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
        stats = json.loads([l for l in o.stdout.splitlines() if l.startswith("{")][-1]) if o.returncode == 0 else {"error": o.stderr[-400:]}
        out_dir = os.path.join(OUT, f"train_hfv{mode}")
        t0 = time.time()
        tr = subprocess.run([sys.executable, "-m", "open_duck_playground_amd.runner", "--task", TASK, "--num_timesteps", str(steps), "--output_dir", out_dir] + (["--hfield_up_normals_only"] if mode == 3 else []),
                            capture_output=True, text=True, env=env, cwd=ROOT)
        wall = time.time() - t0
        evals = []
        mp = os.path.join(out_dir, "metrics.jsonl")
        if os.path.exists(mp):
            for line in open(mp):
                m = json.loads(line)
                if "eval/episode_reward" in m:
                    evals.append((int(m.get("step", 0)), round(m["eval/episode_reward"], 1), round(m.get("eval/avg_episode_length", 0), 1)))
        for f in os.listdir(out_dir) if os.path.isdir(out_dir) else []:
            if f.endswith((".pt", ".onnx")) or f.startswith("events.out"):
                os.remove(os.path.join(out_dir, f))
        res[str(mode)] = dict(kernel_build=lib, zero_action_done_rate_per_step=stats.get("done"), zero_action_done_rate_last20=stats.get("done_last20"),
                              zero_action_mean_reward=stats.get("reward"), train_env_steps=steps, train_wall_s=round(wall, 1),
                              eval_reward_and_episode_length=evals, train_rc=tr.returncode, err=(tr.stderr[-300:] if tr.returncode else None))
        print(mode, json.dumps(res[str(mode)]), flush=True)
    json.dump(res, open(os.path.join(OUT, "hfield_variants_gpu.json"), "w"), indent=1)


if __name__ == "__main__":
    part = sys.argv[1] if len(sys.argv) > 1 else "oracle"
    if part == "oracle":
        oracle_part()
    else:
        gpu_part(int(sys.argv[2]) if len(sys.argv) > 2 else 40_000_000)
