"""Graph-captured PPO minibatch step for the GPU: explicit forward / backward over flat parameter buffers.

brax runs `num_updates_per_batch * num_minibatches` (= 128) clipped-Adam steps per training step
(reference common/runner.py:104-118 -> brax ppo.train); each is ~10 small GEMMs surrounded by ~300
element-wise ops, so under an autograd engine it is launch-bound.  Here one step is a single HIP graph of 8 launches
(the reference architecture, in -> 512 -> 256 -> 128 -> out; csrc/odk_mlp.hip, csrc/odk_learner.hip):

    minibatch gather, forward of both networks (all layers, f32 matrix cores), GAE, loss head (forward + gradients),
    backward-data of both networks, weight gradients of all 8 layers, their finishing launch (slice fold, bias gradients,
    partial sums of the gradient norm), clip + Adam (which also refreshes the packed weight copies the kernels read);
    data-parallel: an RCCL all-reduce of the flat gradient (and the norm in its own launch) before the clip + Adam.

Other architectures (and ODK_LEARNER_FUSED=0) run the library path: one hipBLASLt GEMM per layer with the activation /
bias-gradient kernels between them, policy and value networks as two parallel branches of the graph.

Parameters live in ONE flat fp32 buffer (the modules' `.data` are views of it, so the rollout policy sees
every update), gradients in another: the data-parallel exchange is exactly one all-reduce of 1.97 MB.
The autograd path in `train.py` (`ppo_loss`) is the reference implementation these kernels are tested against.
"""
from __future__ import annotations

import os
import weakref
import tempfile
from typing import Dict

import torch
import torch.nn.functional as F

from .. import engine
from .networks import PPONetworks

_DEBUG_NONFINITE = os.environ.get("ODK_DEBUG_NONFINITE") == "1"   # per-step finiteness check (synchronises: debugging only)


_TUNED_ASSET = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "tunableop_gfx950.csv")


_scratch_files = []
_reported_validators = False


def _cleanup_tunableop():
    """TunableOp rewrites its results file from a static destructor at process exit: point it at /dev/null and delete the
    per-process scratch copy, so that a run leaves no /tmp/odk_tunableop_<pid>.csv behind."""
    try:
        import torch.cuda.tunable as tn
        if _scratch_files:
            tn.set_filename(os.devnull)
            tn.tuning_enable(False)
    except Exception:
        pass
    for f in _scratch_files:
        try:
            os.unlink(f)
        except OSError:
            pass


def _report_validators(tn, path: str) -> None:
    """Says once whether the shipped selections apply: TunableOp silently ignores a file whose validators (PyTorch, HIP,
    hipBLASLt, rocBLAS builds, arch) differ from the running stack and tunes from scratch instead."""
    global _reported_validators
    if _reported_validators:
        return
    _reported_validators = True
    try:
        mine = {str(k): str(v) for k, v in tn.get_validators()}
        theirs = {}
        with open(path) as f:
            for line in f:
                c = line.rstrip("\n").split(",", 2)
                if c[0] == "Validator" and len(c) == 3:
                    theirs[c[1]] = c[2]
        bad = {k: (v, mine.get(k)) for k, v in theirs.items() if mine.get(k) != v}
        if bad:
            print(f"[ppo] shipped GEMM selections NOT applied (validator mismatch {bad}): tuning during warm-up instead")
        else:
            print(f"[ppo] shipped GEMM selections applied ({os.path.basename(path)}, validators match)")
    except Exception as e:
        print(f"[ppo] could not compare GEMM selection validators ({type(e).__name__}: {e})")


def tunable_off() -> None:
    """Turns TunableOp off again (it is process-wide): called when the learner / training loop is torn down."""
    try:
        import torch.cuda.tunable as tn
        tn.tuning_enable(False)
        tn.enable(False)
    except Exception:
        pass


def _tunable(tuning: bool) -> bool:
    """hipBLASLt / rocBLAS kernel selection measured on this GPU for the learner's 30 GEMM shapes (PyTorch TunableOp):
    the library heuristics pick stream-K 32x32 tiles for the K = 5120 weight-gradient GEMMs (30-36 us each); the tuned
    choices run 17-24 us (-150 us per minibatch step).

    The selections for the reference PPO configuration (batch 256 x unroll 20, the duck's observation sizes; also the
    policy-inference GEMMs of an 8192-env rollout and a 128-env evaluator) ship in `assets/tunableop_gfx950.csv`:
    candidates timed alone are within noise of each other and what matters is the step time with the policy / value
    branches overlapping, so the shipped learner entries are the best of 18 tuning runs measured on the whole minibatch
    step (53.9 vs 54.1-58.3 ms per training step at the time).  TunableOp checks the file's version validators
    (PyTorch, HIP, hipBLASLt, rocBLAS, arch) and ignores it on a mismatch; shapes it does not list are tuned during the
    warm-up steps as before.  $ODK_TUNABLEOP_FILE overrides the file (TunableOp rewrites it on exit)."""
    try:
        import torch.cuda.tunable as tn
    except Exception as e:                          # optional speed-up, never a requirement
        print(f"[ppo] GEMM tuning unavailable ({type(e).__name__}: {e})")
        return False
    try:
        if tuning:
            tn.set_max_tuning_duration(int(os.environ.get("ODK_TUNE_MS", "30")))          # ms per candidate kernel
            tn.set_max_tuning_iterations(int(os.environ.get("ODK_TUNE_ITERS", "20")))
            path = os.environ.get("ODK_TUNABLEOP_FILE")
            if not path:                            # per-process scratch copy: the shipped asset is never written to
                path = os.path.join(tempfile.gettempdir(), f"odk_tunableop_{os.getpid()}.csv")
                if path not in _scratch_files:
                    import atexit
                    if not _scratch_files:
                        atexit.register(_cleanup_tunableop)
                    _scratch_files.append(path)
                if os.path.exists(_TUNED_ASSET) and os.environ.get("ODK_TUNABLEOP_PRETUNED", "1") == "1":
                    import shutil
                    shutil.copyfile(_TUNED_ASSET, path)
                    _report_validators(tn, path)
            tn.set_filename(path)
        tn.enable(True)
        tn.tuning_enable(tuning)
        return True
    except Exception as e:
        print(f"[ppo] GEMM tuning unavailable ({type(e).__name__}: {e})")
        try:
            tn.tuning_enable(False)
        except Exception:
            pass
        return False


@torch.no_grad()
def tune_inference_shapes(net: PPONetworks, rows) -> None:
    """Library kernel selection for the policy-inference GEMMs of the rollout (rows = envs per GPU) and of the evaluator
    (rows = eval envs): without it the 128-row evaluator GEMMs run 256x128 macro-tiles (61 us for a 128 x 101 x 512
    product).  The shipped selection file covers the reference sizes (8192 / 128 rows), so this usually tunes nothing."""
    dev = next(net.parameters()).device
    if dev.type != "cuda" or not _tunable(True):
        return
    try:
        for r in rows:
            x = torch.zeros(int(r), net.policy.layers[0].in_features, device=dev)
            for _ in range(2):
                net.policy(net.norm_obs(x))
        torch.cuda.synchronize(dev)
    finally:
        _tunable(False)


_FUSED_MLP = os.environ.get("ODK_LEARNER_FUSED", "1") == "1"   # whole-network forward / backward-data launches (csrc/odk_mlp.hip; 0: one library GEMM per layer)


class _FlatMLP:
    """Views of one MLP's weights / gradients inside the flat buffers + explicit forward / backward over persistent
    activation buffers (`bind`): fixed addresses, so the weight-gradient launch is prepared once and nothing is allocated
    inside the step."""

    def __init__(self, mlp, flat_p, flat_g, off: int):
        self.W, self.b, self.gW, self.gb, self.goff = [], [], [], [], []
        self.zs = self.hs = self.dzs = self.dhs = self.partials = self.fold = None
        for lin in mlp.layers:
            for name in ("weight", "bias"):
                p = getattr(lin, name)
                n = p.numel()
                flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + n].view_as(p)
                (self.W if name == "weight" else self.b).append(p.data)
                (self.gW if name == "weight" else self.gb).append(flat_g[off:off + n].view_as(p))
                if name == "weight":
                    self.goff.append(off)
                off += n
        self.end = off

    def bind(self, x, dz_top, fused: bool, n: int = None, rows: dict = None):
        """Fixes the step's tensors: x [n, in] (network input), dz_top [n, out] (gradient w.r.t. the output, written by the loss
        head).  Library path (`fused` false: one GEMM per layer): allocates z / h per layer, dz / dh per hidden layer and the
        bias-gradient tile sums; the fused path (csrc/odk_mlp.hip) needs only the output buffer here, see `fused_desc`.
        `rows` (fused path): the row sources of `odk_mlp_desc` -- x is then the WHOLE rollout and `n` the minibatch's row count."""
        n, dev = (x.shape[0] if n is None else int(n)), x.device
        self.x, self.dz_top, self.n, self.rows = x, dz_top, n, rows
        self.zs = [torch.empty(n, w.shape[0], device=dev) if (not fused or i == len(self.W) - 1) else None for i, w in enumerate(self.W)]
        if fused:
            return
        self.hs = [x] + [torch.empty(n, w.shape[0], device=dev) for w in self.W[:-1]]
        self.dzs = [torch.empty(n, w.shape[0], device=dev) for w in self.W[:-1]]
        self.dhs = [torch.empty_like(t) for t in self.dzs]
        # per-layer tile sums of dz; folded into the bias gradients by ONE launch after the chain (the bias gradients
        # are first read by the clip + Adam step, so their finalisation need not sit between the GEMMs)
        self.partials = [torch.empty(((n + 63) // 64) * w.shape[0], device=dev) for w in self.W]
        self.fold = engine.ColsumFinalize([(p, self.gb[i]) for i, p in enumerate(self.partials)], n)

    def fused_ok(self) -> bool:
        """The shapes csrc/odk_mlp.hip is built for: in -> 512 -> 256 -> 128 -> out with in <= 224, out <= 32 -- and every weight
        at a flat offset / with an element count that is a multiple of 4 (the weight-gradient launch writes 16-byte pieces: an
        odd action size, 2 A = 14, 18, ..., shifts everything behind the policy's last bias off that grid).  Otherwise the learner
        falls back to the library-GEMM path instead of failing in `engine.DwGemm`."""
        return (_FUSED_MLP and len(self.W) == 4 and tuple(w.shape[0] for w in self.W[:3]) == engine.MLP_HIDDEN
                and self.W[0].shape[1] <= engine.MLP_MAX_IN and self.W[3].shape[0] <= engine.MLP_MAX_OUT
                and all(o % 4 == 0 and w.numel() % 4 == 0 for o, w in zip(self.goff, self.W)))

    def fused_desc(self, table, k0, packed_f, packed_b):
        """This network for `engine.FusedMLP`: the bound buffers + swish' buffers, bias-gradient tile sums (16-row tiles, engine.MLP_TILE) and the
        views of its weights (entries k0 .. k0 + 3 of `table`) inside the packed buffers."""
        n, n_in, n_out = self.n, self.x.shape[1], self.W[-1].shape[0]
        tb = engine.FusedMLP.train_buffers(n, n_in, n_out, self.x.device)
        self.tile_sums, self.tiles = tb["bias_partial"], engine.quad_rows(n) // engine.MLP_TILE
        # weight gradients dW_l = dz_l^T h_{l-1}: (dz, h, n_out, n_in, offset in the flat gradient buffer), quad-row operands
        acts, dzs = [tb["xp"]] + tb["h"], tb["dz"] + [tb["doutp"]]
        self.dw_layers = [(dzs[l], acts[l], w.shape[0], w.shape[1], self.goff[l]) for l, w in enumerate(self.W)]
        return dict(x=self.x, wf=[table.fwd_view(packed_f, k0 + l) for l in range(4)], wb=[table.bwd_view(packed_b, k0 + l) for l in range(4)],
                    b=self.b, out=self.zs[-1], dout=self.dz_top, **tb, **(self.rows or {}))

    def forward(self):
        for i, (W, b) in enumerate(zip(self.W, self.b)):
            torch.addmm(b, self.hs[i], W.t(), out=self.zs[i])
            if i + 1 < len(self.W):
                torch.ops.aten.silu.out(self.zs[i], out=self.hs[i + 1])
        return self.zs[-1]

    def backward(self):
        """Library path: gradients straight into the flat buffer.  Below the top layer, dz and its column sums (the bias
        gradient) come out of one fused pass (csrc silu_bwd_colsum) instead of silu_backward + a separate reduction.
        (Measured and dropped: the weight-gradient GEMMs as a further parallel branch beside the dz -> dh chain made the step
        10 % slower, and a branch forked from a branch crashes hipStreamEndCapture on ROCm 7.2.)"""
        top = len(self.W) - 1
        dz = self.dz_top
        engine.colsum_partial(dz, self.partials[top])   # (a torch.sum over the tall [n, <=28] matrix takes 22-28 us)
        for i in range(top, -1, -1):
            torch.mm(dz.t(), self.hs[i], out=self.gW[i])
            if i > 0:
                dh = torch.mm(dz, self.W[i], out=self.dhs[i - 1])
                dz = self.dzs[i - 1]
                engine.silu_bwd_colsum(dh, self.zs[i - 1], dz, None, self.partials[i - 1])
        self.fold()


class FlatLearner:
    KEYS = ("obs", "priv", "raw_action", "log_prob", "reward", "termination", "truncation")

    def __init__(self, net: PPONetworks, cfg: Dict, B: int, T: int, world: int = 1, group=None, use_graph: bool = True,
                 split_update=None, fused_norm: bool = True, capture_allreduce=None, n_traj: int = None):
        """`split_update` (default: world > 1) runs the step as graph A (loss + gradients) -> all-reduce of the flat gradient
        -> graph B (clip + Adam); tests force it at world size 1 to drive the RCCL stream hand-over on one GPU.
        `capture_allreduce` (default: $ODK_LEARNER_CAPTURE_ALLREDUCE == "1"; needs a process group on the RCCL backend): the
        all-reduce is CAPTURED between the two halves, so a data-parallel step is ONE graph replay again (c10d records the
        collective on its own stream inside the capture; no host-issued call between two replays).  Opt-in: measured on a one-rank
        group only (`bench.py --mode ppo --force-split`), no multi-GPU box has run it.
        `n_traj`: how many trajectories of a rollout the learner's resident copy holds (default B * cfg["num_minibatches"]; grown on
        demand).
        `fused_norm` false: the gradient norm from its own launch even without an all-reduce (the summation order of the split
        path: tests compare the two bit for bit)."""
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            raise engine.OdkError("FlatLearner runs on the GPU only (the autograd path in train.py is the CPU reference)")
        self.net, self.cfg, self.B, self.T, self.world, self.group = net, cfg, B, T, world, group
        A = net.action_size
        n_par = sum(p.numel() for p in list(net.policy.parameters()) + list(net.value.parameters()))
        self.flat_p = torch.empty(n_par, device=dev)
        self.flat_g = torch.zeros(n_par, device=dev)
        self.m, self.v = torch.zeros(n_par, device=dev), torch.zeros(n_par, device=dev)
        self.acc = torch.zeros(engine.ADAM_ACC_FLOATS, device=dev)   # [sum g^2, step count, per-block partials...]
        self.policy = _FlatMLP(net.policy, self.flat_p, self.flat_g, 0)
        self.value = _FlatMLP(net.value, self.flat_p, self.flat_g, self.policy.end)
        assert self.value.end == n_par
        n = B * T
        od, pd = net.policy.layers[0].in_features, net.value.layers[0].in_features
        z = lambda *s: torch.zeros(*s, device=dev)
        self.vs, self.adv, self.stats = z(B, T), z(B, T), z(2)
        self.dlogits, self.dval_all = z(n, 2 * A), z(n + B, 1)
        fused = self.policy.fused_ok() and self.value.fused_ok() and n >= 128   # (the weight-gradient launch wants >= 8 rows per slice)
        self.split_update = world > 1 if split_update is None else bool(split_update)
        if capture_allreduce is None:
            capture_allreduce = os.environ.get("ODK_LEARNER_CAPTURE_ALLREDUCE") == "1"
        self.capture_allreduce = bool(capture_allreduce) and self.split_update and use_graph and (world > 1 or group is not None)
        # INDEXED form of the step (round 5): no gathered copy of the minibatch exists.  The rollout stays resident (`self.roll`), the
        # training step's whole shuffle is a device array of trajectory indices (`self.sched`, one B-slice per minibatch step) with a
        # device-side cursor that the clip + Adam launch advances, the entropy noise of all steps sits in a pool indexed by the same
        # cursor: the forward launch and the fused GAE + head launch read their rows THROUGH the indices, and 128 minibatch steps are
        # 128 graph replays with nothing between them.  (Minibatches beyond the fused GAE + head launch's LDS: the gathered form.)
        self.indexed = fused and n <= 5120 and B <= 1024 and os.environ.get("ODK_LEARNER_INDEXED", "1") == "1"
        self._host_cursor, self._host_steps = 0, 0
        if self.indexed:
            self.steps_cap = max(self.NOISE_POOL, int(cfg.get("num_minibatches", 1)) * int(cfg.get("num_updates_per_batch", 1)))
            self.sched = torch.zeros(self.steps_cap * B, dtype=torch.int64, device=dev)
            self.cursor = torch.zeros(1, dtype=torch.int32, device=dev)
            self._pool = z(self.steps_cap, n, A)
            self.loss_partials = z((n + engine.GAE_HEAD_SAMPLES - 1) // engine.GAE_HEAD_SAMPLES, 4)     # the head's per-workgroup sums, folded by the Adam launch
            self._alloc_rollout(int(n_traj) if n_traj else B * int(cfg.get("num_minibatches", 1)), od, pd, A)
            rows = dict(row_idx=self.sched, cursor=self.cursor, traj_len=T, n_main=n, n_traj=self.cap)
            self.policy.bind(self.roll["obs"].view(self.cap * T, od), self.dlogits, True, n, rows)
            self.value.bind(self.roll["priv"].view(self.cap * T, pd), self.dval_all, True, n + B, dict(rows, x_tail=self.roll["last_priv"]))
        else:
            self.priv_all = z(n + B, pd)                # minibatch privileged obs, then the bootstrap rows
            self.static = dict(obs=z(B, T, od), priv=self.priv_all[:n].view(B, T, pd), last_priv=self.priv_all[n:], raw_action=z(B, T, A),
                               log_prob=z(B, T), reward=z(B, T), termination=z(B, T), truncation=z(B, T))
            self._noise = z(n, A)
            self.policy.bind(self.static["obs"].view(n, -1), self.dlogits, fused)
            self.value.bind(self.priv_all, self.dval_all, fused)
        # whole-network launches (csrc/odk_mlp.hip): forward of both networks = 1 launch, backward-data of both = 1 launch, the
        # weight gradients of all 8 layers = 1 + its finishing launch; the kernels read the weights from packed copies (16-byte
        # pieces along the reduction index) that the Adam launch keeps current
        self.fused = None
        self._fused_norm = fused_norm
        if fused:
            self._n_par = n_par
            self._build_fused()
            self.sync_weights()
        self.losses = z(4)                          # running SUMS of (total, policy, value, entropy) over the steps since metrics()
        self.gae_head = None
        if self.indexed:
            self.gae_head = engine.GaeHead(self.policy.zs[-1], self.value.zs[-1].view(-1), self.roll, self._pool, self.sched, self.cursor, self.dlogits,
                                           self.dval_all.view(-1), self.losses, B, T, cfg, 1.0 / self.world, adv=self.adv.view(-1), vs=self.vs.view(-1),
                                           stats=self.stats, loss_partials=self.loss_partials)
        self.nsteps = 0
        self.graph_a = self.graph_b = self.graph_k = None
        self._gather = None; self._gather_src = ()
        self.side = torch.cuda.Stream() if (not fused and os.environ.get("ODK_LEARNER_BRANCHES", "1") == "1") else None   # library path: policy || value
        self.sample_noise = True                    # plain-launch path only: tests inject self.noise instead
        self._gpool, self._pool_k = None, 0
        if use_graph:
            self._capture()

    def _build_fused(self):
        dev = self.flat_p.device
        kslices = 8       # workspace slices of the weight-gradient launch (each folded from two waves inside the kernel)
        if getattr(self, "dw_ws", None) is None:
            self.dw_ws = torch.empty(kslices * engine.DwGemm.workspace_stride(self._n_par), device=dev)   # split-K partial weight gradients
        nets = (self.policy, self.value)
        self.wtable = engine.WeightTable([(o, w.shape[0], w.shape[1], l > 0) for f in nets for l, (o, w) in enumerate(zip(f.goff, f.W))])
        if getattr(self, "packed_f", None) is None:
            self.packed_f, self.packed_b = torch.zeros(self.wtable.fwd_size, device=dev), torch.zeros(self.wtable.bwd_size, device=dev)
        self.fused = engine.FusedMLP([f.fused_desc(self.wtable, 4 * k, self.packed_f, self.packed_b) for k, f in enumerate(nets)])
        # the launch that folds the weight gradients' row slices also folds the bias gradients' tile sums and -- unless the
        # gradient still has to be all-reduced -- leaves the partial sums of its squared norm for the clip + Adam launch
        self.dw_all = engine.DwGemm([l for f in nets for l in f.dw_layers], self.flat_g, self.dw_ws, kslices,
                                    bias=[(t, f.gb[i], f.tiles) for f in nets for i, t in enumerate(f.tile_sums)],
                                    acc=None if (self.split_update or not self._fused_norm) else self.acc)

    def _alloc_rollout(self, cap: int, od: int, pd: int, A: int):
        dev, T = self.flat_p.device, self.T
        z = lambda *s: torch.zeros(*s, device=dev)
        self.cap = int(cap)
        self.roll = dict(obs=z(cap, T, od), priv=z(cap, T, pd), last_priv=z(cap, pd), raw_action=z(cap, T, A), log_prob=z(cap, T), reward=z(cap, T),
                         termination=z(cap, T), truncation=z(cap, T))

    @property
    def noise(self):
        """The entropy sample the NEXT step reads, [n, A] (tests inject theirs here): the pool slot under the cursor."""
        return self._pool[self._host_cursor] if self.indexed else self._noise

    # ---- the step, as plain stream-ordered launches (captured below) ----
    @torch.no_grad()
    def _loss_and_grads(self):
        B, T, n, cfg = self.B, self.T, self.B * self.T, self.cfg
        if self.indexed:
            self.fused.forward()          # rows through the schedule: no gathered copy
            self.gae_head()               # GAE + advantage statistics + loss head, one launch
            self.fused.backward()
            self.dw_all()                 # weight gradients; its finishing launch: slice fold + bias gradients (+ the norm's partial sums)
            return
        s = self.static
        if self.fused is not None:
            self.fused.forward()
            zp, vals = self.policy.zs[-1], self.value.zs[-1].view(-1)
            baseline, boot = vals[:n], vals[n:]
            engine.gae(s["truncation"], s["termination"], s["reward"], baseline.view(B, T), boot, cfg["gae_lambda"], cfg["discounting"],
                       vs=self.vs, adv=self.adv, stats=self.stats)
            engine.ppo_head(zp, s["raw_action"].view(n, -1), s["log_prob"].view(n), self.adv.view(n),
                            self.stats if cfg["normalize_advantage"] else None, self.vs.view(n), baseline, self.noise, self.dlogits,
                            self.dval_all[:n].view(n), self.losses, cfg["clipping_epsilon"], cfg["entropy_cost"], 1.0 / self.world)
            self.fused.backward()
            self.dw_all()     # weight gradients; its finishing launch: slice fold + bias gradients (+ the norm's partial sums)
            return
        # policy and value networks are independent until the loss head: two branches of the captured graph
        cur = torch.cuda.current_stream()
        if self.side is not None:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                zv = self.value.forward()
            zp = self.policy.forward()
            cur.wait_stream(self.side)
        else:
            zp = self.policy.forward()
            zv = self.value.forward()
        vals = zv.view(-1)
        baseline, boot = vals[:n], vals[n:]
        engine.gae(s["truncation"], s["termination"], s["reward"], baseline.view(B, T), boot, cfg["gae_lambda"], cfg["discounting"],
                   vs=self.vs, adv=self.adv, stats=self.stats)
        engine.ppo_head(zp, s["raw_action"].view(n, -1), s["log_prob"].view(n), self.adv.view(n),
                        self.stats if cfg["normalize_advantage"] else None, self.vs.view(n), baseline, self.noise, self.dlogits,
                        self.dval_all[:n].view(n), self.losses, cfg["clipping_epsilon"], cfg["entropy_cost"], 1.0 / self.world)
        if self.side is not None:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self.value.backward()
            self.policy.backward()
            cur.wait_stream(self.side)
        else:
            self.policy.backward()
            self.value.backward()

    @torch.no_grad()
    def _draw_noise(self):
        self.noise.normal_()

    @torch.no_grad()
    def load_rollout(self, prep: Dict[str, torch.Tensor]):
        """Indexed form: copies a PREPARED rollout (`prepare_rollout`: [N, T, ...] tensors) into the resident buffers the kernels index."""
        N = int(prep["reward"].shape[0])
        self._fit_rollout(N)
        for k, buf in self.roll.items():
            buf[:N].copy_(prep[k])
        self.n_loaded = N

    @torch.no_grad()
    def load_rollout_from(self, net: PPONetworks, data: Dict[str, torch.Tensor], cfg: Dict):
        """`prepare_rollout` written straight into the resident buffers (no intermediate copy of the 200 MB rollout)."""
        N = int(data["reward"].shape[0])
        self._fit_rollout(N)
        prepare_rollout(net, data, cfg, out={k: v[:N] for k, v in self.roll.items()})
        self.n_loaded = N

    def _fit_rollout(self, N: int):
        if N <= self.cap:
            return
        # a larger rollout than the resident copy was sized for: new buffers, new descriptors, new graphs (rare: construction-time sizing is the rule)
        od, pd, A = self.roll["obs"].shape[2], self.roll["priv"].shape[2], self.roll["raw_action"].shape[2]
        had_graph = self.graph_a is not None
        self.graph_a = self.graph_b = self.graph_k = None
        self._alloc_rollout(N, od, pd, A)
        n, B, T = self.B * self.T, self.B, self.T
        rows = dict(row_idx=self.sched, cursor=self.cursor, traj_len=T, n_main=n, n_traj=self.cap)
        self.policy.bind(self.roll["obs"].view(self.cap * T, od), self.dlogits, True, n, rows)
        self.value.bind(self.roll["priv"].view(self.cap * T, pd), self.dval_all, True, n + B, dict(rows, x_tail=self.roll["last_priv"]))
        self._build_fused()
        self.gae_head = engine.GaeHead(self.policy.zs[-1], self.value.zs[-1].view(-1), self.roll, self._pool, self.sched, self.cursor, self.dlogits,
                                       self.dval_all.view(-1), self.losses, B, T, self.cfg, 1.0 / self.world, adv=self.adv.view(-1), vs=self.vs.view(-1),
                                       stats=self.stats, loss_partials=self.loss_partials)
        if had_graph:
            self._capture()

    @torch.no_grad()
    def set_schedule(self, perms: torch.Tensor):
        """Indexed form: the trajectory indices of the next `perms.numel() / B` minibatch steps (the training step's shuffles, in order);
        resets the cursor and draws the entropy noise of all of them."""
        steps = perms.numel() // self.B
        if perms.numel() != steps * self.B or steps < 1 or steps > self.steps_cap:
            raise engine.OdkError(f"set_schedule: {perms.numel()} indices are not 1..{self.steps_cap} minibatches of {self.B} trajectories")
        self.sched[:perms.numel()].copy_(perms.reshape(-1))
        self.cursor.zero_()
        self._host_cursor, self._host_steps = 0, steps
        self._pool[:steps].normal_()
        if self.capture_allreduce and self.world > 1:
            # With the all-reduce captured inside the step graphs, `run()`'s choice between the K-step graph and single steps decides how many
            # collectives a replay issues: every rank must make the same choice, i.e. hold the same (K, schedule length).  One tiny all-reduce per
            # schedule (one per training step) -- a mismatch would otherwise show up as a hang inside a captured collective.
            import torch.distributed as dist
            mine = torch.tensor([float(getattr(self, "K", 0)), float(steps)], device=self.sched.device)
            lo, hi = mine.clone(), mine.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group); dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not (torch.equal(lo, mine) and torch.equal(hi, mine)):
                raise engine.OdkError(f"FlatLearner (captured all-reduce): ranks disagree on (steps per graph, schedule length): mine {mine.tolist()}, "
                                      f"min {lo.tolist()}, max {hi.tolist()}")

    NOISE_POOL = 128   # minibatch steps per refill (= steps per training step in the reference configuration)

    @torch.no_grad()
    def _noise_block(self):
        """Entropy-sample noise for the captured step: inside a graph `normal_` costs three launches per replay (the
        generator's seed / offset fills + the kernel); a pool refilled once per 128 steps costs none -- this step's slice is
        copied into the static noise buffer by the minibatch gather launch itself (a "direct" field of `odk_gather_rows`).
        Returns the slice's first row in the pool viewed as [NOISE_POOL * B, T * A]."""
        if self._gpool is None:
            self._gpool = torch.empty(self.NOISE_POOL, *self.noise.shape, device=self.noise.device)
            self._pool_k = self.NOISE_POOL
            self._gather = None
        if self._pool_k >= self.NOISE_POOL:
            self._gpool.normal_()
            self._pool_k = 0
        k = self._pool_k
        self._pool_k += 1
        return k * self.B

    @torch.no_grad()
    def _update(self):
        if self.fused is not None:
            engine.adam_clip_packed(self.flat_p, self.flat_g, self.m, self.v, self.acc, self.packed_f, self.packed_b, self.wtable,
                                    self.cfg["learning_rate"], self.cfg.get("max_grad_norm") or 0.0, norm_blocks=self.dw_all.norm_blocks,
                                    cursor=self.cursor if self.indexed else None, loss_partials=self.loss_partials if self.indexed else None,
                                    losses=self.losses if self.indexed else None)
        else:
            engine.adam_clip(self.flat_p, self.flat_g, self.m, self.v, self.acc, self.cfg["learning_rate"], self.cfg.get("max_grad_norm") or 0.0)

    @torch.no_grad()
    def sync_weights(self):
        """Rebuilds the packed weight copies from the flat parameter buffer.  The Adam launch keeps them current; call this after
        writing parameters from outside (checkpoint restore, tests) -- `sgd_epoch` does so once per training step."""
        if self.fused is not None:
            engine.pack_weights(self.flat_p, self.packed_f, self.packed_b, self.wtable)

    def _capture(self):
        keep = [t.clone() for t in (self.flat_p, self.m, self.v, self.acc)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # (no library GEMM is captured on the whole-network path: leave PyTorch's process-wide TunableOp switch alone there)
        tuned = _tunable(True) if (self.cfg.get("tune_gemms", True) and self.fused is None) else False
        with torch.cuda.stream(side):               # warm-up outside capture (hipBLASLt workspaces, allocator, GEMM tuning)
            for _ in range(2):
                self._draw_noise(); self._loss_and_grads()
                if self.capture_allreduce:      # (the communicator must exist before a capture can record a collective on it)
                    import torch.distributed as dist
                    dist.all_reduce(self.flat_g, group=self.group)
                self._update()
        torch.cuda.current_stream().wait_stream(side)
        if tuned:
            _tunable(False)                         # keep the selected kernels, never tune inside a capture
        for t, k in zip((self.flat_p, self.m, self.v, self.acc), keep):
            t.copy_(k)                              # the warm-up steps must not train
        self.sync_weights()
        self.losses.zero_()
        if self.indexed:
            self.cursor.zero_()                     # (the warm-up steps advanced it)
            self._host_cursor = 0
        # With a live process group its watchdog thread polls events while collectives are outstanding; under the default (global) capture
        # mode such a query from ANOTHER thread aborts the capture ("operation not permitted when stream is capturing": seen on the
        # one-rank RCCL group once 33 steps were captured instead of one).  So: nothing outstanding when the capture starts, and the
        # capture is thread-local.
        import torch.distributed as _dist
        live_pg = _dist.is_available() and _dist.is_initialized()
        if live_pg:
            torch.cuda.synchronize()
        cap = dict(capture_error_mode="thread_local") if live_pg else {}
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a, **cap):
            self._loss_and_grads()                  # the noise buffer is filled before each replay (load_minibatch)
            if not self.split_update:
                self._update()
            elif self.capture_allreduce:
                import torch.distributed as dist
                dist.all_reduce(self.flat_g, group=self.group)
                self._update()
        # K consecutive steps as ONE graph (indexed form: every step's launches are the same nodes -- the cursor moves on the device): a
        # replay per step left ~8 us of idle GPU between two graphs (the gather launch used to sit there), K = 32 leaves none
        self.graph_k, self.K = None, 0
        K = int(os.environ.get("ODK_LEARNER_STEPS_PER_GRAPH", "32"))
        if self.indexed and K > 1 and (not self.split_update or self.capture_allreduce):
            self.graph_k, self.K = torch.cuda.CUDAGraph(), K
            with torch.cuda.graph(self.graph_k, **cap):
                for _ in range(K):
                    self._loss_and_grads()
                    if self.capture_allreduce:
                        import torch.distributed as dist
                        dist.all_reduce(self.flat_g, group=self.group)
                    self._update()
        if self.split_update and not self.capture_allreduce:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, **cap):
                self._update()

    # ---- public ----
    def load_minibatch(self, data: Dict[str, torch.Tensor], idx: torch.Tensor):
        """ONE minibatch = trajectories `idx` of the prepared ([N, T, ...]) rollout tensors.  Indexed form: the rollout is copied into
        the resident buffers (every call: this is the convenience path of tests and tools -- a training step goes through
        `load_rollout_from` + `set_schedule` once and then only replays) and `idx` becomes a one-step schedule; gathered form: one
        `odk_gather_rows` launch for all eight fields into the static minibatch buffers."""
        if self.indexed:
            if not (idx.is_cuda and idx.dtype == torch.int64 and idx.numel() == self.B):
                raise engine.OdkError(f"load_minibatch: idx must be an int64 CUDA tensor of {self.B} trajectory numbers")
            self.load_rollout(data)
            self.sched[:self.B].copy_(idx)
            self.cursor.zero_()
            self._host_cursor, self._host_steps = 0, 1
            if self.graph_a is not None:
                self._pool[0].normal_()
            return
        keys = self.KEYS + ("last_priv",)
        base = [self._noise_block()] if self.graph_a is not None else []     # (may replace the pool: before the gather is built)
        srcs = tuple(data[k] for k in keys) + ((self._gpool,) if base else ())
        if self._gather is None or len(srcs) != len(self._gather_src) or any(a is not b for a, b in zip(self._gather_src, srcs)):
            direct = [(self._gpool.view(self.NOISE_POOL * self.B, -1), self.noise.view(self.B, -1))] if base else []
            self._gather = engine.RowGather([(data[k], self.static[k]) for k in keys], direct)
            self._gather_src = srcs
        self._gather(idx, base)

    def run(self, steps: int):
        """`steps` consecutive minibatch steps of the schedule: K-step graph replays while K steps are left, single steps for the rest."""
        while steps > 0:
            if self.graph_k is not None and steps >= self.K and self._host_cursor + self.K <= self._host_steps:
                self.graph_k.replay()
                self._host_cursor += self.K; self.nsteps += self.K; steps -= self.K
            else:
                self.step(); steps -= 1
        return self.losses

    def step(self):
        """One clipped-Adam step on the loaded minibatch.  The loss head ADDS this step's (total, policy, value, entropy)
        to `self.losses` (no sync); `metrics()` turns the sums into means."""
        if self.indexed:
            if self._host_cursor >= self._host_steps:
                raise engine.OdkError("FlatLearner.step: the schedule is used up (load_minibatch / set_schedule first)")
            if self.graph_a is None and self.sample_noise:
                self._draw_noise()                     # (before the cursor's host mirror moves: `noise` is the slot under it)
        if self.graph_a is not None:
            self.graph_a.replay()                       # (its entropy noise came with load_minibatch)
        else:
            if self.sample_noise and not self.indexed:
                self._draw_noise()
            self._loss_and_grads()
            if not self.split_update:
                self._update()
        if self.split_update and not (self.capture_allreduce and self.graph_a is not None):
            if self.world > 1 or self.group is not None:
                import torch.distributed as dist
                # RCCL: the collective runs on the process group's own stream; c10d makes that stream wait for the current
                # one (graph A) before it starts and, for a blocking call, the current stream wait for it afterwards, so
                # graph B is ordered behind the reduced gradient without any host synchronisation.
                dist.all_reduce(self.flat_g, group=self.group)   # gradients were pre-scaled by 1/world in the loss head
            if self.graph_b is not None:
                self.graph_b.replay()
            else:
                self._update()
        self.nsteps += 1
        if self.indexed:
            self._host_cursor += 1
        if _DEBUG_NONFINITE:
            self._debug_check()
        return self.losses

    def _debug_check(self):
        torch.cuda.synchronize()
        if bool(torch.isfinite(self.losses).all()) or getattr(self, "_reported", False):
            return
        self._reported = True
        print("NONFINITE losses", self.losses.tolist(), "stats", self.stats.tolist(), "acc", self.acc.tolist(), flush=True)
        for k, v in (self.roll if self.indexed else self.static).items():
            print("   static", k, bool(torch.isfinite(v).all()), float(v.abs().nan_to_num().max()), flush=True)
        for nm in ("noise", "vs", "adv", "dlogits", "dval_all", "flat_g", "flat_p", "m", "v"):
            t = getattr(self, nm)
            print("   ", nm, bool(torch.isfinite(t).all()), float(t.abs().nan_to_num().max()), flush=True)

    def last_step_losses(self):
        """(total, policy, value, entropy) of the most recent `_loss_and_grads()`: the indexed form's loss head leaves per-workgroup sums
        that the clip + Adam launch folds into `losses`; a caller that runs the loss without the update (tests) folds them here."""
        return self.loss_partials.sum(0) if self.indexed else self.losses

    def metrics(self, reset: bool = True):
        """Mean of the four losses over every minibatch step since the last reset, averaged over the data-parallel ranks
        (brax reports the mean over all num_updates_per_batch x num_minibatches steps of the epoch and pmean's it)."""
        l = self.losses / float(max(self.nsteps, 1))
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(l, group=self.group)
            l /= self.world
        if reset:
            self.losses.zero_()
            self.nsteps = 0
        return dict(total_loss=l[0], policy_loss=l[1], v_loss=l[2], entropy_loss=l[3])

    def close(self):
        """Drops the captured graphs and turns the process-wide TunableOp switch off again."""
        self.graph_a = self.graph_b = self.graph_k = None
        tunable_off()

    def optimizer_state(self):
        return dict(m=self.m.clone(), v=self.v.clone(), acc=self.acc.clone())

    def load_optimizer_state(self, st):
        self.m.copy_(st["m"]); self.v.copy_(st["v"]); self.acc.copy_(st["acc"])
        self.sync_weights()


class FusedPolicy:
    """Policy inference for a rollout through the whole-network kernel (csrc/odk_mlp.hip in inference mode: ONE launch for the
    four layers, nothing but the logits written) instead of four library GEMMs + three activation launches.  Holds its own
    forward-packed copy of the policy weights: `refresh()` rebuilds it from the current parameters (once per rollout -- the
    parameters do not move inside one).  `ok` is false for architectures the kernel is not built for."""

    def __init__(self, net: PPONetworks, rows: int):
        lins = list(net.policy.layers)
        dev = lins[0].weight.device
        self.net, self.rows, self.op, self.key = net, int(rows), None, None
        self.ok = (dev.type == "cuda" and _FUSED_MLP and len(lins) == 4 and tuple(l.out_features for l in lins[:3]) == engine.MLP_HIDDEN
                   and lins[0].in_features <= engine.MLP_MAX_IN and lins[3].out_features <= engine.MLP_MAX_OUT)
        if not self.ok:
            return
        self.tables = [engine.WeightTable([(0, l.out_features, l.in_features, False)]) for l in lins]
        self.wf = [torch.zeros(t.fwd_size, device=dev) for t in self.tables]
        self.nobwd = torch.zeros(4, device=dev)
        self.out = torch.empty(self.rows, lins[3].out_features, device=dev)

    @torch.no_grad()
    def refresh(self):
        for lin, t, wf in zip(self.net.policy.layers, self.tables, self.wf):
            engine.pack_weights(lin.weight.data.reshape(-1), wf, self.nobwd, t)

    @torch.no_grad()
    def __call__(self, obs):
        """Logits [rows, 2 A] of the RAW observations `obs` (contiguous [rows, in]); the normaliser (x - mean) / std runs inside
        the kernel's load.  The result is a persistent buffer: consume it before the next call."""
        nrm = self.net.norm_obs
        key = (obs.data_ptr(),) + tuple(l.bias.data_ptr() for l in self.net.policy.layers)     # FlatLearner re-homes the parameters once
        if key != self.key:
            if not (obs.is_contiguous() and tuple(obs.shape) == (self.rows, self.net.policy.layers[0].in_features)):
                raise engine.OdkError("FusedPolicy: contiguous [rows, in] observations expected")
            self.op = engine.FusedMLP([dict(x=obs, in_mean=nrm.mean, in_std=nrm.std, wf=self.wf, b=[l.bias.data for l in self.net.policy.layers],
                                            out=self.out)])
            self.key = key
        self.op.forward()
        return self.out


_FUSED_POLICIES = weakref.WeakKeyDictionary()


def fused_policy(net: PPONetworks, rows: int):
    """The net's cached `FusedPolicy` for `rows` observations per call, or None when the fused kernel does not apply."""
    cache = _FUSED_POLICIES.setdefault(net, {})   # weakly keyed by the module, not stored in it (deepcopy / pickling of the net stay clean)
    fp = cache.get(rows)
    if fp is None:
        fp = cache[rows] = FusedPolicy(net, rows)
    return fp if fp.ok else None


@torch.no_grad()
def prepare_rollout(net: PPONetworks, data: Dict[str, torch.Tensor], cfg: Dict, out: Dict[str, torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Per-training-step preprocessing shared by all 128 minibatch steps: observation normalisation (the
    normaliser is fixed during the SGD epochs, as in brax), reward scaling, termination = done & ~truncation.
    `out`: same-shaped tensors to write into (the learner's resident rollout) -- the same arithmetic, bit for bit, without the
    intermediate allocations."""
    if out is None:
        return dict(obs=net.norm_obs(data["obs"]), priv=net.norm_priv(data["priv"]), last_priv=net.norm_priv(data["last_priv"]),
                    raw_action=data["raw_action"],
                    log_prob=data["log_prob"], reward=data["reward"] * cfg["reward_scaling"],
                    termination=data["done"] * (1.0 - data["truncation"]), truncation=data["truncation"])
    for key, nrm in (("obs", net.norm_obs), ("priv", net.norm_priv), ("last_priv", net.norm_priv)):
        nrm(data[key], out=out[key])
    out["raw_action"].copy_(data["raw_action"]); out["log_prob"].copy_(data["log_prob"]); out["truncation"].copy_(data["truncation"])
    torch.mul(data["reward"], cfg["reward_scaling"], out=out["reward"])
    torch.sub(1.0, data["truncation"], out=out["termination"]); out["termination"].mul_(data["done"])
    return out
