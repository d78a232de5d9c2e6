"""ctypes binding of libodk.so (include/odk.h): the MI355X-native batched env engine.

This is the host side of the drop-in boundary: it owns no arithmetic.  PyTorch-ROCm is
used for device memory and streams only; every number comes out of the HIP kernels in csrc/.
There is no CPU fallback: importing works without the library, but constructing a `Batch` raises
if `csrc/libodk.so` is missing or no HIP device is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

from .model import Model, asset_path

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("ODK_LIB", os.path.join(_CSRC, "libodk.so"))  # ODK_LIB: e.g. the -DODK_PROFILE build

NOBS, NPRIV, NMETRIC, NU = 101, 212, 8, 14      # the duck's sizes; a Batch reports its model's own (nobs, npriv, model.nu)
ADAM_ACC_FLOATS = 2 + 1024   # ODK_ADAM_ACC_FLOATS (include/odk.h)
METRIC_NAMES = ("reward/tracking_lin_vel", "reward/tracking_ang_vel", "cost/torques", "cost/action_rate", "cost/stand_still",
                "reward/alive", "reward/imitation", "swing_peak")

PARAM_BODY_MASS, PARAM_BODY_IPOS_TORSO, PARAM_DOF_FRICTIONLOSS, PARAM_DOF_ARMATURE, PARAM_QPOS0, PARAM_KP = range(6)


class EnvConfig(C.Structure):
    """odk_env_config (include/odk.h) == default_config() of reference joystick.py:49-102."""
    _fields_ = [
        ("ctrl_dt", C.c_float), ("action_scale", C.c_float), ("dof_vel_scale", C.c_float), ("max_motor_velocity", C.c_float),
        ("noise_level", C.c_float), ("noise_gyro", C.c_float), ("noise_accelerometer", C.c_float), ("noise_gravity", C.c_float),
        ("noise_joint_vel", C.c_float),
        ("qpos_noise_scale", C.c_float * 16), ("reward_scales", C.c_float * 7), ("tracking_sigma", C.c_float),
        ("push_enable", C.c_float), ("push_interval_range", C.c_float * 2), ("push_magnitude_range", C.c_float * 2),
        ("cmd_range", (C.c_float * 2) * 7),
        ("use_imitation", C.c_int32), ("use_motor_speed_limits", C.c_int32), ("autoreset", C.c_int32), ("episode_length", C.c_int32),
        ("n_substeps", C.c_int32), ("lanes_per_env", C.c_int32), ("env_kind", C.c_int32), ("reset_base_qvel", C.c_float),
        ("hfield_up_normals_only", C.c_int32),
    ]


class Outputs(C.Structure):
    _fields_ = [("obs_dev", C.c_void_p), ("priv_dev", C.c_void_p), ("reward_dev", C.c_void_p), ("done_dev", C.c_void_p),
                ("truncation_dev", C.c_void_p), ("metrics_dev", C.c_void_p)]


class MlpDesc(C.Structure):
    """odk_mlp_desc (include/odk.h)."""
    _fields_ = [("x", C.c_void_p), ("in_mean", C.c_void_p), ("in_std", C.c_void_p), ("wf", C.c_void_p * 4), ("wb", C.c_void_p * 4), ("b", C.c_void_p * 4), ("xp", C.c_void_p), ("h", C.c_void_p * 3),
                ("g", C.c_void_p * 3), ("out", C.c_void_p), ("dout", C.c_void_p), ("doutp", C.c_void_p), ("dz", C.c_void_p * 3),
                ("bias_partial", C.c_void_p * 4), ("n", C.c_int), ("n_in", C.c_int), ("n_out", C.c_int),
                ("row_idx", C.c_void_p), ("cursor", C.c_void_p), ("x_tail", C.c_void_p), ("traj_len", C.c_int), ("n_main", C.c_int), ("n_traj", C.c_int)]


class StepTail(C.Structure):
    """odk_step_tail (include/odk.h)."""
    _fields_ = [("cursor_dev", C.c_void_p), ("loss_partials_dev", C.c_void_p), ("n_loss_partials", C.c_int), ("losses_dev", C.c_void_p)]


GAE_HEAD_SAMPLES = 32     # ODK_GAE_HEAD_SAMPLES: samples per workgroup of odk_ppo_gae_head (= entries of its loss partials per 32 samples)


class GaeHeadArgs(C.Structure):
    """odk_gae_head_args (include/odk.h)."""
    _fields_ = [(k, C.c_void_p) for k in ("logits", "values", "raw_action", "old_log_prob", "reward", "termination", "truncation", "noise", "row_idx", "cursor",
                                         "dlogits", "dvalues", "losses", "loss_partials", "adv_out", "vs_out", "stats_out")] + \
               [(k, C.c_int) for k in ("B", "T", "action_size", "n_traj", "normalize_advantage")] + \
               [(k, C.c_float) for k in ("gae_lambda", "discount", "clipping_epsilon", "entropy_cost", "grad_scale")]


class GradFinish(C.Structure):
    """odk_grad_finish (include/odk.h)."""
    _fields_ = [("bias_partial", C.c_void_p * 8), ("bias_grad", C.c_void_p * 8), ("width", C.c_int * 8), ("nblk", C.c_int * 8), ("nbias", C.c_int),
                ("sq_partials_dev", C.c_void_p), ("step_counter_dev", C.c_void_p), ("nblocks", C.c_int)]


class WeightTableC(C.Structure):
    """odk_weight_table (include/odk.h)."""
    _fields_ = [("count", C.c_int), ("off", C.c_longlong * 8), ("rows", C.c_int * 8), ("cols", C.c_int * 8), ("fwd_off", C.c_longlong * 8),
                ("bwd_off", C.c_longlong * 8)]


MLP_HIDDEN = (512, 256, 128)     # ODK_MLP_H1..3: the hidden widths the fused network kernels are built for
MLP_MAX_IN, MLP_MAX_OUT, MLP_TILE = 224, 32, 16


class OdkError(RuntimeError):
    pass


def build_library(force: bool = False) -> str:
    """Compiles csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".hip", ".h", ".inc")) or f == "Makefile"]
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "odk.h"))
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    subprocess.check_call(["make", "-C", _CSRC, "-B", "-s", "libodk.so"])
    return LIB_PATH


_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OdkError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the env engine)")
    # torch first: its wheel carries its own HIP runtime (torch/lib/libamdhip64.so), and the process must end up with ONE
    # runtime -- libodk.so takes torch's device pointers and streams.  Loaded the other way round (dlopen here before any
    # `import torch`, e.g. build() then smoke() in one process), libodk.so pulls /opt/rocm's copy in first and its
    # hipSetDevice then reports "no ROCm-capable device" on a box whose GPU torch drives happily.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    P, PP = C.c_void_p, C.POINTER(C.c_void_p)
    FP, DP = C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.odk_last_error.restype = C.c_char_p
    L.odk_default_config.argtypes = [C.POINTER(EnvConfig)]
    L.odk_default_config_standing.argtypes = [C.POINTER(EnvConfig)]
    L.odk_obs_sizes.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.odk_model_load.argtypes = [C.c_char_p, C.c_uint64, PP]
    L.odk_model_free.argtypes = [P]
    L.odk_model_dims.argtypes = [P] + [C.POINTER(C.c_int)] * 4
    L.odk_model_obs_sizes.argtypes = [P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.odk_batch_lanes.argtypes = [P]
    L.odk_model_reduced.argtypes = [P] + [C.POINTER(C.c_int)] * 6
    L.odk_model_env_lds_floats.argtypes = [P]
    L.odk_batch_create.argtypes = [P, C.POINTER(EnvConfig), C.c_int, C.c_int, FP, DP, C.c_int, DP, C.c_int, DP, C.c_int, DP, C.c_int, PP]
    L.odk_batch_destroy.argtypes = [P]
    L.odk_batch_set_config.argtypes = [P, C.POINTER(EnvConfig)]
    L.odk_batch_set_param.argtypes = [P, C.c_int, FP, C.c_int]
    L.odk_reset.argtypes = [P, C.c_uint32, C.c_uint32, C.POINTER(Outputs), P]
    L.odk_step.argtypes = [P, P, C.POINTER(Outputs), P]
    L.odk_physics_step.argtypes = [P, P, C.c_int, P]
    L.odk_batch_get_state.argtypes = [P, FP, FP, FP]
    L.odk_batch_set_state.argtypes = [P, FP, FP, FP]
    L.odk_batch_get_debug.argtypes = [P, FP, FP, FP, FP]
    L.odk_set_debug_dump.argtypes = [C.c_int]
    L.odk_batch_lds_size.argtypes = [P]
    L.odk_batch_get_lds.argtypes = [P, FP]
    L.odk_lds_offset.argtypes = [P, C.c_char_p]
    L.odk_batch_record_size.argtypes = [P]
    L.odk_batch_get_records.argtypes = [P, FP]
    L.odk_batch_set_records.argtypes = [P, FP]
    L.odk_batch_timing.argtypes = [P, C.c_int, FP, C.POINTER(C.c_int)]
    L.odk_record_field.argtypes = [P, C.c_char_p] + [C.POINTER(C.c_int)] * 3
    L.odk_gae.argtypes = [P, P, P, P, P, P, P, P, C.c_int, C.c_int, C.c_float, C.c_float, P]
    L.odk_ppo_head.argtypes = [P] * 11 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, P]
    L.odk_policy_sample.argtypes = [P, P, P, P, P, C.c_int, C.c_int, P]
    L.odk_adam_clip.argtypes = [P, P, P, P, P, C.c_longlong] + [C.c_float] * 5 + [P]
    L.odk_silu_bwd_colsum.argtypes = [P, P, P, P, P, C.c_int, C.c_int, P]
    L.odk_colsum_partial.argtypes = [P, P, C.c_int, C.c_int, P]
    L.odk_colsum_finalize.argtypes = [PP, PP, C.POINTER(C.c_int), C.c_int, C.c_int, P]
    L.odk_dw_gemm.argtypes = [PP, PP, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_int), C.c_int, P, C.c_longlong, P,
                              C.POINTER(GradFinish), P]
    IP, LP = C.POINTER(C.c_int), C.POINTER(C.c_longlong)
    L.odk_mlp_forward.argtypes = [C.POINTER(MlpDesc), C.c_int, P]
    L.odk_mlp_backward.argtypes = [C.POINTER(MlpDesc), C.c_int, P]
    L.odk_mlp_set_profile.argtypes = [P]
    L.odk_mlp_set_profile.restype = None
    L.odk_pack_weights.argtypes = [P, C.c_longlong, P, C.c_longlong, P, C.c_longlong, C.POINTER(WeightTableC), P]
    L.odk_adam_clip_packed.argtypes = [P, P, P, P, P, C.c_longlong] + [C.c_float] * 5 + [P, C.c_longlong, P, C.c_longlong, C.POINTER(WeightTableC), C.c_int, P]
    L.odk_adam_clip_packed_tail.argtypes = [P, P, P, P, P, C.c_longlong] + [C.c_float] * 5 + [P, C.c_longlong, P, C.c_longlong, C.POINTER(WeightTableC), C.c_int,
                                            C.POINTER(StepTail), P]
    L.odk_ppo_gae_head.argtypes = [C.POINTER(GaeHeadArgs), P]
    L.odk_col_moments.argtypes = [P, C.c_longlong, C.c_int, C.c_int, P, P]
    L.odk_moments_update.argtypes = [P, C.c_int, C.c_int, C.c_longlong, P, P, P, P, C.c_float, C.c_float, P]
    L.odk_colsum_fold.argtypes = [PP, PP, IP, IP, C.c_int, P]
    L.odk_gather_rows.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.c_int, P, C.c_int, C.c_longlong, P]
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "odk_last_error", "odk_default_config", "odk_default_config_standing", "odk_obs_sizes", "odk_model_load", "odk_model_free", "odk_model_dims", "odk_model_obs_sizes", "odk_batch_lanes", "odk_model_reduced",
    "odk_model_env_lds_floats", "odk_batch_create",
    "odk_batch_destroy", "odk_batch_set_config", "odk_batch_set_param", "odk_reset", "odk_step", "odk_physics_step",
    "odk_batch_get_state", "odk_batch_set_state", "odk_batch_get_debug", "odk_set_debug_dump", "odk_batch_lds_size",
    "odk_batch_get_lds", "odk_lds_offset", "odk_batch_record_size", "odk_batch_get_records", "odk_batch_set_records", "odk_batch_timing", "odk_gae", "odk_ppo_head",
    "odk_policy_sample", "odk_adam_clip", "odk_silu_bwd_colsum", "odk_colsum_partial", "odk_colsum_finalize", "odk_gather_rows", "odk_dw_gemm",
    "odk_mlp_forward", "odk_mlp_backward", "odk_mlp_set_profile", "odk_pack_weights", "odk_adam_clip_packed", "odk_colsum_fold",
    "odk_adam_clip_packed_tail", "odk_ppo_gae_head", "odk_col_moments", "odk_moments_update")


def _chk(rc: int):
    if rc != 0:
        raise OdkError(f"odk error {rc}: {load_library().odk_last_error().decode()}")


def default_config(standing: bool = False) -> EnvConfig:
    cfg = EnvConfig()
    L = load_library()
    (L.odk_default_config_standing if standing else L.odk_default_config)(C.byref(cfg))
    return cfg


def obs_sizes(env_kind: int):
    """(nobs, npriv) row strides of the observation outputs for an env kind (0 Joystick, 1 Standing) -- of the duck (14 actuators)."""
    a, b = C.c_int(0), C.c_int(0)
    load_library().odk_obs_sizes(int(env_kind), C.byref(a), C.byref(b))
    return a.value, b.value


def model_obs_sizes(model: Model, env_kind: int = 0):
    """(nobs, npriv) of `model`'s env kernels (`odk_model_obs_sizes`): the reference's layout with the robot's actuator count; host-only."""
    L = load_library()
    blob = model.blob()
    h = C.c_void_p()
    _chk(L.odk_model_load(blob, len(blob), C.byref(h)))
    try:
        a, b = C.c_int(0), C.c_int(0)
        _chk(L.odk_model_obs_sizes(h, int(env_kind), C.byref(a), C.byref(b)))
        return a.value, b.value
    finally:
        L.odk_model_free(h)


def model_reduction(model: Model) -> Dict:
    """The twin-dof reduction the kernels use for `model` (odk_model_reduced) + its LDS footprint; host-only, no GPU."""
    L = load_library()
    blob = model.blob()
    h = C.c_void_p()
    _chk(L.odk_model_load(blob, len(blob), C.byref(h)))
    try:
        ints = [C.c_int(0) for _ in range(4)]
        main, twin = (C.c_int * 32)(), (C.c_int * 32)()
        _chk(L.odk_model_reduced(h, *[C.byref(i) for i in ints], main, twin))
        nvr = ints[1].value
        return dict(paired=ints[0].value, nvr=nvr, nMr=ints[2].value, nHr=ints[3].value, main=list(main)[:nvr], twin=list(twin)[:nvr],
                    env_lds_floats=L.odk_model_env_lds_floats(h))
    finally:
        L.odk_model_free(h)


def load_prm() -> Dict[str, np.ndarray]:
    z = np.load(asset_path("prm_table.npz"))
    return {k: z[k] for k in z.files}


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _stream(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f32c(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 4):
            raise OdkError("learner kernels take contiguous float32 CUDA tensors")


def gae(truncation, termination, rewards, values, bootstrap, lambda_: float, discount: float, vs=None, adv=None, stats=None):
    """compute_gae on the device ([B, T] contiguous float32 CUDA tensors) -> (vs, advantages); one HIP launch.
    `truncation` / `termination` are flags: 0 = clear, anything else = set.
    `stats` (2 floats, optional) receives the advantage mean and 1/(std + 1e-8)."""
    import torch
    B, T = rewards.shape
    vs = torch.empty_like(rewards) if vs is None else vs
    adv = torch.empty_like(rewards) if adv is None else adv
    _f32c(truncation, termination, rewards, values, bootstrap, vs, adv, stats)
    _chk(load_library().odk_gae(_ptr(truncation), _ptr(termination), _ptr(rewards), _ptr(values), _ptr(bootstrap), _ptr(vs), _ptr(adv),
                                _ptr(stats), B, T, lambda_, discount, _stream(rewards)))
    return vs, adv


def ppo_head(logits, raw_action, old_log_prob, adv, stats, vs, baseline, noise, dlogits, dbaseline, losses, clipping_epsilon: float,
             entropy_cost: float, grad_scale: float = 1.0):
    """Fused PPO loss head (forward + gradients w.r.t. logits and baseline); `losses` must be zeroed by the caller."""
    n, a2 = logits.shape
    _f32c(logits, raw_action, old_log_prob, adv, stats, vs, baseline, noise, dlogits, dbaseline, losses)
    _chk(load_library().odk_ppo_head(_ptr(logits), _ptr(raw_action), _ptr(old_log_prob), _ptr(adv), _ptr(stats), _ptr(vs), _ptr(baseline),
                                     _ptr(noise), _ptr(dlogits), _ptr(dbaseline), _ptr(losses), n, a2 // 2, clipping_epsilon, entropy_cost,
                                     grad_scale, _stream(logits)))


def adam_clip(params, grads, m, v, acc, lr: float, max_grad_norm: float = 0.0, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """optax clip_by_global_norm + adam on flat float32 buffers; acc = ADAM_ACC_FLOATS scratch (acc[1] = step count)."""
    _f32c(params, grads, m, v, acc)
    if acc.numel() < ADAM_ACC_FLOATS:
        raise OdkError("adam_clip: acc needs ADAM_ACC_FLOATS floats")
    _chk(load_library().odk_adam_clip(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(acc), params.numel(), lr, b1, b2, eps,
                                      max_grad_norm or 0.0, _stream(params)))


def policy_sample(logits, noise, out=None):
    """(raw_action, action, log_prob) of the tanh-normal policy for logits [n, 2A] and standard-normal noise [n, A];
    `out`: the three result tensors to write into (contiguous [n, A], [n, A], [n]) instead of fresh ones."""
    import torch
    n, A = noise.shape
    _f32c(logits, noise)
    if out is not None:
        raw, act, logp = out
        _f32c(raw, act, logp)
        if raw.numel() != n * A or act.numel() != n * A or logp.numel() != n:
            raise OdkError("policy_sample: out tensors of the wrong size")
    else:
        raw, act, logp = torch.empty_like(noise), torch.empty_like(noise), torch.empty(n, device=noise.device)
    _chk(load_library().odk_policy_sample(_ptr(logits), _ptr(noise), _ptr(raw), _ptr(act), _ptr(logp), n, A, _stream(noise)))
    return raw, act, logp


def silu_bwd_colsum(dh, z, dz, colsum, partial):
    """dz = dh * silu'(z) ([n, w]) and colsum = dz.sum(0) in one pass; partial: scratch of ceil(n / 64) * w floats.
    colsum=None leaves the per-tile sums in `partial` for a later `ColsumFinalize` (one launch for several layers)."""
    n, w = z.shape
    _f32c(dh, z, dz, colsum, partial)
    if partial.numel() < ((n + 63) // 64) * w:
        raise OdkError("silu_bwd_colsum: partial scratch too small")
    _chk(load_library().odk_silu_bwd_colsum(_ptr(dh), _ptr(z), _ptr(dz), _ptr(colsum), _ptr(partial), n, w, _stream(z)))


def colsum_partial(x, partial):
    """Tile sums of x ([n, w]) for a later `ColsumFinalize` (partial: ceil(n / 64) * w floats)."""
    n, w = x.shape
    _f32c(x, partial)
    if partial.numel() < ((n + 63) // 64) * w:
        raise OdkError("colsum_partial: partial scratch too small")
    _chk(load_library().odk_colsum_partial(_ptr(x), _ptr(partial), n, w, _stream(x)))


class ColsumFinalize:
    """colsum[f] = fold of partial[f] for a fixed set of layers in one launch (`odk_colsum_finalize`); `n` = rows of dz."""

    def __init__(self, pairs, n: int):
        k = len(pairs)
        self.k, self.n, self.keep = k, int(n), pairs
        for p_, o_ in pairs:
            _f32c(p_, o_)
            if p_.numel() < ((n + 63) // 64) * o_.numel():
                raise OdkError("ColsumFinalize: partial scratch too small")
        self.partial = (C.c_void_p * k)(*[p_.data_ptr() for p_, _ in pairs])
        self.out = (C.c_void_p * k)(*[o_.data_ptr() for _, o_ in pairs])
        self.w = (C.c_int * k)(*[int(o_.numel()) for _, o_ in pairs])

    def __call__(self):
        _chk(load_library().odk_colsum_finalize(self.partial, self.out, self.w, self.k, self.n, _stream(self.keep[0][0])))


def quad_rows(n: int) -> int:
    """Rows of a quad-row buffer for n samples: n rounded up to whole 16-sample tiles (a multiple of 8, as odk_dw_gemm wants)."""
    return (int(n) + MLP_TILE - 1) // MLP_TILE * MLP_TILE


def quad_pack(t):
    """[n, width] -> the quad-row layout [np / 4][width][4] the fused network / weight-gradient kernels use (flat tensor, rows
    past n zero).  Plain torch ops: tests and tools."""
    import torch
    n, w = t.shape
    full = torch.zeros(quad_rows(n), w, device=t.device, dtype=t.dtype)
    full[:n] = t
    return full.view(-1, 4, w).permute(0, 2, 1).contiguous().reshape(-1)


def quad_unpack(buf, n: int, width: int):
    """Inverse of `quad_pack`: the first n rows as [n, width]."""
    return buf.view(-1, width, 4).permute(0, 2, 1).reshape(-1, width)[:n]


class DwGemm:
    """Weight gradients out[off_l : off_l + n_out n_in] = dz_l^T h_l of up to 8 layers in one launch (`odk_dw_gemm`, f32
    matrix cores, split over `kslices` row slices folded in a fixed order).  `layers`: [(dz, h, n_out, n_in, offset in
    `flat_out`)] with dz / h flat QUAD-ROW buffers ([np / 4][width][4], `quad_pack`; np a multiple of 8, >= 8 * kslices);
    `workspace`: kslices * workspace_stride(flat_out.numel()) floats.
    `bias`: [(tile sums [nblk, width], bias gradient [width], nblk)] folded by the same finishing launch (what `ColsumFold` does);
    `acc`: the Adam scratch -- the finishing launch then also leaves the partial sums of the squared gradient norm in acc[2:] and
    advances the step count acc[1]; `norm_blocks` is what `adam_clip_packed` wants to hear about it."""

    def __init__(self, layers, flat_out, workspace, kslices: int = 8, bias=(), acc=None):
        k = len(layers)
        _f32c(flat_out, workspace, *[t for dz, h, _, _, _ in layers for t in (dz, h)])
        if k > 8 or kslices % 8 != 0:
            raise OdkError("DwGemm: at most 8 layers, kslices a multiple of 8")
        rows = []
        for dz, h, no, ni, o in layers:
            np_ = dz.numel() // int(no)
            if dz.numel() != np_ * no or h.numel() != np_ * ni or np_ % 8 or np_ // 8 < 2 * kslices:
                raise OdkError("DwGemm: dz / h must be quad-row buffers of the same row count (a multiple of 8, >= 16 * kslices)")
            if int(o) % 4 or (int(no) * int(ni)) % 4:
                raise OdkError("DwGemm: offsets and element counts must be multiples of 4")
            rows.append(np_)
        stride = self.workspace_stride(flat_out.numel())
        if workspace.numel() < kslices * stride:
            raise OdkError("DwGemm: workspace too small (kslices * workspace_stride(flat_out.numel()) floats)")
        self.keep = (layers, flat_out, workspace)
        self.k, self.kslices, self.stride = k, int(kslices), stride
        self.n = (C.c_int * k)(*rows)
        self.dz = (C.c_void_p * k)(*[l[0].data_ptr() for l in layers])
        self.h = (C.c_void_p * k)(*[l[1].data_ptr() for l in layers])
        self.n_out = (C.c_int * k)(*[int(l[2]) for l in layers])
        self.n_in = (C.c_int * k)(*[int(l[3]) for l in layers])
        self.off = (C.c_longlong * k)(*[int(l[4]) for l in layers])
        self.finish, self.norm_blocks = None, 0
        if bias or acc is not None:
            if len(bias) > 8:
                raise OdkError("DwGemm: at most 8 bias gradients")
            f = GradFinish()
            f.nbias = len(bias)
            for i, (part, grad, nblk) in enumerate(bias):
                _f32c(part, grad)
                if part.numel() < int(nblk) * grad.numel():
                    raise OdkError("DwGemm: bias tile sums too small")
                f.bias_partial[i], f.bias_grad[i], f.width[i], f.nblk[i] = part.data_ptr(), grad.data_ptr(), int(grad.numel()), int(nblk)
            if acc is not None:
                _f32c(acc)
                if acc.numel() < ADAM_ACC_FLOATS:
                    raise OdkError("DwGemm: acc needs ADAM_ACC_FLOATS floats")
                f.sq_partials_dev, f.step_counter_dev = acc.data_ptr() + 8, acc.data_ptr() + 4
            self.finish, self.keep_finish = f, (bias, acc)

    @staticmethod
    def workspace_stride(numel: int) -> int:
        return (int(numel) + 3) // 4 * 4

    def __call__(self):
        _, flat_out, ws = self.keep
        _chk(load_library().odk_dw_gemm(self.dz, self.h, self.n_out, self.n_in, self.off, self.k, self.n, self.kslices, _ptr(ws), self.stride,
                                        _ptr(flat_out), C.byref(self.finish) if self.finish is not None else None, _stream(flat_out)))
        if self.finish is not None and self.finish.sq_partials_dev:
            self.norm_blocks = int(self.finish.nblocks)


def _pad16(k: int) -> int:
    return (int(k) + 15) // 16 * 16


class WeightTable:
    """Where the weight matrices sit inside a flat parameter buffer -- [(float offset, rows = n_out, cols = n_in, backward copy?)],
    at most 8 -- and inside the two packed buffers the fused network kernels read (layouts: include/odk.h, odk_mlp_desc).
    `fwd_size` / `bwd_size`: floats to allocate (zero-initialised) for the packed buffers."""

    def __init__(self, entries):
        if len(entries) > 8:
            raise OdkError("WeightTable: at most 8 weights")
        self.entries = [(int(o), int(r), int(c), bool(bw)) for o, r, c, bw in entries]
        t = WeightTableC()
        t.count = len(entries)
        fo = bo = 0
        self.fwd, self.bwd = [], []
        for k, (o, r, c, bw) in enumerate(self.entries):
            t.off[k], t.rows[k], t.cols[k] = o, r, c
            t.fwd_off[k] = fo; self.fwd.append((fo, _pad16(c) * r)); fo += _pad16(c) * r
            if bw:
                t.bwd_off[k] = bo; self.bwd.append((bo, _pad16(r) * c)); bo += _pad16(r) * c
            else:
                t.bwd_off[k] = -1; self.bwd.append(None)
        self.c, self.fwd_size, self.bwd_size = t, fo, max(bo, 4)

    def fwd_view(self, buf, k):
        o, n = self.fwd[k]
        return buf[o:o + n]

    def bwd_view(self, buf, k):
        if self.bwd[k] is None:
            return None
        o, n = self.bwd[k]
        return buf[o:o + n]


def pack_weights(params, fwd_packed, bwd_packed, table: WeightTable):
    """(Re)builds the packed weight copies from the flat parameter buffer (`odk_pack_weights`); padding is left untouched."""
    _f32c(params, fwd_packed, bwd_packed)
    if fwd_packed.numel() < table.fwd_size or bwd_packed.numel() < table.bwd_size:
        raise OdkError("pack_weights: packed buffers too small")
    _chk(load_library().odk_pack_weights(_ptr(params), params.numel(), _ptr(fwd_packed), fwd_packed.numel(), _ptr(bwd_packed), bwd_packed.numel(),
                                         C.byref(table.c), _stream(params)))


def adam_clip_packed(params, grads, m, v, acc, fwd_packed, bwd_packed, table: WeightTable, lr: float, max_grad_norm: float = 0.0, b1: float = 0.9,
                     b2: float = 0.999, eps: float = 1e-8, norm_blocks: int = 0, cursor=None, loss_partials=None, losses=None):
    """`adam_clip` that also writes every updated weight to its places in the packed copies (`odk_adam_clip_packed`).
    `norm_blocks` > 0: the gradient's finishing launch (`DwGemm(..., acc=acc)`) already left that many partial sums of the
    squared norm in acc[2:] and advanced the step count: no norm launch here.  End-of-step duties of the same launch
    (`odk_adam_clip_packed_tail`, all optional): `cursor` (int32 CUDA tensor of one element) -- the minibatch cursor of the learner's
    schedule, advanced by one; `loss_partials` ([w, 4], what `GaeHead` leaves) folded into `losses` ([4], +=) in workgroup order."""
    _f32c(params, grads, m, v, acc, fwd_packed, bwd_packed)
    if acc.numel() < ADAM_ACC_FLOATS or fwd_packed.numel() < table.fwd_size or bwd_packed.numel() < table.bwd_size:
        raise OdkError("adam_clip_packed: acc needs ADAM_ACC_FLOATS floats, the packed buffers table.fwd_size / bwd_size")
    if cursor is not None and not (cursor.is_cuda and cursor.numel() == 1 and cursor.element_size() == 4 and not cursor.dtype.is_floating_point):
        raise OdkError("adam_clip_packed: cursor must be a one-element int32 CUDA tensor")
    tail = None
    if cursor is not None or loss_partials is not None:
        _f32c(loss_partials, losses)
        if loss_partials is not None and (losses is None or losses.numel() < 4 or loss_partials.numel() % 4):
            raise OdkError("adam_clip_packed: loss_partials [w, 4] need losses [4]")
        tail = StepTail(None if cursor is None else cursor.data_ptr(), None if loss_partials is None else loss_partials.data_ptr(),
                        0 if loss_partials is None else loss_partials.numel() // 4, None if losses is None else losses.data_ptr())
    _chk(load_library().odk_adam_clip_packed_tail(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(acc), params.numel(), lr, b1, b2, eps,
                                                  max_grad_norm or 0.0, _ptr(fwd_packed), fwd_packed.numel(), _ptr(bwd_packed), bwd_packed.numel(),
                                                  C.byref(table.c), int(norm_blocks), C.byref(tail) if tail is not None else None, _stream(params)))


def col_moments(x, slices: int = 1024):
    """(column sums, column sums of squares) of a contiguous float32 CUDA matrix x [rows, w], float64, in one pass (`odk_col_moments` + a
    fixed-order fold of its row slices): the observation normaliser's batch statistics."""
    import torch
    _f32c(x)
    rows, w = x.shape
    slices = max(1, min(int(slices), 1024, int(rows)))
    part = torch.empty(slices, 2, w, dtype=torch.float64, device=x.device)
    _chk(load_library().odk_col_moments(_ptr(x), int(rows), int(w), slices, C.c_void_p(part.data_ptr()), _stream(x)))
    tot = part.sum(0)
    return tot[0], tot[1]


def running_stats_update(x, count, mean, summed_variance, std, std_min: float, std_max: float, slices: int = 1024):
    """brax running_statistics.update of (count float64 [], mean / summed_variance / std float32 [w]) with the batch x [rows, w] (contiguous float32
    CUDA), in place, three launches and no host arithmetic (`odk_col_moments` + `odk_moments_update`)."""
    import torch
    _f32c(x, mean, summed_variance, std)
    rows, w = x.shape
    if not (count.is_cuda and count.dtype == torch.float64 and count.numel() == 1):
        raise OdkError("running_stats_update: count must be a float64 CUDA scalar")
    slices = max(1, min(int(slices), 1024, int(rows)))
    part = torch.empty(slices, 2, w, dtype=torch.float64, device=x.device)
    L = load_library()
    _chk(L.odk_col_moments(_ptr(x), int(rows), int(w), slices, C.c_void_p(part.data_ptr()), _stream(x)))
    _chk(L.odk_moments_update(C.c_void_p(part.data_ptr()), slices, int(w), int(rows), C.c_void_p(count.data_ptr()), _ptr(mean), _ptr(summed_variance), _ptr(std),
                              float(std_min), float(std_max), _stream(x)))


class GaeHead:
    """GAE + advantage statistics + PPO loss head of a minibatch in ONE launch, the rollout read through the schedule's trajectory
    indices (`odk_ppo_gae_head`; include/odk.h names every argument).  Built once over fixed tensors:
      logits [n, 2A], values [n + B], rollout tensors raw_action [n_traj, T, A], log_prob / reward / termination / truncation
      [n_traj, T], noise [steps, n, A], schedule int64 [steps * B], cursor int32 [1], dlogits [n, 2A], dvalues [>= n], losses [4];
      optional adv / vs [n] and stats [2] (copies of the intermediate results); `loss_partials` ([ceil(n / 32), 4]): the per-workgroup loss
      sums go there instead of float atomics on `losses` (`adam_clip_packed(loss_partials=..., losses=...)` folds them)."""

    def __init__(self, logits, values, rollout: dict, noise, schedule, cursor, dlogits, dvalues, losses, B: int, T: int, cfg: dict, grad_scale: float = 1.0,
                 adv=None, vs=None, stats=None, loss_partials=None):
        import torch
        n, A = B * T, logits.shape[1] // 2
        fl = [logits, values, noise, dlogits, dvalues, losses] + [rollout[k] for k in ("raw_action", "log_prob", "reward", "termination", "truncation")]
        _f32c(*fl, adv, vs, stats, loss_partials)
        if loss_partials is not None and loss_partials.numel() < 4 * ((n + GAE_HEAD_SAMPLES - 1) // GAE_HEAD_SAMPLES):
            raise OdkError("GaeHead: loss_partials needs 4 floats per 32 samples")
        n_traj = int(rollout["reward"].shape[0])
        if tuple(logits.shape) != (n, 2 * A) or values.numel() < n + B or dlogits.numel() != n * 2 * A or dvalues.numel() < n or losses.numel() < 4:
            raise OdkError("GaeHead: logits [B T, 2 A], values [B T + B], dlogits like logits, dvalues [>= B T], losses [4]")
        if any(int(rollout[k].shape[0]) != n_traj or int(rollout[k][0].numel()) != T * (A if k == "raw_action" else 1)
               for k in ("raw_action", "log_prob", "reward", "termination", "truncation")):
            raise OdkError("GaeHead: rollout tensors must be [n_traj, T(, A)]")
        if not (schedule.is_cuda and schedule.dtype == torch.int64 and schedule.is_contiguous() and schedule.numel() % B == 0):
            raise OdkError("GaeHead: schedule must be a contiguous int64 CUDA tensor of steps * B entries")
        if not (cursor.is_cuda and cursor.dtype == torch.int32 and cursor.numel() == 1):
            raise OdkError("GaeHead: cursor must be a one-element int32 CUDA tensor")
        if noise.numel() < (schedule.numel() // B) * n * A:
            raise OdkError("GaeHead: the noise pool needs one [n, A] slot per schedule step")
        if n > 5120 or B > 1024:
            raise OdkError("GaeHead: B * T <= 5120, B <= 1024")
        a = GaeHeadArgs()
        a.logits, a.values, a.noise, a.dlogits, a.dvalues, a.losses = (t.data_ptr() for t in (logits, values, noise, dlogits, dvalues, losses))
        a.raw_action, a.old_log_prob, a.reward, a.termination, a.truncation = (rollout[k].data_ptr() for k in ("raw_action", "log_prob", "reward", "termination", "truncation"))
        a.row_idx, a.cursor = schedule.data_ptr(), cursor.data_ptr()
        a.adv_out, a.vs_out, a.stats_out, a.loss_partials = (None if t is None else t.data_ptr() for t in (adv, vs, stats, loss_partials))
        a.B, a.T, a.action_size, a.n_traj, a.normalize_advantage = int(B), int(T), int(A), n_traj, int(bool(cfg["normalize_advantage"]))
        a.gae_lambda, a.discount, a.clipping_epsilon, a.entropy_cost, a.grad_scale = (float(cfg["gae_lambda"]), float(cfg["discounting"]), float(cfg["clipping_epsilon"]),
                                                                                       float(cfg["entropy_cost"]), float(grad_scale))
        self.a, self.keep = a, (fl, rollout, schedule, cursor, adv, vs, stats, loss_partials)

    def __call__(self):
        _chk(load_library().odk_ppo_gae_head(C.byref(self.a), _stream(self.keep[0][0])))


class ColsumFold:
    """colsum[f] = sum over the nblk[f] tile rows of partial[f] ([nblk[f], width]) for up to 8 layers in one launch (`odk_colsum_fold`)."""

    def __init__(self, pairs, nblk):
        k = len(pairs)
        self.k, self.keep = k, pairs
        nblk = [int(nblk)] * k if isinstance(nblk, int) else [int(b) for b in nblk]
        for (p_, o_), nb in zip(pairs, nblk):
            _f32c(p_, o_)
            if p_.numel() < nb * o_.numel():
                raise OdkError("ColsumFold: partial buffer too small")
        self.nblk = (C.c_int * k)(*nblk)
        self.partial = (C.c_void_p * k)(*[p_.data_ptr() for p_, _ in pairs])
        self.out = (C.c_void_p * k)(*[o_.data_ptr() for _, o_ in pairs])
        self.w = (C.c_int * k)(*[int(o_.numel()) for _, o_ in pairs])

    def __call__(self):
        _chk(load_library().odk_colsum_fold(self.partial, self.out, self.w, self.nblk, self.k, _stream(self.keep[0][0])))


class FusedMLP:
    """One or two swish MLPs (n_in -> 512 -> 256 -> 128 -> n_out) whose forward pass is ONE launch and whose backward-data chain is
    ONE launch (`odk_mlp_forward` / `odk_mlp_backward`, csrc/odk_mlp.hip).  Each net is a dict of tensors:
      x [n, n_in], wf[4] (forward-packed weights: `WeightTable.fwd_view`), b[4], out [n, n_out], optionally in_mean / in_std [n_in]
      (the input is normalised on load); for training also wb[4]
      (backward-packed, wb[0] may be None), dout [n, n_out], bias_partial[4] ([ceil(n / 16), width]) and the flat QUAD-ROW
      buffers (`quad_rows(n)` x width floats, layout of `quad_pack`) xp (copy of x), h[3], g[3] (activations, swish'), dz[3],
      doutp (copy of dout) -- `train_buffers` allocates them."""

    @staticmethod
    def train_buffers(n: int, n_in: int, n_out: int, device):
        import torch
        np_, tiles = quad_rows(n), quad_rows(n) // MLP_TILE
        E = lambda k: torch.empty(k, device=device)
        return dict(xp=E(np_ * n_in), h=[E(np_ * w) for w in MLP_HIDDEN], g=[E(np_ * w) for w in MLP_HIDDEN], dz=[E(np_ * w) for w in MLP_HIDDEN],
                    doutp=E(np_ * n_out), bias_partial=[E(tiles * w) for w in MLP_HIDDEN + (n_out,)])

    def __init__(self, nets):
        if not 1 <= len(nets) <= 2:
            raise OdkError("FusedMLP: one or two networks")
        self.k = len(nets)
        self.keep = nets
        self.desc = (MlpDesc * self.k)()
        self.train = []
        for d, nt in zip(self.desc, nets):
            x, out = nt["x"], nt["out"]
            n, n_in = out.shape[0], x.shape[1]      # (with row sources x is the whole rollout: the row count is the output's)
            n_out = out.shape[1]
            widths = (n_in,) + MLP_HIDDEN + (n_out,)
            if n_in > MLP_MAX_IN or n_out > MLP_MAX_OUT or (nt.get("row_idx") is None and x.shape[0] != n):
                raise OdkError("FusedMLP: n_in <= 224, n_out <= 32")
            _f32c(x, out, *nt["wf"], *nt["b"])
            for l in range(4):
                if nt["wf"][l].numel() != _pad16(widths[l]) * widths[l + 1] or nt["b"][l].numel() != widths[l + 1]:
                    raise OdkError(f"FusedMLP: layer {l} is not {widths[l]} -> {widths[l + 1]} (hidden widths are fixed at {MLP_HIDDEN})")
            d.x, d.out, d.n, d.n_in, d.n_out = x.data_ptr(), out.data_ptr(), int(n), int(n_in), int(n_out)
            if nt.get("row_idx") is not None:      # rows through the minibatch schedule (odk_mlp_desc.row_idx): x is the whole rollout
                import torch
                ri, cur = nt["row_idx"], nt["cursor"]
                if not (ri.is_cuda and ri.dtype == torch.int64 and ri.is_contiguous() and cur.is_cuda and cur.dtype == torch.int32 and cur.numel() == 1):
                    raise OdkError("FusedMLP: row_idx int64 / cursor int32[1] CUDA tensors")
                d.row_idx, d.cursor = ri.data_ptr(), cur.data_ptr()
                d.traj_len, d.n_main, d.n_traj = int(nt["traj_len"]), int(nt["n_main"]), int(nt["n_traj"])
                if nt.get("x_tail") is not None:
                    _f32c(nt["x_tail"])
                    d.x_tail = nt["x_tail"].data_ptr()
            if nt.get("in_mean") is not None:      # normalise the input on load: (x - in_mean) / in_std
                _f32c(nt["in_mean"], nt["in_std"])
                if nt["in_mean"].numel() != n_in or nt["in_std"].numel() != n_in:
                    raise OdkError("FusedMLP: in_mean / in_std must have n_in entries")
                d.in_mean, d.in_std = nt["in_mean"].data_ptr(), nt["in_std"].data_ptr()
            for l in range(4):
                d.wf[l], d.b[l] = nt["wf"][l].data_ptr(), nt["b"][l].data_ptr()
            train = "h" in nt
            self.train.append(train)
            if train:
                tiles, np_ = (n + MLP_TILE - 1) // MLP_TILE, quad_rows(n)
                _f32c(nt["dout"], nt["xp"], nt["doutp"], *nt["h"], *nt["g"], *nt["dz"], *nt["bias_partial"], *nt["wb"][1:])
                if nt["xp"].numel() != np_ * n_in or nt["doutp"].numel() != np_ * n_out:
                    raise OdkError("FusedMLP: xp / doutp must be quad-row buffers of quad_rows(n) rows")
                d.xp, d.doutp = nt["xp"].data_ptr(), nt["doutp"].data_ptr()
                for l in range(1, 4):
                    if nt["wb"][l].numel() != _pad16(widths[l + 1]) * widths[l]:
                        raise OdkError(f"FusedMLP: backward-packed weight {l} has the wrong size")
                    d.wb[l] = nt["wb"][l].data_ptr()
                for l in range(3):
                    for key in ("h", "g", "dz"):
                        if nt[key][l].numel() != np_ * MLP_HIDDEN[l]:
                            raise OdkError(f"FusedMLP: {key}[{l}] must be a quad-row buffer of {np_} x {MLP_HIDDEN[l]} floats")
                    d.h[l], d.g[l], d.dz[l] = nt["h"][l].data_ptr(), nt["g"][l].data_ptr(), nt["dz"][l].data_ptr()
                if tuple(nt["dout"].shape) != (n, n_out):
                    raise OdkError("FusedMLP: dout must match out")
                d.dout = nt["dout"].data_ptr()
                for l in range(4):
                    if nt["bias_partial"][l].numel() < tiles * widths[l + 1]:
                        raise OdkError("FusedMLP: bias_partial too small")
                    d.bias_partial[l] = nt["bias_partial"][l].data_ptr()

    def forward(self):
        _chk(load_library().odk_mlp_forward(self.desc, self.k, _stream(self.keep[0]["x"])))

    def backward(self):
        if not all(self.train):
            raise OdkError("FusedMLP.backward: built without training buffers")
        _chk(load_library().odk_mlp_backward(self.desc, self.k, _stream(self.keep[0]["x"])))


class MultiCopy:
    """dst[f] <- src[f] for up to 10 (src, dst) pairs of equally sized contiguous float32 CUDA tensors in ONE launch (the block
    copy mode of `odk_gather_rows`): e.g. the five per-step snapshots of a rollout."""

    def __init__(self, pairs):
        import torch
        n = len(pairs)
        if not 1 <= n <= 10:
            raise OdkError("MultiCopy: 1..10 pairs")
        for s_, d_ in pairs:
            _f32c(s_, d_)
            if s_.numel() != d_.numel():
                raise OdkError("MultiCopy: sizes differ")
        self.n, self.keep = n, pairs
        self.src = (C.c_void_p * n)(*[s_.data_ptr() for s_, _ in pairs])
        self.dst = (C.c_void_p * n)(*[d_.data_ptr() for _, d_ in pairs])
        self.rows = (C.c_int * n)(*[int(s_.numel()) for s_, _ in pairs])
        self.base = (C.c_longlong * n)(*([0] * n))
        self.idx = torch.zeros(1, dtype=torch.int64, device=pairs[0][0].device)   # (unused: every field is a block copy)

    def __call__(self):
        _chk(load_library().odk_gather_rows(self.src, self.dst, self.rows, self.base, self.n, _ptr(self.idx), 1, 1, _stream(self.keep[0][0])))


class RowGather:
    """dst[f][b] = src[f][idx[b]] for a fixed set of (src, dst) float32 tensors in one launch (`odk_gather_rows`).
    `direct`: further (src, dst) pairs copied as plain blocks, dst[b] = src[base + b] with `base` given per call (the
    learner's slice of its noise pool rides along with the minibatch gather instead of being a launch of its own)."""

    def __init__(self, pairs, direct=()):
        allp = list(pairs) + list(direct)
        n = len(allp)
        if n > 10:
            raise OdkError("RowGather: at most 10 fields")
        self.n, self.nidx = n, len(pairs)
        self.keep = allp
        for s_, d_ in allp:
            _f32c(s_, d_)
        self.src = (C.c_void_p * n)(*[s_.data_ptr() for s_, _ in allp])
        self.dst = (C.c_void_p * n)(*[d_.data_ptr() for _, d_ in allp])
        self.rows = (C.c_int * n)(*[int(s_[0].numel()) for s_, _ in allp])
        self.base = (C.c_longlong * n)(*([-1] * n))
        self.nrows = int(pairs[0][1].shape[0])
        self.src_rows = int(pairs[0][0].shape[0])
        for s_, d_ in pairs:
            if int(s_.shape[0]) != self.src_rows or int(d_.shape[0]) != self.nrows or s_[0].numel() != d_[0].numel():
                raise OdkError("RowGather: every source needs the same row count, every destination the same row count, and "
                               "matching row lengths")
        for s_, d_ in direct:
            if int(d_.shape[0]) != self.nrows or s_[0].numel() != d_[0].numel():
                raise OdkError("RowGather: a direct field needs the destinations' row count and matching row lengths")

    def __call__(self, idx, direct_base=()):
        """`idx`: contiguous int64 CUDA tensor of `nrows` source-row numbers (values outside the source are not read: those
        destination rows become NaN); `direct_base`: first source row of every direct field."""
        import torch
        if not (idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == self.nrows):
            raise OdkError(f"RowGather: idx must be a contiguous int64 CUDA tensor of {self.nrows} entries "
                           f"(got {idx.dtype}, {idx.device}, contiguous={idx.is_contiguous()}, {idx.numel()} entries)")
        if len(direct_base) != self.n - self.nidx:
            raise OdkError("RowGather: one base row per direct field")
        for k, b in enumerate(direct_base):
            s_ = self.keep[self.nidx + k][0]
            if b < 0 or b + self.nrows > int(s_.shape[0]):
                raise OdkError("RowGather: direct block out of range")
            self.base[self.nidx + k] = int(b)
        _chk(load_library().odk_gather_rows(self.src, self.dst, self.rows, self.base, self.n, _ptr(idx), self.nrows, self.src_rows, _stream(idx)))


class Batch:
    """`nenv` environments resident on one GPU.  Thin: every method is one C-ABI call."""

    def __init__(self, model: Model, nenv: int, cfg: Optional[EnvConfig] = None, device: int = 0, prm: Optional[dict] = None):
        import torch

        if not torch.cuda.is_available():
            raise OdkError("no HIP device visible: the env engine has no CPU path")
        self.L = load_library()
        self.torch = torch
        self.model, self.nenv, self.device = model, int(nenv), int(device)
        self.cfg = cfg if cfg is not None else default_config()
        blob = model.blob()
        self._m = C.c_void_p()
        _chk(self.L.odk_model_load(blob, len(blob), C.byref(self._m)))
        prm = prm if prm is not None else load_prm()
        self._table = np.ascontiguousarray(prm["table"], np.float32)
        dxs, dys, dths = (np.ascontiguousarray(prm[k], np.float64) for k in ("dxs", "dys", "dthetas"))
        ranges = np.ascontiguousarray(np.concatenate([prm["dx_range"], prm["dy_range"], prm["dtheta_range"]]), np.float64)
        self._b = C.c_void_p()
        _chk(self.L.odk_batch_create(self._m, C.byref(self.cfg), self.nenv, self.device, _fp(self._table), _dp(dxs), len(dxs), _dp(dys),
                                     len(dys), _dp(dths), len(dths), _dp(ranges), int(prm["nb_steps_in_period"][0]), C.byref(self._b)))
        dev = torch.device("cuda", self.device)
        f32 = dict(dtype=torch.float32, device=dev)
        a_, b_ = C.c_int(0), C.c_int(0)
        _chk(self.L.odk_model_obs_sizes(self._m, int(self.cfg.env_kind), C.byref(a_), C.byref(b_)))
        self.nobs, self.npriv = a_.value, b_.value
        self.lanes_per_env = int(self.L.odk_batch_lanes(self._b))      # what the kernels run (cfg.lanes_per_env is a hint)
        self.obs = torch.zeros(self.nenv, self.nobs, **f32)
        self.priv = torch.zeros(self.nenv, self.npriv, **f32)
        self.reward = torch.zeros(self.nenv, **f32)
        self.done = torch.zeros(self.nenv, **f32)
        self.truncation = torch.zeros(self.nenv, **f32)
        self.metrics = torch.zeros(self.nenv, NMETRIC, **f32)
        self._outs = Outputs(self.obs.data_ptr(), self.priv.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                             self.truncation.data_ptr(), self.metrics.data_ptr())
        self.generation = 0          # advanced by every call that rewrites the per-env records (reset / step / set_records): `State.info` checks it

    # -- streams: calls are ordered on torch's current stream
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def set_config(self, cfg: EnvConfig):
        self.cfg = cfg
        _chk(self.L.odk_batch_set_config(self._b, C.byref(cfg)))

    def set_param(self, param: int, values: np.ndarray):
        v = np.ascontiguousarray(values, np.float32).reshape(self.nenv, -1)
        _chk(self.L.odk_batch_set_param(self._b, param, _fp(v), v.shape[1]))

    def reset(self, seed: int, env_id_offset: int = 0):
        self.generation += 1
        _chk(self.L.odk_reset(self._b, seed & 0xFFFFFFFF, env_id_offset, C.byref(self._outs), self._stream()))

    def step(self, action):
        assert action.is_cuda and action.dtype == self.torch.float32 and action.is_contiguous() and tuple(action.shape) == (self.nenv, self.model.nu)
        self.generation += 1
        _chk(self.L.odk_step(self._b, C.c_void_p(action.data_ptr()), C.byref(self._outs), self._stream()))

    def physics_step(self, ctrl, n_substeps: int = 10):
        assert ctrl.is_cuda and ctrl.dtype == self.torch.float32 and ctrl.is_contiguous() and tuple(ctrl.shape) == (self.nenv, self.model.nu)
        _chk(self.L.odk_physics_step(self._b, C.c_void_p(ctrl.data_ptr()), n_substeps, self._stream()))

    # -- synchronous host access (tests, checkpoints)
    def get_state(self):
        qpos = np.zeros((self.nenv, self.model.nq), np.float32); qvel = np.zeros((self.nenv, self.model.nv), np.float32)
        warm = np.zeros((self.nenv, self.model.nv), np.float32)
        _chk(self.L.odk_batch_get_state(self._b, _fp(qpos), _fp(qvel), _fp(warm)))
        return qpos, qvel, warm

    def set_state(self, qpos=None, qvel=None, warm=None):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float32) for a in (qpos, qvel, warm)]
        _chk(self.L.odk_batch_set_state(self._b, *[None if a is None else _fp(a) for a in arrs]))

    def get_debug(self):
        s = np.zeros((self.nenv, 46), np.float32); a = np.zeros((self.nenv, self.model.nu), np.float32)
        c = np.zeros((self.nenv, 12), np.float32); q = np.zeros((self.nenv, self.model.nv), np.float32)
        _chk(self.L.odk_batch_get_debug(self._b, _fp(s), _fp(a), _fp(c), _fp(q)))
        return dict(sensordata=s, actuator_force=a, contact_dist=c, qacc=q)

    def lds_image(self) -> np.ndarray:
        n = self.L.odk_batch_lds_size(self._b)
        img = np.zeros((self.nenv, n), np.float32)
        _chk(self.L.odk_batch_get_lds(self._b, _fp(img)))
        return img

    def lds_offset(self, name: str) -> int:
        off = self.L.odk_lds_offset(self._b, name.encode())
        if off < 0:
            raise KeyError(name)
        return off

    def records(self) -> np.ndarray:
        n = self.L.odk_batch_record_size(self._b)
        r = np.zeros((self.nenv, n), np.float32)
        _chk(self.L.odk_batch_get_records(self._b, _fp(r)))
        return r

    def set_records(self, records: np.ndarray):
        r = np.ascontiguousarray(records, np.float32)
        assert r.shape == (self.nenv, self.L.odk_batch_record_size(self._b))
        self.generation += 1
        _chk(self.L.odk_batch_set_records(self._b, _fp(r)))

    INFO_FIELDS = ("rng", "step", "command", "last_act", "last_last_act", "last_last_last_act", "motor_targets", "feet_air_time", "last_contact",
                   "swing_peak", "push", "push_step", "push_interval_steps", "action_history", "imu_history", "imitation_i",
                   "steps", "truncation", "episode_done", "episode_metrics/sum_reward", "episode_metrics/length", "episode_metrics/reward_terms")

    def record_field(self, name: str):
        """(offset, count, kind) of a named field inside a record (`odk_record_field`; kind 0 float32, 1 int32, 2 bit mask)."""
        o, n, k = C.c_int(0), C.c_int(0), C.c_int(0)
        _chk(self.L.odk_record_field(self._b, name.encode(), C.byref(o), C.byref(n), C.byref(k)))
        return o.value, n.value, k.value

    def info(self, records: np.ndarray = None) -> dict:
        """The carried `info` dict of the reference's State (joystick.py:278-302 + the wrapper's additions), one [nenv, ...] array per
        key, as VIEWS over a host copy of the records (`records()` when none is passed): write through a view, then hand the
        same array to `set_records` to preset a field.  int32 fields come as int32 views; `last_contact` as the packed mask
        (bit f = foot f), with `last_contact_bool(info)` for the reference's bool[2]."""
        r = self.records() if records is None else records
        ri = r.view(np.int32)
        out = {"_records": r}
        for name in self.INFO_FIELDS:
            o, n, k = self.record_field(name)
            v = (r if k == 0 else ri)[:, o:o + n]
            out[name] = v[:, 0] if n == 1 else v
        return out

    @staticmethod
    def last_contact_bool(info: dict) -> np.ndarray:
        m = info["last_contact"]
        return np.stack([(m & 1) != 0, (m & 2) != 0], axis=1)

    def timing(self, enable):
        """Average kernel milliseconds of the timed launches since the last call; `enable`: False / 0 = off, True / 1 = time every
        launch from now on, n = every n-th launch (`odk_batch_timing`)."""
        ms, n = C.c_float(0), C.c_int(0)
        _chk(self.L.odk_batch_timing(self._b, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "_b", None):
            self.L.odk_batch_destroy(self._b); self._b = None
        if getattr(self, "_m", None):
            self.L.odk_model_free(self._m); self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
