"""ctypes wrapper around the CPU oracle (oracle/libodk_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package (open_duck_playground_amd)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> None:
    srcs = [os.path.join(_DIR, f) for f in ("odk_oracle.c", "odk_oracle_env.c", "odk_oracle_convex.inc", "odk_oracle.h", "odk_oracle_env.h", "Makefile")]
    libs = [os.path.join(_DIR, f) for f in ("libodk_oracle.so", "libodk_oracle_f32.so")]
    if not force and all(os.path.exists(l) for l in libs):
        newest = max(os.path.getmtime(s) for s in srcs if os.path.exists(s))
        if all(os.path.getmtime(l) >= newest for l in libs):
            return
    subprocess.check_call(["make", "-C", _DIR, "-B", "-s"])


class _Lib:
    def __init__(self, f32: bool = False):
        build()
        self.f32 = f32
        self.real = C.c_float if f32 else C.c_double
        self.npreal = np.float32 if f32 else np.float64
        # ODK_ORACLE_F32_LIB: bench.py's cpu_baseline points the float32 build at a copy compiled with -march=native on the box
        path = os.path.join(_DIR, "libodk_oracle_f32.so" if f32 else "libodk_oracle.so")
        if f32 and os.environ.get("ODK_ORACLE_F32_LIB") and os.path.exists(os.environ["ODK_ORACLE_F32_LIB"]):
            path = os.environ["ODK_ORACLE_F32_LIB"]
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        P = C.c_void_p
        RP = C.POINTER(self.real)
        L.odko_model_load.restype = P; L.odko_model_load.argtypes = [C.c_char_p, C.c_uint64]
        L.odko_model_free.argtypes = [P]
        L.odko_model_copy.restype = P; L.odko_model_copy.argtypes = [P]
        L.odko_set_tie_bias.restype = None; L.odko_set_tie_bias.argtypes = [C.c_int, self.real, self.real]
        L.odko_set_tie_bias_window.restype = None; L.odko_set_tie_bias_window.argtypes = [C.c_int, self.real, self.real, C.c_int, C.c_int]
        L.odko_model_jitter_hulls.restype = None; L.odko_model_jitter_hulls.argtypes = [P, C.c_uint32, self.real]
        L.odko_model_field.restype = RP; L.odko_model_field.argtypes = [P, C.c_char_p, C.POINTER(C.c_int)]
        L.odko_model_int.restype = C.c_int; L.odko_model_int.argtypes = [P, C.c_char_p]
        L.odko_model_set_int.restype = C.c_int; L.odko_model_set_int.argtypes = [P, C.c_char_p, C.c_int]
        L.odko_model_eq_set_active.restype = C.c_int; L.odko_model_eq_set_active.argtypes = [P, C.c_int, C.c_int]
        IP = C.POINTER(C.c_int)
        L.odko_convex_pair.restype = C.c_int
        L.odko_convex_pair.argtypes = [RP, C.c_int, IP, C.c_int, RP, RP, RP, C.c_int, IP, C.c_int, RP, RP, RP, RP, RP, RP]
        L.odko_model_convex_counts.argtypes = [P, C.c_int, IP, IP, IP]
        L.odko_data_new.restype = P
        L.odko_data_free.argtypes = [P]
        for fn in ("odko_make_data", "odko_forward", "odko_step"):
            getattr(L, fn).argtypes = [P, P]
        L.odko_env_physics_step.argtypes = [P, P, RP, C.c_int]
        L.odko_data_field.restype = RP; L.odko_data_field.argtypes = [P, C.c_char_p, C.POINTER(C.c_int)]
        L.odko_data_int.restype = C.c_int; L.odko_data_int.argtypes = [P, C.c_char_p]
        # env half
        L.odko_prm_new.restype = P
        L.odko_prm_new.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                   C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int]
        L.odko_prm_free.argtypes = [P]
        L.odko_prm_set_table64.argtypes = [P, C.POINTER(C.c_double)]
        L.odko_prm_eval64.argtypes = [P, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
        L.odko_prm_eval.argtypes = [P, self.real, self.real, self.real, C.c_int, RP]
        L.odko_prm_index.argtypes = [P, self.real, self.real, self.real, C.POINTER(C.c_int)]
        L.odko_env_new.restype = P; L.odko_env_new.argtypes = [P, P, P]
        L.odko_env_free.argtypes = [P]
        L.odko_env_clone.restype = P; L.odko_env_clone.argtypes = [P]
        L.odko_env_config.restype = RP; L.odko_env_config.argtypes = [P, C.c_char_p, C.POINTER(C.c_int)]
        L.odko_env_field.restype = RP; L.odko_env_field.argtypes = [P, C.c_char_p, C.POINTER(C.c_int)]
        L.odko_env_int.restype = C.POINTER(C.c_int); L.odko_env_int.argtypes = [P, C.c_char_p, C.POINTER(C.c_int)]
        L.odko_env_data.restype = P; L.odko_env_data.argtypes = [P]
        L.odko_env_reset.argtypes = [P, C.c_uint32, C.c_uint32]
        L.odko_env_set_standing.argtypes = [P]; L.odko_env_set_standing.restype = None
        L.odko_env_nobs.argtypes = [P]; L.odko_env_npriv.argtypes = [P]
        L.odko_env_step.argtypes = [P, RP]
        L.odko_rng_uniform.restype = C.c_float; L.odko_rng_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.odko_env_key.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        for name in ("odko_reward_tracking_lin_vel", "odko_reward_tracking_ang_vel"):
            getattr(L, name).restype = self.real
            getattr(L, name).argtypes = [RP, RP, self.real]
        L.odko_cost_torques.restype = self.real; L.odko_cost_torques.argtypes = [RP, C.c_int]
        L.odko_cost_action_rate.restype = self.real; L.odko_cost_action_rate.argtypes = [RP, RP, C.c_int]
        L.odko_cost_stand_still.restype = self.real; L.odko_cost_stand_still.argtypes = [RP, RP, RP, RP, C.c_int]
        L.odko_cost_stand_still_legs.restype = self.real; L.odko_cost_stand_still_legs.argtypes = [RP, RP, RP, RP, C.c_int]
        L.odko_cost_orientation.restype = self.real; L.odko_cost_orientation.argtypes = [RP]
        L.odko_cost_head_pos.restype = self.real; L.odko_cost_head_pos.argtypes = [RP, RP]
        L.odko_reward_imitation.restype = self.real; L.odko_reward_imitation.argtypes = [RP, RP, RP, RP, RP, RP, RP]
        L.odko_rollout_mt.restype = C.c_double
        L.odko_rollout_mt.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]

    def arr(self, x):
        return np.ascontiguousarray(x, dtype=self.npreal)

    def ptr(self, a):
        return a.ctypes.data_as(C.POINTER(self.real))


_libs = {}


def lib(f32: bool = False) -> _Lib:
    if f32 not in _libs:
        _libs[f32] = _Lib(f32)
    return _libs[f32]


def set_tie_bias(mask: int, eps: float = 0.0, eps_rel: float = 0.0, f32: bool = False, window=None):
    """odko_set_tie_bias: collision decisions of the classes in `mask` take the runner-up inside the band (tests' referee); 0 = off.
    `window` = (first, last): only those collision passes after this call (one per mjx.step)."""
    if window is None:
        lib(f32).lib.odko_set_tie_bias(int(mask), eps, eps_rel)
    else:
        lib(f32).lib.odko_set_tie_bias_window(int(mask), eps, eps_rel, int(window[0]), int(window[1]))


def convex_pair(va, ta, pa, Ra, vb, tb, pb, Rb, f32: bool = False):
    """convex_convex (odk_oracle_convex.inc) on two polytopes given as vertices + outward triangles and poses.
    Returns dict(dist[4], pos[4,3], normal[3], sep_a, sep_b, sep_e, kind) -- kind 0 / 1: face contact with A / B as reference, 2: edge."""
    L = lib(f32)
    A = lambda x: L.arr(np.asarray(x, dtype=np.float64).reshape(-1))
    I = lambda x: np.ascontiguousarray(np.asarray(x, dtype=np.int32).reshape(-1))
    va_, vb_, pa_, pb_, Ra_, Rb_ = A(va), A(vb), A(pa), A(pb), A(Ra), A(Rb)
    ta_, tb_ = I(ta), I(tb)
    dist, pos, nrm, sat = L.arr(np.zeros(4)), L.arr(np.zeros(12)), L.arr(np.zeros(3)), L.arr(np.zeros(3))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    kind = L.lib.odko_convex_pair(L.ptr(va_), len(va_) // 3, ip(ta_), len(ta_) // 3, L.ptr(pa_), L.ptr(Ra_), L.ptr(vb_), len(vb_) // 3, ip(tb_),
                                  len(tb_) // 3, L.ptr(pb_), L.ptr(Rb_), L.ptr(dist), L.ptr(pos), L.ptr(nrm), L.ptr(sat))
    if kind < 0:
        raise ValueError("polytope too large for the oracle's fixed arrays")
    return dict(dist=dist.copy(), pos=pos.reshape(4, 3).copy(), normal=nrm.copy(), sep_a=float(sat[0]), sep_b=float(sat[1]), sep_e=float(sat[2]), kind=kind)


class _Fields:
    """numpy views into a native struct via the *_field(name) accessors."""

    def __init__(self, L: _Lib, handle, getter, int_getter=None):
        self._L, self._h, self._get, self._iget = L, handle, getter, int_getter

    def view(self, name: str) -> np.ndarray:
        n = C.c_int(0)
        p = self._get(self._h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def __getitem__(self, name):
        return self.view(name)


def _blob_ints(blob: bytes, names):
    """int32 records of an ODKM blob by name (layout: open_duck_playground_amd/model.py; the oracle keeps its own reader)"""
    import struct
    out = {}
    if len(blob) < 16 or blob[:4] != b"ODKM":
        return out
    n = struct.unpack_from("<I", blob, 8)[0]
    off = 16
    for _ in range(n):
        nm, code, ndim, s0, s1, s2, s3, nbytes = struct.unpack_from("<32sII4IQ", blob, off)
        off += 64
        key = nm.rstrip(b"\0").decode()
        if key in names and code == 1:
            out[key] = np.frombuffer(blob, dtype="<i4", count=nbytes // 4, offset=off).copy()
        off += nbytes + ((-nbytes) % 8)
    return out


class OracleModel:
    def __init__(self, blob: bytes, f32: bool = False, _handle=None, _named=None):
        self.L = lib(f32)
        self.h = _handle if _handle is not None else self.L.lib.odko_model_load(blob, len(blob))
        if not self.h:
            raise ValueError("odko_model_load failed")
        # the named objects the env logic looks up (reference base.py:63-125, joystick.py:121-181: imu / feet sites, feet / floor geoms, sensor
        # addresses), as the model compiler resolved them: OracleEnv hands them to the C env (whose defaults are the duck's)
        self.named = _named if _named is not None else _blob_ints(blob, ("k_site_imu", "k_site_feet", "k_foot_cgeom", "k_floor_cgeom", "k_adr"))
        self.f = _Fields(self.L, self.h, self.L.lib.odko_model_field)
        for k in ("nq", "nv", "nu", "nbody", "njnt", "nsite", "nsensordata", "ncgeom", "npair", "neq"):
            setattr(self, k, self.L.lib.odko_model_int(self.h, k.encode()))

    def set_int(self, name: str, value: int):
        if self.L.lib.odko_model_set_int(self.h, name.encode(), int(value)) != 0:
            raise KeyError(name)
        setattr(self, name, int(value))

    def eq_set_active(self, e: int, on: bool):
        """mjData.eq_active: equality constraint `e` on / off"""
        if self.L.lib.odko_model_eq_set_active(self.h, int(e), int(bool(on))) != 0:
            raise IndexError(e)

    def convex_counts(self, g: int):
        """(vertices, faces after the coplanar merge, unique edges) of mesh geom g"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        if self.L.lib.odko_model_convex_counts(self.h, g, C.byref(a), C.byref(b), C.byref(c)) != 0:
            raise IndexError(g)
        return a.value, b.value, c.value

    def jitter_hulls(self, seed: int, rel: float):
        """relative noise on the hull vertices (odko_model_jitter_hulls): call on a copy()"""
        self.L.lib.odko_model_jitter_hulls(self.h, int(seed), rel)

    def copy(self) -> "OracleModel":
        return OracleModel(b"", self.L.f32, _handle=self.L.lib.odko_model_copy(self.h), _named=self.named)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.lib.odko_model_free(self.h)
            self.h = None


class OracleData:
    def __init__(self, model: OracleModel, _handle=None, owner=True):
        self.m = model
        self.L = model.L
        self.owner = owner
        self.h = _handle if _handle is not None else self.L.lib.odko_data_new()
        if _handle is None:
            self.L.lib.odko_make_data(model.h, self.h)
        self.f = _Fields(self.L, self.h, self.L.lib.odko_data_field)

    def __getitem__(self, name):
        return self.f.view(name)

    def i(self, name):
        return self.L.lib.odko_data_int(self.h, name.encode())

    def forward(self):
        self.L.lib.odko_forward(self.m.h, self.h)

    def step(self):
        self.L.lib.odko_step(self.m.h, self.h)

    def env_physics_step(self, ctrl, n_substeps=10):
        c = self.L.arr(ctrl)
        self.L.lib.odko_env_physics_step(self.m.h, self.h, self.L.ptr(c), n_substeps)

    def M(self):
        nv = self.m.nv
        return self["qM"][: nv * nv].reshape(nv, nv).copy()

    def J(self):
        nv, nefc = self.m.nv, self.i("nefc")
        return self["efc_J"][: nefc * nv].reshape(nefc, nv).copy()

    def __del__(self):
        if self.owner and getattr(self, "h", None):
            self.L.lib.odko_data_free(self.h)
            self.h = None


class OraclePRM:
    """PolyReferenceMotion restatement (reference poly_reference_motion.py:148-168)."""

    def __init__(self, prm: dict, f32: bool = False):
        self.L = lib(f32)
        self.table = np.ascontiguousarray(prm["table"], dtype=np.float32)
        d = lambda k: np.ascontiguousarray(prm[k], dtype=np.float64)
        self.dxs, self.dys, self.dths = d("dxs"), d("dys"), d("dthetas")
        self.ranges = np.concatenate([d("dx_range"), d("dy_range"), d("dtheta_range")])
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        self.nsteps = int(prm["nb_steps_in_period"][0])
        self.h = self.L.lib.odko_prm_new(self.table.ctypes.data_as(C.POINTER(C.c_float)), dp(self.dxs), len(self.dxs),
                                         dp(self.dys), len(self.dys), dp(self.dths), len(self.dths), dp(self.ranges), self.nsteps)

        if "table64" in prm:
            self.table64 = np.ascontiguousarray(prm["table64"], dtype=np.float64)
            self.L.lib.odko_prm_set_table64(self.h, dp(self.table64))

    def eval64(self, dx, dy, dth, i):
        out = np.zeros(40, dtype=np.float64)
        self.L.lib.odko_prm_eval64(self.h, dx, dy, dth, int(i), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def eval(self, dx, dy, dth, i):
        out = np.zeros(40, dtype=self.L.npreal)
        self.L.lib.odko_prm_eval(self.h, dx, dy, dth, int(i), self.L.ptr(out))
        return out

    def index(self, dx, dy, dth):
        idx = (C.c_int * 3)()
        self.L.lib.odko_prm_index(self.h, dx, dy, dth, idx)
        return tuple(idx)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.lib.odko_prm_free(self.h)
            self.h = None


class OracleEnv:
    """One Joystick environment (reference joystick.py) incl. Episode/AutoReset wrappers."""

    def __init__(self, model: OracleModel, prm: OraclePRM, standing: bool = False, _handle=None):
        self.L = model.L
        self.m, self.prm = model, prm
        self.h = _handle if _handle is not None else self.L.lib.odko_env_new(model.h, prm.h, None)
        if standing and _handle is None:   # reference standing.py defaults on top of the Joystick ones
            self.L.lib.odko_env_set_standing(self.h)
        self.f = _Fields(self.L, self.h, self.L.lib.odko_env_field)
        self.cfg = _Fields(self.L, self.h, self.L.lib.odko_env_config)
        self.data = OracleData(model, _handle=self.L.lib.odko_env_data(self.h), owner=False)
        nm = getattr(model, "named", None) or {}
        if _handle is None and len(nm) == 5:      # (a clone carries its parent's)
            self.ints("imu_site")[0] = nm["k_site_imu"][0]; self.ints("feet_site")[:] = nm["k_site_feet"][:2]
            self.ints("feet_cgeom")[:] = nm["k_foot_cgeom"][:2]; self.ints("floor_cgeom")[0] = nm["k_floor_cgeom"][0]
            a = nm["k_adr"]
            for k, name in enumerate(("adr_gyro", "adr_local_linvel", "adr_accelerometer", "adr_upvector", "adr_global_angvel")):
                self.ints(name)[0] = a[k]
            self.ints("adr_foot_linvel")[:] = a[5:7]

    def clone(self) -> "OracleEnv":
        """an independent copy of the whole env (state, info, wrappers, config) sharing the model and the motion table"""
        return OracleEnv(self.m, self.prm, _handle=self.L.lib.odko_env_clone(self.h))

    def __getitem__(self, name):
        return self.f.view(name)

    def ints(self, name):
        n = C.c_int(0)
        p = self.L.lib.odko_env_int(self.h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    @property
    def nobs(self) -> int:
        return self.L.lib.odko_env_nobs(self.h)

    @property
    def npriv(self) -> int:
        return self.L.lib.odko_env_npriv(self.h)

    def reset(self, seed: int, env_id: int):
        self.L.lib.odko_env_reset(self.h, seed, env_id)

    def step(self, action):
        a = self.L.arr(action)
        self.L.lib.odko_env_step(self.h, self.L.ptr(a))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.lib.odko_env_free(self.h)
            self.h = None


class OracleVecEnv:
    """N oracle envs behind the batched reset / step surface of the product's Joystick (tools only: the height-field hypothesis sweep
    trains against the oracle).  Tensors are torch tensors on `device`; the physics runs on `threads` host threads."""

    METRIC_NAMES = ("reward/tracking_lin_vel", "reward/tracking_ang_vel", "cost/torques", "cost/action_rate", "cost/stand_still",
                    "reward/alive", "reward/imitation", "swing_peak")

    def __init__(self, model: OracleModel, prm: OraclePRM, n: int, device="cpu", threads: int = 8, standing: bool = False, env_id_offset: int = 0):
        import torch
        self.L, self.m, self.prm, self.num_envs, self.device, self.threads, self.offset = model.L, model, prm, int(n), torch.device(device), int(threads), int(env_id_offset)
        lib_ = self.L.lib
        P, FP = C.c_void_p, C.POINTER(C.c_float)
        lib_.odko_vec_new.restype = P; lib_.odko_vec_new.argtypes = [P, P, C.c_int, C.c_int]
        lib_.odko_vec_free.argtypes = [P]
        lib_.odko_vec_env.restype = P; lib_.odko_vec_env.argtypes = [P, C.c_int]
        lib_.odko_vec_reset.argtypes = [P, C.c_uint32, C.c_uint32, C.c_int, FP, FP]
        lib_.odko_vec_step.argtypes = [P, FP, C.c_int, FP, FP, FP, FP, FP, FP]
        self.h = lib_.odko_vec_new(model.h, prm.h, self.num_envs, int(standing))
        e0 = OracleEnv(model, prm, _handle=lib_.odko_vec_env(self.h, 0)); e0.h = None     # (borrowed handle: never freed here)
        self.nobs, self.npriv = lib_.odko_env_nobs(lib_.odko_vec_env(self.h, 0)), lib_.odko_env_npriv(lib_.odko_vec_env(self.h, 0))
        self.action_size = model.nu
        self.observation_size = {"state": (self.nobs,), "privileged_state": (self.npriv,)}
        z = lambda *s: np.zeros(s, np.float32)
        self._obs, self._priv, self._rew, self._done, self._trunc, self._met = z(n, self.nobs), z(n, self.npriv), z(n), z(n), z(n), z(n, 8)

    def config(self, name: str, value):
        """sets a config field (odko_env_config name) on every env"""
        for i in range(self.num_envs):
            e = OracleEnv(self.m, self.prm, _handle=self.L.lib.odko_vec_env(self.h, i))
            e.cfg[name][:] = value
            e.h = None

    def _state(self):
        import torch
        from types import SimpleNamespace
        t = lambda a: torch.from_numpy(a.copy()).to(self.device)
        met = t(self._met)
        return SimpleNamespace(data=None, obs={"state": t(self._obs), "privileged_state": t(self._priv)}, reward=t(self._rew), done=t(self._done),
                               metrics={nm: met[:, i] for i, nm in enumerate(self.METRIC_NAMES)}, info={"truncation": t(self._trunc)})

    def reset(self, seed: int = 0):
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        self.L.lib.odko_vec_reset(self.h, int(seed), self.offset, self.threads, fp(self._obs), fp(self._priv))
        self._rew[:] = 0; self._done[:] = 0; self._trunc[:] = 0; self._met[:] = 0
        return self._state()

    def step(self, state, action):
        a = np.ascontiguousarray(action.detach().to("cpu").numpy(), np.float32)
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        self.L.lib.odko_vec_step(self.h, fp(a), self.threads, fp(self._obs), fp(self._priv), fp(self._rew), fp(self._done), fp(self._trunc), fp(self._met))
        return self._state()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.lib.odko_vec_free(self.h)
            self.h = None
