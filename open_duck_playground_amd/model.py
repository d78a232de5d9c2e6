"""ModelBlob: the flat, versioned wire format between the Python model compiler and the
native engine (`odk_model_load`, include/odk.h) / the oracle (`odko_model_load`).

Plays the role of `mjx.put_model(mj_model)` (reference base.py:61): it is the device-ready
form of the compiled MJCF.  Layout (little endian):

    char  magic[4] = "ODKM";  u32 version;  u32 nrecords;  u32 reserved;
    nrecords x { char name[32]; u32 dtype (0=f64, 1=i32); u32 ndim; u32 shape[4];
                 u64 nbytes;  u8 data[nbytes padded to 8] }

String tables (names_*) are not part of the blob; they stay in the Python `Model`.
"""
from __future__ import annotations

import os
import struct
from typing import Dict

import numpy as np

BLOB_MAGIC = b"ODKM"
BLOB_VERSION = 1

_ASSET_DIR = os.path.join(os.path.dirname(__file__), "assets")


def pack_blob(arrays: Dict[str, np.ndarray]) -> bytes:
    recs = []
    for name, arr in arrays.items():
        arr = np.asarray(arr)
        if arr.dtype.kind in ("U", "S", "O"):
            continue
        if arr.dtype.kind == "f":
            arr = np.ascontiguousarray(arr, dtype="<f8"); code = 0
        else:
            arr = np.ascontiguousarray(arr, dtype="<i4"); code = 1
        if arr.ndim > 4:
            raise ValueError(name)
        shape = list(arr.shape) + [1] * (4 - arr.ndim)
        raw = arr.tobytes()
        pad = (-len(raw)) % 8
        nm = name.encode()
        if len(nm) > 31:
            raise ValueError(f"record name too long: {name}")
        recs.append(struct.pack("<32sII4IQ", nm, code, arr.ndim, *shape, len(raw)) + raw + b"\0" * pad)
    return struct.pack("<4sIII", BLOB_MAGIC, BLOB_VERSION, len(recs), 0) + b"".join(recs)


def unpack_blob(blob: bytes) -> Dict[str, np.ndarray]:
    magic, version, n, _ = struct.unpack_from("<4sIII", blob, 0)
    if magic != BLOB_MAGIC or version != BLOB_VERSION:
        raise ValueError("not an ODKM v1 blob")
    off = 16
    out = {}
    for _ in range(n):
        nm, code, ndim, s0, s1, s2, s3, nbytes = struct.unpack_from("<32sII4IQ", blob, off)
        off += 64
        dt = "<f8" if code == 0 else "<i4"
        shape = (s0, s1, s2, s3)[:ndim]
        out[nm.rstrip(b"\0").decode()] = np.frombuffer(blob, dtype=dt, count=int(np.prod(shape)) if ndim else 1,
                                                       offset=off).reshape(shape).copy()
        off += nbytes + ((-nbytes) % 8)
    return out


class Model:
    """Compiled model: numpy arrays (`.a[name]`), blob bytes and name lookups.  Mirrors the
    slice of `mujoco.MjModel` the reference touches (base.py:63-125, joystick.py:123-181)."""

    def __init__(self, arrays: Dict[str, np.ndarray], xml_path: str = ""):
        self.a = arrays
        self.xml_path = xml_path
        self.nq, self.nv, self.nu = int(arrays["nq"][0]), int(arrays["nv"][0]), int(arrays["nu"][0])
        self.nbody, self.njnt = int(arrays["nbody"][0]), int(arrays["njnt"][0])
        self.nsensordata = int(arrays["nsensordata"][0])

    def blob(self) -> bytes:
        """Wire format: compiled arrays + the kernels' static topology tables (tables.py)."""
        from .tables import build_kernel_tables
        return pack_blob({**self.a, **build_kernel_tables(self.a)})

    # name lookups (mj_name2id equivalents, base.py:136-152)
    def _id(self, table: str, name: str) -> int:
        names = list(self.a[table])
        return names.index(name) if name in names else -1

    def body_id(self, name): return self._id("names_body", name)
    def joint_id(self, name): return self._id("names_jnt", name)
    def geom_id(self, name): return self._id("names_geom", name)
    def site_id(self, name): return self._id("names_site", name)
    def sensor_id(self, name): return self._id("names_sensor", name)
    def actuator_id(self, name): return self._id("names_actuator", name)

    def save(self, path: str):
        np.savez_compressed(path, **{k: v for k, v in self.a.items()})

    @classmethod
    def load(cls, path: str) -> "Model":
        z = np.load(path, allow_pickle=False)
        return cls({k: z[k] for k in z.files}, xml_path=path)

    @classmethod
    def from_xml(cls, xml_path: str, sim_dt: float = 0.002) -> "Model":
        from . import mjcf
        return cls(mjcf.compile_mjcf(xml_path, sim_dt=sim_dt), xml_path=xml_path)


def asset_path(name: str) -> str:
    return os.path.join(_ASSET_DIR, name)


def load_task_model(task: str) -> Model:
    """Pre-compiled models shipped with the package (the GPU box has no MJCF sources).
    Regenerate with tools/compile_models.py from the reference's xmls/ directory."""
    p = asset_path(f"{task}.npz")
    if not os.path.exists(p):
        raise KeyError(task)  # same failure mode as constants.task_to_xml (reference constants.py:28-34)
    return Model.load(p)
