"""Minimal TensorBoard event-file writer (scalars only): the reference logs every metric with
`tensorboardX.SummaryWriter.add_scalar(name, value, step)` (common/runner.py:38-39,58-60); neither tensorboard
nor tensorboardX exists in this image, so the TFRecord framing (length, masked crc32c, payload, masked crc32c) and
the Event / Summary protobufs are written directly.  `read_scalars` reads them back (tests)."""
from __future__ import annotations

import os
import socket
import struct
import time
from typing import Dict, Iterator, Tuple

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _event(wall_time: float, step: int, tag: str = None, value: float = None, file_version: str = None) -> bytes:
    # Event: wall_time=1 (double), step=2 (int64), file_version=3 (string), summary=5 {value=1 {tag=1, simple_value=2 (float)}}
    ev = _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _varint((2 << 3) | 0) + _varint(step & ((1 << 64) - 1))
    if file_version is not None:
        ev += _ld(3, file_version.encode())
    if tag is not None:
        val = _ld(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", value)
        ev += _ld(5, _ld(1, val))
    return ev


class SummaryWriter:
    def __init__(self, logdir: str):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}")
        self._f = open(self.path, "ab")
        self._record(_event(time.time(), 0, file_version="brain.Event:2"))

    def _record(self, data: bytes):
        hdr = struct.pack("<Q", len(data))
        self._f.write(hdr + struct.pack("<I", _masked(hdr)) + data + struct.pack("<I", _masked(data)))

    def add_scalar(self, tag: str, value, step: int):
        self._record(_event(time.time(), int(step), tag, float(value)))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_scalars(path: str) -> Iterator[Tuple[str, int, float]]:
    """(tag, step, value) of every scalar event; verifies both checksums of every record."""
    with open(path, "rb") as f:
        buf = f.read()
    i = 0
    while i < len(buf):
        hdr = buf[i:i + 8]; n = struct.unpack("<Q", hdr)[0]
        assert struct.unpack("<I", buf[i + 8:i + 12])[0] == _masked(hdr)
        data = buf[i + 12:i + 12 + n]
        assert struct.unpack("<I", buf[i + 12 + n:i + 16 + n])[0] == _masked(data)
        i += 16 + n
        step, j = 0, 0
        summary = None
        while j < len(data):
            key = data[j]; j += 1
            field, wire = key >> 3, key & 7
            if wire == 1:
                j += 8
            elif wire == 0:
                v = 0; sh = 0
                while True:
                    b = data[j]; j += 1
                    v |= (b & 0x7F) << sh; sh += 7
                    if not b & 0x80:
                        break
                if field == 2:
                    step = v
            elif wire == 2:
                ln = 0; sh = 0
                while True:
                    b = data[j]; j += 1
                    ln |= (b & 0x7F) << sh; sh += 7
                    if not b & 0x80:
                        break
                if field == 5:
                    summary = data[j:j + ln]
                j += ln
        if summary:
            # summary { value { tag, simple_value } }
            assert summary[0] == (1 << 3) | 2
            k = 1; ln = 0; sh = 0
            while True:
                b = summary[k]; k += 1
                ln |= (b & 0x7F) << sh; sh += 7
                if not b & 0x80:
                    break
            val = summary[k:k + ln]
            assert val[0] == (1 << 3) | 2
            tl = val[1]
            tag = val[2:2 + tl].decode()
            rest = val[2 + tl:]
            assert rest[0] == (2 << 3) | 5
            yield tag, step, struct.unpack("<f", rest[1:5])[0]
