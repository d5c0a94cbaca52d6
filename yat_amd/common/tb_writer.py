"""TensorBoard-compatible event writer (the reference logs through ``torch.utils.tensorboard.SummaryWriter``:
common/trainer.py:4,138 creates it on the main process, :364-367 ``add_scalar('train/loss' | 'train/lr', v, step)``,
train_sana.py:157 ``add_image('validation/{idx}/{prompt}', CHW uint8, step)``).

The ``tensorboard`` package is not a dependency here: this module writes the ``events.out.tfevents.*`` file directly.
[RECALL tensorboard record format]  Each record is ``u64 length | u32 masked_crc32c(length) | payload |
u32 masked_crc32c(payload)`` (little endian), the payload a serialized ``Event`` protobuf:

    Event   { 1: double wall_time; 2: int64 step; 3: string file_version | 5: Summary summary }
    Summary { 1: repeated Value value }
    Value   { 1: string tag; 2: float simple_value | 4: Image image }
    Image   { 1: int32 height; 2: int32 width; 3: int32 colorspace; 4: bytes encoded_image_string (PNG) }

The first record carries ``file_version = "brain.Event:2"``.  ``read_events`` parses a file back (used by the tests and
handy for inspecting a run without TensorBoard).
"""
from __future__ import annotations

import os
import socket
import struct
import time
import zlib

# ---- CRC-32C (Castagnoli), table driven
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    t = _TABLE
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire encoding (only what Event needs)
def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return _varint((field << 3) | wire)


def _bytes_field(field: int, payload: bytes) -> bytes:
    return _key(field, 2) + _varint(len(payload)) + payload


def _event(wall_time: float, step: int | None = None, file_version: str | None = None, summary: bytes | None = None):
    ev = _key(1, 1) + struct.pack("<d", wall_time)
    if step is not None:
        ev += _key(2, 0) + _varint(int(step))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    if summary is not None:
        ev += _bytes_field(5, summary)
    return ev


def encode_png(chw) -> bytes:
    """[C, H, W] uint8 (C = 1, 3 or 4; anything with ``.shape`` and ``.tobytes()`` after HWC transposition) -> PNG bytes."""
    import numpy as np
    a = np.asarray(chw.detach().cpu().numpy() if hasattr(chw, "detach") else chw)
    if a.dtype != np.uint8:                      # SummaryWriter scales float images in [0, 1] to bytes
        a = (np.clip(a.astype(np.float32), 0.0, 1.0) * 255.0).astype(np.uint8)
    if a.ndim == 2:
        a = a[None]
    c, h, w = a.shape
    color_type = {1: 0, 3: 2, 4: 6}[c]
    hwc = np.ascontiguousarray(a.transpose(1, 2, 0))
    raw = np.concatenate([np.zeros((h, 1), np.uint8), hwc.reshape(h, w * c)], axis=1).tobytes()   # filter byte 0 per row

    def chunk(kind: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class SummaryWriter:
    """The subset of ``torch.utils.tensorboard.SummaryWriter`` the reference uses, same argument meaning."""

    def __init__(self, log_dir: str | None = None, comment: str = "", filename_suffix: str = ""):
        if not log_dir:                          # torch: runs/<Mon DD_HH-MM-SS>_<hostname><comment>
            log_dir = os.path.join("runs", time.strftime("%b%d_%H-%M-%S") + "_" + socket.gethostname() + comment)
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        name = f"events.out.tfevents.{int(time.time()):010d}.{socket.gethostname()}.{os.getpid()}.0{filename_suffix}"
        self.path = os.path.join(log_dir, name)
        self._f = open(self.path, "ab")
        self._write(_event(time.time(), file_version="brain.Event:2"))
        self.flush()

    def _write(self, payload: bytes):
        head = struct.pack("<Q", len(payload))
        self._f.write(head + struct.pack("<I", masked_crc32c(head)) + payload + struct.pack("<I", masked_crc32c(payload)))

    def add_scalar(self, tag: str, scalar_value, global_step=None, walltime=None):
        v = float(scalar_value.item() if hasattr(scalar_value, "item") else scalar_value)
        value = _bytes_field(1, tag.encode()) + _key(2, 5) + struct.pack("<f", v)
        self._write(_event(walltime or time.time(), global_step, summary=_bytes_field(1, value)))

    def add_image(self, tag: str, img_tensor, global_step=None, walltime=None, dataformats: str = "CHW"):
        if dataformats == "HWC":
            img_tensor = img_tensor.permute(2, 0, 1) if hasattr(img_tensor, "permute") else img_tensor.transpose(2, 0, 1)
        elif dataformats != "CHW":
            raise ValueError(f"dataformats {dataformats!r} is not supported")
        c, h, w = (int(s) for s in img_tensor.shape)
        image = (_key(1, 0) + _varint(h) + _key(2, 0) + _varint(w) + _key(3, 0) + _varint(c) +
                 _bytes_field(4, encode_png(img_tensor)))
        value = _bytes_field(1, tag.encode()) + _bytes_field(4, image)
        self._write(_event(walltime or time.time(), global_step, summary=_bytes_field(1, value)))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.flush()
            self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- reader (tests / inspection)
def _read_varint(buf: bytes, i: int):
    n = shift = 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, i
        shift += 7


def _fields(buf: bytes):
    i = 0
    while i < len(buf):
        k, i = _read_varint(buf, i)
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, i = _read_varint(buf, i)
        elif wire == 1:
            v, i = buf[i:i + 8], i + 8
        elif wire == 5:
            v, i = buf[i:i + 4], i + 4
        elif wire == 2:
            n, i = _read_varint(buf, i)
            v, i = buf[i:i + n], i + n
        else:
            raise ValueError(f"wire type {wire}")
        yield field, wire, v


def read_events(path: str):
    """Parse an event file -> list of dicts {wall_time, step, file_version | tag, value | image{height,width,colorspace,png}};
    raises ValueError on any CRC mismatch."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    i = 0
    while i < len(data):
        head = data[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", data[i + 8:i + 12])[0] != masked_crc32c(head):
            raise ValueError("length CRC mismatch")
        payload = data[i + 12:i + 12 + n]
        if struct.unpack("<I", data[i + 12 + n:i + 16 + n])[0] != masked_crc32c(payload):
            raise ValueError("payload CRC mismatch")
        i += 16 + n
        ev = {"step": 0}
        for field, _, v in _fields(payload):
            if field == 1:
                ev["wall_time"] = struct.unpack("<d", v)[0]
            elif field == 2:
                ev["step"] = v
            elif field == 3:
                ev["file_version"] = v.decode()
            elif field == 5:
                for f1, _, value in _fields(v):
                    if f1 != 1:
                        continue
                    for f2, _, x in _fields(value):
                        if f2 == 1:
                            ev["tag"] = x.decode()
                        elif f2 == 2:
                            ev["value"] = struct.unpack("<f", x)[0]
                        elif f2 == 4:
                            img = {}
                            for f3, _, y in _fields(x):
                                img[{1: "height", 2: "width", 3: "colorspace", 4: "png"}[f3]] = y
                            ev["image"] = img
        out.append(ev)
    return out
