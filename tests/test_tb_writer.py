"""TensorBoard event writer (yat_amd/common/tb_writer.py): record framing, CRC-32C known answers, protobuf round trip."""
import struct
import zlib

import numpy as np
import torch

from yat_amd.common import tb_writer as tb


def test_crc32c_known_answers():
    assert tb.crc32c(b"123456789") == 0xE3069283                 # the CRC-32C check value (RFC 3720 B.4 family)
    assert tb.crc32c(b"\x00" * 32) == 0x8A9136AA                  # RFC 3720 B.4: 32 bytes of zeros
    assert tb.crc32c(b"\xff" * 32) == 0x62A8AB43                  # RFC 3720 B.4: 32 bytes of ones
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E              # RFC 3720 B.4: incrementing bytes


def test_scalar_and_image_round_trip(tmp_path):
    w = tb.SummaryWriter(str(tmp_path / "run"))
    w.add_scalar("train/loss", torch.tensor(0.125), 3)
    w.add_scalar("train/lr", 1e-4, 3)
    img = (torch.arange(3 * 5 * 7) % 256).to(torch.uint8).view(3, 5, 7)
    w.add_image("validation/0/a prompt", img, 4)
    w.add_image("validation/1/float", torch.linspace(0, 1, 2 * 2).view(1, 2, 2), 4)
    w.close()
    ev = tb.read_events(w.path)                                   # verifies both CRCs of every record
    assert ev[0]["file_version"] == "brain.Event:2"
    assert (ev[1]["tag"], ev[1]["step"], ev[1]["value"]) == ("train/loss", 3, 0.125)
    assert ev[2]["tag"] == "train/lr" and abs(ev[2]["value"] - 1e-4) < 1e-10
    im = ev[3]["image"]
    assert (im["height"], im["width"], im["colorspace"]) == (5, 7, 3) and ev[3]["step"] == 4
    png = im["png"]
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    # decode the single IDAT chunk by hand: filter byte 0 + RGB rows
    i, idat = 8, b""
    while i < len(png):
        (n,) = struct.unpack(">I", png[i:i + 4])
        kind, body = png[i + 4:i + 8], png[i + 8:i + 8 + n]
        assert struct.unpack(">I", png[i + 8 + n:i + 12 + n])[0] == zlib.crc32(kind + body) & 0xFFFFFFFF
        if kind == b"IDAT":
            idat += body
        i += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(5, 1 + 7 * 3)
    assert (rows[:, 0] == 0).all()
    assert np.array_equal(rows[:, 1:].reshape(5, 7, 3), img.permute(1, 2, 0).numpy())
    assert ev[4]["image"]["colorspace"] == 1


def test_corruption_is_detected(tmp_path):
    w = tb.SummaryWriter(str(tmp_path))
    w.add_scalar("x", 1.0, 0)
    w.close()
    data = bytearray(open(w.path, "rb").read())
    data[-6] ^= 1
    open(w.path, "wb").write(bytes(data))
    try:
        tb.read_events(w.path)
    except ValueError:
        return
    raise AssertionError("corrupted payload not detected")
