"""The encoder-free .mp4 writer (tooncrafter_amd/mp4.py; reference: torchvision.io.write_video with h264,
scripts/evaluation/inference.py:154-155): the file is re-read HERE with an independent minimal ISO-BMFF / H.264 I_PCM
parser -- box tree, sample table, NAL framing, emulation prevention, SPS / PPS / slice-header syntax bit by bit -- and
the decoded planes must equal the yuv420p conversion of the frames exactly."""
import struct

import numpy as np
import pytest
import torch

from tooncrafter_amd import mp4


def _boxes(buf, start, end):
    out, o = [], start
    while o < end:
        size, kind = struct.unpack(">I4s", buf[o:o + 8])
        assert size >= 8 and o + size <= end, (kind, size)
        out.append((kind, o + 8, o + size))
        o += size
    assert o == end
    return out


def _find(buf, path, start=0, end=None):
    end = len(buf) if end is None else end
    for kind, s, e in _boxes(buf, start, end):
        if kind == path[0]:
            return (s, e) if len(path) == 1 else _find(buf, path[1:], s, e)
    raise KeyError(path)


class _Rd:
    def __init__(self, data):
        self.bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))
        self.p = 0

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | int(self.bits[self.p])
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k % 2 else -(k // 2)


def _rbsp(nal):
    out, zeros = bytearray(), 0
    for b in nal[1:]:
        if zeros >= 2 and b == 3:
            zeros = 0
            continue
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def _decode(path):
    buf = open(path, "rb").read()
    top = _boxes(buf, 0, len(buf))
    assert [k for k, _, _ in top] == [b"ftyp", b"moov", b"mdat"]
    stbl = (b"moov", b"trak", b"mdia", b"minf", b"stbl")
    s, e = _find(buf, stbl + (b"stsd",))
    entry = s + 8                                             # version/flags + entry_count
    size, kind = struct.unpack(">I4s", buf[entry:entry + 8])
    assert kind == b"avc1"
    w, h = struct.unpack(">HH", buf[entry + 8 + 24:entry + 8 + 28])
    avcc_s, avcc_e = _find(buf, (b"avcC",), entry + 8 + 78, entry + size)
    a = buf[avcc_s:avcc_e]
    assert a[0] == 1 and a[4] & 3 == 3 and a[5] & 31 == 1
    sps_len = struct.unpack(">H", a[6:8])[0]
    sps = a[8:8 + sps_len]
    assert a[8 + sps_len] == 1
    pps_len = struct.unpack(">H", a[9 + sps_len:11 + sps_len])[0]
    pps = a[11 + sps_len:11 + sps_len + pps_len]
    assert sps[0] == 0x67 and pps[0] == 0x68 and (a[1], a[2], a[3]) == (sps[1], sps[2], sps[3])
    # --- SPS
    r = _Rd(_rbsp(sps))
    assert r.u(8) == 66
    r.u(8)
    r.u(8)
    assert r.ue() == 0 and r.ue() == 0 and r.ue() == 2        # sps id, log2_max_frame_num_minus4, poc type
    r.ue()
    assert r.u(1) == 0
    mbw, mbh = r.ue() + 1, r.ue() + 1
    assert r.u(1) == 1 and r.u(1) == 1                        # frame_mbs_only, direct_8x8
    crop = [0, 0, 0, 0]
    if r.u(1):
        crop = [r.ue() for _ in range(4)]
    assert r.u(1) == 0 and r.u(1) == 1                        # no VUI, stop bit
    assert (mbw * 16 - 2 * crop[1], mbh * 16 - 2 * crop[3]) == (w, h)
    # --- PPS
    r = _Rd(_rbsp(pps))
    assert [r.ue(), r.ue(), r.u(1), r.u(1), r.ue(), r.ue(), r.ue(), r.u(1), r.u(2)] == [0] * 9
    assert [r.se(), r.se(), r.se()] == [0, 0, 0] and r.u(1) == 1 and r.u(1) == 0 and r.u(1) == 0 and r.u(1) == 1
    # --- sample table
    s, e = _find(buf, stbl + (b"stsz",))
    _, count = struct.unpack(">II", buf[s + 4:s + 12])
    sizes = struct.unpack(f">{count}I", buf[s + 12:s + 12 + 4 * count])
    s, e = _find(buf, stbl + (b"stco",))
    offs = struct.unpack(f">{count}I", buf[s + 8:s + 8 + 4 * count])
    s, e = _find(buf, stbl + (b"stts",))
    assert struct.unpack(">III", buf[s + 4:s + 16])[1] == count
    s, e = _find(buf, (b"moov", b"trak", b"mdia", b"mdhd"))
    timescale, duration = struct.unpack(">II", buf[s + 12:s + 20])
    frames = []
    for i, (o, n) in enumerate(zip(offs, sizes)):
        ln = struct.unpack(">I", buf[o:o + 4])[0]
        assert ln + 4 == n
        nal = buf[o + 4:o + n]
        assert nal[0] == 0x65                                 # nal_ref_idc 3, IDR slice
        rb = _rbsp(nal)
        r = _Rd(rb[:16])
        assert r.ue() == 0 and r.ue() == 7 and r.ue() == 0 and r.u(4) == 0
        assert r.ue() == i % 2                                # idr_pic_id alternates
        assert r.u(1) == 0 and r.u(1) == 0 and r.se() == 0 and r.ue() == 1
        assert r.ue() == 25                                   # first mb_type: I_PCM
        pos = (r.p + 7) // 8
        mb = np.frombuffer(rb[pos:], dtype=np.uint8)
        nmb = mbw * mbh
        assert mb.size == nmb * 386 - 2 + 1 and mb[-1] == 0x80
        mb = np.concatenate([np.array([0x0D, 0x00], np.uint8), mb[:-1]]).reshape(nmb, 386)
        assert (mb[:, 0] == 0x0D).all() and (mb[:, 1] == 0).all()
        y = mb[:, 2:258].reshape(mbh, mbw, 16, 16).transpose(0, 2, 1, 3).reshape(mbh * 16, mbw * 16)
        cb = mb[:, 258:322].reshape(mbh, mbw, 8, 8).transpose(0, 2, 1, 3).reshape(mbh * 8, mbw * 8)
        cr = mb[:, 322:386].reshape(mbh, mbw, 8, 8).transpose(0, 2, 1, 3).reshape(mbh * 8, mbw * 8)
        frames.append((y[:h, :w], cb[:h // 2, :w // 2], cr[:h // 2, :w // 2]))
    return dict(w=w, h=h, fps=timescale * count / duration, frames=frames)


@pytest.mark.parametrize("t,h,w", [(3, 32, 48), (2, 40, 72), (16, 320, 512)])
def test_mp4_roundtrip(tmp_path, t, h, w):
    g = np.random.default_rng(5)
    frames = g.integers(0, 256, size=(t, h, w, 3), dtype=np.uint8)
    frames[0, :4, :8] = 0                                     # runs of zero bytes: emulation prevention must kick in
    frames[-1] = 255
    path = mp4.write_mp4(str(tmp_path / "clip.mp4"), torch.from_numpy(frames), fps=8)
    d = _decode(path)
    assert (d["w"], d["h"]) == (w, h) and abs(d["fps"] - 8.0) < 1e-9 and len(d["frames"]) == t
    for i in range(t):
        y, cb, cr = mp4.rgb_to_yuv420(frames[i])
        assert np.array_equal(d["frames"][i][0], y) and np.array_equal(d["frames"][i][1], cb) and np.array_equal(d["frames"][i][2], cr)


def test_rgb_to_yuv_reference_points():
    px = np.array([[[0, 0, 0], [255, 255, 255]], [[255, 0, 0], [0, 0, 255]]], dtype=np.uint8)
    y, cb, cr = mp4.rgb_to_yuv420(px)
    assert y.tolist() == [[16, 235], [81, 41]]                # BT.601 limited-range luma of black / white / red / blue
    assert cb.shape == (1, 1) and cr.shape == (1, 1)


def test_save_results_writes_mp4_without_an_encoder(tmp_path, monkeypatch):
    """`save_results_seperate` (inference.py:135-155) on an image without torchvision: a real .mp4 under the reference's
    file name."""
    from tooncrafter_amd import ops, output
    from emu_ops import EmuOps
    old = ops.set_backend(EmuOps())
    try:
        samples = torch.rand(1, 3, 4, 32, 48) * 2 - 1
        written = output.save_results_seperate(["a prompt"], samples, "clip0.png", str(tmp_path / "samples"), fps=8)
    finally:
        ops.set_backend(old)
    assert len(written) == 1 and written[0].endswith("clip0_sample0.mp4")
    d = _decode(written[0])
    assert (d["w"], d["h"], len(d["frames"])) == (48, 32, 4)
