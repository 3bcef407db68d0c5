"""A self-contained `.mp4` writer for images without a video encoder (no torchvision / PyAV / ffmpeg).

The reference writes its clips with `torchvision.io.write_video(path, frames, fps, video_codec='h264',
options={'crf': '10'})` (scripts/evaluation/inference.py:154-155): libx264, yuv420p.  This module produces the same
CONTAINER and CODEC -- ISO base media file, one `avc1` video track, H.264 -- without an encoder library: every macroblock
is coded as I_PCM (ITU-T H.264 7.3.5: raw 8-bit samples, no transform, no entropy coding of residuals), every picture
is an IDR picture.  The stream is lossless with respect to the yuv420p conversion (libx264 at crf 10 is not), about
1.5 bytes per pixel, and decodes with any H.264 decoder (Baseline profile syntax only).

Bitstream (all numbers refer to H.264 clauses):
  SPS 7.3.2.1.1: profile_idc 66, level 5.1, pic_order_cnt_type 2, max_num_ref_frames 1, frame_mbs_only, cropping to the
                 true size when width / height are not multiples of 16.
  PPS 7.3.2.2:   CAVLC, one slice group, deblocking_filter_control_present (the slices switch the loop filter off).
  IDR slice 7.3.3 + 7.3.4: slice_type 7 (I), per macroblock `mb_type` = ue(25) (I_PCM, table 7-11), zero bits up to the
                 byte boundary, 256 luma + 64 Cb + 64 Cr bytes; rbsp_slice_trailing_bits.
  NAL 7.3.1 / 7.4.1: emulation prevention (00 00 0x -> 00 00 03 0x).
Container (ISO/IEC 14496-12 / -15): ftyp, moov (mvhd, trak: tkhd, mdia: mdhd, hdlr, minf: vmhd, dinf, stbl: stsd with
avc1 + avcC, stts, stsc, stsz, stco), mdat with 4-byte length-prefixed NAL units, one sample per chunk.
"""
from __future__ import annotations

import re
import struct
from typing import List

import numpy as np


# ------------------------------------------------------------------------------------------------ bit writer
class _Bits:
    def __init__(self):
        self.bits: List[int] = []

    def u(self, n: int, v: int):
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    def ue(self, v: int):                       # Exp-Golomb, 9.1
        v += 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)

    def se(self, v: int):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def align_zero(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def trailing(self):                         # rbsp_trailing_bits: stop bit + alignment
        self.bits.append(1)
        self.align_zero()

    def bytes(self) -> bytes:
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, dtype=np.uint8)).tobytes()


def _nal(ref_idc: int, unit_type: int, rbsp: bytes) -> bytes:
    """NAL unit = header byte + RBSP with emulation prevention bytes (7.4.1)."""
    body = re.sub(rb"\x00\x00(?=[\x00-\x03])", b"\x00\x00\x03", rbsp)
    return bytes([(ref_idc << 5) | unit_type]) + body


def _sps(width: int, height: int) -> bytes:
    mbw, mbh = (width + 15) // 16, (height + 15) // 16
    b = _Bits()
    b.u(8, 66)                                  # profile_idc: Baseline
    b.u(8, 0b11000000)                          # constraint_set0/1 flags, reserved zero bits
    b.u(8, 51)                                  # level_idc 5.1 (I_PCM bit rates are far above the lower levels)
    b.ue(0)                                     # seq_parameter_set_id
    b.ue(0)                                     # log2_max_frame_num_minus4
    b.ue(2)                                     # pic_order_cnt_type 2: output order = decoding order
    b.ue(1)                                     # max_num_ref_frames (every picture is an IDR picture; IDR pictures are reference pictures)
    b.u(1, 0)                                   # gaps_in_frame_num_value_allowed_flag
    b.ue(mbw - 1)
    b.ue(mbh - 1)
    b.u(1, 1)                                   # frame_mbs_only_flag
    b.u(1, 1)                                   # direct_8x8_inference_flag
    crop_r, crop_b = mbw * 16 - width, mbh * 16 - height
    if crop_r or crop_b:
        if crop_r % 2 or crop_b % 2:
            raise ValueError("4:2:0 cropping needs even width and height")
        b.u(1, 1)
        b.ue(0); b.ue(crop_r // 2); b.ue(0); b.ue(crop_b // 2)      # in chroma sample units (CropUnit = 2)
    else:
        b.u(1, 0)
    b.u(1, 0)                                   # vui_parameters_present_flag
    b.trailing()
    return _nal(3, 7, b.bytes())


def _pps() -> bytes:
    b = _Bits()
    b.ue(0); b.ue(0)                            # pic_parameter_set_id, seq_parameter_set_id
    b.u(1, 0)                                   # entropy_coding_mode_flag: CAVLC
    b.u(1, 0)                                   # bottom_field_pic_order_in_frame_present_flag
    b.ue(0)                                     # num_slice_groups_minus1
    b.ue(0); b.ue(0)                            # num_ref_idx_l0/l1_default_active_minus1
    b.u(1, 0); b.u(2, 0)                        # weighted_pred_flag, weighted_bipred_idc
    b.se(0); b.se(0); b.se(0)                   # pic_init_qp_minus26, pic_init_qs_minus26, chroma_qp_index_offset
    b.u(1, 1)                                   # deblocking_filter_control_present_flag
    b.u(1, 0); b.u(1, 0)                        # constrained_intra_pred_flag, redundant_pic_cnt_present_flag
    b.trailing()
    return _nal(3, 8, b.bytes())


def rgb_to_yuv420(frame: np.ndarray):
    """(h, w, 3) uint8 RGB -> Y (h, w), Cb, Cr (h/2, w/2) uint8: BT.601 limited range, chroma = mean of the 2 x 2 block
    (what swscale's yuv420p conversion of libx264 pipelines computes, up to its rounding)."""
    f = frame.astype(np.float32)
    r, g, b = f[..., 0], f[..., 1], f[..., 2]
    y = 16.0 + (65.481 * r + 128.553 * g + 24.966 * b) / 255.0
    cb = 128.0 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255.0
    cr = 128.0 + (112.0 * r - 93.786 * g - 18.214 * b) / 255.0
    h, w = y.shape
    sub = lambda c: c.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
    q = lambda c: np.clip(np.rint(c), 0, 255).astype(np.uint8)
    return q(y), q(sub(cb)), q(sub(cr))


def _idr_slice(y: np.ndarray, cb: np.ndarray, cr: np.ndarray, idr_pic_id: int) -> bytes:
    """One IDR picture as a single slice of I_PCM macroblocks.  y: (16 mbh, 16 mbw), cb / cr: (8 mbh, 8 mbw)."""
    mbh, mbw = y.shape[0] // 16, y.shape[1] // 16
    b = _Bits()
    b.ue(0)                                     # first_mb_in_slice
    b.ue(7)                                     # slice_type: I, and all slices of the picture are I
    b.ue(0)                                     # pic_parameter_set_id
    b.u(4, 0)                                   # frame_num (IDR: 0)
    b.ue(idr_pic_id)
    b.u(1, 0); b.u(1, 0)                        # dec_ref_pic_marking: no_output_of_prior_pics_flag, long_term_reference_flag
    b.se(0)                                     # slice_qp_delta
    b.ue(1)                                     # disable_deblocking_filter_idc = 1
    b.ue(25)                                    # mb_type of the first macroblock: I_PCM
    b.align_zero()                              # pcm_alignment_zero_bit
    head = b.bytes()
    # every further macroblock starts byte-aligned: ue(25) = 000011010 + 7 alignment zeros = 0x0D 0x00
    ymb = y.reshape(mbh, 16, mbw, 16).transpose(0, 2, 1, 3).reshape(mbh * mbw, 256)
    cbmb = cb.reshape(mbh, 8, mbw, 8).transpose(0, 2, 1, 3).reshape(mbh * mbw, 64)
    crmb = cr.reshape(mbh, 8, mbw, 8).transpose(0, 2, 1, 3).reshape(mbh * mbw, 64)
    mbs = np.empty((mbh * mbw, 2 + 384), dtype=np.uint8)
    mbs[:, 0], mbs[:, 1] = 0x0D, 0x00
    mbs[:, 2:258], mbs[:, 258:322], mbs[:, 322:386] = ymb, cbmb, crmb
    body = mbs.tobytes()[2:]                    # the first macroblock's mb_type sits in `head`
    return _nal(3, 5, head + body + b"\x80")    # rbsp_slice_trailing_bits: stop bit + zeros


# ------------------------------------------------------------------------------------------------ container
def _box(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I", 8 + len(payload)) + kind + payload


def _full(kind: bytes, version: int, flags: int, payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


_MATRIX = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def encode_h264_ipcm(frames: np.ndarray):
    """frames: (t, h, w, 3) uint8 -> (sps NAL, pps NAL, [one IDR NAL per frame])."""
    t, h, w, c = frames.shape
    if c != 3 or frames.dtype != np.uint8:
        raise ValueError("frames must be (t, h, w, 3) uint8")
    if h % 2 or w % 2:
        raise ValueError("yuv420p needs even width and height")
    ph, pw = (h + 15) // 16 * 16, (w + 15) // 16 * 16
    slices = []
    for i in range(t):
        y, cb, cr = rgb_to_yuv420(frames[i])
        if (ph, pw) != (h, w):                  # pad to whole macroblocks by edge replication (cropped away by the SPS)
            y = np.pad(y, ((0, ph - h), (0, pw - w)), mode="edge")
            cb = np.pad(cb, ((0, (ph - h) // 2), (0, (pw - w) // 2)), mode="edge")
            cr = np.pad(cr, ((0, (ph - h) // 2), (0, (pw - w) // 2)), mode="edge")
        slices.append(_idr_slice(y, cb, cr, i & 1))
    return _sps(w, h), _pps(), slices


def write_mp4(path: str, frames, fps: int = 8) -> str:
    """frames: (t, h, w, 3) uint8 (numpy array or torch tensor on the CPU) -> an H.264 .mp4 at `path`."""
    frames = np.ascontiguousarray(frames.numpy() if hasattr(frames, "numpy") else frames)
    t, h, w, _ = frames.shape
    sps, pps, slices = encode_h264_ipcm(frames)
    samples = [struct.pack(">I", len(n)) + n for n in slices]           # AVCC: 4-byte NAL lengths
    timescale, delta = int(fps) * 1000, 1000
    duration = t * delta

    avcc = bytes([1, sps[1], sps[2], sps[3], 0xFC | 3, 0xE0 | 1]) + struct.pack(">H", len(sps)) + sps + \
        bytes([1]) + struct.pack(">H", len(pps)) + pps
    avc1 = _box(b"avc1", b"\x00" * 6 + struct.pack(">H", 1) + b"\x00" * 16 + struct.pack(">HH", w, h) +
                struct.pack(">II", 0x00480000, 0x00480000) + b"\x00" * 4 + struct.pack(">H", 1) + b"\x00" * 32 +
                struct.pack(">Hh", 0x0018, -1) + _box(b"avcC", avcc))
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isomiso2avc1mp41")
    sizes = [len(s) for s in samples]

    def moov(first_offset: int) -> bytes:
        offs, o = [], first_offset
        for s in sizes:
            offs.append(o)
            o += s
        stbl = _box(b"stbl",
                    _full(b"stsd", 0, 0, struct.pack(">I", 1) + avc1) +
                    _full(b"stts", 0, 0, struct.pack(">III", 1, t, delta)) +
                    _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, 1, 1)) +
                    _full(b"stsz", 0, 0, struct.pack(">II", 0, t) + struct.pack(f">{t}I", *sizes)) +
                    _full(b"stco", 0, 0, struct.pack(">I", t) + struct.pack(f">{t}I", *offs)))
        minf = _box(b"minf", _full(b"vmhd", 0, 1, b"\x00" * 8) +
                    _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b""))) + stbl)
        mdia = _box(b"mdia", _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, duration, 0x55C4, 0)) +
                    _full(b"hdlr", 0, 0, struct.pack(">I", 0) + b"vide" + b"\x00" * 12 + b"VideoHandler\x00") + minf)
        tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration) + b"\x00" * 8 +
                     struct.pack(">hhhH", 0, 0, 0, 0) + _MATRIX + struct.pack(">II", w << 16, h << 16))
        mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIIIIH", 0, 0, timescale, duration, 0x10000, 0x100) + b"\x00" * 10 +
                     _MATRIX + b"\x00" * 24 + struct.pack(">I", 2))
        return _box(b"moov", mvhd + _box(b"trak", tkhd + mdia))

    head_len = len(ftyp) + len(moov(0)) + 8                             # moov's size does not depend on the offsets
    with open(path, "wb") as f:
        f.write(ftyp)
        f.write(moov(head_len))
        f.write(struct.pack(">I", 8 + sum(sizes)) + b"mdat")
        for s in samples:
            f.write(s)
    return path
