"""Output sink: ``save_video(frames, path, fps, quality)`` (diffsynth/utils/data/__init__.py:140-145).

The reference writes H.264 mp4 through imageio / ffmpeg.  Neither is installed in this image; with them present the
reference's writer call is used unchanged.  Without them the frames are written as Motion-JPEG (PIL's JPEG encoder) into
the container the requested file name asks for, AT the requested path: ``.mp4`` / ``.m4v`` / ``.mov`` -> an ISO base media
file with one 'mp4v' track of object type 0x6C (JPEG; ffmpeg's own "mjpeg in mp4" form), anything else -> a RIFF AVI.
The path written is returned — never a silent no-op, never a different file name.
"""
import io
import struct

import numpy as np


def _riff_chunk(tag, payload):
    return tag + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")


def _riff_list(tag, kind, payload):
    return _riff_chunk(tag, kind + payload)


def _encode_jpegs(frames, quality):
    """frames: PIL images (or HxWx3 uint8 arrays) of one size.  quality 0..10 like imageio's -> JPEG quality 10..100."""
    from PIL import Image
    jpeg_q = int(min(100, max(10, round(float(quality) * 10))))
    blobs = []
    for f in frames:
        im = f if isinstance(f, Image.Image) else Image.fromarray(np.asarray(f, dtype=np.uint8))
        buf = io.BytesIO()
        im.convert("RGB").save(buf, format="JPEG", quality=jpeg_q)
        blobs.append(buf.getvalue())
    if not blobs:
        raise ValueError("no frames to write")
    w, h = (frames[0].size if isinstance(frames[0], Image.Image) else (np.asarray(frames[0]).shape[1], np.asarray(frames[0]).shape[0]))
    return blobs, w, h


def _box(tag, payload):
    return struct.pack(">I", 8 + len(payload)) + tag + payload


def _full_box(tag, version, flags, payload):
    return _box(tag, struct.pack(">I", (version << 24) | flags) + payload)


def write_mjpeg_mp4(frames, path, fps, quality=9):
    """ISO base media file (ISO/IEC 14496-12): ftyp, mdat (the JPEG frames back to back, one chunk), moov with one video
    track whose sample entry is 'mp4v' + esds objectTypeIndication 0x6C (JPEG, ISO/IEC 14496-1 table 5): every sample is a
    sync sample, constant sample duration 1000 at timescale round(1000 * fps)."""
    blobs, w, h = _encode_jpegs(frames, quality)
    n, delta, timescale = len(blobs), 1000, int(round(fps * 1000))
    duration = n * delta
    total = sum(len(b) for b in blobs)
    if total >= (1 << 32) - 16:
        raise ValueError("clip too large for 32-bit mp4 chunk offsets")
    matrix = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isomiso2mp41")
    mdat = _box(b"mdat", b"".join(blobs))
    mvhd = _full_box(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration) + struct.pack(">IH", 0x10000, 0x0100)
                     + b"\x00" * 10 + matrix + b"\x00" * 24 + struct.pack(">I", 2))
    tkhd = _full_box(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration) + b"\x00" * 8 + struct.pack(">HHHH", 0, 0, 0, 0)
                     + matrix + struct.pack(">II", w << 16, h << 16))
    mdhd = _full_box(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, duration, 0x55C4, 0))
    hdlr = _full_box(b"hdlr", 0, 0, struct.pack(">I", 0) + b"vide" + b"\x00" * 12 + b"VideoHandler\x00")
    vmhd = _full_box(b"vmhd", 0, 1, struct.pack(">HHHH", 0, 0, 0, 0))
    dinf = _box(b"dinf", _full_box(b"dref", 0, 0, struct.pack(">I", 1) + _full_box(b"url ", 0, 1, b"")))
    dec_cfg = bytes([0x04, 13, 0x6C, 0x11]) + b"\x00\x00\x00" + struct.pack(">II", 0, 0)
    es = bytes([0x03, 3 + len(dec_cfg) + 3]) + struct.pack(">HB", 0, 0) + dec_cfg + bytes([0x06, 1, 0x02])
    esds = _full_box(b"esds", 0, 0, es)
    entry = b"\x00" * 6 + struct.pack(">H", 1) + b"\x00" * 16 + struct.pack(">HHII", w, h, 0x00480000, 0x00480000) \
        + struct.pack(">IH", 0, 1) + b"\x00" * 32 + struct.pack(">Hh", 0x0018, -1) + esds
    stsd = _full_box(b"stsd", 0, 0, struct.pack(">I", 1) + _box(b"mp4v", entry))
    stts = _full_box(b"stts", 0, 0, struct.pack(">III", 1, n, delta))
    stsc = _full_box(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1))
    stsz = _full_box(b"stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", len(b)) for b in blobs))
    stco = _full_box(b"stco", 0, 0, struct.pack(">II", 1, len(ftyp) + 8))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
    minf = _box(b"minf", vmhd + dinf + stbl)
    mdia = _box(b"mdia", mdhd + hdlr + minf)
    moov = _box(b"moov", mvhd + _box(b"trak", tkhd + mdia))
    with open(path, "wb") as f:
        f.write(ftyp + mdat + moov)
    return path


def write_mjpeg_avi(frames, path, fps, quality=9):
    blobs, w, h = _encode_jpegs(frames, quality)
    n, biggest = len(blobs), max(len(b) for b in blobs)
    usec = int(round(1e6 / fps))
    avih = struct.pack("<14I", usec, biggest * int(round(fps)), 0, 0x10, n, 0, 1, biggest, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1000, int(round(fps * 1000)), 0, n, biggest, 0xFFFFFFFF, 0) \
        + struct.pack("<4H", 0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
    hdrl = _riff_list(b"LIST", b"hdrl", _riff_chunk(b"avih", avih)
                      + _riff_list(b"LIST", b"strl", _riff_chunk(b"strh", strh) + _riff_chunk(b"strf", strf)))
    movi_payload, index, offset = b"", b"", 4
    for b in blobs:
        chunk = _riff_chunk(b"00dc", b)
        index += b"00dc" + struct.pack("<III", 0x10, offset, len(b))
        movi_payload += chunk
        offset += len(chunk)
    body = b"AVI " + hdrl + _riff_list(b"LIST", b"movi", movi_payload) + _riff_chunk(b"idx1", index)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return path


def save_video(frames, save_path, fps, quality=9, ffmpeg_params=None):
    try:
        import imageio
    except ModuleNotFoundError:
        if str(save_path).lower().endswith((".mp4", ".m4v", ".mov")):
            return write_mjpeg_mp4(frames, save_path, fps, quality)
        return write_mjpeg_avi(frames, save_path, fps, quality)
    writer = imageio.get_writer(save_path, fps=fps, quality=quality, ffmpeg_params=ffmpeg_params)
    for frame in frames:
        writer.append_data(np.array(frame))
    writer.close()
    return save_path
