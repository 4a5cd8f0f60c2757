"""Output sink: ``save_video(frames, path, fps, quality)`` (diffsynth/utils/data/__init__.py:140-145).

The reference writes H.264 mp4 through imageio / ffmpeg.  Neither is installed in this image; with them present the
reference's writer call is used unchanged.  Without them the frames are written as a Motion-JPEG AVI next to the
requested path (``<path>.avi``, plain RIFF written here with PIL's JPEG encoder: every player opens it) and the path
actually written is returned — never a silent no-op.
"""
import io
import struct

import numpy as np


def _riff_chunk(tag, payload):
    return tag + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")


def _riff_list(tag, kind, payload):
    return _riff_chunk(tag, kind + payload)


def write_mjpeg_avi(frames, path, fps, quality=9):
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
        return write_mjpeg_avi(frames, save_path + ".avi", fps, quality)
    writer = imageio.get_writer(save_path, fps=fps, quality=quality, ffmpeg_params=ffmpeg_params)
    for frame in frames:
        writer.append_data(np.array(frame))
    writer.close()
    return save_path
