"""Video files of rendered frame sequences (the reference writes `video_<expid>_iter<k>_<tag>.mp4` with
imageio.mimwrite(path, to8b(rgbs), fps=30, quality=8): /root/reference/main.py:1096-1103, 1480-1484).  There is no
imageio / ffmpeg on this image and an H.264 encoder is out of scope, so the frames go into a Motion-JPEG AVI (RIFF 'AVI ',
one 'MJPG' video stream, 'idx1' index): every frame is a baseline JPEG from PIL, which any player (and ffmpeg, to transcode
to the reference's mp4) reads.  Host-side file writing, not on the hot path."""
import io
import struct

import numpy as np


def _chunk(fourcc, payload):
    pad = b"\x00" if len(payload) & 1 else b""
    return fourcc + struct.pack("<I", len(payload)) + payload + pad


def _jpeg_quality(quality10):
    """imageio's quality 0..10 scale (its ffmpeg plugin maps it to a bitrate) onto PIL's JPEG quality."""
    return int(min(95, max(30, round(35 + 6.5 * float(quality10)))))


def write_mjpeg_avi(path, frames, fps=30, quality=8):
    """frames: uint8 [N,H,W,3] (numpy, or anything np.asarray takes).  Returns the number of bytes written."""
    from PIL import Image
    frames = np.asarray(frames)
    if frames.ndim != 4 or frames.shape[-1] != 3 or frames.dtype != np.uint8:
        raise ValueError("write_mjpeg_avi: expected uint8 frames [N,H,W,3], got %s %s" % (frames.dtype, frames.shape))
    n, h, w, _ = frames.shape
    q = _jpeg_quality(quality)
    jpegs = []
    for f in frames:
        buf = io.BytesIO()
        Image.fromarray(f).save(buf, format="JPEG", quality=q, subsampling=0)
        jpegs.append(buf.getvalue())
    biggest = max((len(j) for j in jpegs), default=0)
    avih = struct.pack("<14I", int(round(1e6 / fps)), biggest * int(fps), 0, 0x10, n, 0, 1, biggest, w, h, 0, 0, 0, 0)
    strh = (b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, int(fps), 0, n, biggest, 0xFFFFFFFF, 0) +
            struct.pack("<4H", 0, 0, w, h))
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
    strl = b"strl" + _chunk(b"strh", strh) + _chunk(b"strf", strf)
    hdrl = b"hdrl" + _chunk(b"avih", avih) + _chunk(b"LIST", strl)
    movi, index, off = [b"movi"], [], 4
    for j in jpegs:
        c = _chunk(b"00dc", j)
        index.append(struct.pack("<4sIII", b"00dc", 0x10, off, len(j)))
        movi.append(c)
        off += len(c)
    body = b"AVI " + _chunk(b"LIST", hdrl) + _chunk(b"LIST", b"".join(movi)) + _chunk(b"idx1", b"".join(index))
    data = b"RIFF" + struct.pack("<I", len(body)) + body
    with open(path, "wb") as fh:
        fh.write(data)
    return len(data)


def read_mjpeg_avi(path):
    """Frames of a file written by write_mjpeg_avi -> (uint8 [N,H,W,3], fps).  Walks the RIFF tree (hdrl for the rate,
    movi for the '00dc' chunks) — the round trip of the writer's own test, and a reader for the files it produces."""
    from PIL import Image
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        raise ValueError("%s: not a RIFF AVI file" % path)
    fps, frames = None, []

    def walk(lo, hi):
        nonlocal fps
        p = lo
        while p + 8 <= hi:
            cc, size = data[p:p + 4], struct.unpack("<I", data[p + 4:p + 8])[0]
            if cc == b"LIST":
                walk(p + 12, p + 8 + size)
            elif cc == b"strh":
                scale, rate = struct.unpack("<II", data[p + 8 + 20:p + 8 + 28])
                fps = rate / max(scale, 1)
            elif cc == b"00dc":
                frames.append(np.asarray(Image.open(io.BytesIO(data[p + 8:p + 8 + size])).convert("RGB")))
            p += 8 + size + (size & 1)

    walk(12, len(data))
    return (np.stack(frames) if frames else np.zeros((0, 0, 0, 3), np.uint8)), fps
