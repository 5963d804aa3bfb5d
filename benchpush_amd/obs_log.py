"""`cfg.log_obs` observation dumps of the single-env adapters (observability only).

The reference writes one PNG per observation channel and step when ``cfg.log_obs`` is set (``log_observation``: ship_ice_env.py:412-479, called from
``step`` :352-353; maze_NAMO_env.py:539-595, :482-483): ``<cfg.output_dir>/t<episode_idx>/<t>_<name>.png``, the channel flipped vertically (``np.flip(axis=0)``),
grey colour map.  It does so through a matplotlib figure (``imshow`` + ``savefig(bbox_inches='tight')``), so its files are the channel resampled to the figure's
size with matplotlib's auto-scaled grey levels.  Here the files carry the channel itself: 8-bit greyscale, one pixel per cell, same names, same directory layout,
same vertical flip -- no plotting library on the path (the PNG encoder below is zlib + struct).  Box-delivery and area-clearing log from inside ``render()``
(area_clearing.py:1151), which is outside the accelerated path: their adapters refuse ``render.log_obs: true`` instead of ignoring it.
"""
import os
import struct
import zlib

import numpy as np

__all__ = ["write_gray_png", "read_gray_png", "dump_channels", "refuse_render_log_obs"]


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_gray_png(path, img):
    """8-bit greyscale PNG of a 2-D array: uint8 as is, floats in [0, 1] scaled like the observation itself (``(x * 255).astype(uint8)``)."""
    a = np.asarray(img)
    if a.ndim != 2:
        raise ValueError("write_gray_png wants a 2-D array, got shape %r" % (a.shape,))
    if a.dtype != np.uint8:
        a = (np.clip(a.astype(np.float64), 0.0, 1.0) * 255).astype(np.uint8)
    h, w = a.shape
    raw = np.empty((h, w + 1), np.uint8)
    raw[:, 0] = 0                      # filter type 0 on every scanline
    raw[:, 1:] = a
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + _chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def read_gray_png(path):
    """Inverse of `write_gray_png` (tests): uint8 [H, W]."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(b):
        n, tag = struct.unpack(">I", b[pos: pos + 4])[0], b[pos + 4: pos + 8]
        data = b[pos + 8: pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", data[:10])
            assert depth == 8 and ctype == 0
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    assert not raw[:, 0].any()
    return raw[:, 1:].copy()


def dump_channels(output_dir, episode_idx, t, channels):
    """One file per (name, 2-D channel): ``<output_dir>/t<episode_idx>/<t>_<name>.png``, flipped vertically like the reference's ``np.flip(..., axis=0)``.
    Returns the paths written."""
    if not output_dir:
        raise ValueError("cfg.log_obs is set but cfg.output_dir is empty: the reference writes to os.path.join(cfg.output_dir, 't<episode>')")
    directory = os.path.join(str(output_dir), "t" + str(episode_idx))
    os.makedirs(directory, exist_ok=True)
    out = []
    for name, img in channels.items():
        fp = os.path.join(directory, "%s_%s.png" % (t, name))
        write_gray_png(fp, np.flip(np.asarray(img), axis=0))
        out.append(fp)
    return out


def refuse_render_log_obs(cfg, env_name):
    """box-delivery-v0 / area-clearing-v0: ``cfg.render.log_obs`` dumps the observation from inside ``render()`` (area_clearing.py:1141-1165), and ``render()`` is
    outside the accelerated path -- a config that asks for it is refused loudly instead of being accepted and ignored."""
    r = cfg.get("render") if hasattr(cfg, "get") else None
    if r is not None and r.get("log_obs"):
        raise NotImplementedError("%s: cfg.render.log_obs is written by render() in the reference, and render() is not part of the accelerated path; "
                                  "log the observations returned by step() instead (benchpush_amd.obs_log.dump_channels)" % env_name)
