"""Object crops resized on the device, bit-exact with the reference's host-side PIL path.

The reference expands every detected box by 20 %, crops it out of the BGR frame, swaps to RGB and hands a PIL image to
the captioner, whose HF processor resizes it with ``Image.resize((S, S), resample=BICUBIC)``
(reference ``detector/pseudolabeler.py:629-643,670-675``; ``captioner/models/blip/blip.py`` processor call).  Here the
crop rectangles stay on the host (integer arithmetic), the integer filter tables of Pillow's resample are built on the
host in doubles (O(S) per box, `pil_bicubic_coeffs` - a restatement of Pillow's ``Resample.c::precompute_coeffs`` +
``normalize_coeffs_8bpc``), and every per-pixel operation runs in ``cap_crop_resize_u8`` (csrc/preprocess.hip).

No CPU fallback: without the HIP library / a GPU this module raises.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _native as N

PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=4096)
def pil_bicubic_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's bicubic tables for resizing `in_size` samples to `out_size` (whole-image box):
    bounds int32 [out, 2] = (first input index, tap count), coeffs int32 [out, ksize] (22-bit fixed point).
    Same double-precision operations in the same order as Resample.c, vectorised over the output index."""
    if in_size < 1 or out_size < 1:
        raise ValueError(f"resize {in_size} -> {out_size}")
    scale = in_size / out_size
    filterscale = scale if scale > 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast: truncation, values are > -1
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    x = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * inv)
    a = -0.5
    w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0, np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))
    valid = np.arange(ksize)[None, :] < xmax[:, None]
    w = np.where(valid, w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):                    # Pillow sums the weights in tap order; padded taps add exactly 0.0
        ww = ww + w[:, t]
    w = np.where((ww != 0.0)[:, None], w / np.where(ww != 0.0, ww, 1.0)[:, None], w)
    fixed = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)   # (int): toward zero
    fixed = np.where(valid, fixed, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    bounds.setflags(write=False); fixed.setflags(write=False)
    return bounds, fixed


def shorter_side_geometry(w: int, h: int, size: int) -> Tuple[int, int, int, int]:
    """open_clip's inference transform as the CoCa plugin applies it (captioner/models/coca/coca.py::preprocess): bicubic
    resize of the shorter side to `size`, then a centre crop.  -> (resized width, resized height, left, top)."""
    scale = size / min(w, h)
    nw, nh = max(size, round(w * scale)), max(size, round(h * scale))
    return nw, nh, (nw - size) // 2, (nh - size) // 2


def _ksize(in_size: np.ndarray, out_size: np.ndarray) -> int:
    """Widest tap row any of the boxes needs: Pillow's ksize = 2 ceil(support) + 1, support = 2 max(in / out, 1)."""
    scale = np.maximum(in_size.astype(np.float64) / out_size.astype(np.float64), 1.0)
    return int((np.ceil(2.0 * scale).astype(np.int64) * 2 + 1).max())


def crop_resize_u8(frame, rects: Sequence[Sequence[int]], size: int, bgr: bool = False,
                   device: str | torch.device = "cuda:0", center_crop: bool = False, tables: str = "device") -> torch.Tensor:
    """frame uint8 [H, W, 3] (numpy or torch, host or device) + integer rectangles (x1, y1, x2, y2) (parts outside the
    frame read as zeros, as Image.crop pads) -> uint8 [n, size, size, 3] RGB on the device, equal to
    ``Image.fromarray(rgb).crop(r).resize((size, size), BICUBIC)`` for every rectangle; with `center_crop` to the
    aspect-preserving form ``resize(shorter side -> size)`` + centre crop (`shorter_side_geometry`): only the kept
    size x size window of the resized crop is computed - its rows of the two filter tables.
    tables: "device" (default) - the filter tables are filled by `cap_crop_resize_tables` (the host only sends rectangles);
    "host" - built here with `pil_bicubic_coeffs` and uploaded (same bits; kept as the cross-check)."""
    if not torch.cuda.is_available():
        raise N.CaptionerHipError("crop_resize_u8 needs a GPU; there is no CPU fallback in the product path")
    lib = N.load_library()
    dev = torch.device(device)
    if isinstance(frame, np.ndarray):
        frame = torch.from_numpy(np.ascontiguousarray(frame))
    if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
        raise ValueError(f"frame must be uint8 [H, W, 3], got {frame.dtype} {tuple(frame.shape)}")
    H, W = int(frame.shape[0]), int(frame.shape[1])
    rects = np.asarray(rects, dtype=np.int64).reshape(-1, 4)
    n = rects.shape[0]
    if n == 0:
        return torch.empty((0, size, size, 3), dtype=torch.uint8, device=dev)
    x1, y1, x2, y2 = rects.T
    if (x2 <= x1).any() or (y2 <= y1).any() or (np.abs(rects) > 1 << 20).any():
        raise ValueError(f"empty or absurd crop rectangle: {rects.tolist()}")
    # a rectangle may leave the frame: Image.crop pads with zeros and so does the kernel (the reference's expand_box clamps
    # x to the frame height and y to its width, so this happens on non-square frames)
    ws, hs = x2 - x1, y2 - y1
    if center_crop:
        geom = np.array([shorter_side_geometry(int(w), int(h), size) for w, h in zip(ws.tolist(), hs.tolist())], dtype=np.int64)
    else:
        geom = np.tile(np.array([size, size, 0, 0], dtype=np.int64), (n, 1))
    KH, KV = _ksize(ws, geom[:, 0]), _ksize(hs, geom[:, 1])
    stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)   # noqa: E731
    with torch.cuda.device(dev):
        frame_d = frame.to(dev, non_blocking=True).contiguous()
        out = torch.empty((n, size, size, 3), dtype=torch.uint8, device=dev)
        sizes = (n * 4, n * 4, n * size * 2, n * size * KH, n * size * 2, n * size * KV)   # rects, geom, hb, hk, vb, vk
        offs = np.concatenate([[0], np.cumsum(sizes)])
        if tables == "device":
            head = torch.from_numpy(np.concatenate([rects.ravel(), geom.ravel()]).astype(np.int32))
            ws_t = torch.empty(int(offs[-1]), dtype=torch.int32, device=dev)
            ws_t[: 8 * n].copy_(head, non_blocking=True)
        elif tables == "host":
            blob = np.zeros(int(offs[-1]), dtype=np.int32)
            blob[: 4 * n] = rects.ravel(); blob[4 * n: 8 * n] = geom.ravel()
            hb = blob[offs[2]:offs[3]].reshape(n, size, 2); hk = blob[offs[3]:offs[4]].reshape(n, size, KH)
            vb = blob[offs[4]:offs[5]].reshape(n, size, 2); vk = blob[offs[5]:offs[6]].reshape(n, size, KV)
            for i in range(n):
                nw, nh, left, top = (int(v) for v in geom[i])
                bh, kh = pil_bicubic_coeffs(int(ws[i]), nw)
                bv, kv = pil_bicubic_coeffs(int(hs[i]), nh)
                hb[i] = bh[left:left + size]; hk[i, :, :kh.shape[1]] = kh[left:left + size]
                vb[i] = bv[top:top + size]; vk[i, :, :kv.shape[1]] = kv[top:top + size]
            ws_t = torch.from_numpy(blob).to(dev, non_blocking=True)
        else:
            raise ValueError(f"tables must be 'device' or 'host', got {tables!r}")
        p = [C.c_void_p(ws_t.data_ptr() + 4 * int(o)) for o in offs[:-1]]
        if tables == "device":
            N.check(lib.cap_crop_resize_tables(p[0], p[1], n, size, KH, KV, p[2], p[3], p[4], p[5], stream()),
                    "cap_crop_resize_tables")
        N.check(lib.cap_crop_resize_u8(C.c_void_p(frame_d.data_ptr()), H, W, int(bool(bgr)), p[0], p[2], p[3], KH, p[4], p[5],
                                       KV, n, size, C.c_void_p(out.data_ptr()), stream()), "cap_crop_resize_u8")
        cur = torch.cuda.current_stream(dev)
        for t in (out, ws_t, frame_d):
            t.record_stream(cur)
    return out


def crop_resize_u8_frames(frames: Sequence[np.ndarray], rects_per_frame: Sequence[Sequence[Sequence[int]]], size: int, bgr: bool = False,
                          device: str | torch.device = "cuda:0", center_crop: bool = False) -> torch.Tensor:
    """Boxes of SEVERAL frames (uint8 [H_f, W_f, 3] host arrays, any sizes) in one go: -> uint8 [n_boxes, size, size, 3] RGB on the
    device, frame-major in box order, every box equal to ``Image.fromarray(rgb).crop(r).resize((size, size), BICUBIC)`` (or the
    shorter-side + centre-crop form).  Only each box's in-frame pixels travel: they are copied into ONE pinned buffer (a box that
    leaves its frame keeps its offset inside the copied patch, the kernel pads with zeros as `Image.crop` does), uploaded once, and
    the whole set is two launches (`cap_crop_resize_u8_frames`).  A 1280 x 1280 frame with three boxes sends ~0.5 MB instead of 4.9."""
    if not torch.cuda.is_available():
        raise N.CaptionerHipError("crop_resize_u8_frames needs a GPU; there is no CPU fallback in the product path")
    lib = N.load_library()
    dev = torch.device(device)
    patches, rects, hw = [], [], []
    for fi, (fr, rs) in enumerate(zip(frames, rects_per_frame)):
        if fr.dtype != np.uint8 or fr.ndim != 3 or fr.shape[2] != 3:
            raise ValueError(f"frame {fi} must be uint8 [H, W, 3], got {fr.dtype} {fr.shape}")
        H, W = int(fr.shape[0]), int(fr.shape[1])
        for r in np.asarray(rs, dtype=np.int64).reshape(-1, 4).tolist():
            x1, y1, x2, y2 = r
            if x2 <= x1 or y2 <= y1 or max(abs(v) for v in r) > 1 << 20:
                raise ValueError(f"empty or absurd crop rectangle: {r}")
            sx1, sy1, sx2, sy2 = min(max(x1, 0), W), min(max(y1, 0), H), min(max(x2, 0), W), min(max(y2, 0), H)
            if sx2 <= sx1 or sy2 <= sy1:                 # wholly outside the frame: one zero pixel stands for the padding
                patches.append(np.zeros((1, 1, 3), dtype=np.uint8))
                rects.append((4, 4, 4 + x2 - x1, 4 + y2 - y1))          # a rectangle of the box's size beside the pixel: all zeros
            else:
                patches.append(fr[sy1:sy2, sx1:sx2])
                rects.append((x1 - sx1, y1 - sy1, x2 - sx1, y2 - sy1))
            hw.append(patches[-1].shape[:2])
    n = len(patches)
    if n == 0:
        return torch.empty((0, size, size, 3), dtype=torch.uint8, device=dev)
    groups = _byte_groups(hw)
    outs = [_resize_patches(patches[i:j], rects[i:j], hw[i:j], size, bgr, dev, center_crop, lib) for i, j in groups]
    return outs[0] if len(outs) == 1 else torch.cat(outs)


_STAGE_BYTES = 256 << 20     # pinned staging per launch: a very long list goes through in groups of at most this many bytes


def _byte_groups(hw):
    """[(i, j)] consecutive patch ranges whose packed bytes stay within _STAGE_BYTES (a single larger patch is a group of its own)."""
    groups, i, acc = [], 0, 0
    for k, (h, w) in enumerate(hw):
        b = int(h) * int(w) * 3
        if k > i and acc + b > _STAGE_BYTES:
            groups.append((i, k))
            i, acc = k, 0
        acc += b
    groups.append((i, len(hw)))
    return groups


def _resize_patches(patches, rects, hw, size, bgr, dev, center_crop, lib) -> torch.Tensor:
    n = len(patches)
    hw = np.asarray(hw, dtype=np.int64)
    rects = np.asarray(rects, dtype=np.int64)
    nbytes = hw[:, 0] * hw[:, 1] * 3
    offs = np.concatenate([[0], np.cumsum(nbytes)])
    packed = torch.empty(int(offs[-1]), dtype=torch.uint8, pin_memory=True)
    pk = packed.numpy()
    for i, a in enumerate(patches):
        pk[offs[i]:offs[i + 1]].reshape(a.shape)[...] = a
    ws, hs = rects[:, 2] - rects[:, 0], rects[:, 3] - rects[:, 1]
    if center_crop:
        geom = np.array([shorter_side_geometry(int(w), int(h), size) for w, h in zip(ws.tolist(), hs.tolist())], dtype=np.int64)
    else:
        geom = np.tile(np.array([size, size, 0, 0], dtype=np.int64), (n, 1))
    KH, KV = _ksize(ws, geom[:, 0]), _ksize(hs, geom[:, 1])
    ftab = np.stack([offs[:-1], hw[:, 0], hw[:, 1]], axis=1).astype(np.int64)
    stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)   # noqa: E731
    with torch.cuda.device(dev):
        packed_d = packed.to(dev, non_blocking=True)
        ftab_d = torch.from_numpy(ftab).to(dev, non_blocking=True)
        out = torch.empty((n, size, size, 3), dtype=torch.uint8, device=dev)
        sizes = (n * 4, n * 4, n * size * 2, n * size * KH, n * size * 2, n * size * KV)   # rects, geom, hb, hk, vb, vk
        o = np.concatenate([[0], np.cumsum(sizes)])
        head = torch.from_numpy(np.concatenate([rects.ravel(), geom.ravel()]).astype(np.int32))
        ws_t = torch.empty(int(o[-1]), dtype=torch.int32, device=dev)
        ws_t[: 8 * n].copy_(head, non_blocking=True)
        p = [C.c_void_p(ws_t.data_ptr() + 4 * int(x)) for x in o[:-1]]
        N.check(lib.cap_crop_resize_tables(p[0], p[1], n, size, KH, KV, p[2], p[3], p[4], p[5], stream()), "cap_crop_resize_tables")
        N.check(lib.cap_crop_resize_u8_frames(C.c_void_p(packed_d.data_ptr()), C.c_void_p(ftab_d.data_ptr()), int(bool(bgr)), p[0], p[2], p[3], KH,
                                              p[4], p[5], KV, n, size, C.c_void_p(out.data_ptr()), stream()), "cap_crop_resize_u8_frames")
        cur = torch.cuda.current_stream(dev)
        for t in (out, ws_t, packed_d, ftab_d):
            t.record_stream(cur)
        cur.synchronize()                                 # the pinned staging buffer is released when this returns
    return out


def resize_u8_list(images: Sequence[np.ndarray], size: int, device: str | torch.device = "cuda:0", center_crop: bool = False) -> torch.Tensor:
    """A list of uint8 RGB images [H_i, W_i, 3] of different sizes (the PIL crops `generate_batch` / `caption_batch` receive) ->
    uint8 [n, size, size, 3] on the device, every image equal to ``Image.fromarray(a).resize((size, size), BICUBIC)`` (or, with
    `center_crop`, to the shorter-side resize + centre crop of `shorter_side_geometry`): `crop_resize_u8_frames` with each image as
    its own frame and its whole extent as the box - one packed upload, two launches."""
    for i, a in enumerate(images):
        if a.ndim != 3 or a.shape[0] < 1 or a.shape[1] < 1:
            raise ValueError(f"image {i} must be uint8 [H, W, 3], got {a.dtype} {a.shape}")
    return crop_resize_u8_frames(images, [[(0, 0, a.shape[1], a.shape[0])] for a in images], size, device=device, center_crop=center_crop)
