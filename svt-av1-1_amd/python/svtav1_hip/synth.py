"""Synthetic pictures and picture-analysis planes for tests and bench (host-side numpy).

Mirrors what the reference's Picture Analysis process hands to ME
(all paths under /root/reference/Source/Lib/Codec):
  * padded full-resolution luma, 68 px of edge replication per side
    (EbEncHandle.c:1006-1009, generate_padding EbMcp.c:173-215);
  * "quarter" plane  = every 2nd pixel / row, padded 32 px;
  * "sixteenth" plane = every 4th pixel / row, padded 16 px
    (Decimation2D EbPictureAnalysisProcess.c:99-124, DecimateInputPicture :4882-4936,
    paddings EbEncHandle.c:1013-1030).
The frame formula is SURVEY.md section 8(d)'s deterministic synthetic input.
"""
from __future__ import annotations

import numpy as np

SEED = 20261004
PAD_FULL = 68
PAD_QUARTER = 32
PAD_SIXTEENTH = 16


def synth_luma(width: int, height: int, t: int = 0, seed: int = SEED) -> np.ndarray:
    """Frame t of the synthetic sequence: smooth field + quadratic texture + LCG noise."""
    x = np.arange(width, dtype=np.int64)[None, :] + 3 * t
    y = np.arange(height, dtype=np.int64)[:, None] + 2 * t
    smooth = 48.0 * np.sin(2 * np.pi * x / 97.0) * np.cos(2 * np.pi * y / 61.0)
    quad = ((x * x + 3 * y * y) >> 9) & 63
    xx = np.arange(width, dtype=np.int64)[None, :]
    yy = np.arange(height, dtype=np.int64)[:, None]
    lcg = (1103515245 * (xx * 7919 + yy * 104729 + seed + 977 * t) + 12345) & 0xFFFFFFFF
    noise = (lcg >> 16) & 31
    v = np.floor(96.0 + smooth).astype(np.int64) + quad + noise - 16
    return np.clip(v, 0, 255).astype(np.uint8)


def pad_plane(img: np.ndarray, pad: int) -> np.ndarray:
    """generate_padding(): horizontal then vertical edge replication == numpy 'edge' mode."""
    return np.ascontiguousarray(np.pad(img, pad, mode="edge"))


def decimate(img: np.ndarray, step: int) -> np.ndarray:
    """Decimation2D(): point-sample every `step`-th pixel and row (no filtering)."""
    return np.ascontiguousarray(img[::step, ::step])


class PaPicture:
    """The three padded luma planes of one EbPaReferenceObject_t, as numpy arrays."""

    def __init__(self, luma: np.ndarray):
        h, w = luma.shape
        assert w % 8 == 0 and h % 8 == 0, "reference rounds picture dims up to multiples of 8"
        self.width, self.height = w, h
        self.full = pad_plane(luma, PAD_FULL)
        self.quarter = pad_plane(decimate(luma, 2)[: h >> 1, : w >> 1], PAD_QUARTER)
        self.sixteenth = pad_plane(decimate(luma, 4)[: h >> 2, : w >> 2], PAD_SIXTEENTH)

    @property
    def stride(self) -> int:
        return self.full.shape[1]

    def sb_grid(self):
        return (self.width + 63) // 64, (self.height + 63) // 64


def clamp_search_window(origin_x, origin_y, x_center, y_center, sw, sh, pic_w, pic_h):
    """Search-window clipping of MotionEstimateLcu (EbMotionEstimation.c:6667-6723).

    Returns (x_search_area_origin, y_search_area_origin, search_area_width, search_area_height).
    """
    pad_w = pad_h = 63
    sw = min(sw, 127)
    sh = min(sh, 127)
    xo = x_center - (sw >> 1)
    yo = y_center - (sh >> 1)
    # The reference evaluates four statements per axis in sequence, each re-reading the origin the
    # previous one just wrote (:6690-6705).  Statement 2 (shrink the width at the left edge) tests the
    # already-corrected origin, so it never fires: the left/top clamp moves the window without
    # shrinking it.  Reproduced literally.
    if origin_x + xo < -pad_w:
        xo = -pad_w - origin_x
    if origin_x + xo < -pad_w:  # dead after the line above, kept for fidelity
        sw = sw - (-pad_w - (origin_x + xo))
    if origin_x + xo > pic_w - 1:
        xo = xo - ((origin_x + xo) - (pic_w - 1))
    if origin_x + xo + sw > pic_w:
        sw = max(1, sw - ((origin_x + xo + sw) - pic_w))
    if origin_y + yo < -pad_h:
        yo = -pad_h - origin_y
    if origin_y + yo < -pad_h:
        sh = sh - (-pad_h - (origin_y + yo))
    if origin_y + yo > pic_h - 1:
        yo = yo - ((origin_y + yo) - (pic_h - 1))
    if origin_y + yo + sh > pic_h:
        sh = max(1, sh - ((origin_y + yo + sh) - pic_h))
    return xo, yo, sw, sh
