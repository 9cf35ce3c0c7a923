"""CPU oracle: semantic-segmentation stitching and patch grids.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates
``tiatoolbox/models/engine/semantic_segmentor.py:1141-1263,1398-1534`` (canvas merge) and
``tiatoolbox/tools/patchextraction.py:356-461,487-613``.
"""

from __future__ import annotations

import numpy as np


def merge_batch_to_canvas(blocks: np.ndarray, output_locations: np.ndarray, merged_shape: tuple[int, int, int]):
    """``semantic_segmentor.py:1141-1183``: add blocks into a row canvas; all-zero blocks are skipped;
    ``count`` is uint8."""
    blocks = np.asarray(blocks)
    output_locations = np.asarray(output_locations)
    canvas = np.zeros(merged_shape, dtype=blocks.dtype)
    count = np.zeros((*merged_shape[:2], 1), dtype=np.uint8)
    for i, block in enumerate(blocks):
        xs, ys, xe, ye = (int(v) for v in output_locations[i])
        if not np.any(block):
            continue
        canvas[0:ye - ys, xs:xe, :] += block[0:ye - ys, 0:xe - xs, :]
        count[0:ye - ys, xs:xe, 0] += 1
    return canvas, count


def merge_wsi(blocks: np.ndarray, output_locations: np.ndarray, out_shape: tuple[int, int]) -> np.ndarray:
    """Whole-slide probabilities as the reference computes them: per patch-row horizontal merge
    (``merge_horizontal`` :1186-1263, width clipped to the canvas), consecutive rows added over their
    overlap and divided by ``max(count, 1)`` (``merge_vertical_chunkwise`` :1398-1534), clipped to
    ``out_shape`` = (H, W)."""
    h, w = out_shape
    locs = np.asarray(output_locations).copy()
    c = blocks.shape[-1]
    probs = np.zeros((h, w, c), dtype=np.float32)
    row_ys = np.unique(locs[:, 1])
    rows = []
    for y0 in row_ys:
        sel = np.flatnonzero(locs[:, 1] == y0)
        rl = locs[sel].copy()
        rl[:, 2] = np.minimum(rl[:, 2], w)
        canvas, count = merge_batch_to_canvas(blocks[sel], rl, (blocks.shape[1], w, c))
        rows.append((int(y0), canvas, count))
    y1s = np.array([r[0] + blocks.shape[1] for r in rows])
    y0s = np.array([r[0] for r in rows])
    overlaps = np.append(y1s[:-1] - y0s[1:], 0)
    curr, curr_cnt = rows[0][1].copy(), rows[0][2].copy()
    written = 0
    for i, overlap in enumerate(overlaps):
        nxt = rows[i + 1] if i + 1 < len(rows) else None
        if nxt is not None and overlap > 0:
            curr[-overlap:] += nxt[1][:overlap]
            curr_cnt[-overlap:] += nxt[2][:overlap]
        cnt = np.where(curr_cnt == 0, 1, curr_cnt)
        p = curr / cnt.astype(np.float32)
        take = min(p.shape[0], h - written)
        if take <= 0:
            break
        probs[written:written + take] = p[:take]
        written += take
        if nxt is not None:
            curr, curr_cnt = nxt[1][overlap:].copy(), nxt[2][overlap:].copy()
    return probs
