"""Axis-aligned stand-ins for the two shapely facilities the reference's tile merge uses.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  shapely (GEOS) is not installed here; the reference
(``tiatoolbox/models/engine/multi_task_segmentor.py:1431-1447, 2889-2893, 2971-3016, 3033-3038, 3287-3294``)
only ever builds ``shapely.box`` rectangles, queries an ``STRtree`` of rectangles with a rectangle and asks
``box.contains(box)``.  For rectangles these have closed-form definitions (documented shapely 2.x semantics):

* ``STRtree.query(geom)`` (no predicate): indices of the tree geometries whose *extent* intersects the
  extent of ``geom`` -- closed intervals, so touching rectangles count;
* ``a.contains(b)``: no point of ``b`` outside ``a`` and the interiors meet; a degenerate ``a`` (zero
  width or height) has an empty interior and contains nothing with positive area.

**Parity with the GEOS implementation is unpinned.**
"""

from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, xmin, ymin, xmax, ymax) -> None:
        self.bounds = (float(xmin), float(ymin), float(xmax), float(ymax))

    def contains(self, other: "Box") -> bool:
        a, b = self.bounds, other.bounds
        if not (a[2] > a[0] and a[3] > a[1]):
            return False
        return a[0] <= b[0] and a[1] <= b[1] and a[2] >= b[2] and a[3] >= b[3]


def box(xmin, ymin, xmax, ymax) -> Box:
    return Box(xmin, ymin, xmax, ymax)


class STRtree:
    def __init__(self, geoms) -> None:
        self._b = np.array([g.bounds for g in geoms], dtype=np.float64).reshape(-1, 4)

    def query(self, geom: Box) -> np.ndarray:
        q = geom.bounds
        b = self._b
        return np.flatnonzero((b[:, 0] <= q[2]) & (b[:, 2] >= q[0]) & (b[:, 1] <= q[3]) & (b[:, 3] >= q[1]))
