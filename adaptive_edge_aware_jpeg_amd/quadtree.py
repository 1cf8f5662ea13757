"""``QuadTree`` / ``QuadNode`` (src/jpeg/quadtree.py:41-165) backed by the HIP quadtree kernels (``aej_quadtree``).

The tree itself is never materialised on the device: the kernels emit the pre-order state symbols and the
leaf table directly (csrc/quadtree.hip).  ``root`` / ``children`` are rebuilt lazily from the symbols when a
caller walks the tree.
"""
import ctypes
from typing import List, Optional, Tuple

import numpy as np

from . import tables
from ._lib import get_context


class QuadNode:
    """Represents a single node in the QuadTree."""

    def __init__(self, x: int, y: int, size: int) -> None:
        self.x = x
        self.y = y
        self.size = size
        self.children: List[Optional["QuadNode"]] = []

    def is_leaf(self) -> bool:
        return len(self.children) == 0


class QuadTree:
    """QuadTree that partitions an edge-detected image adaptively."""

    def __init__(self, edge_image: np.ndarray, max_size: int = 64, min_size: int = 4) -> None:
        if not isinstance(edge_image, np.ndarray):
            raise TypeError("Input must be a numpy array.")
        if edge_image.ndim != 2:
            raise ValueError("Input array must be a 2D with a single channel.")
        self.image = edge_image
        self.max_size = max_size
        self.min_size = min_size
        self.root_size = tables.largest_power_of_2(max(edge_image.shape)) * 2
        self._leaves, self._states = self._run()
        self._root = None

    def _run(self) -> Tuple[np.ndarray, np.ndarray]:
        ctx = get_context()
        t = ctx.torch
        H, W = self.image.shape
        edge = ctx.to_device((self.image == 1.0).astype(np.uint8), t.uint8)     # quadtree.py:38
        lc, sc, cc = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        rc = ctx.lib.aej_quadtree_capacity(H, W, self.min_size, self.max_size, ctypes.byref(lc), ctypes.byref(sc), ctypes.byref(cc))
        if rc != 0:
            raise NotImplementedError(f"quadtree geometry not supported: H={H} W={W} min={self.min_size} max={self.max_size}")
        leaves = ctx.empty((max(lc.value, 1), 4), t.int32)
        states = ctx.empty((max(sc.value, 1),), t.uint8)
        counts = ctx.empty((4,), t.int64)
        nbytes = ctx.lib.aej_quadtree_workspace_bytes(H, W, self.min_size, self.max_size)
        ws = ctx.workspace(nbytes)
        ctx.check(ctx.lib.aej_quadtree(ctx.handle, edge.data_ptr(), H, W, self.min_size, self.max_size, leaves.data_ptr(),
                                       states.data_ptr(), counts.data_ptr(), ws.data_ptr(), ctypes.c_uint64(nbytes)))
        c = counts.cpu().numpy()
        assert int(c[3]) == self.root_size
        return leaves[: int(c[1])].cpu().numpy(), states[: int(c[2])].cpu().numpy()

    @property
    def root(self) -> QuadNode:
        if self._root is None:
            self._root = self._rebuild()
        return self._root

    def _rebuild(self) -> QuadNode:
        # replay the pre-order symbols (the inverse of get_leaves_and_states, cf. jpeg.py:768-800)
        states = self._states.tolist()
        holder = QuadNode(-1, -1, -1)
        holder.children = [None]
        stack = [(0, 0, self.root_size, holder, 0)]
        pos = 0
        while stack:
            x, y, size, parent, idx = stack.pop()
            state = states[pos]
            pos += 1
            if state == 2:
                continue
            node = QuadNode(x, y, size)
            parent.children[idx] = node
            if state == 1:
                h = size // 2
                node.children = [None, None, None, None]
                stack.append((x + h, y + h, h, node, 3))
                stack.append((x, y + h, h, node, 2))
                stack.append((x + h, y, h, node, 1))
                stack.append((x, y, h, node, 0))
        return holder.children[0]

    def get_leaves_and_states(self) -> Tuple[List[QuadNode], List[str]]:
        names = ("00", "01", "10")
        leaves = [QuadNode(int(x), int(y), int(s)) for x, y, s, _ in self._leaves]
        return leaves, [names[v] for v in self._states.tolist()]
