"""Reference-STRUCTURED CPU restatement of ``Jpeg.compress`` up to the zigzag gather.
TEST INFRASTRUCTURE ONLY (CPU baseline of bench.py and a parity check of the oracle's orchestration) -- never imported by
the product.

Where oracle.encode_image() runs each stage as one C call per layer, this module keeps the reference's own control
structure, because that is where the reference spends its time:

* one Python thread, per-layer stages (src/jpeg/jpeg.py:240-272);
* the quadtree as an explicit-stack top-down split with ONE region test per node (quadtree.py:93-134; the reference calls a
  numba `np.any(region == 1.0)` per node) and a pre-order walk for leaves / states (quadtree.py:136-165);
* PER-LEAF Python loops for the gather + ``np.pad(mode='reflect')`` (jpeg.py:393-404), the DCT (one ``cv.dct`` call per leaf,
  jpeg.py:471), the quantisation ``np.round(block / q).astype(int32)`` (jpeg.py:499-502) and the zigzag gather
  ``block.ravel()[idx]`` + one concatenate (jpeg.py:581-588).

The third-party calls the reference makes (numba colour kernels, ``cv.resize``, ``EdgeDetection.canny``'s six OpenCV calls,
``cv.dct``) are served by the C oracle, one call each, exactly where the reference calls the native library -- OpenCV and
numba are absent on both boxes.  Output is identical to oracle.encode_image() (tests/test_oracle_pins.py).
"""
import ctypes

import numpy as np

from . import oracle as O


class _Node:
    __slots__ = ("x", "y", "size", "children", "is_leaf")

    def __init__(self, x, y, size):
        self.x, self.y, self.size = x, y, size
        self.children = [None, None, None, None]
        self.is_leaf = True


def _build_tree(edge, max_size, min_size):
    """quadtree.py:85-134: root = largest_power_of_2(max(H, W)) * 2; pop, skip nodes outside the image, split when
    size > max or (size > min and the region holds an edge pixel); children pushed so that they pop TL, TR, BL, BR."""
    H, W = edge.shape
    root = _Node(0, 0, O.root_size(H, W))
    stack = [(0, 0, root.size, None, 0)]
    while stack:
        x, y, size, parent, idx = stack.pop()
        if x >= W or y >= H:
            continue
        node = root if parent is None else _Node(x, y, size)
        if parent is not None:
            parent.children[idx] = node
        if size > max_size or (size > min_size and bool(np.any(edge[y:y + size, x:x + size] == 1.0))):
            node.is_leaf = False
            h = size // 2
            stack.append((x + h, y + h, h, node, 3))
            stack.append((x, y + h, h, node, 2))
            stack.append((x + h, y, h, node, 1))
            stack.append((x, y, h, node, 0))
    return root


def _leaves_and_states(root):
    """quadtree.py:136-165: pre-order walk; absent child -> 2 ('10'), leaf -> 0 ('00'), internal -> 1 ('01')."""
    leaves, states = [], []
    stack = [root]
    while stack:
        node = stack.pop()
        if node is None:
            states.append(2)
            continue
        if node.is_leaf:
            states.append(0)
            leaves.append(node)
        else:
            states.append(1)
            stack.extend(reversed(node.children))
    return leaves, states


def _dct(block, D):
    """one native call per leaf, as ``cv.dct(block)`` (jpeg.py:471)"""
    s = block.shape[0]
    out = np.empty((s, s), np.float32)
    O.lib().orc_dct_block(D.ctypes.data_as(ctypes.c_void_p), block.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), s)
    return out


def encode_image(rgb, space="YCoCg", qrange=(40, 80), brange=(4, 64)):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    H, W, _ = rgb.shape
    sizes, zz, qm = O.tables(space, qrange, brange)
    Dm = {s: O.dct_matrix(s) for s in sizes}
    conv = O.color_forward(space, rgb.reshape(-1, 3)).reshape(H, W, 3)            # jpeg.py:262, color.convert
    planes = [O.downsample(conv, i, rh, rw) for i, (rh, rw) in enumerate(O.RATIOS[space])]   # jpeg.py:267, cv.resize per layer
    blocks_per_layer, states_per_layer, roots, leaves_per_layer = [], [], [], []
    for i, layer in enumerate(planes):                                             # jpeg.py:356-408 _block_split
        edge = O.edge_pipeline(layer).astype(np.float32)                           # EdgeDetection.canny -> float32 {0, 1}
        root = _build_tree(edge, brange[1], brange[0])
        leaves, states = _leaves_and_states(root)
        roots.append(root.size)
        norm = O.normalize(layer, space, i)                                        # apply_normalization, jpeg.py:387-390
        blocks = []
        for leaf in leaves:                                                        # jpeg.py:393-404
            x, y, s = leaf.x, leaf.y, leaf.size
            block = norm[y:y + s, x:x + s]
            if block.shape != (s, s):
                block = np.pad(block, ((0, s - block.shape[0]), (0, s - block.shape[1])), mode="reflect")
            blocks.append(np.ascontiguousarray(block))
        blocks_per_layer.append(blocks)
        states_per_layer.append(states)
        leaves_per_layer.append(leaves)
    dct = [[_dct(b, Dm[b.shape[0]]) for b in blocks] for blocks in blocks_per_layer]           # jpeg.py:461-471
    quant = [[np.round(b / qm[i][b.shape[0]]).astype(np.int32) for b in blocks] for i, blocks in enumerate(dct)]   # jpeg.py:485-506
    out = []
    for i, blocks in enumerate(quant):                                             # jpeg.py:579-588
        coeffs = np.concatenate([b.ravel()[zz[b.shape[0]]] for b in blocks]) if blocks else np.zeros(0, np.int32)
        out.append({"root_size": roots[i], "states": np.asarray(states_per_layer[i], np.uint8),
                    "leaves": np.array([[n.x, n.y, n.size] for n in leaves_per_layer[i]], np.int32).reshape(-1, 3), "coeffs": coeffs})
    return out


def fan_out(images, space, qrange, brange, workers):
    """one image per worker PROCESS over the host cores, as the reference's sweep does
    (test/analysis/metrics_computation.py:253, a process pool with one ``Jpeg`` per worker).  Fresh interpreters (no fork of
    the calling process, which may hold a GPU context); every worker loads its image from a shared .npy, encodes it with
    encode_image() above and exits.  Returns the wall seconds from the first spawn to the last exit."""
    import os, subprocess, sys, tempfile, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "imgs.npy")
        np.save(path, np.ascontiguousarray(images, dtype=np.float32))
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from oracle import reference_structured as RS; "
                "im = np.load(%r, mmap_mode='r')[int(sys.argv[1])]; RS.encode_image(np.array(im), %r, %r, %r)" % (root, path, space, tuple(qrange), tuple(brange)))
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-c", code, str(i)], env=dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1"))
                 for i in range(min(workers, len(images)))]
        rcs = [p.wait() for p in procs]
        dt = time.perf_counter() - t0
    if any(rcs):
        raise RuntimeError(f"reference-structured worker failed: exit codes {rcs}")
    return dt, len(procs)
