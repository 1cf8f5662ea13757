"""``Jpeg`` -- the reference's codec object (src/jpeg/jpeg.py:177-800) with the encode hot path on the GPU.

``compress(img)`` / ``decompress(bytes)`` keep the reference signatures (Image <-> .ajpg bytes).  Stages a-1 ... a-15 of SURVEY.md
section 8 (colour convert, down-sample, Canny, quadtree, normalise, DCT, quantise, zigzag) run as HIP kernels
behind ``aej_encode_batch``; the ``.ajpg`` container (JSON header, 2-bit state packing, per-layer zlib-9;
jpeg.py:531-597) is written on the host from the kernel outputs.  ``compress_batch`` is the throughput entry:
a device-resident float32 [B, H, W, 3] batch in, device-resident coefficient / leaf / state arrays out.
"""
import ctypes
import json
import os
import zlib
from io import BytesIO
from typing import List, Optional, Tuple

import numpy as np

from . import tables
from ._lib import AejError, get_context, warn_if_queue_limited
from .image import Image
from .settings import JpegCompressionSettings


def usable_cpus() -> int:
    """Cores this process may run on (its affinity mask; a container's share is usually far below ``os.cpu_count()``)."""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


class EncodedBatch:
    """Device-resident result of the encode hot path for a batch (layout: include/aej.h, aej_plan)."""

    def __init__(self, plan, coeffs, leaves, states, counts, dct=None):
        self.plan, self.coeffs, self.leaves, self.states, self.counts, self.dct = plan, coeffs, leaves, states, counts, dct
        self._counts_host = None

    @property
    def counts_host(self) -> np.ndarray:
        if self._counts_host is None:
            self._counts_host = self.counts.cpu().numpy()
        return self._counts_host

    def layer(self, b: int, l: int, want_dct: bool = False):
        """-> dict(root_size, states uint8[n], leaves int32[n,3] (x,y,size), coeffs int32[sum s*s])"""
        p = self.plan
        n_coef, n_leaf, n_state, root = (int(v) for v in self.counts_host[b, l])
        co = b * p.coeff_stride + p.coeff_off[l]
        lo = b * p.leaf_stride + p.leaf_off[l]
        so = b * p.state_stride + p.state_off[l]
        out = {
            "root_size": root,
            "coeffs": self.coeffs[co:co + n_coef].cpu().numpy(),
            "leaves": self.leaves[lo:lo + n_leaf, :3].cpu().numpy(),
            "leaf_coeff_offsets": self.leaves[lo:lo + n_leaf, 3].cpu().numpy(),
            "states": self.states[so:so + n_state].cpu().numpy(),
        }
        if want_dct and self.dct is not None:
            out["dct"] = self.dct[co:co + n_coef].cpu().numpy()
        return out


class Jpeg:
    """JPEG compression and decompression implementation with adaptive blocking."""

    def __init__(self, settings: JpegCompressionSettings, device: int = 0) -> None:
        self._device = device
        self.update_settings(settings)

    # ------------------------------------------------------------------ settings / caches (jpeg.py:189-238)
    def update_settings(self, settings: JpegCompressionSettings, layer_shape: Optional[Tuple[int, int]] = None) -> None:
        self.settings = settings
        if layer_shape is not None:
            self.update_layer_shapes(layer_shape)
        self.precompute_caches()

    def update_layer_shapes(self, layer_shape: Tuple[int, int]) -> None:
        self.layer_shape = layer_shape
        self.layer_shapes = self._compute_downsampled_shapes(self.layer_shape)

    def precompute_caches(self) -> None:
        sizes = tables.block_sizes(self.settings.block_size_range)
        if not hasattr(self, "zigzag_cache"):
            self.zigzag_cache = {}
        for size in sizes:
            if size not in self.zigzag_cache:
                self.zigzag_cache[size] = Jpeg._zigzag_ordering(size)
        self.quantization_matrix_cache = {}
        for i, qm in enumerate(self.settings.quantization_matrices):
            self.quantization_matrix_cache[i] = {
                size: Jpeg._get_quantization_matrix(qm, size, self._get_quality_factor(size)) for size in sizes}
        self._block_sizes = sizes

    def _qmats_blob(self) -> np.ndarray:
        """[layer][size][s*s] int32, the layout aej_set_settings expects."""
        return np.concatenate([self.quantization_matrix_cache[l][s].ravel() for l in range(3) for s in self._block_sizes]).astype(np.int32)

    def _bind(self):
        ctx = get_context(self._device)
        bmin, bmax = self._block_sizes[0], self._block_sizes[-1]
        ctx.set_settings(self.settings.color_space, bmin, bmax, self._qmats_blob())
        return ctx

    # ------------------------------------------------------------------ encode
    def compress_batch(self, batch, want_dct: bool = False) -> EncodedBatch:
        """batch: float32 [B, H, W, 3] in [0, 1], or uint8 [B, H, W, 3] (8-bit ingest: the GPU forms float32(v) / 255 as
        Image.load does, image.py:80, at a quarter of the input traffic) -- a torch tensor already on the GPU (throughput
        path) or a numpy array (copied).  Returns device-resident outputs; nothing is copied back."""
        ctx = self._bind()
        t = ctx.torch
        is_u8 = str(getattr(batch, "dtype", "")) in ("uint8", "torch.uint8")
        x = ctx.to_device(batch, t.uint8 if is_u8 else t.float32)
        if x.ndim != 4 or x.shape[3] != 3:
            raise ValueError("Input batch must be [B, H, W, 3].")
        B, H, W, _ = x.shape
        plan = ctx.plan(B, H, W)
        warn_if_queue_limited(ctx, B, H, W)
        coeffs = ctx.empty((B * plan.coeff_stride,), t.int32)
        leaves = ctx.empty((B * plan.leaf_stride, 4), t.int32)
        states = ctx.empty((B * plan.state_stride,), t.uint8)
        counts = ctx.empty((B, 3, 4), t.int64)
        dct = ctx.empty((B * plan.coeff_stride,), t.float32) if want_dct else None
        self.encode_into(ctx, x, plan, coeffs, leaves, states, counts, dct)
        return EncodedBatch(plan, coeffs, leaves, states, counts, dct)

    def compress_batches(self, batches, in_flight: int = 4, inputs_ready: bool = False):
        """``compress_batch`` for a SEQUENCE of batches at the throughput the schedule allows: a generator that keeps up to ``in_flight`` calls
        running, each on a private stream with a context, outputs and workspace of its own (``aej_encode_batch_begin`` / ``_end``), and yields
        one ``EncodedBatch`` per input batch, in order.  With four 64 x 4K-sized calls in flight each is cut into two sub-batches -- what
        bench.py times; one blocking ``compress_batch`` after another is 10-15 % slower.  Inputs may be device tensors
        (produced on the caller's current stream: every call waits for what that stream has been given so far, unless ``inputs_ready`` says the
        tensors are complete already) or numpy arrays; a result is complete when it is yielded."""
        if in_flight < 1:
            raise ValueError("in_flight must be at least 1.")
        t = get_context(self._device).torch
        dev = t.device("cuda", self._device)
        caller = t.cuda.current_stream(dev)
        # the private streams live as long as this codec: their contexts (get_context keys them by stream) and workspaces are reused by later calls
        pool = self.__dict__.setdefault("_pipe_streams", [])
        while len(pool) < in_flight:
            pool.append(t.cuda.Stream(device=dev))
        streams = pool[:in_flight]
        ctxs = [None] * in_flight
        ring = []                                              # (slot, plan, outputs) of the calls in flight, oldest first

        def finish(entry):
            slot, plan, out = entry
            with t.cuda.stream(streams[slot]):
                self.encode_end(ctxs[slot])
            for o in out:
                o.record_stream(caller)                        # allocated on the private stream, used by the caller from here on
            return EncodedBatch(plan, *out)

        try:
            for i, batch in enumerate(batches):
                slot = i % in_flight
                if len(ring) == in_flight:
                    yield finish(ring.pop(0))
                s = streams[slot]
                if not inputs_ready and isinstance(batch, t.Tensor):
                    s.wait_stream(caller)                      # the input may have been produced on the caller's stream
                with t.cuda.stream(s):
                    ctx = ctxs[slot] = self._bind()
                    is_u8 = str(getattr(batch, "dtype", "")) in ("uint8", "torch.uint8")
                    x = ctx.to_device(batch, t.uint8 if is_u8 else t.float32)
                    if x.ndim != 4 or x.shape[3] != 3:
                        raise ValueError("Input batch must be [B, H, W, 3].")
                    B, H, W, _ = x.shape
                    plan = ctx.plan(B, H, W)
                    warn_if_queue_limited(ctx, B, H, W)
                    ctx.set_sub_batches(2 if in_flight >= 4 and B * H * W >= 384_000_000 else 0)
                    out = (ctx.empty((B * plan.coeff_stride,), t.int32), ctx.empty((B * plan.leaf_stride, 4), t.int32),
                           ctx.empty((B * plan.state_stride,), t.uint8), ctx.empty((B, 3, 4), t.int64))
                    self.encode_begin(ctx, x, plan, *out)
                ring.append((slot, plan, out))
            while ring:
                yield finish(ring.pop(0))
        finally:
            for entry in ring:                                 # the consumer stopped early or a call failed: nothing may stay in flight
                try:
                    finish(entry)
                except Exception:
                    pass
            for c in ctxs:
                if c is not None:
                    c.set_sub_batches(0)

    def encode_into(self, ctx, x, plan, coeffs, leaves, states, counts, dct=None) -> None:
        """One pass of the hot path into caller-owned device buffers (what bench.py times)."""
        ws = ctx.workspace(plan.workspace_bytes)
        entry = ctx.lib.aej_encode_batch_u8 if x.dtype == ctx.torch.uint8 else ctx.lib.aej_encode_batch
        ctx.check(entry(
            ctx.handle, x.data_ptr(), plan.batch, plan.H, plan.W, coeffs.data_ptr(), leaves.data_ptr(), states.data_ptr(),
            counts.data_ptr(), dct.data_ptr() if dct is not None else None, ws.data_ptr(), ctypes.c_uint64(plan.workspace_bytes)))

    def encode_begin(self, ctx, x, plan, coeffs, leaves, states, counts, dct=None) -> None:
        """First half of encode_into (``aej_encode_batch_begin``): enqueues the call and returns without waiting.  The buffers and
        ``ctx`` belong to the call until ``encode_end(ctx)``; a second context (obtained under another ``torch.cuda.stream``)
        with its own buffers can take the next batch meanwhile."""
        ws = ctx.workspace(plan.workspace_bytes)
        ctx._in_flight = (x, coeffs, leaves, states, counts, dct, ws)          # keep the tensors alive
        ctx.check(ctx.lib.aej_encode_batch_begin(
            ctx.handle, x.data_ptr(), 1 if x.dtype == ctx.torch.uint8 else 0, plan.batch, plan.H, plan.W, coeffs.data_ptr(), leaves.data_ptr(),
            states.data_ptr(), counts.data_ptr(), dct.data_ptr() if dct is not None else None, ws.data_ptr(), ctypes.c_uint64(plan.workspace_bytes)))

    @staticmethod
    def encode_end(ctx) -> None:
        """Second half (``aej_encode_batch_end``): waits for the call in flight on ``ctx`` and verifies it."""
        try:
            ctx.check(ctx.lib.aej_encode_batch_end(ctx.handle))
        finally:
            ctx._in_flight = None

    def compress(self, img: Image, *, zlib_level: int = 9, entropy: str = "host") -> bytes:
        """Compresses the input image (jpeg.py:240-272).  The keyword-only arguments are the opt-ins of ``compress_many`` (a lower host
        deflate level, or the zlib streams written on the GPU); with the defaults the bytes are the reference's."""
        if not isinstance(img, Image):
            raise TypeError("Input must be an Image object.")
        if img.data.ndim != 3:
            raise ValueError("Input array must be a 3D.")
        self.update_layer_shapes(img.original_shape[:2])
        self.extension = img.extension
        data = np.ascontiguousarray(img.data, dtype=np.float32).reshape(img.original_shape)
        if entropy != "host" or zlib_level != 9:
            return self.compress_many(data[None], extension=img.extension, workers=3, zlib_level=zlib_level, entropy=entropy)[0]
        enc = self.compress_batch(self._host_batch_for_upload(data[None]))
        layers = [enc.layer(0, l) for l in range(3)]
        return self._entropy_encode(layers)

    # ------------------------------------------------------------------ decode (jpeg.py:274-297)
    def decompress_batch(self, enc: EncodedBatch):
        """Device round trip: decode an EncodedBatch in place -> torch float32 [B, H, W, 3] in [0, 1] on the GPU."""
        ctx = self._bind()
        t = ctx.torch
        p = enc.plan
        out = ctx.empty((p.batch, p.H, p.W, 3), t.float32)
        nbytes = ctx.lib.aej_decode_workspace_bytes(ctx.handle, p.batch, p.H, p.W)
        ws = ctx.workspace(nbytes)
        ctx.check(ctx.lib.aej_decode_batch(ctx.handle, enc.coeffs.data_ptr(), enc.leaves.data_ptr(), enc.counts.data_ptr(), p.batch,
                                           p.H, p.W, out.data_ptr(), ws.data_ptr(), ctypes.c_uint64(nbytes)))
        return out

    def decompress(self, img_encoded: bytes) -> Image:
        """Decompresses encoded image data (jpeg.py:274-297): container parsing and zlib on the host, everything else on the GPU."""
        meta, layers = self._entropy_decode(img_encoded)
        H, W = self.layer_shape
        ctx = self._bind()
        t = ctx.torch
        plan = ctx.plan(1, H, W)
        coeffs = np.zeros(plan.coeff_stride, np.int32)
        leaves = np.zeros((plan.leaf_stride, 4), np.int32)
        counts = np.zeros((1, 3, 4), np.int64)
        for l, L in enumerate(layers):
            h, w = (int(v) for v in self.layer_shapes[l])
            sizes = np.asarray(Jpeg._decode_leaf_sizes(L["states"], L["root_size"]), dtype=np.int32)
            root = tables.largest_power_of_2(max(h, w)) * 2                     # jpeg.py:425
            # a stream whose header does not describe a tiling with the settings' block sizes must not reach the GPU tables
            leaf_cap = (plan.leaf_off[l + 1] if l < 2 else plan.leaf_stride) - plan.leaf_off[l]
            if len(sizes) > leaf_cap or (len(sizes) and (sizes.min() < self._block_sizes[0] or sizes.max() > self._block_sizes[-1])):
                raise ValueError("corrupt stream: leaf sizes outside the block-size range of the header, or more leaves than the layer holds")
            xy = np.zeros((len(sizes), 2), np.int32)
            placed = ctx.lib.aej_leaf_positions_host(sizes.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(sizes)), root, h, w,
                                                     xy.ctypes.data_as(ctypes.c_void_p))
            if placed != len(sizes):
                raise ValueError("corrupt stream: quadtree header does not tile the layer")
            offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64) ** 2)])
            if offs[-1] != L["coeffs"].size or offs[-1] > plan.coeff_stride:
                raise ValueError("corrupt stream: coefficient count does not match the quadtree header")
            co, lo = plan.coeff_off[l], plan.leaf_off[l]
            coeffs[co:co + offs[-1]] = L["coeffs"]
            leaves[lo:lo + len(sizes), 0:2] = xy
            leaves[lo:lo + len(sizes), 2] = sizes
            leaves[lo:lo + len(sizes), 3] = offs[:-1]
            counts[0, l] = (offs[-1], len(sizes), len(L["states"]), L["root_size"])
        enc = EncodedBatch(plan, ctx.to_device(coeffs, t.int32), ctx.to_device(leaves, t.int32), None, ctx.to_device(counts, t.int64))
        rgb = self.decompress_batch(enc)[0].cpu().numpy()
        return Image.from_array(rgb, rgb.shape, self.extension)

    def _entropy_decode(self, encoded_data: bytes):
        """Container parsing of jpeg.py:599-661: restores settings from the JSON header (jpeg.py:613-631) and returns the
        per-layer state symbols, root size and zigzag-ordered coefficients."""
        s = BytesIO(encoded_data)
        mlen = int.from_bytes(s.read(4), byteorder="big")
        meta = json.loads(s.read(mlen).decode("utf-8"))
        self.extension = meta["extension"]
        self.update_settings(JpegCompressionSettings(color_space=meta["color_space"],
                                                     quality_range=(meta["quality_min"], meta["quality_max"]),
                                                     block_size_range=(meta["block_size_min"], meta["block_size_max"])),
                             (meta["height"], meta["width"]))
        layers = []
        for _ in range(meta["num_layers"]):
            bits_len = int.from_bytes(s.read(4), byteorder="big")
            root_size = int.from_bytes(s.read(4), byteorder="big")
            packed = np.frombuffer(s.read((bits_len + 7) // 8), dtype=np.uint8)
            st = np.stack([(packed >> 6) & 3, (packed >> 4) & 3, (packed >> 2) & 3, packed & 3], 1).reshape(-1)[: bits_len // 2]
            clen = int.from_bytes(s.read(4), byteorder="big")
            coeffs = np.frombuffer(zlib.decompress(s.read(clen)), dtype=np.int32)
            layers.append({"states": st.tolist(), "root_size": root_size, "coeffs": coeffs})
        return meta, layers

    # ------------------------------------------------------------------ .ajpg container (jpeg.py:531-597)
    def _header_bytes(self, num_layers: int) -> bytes:
        metadata = {
            "height": int(self.layer_shape[0]), "width": int(self.layer_shape[1]), "num_layers": num_layers,
            "color_space": self.settings.color_space,
            "quality_min": self.settings.quality_range[0], "quality_max": self.settings.quality_range[1],
            "block_size_min": self.settings.block_size_range[0], "block_size_max": self.settings.block_size_range[1],
            "extension": self.extension,
        }
        mb = json.dumps(metadata).encode("utf-8")
        return len(mb).to_bytes(4, byteorder="big") + mb

    @staticmethod
    def _layer_pieces(L, zlib_level: int = 9, stream=None) -> tuple:
        """The pieces of one layer record of the container (jpeg.py:561-595): state bits, root size, zlib-9 of the int32 coefficients
        (``zlib_level`` other than the reference's 9 is an opt-in: any level gives a stream the reference's ``zlib.decompress`` reads).
        ``stream``: the finished zlib stream (bytes or a buffer view), when it was written elsewhere."""
        st = L["states"]
        bits_len = 2 * len(st)
        pad = (-len(st)) % 4
        quad = np.concatenate([st, np.zeros(pad, np.uint8)]).reshape(-1, 4)
        packed = ((quad[:, 0] << 6) | (quad[:, 1] << 4) | (quad[:, 2] << 2) | quad[:, 3]).astype(np.uint8)
        comp = stream if stream is not None else zlib.compress(np.ascontiguousarray(L["coeffs"], dtype=np.int32).tobytes(), level=zlib_level)
        return (bits_len.to_bytes(4, byteorder="big"), int(L["root_size"]).to_bytes(4, byteorder="big"), packed.tobytes(),
                len(comp).to_bytes(4, byteorder="big"), comp)

    @staticmethod
    def _layer_bytes(L, zlib_level: int = 9, stream=None) -> bytes:
        return b"".join(Jpeg._layer_pieces(L, zlib_level, stream))

    def _entropy_encode(self, layers) -> bytes:
        """Header + one record per layer (jpeg.py:531-597).  The layers' zlib streams are independent (jpeg.py:588-595 writes them one
        after the other) and ``zlib.compress`` releases the GIL, so they are deflated on one thread each: the same bytes, in the time
        of the largest layer (luma: two thirds of the coefficients) instead of the sum."""
        if len(layers) <= 1:
            recs = [self._layer_bytes(L) for L in layers]
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=len(layers)) as ex:
                recs = list(ex.map(self._layer_bytes, layers))
        return self._header_bytes(len(layers)) + b"".join(recs)

    def _host_batch_for_upload(self, batch):
        """A host float32 batch whose values are all exactly ``float32(k) / 255`` -- what ``Image.load`` produces (image.py:80) -- crosses
        PCIe as the uint8 levels ``k`` (3 B per pixel instead of 12) and takes the 8-bit ingest entry, which forms the same float32
        quotients on the GPU: identical outputs.  One threaded pass of a host helper of the library decides (it stops at the first value
        that is not such a quotient); device tensors, uint8 input and everything else are returned unchanged."""
        if not isinstance(batch, np.ndarray) or batch.dtype != np.float32 or batch.size < (1 << 16):
            return batch
        ctx = self._bind()
        src = np.ascontiguousarray(batch)
        out = np.empty(src.shape, np.uint8)
        ok = ctx.lib.aej_pack_u8_levels_host(src.ctypes.data, ctypes.c_int64(src.size), out.ctypes.data, usable_cpus())
        return out if ok == 1 else batch

    def deflate_batch(self, enc: EncodedBatch, adaptive: bool = True, as_views: bool = False, tables: Optional[np.ndarray] = None) -> List[List[bytes]]:
        """OPT-IN GPU entropy stage (``aej_deflate_histogram`` / ``aej_deflate_batch``, csrc/deflate.hip): the zlib stream of every layer
        of an encoded batch, written on the GPU -- one deflate block per stream, LZ77 matches from an exact hash-chain search over 32 KiB
        chunks, tokens chosen by dynamic programming -- which ``zlib.decompress`` (the reference's decoder, jpeg.py:659) reads like any
        other stream.  ``adaptive``: one dynamic Huffman code per layer, built by the library's host helper from the token histogram the
        GPU counts (a 4 KB round trip); otherwise RFC 1951's fixed code.  Only the compressed bytes cross to the host.
        -> [image][layer] bytes (``as_views``: memoryviews into the page-locked host buffer the device-to-host copy filled -- valid until
        this context's next ``deflate_batch`` -- for callers that assemble larger records at once and want no intermediate copy; ``tables``: use these codes ([3][deflate_tables.TABLE_WORDS] uint32, e.g.
        kept from an earlier batch) instead of counting -- a stream that needs a symbol they have no code for takes the fixed code)."""
        from . import deflate_tables as DT
        ctx = self._bind()
        t = ctx.torch
        p = enc.plan
        nbytes = int(ctx.lib.aej_deflate_workspace_bytes(ctx.handle, p.batch, p.H, p.W))
        ws = ctx.workspace(nbytes)
        parsed = 0
        if tables is not None:
            tables = ctx.to_device(np.ascontiguousarray(tables, dtype=np.uint32).view(np.int32), t.int32)
        elif adaptive:
            hist = ctx.empty((3, DT.HIST_BINS), t.int32)
            ctx.check(ctx.lib.aej_deflate_histogram(ctx.handle, enc.coeffs.data_ptr(), enc.counts.data_ptr(), p.batch, p.H, p.W, hist.data_ptr(),
                                                    ws.data_ptr(), ctypes.c_uint64(nbytes)))
            parsed = 1                              # the tokens stay in the workspace: aej_deflate_batch does not parse again
            h = np.ascontiguousarray(hist.cpu().numpy(), dtype=np.int32)
            # codes for exactly the symbols that occur: the table is used on the very data (and parse) it was counted on
            cover = np.zeros(3, np.int32)
            tab = np.empty((3, DT.TABLE_WORDS), np.uint32)
            if ctx.lib.aej_deflate_build_tables(h.ctypes.data, cover.ctypes.data, tab.ctypes.data):
                raise AejError("aej_deflate_build_tables failed")
            tables = ctx.to_device(tab.view(np.int32), t.int32)
        cap = max((p.coeff_off[l + 1] if l < 2 else p.coeff_stride) - p.coeff_off[l] for l in range(3))
        stride = int(ctx.lib.aej_deflate_stream_bound(ctypes.c_uint64(4 * cap)))
        stride = (stride + 255) // 256 * 256
        n = p.batch * 3
        streams = ctx.empty((n, stride), t.uint8)
        sizes = ctx.empty((n,), t.int64)
        ctx.check(ctx.lib.aej_deflate_batch(ctx.handle, enc.coeffs.data_ptr(), enc.counts.data_ptr(), p.batch, p.H, p.W,
                                            tables.data_ptr() if tables is not None else None, parsed, streams.data_ptr(),
                                            ctypes.c_uint64(stride), sizes.data_ptr(), ws.data_ptr(), ctypes.c_uint64(nbytes)))
        sz = sizes.cpu().numpy()
        off = np.concatenate([[0], np.cumsum(sz)])
        # compacted on the device by ONE gather launch, then one device-to-host copy of the compressed bytes into page-locked memory
        packed = t.cat([streams[i, :int(sz[i])] for i in range(n)])
        stage = ctx.pinned(packed.numel())[:packed.numel()]
        stage.copy_(packed)
        host = stage.numpy()                        # (as_views: valid until this context's next deflate_batch)
        view = memoryview(host)
        cut = (lambda a, b: view[a:b]) if as_views else (lambda a, b: host[a:b].tobytes())
        return [[cut(int(off[3 * b + l]), int(off[3 * b + l + 1])) for l in range(3)] for b in range(p.batch)]

    def compress_many(self, batch, extension: Optional[str] = None, workers: Optional[int] = None, zlib_level: int = 9,
                      entropy: str = "host") -> List[bytes]:
        """``compress`` for a batch: one GPU pass (``compress_batch``), then the container of every image with the
        per-layer zlib-9 streams -- the part of ``compress`` that dominates end to end -- deflated on a thread pool (zlib
        releases the GIL).  Each element equals ``compress(Image(batch[i]))`` byte for byte.

        ``zlib_level`` is an opt-in for throughput: with the hot path on the GPU, level 9 deflate of the int32 coefficients IS the
        call (98 % of it on natural 4K images: 1.4 MP/s per host core against 80 000 MP/s for everything before it,
        profiles/r04_bench_extra*.json); a lower level writes a larger container that ``Jpeg.decompress`` -- the reference's included,
        jpeg.py:659 -- reads unchanged.  ``entropy="gpu"`` goes further: the streams are written on the GPU (``deflate_batch``) and only
        compressed bytes cross to the host.  The default stays host zlib level 9, the reference's (jpeg.py:590), so the bytes stay the
        reference's."""
        from concurrent.futures import ThreadPoolExecutor
        if entropy not in ("host", "gpu", "gpu-fixed"):
            raise ValueError("entropy must be 'host', 'gpu' or 'gpu-fixed'")
        enc = self.compress_batch(self._host_batch_for_upload(batch))
        p = enc.plan
        self.update_layer_shapes((p.H, p.W))
        self.extension = extension
        header = self._header_bytes(3)
        if entropy in ("gpu", "gpu-fixed"):
            streams = self.deflate_batch(enc, adaptive=entropy == "gpu", as_views=True)
            cnt = enc.counts_host
            aligned = p.state_stride % 4 == 0 and all(p.state_off[l] % 4 == 0 for l in range(3))
            if aligned:
                # the 2-bit packing of the state symbols (jpeg.py:561-578) for the whole batch in a few device operations: the layers' slots
                # start on multiples of four symbols, so packing the buffer four by four packs every layer; only a layer's last byte can
                # hold symbols from beyond its end, and those bits are cleared below (the reference pads with zeros)
                q = enc.states.view(-1, 4) & 3              # (what lies beyond a layer's last symbol is not ours: keep it out of the neighbouring bit fields)
                packed_all = ((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).cpu().numpy()
            else:
                states = enc.states.cpu().numpy()       # every image's state symbols in ONE device-to-host copy
            out = []
            for b in range(p.batch):
                pieces = [header]
                for l in range(3):
                    so, n_st = b * p.state_stride + p.state_off[l], int(cnt[b, l, 2])
                    if aligned:
                        seg = packed_all[so // 4:so // 4 + (n_st + 3) // 4]
                        if n_st % 4:
                            seg = seg.copy()
                            seg[-1] &= (0xFF << (2 * (4 - n_st % 4))) & 0xFF
                        comp = streams[b][l]
                        pieces += [(2 * n_st).to_bytes(4, byteorder="big"), int(cnt[b, l, 3]).to_bytes(4, byteorder="big"), seg.data,
                                   len(comp).to_bytes(4, byteorder="big"), comp]
                    else:
                        pieces += self._layer_pieces({"states": states[so:so + n_st], "root_size": int(cnt[b, l, 3])}, stream=streams[b][l])
                out.append(b"".join(pieces))            # the one copy of the compressed bytes on the host
            return out
        jobs = [(b, l) for b in range(p.batch) for l in range(3)]
        with ThreadPoolExecutor(max_workers=workers or usable_cpus()) as ex:
            recs = list(ex.map(lambda bl: self._layer_bytes(enc.layer(bl[0], bl[1]), zlib_level), jobs))
        return [header + b"".join(recs[3 * b:3 * b + 3]) for b in range(p.batch)]

    # ------------------------------------------------------------------ small helpers, reference names kept
    def _compute_downsampled_shapes(self, layer_shapes) -> np.ndarray:
        return np.asarray(layer_shapes) // self.settings.downsampling_ratios      # jpeg.py:676-686

    def _get_quality_factor(self, block_size: int) -> int:
        return tables.quality_factor(block_size, self.settings.block_size_range, self.settings.quality_range)

    @staticmethod
    def _get_quantization_matrix(default_matrix: np.ndarray, size: int, quality: int) -> np.ndarray:
        return tables.quantization_matrix(default_matrix, size, quality)

    @staticmethod
    def _zigzag_ordering(size: int) -> np.ndarray:
        return tables.zigzag_ordering(size)

    @staticmethod
    def _decode_leaf_sizes(states: List[int], root_size: int) -> List[int]:
        """jpeg.py:768-800"""
        sizes, stack, i = [], [root_size], 0
        while stack and i < len(states):
            size = stack.pop()
            s = states[i]
            i += 1
            if s == 0:
                sizes.append(size)
            elif s == 1:
                stack.extend([size // 2] * 4)
        return sizes
