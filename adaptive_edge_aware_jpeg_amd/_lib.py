"""ctypes binding of libaejpeg_hip.so (include/aej.h) + a small device-context wrapper.

PyTorch-ROCm is used for device memory and streams only: every tensor handed to the library is a
``torch`` CUDA(=HIP) tensor and the library receives raw device pointers.  There is no CPU fallback: if
the shared library is missing or no GPU is visible, the calls raise.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AEJ_LIBRARY: developer override used by tools/profiling/variants.py to time experimental builds of the same ABI side by side
# (they live under build/variants/, never in place of the in-tree library)
LIB_PATH = os.environ.get("AEJ_LIBRARY") or os.path.join(_HERE, "libaejpeg_hip.so")

SPACE_IDS = {"YCbCr": 0, "YCoCg": 1, "YCoCg-R": 2, "OKLAB": 3, "ICtCp": 4, "ICaCb": 5, "JzAzBz": 6}
CONVERT_IDS = dict(SPACE_IDS, XYZ=7)     # color.convert also serves the helper space XYZ (conversion.py:63-68); not a codec space

AEJ_ERR_ARG, AEJ_ERR_HIP, AEJ_ERR_STATE, AEJ_ERR_CAPACITY, AEJ_ERR_UNSUPPORTED = -1, -2, -3, -4, -5


class AejError(RuntimeError):
    pass


class AejPlan(ctypes.Structure):
    _fields_ = [
        ("batch", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("layer_h", ctypes.c_int32 * 3), ("layer_w", ctypes.c_int32 * 3), ("root_size", ctypes.c_int32 * 3),
        ("coeff_off", ctypes.c_int64 * 3), ("coeff_stride", ctypes.c_int64),
        ("leaf_off", ctypes.c_int64 * 3), ("leaf_stride", ctypes.c_int64),
        ("state_off", ctypes.c_int64 * 3), ("state_stride", ctypes.c_int64),
        ("workspace_bytes", ctypes.c_uint64),
    ]


_lib = None
_lib_lock = threading.Lock()

# ---- hardware queues ------------------------------------------------------------------------------------------------
# HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams on one queue run one after the other.
# The library overlaps sub-batches and calls on several streams and picks its schedule by the number of queues --
# which only the host can know: the runtime reads the variable once, when it initialises (any HIP call does that,
# torch.cuda.is_available() included).  IMPORTING THIS PACKAGE CHANGES NOTHING IN THE PROCESS: it only looks.
#   * the variable is in the environment when the package is imported (set by the launcher, or by a script before it imported torch,
#     as bench.py does): that is what the runtime starts with -- its value is what the library is told;
#   * otherwise the library is told HIP's default of 4 (its conservative two-sub-batch schedule).  An application that wants the wide
#     schedule calls configure_hw_queues() BEFORE anything initialises HIP (it sets the variable, explicitly, at the caller's request),
#     or states the value its runtime really started with through set_hw_queues().
HIP_DEFAULT_HW_QUEUES = 4
_hw_queues = HIP_DEFAULT_HW_QUEUES
_hw_queues_source = "HIP default (assumed)"


def _look_at_hw_queues():
    """Called once at package import: read-only (see the policy above)."""
    global _hw_queues, _hw_queues_source
    present = os.environ.get("GPU_MAX_HW_QUEUES")
    if present is not None:
        try:
            v = int(present)
        except ValueError:
            v = 0
        if v > 0:
            _hw_queues, _hw_queues_source = v, "GPU_MAX_HW_QUEUES was in the environment when the package was imported"
    return _hw_queues


def configure_hw_queues(want=16):
    """Explicit opt-in, to be called BEFORE `import torch` (the HIP runtime reads GPU_MAX_HW_QUEUES once, when it initialises, and
    torch.cuda.is_available() / device_count() already initialise it): puts GPU_MAX_HW_QUEUES=<want> into this process's environment so
    that every stream of the library gets a hardware queue of its own, and tells the library.  If torch is already imported nobody can
    know whether the runtime has read the variable: it is still set (it cannot hurt) but the library keeps HIP's default of 4 -- told 16
    while the runtime has 4 it would put four sub-batches on streams that serialise, measured slower than the conservative schedule --
    unless the application states the real value through set_hw_queues().  Returns the queue count the library will schedule for."""
    global _hw_queues, _hw_queues_source
    import sys
    if os.environ.get("GPU_MAX_HW_QUEUES") is not None:
        return _look_at_hw_queues()
    trusted = "torch" not in sys.modules
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(want))
    if trusted:
        _hw_queues, _hw_queues_source = int(want), "configure_hw_queues() before torch was imported"
        for ctx in _contexts.values():
            ctx.check(ctx.lib.aej_set_hw_queues(ctx.handle, _hw_queues))
    return _hw_queues


def set_hw_queues(n):
    """For applications that initialise HIP before importing this package: the value of GPU_MAX_HW_QUEUES their runtime started
    with.  Applies to contexts created afterwards and to the existing ones."""
    global _hw_queues, _hw_queues_source
    _hw_queues, _hw_queues_source = int(n), "stated by the application (set_hw_queues)"
    for ctx in _contexts.values():
        ctx.check(ctx.lib.aej_set_hw_queues(ctx.handle, _hw_queues))


def hw_queues():
    """-> (queues the library schedules for, where the number comes from)"""
    return _hw_queues, _hw_queues_source


_warned_queue_limited = False


def warn_if_queue_limited(ctx, batch, H, W):
    """Once per process: a call large enough for the library to cut it into four overlapping sub-batches runs the conservative schedule
    because the library was told the HIP runtime has fewer than 8 hardware queues (its default is 4) -- the drop-in then runs about 10 %
    below what bench.py measures.  Says how to get the wide schedule."""
    global _warned_queue_limited
    if _warned_queue_limited:
        return
    sch = ctx.schedule(batch, H, W)
    if sch["limited_by_hw_queues"]:
        _warned_queue_limited = True
        import warnings
        warnings.warn(f"adaptive_edge_aware_jpeg_amd: this call ({batch} x {W}x{H}) would run as more overlapping sub-batches with more hardware "
                      f"queues; the library schedules for {sch['hw_queues']} ({_hw_queues_source}).  Put GPU_MAX_HW_QUEUES=16 into the environment "
                      "(or call adaptive_edge_aware_jpeg_amd.configure_hw_queues()) before `import torch`, or state the runtime's real value with "
                      "set_hw_queues().", RuntimeWarning, stacklevel=3)


# name -> (restype, argtypes); exactly the symbols declared in include/aej.h
_P, _I, _I64, _U64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64
class CannyParams(ctypes.Structure):
    """aej_canny_params (include/aej.h)"""
    _fields_ = [("canny_low_ratio", ctypes.c_double), ("canny_high_ratio", ctypes.c_double), ("clahe_clip_limit", ctypes.c_double),
                ("bilateral_sigma_color", ctypes.c_double), ("bilateral_sigma_space", ctypes.c_double), ("use_l2_gradient", ctypes.c_int)]


SIGNATURES = {
    "aej_abi_version": (_I, []),
    "aej_create": (_P, [_I, _P]),
    "aej_destroy": (None, [_P]),
    "aej_last_error": (ctypes.c_char_p, [_P]),
    "aej_synchronize": (_I, [_P]),
    "aej_set_stream": (_I, [_P, _P]),
    "aej_get_hysteresis_stats": (_I, [_P, _P]),
    "aej_set_graph_mode": (_I, [_P, _I]),
    "aej_set_sub_batches": (_I, [_P, _I]),
    "aej_set_hw_queues": (_I, [_P, _I]),
    "aej_get_schedule_host": (_I, [_P, _I, _I, _I, _P]),
    "aej_set_option": (_I, [_P, ctypes.c_char_p, _I64]),
    "aej_get_option": (_I, [_P, ctypes.c_char_p, _P]),
    "aej_test_fail_after_stage": (_I, [_P, _I]),            # include/aej_testing.h (tests only)
    "aej_encode_batch_begin": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _U64]),
    "aej_encode_batch_end": (_I, [_P]),
    "aej_get_split_calls": (ctypes.c_int64, [_P]),
    "aej_get_graph_stats": (_I, [_P, _P]),
    "aej_set_profiling": (_I, [_P, _I]),
    "aej_get_stage_ms": (_I, [_P, _P]),
    "aej_stage_name": (ctypes.c_char_p, [_I]),
    "aej_set_settings": (_I, [_P, _I, _I, _I, _P]),
    "aej_encode_plan": (_I, [_P, _I, _I, _I, ctypes.POINTER(AejPlan)]),
    "aej_encode_batch": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _U64]),
    "aej_encode_batch_u8": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _U64]),
    "aej_metrics_workspace_bytes": (_U64, [_I, _I, _I]),
    "aej_metrics_batch": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _U64]),
    "aej_color_convert": (_I, [_P, _I, _P, _P, _I64]),
    "aej_color_planes": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "aej_canny_workspace_bytes": (_U64, [_I, _I]),
    "aej_canny": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _U64]),
    "aej_set_canny_params": (_I, [_P, _P]),
    "aej_quadtree_workspace_bytes": (_U64, [_I, _I, _I, _I]),
    "aej_quadtree_capacity": (_I, [_I, _I, _I, _I, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "aej_quadtree": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _U64]),
    "aej_dct_quant_zigzag": (_I, [_P, _P, _I, _I, _I, _P, _I64, _P, _P]),
    "aej_deflate_stream_bound": (_U64, [_U64]),
    "aej_deflate_workspace_bytes": (_U64, [_P, _I, _I, _I]),
    "aej_deflate_histogram": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _U64]),
    "aej_deflate_batch": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _U64, _P, _P, _U64]),
    "aej_deflate_build_tables": (_I, [_P, _P, _P]),
    "aej_pack_u8_levels_host": (_I, [_P, _I64, _P, _I]),
    "aej_decode_workspace_bytes": (_U64, [_P, _I, _I, _I]),
    "aej_decode_batch": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _U64]),
    "aej_color_convert_inverse": (_I, [_P, _I, _P, _P, _I64]),
    "aej_leaf_positions_host": (_I64, [_P, _I64, _I, _I, _I, _P]),
}


def load_library():
    """Load libaejpeg_hip.so and declare every signature.  Raises if the HIP extension is not built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise AejError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
            # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  The
            # device pointers we receive come from that runtime, so it must be the one this library binds to:
            # import torch first, then the loader resolves our NEEDED libamdhip64.so.7 to the loaded copy.
            import torch  # noqa: F401
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise AejError("no HIP device visible to PyTorch-ROCm: the adaptive-JPEG encode path runs on the GPU only")
    return torch


class Context:
    """One aej_ctx bound to a device and to ONE torch stream of it (the stream that was current when get_context()
    created it).  Everything torch does for a call -- input copies, output and workspace allocation -- happens on the stream
    current at call time, so the library must enqueue on that same stream; get_context() therefore hands out one context per
    (device, stream), each with its own workspace, and `check_stream()` refuses a context used under another stream."""

    def __init__(self, device=0):
        self.torch = _torch()
        self.lib = load_library()
        self.device = self.torch.device("cuda", device if isinstance(device, int) else device.index or 0)
        self.stream = self.torch.cuda.current_stream(self.device).cuda_stream
        self.handle = self.lib.aej_create(self.device.index, ctypes.c_void_p(self.stream))
        if not self.handle:
            raise AejError(self.lib.aej_last_error(None).decode())
        self.check(self.lib.aej_set_hw_queues(self.handle, _hw_queues))
        self.settings_key = None
        self._ws = None

    def check_stream(self):
        cur = self.torch.cuda.current_stream(self.device).cuda_stream
        if cur != self.stream:
            raise AejError(f"context bound to stream {self.stream:#x} used while torch's current stream is {cur:#x}: "
                           "obtain the context with get_context() inside the `torch.cuda.stream(...)` block")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.aej_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- helpers
    def check(self, rc):
        if rc == 0:
            return
        msg = self.lib.aej_last_error(self.handle).decode()
        if rc == AEJ_ERR_ARG:
            raise ValueError(msg)
        if rc == AEJ_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise AejError(f"aej error {rc}: {msg}")

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    def workspace(self, nbytes):
        """Scratch of at least `nbytes`, reused across calls.  It is allocated on, and only ever used by kernels of, this
        context's stream, so dropping the old tensor when a larger one is needed is safe: torch's caching allocator hands a
        freed block only to later work of the same stream."""
        self.check_stream()
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
        return self._ws

    def pinned(self, nbytes):
        """Page-locked host staging buffer of at least `nbytes`, reused across calls (device-to-host copies into pageable memory run at
        a third of the PCIe rate).  Its contents are only valid until the next call that uses it."""
        if getattr(self, "_pinned", None) is None or self._pinned.numel() < nbytes:
            self._pinned = None
            self._pinned = self.torch.empty(int(max(nbytes, 1 << 20)), dtype=self.torch.uint8, pin_memory=True)
        return self._pinned

    def hysteresis_stats(self):
        """-> dict(calls, queued): whole-path calls on this context, tiles that went through the device-side hysteresis work queue
        in the last completed call (include/aej.h)."""
        buf = (ctypes.c_int64 * 4)()      # (4: round 3 libraries, loaded through AEJ_LIBRARY for A / B runs, wrote four)
        self.check(self.lib.aej_get_hysteresis_stats(self.handle, ctypes.cast(buf, ctypes.c_void_p)))
        return {"calls": int(buf[0]), "queued": int(buf[1])}

    def set_canny_params(self, params=None):
        """aej_set_canny_params: (low_ratio, high_ratio, clip_limit, sigma_color, sigma_space, use_l2) or None for the defaults"""
        if params is None:
            self.check(self.lib.aej_set_canny_params(self.handle, None))
            return
        lo, hi, clip, sc, ss, l2 = params
        v = CannyParams(float(lo), float(hi), float(clip), float(sc), float(ss), 1 if l2 else 0)
        self.check(self.lib.aej_set_canny_params(self.handle, ctypes.cast(ctypes.pointer(v), ctypes.c_void_p)))

    def set_sub_batches(self, n):
        """0 = automatic (default), 1 = never split a call, 2..8 = that many sub-batches on private streams (include/aej.h)"""
        self.check(self.lib.aej_set_sub_batches(self.handle, int(n)))

    def schedule(self, batch, H, W):
        """-> dict(hw_queues, sub_batches, limited): what the automatic mode would do for this call (aej_get_schedule_host)"""
        buf = (ctypes.c_int32 * 4)()
        self.check(self.lib.aej_get_schedule_host(self.handle, int(batch), int(H), int(W), ctypes.cast(buf, ctypes.c_void_p)))
        return {"hw_queues": int(buf[0]), "sub_batches": int(buf[1]), "limited_by_hw_queues": bool(buf[2]), "hw_queues_source": _hw_queues_source}

    def set_option(self, name, value):
        """aej_set_option: the library's tuning / A-B choices (table in include/aej.h); it reads no environment variable."""
        self.check(self.lib.aej_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name):
        v = ctypes.c_int64()
        self.check(self.lib.aej_get_option(self.handle, name.encode(), ctypes.cast(ctypes.pointer(v), ctypes.c_void_p)))
        return int(v.value)

    def split_calls(self):
        return int(self.lib.aej_get_split_calls(self.handle))

    def set_graph_mode(self, mode):
        """0 = never replay a captured hipGraph (default), 1 = automatic (small calls), 2 = whenever possible (include/aej.h)"""
        self.check(self.lib.aej_set_graph_mode(self.handle, int(mode)))

    def graph_stats(self):
        buf = (ctypes.c_int64 * 3)()
        self.check(self.lib.aej_get_graph_stats(self.handle, ctypes.cast(buf, ctypes.c_void_p)))
        return {"launches": int(buf[0]), "captures": int(buf[1]), "cached": int(buf[2])}

    def to_device(self, arr, dtype):
        t = self.torch
        if isinstance(arr, t.Tensor):
            return arr.to(device=self.device, dtype=dtype).contiguous()
        return t.from_numpy(np.ascontiguousarray(arr)).to(device=self.device, dtype=dtype).contiguous()

    def set_settings(self, space, bmin, bmax, qmats):
        """qmats: int32 numpy array laid out [layer][size][s*s] (see include/aej.h)."""
        key = (space, bmin, bmax, qmats.tobytes())
        if key == self.settings_key:
            return
        q = np.ascontiguousarray(qmats, dtype=np.int32)
        self.check(self.lib.aej_set_settings(self.handle, SPACE_IDS[space], bmin, bmax, q.ctypes.data_as(ctypes.c_void_p)))
        self.settings_key = key

    N_STAGES = 18

    def set_profiling(self, on):
        self.check(self.lib.aej_set_profiling(self.handle, 1 if on else 0))

    def stage_ms(self):
        """-> {stage name: milliseconds} for the last aej_encode_batch call (profiling must be on)."""
        buf = (ctypes.c_float * self.N_STAGES)()
        self.check(self.lib.aej_get_stage_ms(self.handle, ctypes.cast(buf, ctypes.c_void_p)))
        return {self.lib.aej_stage_name(i).decode(): float(buf[i]) for i in range(self.N_STAGES)}

    def plan(self, batch, H, W):
        p = AejPlan()
        self.check(self.lib.aej_encode_plan(self.handle, batch, H, W, ctypes.byref(p)))
        return p


_contexts = {}


def get_context(device=0):
    """The context of (device, torch's current stream on it); created on first use."""
    idx = device if isinstance(device, int) else (device.index or 0)
    torch = _torch()
    key = (idx, torch.cuda.current_stream(torch.device("cuda", idx)).cuda_stream)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _contexts[key] = Context(idx)
    return ctx


def release_context(ctx):
    """Destroy a context handed out by get_context() and forget it: the library's state for it is freed (aej_destroy) and its
    workspace goes back to torch's allocator.  For callers that walk through many streams or batch shapes (bench.py's other_configs)."""
    for key, known in list(_contexts.items()):
        if known is ctx:
            del _contexts[key]
    ctx._ws = None
    ctx._pinned = None
    ctx._in_flight = None
    if getattr(ctx, "handle", None):
        ctx.lib.aej_destroy(ctx.handle)
        ctx.handle = None
