"""One benchmarked workload: a batch shape + settings, two device-resident input batches, and the contexts its steps rotate over.

A *step* is one pass of the whole encode hot path over one batch.  Step i is enqueued (aej_encode_batch_begin) on context i % n, each
context on its own stream with its own output buffers and workspace, after the step that used that context before has been ended
(aej_encode_batch_end: waited for, device counters checked).  Inputs alternate between two batches of different seeds, so nothing
data-dependent can be remembered from one call to the next.  `sync()` ends every call in flight.
"""
import numpy as np


class Pipe:
    """one context on one stream with its own outputs (and, inside the context, its own workspace)"""

    def __init__(self, wl, stream):
        torch, jpeg = wl.torch, wl.jpeg
        self.wl, self.stream = wl, stream
        with torch.cuda.stream(stream):
            self.ctx = jpeg._bind()
            self.ctx.set_graph_mode(wl.graph)
            self.ctx.set_sub_batches(wl.pipelined_sub_batches)
            for kv in wl.options:
                self.ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            self.plan = self.ctx.plan(wl.B, wl.H, wl.W)
            p = self.plan
            self.out = (self.ctx.empty((wl.B * p.coeff_stride,), torch.int32), self.ctx.empty((wl.B * p.leaf_stride, 4), torch.int32),
                        self.ctx.empty((wl.B * p.state_stride,), torch.uint8), self.ctx.empty((wl.B, 3, 4), torch.int64))
        self.pending = False
        self.input = 0

    def end(self):
        if self.pending:
            self.pending = False
            with self.wl.torch.cuda.stream(self.stream):
                self.wl.jpeg.encode_end(self.ctx)

    def begin(self, which):
        self.end()
        with self.wl.torch.cuda.stream(self.stream):
            self.wl.jpeg.encode_begin(self.ctx, self.wl.batches[which], self.plan, *self.out)
        self.pending, self.input = True, which


class Workload:
    def __init__(self, torch, A, dev, batches_f32, *, space, qrange, brange, ingest="f32", n_pipe=3, graph=0, sub_batches=0, pipelined_sub_batches=None,
                 options=()):
        from .data import to_u8
        self.torch, self.dev = torch, dev
        self.batches_f32 = batches_f32
        self.B, self.H, self.W = (int(v) for v in batches_f32[0].shape[:3])
        self.space, self.qrange, self.brange, self.ingest = space, tuple(qrange), tuple(brange), ingest
        # aej_set_sub_batches: `sub_batches` for the blocking calls (0 = the library's automatic choice), `pipelined_sub_batches` for the steps
        # that rotate over the contexts (a caller that keeps four calls in flight wants each cut in two, a lone blocking call in four)
        self.graph, self.sub_batches, self.options = graph, sub_batches, list(options)
        self.pipelined_sub_batches = sub_batches if pipelined_sub_batches is None else pipelined_sub_batches
        self.batches = batches_f32 if ingest == "f32" else [to_u8(torch, x) for x in batches_f32]
        self.jpeg = A.Jpeg(A.JpegCompressionSettings(space, self.qrange, self.brange), device=dev.index)
        # every context on a stream of its own, none on the legacy null stream: once other streams exist, launches on the null stream shift
        # the HIP-event stage attribution (colour planes +0.25 ms, blur -0.09 ms, profiles/r02_null_stream_stage_attribution.txt)
        self.pipes = [Pipe(self, torch.cuda.Stream(device=dev)) for _ in range(n_pipe)]
        torch.cuda.synchronize()
        self.n_pipe = n_pipe
        self.ctx, self.plan = self.pipes[0].ctx, self.pipes[0].plan
        if self.sub_batches != self.pipelined_sub_batches:
            # the blocking calls split differently from the pipelined steps: let context 0 create the sub-batch streams of BOTH shapes now.  Streams
            # created after every other stream of the process exists measured 0.3-0.4 ms per blocking call slower (profiles/r05_sched_sweep.txt)
            self.serial_step(0)
            torch.cuda.synchronize()

    @property
    def pixels_per_step(self):
        return self.B * self.H * self.W

    def input_of(self, i):
        # inputs alternate on EVERY context (a context that saw the same batch each time could keep something from it)
        return (i // self.n_pipe + i) & 1

    def step(self, i):
        """throughput loop: enqueue step i on context i % n; the step that used it before is ended first"""
        self.pipes[i % self.n_pipe].begin(self.input_of(i))

    def serial_step(self, i):
        """one blocking call at a time on context 0"""
        p = self.pipes[0]
        with self.torch.cuda.stream(p.stream):
            if self.sub_batches != self.pipelined_sub_batches:
                p.ctx.set_sub_batches(self.sub_batches)
            self.jpeg.encode_into(p.ctx, self.batches[i & 1], p.plan, *p.out)
            if self.sub_batches != self.pipelined_sub_batches:
                p.ctx.set_sub_batches(self.pipelined_sub_batches)

    def sync(self):
        for p in self.pipes:
            p.end()
        self.torch.cuda.synchronize()

    def warm(self, n):
        n = max(n, 2 * self.n_pipe)               # every context has seen both inputs
        for i in range(n):
            self.step(i)
        self.sync()
        return n

    def hyst_stats(self):
        tot = {}
        for p in self.pipes:
            for k, v in p.ctx.hysteresis_stats().items():
                tot[k] = (tot.get(k, 0) + v) if k != "queued" else max(tot.get(k, 0), v)
        return tot

    def last_step(self, steps):
        """-> (pipe, which input batch) of timed step `steps - 1`"""
        return self.pipes[(steps - 1) % self.n_pipe], self.input_of(steps - 1)

    def verify(self, O, steps, picks):
        """The outputs of the LAST timed step against the CPU oracle for the images `picks`: quadtree states, leaf table and quantised
        zigzag coefficients of all three layers, bit for bit."""
        from concurrent.futures import ThreadPoolExecutor
        from adaptive_edge_aware_jpeg_amd.jpeg import EncodedBatch
        pipe, which = self.last_step(steps)
        enc = EncodedBatch(pipe.plan, *pipe.out)
        with ThreadPoolExecutor(max_workers=len(picks)) as ex:
            refs = list(ex.map(lambda b: O.encode_image(self.batches_f32[which][b].cpu().numpy(), self.space, self.qrange, self.brange), picks))
        ok = True
        for b, ref in zip(picks, refs):
            for l in range(3):
                got = enc.layer(b, l)
                ok = ok and got["root_size"] == ref[l]["root_size"] and all(np.array_equal(got[k], ref[l][k]) for k in ("states", "leaves", "coeffs"))
        return {"ok": bool(ok), "images": list(picks), "of_batch": "A" if which == 0 else "B",
                "what": "quadtree states, leaf table and quantised zigzag coefficients of all 3 layers, bit-exact vs the CPU oracle"}

    def stage_ms(self, n_prof=4):
        """per-stage times: separate, untimed, profiled blocking steps (HIP events on the launch stream inside the library)"""
        ctx = self.ctx
        ctx.set_profiling(True)
        acc = {}
        self.serial_step(0); self.serial_step(1)
        for i in range(n_prof):
            self.serial_step(i)
            for k, v in ctx.stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v
        ctx.set_profiling(False)
        return {k: v / n_prof for k, v in acc.items()}

    def leaf_histogram(self):
        """leaf-size histogram of context 0's last batch ("report the leaf-size histogram with every number") -> (hist, counts)"""
        torch, plan, dev = self.torch, self.plan, self.dev
        coeffs, leaves, states, counts = self.pipes[0].out
        cnt = counts.cpu().numpy()
        lv = leaves.view(self.B, plan.leaf_stride, 4)
        hist = {}
        for l in range(3):
            n_l = torch.from_numpy(cnt[:, l, 1].copy()).to(dev)
            lo = int(plan.leaf_off[l])
            cap = int(cnt[:, l, 1].max())
            sz = lv[:, lo:lo + cap, 2]
            valid = torch.arange(cap, device=dev)[None, :] < n_l[:, None]
            s = self.brange[0]
            while s <= self.brange[1]:
                hist[s] = hist.get(s, 0) + int(((sz == s) & valid).sum().item())
                s *= 2
        return hist, cnt

    def close(self):
        from adaptive_edge_aware_jpeg_amd._lib import release_context
        self.sync()
        for p in self.pipes:
            release_context(p.ctx)
            p.out = None
        self.pipes = []
        self.ctx = self.plan = None
        self.batches = self.batches_f32 = None
        self.torch.cuda.empty_cache()
