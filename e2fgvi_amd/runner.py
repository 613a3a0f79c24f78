"""Clip-sharded execution: one process per GPU, clips partitioned over ranks, one RCCL all-gather
of the output frames per step (the only collective: clips are independent).
Also the HIP-graph replay of the forward (launch-bound at one clip per GPU: ~600 short kernels)."""
import torch


def shard_range(n_items, rank, world):
    """Contiguous balanced partition: ranks [0, n_items % world) get one extra item."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_frames(local_out, world, group=None, out=None, force=False):
    """All-gather equal-sized per-rank outputs [n,...] into [world*n,...] (rank-major = clip order)."""
    if world == 1 and not force:
        return local_out
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world * local_out.shape[0],) + tuple(local_out.shape[1:]), dtype=local_out.dtype,
                          device=local_out.device)
    dist.all_gather_into_tensor(out, local_out.contiguous(), group=group)
    return out


def inpaint_sharded(net, clips, num_local_frames, rank, world, group=None, pack_u8=False):
    """clips: [B,t,3,H,W] (same on every rank, or at least this rank's slice valid).  Every rank runs its
    contiguous share and all ranks receive all output frames [B*t,3,H,W].  B must be divisible by world.

    pack_u8: convert this rank's frames to the uint8 NHWC form test.py saves (uint8((pred+1)/2*255), HIP kernel) BEFORE
    the all-gather -- a quarter of the bytes over xGMI; returns uint8 [B*t,H,W,3]."""
    B = clips.shape[0]
    if B % world:
        raise ValueError("number of clips (%d) must be divisible by the number of ranks (%d)" % (B, world))
    lo, hi = shard_range(B, rank, world)
    out, _ = net(clips[lo:hi], num_local_frames)
    if pack_u8:
        from . import ops
        out = ops.pred_to_u8(out.contiguous())
    return gather_frames(out, world, group)


class whole_propagation:
    """context: the engine of `net` (if it has one) runs conv_offset.0 / backbone.0 whole, without the side-stream split of the
    one-clip forward (engine.PROP_SPLIT) -- what ShardedStep's pipelines capture, and what their frames are bit-equal to"""

    def __init__(self, net):
        eng = getattr(net, "engine", None)
        try:
            self.eng = eng() if callable(eng) else None
        except Exception:
            self.eng = None

    def __enter__(self):
        self.keep = getattr(self.eng, "prop_split", None)
        if self.keep:
            self.eng.prop_split = {}
        return self

    def __exit__(self, *a):
        if self.keep:
            self.eng.prop_split = self.keep
        return False


class ShardedStep:
    """One benchmark / serving step: forward of this rank's clips (+ optional uint8 packing) + all-gather of the frames.

    The forward (and the packing kernel) replays from a HIP graph once captured; the collective is NOT part of the graph.
    With a gather, step k's frames are first copied out of the forward's (static, under graph replay) output into one of
    two staging buffers, the all-gather of that buffer is issued asynchronously (RCCL's own stream) and runs under step
    k+1's forward; ``run()`` returns the gathered frames of the PREVIOUS step (None on the first call) and ``finish()``
    those of the last one; a returned buffer is valid until the next ``run()`` (two gather buffers alternate).
    Without a gather ``run()`` / ``finish()`` return this step's frames directly.

    in_flight = K > 1 (without a gather; round 6): K HIP graphs of the same forward, each with its own static buffers and its own
    stream, replayed round-robin -- step n + 1 starts while step n is still in its latency-bound propagation chain (the engine's
    state is read-only under replay: weights, key tables, zero buffers).  ``run()`` then returns the frames of the step issued
    K - 1 calls earlier (None for the first K - 1 calls), valid until the next ``run()``; ``finish()`` drains the pipelines and
    returns the last step's frames.  Ordering: a pipeline's replay waits for everything the caller has enqueued on the current
    stream so far (the consumer of the buffer it is about to overwrite), the current stream waits for the step it hands out."""

    def __init__(self, net, x, lt, group_world=1, use_graph=True, force_gather=False, pack_u8=False, group=None, in_flight=1):
        self.net, self.x, self.lt, self.world = net, x, lt, group_world
        self.in_flight = max(1, int(in_flight))
        self._pipes = None         # in_flight > 1: [(graph, static output, stream, event)]
        self.gather = group_world > 1 or force_gather
        self.pack_u8, self.group = pack_u8, group
        self.graph = None
        self.graphed = False
        self.out = None            # output of the last forward (static buffer once graphed)
        self.use_graph = use_graph
        self._calls = 0
        self._stage = None         # two staging buffers (this rank's frames), two gather buffers
        self._gathered = None
        self._pending = None       # (work, gathered buffer) of the step whose gather is in flight
        self._last = None
        self._synced = group_world <= 1

    def _forward(self):
        out, _ = self.net(self.x, self.lt)
        if self.pack_u8:
            from . import ops
            out = ops.pred_to_u8(out.contiguous())
        return out

    def _capture(self):
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._forward()                       # warm the allocator on the capture stream
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            # Captured on `s`, the stream the warm-up forward just ran on -- NOT on torch.cuda.graph's process-wide default capture
            # stream: the engine keeps one side stream per main stream (engine._side_stream), and on torch's stream every new engine
            # forked a side stream into the capture that had never run anything.  After earlier graphs of the process had been
            # destroyed, the replay of such a graph died in hip::Graph::UpdateStreams (SIGSEGV under hipGraphLaunch: `python bench.py
            # --no-cpu-baseline`, first secondary line, 5 of 5 runs).  On `s` both streams of the capture have run the
            # same forward eagerly one statement earlier.
            # thread_local: RCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                self.out = self._forward()
            self.graph, self.graphed = g, True
        except Exception as e:                         # capture unsupported -> stay eager, say so
            print("HIP graph capture failed (%s); running eagerly" % (str(e).splitlines()[0],), flush=True)
            self.use_graph = False
            self.graph = None
            torch.cuda.synchronize()

    def _sync_decisions(self):
        """Every rank replays rank 0's tile decisions: same kernels, same accumulation order, bit-identical frames from every
        rank for identical clips.  With the default table-driven selection (ops.py) the ranks agree by construction; under
        E2FGVI_AUTOTUNE=1 each rank times its own candidates during its FIRST eager forward, so the broadcast has to come
        after that forward and before the graph capture (or the second eager forward) freezes the choices: the first call of
        run() is this rank's tuning forward, its output is recomputed with rank 0's table when the tables differed."""
        from . import ops
        self._synced = True
        mine = dict(ops._TUNED)
        if ops.sync_tile_decisions(self.group) and ops._TUNED != mine:
            self.out = self._forward()

    def _pipeline_setup(self):
        """K graphs (each with its own static buffers) and K replay streams.  Which hardware queue HIP gives a stream depends on
        how many streams the process has made before, and two pipelines whose streams (or whose graphs' internal side branches)
        share a queue run in turn instead of side by side: measured 808 ... 900 frames/s for K = 3 on one box, by creation order
        alone (profiles/r06_forwards_in_flight.txt).  So the streams are CHOSEN: K graphs are captured once (a graph replays on
        whatever stream is current), a window of K consecutive streams slides over a dozen candidates, each window replays 2 K
        steps, the fastest one stays.  The choice moves scheduling only: every window returns the same bits."""
        import time
        keep = self.out
        graphs = []
        # The pipelines run the propagation layers WHOLE: engine.PROP_SPLIT takes the non-recurrent input channels of conv_offset.0 /
        # backbone.0 out of the one-clip chain into four batched side launches -- +2.3 % for a forward that has the GPU to itself,
        # -2 ... -4 % once another forward fills the chain's idle CUs and the side launches only compete with it
        # (tools/inflight_ab.py: 913.7 with the split, 933.6 without, two in flight, same box; sequential: 818.5 / 799.4).
        with whole_propagation(self.net):
            for _ in range(self.in_flight):
                self.graph = None
                self._capture()
                if self.graph is None:                 # capture unsupported: sequential eager steps
                    self.in_flight, self.out = 1, keep
                    return False
                graphs.append((self.graph, self.out))
        K = self.in_flight
        cand = [torch.cuda.Stream() for _ in range(K + 9)]
        cur = torch.cuda.current_stream()
        # every candidate runs one eager kernel before a graph is launched on it (no stream meets the graph runtime new;
        # the one unexplained crash of the calibration happened on a box's first process, in a replay on a never-used stream)
        tick = torch.zeros(1, device=self.x.device)
        for c in cand:
            c.wait_stream(cur)
            with torch.cuda.stream(c):
                tick.add_(1.0)
            cur.wait_stream(c)
        torch.cuda.synchronize()
        best, best_t = 0, None
        for off in range(len(cand) - K + 1):
            sts = cand[off:off + K]
            ts = []
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for n in range(2 * K):
                    st = sts[n % K]
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        graphs[n % K][0].replay()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            if best_t is None or min(ts) < best_t:
                best, best_t = off, min(ts)
        self.stream_window = (best, round(1e3 * best_t / (2 * K), 3))
        self._pipes = [[g, o, cand[best + k], torch.cuda.Event(), False] for k, (g, o) in enumerate(graphs)]
        self._issued = 0
        return True

    def _run_pipelined(self):
        if self._pipes is None and not self._pipeline_setup():
            return self.run()
        K, n = self.in_flight, self._issued
        cur = torch.cuda.current_stream()
        g, out, st, ev, _ = self._pipes[n % K]
        st.wait_stream(cur)                            # the consumer of the buffer this replay overwrites
        with torch.cuda.stream(st):
            g.replay()
            ev.record(st)
        self._pipes[n % K][4] = True
        self._issued = n + 1
        self._calls += 1
        oldest = self._pipes[(n + 1) % K]
        if not oldest[4]:
            return None
        cur.wait_event(oldest[3])
        self._last = self.out = oldest[1]
        return oldest[1]

    def run(self):
        if self.in_flight > 1 and not self.gather and self.use_graph:
            if self._calls >= 1:
                return self._run_pipelined()
            with whole_propagation(self.net):          # the first (eager) step returns what the pipelines will return
                self._calls += 1
                self.out = self._last = self._forward()
                return self.out
        if self.use_graph and self.graph is None and self._calls >= 1:
            self._capture()
        self._calls += 1
        if self.graph is not None:
            self.graph.replay()
        else:
            self.out = self._forward()
            if not self._synced:
                self._sync_decisions()
        if not self.gather:
            self._last = self.out
            return self.out
        return self._gather_pipelined(self.out)

    def _gather_pipelined(self, out):
        import torch.distributed as dist
        if self._stage is None:
            self._stage = [torch.empty_like(out) for _ in range(2)]
            self._gathered = [torch.empty((self.world * out.shape[0],) + tuple(out.shape[1:]), dtype=out.dtype,
                                          device=out.device) for _ in range(2)]
        k = self._calls & 1
        prev, self._pending = self._pending, None
        if prev is not None:
            prev[0].wait()                  # step k-1's gather (it had a whole forward to finish): frees stage / gather [k^1]...
        # ... and stage[k] / gathered[k] were released when step k-2's gather was waited for, one call ago
        self._stage[k].copy_(out)           # the forward's output buffer is free for the next replay after this copy
        work = dist.all_gather_into_tensor(self._gathered[k], self._stage[k], group=self.group, async_op=True)
        self._pending = (work, self._gathered[k])
        if prev is None:
            return None
        self._last = prev[1]
        return prev[1]

    def finish(self):
        """Frames of the last step (all ranks' frames when gathering); waits for the gather / the pipelines in flight."""
        if self._pipes is not None and self._issued:
            cur = torch.cuda.current_stream()
            for p in self._pipes:
                if p[4]:
                    cur.wait_event(p[3])
            self._last = self.out = self._pipes[(self._issued - 1) % self.in_flight][1]
            return self._last
        if self._pending is not None:
            work, buf = self._pending
            work.wait()
            self._pending = None
            self._last = buf
        return self._last


def dominant_kernel_probe(net, dev, iters=20):
    """Device time of the dominant kernel on its heaviest single launch of the north-star clip: encoder layer 10 (640 -> 512
    in 2 groups, 3x3, 10 frames of 60x108), hip events on the launch stream.  Whatever kernel the layer's tile decision names
    is what is timed and described (round 2: the fp32 Winograd F(2x4,3x3) kernel; round 3: the Winograd F(2x2,3x3) kernel
    with exactly split operands on the bf16 matrix pipe where it is faster).  `achieved` / `frac` count the work ISSUED to the
    matrix pipe in fp32-MFMA equivalents (a bf16 MAC of a split-operand kernel occupies the pipe for 157.3 / 2500 of the time
    of an fp32 MAC, so `frac` is the fraction of the time the pipe is busy at its peak rate in either case);
    `achieved_algorithmic` counts the direct-convolution FLOPs; `frac_effective` prices them against 2500 / 6 TF (split operands)
    or 157.3 TF (fp32 MFMA)."""
    from . import lib, ops
    eng = net.engine()
    layer = eng.enc[5]
    x0 = torch.randn(10, 60, 108, 256, device=dev)
    x1 = torch.randn(10, 60, 108, 384, device=dev)
    out = layer([x0, x1], act=ops.ACT_LRELU, slope=0.2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        layer([x0, x1], out=out, act=ops.ACT_LRELU, slope=0.2)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    saved, lib.TRACE = lib.TRACE, []
    try:                                                   # the trace record of the launch: kernel, algorithmic / issued MACs
        layer([x0, x1], out=out, act=ops.ACT_LRELU, slope=0.2)
        torch.cuda.synchronize()
        wk = [r["meta"] for r in lib.TRACE if r["meta"] and "macs" in r["meta"]][-1]
    finally:
        lib.TRACE = saved
    gflop, gflop_iss = 2e-9 * wk["macs"], 2e-9 * wk["issued"]
    gflop_use = 2e-9 * wk.get("useful", min(wk["macs"], wk["issued"]))
    tf, tf_iss = gflop / (us * 1e-6) / 1e3, gflop_iss / (us * 1e-6) / 1e3
    eff_peak = 2500.0 / 6 if "x3" in wk["kernel"] else 157.3
    rec = {"kernel": "%s (encoder.layers.10: 3x3 640->512 g2 on 10x60x108)" % wk["kernel"], "avg_us": round(us, 2),
           "gflop_per_launch": round(gflop_iss, 3), "gflop_per_launch_algorithmic": round(gflop, 3),
           "achieved": round(tf_iss, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf_iss / 157.3, 4),
           "frac_useful": round(gflop_use / (us * 1e-6) / 1e3 / 157.3, 4),
           "achieved_algorithmic": round(tf, 2), "effective_peak": round(eff_peak, 1), "frac_effective": round(tf / eff_peak, 4),
           "note": "achieved = work issued to the matrix pipe / time, in fp32-MFMA equivalents (frac_useful: without the MFMAs of the "
                   "tile padding -- 64x112-pixel blocks over 60x108 frames); achieved_algorithmic = direct-convolution FLOPs / time, "
                   "frac_effective = that against the rate the pipe retires fp32 products in this kernel's arithmetic (2500 / 6 TF as "
                   "six bf16 terms, 157.3 TF as fp32 MFMA)"}
    if "x3" in wk["kernel"]:
        rec["bf16_mfma"] = {"achieved": round(tf_iss * 2500.0 / 157.3, 1), "peak": 2500.0, "unit": "TFLOP/s",
                            "note": "the same launch counted as what it issues: six v_mfma_f32_32x32x16_bf16 terms per product of "
                                    "exactly split fp32 operands"}
    return rec
