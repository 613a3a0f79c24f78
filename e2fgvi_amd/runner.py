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
    """All-gather equal-sized per-rank outputs [n,3,H,W] into [world*n,3,H,W] (rank-major = clip order)."""
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


class ShardedStep:
    """One benchmark / serving step: forward of this rank's clips (+ all-gather)."""

    def __init__(self, net, x, lt, group_world=1, use_graph=True, force_gather=False):
        self.net, self.x, self.lt, self.world = net, x, lt, group_world
        self.force_gather = force_gather
        self.graph = None
        self.graphed = False
        self.out = None
        self.gathered = None
        self.use_graph = use_graph
        self._calls = 0

    def _forward(self):
        out, _ = self.net(self.x, self.lt)
        return out

    def run(self):
        if self.use_graph and self.graph is None and self._calls >= 1:
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._forward()                       # warm allocator on the side stream
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.out = self._forward()
                self.graph, self.graphed = g, True
            except Exception as e:                         # capture unsupported -> stay eager, say so
                print("HIP graph capture failed (%s); running eagerly" % (str(e).splitlines()[0],), flush=True)
                self.use_graph = False
                torch.cuda.synchronize()
        self._calls += 1
        if self.graph is not None:
            self.graph.replay()
            out = self.out
        else:
            out = self._forward()
        if self.world > 1 or self.force_gather:
            return self._gather_pipelined(out)
        return out

    def _gather_pipelined(self, out):
        """All-gather of this step's frames on RCCL's stream while the NEXT step's forward runs: the call returns the
        gathered frames of the previous step (None on the first call); `finish()` returns the last ones.  Two gather
        buffers alternate; the forward's output tensor is kept alive until its gather has completed."""
        import torch.distributed as dist
        if self.graph is not None:                         # static output buffer: it would be overwritten under the gather
            if self.gathered is None:
                self.gathered = [torch.empty((self.world * out.shape[0],) + tuple(out.shape[1:]), dtype=out.dtype,
                                             device=out.device)]
            gather_frames(out, self.world, out=self.gathered[0], force=self.force_gather)
            return self.gathered[0]
        if self.gathered is None:
            self.gathered = [torch.empty((self.world * out.shape[0],) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
                             for _ in range(2)]
            self._pending = None
        buf = self.gathered[self._calls & 1]
        work = dist.all_gather_into_tensor(buf, out.contiguous(), async_op=True)
        prev, self._pending = self._pending, (work, out, buf)
        if prev is None:
            return None
        prev[0].wait()                                     # stream-level wait; that gather finished a whole forward ago
        return prev[2]

    def finish(self):
        """Frames of the last step (waits for its gather)."""
        if getattr(self, "_pending", None) is None:
            return self.out if self.graph is not None else None
        work, _, buf = self._pending
        work.wait()
        self._pending = None
        return buf


def dominant_kernel_probe(net, dev, iters=20):
    """Device time of the dominant kernel (conv_wino_kernel, fp32 Winograd F(2x2,3x3) on the fp32 MFMA pipe) on its
    heaviest single launch of the north-star clip: encoder layer 10 (640 -> 512 in 2 groups, 3x3, 10 frames of 60x108),
    hip events on the launch stream.  `achieved` is ALGORITHMIC (direct-convolution) FLOP/s, so it can exceed the MFMA
    peak: the kernel executes 16/36 of those multiplies (x the block padding)."""
    from . import ops
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
    gflop = 2 * 10 * 60 * 108 * 9 * 320 * 512 * 1e-9
    tf = gflop / (us * 1e-6) / 1e3
    wino = layer.algo == "auto"
    # multiplies actually issued: 16 per 2x2 outputs and input channel instead of 36, on 64x112 padded pixels per frame
    executed = tf * (16.0 / 36.0) * (64 * 112) / (60 * 108) if wino else tf
    name = ("conv_wino_kernel<2,64,2> (Winograd F(2x2,3x3), encoder.layers.10: 3x3 640->512 g2 on 10x60x108)" if wino else
            "conv_igemm_kernel (encoder.layers.10: 3x3 640->512 g2 on 10x60x108)")
    return {"kernel": name, "avg_us": round(us, 2), "gflop_per_launch": round(gflop, 3), "achieved": round(tf, 2),
            "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
            "mfma_executed": round(executed, 2), "mfma_frac": round(executed / 157.3, 4),
            "note": "achieved = algorithmic (direct conv) FLOPs / time; Winograd issues 16/36 of them, mfma_executed is what "
                    "the matrix pipe actually did"}
