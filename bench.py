#!/usr/bin/env python
"""Headline benchmark: inpainted frames/sec at 432x240, T=10 (BASELINE.json), fp32, on N MI355X.

One "step" = one InpaintGenerator forward over this rank's batch of synthetic clips (all frames local, l_t = t: the
configuration that maximises propagation work), inputs already resident in HBM.

    N = 1 (default)  BASELINE.json configs[1]: e2fgvi 432x240 T=10, ONE clip per forward, fp32.
    N > 1 (default)  BASELINE.json configs[2]: the same clip, 8 clips per GPU per forward (64 at N=8), clips sharded
                     over ranks (one process per GPU), every step ends with the RCCL all-gather of the output frames
                     (uint8, the form test.py saves) over xGMI; per-GPU work is fixed (weak scaling).  Note for anyone
                     computing a scaling efficiency: the like-for-like single-GPU number of config 3 is
                     `--gpus 1 --clips-per-gpu 8`, not the 1-clip default of N = 1.
    --model e2fgvi_hq --hw 720x1296 --precision bf16      BASELINE.json configs[3] (and [4] with --hw 1080x1944 --t 20)

Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus 8 --steps 20 --warmup 3
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

PEAK_TFLOPS = {"fp32": 157.3,     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 2.4 GHz (= the fp32 vector peak)
               "bf16": 2500.0}    # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16); AMD's 5 PF figure includes 2:1 sparsity


def flops_per_clip(H, W, t, lt, hq):
    """analytic GFLOP (2 x MAC of conv / linear / matmul) per clip, formulas of SURVEY.md 8(d)"""
    p4, p2 = (H // 4) * (W // 4), (H // 2) * (W // 2)
    fh, fw = (H // 4 + 6 - 7) // 3 + 1, (W // 4 + 6 - 7) // 3 + 1
    n = fh * fw
    nw = n // 45
    hu, wu = -(-(H // 4) // 32) * 32, -(-(W // 4) // 32) * 32
    enc = t * (p2 * (64 * 27 + 64 * 576) + p4 * (128 * 576 + 256 * 1152 + 384 * 2304 + 512 * 2880 + 384 * 1728 + 256 * 720 + 128 * 4608))
    spy = 2 * (lt - 1) * 239904 * sum(hu * wu // 4 ** l for l in range(6))
    off = 2 * (lt - 1) * p4 * 9 * (388 * 128 + 2 * 128 * 128 + 128 * 432)
    dcn = 2 * (lt - 1) * p4 * 128 * 2304
    bb = lt * p4 * 9 * (256 * 128 + 384 * 128 + 2 * 128 * 128)
    fus = lt * p4 * 128 * 256
    ss = 2 * t * n * 6272 * 512 + (t * p4 * 128 * 1152 if hq else 0)
    blk = 8 * (t * n * 512 * 2048 + nw * t * (512 * 1536 + 45 * 512) + nw * 4 * (45 * t) * (210 * t) * 128 * 2 + t * n * 512 * 1960 * 2)
    dec = t * (p2 * 9 * (128 * 128 + 128 * 64) + 4 * p2 * 9 * (64 * 64 + 64 * 3))
    return 2e-9 * (enc + spy + off + dcn + bb + fus + ss + blk + dec)


KERNELS = {}      # layer -> kernel of the last traced forward (bench line: config.kernels)
USEFUL_GFLOP = 0.0   # of the last traced forward: issued work without tile / channel padding (roofline.frac_useful)
# the roof the reference's direct-convolution FLOPs are priced against (`frac_effective`): an fp32 product costs the matrix pipe one
# fp32 MFMA MAC (157.3 TF), or six bf16 MFMA MACs on exactly split operands (2500 / 6 = 416.7 TF), or one bf16 MAC on the bf16 path
EFFECTIVE_PEAK = {"fp32": 157.3, "fp32+x3": 2500.0 / 6, "bf16": 2500.0}


def launch_command(n, argv, script=None, port=None):
    """argv of the one-process-per-GPU launch of this script on ONE node: the driver's own form
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`)."""
    if port is None:
        import socket
        with socket.socket() as sk:                       # a free port, so two jobs on one box do not collide
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv, script=None, env=None):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks here and hand their exit code back.
    Rank 0 prints the JSON line on the inherited stdout (the other ranks' stdout goes to /dev/null in main())."""
    import subprocess
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL across processes)
    e.setdefault("OMP_NUM_THREADS", "8")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    sys.stdout.flush()
    return subprocess.call(launch_command(n, argv, script), env=e)


def traced_work(net, x, lt):
    """One eager forward with launch tracing on: sums of the algorithmic MACs (what the reference's layers compute) and of
    the MACs actually ISSUED to the matrix pipe (Winograd layers issue 16/36 of theirs, x block / channel padding) over
    every conv / linear / deformable-conv / attention launch.  Side effect: KERNELS = {layer: kernel that ran} of that forward
    (the eight transformer blocks and the two propagation directions collapsed when they agree)."""
    import re
    from e2fgvi_amd import lib
    lib.TRACE = []
    try:
        net(x, lt)
        torch.cuda.synchronize()
        rows = [r["meta"] for r in lib.TRACE if r["meta"] and "macs" in r["meta"]]
    finally:
        lib.TRACE = None
    names = {}
    for r in rows:
        layer = re.sub(r"^transformer\.\d+\.", "transformer.*.", str(r.get("layer", "?")))
        layer = re.sub(r"(backward_|forward_)", "*_", layer)
        names.setdefault(layer, set()).add(str(r.get("kernel", "?")))
    KERNELS.clear()
    KERNELS.update({k: " | ".join(sorted(v)) for k, v in sorted(names.items()) if not k.startswith("spynet.")})
    KERNELS["spynet.*"] = " | ".join(sorted({kk for k, v in names.items() if k.startswith("spynet.") for kk in v}))
    global USEFUL_GFLOP
    USEFUL_GFLOP = 2e-9 * sum(r.get("useful", min(r["macs"], r["issued"])) for r in rows)
    return 2e-9 * sum(r["macs"] for r in rows), 2e-9 * sum(r["issued"] for r in rows), len(rows)


def _parity(hip_out, ref, what):
    d = (hip_out.double() - ref.double()).abs()
    rms = float(ref.double().pow(2).mean().sqrt())
    return {"max_abs": float("%.3e" % d.max()), "max_abs_over_rms": float("%.3e" % (float(d.max()) / rms)),
            "rms_of_reference": float("%.3e" % rms), "weights": what}


def cpu_baseline(sd, model, H, W, t, lt, hip_out=None, stress=None):
    """The CPU restatement of the reference forward (oracle/e2fgvi_oracle.py, kind "port": the reference's own Python
    cannot travel to the GPU box) timed on this host's cores on a bounded sample of the same workload: ONE clip,
    1 warm-up + median of 3 forwards for the 432x240 workload; the HQ resolutions take minutes per
    clip on a CPU, there the sample is one un-warmed forward of a 2-frame clip of the same resolution.
    hip_out: the frames the TIMED configuration (same clip, same weights, the timed kernels) produced -- the oracle's output of
    the first forward is compared with them and returned as the second value: the bench line carries its own parity.
    stress: {kind: (state_dict, hip frames)} of the same clip under the stress weights of SURVEY.md 8(c) T3 (O(1) activations, non-zero
    conv_offset[-1]: real DCN offsets and masks) and, round 6, the peaked weights (synth.py: sharp attention, saturated offsets) --
    one more oracle forward each, compared the same way (round 5: at default init conv_offset[-1] == 0 makes offsets = flow and
    mask = 0.5, a regime in which a 1e-3 absolute bound is weak)."""
    from e2fgvi_amd.synth import synth_clip
    from oracle import e2fgvi_oracle as O
    host = os.cpu_count() or 1
    # torch's intra-op pool stops scaling on these small ops (with the GPU box's 256 hardware threads it collapses to
    # minutes per forward): 16 threads is the fastest setting measured there
    cores = max(1, min(host, 16))
    torch.set_num_threads(cores)
    small = (H, W) == (240, 432)
    ts, ls = (t, lt) if small else (2, 2)
    x, _ = synth_clip(1, ts, H, W, seed=0, smooth=False)
    times = []
    parity = None
    for k in range(4 if small else 1):
        t0 = time.perf_counter()
        ref, _ = O.forward(sd, x, ls, model)
        times.append(time.perf_counter() - t0)
        if k == 0 and hip_out is not None and tuple(hip_out.shape) == tuple(ref.shape):
            parity = {"default": _parity(hip_out, ref, "reference's init_weights distribution (the timed weights)")}
    what = {"stress": "stress weights (synth_state_dict 'stress': O(1) activations, non-zero conv_offset[-1], random sc.bias; "
                      "SURVEY.md 8c T3)",
            "peaked": "peaked weights (synth_state_dict 'peaked', the stand-in for trained weights: mean largest attention "
                      "probability 0.24-0.51, 46 % of the residual DCN offsets beyond +-9 px, saturated masks, non-uniform pool_layers, "
                      "flows up to 10 px)"}
    for kind, (sd_k, frames_k) in (stress or {}).items() if parity is not None else ():
        ref, _ = O.forward(sd_k, x, ls, model)
        if tuple(frames_k.shape) == tuple(ref.shape):
            parity[kind] = _parity(frames_k, ref, what[kind])
    if parity is not None:
        parity.update({"max_abs": max(v["max_abs"] for v in parity.values()),
                       "max_abs_over_rms": max(v["max_abs_over_rms"] for v in parity.values()), "bound_max_abs": 1e-3,
                       "vs": "oracle port (oracle/e2fgvi_oracle.py, torch CPU fp32) on the timed clip; the HIP frames are those of the "
                             "timed configuration (same engine class, same kernel decisions as config.kernels; `default` = the very "
                             "engine / HIP graph that was timed, `stress` / `peaked` = further engines on the same clip)"})
    timed = times[1:] if small else times
    dt = statistics.median(timed)
    return {"value": round(ts / dt, 4), "unit": "frames/s", "cores": cores, "host_cores": host, "kind": "port",
            "sample": "one %s %dx%d T=%d l_t=%d clip, torch CPU fp32 on %d threads (host has %d): %s; frames/s = %d frames / "
                      "that time.  A lower bound on the reference's own CPU speed: the port's deformable conv is a pure-torch gather, "
                      "the reference calls mmcv's C++ CPU op (mmcv is not in this image)"
                      % (model, W, H, ts, ls, cores, host,
                         "1 warm-up + median of 3 forwards (%.1f s each)" % dt if small else
                         "one un-warmed forward (%.1f s)" % dt, ts)}, parity


def time_local(net, x, lt, steps, warmup, use_graph=True, in_flight=1):
    """`steps` forwards of this GPU's clips replayed from a HIP graph, no collective: (wall seconds, device ms, graphed)"""
    from e2fgvi_amd import runner
    step = runner.ShardedStep(net, x, lt, group_world=1, use_graph=use_graph, in_flight=in_flight)
    for _ in range(max(warmup, 2 if in_flight == 1 else 2 + in_flight)):       # the second call captures the graph(s)
        step.run()
    step.finish()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step.run()
    step.finish()
    ev1.record()
    torch.cuda.synchronize()
    wall, dev_ms = time.perf_counter() - t0, ev0.elapsed_time(ev1)
    LAST_FRAMES[0] = step.finish()          # the frames of the last timed step (bf16 lines: compared with the real reference's fixture)
    return wall, dev_ms, bool(step.graphed)


LAST_FRAMES = [None]
GOLDEN = os.path.join(ROOT, "tests", "golden")
# fixtures of the REAL reference (tests/golden/make_golden.py) for the bf16 lines: (H, W, T) -> (the bench clip under the timed
# default-init weights | None, a stress-weights clip of the same resolution)
HQ_FIXTURES = {(720, 1296, 10): ("g14_hq_default_720x1296_t10_lt10_benchclip.npz", "g9_hq_stress_720x1296_t10_lt10.npz"),
               (1080, 1944, 20): (None, "g10_hq_stress_1080x1944_t8_lt8.npz")}


def _fixture_parity(frames, path, what):
    """frames [b*t, 3, H, W] of the HIP path against the strided sub-sample a fixture holds of the real reference's output"""
    import numpy as np
    z = np.load(path)
    so = int(z["meta"][6])
    diff = frames[:, :, ::so, ::so].float().cpu().numpy().astype(np.float64) - z["out_sub"]
    rms = float(z["out_stats"][2])
    return {"max_abs": float("%.3e" % np.abs(diff).max()), "rms_of_difference_over_rms": float("%.3e" % (np.sqrt((diff ** 2).mean()) / rms)),
            "rms_of_reference": float("%.3e" % rms), "fixture": "tests/golden/" + os.path.basename(path), "weights": what}


def hq_parity(dev, model, H, W, t, precision, timed_frames):
    """`parity` of a bf16 secondary line, against the REAL reference (fixtures made in the build container; the reference cannot
    travel): `default` = the frames of the timed engine on the timed clip (720x1296 T=10: the fixture IS bench.py's clip),
    `stress` = a second engine of the same class and kernel decisions on the stress-weights fixture clip of that resolution
    (T = l_t = 10 at 720x1296, T = l_t = 8 at 1080x1944: the reference's materialised attention scores at T = 20 need > 60 GB)."""
    import importlib
    import numpy as np
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    fx = HQ_FIXTURES.get((H, W, t))
    if fx is None or not all(f is None or os.path.exists(os.path.join(GOLDEN, f)) for f in fx):
        return None
    par = {}
    if fx[0] is not None and timed_frames is not None:
        par["default"] = _fixture_parity(timed_frames, os.path.join(GOLDEN, fx[0]), "reference's init_weights distribution (the timed weights, "
                                         "the timed clip, the timed engine's last step)")
    z = np.load(os.path.join(GOLDEN, fx[1]))
    fh, fw, ft, flt, fb, seed = [int(v) for v in z["meta"][:6]]
    net = importlib.import_module("model." + model).InpaintGenerator()
    net.load_state_dict(synth_state_dict(model, "stress", 0))
    net = net.to(dev).eval()
    net.precision = precision
    xs = synth_clip(fb, ft, fh, fw, seed=seed, moving=True)[0].to(dev)
    out = net(xs, flt)[0]
    torch.cuda.synchronize()
    par["stress"] = _fixture_parity(out, os.path.join(GOLDEN, fx[1]), "stress weights, %dx%d T=%d l_t=%d (a second engine, same kernel table)"
                                    % (fw, fh, ft, flt))
    del net, xs, out
    par.update({"max_abs": max(v["max_abs"] for v in par.values()),
                "rms_of_difference_over_rms": max(v["rms_of_difference_over_rms"] for v in par.values()),
                "bound_max_abs": 1e-2, "bound_rms_of_difference_over_rms": 2e-2,
                "vs": "the REAL reference's CPU forward (sub-sampled fixtures, tests/golden/make_golden.py); bf16 data path: the bound is "
                      "DESIGN.md section 4's (not the fp32 contract's 1e-3)"})
    return par


ARITHMETIC = {
    "fp32+x3": "fp32 tensors; contractions at fp32-level rounding: per layer either fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-equivalent to "
               "an fp32 FMA chain) or the bf16 matrix pipe with EXACTLY split operands (3 bf16 pieces per fp32 value, 6 of the 9 cross "
               "terms as v_mfma_f32_32x32x16_bf16, fp32 accumulate; dropped terms < 2^-22 of a product): fp32-level, not bit-identical "
               "to an FMA chain (tests/test_gpu_x3.py); which layer runs which: config.kernels",
    "fp32": "fp32 tensors, every contraction on fp32 MFMA (v_mfma_f32_32x32x2_f32): bit-equivalent to an fp32 FMA chain (E2FGVI_X3=0)",
    "bf16": "bf16 tensors between kernels, bf16 MFMA (v_mfma_f32_32x32x16_bf16) with fp32 accumulation; flows, DCN offsets / masks, the "
            "token residual stream and the output frames stay fp32 (DESIGN.md 1b)",
}

SECONDARY = [   # (label, model, H, W, clips, t, l_t, precision, x3, steps, warmup, forwards in flight)
    ("BASELINE.json configs[1] with E2FGVI_X3=0: every fp32 layer on fp32 MFMA (no split-operand kernels)", "e2fgvi", 240, 432, 1, 10, 10, "fp32", False, 10, 4, 2),
    ("SURVEY.md 8(d) C2 second split: T=10 with 5 local + 5 reference frames (configs/train_e2fgvi.json:9-10)", "e2fgvi", 240, 432, 1, 10, 5, "fp32", True, 10, 4, 2),
    ("BASELINE.json configs[2] per-GPU work on ONE GPU: 8 clips per forward, no collective (the N = 1 point of the 8-GPU job runs one "
     "forward at a time: its `sequential`)", "e2fgvi", 240, 432, 8, 10, 10, "fp32", True, 6, 4, 2),
    ("BASELINE.json configs[1] with THREE forwards in flight (the headline keeps two in flight)", "e2fgvi", 240, 432, 1, 10, 10, "fp32", True, 30, 6, 3),
    ("BASELINE.json configs[3]", "e2fgvi_hq", 720, 1296, 1, 10, 10, "bf16", True, 20, 4, 2),
    ("BASELINE.json configs[4] (one GPU's clip)", "e2fgvi_hq", 1080, 1944, 1, 20, 20, "bf16", True, 6, 4, 2),
]


def secondary_line(dev, label, model, H, W, b, t, lt, precision, x3, steps, warmup, in_flight=1):
    """One more configuration timed in the same process after the headline (inputs resident, HIP-graph replay)."""
    import gc
    import importlib
    from e2fgvi_amd import engine, ops
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    saved = (ops.X3_ENABLED, engine.FC2_CONV)
    if precision == "fp32" and not x3:
        ops.X3_ENABLED, engine.FC2_CONV = False, False
    try:
        net = importlib.import_module("model." + model).InpaintGenerator()
        net.load_state_dict(synth_state_dict(model, "default", 0))
        net = net.to(dev).eval()
        net.precision = precision
        x = synth_clip(b, t, H, W, seed=0, smooth=False)[0].to(dev)
        torch.cuda.reset_peak_memory_stats(dev)
        net(x, lt)
        torch.cuda.synchronize()
        from e2fgvi_amd import runner
        if in_flight > 1:
            with runner.whole_propagation(net):
                gflop_alg, gflop_issued, nlaunch = traced_work(net, x, lt)
        else:
            gflop_alg, gflop_issued, nlaunch = traced_work(net, x, lt)
        gflop_useful = USEFUL_GFLOP
        kernels = dict(KERNELS)
        sequential = None
        if in_flight > 1:
            sq_steps = max(3, min(steps, 10))
            el, _, _ = time_local(net, x, lt, sq_steps, 2)
            sequential = {"value": round(b * t * sq_steps / el, 3), "unit": "frames/s", "ms_per_step": round(1e3 * el / sq_steps, 3),
                          "steps": sq_steps, "forwards_in_flight": 1}
        elapsed, dev_ms, graphed = time_local(net, x, lt, steps, warmup, in_flight=in_flight)
        timed_frames, LAST_FRAMES[0] = LAST_FRAMES[0], None
    finally:
        ops.X3_ENABLED, engine.FC2_CONV = saved
    secs = dev_ms * 1e-3 / steps
    peak = PEAK_TFLOPS[precision]
    arith = "bf16" if precision == "bf16" else ("fp32+x3" if x3 and saved[0] else "fp32")
    line = {"config": {"workload": "%s: %s %dx%d T=%d l_t=%d, %d clip(s) per forward on one GPU, random-init weights, torch.rand "
                                   "frames + box mask (SURVEY.md 8d)" % (label, model, W, H, t, lt, b),
                       "precision": precision, "arithmetic": ARITHMETIC[arith], "hip_graph": graphed, "forwards_in_flight": in_flight},
            "metric": "inpainted frames/sec at %dx%d T=%d" % (W, H, t), "value": round(b * t * steps / elapsed, 3), "unit": "frames/s",
            "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 3),
            "dtype": "f32" if precision == "fp32" else "bf16",
            "roofline": {"bound": "mfma", "achieved": round(gflop_issued / secs / 1e3, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(gflop_issued / secs / 1e3 / peak, 4),
                         "frac_useful": round(gflop_useful / secs / 1e3 / peak, 4),
                         "achieved_algorithmic": round(gflop_alg / secs / 1e3, 2),
                         "effective_peak": round(EFFECTIVE_PEAK[arith], 1),
                         "frac_effective": round(gflop_alg / secs / 1e3 / EFFECTIVE_PEAK[arith], 4),
                         "gflop_per_forward": {"algorithmic": round(gflop_alg, 1), "issued": round(gflop_issued, 1),
                                               "useful": round(gflop_useful, 1), "mfma_launches": nlaunch}},
            "peak_memory_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)}
    if precision == "fp32" and not x3:
        line["config"]["kernels"] = kernels
    if sequential is not None:
        line["sequential"] = sequential
    del net, x
    gc.collect()
    torch.cuda.empty_cache()
    if precision == "bf16":
        par = hq_parity(dev, model, H, W, t, precision, timed_frames)
        if par is not None:
            line["parity"] = par
        del timed_frames
        gc.collect()
        torch.cuda.empty_cache()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips-per-gpu", type=int, default=0,
                    help="clips per forward on each GPU; default 1 at N=1 (BASELINE config 2), 8 at N>1 (config 3)")
    ap.add_argument("--model", default="e2fgvi", choices=("e2fgvi", "e2fgvi_hq"))
    ap.add_argument("--hw", default="240x432", help="HxW of the (already padded) clip; the base model is fixed to 240x432")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16"))
    ap.add_argument("--t", type=int, default=10)
    ap.add_argument("--lt", type=int, default=0, help="local frames (default: all t)")
    ap.add_argument("--gather", default="u8", choices=("u8", "f32"), help="dtype of the frames in the all-gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="disable HIP-graph replay of the forward")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="forwards in flight on one GPU (runner.ShardedStep(in_flight=K): K HIP graphs of the same one-clip forward "
                         "replayed round-robin on K streams -- step n + 1 starts while step n is in its latency-bound propagation "
                         "chain; every step returns the bits of the sequential forward).  Default: 2 on a single GPU without a "
                         "gather (the line then also carries `sequential`: one graph replayed back to back, rounds 1-5's headline "
                         "mode, timed in the same process), 1 with a gather (N > 1 / --force-dist)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gather even with one rank (self-test)")
    ap.add_argument("--no-dominant-probe", action="store_true",
                    help="skip the 21 extra launches of encoder.layers.10 behind the timed region (PMC passes count whole processes: "
                         "tools/pmc.sh divides by the number of forwards)")
    ap.add_argument("--launcher", action="store_true",
                    help="start the ranks through torch.distributed.run even for one GPU (the N > 1 launch path, testable on one GPU)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the BASELINE configs[3] / [4] lines (e2fgvi_hq 720x1296 T=10 and 1080x1944 T=20, bf16) that the "
                         "default single-GPU invocation times after the headline and attaches as `secondary`")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "WORLD_SIZE" in os.environ
    if launched and args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not launched and (args.gpus > 1 or args.launcher):
        # plain `python bench.py --gpus N`: become the launcher of the N ranks (round 6; before, this form exited with a hint)
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("--gpus %d but %d GPU(s) visible on this node" % (args.gpus, have))
        argv = [a for a in sys.argv[1:] if a != "--launcher"]
        if args.gpus == 1 and "--force-dist" not in argv:
            argv.append("--force-dist")
        raise SystemExit(self_launch(args.gpus, argv))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import importlib
    from e2fgvi_amd import ops, runner
    from e2fgvi_amd.synth import synth_clip, synth_state_dict

    H, W = [int(v) for v in args.hw.lower().split("x")]
    if H > W:                                            # tolerate WxH
        H, W = W, H
    t = args.t
    lt = args.lt or t
    b = args.clips_per_gpu or (1 if world == 1 else 8)
    hq = args.model == "e2fgvi_hq"
    sd = synth_state_dict(args.model, "default", 0)        # random-init distribution of the reference
    net = importlib.import_module("model." + args.model).InpaintGenerator()
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    net.precision = args.precision
    # frames = torch.rand(b,t,3,H,W, generator=seed) * 2 - 1, box mask [H/4:H/2, W/4:W/2]
    x, _ = synth_clip(b, t, H, W, seed=rank, smooth=False)
    x = x.to(dev)
    # Build the engine (weight re-layout, tile tuning, stream creation) BEFORE RCCL comes up: measured on MI355X, a
    # forward whose engine was built after init_process_group runs ~4 % slower (17.8 vs 17.15 ms)
    torch.cuda.reset_peak_memory_stats(dev)
    net(x, lt)
    torch.cuda.synchronize()
    want_in_flight = 1 if (world > 1 or args.force_dist or args.no_graph) else (args.in_flight or 2)
    if want_in_flight > 1:
        with runner.whole_propagation(net):                         # what the pipelines' graphs run (runner.ShardedStep._pipeline_setup)
            gflop_alg, gflop_issued, nlaunch = traced_work(net, x, lt)
    else:
        gflop_alg, gflop_issued, nlaunch = traced_work(net, x, lt)      # per forward of b clips
    gflop_useful = USEFUL_GFLOP
    kernels = dict(KERNELS)
    same_work = None
    if world > 1 or args.force_dist:
        # the like-for-like single-GPU number of this job's per-GPU work: the same clips, the same HIP-graph replay, no
        # collective, timed on this GPU before RCCL comes up (a scaling efficiency can then be read off ONE record)
        sw_steps = max(3, min(args.steps, 10))
        el, _, _ = time_local(net, x, lt, sw_steps, 2, use_graph=not args.no_graph)
        same_work = {"value": round(b * t * sw_steps / el, 3), "unit": "frames/s", "ms_per_step": round(1e3 * el / sw_steps, 3),
                     "steps": sw_steps, "note": "rank 0's GPU alone on its own share of the job (%d clips per forward, graph "
                                                "replay, no collective), timed before init_process_group: the N = 1 point of "
                                                "this workload; efficiency at N = value / (N x this)" % b}

    dist = None
    if world > 1 or args.force_dist:
        if rank != 0:
            # only rank 0 reports: keep the other ranks' stdout (RCCL prints its version banner there when NCCL_DEBUG is set)
            # from landing after the JSON line
            sys.stdout.flush()
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    # the forward replays from a HIP graph in every mode; the collective stays outside the graph (runner.ShardedStep)
    in_flight = 1 if (dist is not None or args.no_graph) else (args.in_flight or 2)
    sequential = None
    if in_flight > 1:
        # rounds 1-5's headline mode on this box, in this process: ONE graph replayed back to back
        sq_steps = max(5, min(args.steps, 20))
        el, _, _ = time_local(net, x, lt, sq_steps, 3)
        LAST_FRAMES[0] = None
        sequential = {"value": round(b * t * sq_steps / el, 3), "unit": "frames/s", "ms_per_step": round(1e3 * el / sq_steps, 3),
                      "steps": sq_steps, "forwards_in_flight": 1,
                      "note": "the same forward, one HIP graph replayed back to back on one stream (the headline mode of rounds 1-5; "
                              "`ms_per_step` here is also the latency of a forward that has the GPU to itself)"}
    step = runner.ShardedStep(net, x, lt, group_world=world, use_graph=not args.no_graph, force_gather=args.force_dist,
                              pack_u8=(dist is not None and args.gather == "u8"), in_flight=in_flight)
    for _ in range(max(args.warmup, 2 if step.in_flight == 1 else 2 + step.in_flight)):   # the second untimed step captures the HIP graph(s)
        step.run()
    step.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step.run()
    step.finish()                         # the last step's (pipelined) all-gather joins the timed region
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    frames = world * b * t * args.steps
    value = frames / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    peak = PEAK_TFLOPS[args.precision]
    analytic = b * flops_per_clip(H, W, t, lt, hq)
    secs = dev_ms * 1e-3 / args.steps
    tf_alg, tf_iss = gflop_alg / secs / 1e3, gflop_issued / secs / 1e3          # this rank, device-timed
    arith = "bf16" if args.precision == "bf16" else ("fp32+x3" if ops.X3_ENABLED else "fp32")
    eff_peak = EFFECTIVE_PEAK[arith]
    config_no = 2 if (world == 1 and b == 1) else 3
    if hq:
        config_no = 4 if t <= 10 else 5
    out = {
        "metric": "inpainted frames/sec at %dx%d T=%d" % (W, H, t), "value": round(value, 3), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[%d]: %s %dx%d T=%d l_t=%d, %d clip(s) per GPU per forward (%d in the job), "
                               "random-init weights, torch.rand frames + box mask (SURVEY.md 8d)"
                               % (config_no - 1, args.model, W, H, t, lt, b, b * world),
                   "clips_per_gpu": b, "precision": args.precision,
                   "arithmetic": ARITHMETIC[arith],
                   "kernels": kernels,
                   "kernel_selection": ("timed on this box (E2FGVI_AUTOTUNE=1)" if ops.AUTOTUNE else
                                        "e2fgvi_amd/tile_table.py (checked in, deterministic: ops.py)"),
                   "parallelism": "clip-shard x%d + all-gather of the %s frames" % (world, args.gather) if dist is not None
                                  else ("single GPU, no collective" + ("; %d forwards in flight (that many HIP graphs of the one-clip "
                                        "forward replayed round-robin on as many streams, bit-identical steps; `sequential` = one at a time)"
                                        % step.in_flight if step.in_flight > 1 else "")),
                   "hip_graph": bool(step.graphed), "forwards_in_flight": step.in_flight},
        "roofline": {"bound": "mfma", "achieved": round(tf_iss, 3), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(tf_iss / peak, 4), "frac_useful": round(gflop_useful / secs / 1e3 / peak, 4),
                     "achieved_algorithmic": round(tf_alg, 3), "effective_peak": round(eff_peak, 1),
                     "frac_effective": round(tf_alg / eff_peak, 4),
                     "gflop_per_forward": {"algorithmic": round(gflop_alg, 1), "issued": round(gflop_issued, 1),
                                           "useful": round(gflop_useful, 1),
                                           "analytic_survey_8d": round(analytic, 1), "mfma_launches": nlaunch},
                     "traffic": None,
                     "note": "whole forward, device time of one forward (hip events on the launch stream). `achieved` / `frac` count "
                             "the FLOPs ISSUED to the matrix pipe (the Winograd F(2x2,3x3) layers issue 16/36 of their direct-"
                             "convolution multiplies, plus block / channel padding: per-launch accounting of ops.PackedConv._work), "
                             "`frac_useful` the same without the tile / channel padding (lib.useful_macs).  fp32 layers that run on "
                             "the bf16 matrix pipe with exactly split operands (six bf16 MFMA terms per fp32 product, fp32-level "
                             "rounding: tests/test_gpu_x3.py) are counted in fp32-MFMA equivalents (bf16 MACs x 157.3 / 2500), so "
                             "`frac` is the share of the time the matrix pipe is busy at its peak rate (what rocprofv3's "
                             "SQ_VALU_MFMA_BUSY_CYCLES measures).  `achieved_algorithmic` counts the reference's direct-convolution "
                             "FLOPs (SURVEY.md 8d); `frac_effective` prices them against `effective_peak`, the rate at which the "
                             "matrix pipe can retire fp32 products in this arithmetic (157.3 TF as fp32 MFMA, 2500 / 6 = 416.7 TF as "
                             "six bf16 terms, 2500 TF on the bf16 data path) -- Winograd's saved multiplies count in its favour, "
                             "so it is an efficiency against the direct algorithm, not a pipe utilisation"},
    }
    # HBM-side traffic of one forward: rocprofv3 PMC passes (tools/pmc.sh) cannot run inside the timed process; the summary of the
    # last collection is committed under profiles/ TOGETHER WITH the key of the library it measured (lib.library_key()), and it is
    # quoted here only when that key is the key of the library this process has just timed -- a figure of an older library is
    # named in the note and NOT reported as this tree's traffic (round 4's line carried round 3's number)
    from e2fgvi_amd import lib as _lib
    lib_key = _lib.library_key()
    out["library_sha16"] = lib_key
    traffic_cfg = None
    if b == 1 and (t, lt) == (10, 10) and not hq and args.precision == "fp32":
        traffic_cfg = ""
    elif b == 1 and (t, lt) == (10, 10) and hq and (H, W) == (720, 1296) and args.precision == "bf16":
        traffic_cfg = "_hq720_bf16"
    elif b == 1 and (t, lt) == (20, 20) and hq and (H, W) == (1080, 1944) and args.precision == "bf16":
        traffic_cfg = "_hq1080_bf16"
    if traffic_cfg is not None:
        import glob
        stale = []
        for tfile in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic%s.json" % traffic_cfg)), reverse=True):
            try:
                tj = json.load(open(tfile))
            except Exception:
                continue
            if tj.get("library_sha16") != lib_key:
                stale.append(os.path.basename(tfile))
                continue
            out["roofline"]["traffic"] = round(tj["hbm_bytes_per_forward"])
            out["roofline"]["traffic_note"] = ("FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE per forward, rocprofv3 --pmc, separate "
                                               "passes (profiles/%s, measured on this library: sha16 %s); fabric-side, includes Infinity-"
                                               "Cache hits; compulsory minimum (input + weights + output) is %s GB/clip; counted on eager one-at-a-time forwards "
                                               "(PMC passes serialise the kernels), the same kernels the in-flight graphs replay"
                                               % (os.path.basename(tfile), lib_key, "0.19" if not hq else "0.39"))
            break
        else:
            out["roofline"]["traffic_note"] = ("no PMC collection of this library (sha16 %s) under profiles/: bash tools/pmc.sh <tag>; "
                                               "collections of other builds, not quoted: %s" % (lib_key, ", ".join(stale) or "none"))
    if rank == 0:
        if not hq and args.precision == "fp32" and not args.no_dominant_probe:
            dom = runner.dominant_kernel_probe(net, dev)
            dom["traffic"] = None
            # PMC traffic of the kernel that runs: the newest profiles/*dominant_kernel_traffic*.json whose kernel_tag is this
            # kernel's trace name (round 3's files carry it; older ones are the fp32 F(2x4) kernel's)
            import glob
            for dfile in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_dominant_kernel_traffic*.json")), reverse=True):
                try:
                    dj = json.load(open(dfile))
                    if dj.get("kernel_tag", "conv_wino4<F(2x4),64>") != dom["kernel"].split(" (")[0] or dj.get("library_sha16") != lib_key:
                        continue
                    dom["traffic"] = round(dj["hbm_bytes_per_launch"])
                    dom["traffic_note"] = "bytes per launch, PMC FETCH_SIZE x2 + WRITE_SIZE (profiles/%s)" % os.path.basename(dfile)
                    break
                except Exception:
                    pass
            out["roofline"]["dominant_kernel"] = dom
        out["peak_memory_gb"] = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)
        eng = getattr(net, "_engine", None)
        if eng is not None:
            # packed weights the layers hold after the timed run (each layer packs what the kernels it runs need) next to the checkpoint
            out["weight_memory_gb"] = {"packed": round(eng.weight_bytes() / 2 ** 30, 3),
                                       "checkpoint_fp32": round(sum(p.numel() * 4 for p in net.parameters()) / 2 ** 30, 3)}
        if world == 1 and not args.no_cpu_baseline:
            # the frames of the timed configuration (clip 0 of this rank = the oracle's clip: synth_clip seed 0), computed by the
            # very engine / kernel decisions / HIP graph that were timed
            hip_frames = step.finish()
            hip_frames = (hip_frames if hip_frames is not None else net(x, lt)[0])[:t].float().cpu() if not step.pack_u8 else None
            stress = None
            if hip_frames is not None and (H, W) == (240, 432):
                # the same clip under the stress and the peaked weights: a second / third engine, the same (table-driven) kernel decisions
                stress = {}
                for kind in ("stress", "peaked"):
                    sd_s = synth_state_dict(args.model, kind, 0)
                    net_s = importlib.import_module("model." + args.model).InpaintGenerator()
                    net_s.load_state_dict(sd_s)
                    net_s = net_s.to(dev).eval()
                    net_s.precision = args.precision
                    stress[kind] = (sd_s, net_s(x[:1], lt)[0][:t].float().cpu())
                    del net_s
            out["cpu_baseline"], parity = cpu_baseline(sd, args.model, H, W, t, lt, hip_out=hip_frames, stress=stress)
            if parity is not None:
                out["parity"] = parity
    if same_work is not None:
        out["single_gpu_same_work"] = same_work
    if sequential is not None:
        out["sequential"] = sequential
        out["config"]["stream_window"] = getattr(step, "stream_window", None)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    default_headline = (world == 1 and not args.force_dist and not hq and args.precision == "fp32" and b == 1 and (t, lt) == (10, 10))
    if rank == 0 and default_headline and not args.no_secondary:
        del step, net, x
        torch.cuda.empty_cache()
        out["secondary"] = []
        for cfg in SECONDARY:
            try:
                out["secondary"].append(secondary_line(dev, *cfg))
            except Exception as e:                    # the headline line must survive a failure here
                out["secondary"].append({"config": {"workload": "%s: %s %dx%d T=%d %s" % (cfg[0], cfg[1], cfg[3], cfg[2], cfg[5], cfg[7])},
                                         "error": str(e).splitlines()[0][:300]})
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)        # C stdio first (RCCL's banner), so that the JSON line is the last line on stdout
        print(json.dumps(out), flush=True)
        if out.get("parity") and not out["parity"]["max_abs"] <= out["parity"]["bound_max_abs"]:
            raise SystemExit("parity of the timed configuration: max |hip - oracle| = %g > %g"
                             % (out["parity"]["max_abs"], out["parity"]["bound_max_abs"]))


if __name__ == "__main__":
    main()
