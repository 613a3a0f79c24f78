"""bench.py's contract on the GPU box, as subprocesses (what the driver runs): the default single-GPU line carries the
BASELINE configs[3] / [4] lines as `secondary`; the N > 1 flow -- RCCL process group, HIP-graph replay, uint8 all-gather
pipelined under the next forward, 8 clips per GPU (BASELINE configs[2]'s per-GPU work) -- runs here with world size 1
(--force-dist) and reports the like-for-like single-GPU number of the same work."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29677", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:]
    return json.loads(lines[-1])


def _fracs(node, path=""):
    """every `frac*` number anywhere in a bench line"""
    if isinstance(node, dict):
        for k, v in node.items():
            if k.startswith("frac") and isinstance(v, (int, float)):
                yield path + "/" + k, v
            else:
                yield from _fracs(v, path + "/" + k)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from _fracs(v, "%s[%d]" % (path, i))


def test_plain_gpus_flag_launches_the_ranks_itself(dev):
    """`python bench.py --gpus N` with no launcher around it (the driver's 8-GPU form minus torch.distributed.run) must start its
    own ranks: --launcher takes the same path with one GPU (torch.distributed.run -> one rank, RCCL group, gather, JSON line)"""
    j = _bench("--gpus", "1", "--launcher", "--clips-per-gpu", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert j["n_gpus"] == 1 and j["config"]["clips_per_gpu"] == 2 and "all-gather" in j["config"]["parallelism"]
    assert j["single_gpu_same_work"]["value"] > 0 and j["value"] > 30


def test_multi_gpu_flow_with_one_rank(dev):
    """config 3's per-GPU flow: `bench.py --force-dist --clips-per-gpu 8`"""
    j = _bench("--force-dist", "--clips-per-gpu", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert j["n_gpus"] == 1 and j["unit"] == "frames/s" and j["scaling"] == "weak" and j["dtype"] == "f32"
    assert j["config"]["clips_per_gpu"] == 8 and j["config"]["hip_graph"] is True
    assert "all-gather of the u8 frames" in j["config"]["parallelism"]
    assert abs(j["value"] - 80 * 3 / (j["ms_per_step"] * 3e-3)) <= 1e-2 * j["value"]
    sw = j["single_gpu_same_work"]
    assert sw["unit"] == "frames/s" and sw["value"] > 0
    # one rank: the gather is a device copy that runs under the next forward -- the pipelined step costs what the bare forward costs
    assert 0.8 <= j["value"] / sw["value"] <= 1.1, (j["value"], sw["value"])
    assert j["roofline"]["frac"] > 0.3


def test_default_line_carries_the_hq_configs(dev):
    """`python bench.py` (what the driver records): headline fields incl. its own arithmetic tag and kernel map, and `secondary` =
    the pure fp32-MFMA configuration (E2FGVI_X3=0), SURVEY.md 8(d)'s second split (T=10, l_t=5), config 3's per-GPU work on one
    GPU (8 clips), e2fgvi_hq 720x1296 T=10 and 1080x1944 T=20 in bf16"""
    j = _bench("--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert j["metric"].startswith("inpainted frames/sec at 432x240 T=10") and j["dtype"] == "f32" and j["value"] > 30
    assert j["roofline"]["bound"] == "mfma" and 0 < j["roofline"]["frac"] < 1
    # round 6: no fraction above 1 anywhere on the line (round 5 printed direct-convolution FLOPs over the fp32 peak as a `frac_*`)
    fr = dict(_fracs(j))
    assert fr and all(0 < v < 1 for v in fr.values()), fr
    assert 0 < j["roofline"]["frac_useful"] <= j["roofline"]["frac"] and j["roofline"]["effective_peak"] == 416.7
    assert j["roofline"]["dominant_kernel"]["frac_useful"] <= j["roofline"]["dominant_kernel"]["frac"]
    assert "split operands" in j["config"]["arithmetic"] and j["config"]["kernel_selection"].startswith("e2fgvi_amd/tile_table.py")
    assert "encoder.layers.10" in j["config"]["kernels"] and j["peak_memory_gb"] > 0
    sec = j["secondary"]
    assert len(sec) == 6 and all("error" not in s for s in sec), sec
    x30, lt5, c8, fl3, hq720, hq1080 = sec
    # round 6: three forwards in flight (three graphs on three streams) -- same clip, same kernels, more of the chip busy
    # (the headline itself keeps two in flight since round 6 and carries the one-at-a-time number of the same process beside it)
    assert fl3["config"]["forwards_in_flight"] == 3 and j["config"]["forwards_in_flight"] == 2
    for s_ in (x30, lt5, c8, hq720, hq1080):          # ... and so do the other lines, each with its one-at-a-time number
        assert s_["config"]["forwards_in_flight"] == 2 and 0.75 * s_["value"] <= s_["sequential"]["value"] <= 1.08 * s_["value"], s_
    sq = j["sequential"]
    assert sq["forwards_in_flight"] == 1 and 0.75 * j["value"] <= sq["value"] <= 1.08 * j["value"], (sq, j["value"])
    assert fl3["value"] > 0.95 * sq["value"] and "in flight" in j["config"]["parallelism"]
    assert "E2FGVI_X3=0" in x30["config"]["workload"] and x30["dtype"] == "f32" and "fp32 MFMA" in x30["config"]["arithmetic"]
    assert not any("x3" in k for k in x30["config"]["kernels"].values()), x30["config"]["kernels"]
    assert "l_t=5" in lt5["config"]["workload"] and lt5["value"] > 30
    assert "8 clip(s)" in c8["config"]["workload"] and c8["value"] > 30 and abs(c8["value"] - 80e3 / c8["ms_per_step"]) <= 1e-2 * c8["value"]
    assert "720" in hq720["metric"] and "T=10" in hq720["metric"] and hq720["dtype"] == "bf16" and hq720["value"] > 30
    assert "1080" in hq1080["metric"] and "T=20" in hq1080["metric"] and hq1080["dtype"] == "bf16" and hq1080["value"] > 10
    for s in (hq720, hq1080):
        assert s["roofline"]["peak"] == 2500.0 and 0 < s["roofline"]["frac"] < 1 and s["config"]["hip_graph"] is True
        # round 6: the bf16 lines carry their parity against the REAL reference's fixtures (the timed engine's frames on the timed
        # clip at 720x1296 T=10; a stress-weights clip at the line's resolution and T = 10 / 8)
        p = s["parity"]
        assert 0 < p["max_abs"] <= p["bound_max_abs"] == 1e-2 and 0 < p["rms_of_difference_over_rms"] <= 2e-2, p
        assert p["stress"]["fixture"].startswith("tests/golden/g") and p["stress"]["rms_of_reference"] > 0.05
    assert hq720["parity"]["default"]["fixture"].endswith("benchclip.npz")


def test_bench_line_carries_its_own_parity(dev):
    """with the CPU baseline on, the line compares the frames of the TIMED configuration with the oracle's on the same clip
    (`parity`), and bench.py exits non-zero above the north-star bound"""
    j = _bench("--steps", "3", "--warmup", "1", "--no-secondary")
    p = j["parity"]
    assert p["bound_max_abs"] == 1e-3 and 0 <= p["max_abs"] <= 1e-3 and p["max_abs_over_rms"] <= 2e-3, p
    # round 5: the default-init weights (conv_offset[-1] == 0: offsets = flow, mask = 0.5) AND the stress weights
    for k in ("default", "stress", "peaked"):       # round 6: three weight regimes on the line
        assert 0 <= p[k]["max_abs"] <= 1e-3 and p[k]["max_abs_over_rms"] <= (2e-3 if k == "peaked" else 2e-4), (k, p[k])
    assert p["stress"]["rms_of_reference"] > 10 * p["default"]["rms_of_reference"]
    # the traffic figure is this library's or absent (never another build's)
    assert j["library_sha16"] and (j["roofline"]["traffic"] is None or j["library_sha16"] in j["roofline"]["traffic_note"])
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0


def test_two_processes_return_the_same_bits(dev):
    """deterministic kernel selection (ops.py, e2fgvi_amd/tile_table.py): two fresh processes run the same kernels in the same
    order on the headline configuration and return torch.equal frames"""
    code = ("import sys, importlib, hashlib, torch; sys.path.insert(0, %r);"
            "from e2fgvi_amd.synth import synth_clip, synth_state_dict;"
            "net = importlib.import_module('model.e2fgvi').InpaintGenerator(); net.load_state_dict(synth_state_dict('e2fgvi', 'stress', 0));"
            "net = net.cuda().eval(); x = synth_clip(1, 10, 240, 432, seed=5, moving=True)[0].cuda();"
            "o, (ff, fb) = net(x, 10); torch.cuda.synchronize();"
            "print('HASH', hashlib.sha256(o.cpu().numpy().tobytes() + ff.cpu().numpy().tobytes()).hexdigest())" % ROOT)
    hashes = []
    for _ in range(2):
        env = dict(os.environ)
        env.pop("E2FGVI_AUTOTUNE", None)
        p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        hashes.append([ln for ln in p.stdout.splitlines() if ln.startswith("HASH")][-1])
    assert hashes[0] == hashes[1], hashes
