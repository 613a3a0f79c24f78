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
    """`python bench.py` (what the driver records): headline fields + `secondary` = e2fgvi_hq 720x1296 T=10 and 1080x1944 T=20, bf16"""
    j = _bench("--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert j["metric"].startswith("inpainted frames/sec at 432x240 T=10") and j["dtype"] == "f32" and j["value"] > 30
    assert j["roofline"]["bound"] == "mfma" and 0 < j["roofline"]["frac"] < 1
    sec = j["secondary"]
    assert len(sec) == 2 and all("error" not in s for s in sec), sec
    assert "720" in sec[0]["metric"] and "T=10" in sec[0]["metric"] and sec[0]["dtype"] == "bf16" and sec[0]["value"] > 30
    assert "1080" in sec[1]["metric"] and "T=20" in sec[1]["metric"] and sec[1]["dtype"] == "bf16" and sec[1]["value"] > 10
    for s in sec:
        assert s["roofline"]["peak"] == 2500.0 and 0 < s["roofline"]["frac"] < 1 and s["config"]["hip_graph"] is True
