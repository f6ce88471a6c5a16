"""bench.py emits exactly one JSON line with the driver's contract keys, for every --config, and starts its own
ranks when asked for more than one GPU (run small on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline")


def _run(cmd, env=None, timeout=900):
    out = subprocess.run(cmd, capture_output=True, text=True, env=env or dict(os.environ), timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout          # ONE line on stdout, nothing else
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    assert d["data"] == "synthetic" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] <= 1.0, r
    return d


def test_bench_json_contract(hip):
    d = _run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64",
              "--time-steps", "20", "--cpu-seconds", "1"])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64"
    r = d["roofline"]
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert "streaming_model" in r and r["traffic"] is None              # not the profiled configuration -> no static traffic
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["cpu_model"]
    assert [l["cores"] for l in c["legs"]][:2] == [1, 1] and "NumPy" in c["legs"][0]["sample"]
    assert d["value"] > 0 and d["rel_l2_vs_cpu_ref"] < 1e-10 and d["iters_match_cpu_ref"] is True


@pytest.mark.parametrize("config,batch", [("pod_galerkin", 300), ("pod_lspg", 300), ("quadratic", 280), ("ann", 300)])
def test_bench_rom_configs(hip, config, batch):
    """BASELINE configs[2..4] through the same contract: MFMA roofline, parity against the oracle outside the timed region."""
    d = _run([sys.executable, os.path.join(REPO, "bench.py"), "--config", config, "--steps", "1", "--warmup", "1", "--batch",
              str(batch), "--time-steps", "12", "--cpu-seconds", "1"])
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["peak"] == 78.6
    assert d["roofline"]["measured_ceiling_of_the_instruction_used"]["TFLOP/s"] == 62.0
    assert d["rel_l2_vs_cpu_ref"] < d["parity_tolerance"] and d["nonfinite_samples"] == 0
    if config != "ann":                                     # fp32 closure: counts may differ by one at the threshold
        assert d["iters_match_cpu_ref"] is True
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0


def test_bench_default_line_carries_the_other_configs(hip):
    """The driver's default command line also measures configs[2..4] (other_configs): here at the small size."""
    d = _run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "64",
              "--time-steps", "20", "--no-cpu-baseline", "--other-configs", "small", "--other-steps", "1"])
    oc = d["other_configs"]
    assert [e["config"]["name"] for e in oc] == ["pod_galerkin", "pod_lspg", "quadratic", "ann", "decoder_bf16", "pod_r96_galerkin",
                                                   "pod_r96_lspg"]
    for e in oc:
        assert "error" not in e, e
        assert e["value"] > 0 and e["ms_per_step"] > 0 and e["steps"] == 1 and 0.0 < e["roofline"]["frac"] <= 1.0
        assert e["rel_l2_vs_cpu_ref"] < e["parity_tolerance"] and e["nonfinite_samples"] == 0
        if e["config"]["name"] in ("pod_galerkin", "pod_lspg", "quadratic", "pod_r96_galerkin", "pod_r96_lspg"):
            assert e["iters_match_cpu_ref"] is True


def test_bench_decoder_bf16(hip):
    d = _run([sys.executable, os.path.join(REPO, "bench.py"), "--config", "decoder_bf16", "--steps", "1", "--warmup", "1",
              "--batch", "300", "--time-steps", "100"])
    assert d["dtype"] == "bf16" and d["roofline"]["bound"] == "hbm" and d["rel_l2_vs_cpu_ref"] < d["parity_tolerance"]


def test_bench_starts_its_own_ranks(hip):
    """`python bench.py --gpus 2` with no RANK in the environment: the parent spawns two rank processes before touching
    the GPU (gloo rehearsal on this box's one GPU: RCCL refuses two ranks per device), relays one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    d = _run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64",
              "--time-steps", "20"], env=env)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and "cpu_baseline" not in d
    assert d["rel_l2_vs_cpu_ref"] < 1e-10 and d["iters_match_cpu_ref"] is True
    assert d["allgather_svd_ms"] > 0 and d["allgather_svd"]["snapshots"] == 2 * 8 * 5


def test_bench_under_torchrun(hip):
    """The driver's launch line (torch.distributed.run, RANK in the environment) on a free port."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, BG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
              "--batch", "64", "--time-steps", "20"], env=env)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["global_batch"] == 128


def test_bench_failed_rank_is_a_failure(hip):
    """A rank that dies makes the self-launching parent exit non-zero (never a silent 1-GPU number)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--batch", "8", "--n", "99999", "--time-steps", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
