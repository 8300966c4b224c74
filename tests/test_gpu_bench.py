"""bench.py through its command line on the GPU box: the contract line at N = 1, and the N > 1 control flow (barriers, the RCCL /
gloo exchange, max-over-ranks timing, rank-0 printing) with two ranks on ONE device (YGZ_BENCH_ONE_DEVICE=1 -> gloo; RCCL refuses two
ranks on one GPU, and 8-GPU runs are the driver's).  Small batches: this checks plumbing, not speed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_contract_line_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "32", "--no-cpu-baseline",
                        "--offline-frames", "64"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "roofline_valu"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["unit"] == "frames/s"
    assert d["config"]["frames_per_gpu_per_step"] == 32 and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "valu" and rf["kernel"] == "k_klt3" and rf["launches"] == 3 and rf["avg_launch_us"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert d["roofline_valu"]["mfma"]["kernel"].startswith("k_hamming_f4") and 0 < d["roofline_valu"]["mfma"]["frac"] < 1
    # the extra blocks measured after the timed region: BASELINE configs[4] (here a 64-frame sequence) and the transfer-inclusive mode
    off, st = d["offline"], d["stream"]
    assert "error" not in off and off["value"] > 0 and off["config"]["frames_total"] == 64 and off["result_check"]["ba_windows"] == 1
    assert off["result_check"]["max_abs_trajectory_error_vs_ground_truth"] < 0.05 and off["gray"]["value"] > 0
    assert "error" not in st and st["bgr"]["value"] > 0 and st["gray"]["value"] > 0 and st["batches_in_flight"] == 3
    # value = frames / time: consistent with ms_per_step
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    # per-step device times, and the PCIe-inclusive rate next to the HBM-resident value
    sm = d["step_ms"]
    assert sm["steps"] == 3 and 0 < sm["p10"] <= sm["median"] <= sm["p90"] and sm["median"] < 2.0 * d["ms_per_step"]
    assert d["value_with_transfers"] == st["bgr"]["value"] and "resident in HBM" in d["metric"]
    assert "frac_of_int8_peak" not in d["roofline_valu"]["mfma"]
    assert d["ranks"]["world"] == 1 and d["ranks"]["distinct_devices"] == 1 and len(d["ranks"]["per_rank"]) == 1


@pytest.mark.parametrize("mode", ["step", "offline"])
def test_bench_two_ranks_on_one_device(mode):
    env = dict(os.environ, YGZ_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    port = {"step": "29631", "offline": "29632"}[mode]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    cmd += ["--batch", "16"] if mode == "step" else ["--mode", "offline", "--frames", "32"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    if mode == "step":
        assert d["scaling"] == "weak" and abs(d["value"] - 2 * 16 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
        # who took part: two ranks (here on one device, through gloo); the metric says what the N > 1 value is
        assert d["ranks"]["world"] == 2 and [r["rank"] for r in d["ranks"]["per_rank"]] == [0, 1] and d["ranks"]["backend"] == "gloo"
        assert "replicas" in d["metric"]
    else:
        assert d["scaling"] == "strong" and d["config"]["frames_total"] == 32
        assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
        assert d["result_check"]["max_abs_trajectory_error_vs_ground_truth"] < 0.05
