"""bench.py prints ONE JSON line with the fields the driver reads (the contract in the task
statement): metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
scaling / vs_baseline / dtype / data / config, plus the roofline and cpu_baseline objects."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_contract(d, steps, warmup):
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == baseline["metric"]
    for key in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 and 0.0 < r["frac"] <= 1.0
    assert r["algorithmic_bytes"] > 0 and "traffic" in r
    return r


@pytest.mark.parametrize("workload", ["sym5-l8", "coif4-l14"])
def test_bench_line_has_the_contract_fields(workload):
    """The headline workload (coif4-l14) included: its roofline must be a fraction of the peak."""
    d = _run("--workload", workload, "--steps", "2", "--warmup", "2", "--cpu-frames", "1", "--cpu-fe-frames", "2")
    r = _check_contract(d, 2, 2)
    assert r["bound"] == "mfma" and r["algorithmic_TFLOPs"] > 0
    # every matrix-core class reports an issued-flops fraction below the peak
    for name, c in d["classes"].items():
        if "mfma_frac" in c:
            assert 0.0 < c["mfma_frac"] <= 1.0, (name, c)
    # no hole on the line: every class states its algorithmic bytes, and when a committed counter summary exists for
    # the workload every class whose kernels the summary knows has its HBM traffic beside them
    for name, c in d["classes"].items():
        assert c["algorithmic_bytes_per_step"] > 0, (name, c)
        assert c.get("mfma_frac") or c.get("achieved_GBps"), (name, c)
    if r.get("traffic_source"):
        assert r["traffic"] and r["traffic"] > 0
        for name, c in d["classes"].items():
            assert c["hbm_bytes_per_step_pmc"] and c["hbm_bytes_per_step_pmc"] > 0, (name, c)
            # tensors that do not fit the 256 MiB Infinity Cache cannot be moved with fewer HBM bytes than they hold:
            # a class whose counter bytes fall below its algorithmic bytes is a calibration error (tools/pmc_json.py)
            if c["algorithmic_bytes_per_step"] > 2 * 256 * 2 ** 20:
                assert c["hbm_bytes_per_step_pmc"] >= 0.9 * c["algorithmic_bytes_per_step"], (name, c)
    if workload == "coif4-l14":
        # every launch of the library belongs to a timing class: the classes add up to the step
        assert 0.97 <= d["classes_share_of_step"] <= 1.05, (d["classes_sum_ms_per_step"], d["ms_per_step"])
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["front_end_batch"] == 2 and c["step_batch"] == 1
    assert "cropped" not in c["sample"] and "scaled" not in c["sample"]
    # the same step fed from WAV files through the native loader (host-to-device copy included)
    e = d["end_to_end"]
    assert e["value"] > 0 and e["steps"] >= 1 and 0.5 * d["value"] <= e["value"] <= 1.2 * d["value"], e
    if workload == "coif4-l14":
        # the front-end-only workloads ride on the default workload's line (BASELINE configs[3] among them)
        fe = {(f["workload"], f["batch"]): f for f in d["frontend_only"]}
        haar = fe[("packets-haar level-14 front end only", 4096)]
        assert haar["algorithmic_bytes_per_frame"] == 4 * (22050 + 32768) and 0.3 < haar["frac_of_hbm_peak"] <= 1.0
        assert all(0.0 < f["frac_of_hbm_peak"] <= 1.0 and 0.0 < f["frac_of_f32_peak"] <= 1.0 for f in d["frontend_only"])
        # and so do the other BASELINE configurations (configs[0], [2], [4] and the shipped level-8 models)
        sec = {f["workload"]: f for f in d["secondary"]}
        assert len(sec) == 6 and not any("error" in f for f in d["secondary"]), d["secondary"]
        for f in d["secondary"]:
            assert f["ms_per_step"] > 0 and 0.0 < f["frac"] <= 1.0 and f["dominant_class"], f
        # configs[0] names a CPU path: its own CPU leg rides on its entry, at the protocol's batch
        c0 = sec["STFT(n_fft 511, hop 220) + DCNN train step"]["cpu_baseline"]
        assert c0["kind"] == "port" and c0["value"] > 0 and c0["front_end_batch"] == 128 and c0["step_batch"] == 128
        lcnn = sec["STFT(n_fft 511, hop 220) + LCNN eval forward (bf16 matrix products)"]
        assert lcnn["dtype"] == "bf16"
        # configs[4]: accuracy / EER of the evaluation loop on the synthetic cross-generator labels, and the batch a
        # GPU wants beside the protocol's 128
        assert 0.0 <= lcnn["accuracy"] <= 1.0 and 0.0 <= lcnn["eer"] <= 1.0 and lcnn["eval_frames"] == 1024
        assert set(lcnn["per_label_accuracy"]) == {"A_real", "B_melgan", "C_hifigan"}
        assert lcnn["large_batch"]["batch"] == 1024 and lcnn["large_batch"]["ms_per_step"] > 0
    else:
        assert d["frontend_only"] is None and d["secondary"] is None


def test_frontend_workload_line():
    """BASELINE configs[3] (Haar level 14, front end only) at a reduced batch: HBM-bound roofline
    from >= 20 timed launches."""
    d = _run("--workload", "haar-l14-frontend", "--batch", "256", "--steps", "5", "--warmup", "2",
             "--cpu-frames", "1", "--cpu-fe-frames", "2")
    r = _check_contract(d, 5, 2)
    assert r["bound"] == "hbm" and r["kernel"] == "wpt"
    assert d["frontend"]["launches_timed"] >= 20
    assert d["frontend"]["algorithmic_bytes_per_frame"] == 4 * (22050 + 32768)
    assert d["config"]["features"] == [1, 16384, 2]


def test_bench_spawns_its_own_ranks():
    """`--gpus N` without a launcher starts the ranks as a child torch.distributed.run (here one
    rank, the only GPU of the box) and relays rank 0's line; the RCCL process group is up."""
    d = _run("--spawn", "--gpus", "1", "--workload", "sym5-l8", "--steps", "2", "--warmup", "1",
             "--cpu-frames", "0")
    assert d["n_gpus"] == 1 and d["world"]["size"] == 1
    assert d["world"]["backend"] and d["world"]["rccl_version"]
    assert len(d["world"]["devices"]) == 1 and "cuda:0" in d["world"]["devices"][0]
    # the line carries what the step exchanged (nothing with one rank; the first multi-GPU run records its own)
    assert d["world"]["collectives"]["per_step"] == 0 and d["world"]["collectives"]["bytes_per_step"] == 0
