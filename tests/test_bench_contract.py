"""CPU: the bench line's CONTRACT, checked on the newest line committed under profiles/ (bench.py itself needs a GPU).  The
keys the driver reads, the two objects this tier adds (`roofline`, `cpu_baseline`) with their fields, and the round-5 / round-6 additions
the documents cite (both roofline fractions with their sources, the N = 256 CPU forward, the 32-row parity, the audits, the
config-4 leg's rank fields)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_r100_driver_flags.json")) or
                   glob.glob(os.path.join(ROOT, "profiles", "r*_bench_r100.json")))
    assert files, "no committed bench line under profiles/"
    return files[-1], json.loads(open(files[-1]).read().strip().splitlines()[-1])


def test_committed_bench_line_keeps_the_contract():
    name, l = _newest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in l, (name, k)
    assert l["unit"] == "embeddings/s" and l["higher_is_better"] is True and l["scaling"] == "weak" and l["vs_baseline"] is None
    assert l["data"] == "synthetic" and l["dtype"] in ("bf16", "f16", "f16x2", "f32") and "workload" in l["config"] and "model" not in l["config"]
    assert abs(l["value"] - l["n_gpus"] * l["steps"] * l["config"]["batch_per_gpu"] / (l["ms_per_step"] * l["steps"] / 1e3)) < 1e-6 * l["value"]
    r = l["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 0.6
    assert r["frac_hip_events"] == r["frac"] and 0.3 < r["frac_rocprof"] < 0.6 and abs(r["frac_rocprof"] - r["frac"]) < 0.05
    assert os.path.exists(os.path.join(ROOT, "profiles", r["rocprof"]["source"]))
    assert os.path.exists(os.path.join(ROOT, "profiles", r["traffic"]["source"])) and r["traffic"]["measured_in_this_run"] is False
    c = l["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["batch"] == 256 and c["cores"] >= 1 and c["fastest_batch"]["value"] >= c["value"] * 0.5
    p = l["parity"]
    assert p["oracle_rows"] == 32 and p["one_minus_cos_vs_cpu_oracle_max"] < 1e-3 and p["one_minus_cos_vs_cpu_oracle_mean"] <= p["one_minus_cos_vs_cpu_oracle_max"]
    assert p["batch1_rows_bit_equal_to_timed_batch"] is True
    for leg in (l["config3"]["screen_settle"], l["config4"]["screen_settle"]):
        a = leg["audit"]
        assert a["m"] > 0 and a["passes"] >= 1 and "3/m" in a["claim"] and leg["identical_to_exact_all"] is True
    c4 = l["config4"]
    assert c4["n_gpus"] == l["n_gpus"] and len(c4["pair_rows_per_rank"]) == l["n_gpus"] and c4["ranks_end_with_identical_student_weights"] is True
    assert sum(c4["pair_rows_per_rank"]) == 3840
    assert l["exact_selection"]["dtype"] == "f16x2" and l["exact_selection"]["max_abs_diff_vs_cpu_oracle"] < 2e-5
    # BASELINE.md §4 leg 3: the head's fine-tune step AND the SmallRes 32 x 32 end-to-end step, on the GPU and on the host
    assert 0 < l["finetune_step_ms"] < 1 and 0 < l["smallres32_train_step_ms"] < 20
    assert l["cpu_baseline"]["finetune_step_ms_cpu"] > l["finetune_step_ms"] and l["cpu_baseline"]["smallres32_train_step_ms_cpu"] > l["smallres32_train_step_ms"]
    # round 6: the at-reference-precision rates at the top level, the few-pixel attack leg, the training loops
    assert "bf16 < reference f32" in l["dtype_note"] and l["value_exact"]["dtype"] == "f16x2" and 0 < l["value_exact"]["embeddings_per_s"] < l["value"]
    vi = l["value_identical_selection"]
    assert vi["all_exact_embeddings_per_s"] < vi["embeddings_per_s"] < l["value"] and "bit-equal" in vi["workload"]
    c5 = l["config5"]
    assert c5["lockstep_images_identical_to_one_after_another"] is True and c5["n_gpus"] == l["n_gpus"]
    for mode in ("screen", "bf16", "exact"):
        assert c5[mode]["backbone_forwards_per_pair"] == 20400 and 0.9 < c5[mode]["frac_of_in_batch_rate"] <= 1.02, (mode, c5[mode])
    assert c5["screen"]["backbone_forwards_per_s"] >= 40000 and c5["exact"]["search_dtype"] == "f16x2"
    assert 0 < l["custom_train_step_ms"] < 0.08 and l["smallres32_train_step_ms"] < 0.6
    assert 0 < l["smallres32_train_step_resident_ms"] <= l["smallres32_train_step_ms"]          # operands already in HBM: no staging, no upload
    assert l["parity"]["normalized_weights"]["one_minus_cos_vs_cpu_oracle_max"] < l["parity"]["one_minus_cos_vs_cpu_oracle_max"]
    assert l["config4"]["finetune_steps"]["rows_per_step"] == 16
