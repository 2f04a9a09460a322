"""The driver's contract with bench.py, checked on the GPU box: one JSON line on stdout with the agreed keys, the headline
metric on BASELINE.json's configuration, `roofline` and `cpu_baseline` objects, a parity verdict, and the nested records
of configs 3-5.  Short runs (few steps); the numbers themselves are not asserted, their consistency is."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


LINE_MAX_BYTES = 6144         # the driver could not read a 22 KB line; bench.compact_line refuses to print more than this


def run_bench(tmp_path, *args):
    """(the ONE stdout line, the full record bench.py wrote beside it)"""
    # the contract is about the DEFAULT line: routing and selection knobs of the caller's shell do not reach the child
    env = {k: v for k, v in os.environ.items() if not (k.startswith("FOURQ_") and k != "FOURQ_AMD_LIB")}
    env["FOURQ_BENCH_SETTLE_MS"] = "10"
    full_path = str(tmp_path / "bench_full.json")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-json", full_path] + list(args), capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line: %r" % lines[:3]
    assert len(lines[0]) <= LINE_MAX_BYTES, "the line is %d bytes" % len(lines[0])
    assert proc.stdout.rstrip("\n").endswith(lines[0])                    # ... and it is the last thing on stdout
    with open(full_path) as fh:
        full = json.load(fh)
    assert json.loads([ln for ln in proc.stderr.splitlines() if ln.startswith('{"metric"')][-1]) == full      # the same record on stderr
    return json.loads(lines[0]), full


def test_the_compact_line_is_what_the_driver_reads(tmp_path):
    """one line, numbers only, <= 6 KB, with the contract keys, `roofline` and `cpu_baseline` at top level and one flat
    record per nested config; everything else in the full record."""
    short, full = run_bench(tmp_path, "--steps", "20", "--warmup", "5")         # the driver's own arguments
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                     ("cpu_baseline", dict), ("valu_roofline", dict), ("configs", dict), ("pcie", dict), ("clock_mhz", float), ("cycles_per_unit", float),
                     ("ct_value", float), ("ct_ratio", float), ("parity_ok", bool)):
        assert isinstance(short[key], typ), key
    assert short["vs_baseline"] is None and short["steps"] == 20 and short["warmup"] == 5 and short["parity_ok"] is True
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert short[k] == full[k], k
    assert set(short["config"]) == {"workload", "batch_per_gpu", "ranks_seen", "backend", "build_id", "built_from_sources", "table_selection"}
    assert len(short["config"]["workload"]) <= 120 and "configs[1]" in short["config"]["workload"]
    assert set(short["roofline"]) == {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch"}
    assert short["roofline"] == {k: full["roofline"][k] for k in short["roofline"]} and len(short["roofline"]["kernel"]) < 40
    c = short["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] <= c["cores_cap"] and c["cores"] <= c["host_cores_granted"] <= c["host_cores_total"]
    assert c["per_core_reference_survey"] == 410.0 and c["c_restatement_value"] > c["value"] > 0 and abs(c["per_core"] * c["cores"] - c["value"]) / c["value"] < 0.25
    assert set(short["configs"]) == {"cfg3", "cfg4", "cfg5"}
    for name, rec in short["configs"].items():
        assert len(json.dumps(rec, separators=(",", ":"))) <= 300, name
        assert rec["parity_ok"] is True and rec["value"] == full["configs"][name]["value"]
        assert rec["issue_frac"] is None or 0 < rec["issue_frac"] <= 1.05          # None until the PMC profile of THIS build is committed
        assert rec["pcie_value"] > 0 and 0.95 < rec["ct_ratio"] < 3.0 and 5 < rec["cycles_per_unit"] < 40
    assert short["ct_value"] == full["ct_select"]["cfg2"]["value"] and short["ct_ratio"] == full["ct_select"]["cfg2"]["ratio_vs_default"]
    p = short["pcie"]
    assert 0 < p["r1_2p20_ms"] and 0 < p["affine_2p20_ms"] and 0 < p["bytes_2p20_ms"] and p["ms"] > 0
    # no prose: every string of the line is a short label
    def strings(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, str):
            yield x
    assert max(len(t) for t in strings(short)) <= 160
    assert short["fields"] == "profiles/BENCH_FIELDS.md" and os.path.exists(os.path.join(ROOT, short["fields"]))


def test_default_line_has_the_contract_keys(tmp_path):
    _, line = run_bench(tmp_path, "--steps", "20", "--warmup", "2")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict), ("valu_roofline", dict), ("parity", dict),
                     ("configs", dict), ("pcie_inclusive", dict), ("alongside", dict)):
        assert isinstance(line[key], typ), key
    assert line["vs_baseline"] is None and line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 2
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic" and line["dtype"] == "u64"
    assert "configs[1]" in line["config"]["workload"] and line["config"]["batch_per_gpu"] == 1 << 16
    assert abs(line["value"] - (1 << 16) / (line["ms_per_step"] * 1e-3)) / line["value"] < 0.01       # units / time, whole job
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert r["algorithmic_bytes_per_launch"] == 352 << 16 and r["kernel_ms"] <= line["ms_per_step"] * 1.05
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    v = line["valu_roofline"]
    assert 0 < v["algorithmic_frac"] < v["executed_frac"] < 1 and v["algorithmic_mads_per_unit"] == 49440
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "oracle/curve4q_oracle.py" in c["sample"]
    assert c["c_restatement"]["threads"] == c["c_restatement"]["cores"]                              # OpenMP took the thread count
    assert line["parity"]["ok"] is True and line["parity"]["units"] == 1 << 16 and line["parity"]["edge_cases_checked"] == 54
    assert set(line["configs"]) == {"cfg3", "cfg4", "cfg5"}
    for name, rec in line["configs"].items():
        assert rec["parity"]["ok"] is True and rec["ms_per_step"] > 0 and rec["roofline"]["frac"] > 0, name
        assert 0 < rec["valu_roofline"]["algorithmic_frac"] < 1
    for rec, op in ((line["alongside"], "MUL_windowed"), (line["configs"]["cfg3"]["alongside"], "MUL_endo")):   # the other ladder, reported alongside
        assert rec["op"].startswith(op) and rec["parity_ok"] is True and rec["ms_per_step"] > 0 and 0 < rec["algorithmic_frac"] < rec["executed_frac"] < 1
    p = line["pcie_inclusive"]
    assert p["value"] < line["value"] and p["gbs_h2d"] > 10 and p["gbs_d2h"] > 10 and p["pageable_caller"]["value"] > 0
    for name, rec in line["configs"].items():                     # wall-clock first H2D byte to last D2H byte, every config
        q = rec["pcie_inclusive"]
        assert 0 < q["value"] < rec["value"] * 1.02 and q["pageable_caller"]["value"] > 0, name
    # round 3: what the line says about itself
    assert v["peak_measured"] < v["peak"] and v["executed_frac_of_measured_peak"] > v["executed_frac"]
    v4 = line["configs"]["cfg4"]["valu_roofline"]
    assert v4["algorithmic_mads_per_unit_of_the_algorithm_run"] < v4["algorithmic_mads_per_unit"]
    assert v4["algorithmic_frac_of_the_algorithm_run"] < v4["algorithmic_frac"]
    lib = line["config"]["library"]
    assert lib["version"] == 600 and len(lib["build_id"]) == 16
    from fourq_amd import build
    assert lib["build_id"] == build.source_id() and lib["built_from_these_sources"] is True         # what was timed is what the sources say
    src = r["traffic_source"]
    assert src["loaded_library_build_id"] == lib["build_id"]
    assert (r["traffic"] is None) == (src.get("profiled_library_build_id") != lib["build_id"])      # a figure only for the build it was measured on
    assert line["config"]["table_selection"] == "indexed"
    # round 5: every workload record carries the in-kernel clock it was measured at and its cost in cycles (comparable across boxes)
    for name, rec in [("cfg2", line)] + sorted(line["configs"].items()):
        clk = rec["clock"]
        cpu = rec["cycles_per_unit"]
        assert 1200 < clk["min_mhz"] <= clk["in_kernel_mhz"] <= clk["max_mhz"] < 2600, (name, clk)
        assert clk["mode"] == "bracket" and clk["valid"] is True and 0.97 < clk["window_over_timed_span"] < 1.05, (name, clk)    # round 6: the probe's window IS the timed region
        ms = rec["roofline"]["kernel_ms"]
        n_units = rec["config"]["batch_per_gpu"] if name == "cfg2" else rec["batch_per_gpu"]
        assert abs(cpu - ms * 1e-3 * clk["in_kernel_mhz"] * 1e6 / n_units) / cpu < 1e-3, name
    assert 8.0 < line["cycles_per_unit"] < 13.0                   # MUL_endo, variable base: ~ 650 k wave cycles per generation of 65 536 lanes
    sw = line["size_sweep"]                                        # small batches and remainders run two or four lanes per element: the cliff, driver-visible
    assert sw["1"] < 0.6 * sw["65536"] and sw["1024"] < 0.6 * sw["65536"] and sw["16384"] < sw["32768"] < 0.8 * sw["65536"] and sw["t(65792)/t(65536)"] < 1.75
    sb = line["small_batches"]                                     # the other operations at 1 and 1 024 elements, checked against the oracle
    assert sb["parity_ok"] is True
    assert sb["keygen: comb of [392]G == DH_endo(m, G)"]["1"] < 0.6 * sb["DH_endo(m, Q)"]["1"] < sb["MUL_windowed(m, P)"]["1"]
    assert sb["MUL_endo mixed 50/50"]["1024"] < 0.8 * sw["65536"] and sb["MUL_endo(m, G, table) fixed base"]["1"] < sw["1"]
    ct = line["ct_select"]                                         # the constant-time mode, driver-visible: same outputs, its price
    assert set(ct) == {"cfg2", "cfg3", "cfg4", "cfg5"}
    for name in ("cfg2", "cfg3", "cfg4", "cfg5"):
        assert ct[name]["parity_ok"] is True and 0.95 < ct[name]["ratio_vs_default"] < 3.0, name


def test_single_workload_line(tmp_path):
    short, line = run_bench(tmp_path, "--workload", "cfg5", "--steps", "10", "--warmup", "1", "--no-configs", "--no-cpu-baseline", "--no-pcie", "--no-ct")
    assert "configs[4]" in line["config"]["workload"] and "configs" not in line and "cpu_baseline" not in line and "alongside" not in line
    assert "ct_select" not in line and "pcie_inclusive" not in line
    assert line["parity"]["ok"] is True and line["config"]["batch_per_gpu"] == 1 << 17
    assert "configs" not in short and "cpu_baseline" not in short and "pcie" not in short and short["parity_ok"] is True and "configs[4]" in short["config"]["workload"]
