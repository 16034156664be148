"""CPU: the host-side arithmetic of bench.py — the byte formulas its roofline block prints,
the CPU count its baseline leg runs on, the power / clock reader.  No GPU, nothing under oracle/."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_are_the_survey_figures():
    """bytes per block-channel = 12 P + 8 (P+1)(K_eff+1) + 8 (P+1) K_eff / S.
    cfg3/cfg5 (K = 32, S = 64): 2 294 028; cfg1 (K = 8, S = 1): 1 212 552; cfg4 (K = 64, S = 1): 8 553 480."""
    b = _bench()
    assert int(b.alg_bytes(8192, 32, 64)["total"]) == 2294028
    assert int(b.alg_bytes(8192, 8, 1)["total"]) == 1212552
    assert int(b.alg_bytes(8192, 64, 1)["total"]) == 8553480
    ab = b.alg_bytes(8192, 32, 64)
    assert ab["mac"] == 8 * 8193 * 32 + 8 * 8193 * 32 / 64          # K2's share: K rows of the stream + the shared filter
    # a T-block call needs at least 8P(T+K)/T + 8P in K2 (every row once, plus the K rows of history): a ninth of that
    tb = b.tiled_bytes(8192, 32, 256)
    assert tb["mac"] == 8 * 8192 * (256 + 32) / 256 + 8 * 8192 and tb["mac"] < ab["mac"] / 8
    assert tb["total"] == tb["forward"] + tb["mac"] + tb["inverse"]


def _canned_full_result(n_gpus=8):
    """A full result dict of the size a real run produces: round 5's 23.9 KB line (profiles/r05_bench_20steps.json, the
    one the driver could not parse) plus this round's keys and an 8-rank job's shards."""
    import json
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_20steps.json")).read().strip().splitlines()[-1])
    full["n_gpus"] = n_gpus
    full["shards"] = [[r + n_gpus * i for i in range(64)] for r in range(n_gpus)]
    full["process_group"] = {"backend": "nccl", "world_size": n_gpus, "forced_at_world_size_1": False}
    full["roofline"]["k2_valu"] = {"tflops": 60.12, "peak": 157.3, "frac": 0.3822, "flops_per_launch": 53150220288, "sclk_mhz": 1730.0, "frac_at_sclk": 0.5302}
    full["roofline"]["traffic_note"] = "x" * 900                       # prose of any length stays out of the line
    for c in full["configs"].values():
        c["roofline"]["k2_valu"] = {"tflops": 66.6, "peak": 157.3, "frac": 0.4234, "frac_at_sclk": 0.51, "sclk_mhz": 1990.0}
        c["roofline"]["bound"] = "valu"
    return full


def test_contract_line_is_small():
    """The driver keeps an 8 KB tail of stdout and parses the LAST line: round 5's 23.9 KB line came back `parsed: null`.
    The line built from a full-size result must stay under 4 KB, round-trip through json, and carry the contract's keys,
    `roofline` and `cpu_baseline` with the fields the contract names; everything else is in the details file it names."""
    import json
    from benchlib import line as L
    full = _canned_full_result()
    assert len(json.dumps(full)) > 20000                               # the input really is round 5's size
    text = L.render(full)
    assert len(text) < 4096 and "\n" not in text, len(text)
    out = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "details"):
        assert k in out, k
    assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"] and out["unit"] == "Msamples/s"
    assert "workload" in out["config"] and "model" not in out["config"] and out["config"]["partitions"] == 32
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_name", "kernel_ms", "kernels_ms",
              "min_bytes_per_launch", "frac_lower_bound", "path", "k2_valu"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "valu") and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = out["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "one_core", "zita_convolver_on_this_box"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert set(out["configs"]) == {"cfg1", "cfg2", "cfg4", "matrix"}
    for c in out["configs"].values():
        assert set(c) == {"msamples_per_s", "ms_per_call", "path_frac", "parity_rms", "bound", "k2_valu_frac"}
    assert out["shards"]["streams_per_rank"] == [64] * 8 and out["process_group"]["world_size"] == 8
    assert out["details"] == "bench_details.json"
    # a failed leg and a missing CPU leg still give a parseable, small line
    full["configs"]["cfg4"] = {"error": "RuntimeError('" + "y" * 500 + "')"}
    full["cpu_baseline"] = None
    out2 = json.loads(L.render(full))
    assert len(L.render(full)) < 4096 and out2["cpu_baseline"] is None and "error" in out2["configs"]["cfg4"]


def test_emit_prints_one_line_and_writes_the_details(tmp_path, capsys):
    import json
    from benchlib import line as L
    full = _canned_full_result(2)
    path = str(tmp_path / "details.json")
    L.emit(full, path)
    cap = capsys.readouterr()
    lines = cap.out.splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096                    # ONE line on stdout
    assert json.loads(lines[0])["details"] == path
    det = json.load(open(path))
    assert det["shards"] == full["shards"] and det["drop_in_threads"] == full["drop_in_threads"]     # nothing is lost
    assert det["roofline"]["all_kernels"] == full["roofline"]["all_kernels"]


def test_k2_flops_and_the_roof_that_binds():
    """K2's issued flops: 6 per complex multiply-add in the three-FMA walk, 8 otherwise; cfg4 6.54 Gflop,
    the 2 x 2 matrix 106 Gflop; `bound` is the larger fraction."""
    from benchlib import formulas as F
    from benchlib.configs import choose_bound
    assert F.walk_flops("mac_walk3_kernel<33, 7, true, 2, 1>", 8192, 64, 256, 8) == 6 * 65 * 8192 * 256 * 8
    assert round(F.walk_flops("mac_walk3_kernel<33, 7, true, 2, 2>", 8192, 32, 256, 128, 2) / 1e9) == 106
    assert F.walk_flops("mac_walk_kernel<33, 7, true, 4, 1, 1>", 8192, 32, 256, 128) == 8 * 33 * 8192 * 256 * 128
    v = F.valu_fractions(6.543e9, 0.0983, 1990.0)
    assert abs(v["tflops"] - 66.56) < 0.1 and abs(v["frac"] - 0.423) < 2e-3 and abs(v["frac_at_sclk"] - 0.5104) < 2e-3
    assert choose_bound(0.43, v) == "hbm" and choose_bound(0.40, v) == "valu" and choose_bound(None, v) == "valu"
    assert choose_bound(0.66, None) == "hbm"


def test_usable_cpus_is_bounded_by_the_affinity_mask():
    b = _bench()
    n, host, why = b.usable_cpus()
    assert 1 <= n <= host and n <= len(os.sched_getaffinity(0)) and why in ("affinity mask", "cgroup CPU quota")


def test_power_reader_on_a_fake_hwmon_tree(tmp_path):
    b = _bench()
    h = tmp_path / "hwmon" / "hwmon3"
    h.mkdir(parents=True)
    (h / "power1_input").write_text("1400000000\n")
    (h / "power1_cap").write_text("1400000000\n")
    (h / "freq1_input").write_text("1690000000\n")
    (tmp_path / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 1690Mhz *\n2: 2400Mhz\n")
    w = b.PowerWatch.__new__(b.PowerWatch)
    w.dirs, w.by_address, w.samples, w._thread, w._stop = [str(h)], False, [], None, None
    with w:
        time.sleep(0.25)
    s = w.summary()
    assert s["socket_w"] == 1400.0 and s["cap_w"] == 1400.0 and s["sclk_mhz"] == 1690.0
    assert s["sclk_max_mhz"] == 2400 and s["at_power_cap"] is True and s["samples"] >= 3


def test_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 4 --steps 7` outside torch.distributed.run: main() must hand over to a CHILD process running the
    launcher (never an exec, never after importing the engine), with the same arguments, rendezvous on 127.0.0.1."""
    import subprocess
    import sys
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    try:
        b.main()
        raise AssertionError("main() went on past the self-launch")
    except SystemExit as e:
        assert e.code == 7                                            # the child's exit code is ours
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_profile_applies_only_to_the_kernels_it_was_taken_from():
    """bench.py takes HBM bytes from profiles/traffic.json only for launches of the SAME kernels (by instantiation name, as
    the engine reports them) whose times agree: a kernel change that keeps the time must still drop the old profile."""
    b = _bench()
    entry = {"profile": "rXX", "bytes": {"mac": 100, "forward": 50}, "avg_ns": {"mac": 1000e3, "forward": 600e3},
             "kernels": {"mac": "mac_walk_kernel<33, 7, true, 4, 1, 1> grid=1048576", "forward": "forward_walker_kernel<13, true, false> grid=262144"}}
    ran = {"mac": "mac_walk_kernel<33, 7, true, 4, 1, 1>", "forward": "forward_walker_kernel<13, true, false>", "inverse": "x"}
    kms = {"mac": 1.05, "forward": 0.61, "inverse": 0.6}
    assert b.profile_applies(entry, ran, kms, 0.15, 0.0) == (True, None)
    ok, note = b.profile_applies(entry, dict(ran, mac="mac_walk3_kernel<33, 7, true, 1, 1>"), kms, 0.15, 0.0)
    assert not ok and "other kernels" in note and "mac_walk3_kernel<33,7,true,1,1>" in note          # same time, other kernel
    ok, note = b.profile_applies(entry, ran, dict(kms, mac=1.3), 0.15, 0.0)
    assert not ok and "differs" in note                                                              # same kernel, other time
    assert b.profile_applies(entry, ran, dict(kms, mac=1.3), 0.15, 0.0, roles=["forward"])[0]         # asked about K1 only
    assert b.profile_applies({}, ran, kms, 0.15, 0.0) == (False, None)
    assert b.norm_kernel("mac_kernel<1> grid=524288") == "mac_kernel<1>"


def test_every_profiled_kernel_exists_in_the_built_library():
    """profiles/traffic.json names the kernels its bytes were counted on; each of them must be an instantiation of the
    library as built (a profile of kernels that no longer exist cannot be matched by bench.py and is dead weight)."""
    import json
    import sys
    import tempfile
    import pytest
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    objs = [o for o in check_isa.default_objects() if os.path.exists(o)]
    if not (check_isa.tools_present() and len(objs) == 2):
        pytest.skip("needs the ROCm llvm tools and the built kernel objects")
    import ctypes
    cxa = ctypes.CDLL("libstdc++.so.6").__cxa_demangle
    cxa.restype = ctypes.c_void_p
    libc = ctypes.CDLL(None)

    def demangle(m):
        st = ctypes.c_int(0)
        p = cxa(m.encode(), None, None, ctypes.byref(st))
        try:
            return ctypes.string_at(p).decode() if (st.value == 0 and p) else m
        finally:
            if p:
                libc.free(ctypes.c_void_p(p))
    b = _bench()
    built = set()
    for o in objs:
        with tempfile.TemporaryDirectory() as tmp:
            names = [k["name"] for k in check_isa.kernel_metadata(check_isa.extract_code_object(o, tmp))]
        for line in map(demangle, names):
            head = line.split("(fk::")[0].split("(float")[0]                 # cut the argument list
            built.add(b.norm_kernel(head.split("::")[-1]))
    assert any(n.startswith("mac_walk3_kernel<") for n in built) and any(n.startswith("forward_walker_kernel<") for n in built)
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    missing = [(shape, role, name) for shape, e in tj.items() if isinstance(e, dict)
               for role, name in (e.get("kernels") or {}).items() if b.norm_kernel(name) not in built]
    assert not missing, missing


def test_the_matrix_configuration_has_its_own_profile_entry():
    """`bench.py --only-config matrix` (cfg3's batch through a full 2 x 2 filter matrix) shares S / T / K / C with the headline
    shape: its traffic.json entry has a key of its own and was counted on the two-paths-per-output walk."""
    import json
    b = _bench()
    assert b.traffic_key(64, 256, 32, 2) == "S64_T256_K32_C2" and b.traffic_key(64, 256, 32, 2, True) == "S64_T256_K32_C2_full"
    cfg = b.OTHER_CONFIGS["matrix"]
    assert cfg["full"] and (cfg["S"], cfg["C"], cfg["size"]) == (64, 2, 262144)
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    e, d = tj["S64_T256_K32_C2_full"], tj["S64_T256_K32_C2"]
    assert b.norm_kernel(e["kernels"]["mac"]) == b.norm_kernel("mac_walk3_kernel<33, 7, true, 2, 2>")
    assert b.norm_kernel(d["kernels"]["mac"]) == b.norm_kernel("mac_walk3_nt_kernel<33, 7, true, 1, 1>")      # (the streaming form: 2.1 GB of Y)
    assert 1.0 < e["bytes"]["mac"] / d["bytes"]["mac"] < 1.1            # the same rows, the second input's read beside the first


def test_kernel_times_in_the_tables_are_this_runs():
    """a per-kernel `ms` / `frac` must be of THIS run — the dispatch-bound event time — with the committed
    profile's kernel-trace average beside it under its own key, never substituted or scaled to fit."""
    from benchlib.configs import kernel_table
    from benchlib.formulas import tiled_bytes
    kms = {"forward": 0.0540, "mac": 0.1100, "inverse": 0.0630}             # K2 regressed from the profile's 98.3 us to 110
    ev = {"forward": 0.0590, "mac": 0.1150, "inverse": 0.0690}
    entry = {"profile": "r05_cfg4", "avg_ns": {"forward": 52600, "mac": 98300, "inverse": 62100},
             "kernels": {"mac": "mac_walk3_kernel<33, 7, true, 2, 1> grid=262144"}}
    by = {"forward": 202e6, "mac": 337e6, "inverse": 219e6}
    t = kernel_table(kms, ev, {"mac": "mac_walk3_kernel<33, 7, true, 2, 1>"}, entry, by, tiled_bytes(8192, 64, 256), 8 * 256)
    assert t["mac"]["ms"] == 0.11 and t["mac"]["event_ms"] == 0.115 and t["mac"]["trace_us"] == 98.3
    assert abs(t["mac"]["frac"] - 337e6 / 0.11e-3 / 8e12) < 1e-4            # the regression shows in the fraction
    assert not hasattr(_bench(), "fit_trace_to_wall")


def test_profile_summary_tells_the_streaming_walk_apart_and_survives_a_missing_trace(tmp_path):
    """tools/summarize_profile.py on a synthetic rocprofv3 output: `mac_walk3_nt_kernel` is a kernel of its own (not folded
    into `mac_walk3_kernel`), FETCH_SIZE is doubled (gfx950, MI355X_MICROARCH.md), and when the per-dispatch trace csv is
    missing (tools/profile.sh deletes csv files over 3 MB) the stats csv's averages of the same run stand in, marked as such."""
    import json
    import subprocess
    import sys
    nt = "void fk::(anonymous namespace)::mac_walk3_nt_kernel<33, 7, true, 1, 1>(fk::(anonymous namespace)::JobRef, fk::FilterDev, HIP_vector_type<float, 2u>*, int, int)"
    pl = "void fk::(anonymous namespace)::mac_walk3_kernel<26, 8, true, 1, 1>(fk::(anonymous namespace)::JobRef, fk::FilterDev, HIP_vector_type<float, 2u>*, int, int)"
    d = tmp_path / "prof"
    (d / "trace" / "r").mkdir(parents=True)
    (d / "pmc_FETCH_SIZE" / "r").mkdir(parents=True)
    (d / "pmc_WRITE_SIZE" / "r").mkdir(parents=True)
    (d / "trace" / "r" / "1_kernel_stats.csv").write_text(
        '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
        '"%s",10,8870000,887000.0,60.0,880000,890000,100.0\n"%s",20,322000,16100.0,2.0,15000,17000,100.0\n' % (nt, pl))
    (d / "trace" / "r" / "1_kernel_trace.csv").write_text(                       # only the nt kernel's dispatches survived
        '"Kernel_Name","Grid_Size","Start_Timestamp","End_Timestamp"\n' + "".join('"%s",1048576,%d,%d\n' % (nt, 1000 * i, 1000 * i + 887) for i in range(10)))
    for c, kib in (("FETCH_SIZE", 1000.0), ("WRITE_SIZE", 2000.0)):
        (d / ("pmc_" + c) / "r" / "2_counter_collection.csv").write_text(
            '"Kernel_Name","Grid_Size","Counter_Name","Counter_Value"\n' +
            "".join('"%s",%d,"%s",%f\n' % (k, g, c, kib) for k, g in ((nt, 1048576), (pl, 131072)) for _ in range(3)))
    out = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), str(d)], text=True))
    knt, kpl = "mac_walk3_nt_kernel<33, 7, true, 1, 1> grid=1048576", "mac_walk3_kernel<26, 8, true, 1, 1> grid=131072"
    assert out["kernel_trace"][knt]["avg_ns"] == 887 and out["kernel_trace"][knt]["dispatches"] == 10
    assert out["kernel_trace"][kpl]["avg_ns"] == 16100.0 and "kernel_stats.csv" in out["kernel_trace"][kpl]["source"]
    assert out["hbm_per_dispatch"][knt] == {"hbm_read_bytes_corrected": 2048000.0, "hbm_write_bytes": 2048000.0, "hbm_bytes": 4096000.0}
