"""CPU: the host-side arithmetic of bench.py — the byte formulas its roofline block prints,
the CPU count its baseline leg runs on, the power / clock reader.  No GPU, nothing under oracle/."""
import importlib.util
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    assert b.norm_kernel(d["kernels"]["mac"]) == b.norm_kernel("mac_walk3_kernel<33, 7, true, 1, 1>")
    assert 1.0 < e["bytes"]["mac"] / d["bytes"]["mac"] < 1.1            # the same rows, the second input's read beside the first


def test_trace_times_never_exceed_the_calls_wall_time():
    """A launch of a few tens of microseconds is timed by the committed profile's kernel trace; on a box faster than the one
    that took the profile those times are scaled to the call's wall time of this run (their sum cannot exceed it)."""
    b = _bench()
    ev = {"forward": 0.025, "mac": 0.020, "inverse": 0.026}
    tr = {"forward": 0.0217, "mac": 0.0158, "inverse": 0.0216}
    assert b.fit_trace_to_wall(tr, ev, 0.0617) == 1.0                       # 59.1 us of kernels in a 61.7 us call
    f = b.fit_trace_to_wall(tr, ev, 0.0570)                                  # a faster box: 57.0 us per call
    assert 0.96 < f < 0.97 and abs(sum(v * f for v in tr.values()) - 0.0570) < 1e-9
    assert b.fit_trace_to_wall({"mac": 0.0158}, ev, 0.0617) == 1.0          # (only when all three are trace times)
    assert b.fit_trace_to_wall({}, ev, 0.05) == 1.0
