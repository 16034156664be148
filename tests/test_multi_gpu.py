"""GPU: the multi-GPU code paths on a one-GPU box.

(a) The in-process sharder (folve::DeviceRouter under ProcessorPool, the reference's
    processor-pool.cc:48-91 plus "which GPU") with TWO slots — both on device 0, which exercises
    everything except a second physical GPU: least-loaded placement, one engine and one committed
    filter per (configuration, slot), the per-slot combiner, parity.
(b) bench.py's N = 2 path under torch.distributed.run (gloo, both ranks on device 0): sharding by
    stream index, max-over-ranks timing, one JSON line from rank 0.
Both run in fresh subprocesses: FOLVE_AMD_DEVICES is read once per process, and the parent must not
have touched the GPU for the launcher's children."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_router_slots_share_the_load(tmp_path):
    env = dict(os.environ, FOLVE_AMD_DEVICES="0,0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "router_worker.py"), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("ROUTER_JSON ")][-1]
    out = json.loads(line[len("ROUTER_JSON "):])
    assert out["slots"] == 2 and out["engines"] == 2
    assert out["live_while_held"] == [4, 4] and out["per_engine"] == [4, 4]     # least-loaded placement alternates
    assert out["cached_filters"] == 2                                           # one committed filter per (config, slot)
    assert out["max_rms"] <= 1e-5
    assert out["pooled"] == 8


def test_cfg5_shape_eight_slots_512_streams_and_one_gpu_going_bad(tmp_path):
    """BASELINE.json configs[4] on one device: eight router slots, 512 SoundProcessors of cfg3's filter opened through
    ProcessorPool from 64 threads (64 per slot exactly), all 512 converting at once with spot parity against float64;
    then one slot's engine starts failing every call while its 64 files are in mid-conversion: they move to the other
    slots from their kept input and come out equal to the float64 convolution, the slot is fenced, the next opens land on
    the other seven and none returns NULL; when the engine works again a probe puts the slot back in service.  (tests/cfg5_worker.py; /root/reference/processor-pool.cc:48-91)"""
    env = dict(os.environ, FOLVE_AMD_DEVICES="0,0,0,0,0,0,0,0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cfg5_worker.py"), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("CFG5_JSON ")][-1][len("CFG5_JSON "):])
    bad = out["bad_slot"]
    assert out["slots"] == 8 and out["distinct_engines"] == 8
    assert out["live"] == [64] * 8 and out["per_slot"] == [64] * 8              # cfg5: 64 streams per GPU
    assert out["cached_filters"] == 8                                            # one committed filter per slot
    assert out["checked"] >= 12 and out["max_rms"] <= 1e-5
    assert out["ok_before"] == 512 and out["states_before"] == [0] * 8
    # one GPU goes bad in mid-conversion: its 64 files MOVE to the other slots and come out right — every one of them checked
    # against float64, peaks included, no block of silence; the slot is fenced, the others never noticed
    assert out["files_on_bad"] == 64 and out["moved"] == 64 and out["max_moves"] == 1 and out["still_on_bad"] == 0
    assert out["ok_after"] == 512
    assert out["checked_bad_phase"] >= 64 and out["max_rms_bad_phase"] <= 1e-5 and out["silent_blocks"] == 0 and out["peak_err"] <= 1e-6
    assert out["live_after_move"][bad] == 0 and sum(out["live_after_move"]) == 512
    assert out["states_bad"] == [2 if s == bad else 0 for s in range(8)] and out["failures_bad"] >= 3
    assert out["more_null"] == 0 and out["more_on_bad"] == 0 and out["rms_more"] <= 1e-5
    assert out["live_more"][bad] == 0 and sum(out["live_more"]) == 512 + 56
    assert out["pooled"] == 512 + 56 and out["live_pooled"][bad] == 0            # nothing was lost: every processor went back to the pool
    assert out["again_on_bad"] == 0
    # and back in service
    assert out["state_back"] == 0 and out["back_on_bad"] == 8 and out["rms_back"] <= 1e-5


def test_an_open_file_moves_at_every_depth_and_position(tmp_path):
    """tests/survive_worker.py: one file, two router slots on device 0; the slot's engine dies after 0 / 2 / 40 / all whole
    blocks have been handed out, at run-ahead depths 1 / 4 / 64, through a stereo K = 25 filter and a 2 -> 3 channel one.  Every
    file equals its reference (float64 convolution / closed form), moved exactly once, no silent block, same peak."""
    env = dict(os.environ, FOLVE_AMD_DEVICES="0,0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "survive_worker.py"), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("SURVIVE_JSON ")][-1][len("SURVIVE_JSON "):])
    assert len(out["cases"]) == 24
    for c in out["cases"]:
        assert c["rms"] <= 1e-5 and c["silent_blocks"] == 0 and c["ok"] == 1 and c["peak_err"] <= 1e-6, c
        assert c["moves"] == 1 and c["engine_changed"], c


def test_a_file_beyond_the_old_history_budget_moves_and_a_stale_one_does_not(tmp_path):
    """VERDICT r05 weak #10 and ADVICE r05: (a) a 16-channel K = 128 file (135 MB of input history, beyond the 64 MB the
    history was budgeted with) loses its GPU at block 150 and moves — equal to its closed form, no silent block; (b) a file
    whose configuration was edited while it was open does NOT move (the other GPU would get the new taps): silence,
    ok() == false, and the next open gets the new configuration.  tests/survive_big_worker.py, a fresh process."""
    env = dict(os.environ, FOLVE_AMD_DEVICES="0,0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "survive_big_worker.py"), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("SURVIVE_BIG_JSON ")][-1][len("SURVIVE_BIG_JSON "):])
    big = out["big"]
    assert big["history_bytes"] > (64 << 20) and big["partitions"] == 128
    assert big["moves"] == 1 and big["engine_changed"] and big["ok"] == 1 and big["silent_blocks"] == 0, big
    assert big["rms"] <= 1e-5 and big["max_err"] <= 1e-4 and big["peak_err"] <= 1e-6, big
    st = out["stale"]
    assert st["moves"] == 0 and st["ok"] == 0 and not st["config_up_to_date"], st
    assert 10 <= st["first_bad_block"] <= 18 and st["tail_is_silence"], st          # (up to two run-ahead chunks of 4 were already computed)
    assert st["next_open_gain_err"] <= 1e-6, st


def run_bench(args, env, tmp_path, timeout=900, launcher=None):
    """bench.py as the driver runs it (or under `launcher`): returns (the ONE stdout line parsed, the details file parsed).
    The line must be the only JSON line, under 4 KB, and name the details file."""
    details = str(tmp_path / "bench_details.json")
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + list(args) + ["--details", details]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                           # rank 0 only, one line
    assert len(lines[0]) < 4096, len(lines[0])                                 # the driver keeps an 8 KB tail (round 5's line was lost)
    out = json.loads(lines[0])
    assert out["details"] == details
    return out, json.load(open(details))


def test_the_line_the_driver_records(tmp_path):
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5`, every leg on — the command behind BENCH_rNN.json.  Round 5's line
    of this command was 23.9 KB and came back unparsed; it must be ONE line under 4 KB that carries the contract's keys,
    `roofline` (this run's kernel time, a fraction of a roof) and `cpu_baseline`, with one small entry per other
    configuration; the per-kernel tables live in the details file, whose kernel times fit inside their calls."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "FOLVE_BENCH_DEVICE", "FOLVE_BENCH_BACKEND", "FOLVE_BENCH_FORCE_DIST"):
        env.pop(k, None)
    out, det = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "4", "--skip", "drop_in"], env, tmp_path, timeout=1200)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in out, k
    assert out["steps"] == 20 and out["warmup"] == 5 and out["n_gpus"] == 1 and out["unit"] == "Msamples/s" and out["dtype"] == "f32"
    assert abs(out["value"] - 64 * 256 * 8192 * 2 / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * out["value"]     # value IS samples / time
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "valu") and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["kernel_ms"] > 0
    assert rf["kernel_name"].startswith("mac_walk3_nt_kernel<33,") and 0.3 < rf["frac_lower_bound"] < 1.0
    assert rf["frac"] is None or (0.3 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["traffic"] / (rf["kernel_ms"] * 1e-3) / 8e12) < 2e-3)
    assert sum(rf["kernels_ms"].values()) <= out["ms_per_step"] * 1.02                       # dispatch times fit inside the step
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["sample"] and cb["one_core"]["value"] > 0
    assert set(out["configs"]) == {"cfg1", "cfg2", "cfg4", "matrix"}
    for name, c in out["configs"].items():
        assert c["parity_rms"] <= 1e-5 and c["msamples_per_s"] > 0 and 0.2 < c["path_frac"] < 1.0, (name, c)
        d = det["configs"][name]
        assert d["kernels_sum_ms"] <= d["ms_per_call"] * 1.02, (name, d["kernels_sum_ms"], d["ms_per_call"])
        for k in d["roofline"]["kernels"].values():
            assert k["ms"] > 0 and (k["event_ms"] is None or k["ms"] <= k["event_ms"] * 1.10 + 0.002), k    # an event behind the kernel holds the boundary too
    assert det["roofline"]["all_kernels"] and det["steady_state"]["steps"] == 400 and det["roofline_streaming"]["frac"] > 0.3


def test_bench_two_ranks_on_one_gpu(tmp_path):
    env = dict(os.environ, FOLVE_BENCH_DEVICE="0", FOLVE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29611"]
    out, det = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--streams", "8", "--blocks", "16", "--no-cpu-baseline"],
                         env, tmp_path, launcher=launcher)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    assert out["config"]["streams_per_gpu"] == 8 and out["config"]["total_streams"] == 16
    assert out["shards"] == {"rule": "gpu = stream mod N", "streams_per_rank": [8, 8], "first_of_rank": [0, 1]}
    assert det["shards"] == [[0, 2, 4, 6, 8, 10, 12, 14], [1, 3, 5, 7, 9, 11, 13, 15]]   # gpu = stream mod N
    assert out["parity_rms"] is not None and out["parity_rms"] <= 1e-5
    assert out["value"] > 0 and det["value"] == out["value"]


def test_bench_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it (what the driver's SCALE step may run): bench.py starts its two
    ranks as a child process itself, and the N = 2 line carries everything the N = 1 line does — roofline, cpu_baseline
    (timed by rank 0 after the process group is gone), shards.  Both ranks on device 0 over gloo: a one-GPU box."""
    env = dict(os.environ, FOLVE_BENCH_DEVICE="0", FOLVE_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out, det = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--streams", "8", "--blocks", "16", "--cpu-seconds", "2"], env, tmp_path)
    assert out["n_gpus"] == 2 and out["config"]["total_streams"] == 16 and out["value"] > 0
    assert det["shards"] == [[0, 2, 4, 6, 8, 10, 12, 14], [1, 3, 5, 7, 9, 11, 13, 15]]
    assert out["process_group"] == {"backend": "gloo", "world_size": 2, "forced_at_world_size_1": False}
    sp = det["stream_peaks"]
    assert sp["streams"] == 16 and 0.5 < sp["min_abs"] <= sp["max_abs"] < 20   # gathered over both ranks
    assert out["roofline"]["bound"] in ("hbm", "valu") and out["roofline"]["kernel_ms"] > 0 and out["roofline"]["frac_lower_bound"] > 0
    cpu = out["cpu_baseline"]
    assert cpu and cpu["value"] > 0 and cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["sample"]
    assert out["parity_rms"] <= 1e-5


def test_the_two_rank_line_of_the_benchmarked_shape_against_one_rank(tmp_path):
    """The command the driver's SCALE step runs, `python3 bench.py --gpus 2 --steps 20 --warmup 5`, on a one-GPU box (both
    ranks on device 0 over gloo): the line is small and complete (roofline, cpu_baseline, shards, process group), and the
    job's aggregate rate agrees with the N = 1 line's measured in the same session on the same device — two ranks of 32
    streams share the GPU that one rank of 64 streams has to itself, so value(N = 2) / 2 per rank = value(N = 1) / 2:
    the sharded path (barriers, max over ranks, two engines) costs nothing beside the one-process path.  The pool's own
    semantics: one processor per open file, whichever GPU it lands on (/root/reference/processor-pool.cc:48-91)."""
    env = dict(os.environ, FOLVE_BENCH_DEVICE="0", FOLVE_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    (tmp_path / "n1").mkdir(); (tmp_path / "n2").mkdir()
    common = ["--steps", "20", "--warmup", "5", "--no-extras", "--cpu-seconds", "2"]
    one, _ = run_bench(["--gpus", "1"] + common, env, tmp_path / "n1")
    two, det = run_bench(["--gpus", "2", "--streams", "32"] + common, env, tmp_path / "n2", timeout=1500)
    assert two["n_gpus"] == 2 and two["config"]["total_streams"] == 64 and two["config"]["blocks_per_step"] == 256
    assert two["process_group"]["world_size"] == 2 and two["shards"]["streams_per_rank"] == [32, 32]
    assert two["roofline"]["kernel_ms"] > 0 and two["roofline"]["frac_lower_bound"] > 0 and two["cpu_baseline"]["value"] > 0
    assert two["parity_rms"] <= 1e-5 and one["parity_rms"] <= 1e-5
    per_rank_share = two["value"] / 2
    assert abs(per_rank_share - one["value"] / 2) <= 0.10 * (one["value"] / 2), (one["value"], two["value"])


def test_bench_cfg5_shape_eight_ranks_512_streams_on_one_gpu(tmp_path):
    """BASELINE.json configs[4] through bench.py itself: `python bench.py --gpus 8` — eight ranks x 64 streams x 256 blocks,
    cfg5's 512 streams sharded gpu = stream mod 8 — with all eight ranks on device 0 over gloo (a one-GPU box: the rate it
    prints is eight processes sharing one GPU and means nothing; what is checked is the launch, the sharding, the parity
    gate on every rank at the full per-GPU shape, the barriers and reductions, and the line's completeness)."""
    env = dict(os.environ, FOLVE_BENCH_DEVICE="0", FOLVE_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out, det = run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env, tmp_path, timeout=1500)
    assert out["n_gpus"] == 8 and out["scaling"] == "weak"
    assert out["config"]["streams_per_gpu"] == 64 and out["config"]["total_streams"] == 512 and out["config"]["blocks_per_step"] == 256
    assert out["shards"]["streams_per_rank"] == [64] * 8 and out["shards"]["first_of_rank"] == list(range(8))
    assert len(det["shards"]) == 8 and all(len(s) == 64 for s in det["shards"])
    assert sorted(i for s in det["shards"] for i in s) == list(range(512)) and det["shards"][3][:3] == [3, 11, 19]
    assert out["process_group"] == {"backend": "gloo", "world_size": 8, "forced_at_world_size_1": False}
    assert out["parity_rms"] <= 1e-5 and out["value"] > 0 and out["roofline"]["kernel_ms"] > 0
    assert det["stream_peaks"]["streams"] == 512 and det["stream_peaks"]["min_abs"] > 0.5      # every stream's maximum arrived


def test_bench_rccl_branch_runs_on_one_gpu(tmp_path):
    """The RCCL branch of bench.py — init_process_group("nccl", device_id), the barriers around the timed region and the
    reductions of sharding.aggregate_throughput, beside the engine's private HIP streams — behind FOLVE_BENCH_FORCE_DIST=1
    at world size 1: everything of the N > 1 launch that one GPU can run."""
    env = dict(os.environ, FOLVE_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "FOLVE_BENCH_BACKEND"):
        env.pop(k, None)
    out, det = run_bench(["--steps", "3", "--warmup", "1", "--streams", "8", "--blocks", "16", "--no-cpu-baseline", "--no-extras"], env, tmp_path)
    assert out["process_group"] == {"backend": "nccl", "world_size": 1, "forced_at_world_size_1": True}
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["parity_rms"] <= 1e-5
    assert det["stream_peaks"]["streams"] == 8 and det["stream_peaks"]["min_abs"] > 0.5       # (the gather's reduction ran over RCCL)


def test_harness_over_two_router_slots_with_numa_placement(tmp_path):
    """bench.py's drop_in_threads_multi_gpu leg on a one-GPU box: the C++ harness with two router slots (both device 0),
    run-ahead on, NUMA placement on (a no-op where sysfs says nothing): the streams split evenly, both engines work."""
    exe = os.path.join(ROOT, "tools", "dropin", "dropin_threads")
    if not os.path.exists(exe):
        pytest.skip("tools/dropin/dropin_threads not built")
    sys.path.insert(0, os.path.join(ROOT, "tools", "dropin"))
    import make_conf
    conf = make_conf.write(str(tmp_path), 65536)
    env = dict(os.environ, FOLVE_AMD_DEVICES="0,0")
    r = subprocess.run([exe, conf, "8", "256", "1", "json", "run_ahead=8", "pin=1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["threads"] == 8 and out["run_ahead"] == 8 and out["numa_pin"]
    assert out["gpus"]["0"]["streams"] == 8                                    # (two slots, one physical device)
    assert out["blocks_per_s"] > 0 and out["requests"] < 8 * 256              # blocks travelled in chunks


def test_harness_verifies_the_duplex_pipeline_under_contention(tmp_path):
    """48 native file threads with run-ahead 32 — big combined batches, the engine's duplex DMA pipeline, per-block maxima
    from the GPU — and afterwards every thread's output and peak compared (verify=1) with the same file pulled one block per
    engine call by a single thread without the combiner."""
    exe = os.path.join(ROOT, "tools", "dropin", "dropin_threads")
    if not os.path.exists(exe):
        pytest.skip("tools/dropin/dropin_threads not built")
    sys.path.insert(0, os.path.join(ROOT, "tools", "dropin"))
    import make_conf
    conf = make_conf.write(str(tmp_path), 100000)
    r = subprocess.run([exe, conf, "48", "160", "1", "json", "run_ahead=32", "verify=1", "file_blocks=160"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["verified_rms"] is not None and out["verified_rms"] <= 2e-6
    assert out["largest_batch_blocks"] >= 512                                  # (>= 32 MB each way: the duplex path ran)
