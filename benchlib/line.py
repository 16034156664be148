"""The ONE line bench.py prints on stdout, and the file everything else goes to.

The driver keeps an 8 KB tail of stdout and parses its last line: a line longer than that is lost (round 5's 23.9 KB line
was: BENCH_r05.json `parsed: null`).  contract_line() keeps the contract's keys, `roofline`, `cpu_baseline` and one number
per other configuration — under 4 KB, bounded by tests/test_bench_cpu.py::test_contract_line_is_small — and names
bench_details.json (beside bench.py), which holds every per-kernel table, leg and note of the run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DETAILS_NAME = "bench_details.json"
LINE_LIMIT = 4096


def _pick(d, keys):
    return {k: d.get(k) for k in keys if isinstance(d, dict) and k in d}


def _short(text, n):
    text = text or ""
    return text if len(text) <= n else text[:n - 3] + "..."


def contract_line(full, details_name=DETAILS_NAME):
    """The compact dict of the contract line from the run's full result dict."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    cfg = full.get("config") or {}
    out["config"] = dict(_pick(cfg, ("streams_per_gpu", "total_streams", "channels", "taps", "block", "partitions", "blocks_per_step")),
                         workload=_short(cfg.get("workload"), 260), pcm=_short(cfg.get("pcm"), 80), sharding=_short(cfg.get("sharding"), 80))
    out.update(_pick(full, ("mframes_per_s", "realtime_streams", "parity_rms", "parity_rel")))
    rf = full.get("roofline") or {}
    line_rf = _pick(rf, ("bound", "kernel", "kernel_name", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                         "kernel_ms", "kernels_ms", "min_bytes_per_launch", "frac_lower_bound"))
    line_rf["time_source"] = "events bound to the dispatches, this run"
    if rf.get("traffic") is None and rf.get("traffic_note"):
        line_rf["traffic_note"] = _short(rf["traffic_note"], 200)
    line_rf["path"] = _pick(rf.get("path") or {}, ("frac_of_min_bytes",))
    if rf.get("k2_valu"):
        line_rf["k2_valu"] = _pick(rf["k2_valu"], ("tflops", "peak", "frac", "frac_at_sclk", "sclk_mhz"))
    out["roofline"] = line_rf
    st = full.get("steady_state")
    if st:
        pw = st.get("power") or {}
        out["steady_state"] = dict(_pick(st, ("steps", "ms_per_step", "msamples_per_s")), **_pick(pw, ("socket_w", "cap_w", "sclk_mhz")))
    e2e = full.get("end_to_end")
    if isinstance(e2e, dict) and e2e.get("msamples_per_s"):
        out["end_to_end"] = _pick(e2e, ("msamples_per_s", "pcie_GBs_each_way"))
    sb = full.get("single_block")
    if isinstance(sb, dict) and sb.get("single_block_us"):
        out["single_block_us"] = sb["single_block_us"]
    cfgs = full.get("configs")
    if isinstance(cfgs, dict):
        small = {}
        for name, c in cfgs.items():
            if not isinstance(c, dict) or "error" in c:
                small[name] = {"error": _short((c or {}).get("error", "?"), 80)}
                continue
            r = c.get("roofline") or {}
            p = r.get("path") or {}
            small[name] = {"msamples_per_s": c.get("msamples_per_s"), "ms_per_call": c.get("ms_per_call"),
                           "path_frac": p.get("frac") if p.get("frac") is not None else p.get("frac_of_min_bytes"),
                           "parity_rms": float("%.3g" % c["parity_rms"]) if c.get("parity_rms") is not None else None,
                           "bound": r.get("bound"),
                           "k2_valu_frac": (r.get("k2_valu") or {}).get("frac")}
        out["configs"] = small
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line_cb = _pick(cb, ("value", "unit", "cores", "kind"))
        line_cb["what"] = "CPU restatement of zita-convolver's one-level 8192-frame partitioned algorithm, vectorised (oracle/fastcpu.c); zita itself is not on the box"
        if (cb.get("zita_convolver_on_this_box") or {}).get("available"):
            line_cb["what"] = "CPU restatement of zita-convolver's algorithm (oracle/fastcpu.c); the real libzita-convolver is timed beside it"
        line_cb["sample"] = _short(cb.get("sample"), 120)
        line_cb["one_core"] = _pick(cb.get("one_core") or {}, ("value",))
        z = cb.get("zita_convolver_on_this_box") or {}
        line_cb["zita_convolver_on_this_box"] = _pick(z, ("available", "value", "cores"))
        out["cpu_baseline"] = line_cb
    else:
        out["cpu_baseline"] = None
    if full.get("shards") is not None:
        sh = full["shards"]
        out["shards"] = {"rule": "gpu = stream mod N", "streams_per_rank": [len(x) for x in sh], "first_of_rank": [x[0] if x else None for x in sh]}
    if full.get("process_group") is not None:
        out["process_group"] = full["process_group"]
    out["details"] = details_name
    return out


def render(full, details_name=DETAILS_NAME):
    """The line as text.  Should it ever exceed the limit (it does not for any shape the tests build), optional keys go,
    most dispensable first; the contract keys, `roofline` and `cpu_baseline` always stay."""
    out = contract_line(full, details_name)
    line = json.dumps(out, separators=(", ", ": "))
    for drop in ("single_block_us", "end_to_end", "steady_state", "process_group", "shards", "configs"):
        if len(line) < LINE_LIMIT:
            break
        out.pop(drop, None)
        out["dropped_for_length"] = out.get("dropped_for_length", []) + [drop]
        line = json.dumps(out, separators=(", ", ": "))
    return line


def write_details(full, path=None):
    """Everything the run measured, as indented JSON beside bench.py (or in the temp directory if the tree is read-only).
    Returns the path written, or None."""
    import tempfile
    for p in ([path] if path else [os.path.join(ROOT, DETAILS_NAME), os.path.join(tempfile.gettempdir(), DETAILS_NAME)]):
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
                f.write("\n")
            return p
        except OSError:
            continue
    return None


def emit(full, details_path=None):
    """Write the details file, echo the details to stderr (one line, for logs that keep stderr), print the contract line."""
    p = write_details(full, details_path)
    name = DETAILS_NAME if (p and os.path.dirname(p) == ROOT) else (p or "(not written)")
    try:
        sys.stderr.write("bench.py details (%s): %s\n" % (name, json.dumps(full)))
    except Exception:  # noqa: BLE001
        pass
    line = render(full, name)
    print(line)
    sys.stdout.flush()
    return line
