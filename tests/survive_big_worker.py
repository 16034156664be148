"""Runs in a FRESH process (tests/test_multi_gpu.py::test_a_file_beyond_the_old_history_budget_moves_and_a_stale_one_does_not):
two router slots on device 0 (FOLVE_AMD_DEVICES=0,0).

(a) A 16-channel stream through a 2^20-tap configuration (K = 128, the reference's MAXSIZE, zita-config.h:61): its input
    history — 2K + 2 = 258 blocks x 8192 frames x 16 channels = 135 MB — is beyond the 64 MB the history was budgeted with until
    round 6, when such a file fell back to silence on a GPU failure.  The engine under it starts
    failing at block 150: the file must move — once — and come out equal to its closed form (every path a dirac:
    y_c[n] = g_c x_c[n - d_c], delays up to 1 048 575 frames = the last tap of the last partition), no silent block.
    The reference never emits silence from Process() (/root/reference/sound-processor.cc:98-127).
(b) A stereo file whose configuration is edited (other taps, newer mtime) while it is open, then loses its GPU: the move is
    REFUSED (the other GPU would be given the new taps under the old key) — silence from there, ok() == false, and the next
    open of that configuration gets the new taps.
Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    tmp = sys.argv[1]
    assert os.environ.get("FOLVE_AMD_DEVICES") == "0,0"
    import folve_amd.capi as capi
    import folve_amd.host as H
    from fixtures import seeded_input
    L = H._L()
    L.fh_router_health_policy(3, 0.05)
    out = {}

    # ---- (a) 16 channels, K = 128 ----
    C, size, P = 16, 1 << 20, 8192
    delays = [0, 1, 8191, 8192, 100000, 262144, 524287, 524288, 777777, 1000000, 1048575, 5, 300000, 65536, 912345, 1048574]
    gains = [0.5 + 0.03125 * c for c in range(C)]
    d = os.path.join(tmp, "big")
    os.makedirs(d)
    conf = os.path.join(d, "filter-96000.conf")
    with open(conf, "w") as f:
        f.write("/convolver/new %d %d 256 %d\n" % (C, C, size))
        for c in range(C):
            f.write("/impulse/dirac %d %d %.6f %d\n" % (c + 1, c + 1, gains[c], delays[c]))
    H.set_run_ahead(8)                                   # chunks far shorter than K: every block goes through the host ring
    p = H.SoundProcessor.create(conf, 96000, C)
    assert p is not None
    nblocks, short, trigger = 180, 999, 150
    x = seeded_input(31, nblocks * P + short, C)
    eng0 = int(L.fh_processor_engine(p.h))
    outs, done, blocks_out, killed = [], 0, 0, False
    while done < len(x):
        if not killed and blocks_out >= trigger:
            assert L.fe_engine_set_tuning(eng0, capi.FE_TUNE_FAIL_NEXT, -1) == 0
            killed = True
        r = p.fill_buffer(x[done:])
        assert r > 0
        outs.append(p.write_processed(r))
        done += r
        blocks_out += 1
    assert L.fe_engine_set_tuning(eng0, capi.FE_TUNE_FAIL_NEXT, 0) == 0
    y = np.concatenate(outs, 0)
    ref = np.zeros_like(x, dtype=np.float64)
    for c in range(C):
        n = len(x) - delays[c]
        if n > 0:
            ref[delays[c]:, c] = np.float32(gains[c]) * x[:n, c].astype(np.float64)
    err = y - ref
    out["big"] = {"channels": C, "partitions": 128, "history_bytes": 258 * P * C * 4, "blocks": nblocks, "trigger": trigger,
                  "rms": float(np.sqrt(np.mean(err * err))), "max_err": float(np.abs(err).max()),
                  "moves": int(L.fh_processor_moves(p.h)), "ok": int(L.fh_processor_ok(p.h)),
                  "engine_changed": int(L.fh_processor_engine(p.h)) != eng0,
                  "silent_blocks": sum(1 for b in range(0, len(y), P) if not y[b:b + P].any()),
                  "peak_err": abs(p.max_output_value() - max(0.0, float(y.max()))), "run_ahead": p.run_ahead()}
    p.close()
    time.sleep(0.12)

    # ---- (b) the configuration changes under an open file, then its GPU fails ----
    d2 = os.path.join(tmp, "stale")
    os.makedirs(d2)
    conf2 = os.path.join(d2, "filter-44100.conf")

    def write_conf(g):
        with open(conf2, "w") as f:
            f.write("/convolver/new 2 2 256 20000\n/impulse/dirac 1 1 %.3f 0\n/impulse/dirac 2 2 %.3f 12345\n" % (g, g))
    write_conf(0.5)
    old = time.time() - 100
    os.utime(conf2, (old, old))
    H.set_run_ahead(4)
    q = H.SoundProcessor.create(conf2, 44100, 2)
    assert q is not None
    x2 = seeded_input(5, 40 * P, 2)
    eng1 = int(L.fh_processor_engine(q.h))
    outs, done, blocks_out = [], 0, 0
    while done < len(x2):
        if blocks_out == 10:
            write_conf(0.25)                              # an edit: other taps, a newer mtime
            assert L.fe_engine_set_tuning(eng1, capi.FE_TUNE_FAIL_NEXT, -1) == 0
        r = q.fill_buffer(x2[done:])
        assert r > 0
        outs.append(q.write_processed(r))
        done += r
        blocks_out += 1
    assert L.fe_engine_set_tuning(eng1, capi.FE_TUNE_FAIL_NEXT, 0) == 0
    y2 = np.concatenate(outs, 0)
    good = 0.5 * x2[:, 0]
    first_bad = next((b for b in range(40) if np.abs(y2[b * P:(b + 1) * P, 0] - good[b * P:(b + 1) * P]).max() > 1e-5), 40)
    tail = y2[first_bad * P:]
    out["stale"] = {"moves": int(L.fh_processor_moves(q.h)), "ok": int(L.fh_processor_ok(q.h)), "first_bad_block": first_bad,
                    "tail_is_silence": bool(not tail.any()), "config_up_to_date": bool(q.config_still_up_to_date())}
    q.close()
    time.sleep(0.12)
    r2 = H.SoundProcessor.create(conf2, 44100, 2)         # the next open: the edited configuration
    assert r2 is not None
    z = []
    done = 0
    xs = x2[:3 * P]
    while done < len(xs):
        n = r2.fill_buffer(xs[done:])
        z.append(r2.write_processed(n))
        done += n
    z = np.concatenate(z, 0)
    out["stale"]["next_open_gain_err"] = float(np.abs(z[:, 0] - 0.25 * xs[:, 0]).max())
    r2.close()
    print("SURVIVE_BIG_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
