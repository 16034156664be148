"""Runs in a FRESH process (tests/test_multi_gpu.py::test_an_open_file_moves_at_every_depth_and_position): two router slots on
device 0 (FOLVE_AMD_DEVICES=0,0).  One file thread converts a file through folve::SoundProcessor; after a chosen number of
blocks have been handed out, the engine it runs on starts failing every launch round (FE_TUNE_FAIL_NEXT = -1).  The file must
come out equal to the float64 convolution — no block of silence, the same peak — with exactly one move, whatever the run-ahead
depth (1: Process() block by block, the K blocks of state replayed through a stream opened for one-block calls; 4: chunks
shorter than K, kept in the host ring; 64: long chunks kept in place) and wherever the failure falls (the very first call:
nothing to replay; the ramp; the steady state; the file's short last block, which goes through Process()).  A 2 -> 3 channel
configuration (input and output of different widths) beside the stereo one.  The reference never emits silence from Process()
(/root/reference/sound-processor.cc:98-127).  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    tmp = sys.argv[1]
    assert os.environ.get("FOLVE_AMD_DEVICES") == "0,0"
    import folve_amd.capi as capi
    import folve_amd.host as H
    from fixtures import make_santalucia_shaped_dir, seeded_input
    from oracle import oracle as O
    L = H._L()
    L.fh_router_health_policy(3, 0.05)
    d, hs = make_santalucia_shaped_dir(os.path.join(tmp, "sl"))
    stereo = os.path.join(d, "filter-44100.conf")
    os.makedirs(os.path.join(tmp, "m"))
    matrix = os.path.join(tmp, "m", "filter-44100.conf")
    with open(matrix, "w") as f:
        f.write("/convolver/new 2 3 256 30000\n/impulse/dirac 1 1 0.5 0\n/impulse/dirac 2 2 0.25 17000\n"
                "/impulse/dirac 1 3 1.0 29999\n/impulse/dirac 2 3 -0.5 3\n")

    def matrix_ref(x):
        exp = np.zeros((len(x), 3))
        exp[:, 0] = 0.5 * x[:, 0]
        exp[17000:, 1] = 0.25 * x[:-17000, 1]
        exp[29999:, 2] += x[:-29999, 0]
        exp[3:, 2] += -0.5 * x[:-3, 1]
        return exp

    cases = []
    nblocks, short = 70, 1234
    for conf, ch, name in ((stereo, 2, "stereo K=25"), (matrix, 2, "2->3 K=4")):
        for depth in (1, 4, 64):
            for trigger in (0, 2, 40, nblocks):                  # blocks handed out before the engine dies (nblocks: only the short last block is left)
                H.set_run_ahead(depth)
                p = H.SoundProcessor.create(conf, 44100, ch)
                assert p is not None
                x = seeded_input(7 + depth + trigger, nblocks * 8192 + short, ch)
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
                ref = O.linear_convolution_f64(x, hs, 2) if conf == stereo else matrix_ref(x)
                silent = sum(1 for b in range(0, len(y), 8192) if not y[b:b + 8192].any())
                cases.append({"config": name, "depth": depth, "trigger": trigger, "rms": float(O.rms(y - ref)),
                              "moves": int(L.fh_processor_moves(p.h)), "ok": int(L.fh_processor_ok(p.h)),
                              "engine_changed": int(L.fh_processor_engine(p.h)) != eng0, "silent_blocks": silent,
                              "peak_err": abs(p.max_output_value() - max(0.0, float(y.max()))), "run_ahead": p.run_ahead()})
                p.close()
                # back in service for the next case: past the re-probe interval, a probe that answers
                import time
                time.sleep(0.12)
                q = H.SoundProcessor.create(conf, 44100, ch)
                assert q is not None
                q.close()
    print("SURVIVE_JSON " + json.dumps({"cases": cases}))


if __name__ == "__main__":
    main()
