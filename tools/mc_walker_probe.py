"""Dev aid: is the channel-pair walkers' rate (K1 / K3 of a lone 8-channel stream) the 8-byte strided PCM accesses or the
launch shape?  The same number of (block, channel) units as stereo streams (whole 16-byte quads), as one 8-channel stream,
and with the per-pair non-walking kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.configs import measure_config

for name, S, C, T, tune in (("4 stereo streams x 256", 4, 2, 256, None), ("1 x 8ch x 256 (walkers per pair)", 1, 8, 256, None),
                            ("1 x 8ch x 256 (fft_form 3: chpair kernels)", 1, 8, 256, {"fft_form": 3}),
                            ("16 stereo x 256", 16, 2, 256, None), ("4 x 8ch x 256", 4, 8, 256, None),
                            ("4 stereo x 1024", 4, 2, 1024, None), ("1 x 8ch x 1024", 1, 8, 1024, None)):
    r = measure_config(T=T, tune=tune, steps=200, check=False, S=S, C=C, size=524288)
    print("%-44s %.4f ms/call, kernels %s" % (name, r["ms_per_call"], {k: round(v * 1e3, 1) for k, v in r["kernels_ms"].items()}), flush=True)
