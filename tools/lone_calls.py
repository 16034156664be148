"""Dev aid: wall time per call of the lone-stream configurations (cfg1, cfg2, cfg4), asynchronous back-to-back calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.configs import measure_config
for name, kw in (("cfg1", dict(S=1, C=2, size=65536, T=323)), ("cfg2", dict(S=1, C=2, size=204800, T=256, populated=178193)),
                 ("cfg4", dict(S=1, C=8, size=524288, T=256)), ("cfg4 x1024", dict(S=1, C=8, size=524288, T=1024))):
    for rep in range(2):
        r = measure_config(tune=None, steps=400, check=False, **kw)
        print("%-12s %.4f ms/call  kernels(ev) %s" % (name, r["ms_per_call"], {k: round(v * 1e3, 1) for k, v in r["kernels_ms"].items()}), flush=True)
