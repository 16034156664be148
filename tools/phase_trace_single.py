"""Phase cycles of the walkers inside the ONE-block call (dev aid; TRACE build: make -C folve_amd/csrc TRACE=1).
fft_form = 2 pins the walkers so that K1 is the instrumented forward_walker too."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FOLVE_AMD_LIB", os.path.join(ROOT, "folve_amd", "libfolve_amd_trace.so"))
sys.path.insert(0, ROOT)
import numpy as np
import folve_amd as fa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
form = int(sys.argv[2]) if len(sys.argv) > 2 else 2        # 2: walkers everywhere (phases); 0: the product's choice (stamps)
size, C = 262144, 2
eng = fa.Engine(0)
eng.set_tuning(fft_form=form)
flt = fa.Filter(eng, C, C, size)
rng = np.random.default_rng(3)
for c in range(C):
    h = rng.standard_normal(size).astype(np.float32); h /= np.linalg.norm(h)
    flt.add(c, c, h)
flt.commit()
P = flt.block_size
L = fa.lib()
L.fe_debug_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = ctypes.c_void_p()
assert L.fe_host_alloc(P * C * 4, ctypes.byref(buf)) == 0
st = flt.open_stream(1)
assert L.fe_stream_bind_host_buffer(st.h, buf, P * C * 4) == 0
arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_float)), shape=(P * C,))
arr[:] = rng.uniform(-1, 1, P * C).astype(np.float32)
for _ in range(40):
    L.fe_stream_process(st.h, buf, P, buf, None, None)
ph = (ctypes.c_ulonglong * 16)()
assert L.fe_debug_phases(ph, 1) == 0
hostt = (ctypes.c_ulonglong * 8)()
L.fe_debug_host_times.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
L.fe_debug_host_times(hostt, 1)
t0 = time.perf_counter()
for _ in range(n):
    L.fe_stream_process(st.h, buf, P, buf, None, None)
dt = (time.perf_counter() - t0) / n
assert L.fe_debug_phases(ph, 1) == 0
print("%.1f us per block (instrumented build)" % (dt * 1e6))
L.fe_debug_host_times(hostt, 1)
nc = max(1, hostt[2])
print("host: entry -> three launches enqueued %.2f us | -> completion seen %.2f us (%.1f polls) | return -> next entry (caller's loop) %.2f us"
      % (hostt[0] / nc / 1e3, hostt[1] / nc / 1e3, hostt[3] / nc, hostt[4] / max(1, nc - 1) / 1e3))
names = [
    ("forward_walker", ["PCM wait + stage A", "barrier", "stage B", "barrier", "split + stores", "barrier", "prologue: tables -> LDS", "prologue: job descriptor"]),
    ("inverse_walker", ["Y wait + fold", "prefetch + stage A", "barrier", "stage B", "barrier", "read + stores", "barrier", "prologue"]),
]
for k, (kn, phn) in enumerate(names):
    v = [ph[k * 8 + i] for i in range(len(phn))]
    print("%s: %.0f ticks of 10 ns per call, all workgroups of the launch" % (kn, sum(v) / n))
    for nm, c in zip(phn, v):
        print("   %-20s %8.0f x 10 ns" % (nm, c / n))

# the GPU's own timeline of the last calls: stamps 1/2 = forward_dual entry/exit, 3/4 = mac_small, 5/6 = inverse_walker
L.fe_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_uint)]
ring = (ctypes.c_ulonglong * 256)(); cnt = ctypes.c_uint()
assert L.fe_debug_stamps(ring, ctypes.byref(cnt)) == 0
seq = [ring[i & 255] for i in range(max(0, cnt.value - 240), cnt.value)]
ev = sorted(((v & ((1 << 56) - 1)) * 0.01, v >> 56) for v in seq)       # (us, id)
import statistics as stt
pairs = {}
for i in range(len(ev) - 1):
    pairs.setdefault((ev[i][1], ev[i + 1][1]), []).append(ev[i + 1][0] - ev[i][0])
print("GPU timeline, stamp -> next stamp (median us over the last calls; 1/2 K1 entry/exit, 3/4 K2, 5/6 K3, 10+ inside K1):")
for k in sorted(pairs, key=lambda k: (k[0] if k[0] < 10 else 1.5 + k[0] / 100.0)):
    if len(pairs[k]) >= 5:
        print("   %2d -> %2d  %6.2f us  (n=%d)" % (k[0], k[1], stt.median(pairs[k]), len(pairs[k])))
