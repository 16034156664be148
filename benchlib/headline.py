"""The headline workload — BASELINE.json configs[2] / cfg5 per GPU: 64 concurrent 44.1 kHz stereo streams through one shared
2-path 262 144-tap FIR, PCM resident in HBM, one step = one batched pass of K1 -> K2 -> K3 over `blocks` consecutive
blocks of every stream.  Setup, the parity gate, the timed region (the contract's barrier + synchronize on both sides, max
over ranks), and the roofline block of the dominant kernel."""
import os
import sys
import time

import numpy as np

from .configs import ROLE_NAMES, choose_bound, profile_kernels
from .formulas import HBM_PEAK_GBS, PARITY_TOL, alg_bytes, conv_f64, rms, tiled_bytes, valu_fractions, walk_flops
from .power import PowerWatch
from .profiles import load_traffic, profile_applies


def parse_tune(text):
    return {k: int(v) for k, v in (kv.split("=") for kv in text.split(","))} if text else None


class Headline:
    def __init__(self, args, world, rank, dev, dist, red_dev):
        import torch
        import folve_amd as fa
        from folve_amd import sharding
        from folve_amd.capi import BatchPlan, FE_ASYNC, FE_DEVICE_PTRS
        self.torch, self.fa, self.sharding = torch, fa, sharding
        self.args, self.world, self.rank, self.dev, self.dist, self.red_dev = args, world, rank, dev, dist, red_dev
        # streams are sharded by index, as the pool hands out processors: gpu = stream % world
        self.my_streams = sharding.shard_streams(args.streams * world, world, rank)
        assert len(self.my_streams) == args.streams
        S, T, C, size = args.streams, args.blocks, args.channels, args.taps
        self.S, self.T, self.C, self.size = S, T, C, size
        self.ts = torch.cuda.Stream()
        self.eng = fa.Engine(dev, self.ts.cuda_stream)
        if args.tune:
            self.eng.set_tuning(**parse_tune(args.tune))
        self.flt = fa.Filter(self.eng, C, C, size)
        self.P, self.K = self.flt.block_size, self.flt.partitions
        rng = np.random.default_rng(3)
        self.taps = []
        for c in range(C):                               # one shared filter, C diagonal paths, unit L2 norm
            h = rng.standard_normal(size).astype(np.float32)
            h /= np.linalg.norm(h)
            self.taps.append(h)
            self.flt.add(c, c, h)
        self.flt.commit()
        self.streams = [self.flt.open_stream(T) for _ in range(S)]
        P = self.P
        with torch.cuda.stream(self.ts):
            self.xs, self.ys = [], []
            for s in range(S):
                g = torch.Generator(device="cuda")
                g.manual_seed(100 + self.my_streams[s])
                self.xs.append(torch.rand(T * P, C, device="cuda", generator=g) * 2 - 1)   # U(-1, 1)
                self.ys.append(torch.empty(T * P, C, device="cuda"))
        self.plan = BatchPlan(self.streams, [x.data_ptr() for x in self.xs], [y.data_ptr() for y in self.ys], [T * P] * S,
                              FE_DEVICE_PTRS | FE_ASYNC)
        self.y1 = {}
        self.kms = self.event_ms = self.launched = None
        self.sclk_mhz = None

    def sync(self):
        self.eng.synchronize()
        self.torch.cuda.synchronize()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    # ---- parity gate: nothing is timed unless the benchmarked launch is right ----
    def parity_gate(self):
        """Step 1 runs from zeroed state, step 2 carries it: outputs of two streams against the float64 linear
        convolution of [x | x] with the taps.  The references are computed FIRST (seconds of CPU work), so that the GPU
        does not sit idle between the gate's two steps and the warm-up.  Returns (abs rms, relative rms, streams checked),
        max over ranks."""
        S = self.S
        check = sorted({0, S - 1})
        self.sync()
        refs = {}
        for s in check:
            x = self.xs[s].cpu().numpy()
            refs[s] = conv_f64(np.concatenate([x, x]), self.taps)
        self.plan.run(); self.sync()
        y1 = {s: self.ys[s].cpu().numpy().copy() for s in check}
        self.plan.run(); self.sync()
        y2 = {s: self.ys[s].cpu().numpy().copy() for s in check}
        self.y1 = y1
        parity_abs, parity_rel = 0.0, 0.0
        for s in check:
            got = np.concatenate([y1[s], y2[s]])
            e = rms(got - refs[s])
            parity_abs = max(parity_abs, e)
            parity_rel = max(parity_rel, e / rms(refs[s]))
        del refs
        if self.dist is not None:
            t = self.torch.tensor([parity_abs, parity_rel], dtype=self.torch.float64, device=self.red_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            parity_abs, parity_rel = float(t[0]), float(t[1])
        return parity_abs, parity_rel, check

    def gate_or_exit(self):
        import json
        parity_abs, parity_rel, check = self.parity_gate()
        if not (parity_abs <= PARITY_TOL and parity_rel <= PARITY_TOL):
            if self.rank == 0:
                print(json.dumps({"error": "parity gate failed", "parity_rms": parity_abs, "parity_rel": parity_rel,
                                  "tolerance": PARITY_TOL}))
            sys.stderr.write("bench.py: PARITY GATE FAILED (rms %.3e, rel %.3e > %.0e): nothing was timed\n"
                             % (parity_abs, parity_rel, PARITY_TOL))
            sys.exit(1)
        return parity_abs, parity_rel, check

    # ---- the timed region of the contract ----
    def timed(self):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; seconds, max over ranks."""
        a = self.args
        for _ in range(a.warmup):
            self.plan.run()
        self.sync(); self.barrier(); self.sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            self.plan.run()
        self.sync(); self.barrier(); self.sync()
        dt = time.perf_counter() - t0
        _, dt, _ = self.sharding.aggregate_throughput(self.S * self.T * self.P * a.steps, dt, self.dist, self.red_dev)   # max over ranks
        return dt

    def stream_peaks(self):
        """The hot path's one metric, max_output_value() (/root/reference/sound-processor.cc:116-125), of every stream of the
        job in global stream order: each rank's K3 keeps its streams' running maxima on the GPU; the ranks' shards are
        disjoint, so one sum-reduction of a 64 N-vector is the gather (the only inter-GPU traffic besides the barrier)."""
        self.sync()
        pk = [st.peaks() for st in self.streams]
        return self.sharding.gather_stream_values(self.my_streams, [p_[1] for p_ in pk], self.S * self.world, self.dist, self.red_dev)

    def steady_state(self, nlong=400):
        """The same loop once more, long enough for the GPU's clocks to settle (a 20-step region is over in 45 ms), with
        socket power and shader clock: reported beside `value`, never instead of it."""
        for _ in range(50):
            self.plan.run()
        self.sync()
        watch = PowerWatch(self.dev)
        tl = time.perf_counter()
        with watch:
            for _ in range(nlong):
                self.plan.run()
            self.sync()
        dl = (time.perf_counter() - tl) / nlong
        pw = watch.summary()
        if pw:
            self.sclk_mhz = pw.get("sclk_mhz")
        return {"steps": nlong, "ms_per_step": round(dl * 1e3, 4), "msamples_per_s": round(self.S * self.T * self.P * self.C / dl / 1e6, 1),
                "power": pw,
                "note": "same launches, 400 steps after 50 more warm-up steps: the timed region above is too short "
                        "for the clocks to settle"}

    # ---- the roofline block ----
    def roofline(self, dt):
        """The dominant kernel against the roof that binds it.  Kernel times: this run's, events bound to the dispatches.
        HBM bytes per launch: PMC counters cannot be read from inside this process, so they come from the committed
        rocprofv3 --pmc passes of this same command (profiles/traffic.json, written by tools/profile_all.sh with the
        profile's tag and its kernel-trace averages), used only for the same kernels (by name) at times that agree (15 %)."""
        a, S, T, C, P, K, world = self.args, self.S, self.T, self.C, self.P, self.K, self.world
        units = S * C * T                                    # block-channels one launch processes
        ab, tb = alg_bytes(P, K, S), tiled_bytes(P, K, T)
        watch = PowerWatch(self.dev)
        with watch:
            kms, event_ms = profile_kernels(self.eng, self.plan.run, max(a.steps, 200))
            self.sync()
        if self.sclk_mhz is None and watch.summary():
            self.sclk_mhz = watch.summary().get("sclk_mhz")
        launched = self.eng.last_kernels()
        self.kms, self.event_ms, self.launched = kms, event_ms, launched
        dominant = max(kms, key=kms.get)
        shape_key = "S%d_T%d_K%d_C%d" % (S, T, K, C)
        entry = load_traffic().get(shape_key) or {}
        traffic = (entry.get("bytes") or {}).get(dominant)
        traffic_note = None
        if traffic is not None:
            ok, traffic_note = profile_applies(entry, launched, kms, 0.15, 0.0, roles=[dominant])
            if not ok:
                sys.stderr.write("bench.py: WARNING " + str(traffic_note) + "\n")
                traffic = None
        applies = {k: profile_applies(entry, launched, kms, 0.15, 0.0, roles=[k])[0] for k in kms}   # per kernel: same name, same time
        t_dom = kms[dominant] * 1e-3
        achieved = (traffic / t_dom / 1e9) if traffic else None
        floor = tb[dominant] * units / t_dom / 1e9 / HBM_PEAK_GBS
        valu = valu_fractions(walk_flops(launched.get("mac"), P, K, T, S * C, 1), kms["mac"], self.sclk_mhz)
        hbm_frac = (achieved / HBM_PEAK_GBS) if achieved else floor
        roofline = {
            "bound": choose_bound(hbm_frac, valu) if dominant == "mac" else "hbm",
            "kernel": ROLE_NAMES[dominant], "kernel_name": launched.get(dominant),
            "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
            "traffic": traffic, "traffic_source": entry.get("profile"), "traffic_note": traffic_note or entry.get("note"),
            "kernel_ms": round(kms[dominant], 4), "kernels_ms": {k: round(v, 4) for k, v in kms.items()},
            "time_source": "events bound to the dispatches in this run (hipExtLaunchKernelGGL start / stop: the packet's own begin-to-end)",
            "event_ms": {k: round(v, 4) for k, v in event_ms.items()},
            "kernels_launched": launched,
            "profiled_kernel": (entry.get("kernels") or {}).get(dominant),
            "profile_trace_us": {k: round(v / 1e3, 1) for k, v in (entry.get("avg_ns") or {}).items()},
            "min_bytes_per_launch": int(tb[dominant] * units),
            # never "no roofline": when the committed PMC traffic does not apply to this run (`frac` null, the reason in
            # traffic_note) the fraction by the minimum bytes the launch must move still bounds the true one from below
            "frac_lower_bound": round(floor, 4),
            "frac_lower_bound_why": ("`frac` is null: " + (traffic_note or "profiles/traffic.json has no entry for shape " + shape_key)
                                     + "; this is min_bytes_per_launch / kernel time / peak, a floor of the true fraction")
            if traffic is None else "counter bytes are in use: `frac` is the measured fraction, this its floor",
            "frac_alg": {"applicable": T == 1,
                         "value": round(ab[dominant] * units / t_dom / 1e9 / HBM_PEAK_GBS, 4),
                         "why": "SURVEY.md 8(d)'s streaming formula re-reads K spectra per output block; a "
                                "run-ahead call re-uses them on chip, so this figure is not a roofline fraction"},
            "k2_valu": valu,
            "all_kernels": {k: {"ms": round(kms[k], 4), "event_ms": round(event_ms[k], 4), "kernel": launched.get(k),
                                "traffic": (entry.get("bytes") or {}).get(k) if applies[k] else None,
                                "frac": round((entry.get("bytes") or {}).get(k, 0) / (kms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                if applies[k] and (entry.get("bytes") or {}).get(k) else None,
                                "frac_of_min_bytes": round(tb[k] * units / (kms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                            for k in kms},
            "path": {"min_bytes_per_block_channel": int(tb["total"]),
                     "frac_of_min_bytes": round(tb["total"] * units * world * a.steps / dt / 1e9 / (HBM_PEAK_GBS * world), 4)}}
        self._entry, self._applies = entry, applies
        return roofline

    def measured_hbm(self):
        """What this GPU's HBM gives ANY kernel, reads and writes apart (fe_engine_hbm_rates2: plain 16-byte streaming
        kernels over 2 GiB, HIP events), and each kernel's time against its own bytes at those rates.  The best float4
        copy found on this pool's boxes (tools/micro/copy_rate.hip, profiles/r06_copy_rate.txt) is 5.99 TB/s, non-temporal
        on both sides; MI355X_MICROARCH.md:36 quotes 6.29."""
        try:
            rates = self.eng.hbm_rates3(1 << 31, 20)
            entry, applies, kms = self._entry, self._applies, self.kms
            rd, wr = (entry.get("read") or {}), (entry.get("write") or {})
            wrate = max(rates["write"], rates["write_regions"])
            crate = max(rates["copy"], rates["copy_regions"], rates["copy_best"])
            model = {}
            for k in kms:
                if applies[k] and rd.get(k) and wr.get(k):
                    t_model = rd[k] / (rates["read"] * 1e9) + wr[k] / (wrate * 1e9)
                    t_copy = (rd[k] + wr[k]) / (crate * 1e9)
                    t_guide = (rd[k] + wr[k]) / 6.29e12
                    model[k] = {"model_ms": round(t_model * 1e3, 4), "frac": round(t_model / (kms[k] * 1e-3), 4),
                                "at_copy_rate_ms": round(t_copy * 1e3, 4), "frac_at_copy_rate": round(t_copy / (kms[k] * 1e-3), 4),
                                "frac_of_6290_GBs": round(t_guide / (kms[k] * 1e-3), 4)}
            return {"read_GBs": round(rates["read"], 1), "write_GBs": round(rates["write"], 1), "copy_GBs": round(rates["copy"], 1),
                    "write_own_regions_GBs": round(rates["write_regions"], 1), "copy_own_regions_GBs": round(rates["copy_regions"], 1),
                    "copy_best_GBs": round(rates["copy_best"], 1),
                    "what": "plain streaming kernels on this GPU in this run: 16 bytes per lane over 2 GiB, 20 passes, HIP events "
                            "(copy counts bytes read + written); *_own_regions: every workgroup its own contiguous region; copy_best: non-temporal loads "
                            "and stores in 4 KiB bursts per wave, the best shape of tools/micro/copy_rate.hip",
                    "kernel_time_at_these_rates": model or None,
                    "frac_meaning": "frac: (PMC read bytes / read rate + PMC write bytes / the better write rate) / kernel time; "
                                    "frac_at_copy_rate: PMC bytes / the best of this run's copy rates / kernel time; "
                                    "frac_of_6290_GBs: PMC bytes / 6.29 TB/s (MI355X_MICROARCH.md:36) / kernel time"}
        except Exception as ex:                                  # a measurement aid: never fails the bench line
            return {"error": str(ex)}

    # ---- streaming form (one block per stream per call = SoundProcessor::Process granularity): here K2
    # really streams K spectra per block, so algorithmic and moved bytes coincide ----
    def streaming(self):
        from folve_amd.capi import BatchPlan, FE_ASYNC, FE_DEVICE_PTRS
        a, S, C, P, K = self.args, self.S, self.C, self.P, self.K
        ab = alg_bytes(P, K, S)
        st1 = [self.flt.open_stream(1) for _ in range(S)]
        plan1 = BatchPlan(st1, [x.data_ptr() for x in self.xs], [y.data_ptr() for y in self.ys], [P] * S,
                          FE_DEVICE_PTRS | FE_ASYNC)
        for _ in range(K + 2):
            plan1.run()
        self.sync()
        n1 = max(50, a.steps * 2)
        t1 = time.perf_counter()
        for _ in range(n1):
            plan1.run()
        self.sync()
        d1 = (time.perf_counter() - t1) / n1
        k1ms, ev1 = profile_kernels(self.eng, plan1.run, 50)
        self.sync()
        mac1_gbs = ab["mac"] * S * C / (k1ms["mac"] * 1e-3) / 1e9
        e1 = load_traffic().get("S%d_T1_K%d_C%d" % (S, K, C)) or {}
        launched1 = self.eng.last_kernels()
        ok1, note1 = profile_applies(e1, launched1, k1ms, 0.20, 3.0, roles=["mac"])
        out = {"bound": "hbm", "kernel": "K2 mac (one block per call)", "blocks_per_call": 1,
               "achieved": round(mac1_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(mac1_gbs / HBM_PEAK_GBS, 4), "alg_bytes_per_launch": int(ab["mac"] * S * C),
               "traffic": (e1.get("bytes") or {}).get("mac") if ok1 else None, "traffic_source": e1.get("profile"), "traffic_note": note1,
               "kernels_launched": launched1,
               "kernel_ms": round(k1ms["mac"], 4), "kernels_ms": {k: round(v, 4) for k, v in k1ms.items()},
               "event_ms": {k: round(v, 4) for k, v in ev1.items()},
               "ms_per_step": round(d1 * 1e3, 4), "msamples_per_s": round(S * P * C / d1 / 1e6, 1),
               "path_frac": round(ab["total"] * S * C / d1 / 1e9 / HBM_PEAK_GBS, 4)}
        for s_ in st1:
            s_.close()
        return out


def init_process_group(world, rank, dev, backend, force_dist):
    """torch.distributed (RCCL; gloo for the one-GPU exercises) for the barrier, the max over ranks and bookkeeping only."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if force_dist and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(so.getsockname()[1])
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist
