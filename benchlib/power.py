"""What the box gives while a loop runs: socket power and shader clock of one GPU, and the CPUs this process may use."""
import os


class PowerWatch:
    """Socket power and shader clock of one GPU while a loop runs: a thread reading the amdgpu hwmon files
    (power1_input, power1_cap, freq1_input).  The device is found by PCI address; failing that, the
    busiest amdgpu card at sampling time.  Informational: says whether a step runs against the power cap."""

    def __init__(self, device):
        import glob
        self.dirs = []
        try:
            import torch
            pr = torch.cuda.get_device_properties(device)
            pat = "/sys/bus/pci/devices/%04x:%02x:%02x.0/hwmon/hwmon*" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            self.dirs = glob.glob(pat)
        except Exception:
            self.dirs = []
        self.by_address = bool(self.dirs)
        if not self.dirs:
            self.dirs = [d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
                         if os.path.exists(os.path.join(d, "power1_input"))]
        self.samples = []
        self._stop = None
        self._thread = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return int(f.read().strip())
        except Exception:
            return None

    def _sample(self):
        best = None
        for d in self.dirs:
            pw = self._read(os.path.join(d, "power1_input"))
            if pw is None:
                pw = self._read(os.path.join(d, "power1_average"))
            if pw is not None and (best is None or pw > best[0]):
                best = (pw, self._read(os.path.join(d, "freq1_input")), self._read(os.path.join(d, "power1_cap")), d)
        if best:
            self.samples.append(best)

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                self._sample()
                self._stop.wait(0.02)
        if self.dirs:
            self._thread = threading.Thread(target=run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread:
            self._stop.set()
            self._thread.join()

    def summary(self):
        if len(self.samples) < 3:
            return None
        mid = self.samples[len(self.samples) // 4:]              # the controller needs a moment to react
        pw = sorted(x[0] for x in mid)[len(mid) // 2] / 1e6
        fr = sorted(x[1] or 0 for x in mid)[len(mid) // 2] / 1e6
        cap = (mid[-1][2] or 0) / 1e6
        top = None
        try:
            with open(os.path.join(os.path.dirname(os.path.dirname(mid[-1][3])), "pp_dpm_sclk")) as f:
                top = max(int(l.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", "")) for l in f if ":" in l)
        except Exception:
            top = None
        return {"socket_w": round(pw, 1), "cap_w": round(cap, 1), "sclk_mhz": round(fr, 1), "sclk_max_mhz": top,
                "at_power_cap": bool(cap and pw >= 0.97 * cap), "samples": len(mid),
                "source": "amdgpu hwmon (power1_input, freq1_input), device by " + ("PCI address" if self.by_address else "highest power")}

def usable_cpus():
    """(cpus this process may run on at once, host cpus, why): the affinity mask cut to the cgroup's CPU quota —
    256 threads inside a 16-CPU quota are 16 cores' worth of work with a throttle on top."""
    host = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = host
    why = "affinity mask"
    try:
        quota = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):                      # cgroup v2
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):      # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        if quota is not None and quota < n:
            n, why = max(1, int(quota)), "cgroup CPU quota"
    except Exception:
        pass
    return n, host, why

