"""benchlib — the pieces of bench.py (the repo-root entry the driver runs), one module per concern:

  formulas.py   byte / flop formulas of SURVEY.md section 8(d), the float64 ground truth, the peaks the fractions divide by
  power.py      socket power / shader clock reader (amdgpu hwmon), the CPUs this process may use
  profiles.py   profiles/traffic.json: which committed PMC byte counts apply to the kernels THIS run launched
  configs.py    one configuration measured (rate, in-run kernel times, parity) and its `configs` entry; mixed-filter batches
  headline.py   the headline workload (BASELINE.json configs[2]): setup, parity gate, timed region, roofline block
  dropin.py     the legs through the drop-in seam: one block per call, end to end over PCIe, file threads
  cpu.py        the CPU baseline legs (the only code here that touches oracle/)
  line.py       the ONE contract line (< 4 KB, bounded by tests/test_bench_cpu.py) and bench_details.json beside it

Measurement only: nothing here is on the product path, and only cpu.py imports the oracle."""
