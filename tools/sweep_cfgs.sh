# all four config legs with the library named by FOLVE_AMD_LIB (or the product library): bash tools/sweep_cfgs.sh
for cfg in cfg1 cfg2 cfg4; do
  python bench.py --only-config $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
v=list(d.values())[0]
k=v['roofline']['kernels']
print('$cfg', v['ms_per_call'], v['msamples_per_s'], {a:k[a]['ms'] for a in k}, v['parity_rms'])
"
done
python tools/quick_bench.py 64 256 30 2>/dev/null | tail -3
