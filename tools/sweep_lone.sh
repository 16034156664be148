# one-stream configurations under engine tunings (run on the GPU box): bash tools/sweep_lone.sh "tune1" "tune2" ..
for cfg in cfg2 cfg4; do
for t in "$@"; do
  [ "$t" = "-" ] && t=""
  echo "== $cfg tune=[$t]"
  python bench.py --only-config $cfg ${t:+--tune $t} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
v=list(d.values())[0]
k=v['roofline']['kernels']
print(v['ms_per_call'], v['msamples_per_s'], {a:k[a]['ms'] for a in k}, v['parity_rms'], v['roofline']['path']['frac_of_min_bytes'])
"
done; done
