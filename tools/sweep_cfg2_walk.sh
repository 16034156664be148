for t in "-" "walk_lpb=2" "walk_lpb=4" "walk_lpb=4,walk_tiles=2" "walk_lpb=4,walk_tiles=4" "walk_lpb=2,walk_tiles=4" "walk_lpb=2,walk_tiles=8" "walk_tiles=6" "walk_tiles=12"; do
  [ "$t" = "-" ] && t=""
  echo "== cfg2 tune=[$t]"
  python bench.py --only-config cfg2 ${t:+--tune $t} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
v=list(d.values())[0]
k=v['roofline']['kernels']
print(v['ms_per_call'], v['msamples_per_s'], {a:k[a]['ms'] for a in k}, v['parity_rms'])
"
done
