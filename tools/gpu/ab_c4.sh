# A/B of BASELINE configs[4] (200k all-vs-all hp k=24) across library builds on one box.   bash tools/gpu/ab_c4.sh lib1.so ... ("cur" = in-tree)
for r in 1 2; do for lib in "$@"; do
  if [ "$lib" = cur ]; then unset KMERSEEK_AMD_LIB; else export KMERSEEK_AMD_LIB=$(realpath $lib); fi
  python3 bench.py --mode index-sharded --no-cpu-baseline --no-aux --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$lib', 'step', round(d['ms_per_step'],4), {a:round(b['ms_per_step'],3) for a,b in k.items() if b['ms_per_step']>0.05}, d['self_check']['ok'])"
done; done
