# A/B of the headline step across library builds on one box: per-kernel ms per step.   bash tools/gpu/ab_step.sh lib1.so lib2.so ... ("cur" = in-tree)
for r in 1 2; do for lib in "$@"; do
  if [ "$lib" = cur ]; then unset KMERSEEK_AMD_LIB; else export KMERSEEK_AMD_LIB=$(realpath $lib); fi
  python3 bench.py --no-cpu-baseline --no-config4 --no-aux --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$lib', 'step', round(d['ms_per_step'],4), 'sketch', round(k['sketch_tiles']['ms_per_step'],4), 'scatter', round(k['bucket_scatter']['ms_per_step'],4), 'join', round(k['join_buckets']['ms_per_step'],4), 'self', d['self_check']['ok'])"
done; done
