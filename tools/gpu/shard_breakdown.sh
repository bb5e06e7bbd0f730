mkdir -p gpurun_out
for cfg in "1000000 1000000 10 1 protein s1m" "125000 1000000 10 1 protein s125k" "10000 10000 7 1 protein c2" "250000 1000000 10 1 protein s250k"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --queries $1 --targets $2 --ksize $3 --scaled $4 --moltype $5 --no-cpu-baseline --no-aux --no-config4 > gpurun_out/r2_bd_$6.json 2>gpurun_out/r2_bd_$6.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r2_bd_$6.json')); k=d['kernels']; print('$6', round(d['ms_per_step'],3)); print({n: round(v['ms_per_step'],3) for n,v in sorted(k.items(), key=lambda x:-x[1]['ms_per_step'])})"
done
