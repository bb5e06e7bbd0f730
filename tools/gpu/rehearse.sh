# N ranks on ONE GPU (gloo collectives on host tensors): rehearses the launcher / sharding / exchange plumbing of bench.py's
# three modes and asserts sharded == unsharded rows (bench.py: sharded_equals_unsharded; a mismatch raises under
# KS_BENCH_REHEARSE).  Numbers from these runs mean nothing.  R = round tag of the output files.
R=${R:-r4}
mkdir -p gpurun_out
set -x
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 4 --steps 3 --warmup 1 --queries 100000 --targets 400000 --c4-proteins 50000 --no-cpu-baseline --no-aux > gpurun_out/${R}_reh4.json 2> gpurun_out/${R}_reh4.err; echo "rc=$?"
tail -c 600 gpurun_out/${R}_reh4.err
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 4 --steps 3 --warmup 1 --scaling weak --queries 100000 --targets 200000 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/${R}_reh4w.json 2> gpurun_out/${R}_reh4w.err; echo "rc=$?"
tail -c 300 gpurun_out/${R}_reh4w.err
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 4 --steps 3 --warmup 1 --mode index-sharded --c4-proteins 50000 > gpurun_out/${R}_reh4i.json 2> gpurun_out/${R}_reh4i.err; echo "rc=$?"
tail -c 300 gpurun_out/${R}_reh4i.err
# a rank whose peers never come up (nobody listens on the port): must exit non-zero within its timeout, not hang
t0=$(date +%s)
KS_BENCH_REHEARSE=1 KS_BENCH_PG_TIMEOUT_S=20 RANK=1 LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29999 timeout -k 10 150 python bench.py --gpus 2 --steps 1 --warmup 1 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/${R}_reh_badport.out 2> gpurun_out/${R}_reh_badport.err
echo "bad-port rank: rc=$? after $(( $(date +%s) - t0 )) s"
tail -n 3 gpurun_out/${R}_reh_badport.err
set +x
R=$R python - <<'PY'
import json, os
R = os.environ["R"]
for f in (f"{R}_reh4", f"{R}_reh4w", f"{R}_reh4i"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        c4 = d.get("config4_index_sharded") or {}
        print(f, d["n_gpus"], d["scaling"], round(d["ms_per_step"], 3), d["config"].get("parallelism", "")[:60],
              "gate:", d.get("sharded_equals_unsharded"), "c4 gate:", c4.get("sharded_equals_unsharded"),
              "self:", (d.get("self_check") or {}).get("ok"), (c4.get("self_check") or {}).get("ok"))
    except Exception as e:
        print(f, "ERR", e)
PY
