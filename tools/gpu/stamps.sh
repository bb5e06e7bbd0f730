mkdir -p gpurun_out
( python tools/microbench.py --variants "SK_STAMP" --steps 3
  python tools/microbench.py --variants "SK_STAMP" --steps 3 --plain
  python tools/microbench.py --variants "SK_STAMP" --steps 3 --plain --ksize 16 --scaled 5 --moltype dayhoff
) > gpurun_out/r2_stamps.log 2>&1
