# Differential fuzz campaign (tools/fuzz_parity.py against the oracle) over the forced-path configurations.  R = round tag of the
# logs; PART=1 / PART=2 run half of it (one gpurun call holds ~20 minutes).
R=${R:-r4}
PART=${PART:-0}
CASES=${CASES:-350}
OFS=${OFS:-0}   # added to every seed: a fresh campaign
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" timeout -k 10 500 python tools/fuzz_parity.py --cases $CASES --seed $((SEED + OFS)) > gpurun_out/${R}_fz_$name.log 2>&1; echo "$name rc=$? $(tail -1 gpurun_out/${R}_fz_$name.log)"; }
if [ "$PART" != 2 ]; then
SEED=301 run default A=1
SEED=315 run key_columns KS_DEBUG_JOIN_FP=0
SEED=318 run key_columns_split KS_DEBUG_JOIN_FP=0 KS_DEBUG_JOIN_SPLIT=3
SEED=302 run fp_staged KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SPARSE=0
SEED=303 run fp_sparse KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SPARSE=1
SEED=304 run fp_sparse_segs_coarse KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SPARSE=1 KS_DEBUG_JOIN_SEGS=1 KS_DEBUG_FP_COARSEN=18
SEED=305 run fp_staged_segs_coarse KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SPARSE=0 KS_DEBUG_JOIN_SEGS=1 KS_DEBUG_FP_COARSEN=10
SEED=306 run rows_ticket_planless KS_DEBUG_ROWS_TICKET=1 KS_DEBUG_NO_PLAN=1
SEED=307 run nocompact_nopack KS_DEBUG_NO_COMPACT=1 KS_DEBUG_NO_PACK=1
SEED=308 run lsd_paths KS_DEBUG_PAIRS_LSD=1 KS_DEBUG_INDEX_LSD=1 KS_DEBUG_UNPACKED_PAIRS=1
fi
if [ "$PART" != 1 ]; then
SEED=310 run full_lists KS_DEBUG_QCAP=2
SEED=311 run nopack KS_DEBUG_NO_PACK=1
SEED=312 run nopack_full_lists KS_DEBUG_NO_PACK=1 KS_DEBUG_QCAP=1
SEED=309 timeout -k 10 600 python tools/fuzz_parity.py --big --cases 20 --seed $((309 + OFS)) > gpurun_out/${R}_fz_big.log 2>&1; echo "big rc=$? $(tail -1 gpurun_out/${R}_fz_big.log)"
# big batches against an index in the fingerprint layout: 10-byte query postings at scaled = 1 (both fingerprint join kernels)
SEED=313 KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SPARSE=0 timeout -k 10 500 python tools/fuzz_parity.py --big --cases 12 --seed $((313 + OFS)) > gpurun_out/${R}_fz_big_fp_staged.log 2>&1; echo "big_fp_staged rc=$? $(tail -1 gpurun_out/${R}_fz_big_fp_staged.log)"
SEED=314 KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SPARSE=1 timeout -k 10 500 python tools/fuzz_parity.py --big --cases 12 --seed $((314 + OFS)) > gpurun_out/${R}_fz_big_fp_sparse.log 2>&1; echo "big_fp_sparse rc=$? $(tail -1 gpurun_out/${R}_fz_big_fp_sparse.log)"
# ... and with the pair list forced into segments: the match sort's first level reads the segments in place (lists of >= 65,536 records)
SEED=316 KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SEGS=1 timeout -k 10 500 python tools/fuzz_parity.py --big --cases 12 --seed $((316 + OFS)) > gpurun_out/${R}_fz_big_fp_segs.log 2>&1; echo "big_fp_segs rc=$? $(tail -1 gpurun_out/${R}_fz_big_fp_segs.log)"
# ... and against indexes joined on 16 prefix bits (forced: KS_DEBUG_BUCKET): 9-byte postings behind the bucket scatter, both joins
SEED=319 KS_DEBUG_JOIN_FP=1 KS_DEBUG_BUCKET=64 KS_DEBUG_JOIN_SPARSE=0 timeout -k 10 500 python tools/fuzz_parity.py --big --cases 12 --seed $((319 + OFS)) > gpurun_out/${R}_fz_big_nine_staged.log 2>&1; echo "big_nine_staged rc=$? $(tail -1 gpurun_out/${R}_fz_big_nine_staged.log)"
SEED=320 KS_DEBUG_JOIN_FP=1 KS_DEBUG_BUCKET=64 KS_DEBUG_JOIN_SPARSE=1 timeout -k 10 500 python tools/fuzz_parity.py --big --cases 12 --seed $((320 + OFS)) > gpurun_out/${R}_fz_big_nine_sparse.log 2>&1; echo "big_nine_sparse rc=$? $(tail -1 gpurun_out/${R}_fz_big_nine_sparse.log)"
fi
