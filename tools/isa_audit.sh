#!/bin/bash
# What the compiler made of the kernels: per kernel of every .hip file, registers / scratch, ds_bpermute count and the number of
# "one load, one wait" pairs (a global load followed by s_waitcnt vmcnt(0) with no other load between: a guarded load inside an
# unrolled loop is issued and waited for in every turn).  Runs here (hipcc cross-compiles; no GPU):
#   bash tools/isa_audit.sh [min_pairs]        -> one line per kernel with >= min_pairs such pairs (default 6) or any ds_bpermute
R=$(cd "$(dirname "$0")/.." && pwd)
MIN=${1:-6}
T=$(mktemp -d)
for f in $R/kmerseek_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o $T/$b.s $f 2>/dev/null || { echo "$b: compile failed"; continue; }
  awk -v F=$b -v MIN=$MIN '
    /^_Z.*:/ && !/^\.L/ { name=$1; sub(/:$/, "", name); serial=0; pend=0; bp=0 }
    /^\tglobal_load|^\tbuffer_load|^\tflat_load/ { pend = pend ? 2 : 1 }
    /s_waitcnt vmcnt\(0\)/ { if (pend == 1) serial++; pend = 0 }
    /ds_bpermute|ds_permute/ { bp++ }
    /^\.Lfunc_end/ { if (serial >= MIN || bp > 0) printf "%-10s %-70.70s load-wait pairs %3d  ds_bpermute %3d\n", F, name, serial, bp }
  ' $T/$b.s
  grep -E "\.(num_vgpr|private_seg_size)," $T/$b.s | awk -v F=$b '/private_seg_size/ { n=$2; v=$3; if (v+0 > 0) { sub(/\.private_seg_size,/, "", n); printf "%-10s %-70.70s SCRATCH %s bytes\n", F, n, v } }'
done
rm -rf $T
