#!/bin/bash
# rocprofv3 kernel statistics for the kernels outside the headline path: mash sketch + pairs (C5) and
# FASTA ingest.   gpurun -- scripts/profile_other.sh r02_x
set -u
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
db() { find $out/$1 -name '*.db' 2>/dev/null | head -1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/kt_c5 -- python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py C5 > $out/c5.jsonl 2> $out/c5.err || [ -n "$(db kt_c5)" ] || { echo C5 failed; tail -3 $out/c5.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/kt_ing -- python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py ING > $out/ing.jsonl 2> $out/ing.err || [ -n "$(db kt_ing)" ] || { echo ING failed; tail -3 $out/ing.err; exit 1; }
cd $GRAFT_REPO_ROOT
python3 scripts/rocpd_summary.py stats $(db kt_c5) > $out/c5_kernel_stats.csv
python3 scripts/rocpd_summary.py stats $(db kt_ing) > $out/ing_kernel_stats.csv
rm -rf $out/kt_c5 $out/kt_ing
cat $out/c5.jsonl $out/ing.jsonl
head -8 $out/c5_kernel_stats.csv | cut -c1-150
head -8 $out/ing_kernel_stats.csv | cut -c1-150
