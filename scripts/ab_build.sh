#!/bin/bash
# A/B of BUILD switches (persist.hip) on one box:   gpurun -- scripts/ab_build.sh <tag> "-|-DDVS_ACC_STRIDE=32" "ENV=..." ["ENV=..."]
# every build ("-": the plain one) runs scripts/ab.sh with the environment settings given; the plain library is rebuilt at the end
# AB_OBJ (default persist.hip.o): the object the switch is compiled into; AB_CMD (default scripts/ab.sh): what is run per build
set -u
tag=$1; builds=$2; shift; shift
obj=${AB_OBJ:-persist.hip.o}; cmd=${AB_CMD:-scripts/ab.sh}
out=gpurun_out/${tag}_ab_build.txt
: > $out
IFS='|' read -ra B <<< "$builds"
for b in "${B[@]}"; do
  x=$b; [ "$b" = "-" ] && x=""
  rm -f diverseseq_amd/csrc/build/$obj
  make -C diverseseq_amd/csrc -j8 EXTRA="$x" > gpurun_out/${tag}_build.log 2>&1 || { tail -5 gpurun_out/${tag}_build.log; exit 1; }
  echo "== build: $b" >> $out
  bash $cmd "$@" >> $out 2>&1 || { cat $out; exit 1; }
done
rm -f diverseseq_amd/csrc/build/$obj
make -C diverseseq_amd/csrc -j8 > /dev/null 2>&1
cat $out
