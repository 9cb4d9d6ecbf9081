#!/bin/bash
# One gpurun call -> everything profiles/ needs for a build tag:
#   gpurun -- scripts/profile_round.sh r01_f
# bench line, rocprofv3 kernel stats, and FETCH_SIZE / WRITE_SIZE in separate --pmc passes
# (never combined with other trace domains).  Summaries land in gpurun_out/<tag>/.
set -u
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py > $out/bench.json 2> $out/bench.err || { echo bench failed; tail -3 $out/bench.err; exit 1; }
# (the profiled process may die in the tool's own teardown AFTER its database is written: a run counts
# as failed only when it left no database)
db() { find $out/$1 -name '*.db' 2>/dev/null | head -1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-runs > $out/bench_kt.json 2> $out/kt.err || [ -n "$(db kt)" ] || { echo kernel-trace failed; tail -3 $out/kt.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs > /dev/null 2> $out/pmc_fetch.err || [ -n "$(db pmc_fetch)" ] || { echo pmc fetch failed; tail -3 $out/pmc_fetch.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs > /dev/null 2> $out/pmc_write.err || [ -n "$(db pmc_write)" ] || { echo pmc write failed; tail -3 $out/pmc_write.err; exit 1; }
cd $GRAFT_REPO_ROOT
python3 scripts/rocpd_summary.py stats $(db kt) > $out/kernel_stats.csv
python3 scripts/rocpd_summary.py pmc $(db pmc_fetch) > $out/pmc_fetch_size.csv
python3 scripts/rocpd_summary.py pmc $(db pmc_write) > $out/pmc_write_size.csv
rm -rf $out/kt $out/pmc_fetch $out/pmc_write   # the databases are large; the summaries are what is kept
# profiles/pmc_traffic.json of this build, with the algorithmic bytes per launch of the kernel-trace run beside the counters
python3 scripts/make_pmc_traffic.py $out/pmc_fetch_size.csv $out/pmc_write_size.csv $tag $out/bench_kt.json > /dev/null && cp profiles/pmc_traffic.json $out/pmc_traffic.json
head -4 $out/kernel_stats.csv | cut -c1-160
cat $out/bench.json | cut -c1-400
