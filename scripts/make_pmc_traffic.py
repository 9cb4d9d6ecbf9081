#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two PMC summaries (scripts/rocpd_summary.py pmc ...):

  python scripts/make_pmc_traffic.py profiles/r01_e_pmc_fetch_size.csv profiles/r01_e_pmc_write_size.csv r01_e [bench.json]

The optional fourth argument is the bench line of the SAME build and command line family (the kernel-trace run of
scripts/profile_round.sh, `bench_kt.json`): its `roofline.bytes_per_launch` -- rows scored x 4^k x sizeof(count) per
scan launch -- goes into the file as `algorithmic_bytes_per_launch`, which is what bench.py's `roofline.traffic_stale`
compares a later run's own bytes per launch with.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> B): rocprofv3 reports both in KiB and,
on gfx950, FETCH_SIZE prices the 128-B requests of wide coalesced reads at 64 B
(MI355X_MICROARCH.md, HBM section), hence the factor 2 on the read side.
"""
import csv
import json
import sys

KERNELS = {"persist_nmost_kernel": "persist_nmost_kernel", "kmer_hist_kernel": "kmer_hist_kernel",
           "scan_kernel": "scan_kernel_streaming"}


def load(path):
    out = {}
    for row in csv.DictReader(open(path)):
        for key, label in KERNELS.items():
            if key + "<" in row["Kernel"]:
                out[label] = float(row["MeanValue"])
    return out


fetch, write = load(sys.argv[1]), load(sys.argv[2])
tag = sys.argv[3]
alg = None
if len(sys.argv) > 4:
    for line in open(sys.argv[4]):
        line = line.strip()
        if line.startswith("{"):
            alg = json.loads(line)["roofline"]["bytes_per_launch"]
doc = {
    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
              f"bench.py --steps 2 --warmup 1 --no-side-runs, MI355X (build {tag}); summaries in "
              f"profiles/{tag}_pmc_fetch_size.csv and profiles/{tag}_pmc_write_size.csv",
    "units": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts the 128-B "
             "requests of wide coalesced reads as 64 B, so read bytes = 2 x FETCH_SIZE "
             "(MI355X_MICROARCH.md, HBM section). Checks: kmer_hist_kernel WRITE_SIZE vs the "
             "100000 x 4096 x sizeof(count) matrix (800000 KiB with 16-bit rows, 1600000 KiB with 32-bit), its "
             "2 x FETCH_SIZE vs 488281 KiB of sequence bytes.",
    "workload": "nmost n=10, 100000 x 5000 bp, k=6",
}
for label in KERNELS.values():
    if label not in fetch or label not in write:
        continue  # (the profiled run skips the side passes: bench.py --no-side-runs)
    f, w = fetch[label], write[label]
    doc[label] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
    if label == "persist_nmost_kernel" and alg:
        doc[label]["algorithmic_bytes_per_launch"] = alg
        doc[label]["traffic_over_algorithmic"] = (2 * f + w) * 1024 / alg
json.dump(doc, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: doc[k] for k in KERNELS.values() if k in doc}, indent=1))
