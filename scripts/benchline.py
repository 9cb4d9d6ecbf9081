import json,sys
d=json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()); c=d["config"]  # file argument, or stdin
print("value %.3gM/s ms/step %.2f | scan: %.1f us/launch %.0f GB/s (%d launches, %.2f ms) | hist %.2f ms | rows %.0f accepts %.0f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["achieved"], c["scan_launches_per_step"], c["scan_ms_per_step"], c["hist_host_ms_per_step"], c["rows_scored_per_step"], c["accepts_per_step"]))
