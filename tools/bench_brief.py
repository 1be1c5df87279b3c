"""Prints the numbers of a bench.py JSON line (primary and secondary entries) in one line each."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("primary  %-44s %9.4f ms/step  kernel %9.4f ms  frac %.4f  %.3e %s" % (
    d["config"]["workload"][:44], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"], d["unit"]))
for e in d.get("secondary", []):
    print("secondary %-43s %9.4f ms/step  kernel %9.4f ms  frac %.4f  %.3e %s  [%s]" % (
        e["config"]["workload"][:43], e["ms_per_step"], e["roofline"]["kernel_ms"], e["roofline"]["frac"], e["value"], e["unit"],
        e["roofline"]["kernel"]))
if "cpu_baseline" in d:
    print("cpu_baseline %.3e pairs/s on %d cores; 1 thread %.3e" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"],
                                                                     d["cpu_baseline"]["single_thread"]["value"]))
