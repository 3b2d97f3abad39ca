"""tools/show_bench.py FILE -- one line from a bench.py output file (takes the JSON line, whatever else a library printed around it)"""
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
d = json.loads(line)
print(sys.argv[1], round(d["value"]), "Mrays/s", round(d["ms_per_step"], 1), "ms/step; kernel ms", {k: round(v) for k, v in d["kernel_ms"].items()}, "avg", round(d["roofline"]["avg_launch_ms"], 3))
