import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "Mrays/s", round(d["ms_per_step"],1), "ms/step; kernel ms", {k:round(v) for k,v in d["kernel_ms"].items()}, "avg", round(d["roofline"]["avg_launch_ms"],3))
