"""Timeline of one training step out of a rocprofv3 --kernel-trace CSV: per kernel start / duration / gap to the launch
before, for the last complete step of the trace (steps are delimited by the optimizer kernel).
  python tools/step_timeline.py gpurun_out/tl/.../*_kernel_trace.csv [step index, e.g. 8; negative counts from the end]
(bench.py brackets kernels with HIP events in its last, untimed steps: a step out of the timed region shows the step itself)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
import glob, os
for f in glob.glob(os.path.join(os.path.dirname(sys.argv[1]), "*memory_copy_trace.csv")):   # copies, when traced
    for r in csv.DictReader(open(f)):
        rows.append({"Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"],
                     "Kernel_Name": "[copy] " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))})
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
if len(ends) < 3: sys.exit("fewer than three steps in the trace")
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3     # which step of the trace (default: third from the end)
lo, hi = ends[which] + 1, ends[which + 1]
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = t0; busy = 0; gaps = 0
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("dmet::(anonymous namespace)::", "").replace("void ", "")
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name[:70]}")
    busy += e - s; gaps += max(0, s - prev_end); prev_end = max(prev_end, e)
print(f"step span {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us, launches {hi - lo + 1}")
