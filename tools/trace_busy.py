"""GPU busy fraction over the tail of a rocprofv3 kernel trace: tools/trace_busy.py <kernel_trace.csv> [fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seg = rows[int(len(rows) * (1 - frac)):]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
print(f"kernels {len(seg)}  span {span:.2f} ms  busy {busy:.2f} ms  idle {100 * (1 - busy / span):.1f} %  mean kernel {1e3 * busy / len(seg):.1f} us")
