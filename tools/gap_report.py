"""Summarise a rocprofv3 kernel trace CSV: per-kernel busy time and the idle gaps between consecutive kernels.
usage: gap_report.py <kernel_trace.csv> [skip_first_n]"""
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 50000]
print(f"kernels {len(rows)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us ({100.0*busy/span:.1f} %)")
print(f"gaps < 50 us: n {len(small)}  mean {sum(small)/max(1,len(small))/1e3:.2f} us  total {sum(small)/1e3:.1f} us ({100.0*sum(small)/span:.1f} % of span)")
by = collections.defaultdict(list)
for (s, e, n), g in zip(rows[:-1], gaps):
    by[n.split("(")[0][:70]].append((e - s, g))
for n, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:14]:
    print(f"{n:70s} n {len(v):5d}  avg {sum(x[0] for x in v)/len(v)/1e3:7.1f} us  gap after {sum(min(x[1],50000) for x in v)/len(v)/1e3:5.2f} us")
