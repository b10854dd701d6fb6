"""Kernels of the last steps of a rocprofv3 --kernel-trace csv in launch order: name, start offset, duration, gap to the previous one (us).
Usage: trace_step.py <kernel_trace.csv> [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
prev = None
for r in rows:
	s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
	print('%-46s start %9.1f  dur %8.1f  gap %6.1f' % (r['Kernel_Name'][:46], (s - t0) / 1e3, (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3))
	prev = e
