"""Kernels and memory copies of the last `span_ms` of a rocprofv3 run (--kernel-trace --memory-copy-trace, csv) on one timeline.
Usage: trace_timeline.py <dir> [span_ms]"""
import csv, glob, sys
d = sys.argv[1]
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 9e6
ev = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
	for r in csv.DictReader(open(f)):
		ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:40]))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
	for r in csv.DictReader(open(f)):
		ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C %s %s B' % (r.get('Direction', '?'), r.get('Size', r.get('Bytes', '?')))))
ev.sort()
t1 = max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= t1 - span]
t0 = ev[0][0]
for s, e, name in ev:
	print('%9.1f +%8.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, name))
