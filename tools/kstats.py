"""Print kernel name (shortened), calls and average duration (us) of a rocprofv3 kernel_stats.csv: kstats.py <file.csv> [substring ...]"""
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:]:
	if len(sys.argv) < 3 or any(k in r[0] for k in sys.argv[2:]):
		print('%-60s calls %5s  avg %10.2f us' % (r[0][:60], r[1], float(r[3]) / 1e3))
