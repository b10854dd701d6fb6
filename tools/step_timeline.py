#!/usr/bin/env python3
"""Timeline of the last launches of a rocprofv3 --kernel-trace run: start offset, duration and the idle gap in front of every kernel.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --workload de_c4_single4 --steps 3 --warmup 1 --no-extras --cpu-seconds 0 --e2e 0
    python3 tools/step_timeline.py gpurun_out/tl 120

Where a resident step's time goes when the sum of its kernels is well below the step (host read-backs between launches show as gaps)."""
import glob
import os
import sys

import pandas as pd


def main(d, last):
	f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
	df = pd.read_csv(f).sort_values('Start_Timestamp')
	df = df.tail(last)
	t0 = df['Start_Timestamp'].iloc[0]
	prev = None
	busy = 0
	for _, r in df.iterrows():
		name = r['Kernel_Name']
		name = name[name.find('k_'):][:60] if 'k_' in name else name[:60]
		gap = 0 if prev is None else (r['Start_Timestamp'] - prev) / 1e3
		dur = (r['End_Timestamp'] - r['Start_Timestamp']) / 1e3
		busy += dur
		print('{:10.1f} us  +{:8.1f} gap  {:9.1f} us  {}'.format((r['Start_Timestamp'] - t0) / 1e3, gap, dur, name))
		prev = r['End_Timestamp']
	span = (df['End_Timestamp'].iloc[-1] - t0) / 1e3
	print('span {:.1f} us, kernels {:.1f} us, idle {:.1f} us'.format(span, busy, span - busy))


if __name__ == '__main__':
	main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100)
