#!/usr/bin/env python3
"""Per-step view of a rocprofv3 --kernel-trace (--memory-copy-trace) run: the launches of one ANCHOR kernel (one per step, e.g. k_s1_cells) cut the
trace into steps; for every step: its span, the time its kernels were busy, the largest idle gap between two kernels, and the memory copies that
started inside it.  What "nothing of a resident step on the host" looks like in a trace: span ~ busy, no gap above a launch latency, no copies.

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --workload de_c4_single1 ...
    python3 tools/step_gaps.py gpurun_out/tl k_s1_cells"""
import glob
import os
import sys

import pandas as pd


def main(d, anchor):
	f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
	k = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
	cp = None
	m = glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True)
	if m:
		cp = pd.read_csv(m[0])
	at = k.index[k['Kernel_Name'].str.contains(anchor, regex=False)].tolist()
	print('{} launches, {} of {} (steps)'.format(len(k), len(at), anchor))
	print('%5s %10s %10s %10s %8s %s' % ('step', 'span us', 'busy us', 'max gap', 'kernels', 'copies started inside (direction: bytes)'))
	for s in range(1, len(at)):
		seg = k.iloc[at[s - 1] + 1: at[s] + 1]
		t0, t1 = k['End_Timestamp'].iloc[at[s - 1]], seg['End_Timestamp'].iloc[-1]
		busy = float((seg['End_Timestamp'] - seg['Start_Timestamp']).sum()) / 1e3
		prev = [t0] + seg['End_Timestamp'].tolist()[:-1]
		gap = max((a - b) / 1e3 for a, b in zip(seg['Start_Timestamp'].tolist(), prev))
		copies = ''
		if cp is not None:
			inside = cp[(cp['Start_Timestamp'] >= t0) & (cp['Start_Timestamp'] < t1)]
			col = 'Direction' if 'Direction' in inside.columns else inside.columns[1]
			copies = ', '.join('{}: {}'.format(r[col], r.get('Bytes', r.get('Size', '?'))) for _, r in inside.iterrows()) or 'none'
		print('%5d %10.1f %10.1f %10.1f %8d %s' % (s, (t1 - t0) / 1e3, busy, gap, len(seg), copies))


if __name__ == '__main__':
	main(sys.argv[1], sys.argv[2])
