#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (separate FETCH_SIZE / WRITE_SIZE passes) per kernel.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE [more dirs] > profiles/rNN_pmc_c2.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced stream, so it is doubled for kernels
whose loads are that wide (K2, vectorised K1).  Round 5 calibrated two narrower patterns on known byte counts, as the guide asks for other widths: the
normvar kernels' 4 B/lane coalesced row reads (200 MB read, 102.6 MB reported) and K3's 8 B/lane reads of the product matrix (120 MB read, 61.1 MB
reported) are halved in the same way -- a wave's contiguous 256 / 512 B are 128-byte requests tallied at 64 -- and doubled here as well."""
import glob
import json
import os
import sys

import pandas as pd

WIDE = {'k_dl_count': True, 'k_dl_fill': True, 'k_ds_ct': False, 'k_nv_moments': True, 'k_nv_apply': True, 'k_nv_weights': True, 'k_de_sparse': True, 'k_s1_stream': True, 'k_s1_cells': False, 'k_binnet_rows': True, 'k_residualize_res': True, 'k_s4_sweep': False, 'k_fix_dot': False, 'k_gram_skinny': True, 'k_gram_f64': True, 'k_gram_i8': True, 'k_quantize_rows': True, 'k_residualize_v4': True, 'k_assoc_sweep_sym': True, 'k_assoc_sweep': True, 'k_residualize': False}


def main(dirs):
	rows = []
	for d in dirs:
		for f in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
			df = pd.read_csv(f)
			df['kernel'] = df['Kernel_Name'].str.extract(r'\b(k_[a-z0-9_]+(?:<[^>]*>)?)')  # template arguments kept: instantiations differ
			rows.append(df[df['kernel'].notna()])
	df = pd.concat(rows)
	# one instantiation launched on inputs of very different size with the same persistent grid (the de step's 20000-gene
	# pass and its 1-row grouping pass) is reported as two classes split at the geometric mean of its durations, never
	# averaged together
	df['ns'] = df['End_Timestamp'] - df['Start_Timestamp']
	span = df.groupby('kernel')['ns'].agg(['min', 'max'])
	cut = {k: (r['min'] * r['max'])**0.5 if r['max'] > 5 * r['min'] else None for k, r in span.iterrows()}
	df['kernel'] = [k if cut[k] is None else k + (' @long' if ns > cut[k] else ' @short') for k, ns in zip(df['kernel'], df['ns'])]
	mean = df.groupby(['kernel', 'Counter_Name'])['Counter_Value'].mean().unstack()
	# effective shader clock (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / duration
	gui = df[df['Counter_Name'] == 'GRBM_GUI_ACTIVE']
	clock = (gui['Counter_Value'] / 8 / gui['ns'].clip(lower=1)).groupby(gui['kernel']).mean() if len(gui) else {}
	out = {}
	for k, r in mean.iterrows():
		e = {c: float(v) for c, v in r.items() if v == v}
		if 'FETCH_SIZE' in e:
			e['fetch_bytes_raw'] = e['FETCH_SIZE'] * 1024
			wide = WIDE.get(k.split('<')[0].split(' ')[0], False)
			e['fetch_bytes'] = e['fetch_bytes_raw'] * (2 if wide else 1)
			e['fetch_corrected_x2'] = bool(wide)
		if 'WRITE_SIZE' in e:
			e['write_bytes'] = e['WRITE_SIZE'] * 1024
		if 'fetch_bytes' in e and 'write_bytes' in e:
			e['hbm_bytes_per_launch'] = e['fetch_bytes'] + e['write_bytes']
		if k in clock:
			e['effective_clock_ghz'] = float(clock[k])
		out[k] = e
	json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
	main(sys.argv[1:])
