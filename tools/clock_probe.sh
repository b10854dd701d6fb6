#!/bin/bash
# effective shader clock of a kernel = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back)
# usage: [PROBE=tools/k2s_time.py] tools/clock_probe.sh <lib.so> <kernel-name-prefix> [args of the timing script (default tools/k2i8_time.py)]
export TMPDIR=/tmp
lib=$1; pref=$2; shift 2
out=/tmp/clk_$$
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out -o c -- python3 ${PROBE:-tools/k2i8_time.py} $lib "$@" > /dev/null 2>&1
python3 - "$out" "$pref" "$lib" <<'PY'
import sys, glob, pandas as pd
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
d = pd.read_csv(f)
d = d[d.Kernel_Name.str.contains(sys.argv[2]) & (d.Counter_Name == 'GRBM_GUI_ACTIVE')]
d['ns'] = d.End_Timestamp - d.Start_Timestamp
d = d[d.ns > 0.5 * d.ns.max()]
print('%s %s: %d launches, %.3f ms, effective clock %.2f GHz' % (sys.argv[3], sys.argv[2], len(d), d.ns.mean() / 1e6, (d.Counter_Value / 8 / d.ns).mean()))
PY
