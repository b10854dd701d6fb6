#!/bin/bash
# Wall time of a small `normalisr coex` / `de` call from the shell: with the torch engine, and through the library's own entries (no torch import)
python - <<'PY'
import numpy as np
rng = np.random.default_rng(0)
np.save('/tmp/exp.npy', rng.standard_normal((2000, 5000)).astype(np.float32))
np.save('/tmp/cov.npy', np.vstack([rng.standard_normal((2, 5000)), np.ones((1, 5000))]))
np.save('/tmp/dg.npy', (rng.random((20, 5000)) < 0.1).astype(np.float32))
PY
for mode in 0 1 default; do
	if [ $mode = default ]; then unset NRM_HOST_ENTRY; else export NRM_HOST_ENTRY=$mode; fi
	for rep in 1 2 3; do
		S=$(date +%s%N)
		bin/normalisr coex /tmp/exp.npy /tmp/cov.npy /tmp/pv$mode.npy --dot_out /tmp/dot$mode.npy > /dev/null 2>&1
		E=$(date +%s%N)
		echo "coex NRM_HOST_ENTRY=$mode: $(( (E - S) / 1000000 )) ms"
	done
	S=$(date +%s%N)
	bin/normalisr de /tmp/dg.npy /tmp/exp.npy /tmp/cov.npy /tmp/dpv$mode.npy /tmp/lfc$mode.npy > /dev/null 2>&1
	E=$(date +%s%N)
	echo "de   NRM_HOST_ENTRY=$mode: $(( (E - S) / 1000000 )) ms"
done
python - <<'PY'
import numpy as np
for f in ('pv', 'dot', 'dpv', 'lfc'):
    a, b = np.load('/tmp/%s0.npy' % f), np.load('/tmp/%sdefault.npy' % f)
    print(f, 'identical' if np.array_equal(a, b) else 'max rel diff %.2e' % np.max(np.abs(a - b) / (np.abs(a) + 1e-300)))
PY
