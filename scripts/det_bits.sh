#!/bin/bash
# determinism of scripts/dump_grad.py with the tree's library: three runs
for i in 1 2 3; do timeout 200 python3 scripts/dump_grad.py /tmp/gd_$i.npz 2>/dev/null; done
python3 - <<'PY'
import numpy as np
a = [np.load(f"/tmp/gd_{i}.npz") for i in (1, 2, 3)]
for k in a[0].files:
    print(k, [bool(np.array_equal(a[0][k], a[i][k])) for i in (1, 2)])
PY
