#!/bin/bash
# round 6, first lease: (1) the non-finite parity tests + the suites the round's kernel edits touch, (2) the NaN hunt of VERDICT r5 #1:
# RotatE fits of the many-seed learning test with an isfinite watch around every step (scripts/nan_hunt.py) -- default mode, the rare
# paths forced (overflow list through a 16-entry bucket cap, hot-row replicas, poisoned scratch), deterministic mode.
set -u
O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_nonfinite.py -q -p no:cacheprovider > $O/pytest_nonfinite.log 2>&1; echo "rc=$?" >> $O/pytest_nonfinite.log
tail -25 $O/pytest_nonfinite.log | cut -c1-600
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_deterministic.py tests/test_gpu_cols.py tests/test_gpu_lazy.py tests/test_gpu_tile_direct.py -q -p no:cacheprovider > $O/pytest_touched.log 2>&1; echo "rc=$?" >> $O/pytest_touched.log
tail -8 $O/pytest_touched.log | cut -c1-600
H=$O/hunt; mkdir -p $H
timeout 1100 python scripts/nan_hunt.py RotatE nll 0 16000 $H/rotate_nll_default.jsonl default > $H/rotate_nll_default.log 2>&1
tail -3 $H/rotate_nll_default.log | cut -c1-1500
AMDKGE_DEBUG_BUCKET_CAP=16 timeout 500 python scripts/nan_hunt.py RotatE nll 100000 6000 $H/rotate_nll_forced.jsonl forced > $H/rotate_nll_forced.log 2>&1
tail -2 $H/rotate_nll_forced.log | cut -c1-1500
timeout 400 python scripts/nan_hunt.py RotatE nll 200000 4000 $H/rotate_nll_det.jsonl det > $H/rotate_nll_det.log 2>&1
tail -2 $H/rotate_nll_det.log | cut -c1-1500
timeout 300 python scripts/nan_hunt.py RotatE self_adversarial 300000 3000 $H/rotate_sa_default.jsonl default > $H/rotate_sa_default.log 2>&1
tail -2 $H/rotate_sa_default.log | cut -c1-1500
