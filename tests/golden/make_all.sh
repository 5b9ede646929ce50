#!/bin/bash
# Regenerates every solver fixture of tests/golden/ (run from the repository root; needs the C oracle built: make -C oracle).  The base set first (the warm-start and
# closed-loop fixtures start from it), the other modes in parallel, the `generator` records last.
set -e
cd "$(dirname "$0")/../.."
export OMP_NUM_THREADS=2
python tests/golden/make_golden.py > /tmp/make_golden_base.log 2>&1
pids=()
for mode in --warm --integral --closed-loop --config3 --midpoint --cn --ball --via --line --two --integral-free --dynamic --polygon --monotone --merit; do
    python tests/golden/make_golden.py $mode > /tmp/make_golden$mode.log 2>&1 &
    pids+=($!)
    while [ "$(jobs -rp | wc -l)" -ge 6 ]; do sleep 1; done
done
for p in "${pids[@]}"; do wait $p; done
python tests/golden/make_golden.py --stamp > /tmp/make_golden--stamp.log 2>&1
tail -n 3 /tmp/make_golden*.log
