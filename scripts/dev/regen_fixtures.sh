cd /root/repo
python tests/golden/make_golden.py 2>&1 | tail -5
for f in --warm --integral --closed-loop --config3 --midpoint --cn --ball --via --line --two --integral-free --dynamic --polygon; do python tests/golden/make_golden.py $f 2>&1 | tail -3 | cut -c1-200; done
python tests/golden/make_ref_vectors.py 2>&1 | tail -5
python -c "from mpc_local_planner_amd import _lib; _lib.build(verbose=False)" 2>&1 | tail -2
python -m pytest tests/ -q -m "not gpu" -n 6 2>&1 | tail -15
