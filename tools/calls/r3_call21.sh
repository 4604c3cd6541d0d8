R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-amp-leg --no-h2d-leg 2>&1 | grep -v "^{" | tail -25
