# scratch job file for /usr/local/graft/bin/gpurun -- 'bash scripts/_gpurun_job.sh' (rewritten per experiment); this is the
# standard end-of-change verification: the GPU test suite, the smoke entry and one default bench line
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-300
