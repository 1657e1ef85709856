#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2f/tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r2f/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
