#!/bin/bash
# the GPU suite, not stopping at the first failure; smoke()
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-r4g}
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q ${TEST_ARGS} > $OUT/tests.txt 2>&1; echo "tests rc=$?"
tail -12 $OUT/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
