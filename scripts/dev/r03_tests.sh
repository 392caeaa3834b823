#!/bin/bash
# the GPU suite (or the tests named on the command line) with durations; log under gpurun_out/$OUT (default r03_tests)
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r03_tests}
mkdir -p $O
timeout 1500 python -m pytest ${@:-tests} -m gpu -q --durations=12 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -${TAIL:-60} $O/pytest.log
