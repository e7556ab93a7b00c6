#!/bin/bash
# GPU test suite under a time limit; extra pytest arguments are passed through.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 ${GPU_TEST_LIMIT:-900} python -u -m pytest -x -v --timeout 180 --timeout-method thread -m gpu "$@" > gpurun_out/${GPU_TEST_LOG:-gputests.log} 2>&1
rc=$?
tail -n 15 gpurun_out/${GPU_TEST_LOG:-gputests.log}
exit $rc
