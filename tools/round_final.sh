#!/bin/bash
# round end, one gpurun call: the GPU suite, then everything DESIGN section 4 quotes (tools/round_profiles.sh)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TAG="${1:-r06}"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu_final.log 2>&1
echo "pytest rc $?" >> gpurun_out/${TAG}_pytest_gpu_final.log
tail -3 gpurun_out/${TAG}_pytest_gpu_final.log
timeout 2400 bash tools/round_profiles.sh ${TAG} 2>&1 | tail -40
