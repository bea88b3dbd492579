#!/bin/bash
# round 3, GPU call 25: the whole GPU suite + smoke on the final tree
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
timeout 280 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r3_tests_final_tree.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r3_tests_final_tree.log
cp gpurun_out/parity.jsonl gpurun_out/r3_parity_final_tree.jsonl
cat gpurun_out/r3_tests_final_tree.log
