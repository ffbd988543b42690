#!/bin/bash
# round-2 call 1: CU-mask probe, new big-nb parity tests, then the whole GPU suite
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 120 ./bin_tmp/cumask_probe > $OUT/cumask_probe.txt 2>&1; echo "probe rc=$?"; cat $OUT/cumask_probe.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "config_tile_sizes" > $OUT/pytest_big.log 2>&1; rc=$?; echo "big rc=$rc"; tail -5 $OUT/pytest_big.log
[ $rc -gt 1 ] && exit $rc
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_config_tile_sizes_partitioned --deselect tests/test_gpu_parity.py::test_config_tile_sizes_multicolumn_nodes --durations=15 > $OUT/pytest_all.log 2>&1; rc=$?; echo "all rc=$rc"; tail -25 $OUT/pytest_all.log
exit 0
