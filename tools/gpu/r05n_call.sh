# round 5, last GPU call: the whole GPU suite and smoke() on the final tree (the engine changed after the final set's suite run: Conv2D units on
# split operands, padded weight gradients, the driver-level bf16x3 test)
cd $GRAFT_REPO_ROOT
timeout 2600 python -m pytest tests -q -m gpu -s > gpurun_out/r05n_gpu_tests.log 2>&1; tail -4 gpurun_out/r05n_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
