#!/bin/bash
# round 5 session 3: the whole GPU suite on the tree with strict_order = 3 / time-bounded completion waits / no hipGraph option
O=gpurun_out/r05c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
