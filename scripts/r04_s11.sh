#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_map_path.py -x -q 2>&1 | tail -4 | cut -c1-250
timeout 300 python scripts/time_map_update.py 2>&1 | tail -8
