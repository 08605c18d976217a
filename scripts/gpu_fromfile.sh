#!/bin/bash
timeout 900 python -m pytest tests/test_api.py tests/test_sharded_reader.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python scripts/exp/exp_fromfile.py 2>&1 | grep -v Warn
timeout 600 python bench.py --from-file /dev/shm/bnpk_bench.fq --reads 8000000 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
rm -f /dev/shm/bnpk_bench.fq
