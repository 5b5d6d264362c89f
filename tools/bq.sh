#!/bin/bash
cd /root/repo
TBK_SMALL_CALL_PER_ORBITAL=0 python tools/bench_crossover.py 64 4096
TBK_SMALL_CALL_PER_ORBITAL=100000 python tools/bench_crossover.py 64 4096
TBK_SMALL_CALL_PER_ORBITAL=0 python tools/bench_crossover.py 32 256
TBK_SMALL_CALL_PER_ORBITAL=100000 python tools/bench_crossover.py 32 256
