#!/bin/bash
cd /root/repo
bash tools/run_small.sh "40 32768" "33 32768" "48 32768" "64 32768" 2>&1 | grep "n="
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
