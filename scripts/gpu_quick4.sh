#!/bin/bash
export TMPDIR=/tmp
( timeout 300 python bench.py --steps 8 --warmup 2 --cpu-baseline off 2>&1 | grep -o 'timed.*\|"value": [0-9.]*' )
( timeout 300 python bench.py --steps 8 --warmup 2 --cpu-baseline off --no-roofline 2>&1 | grep -o 'timed.*\|"value": [0-9.]*' )
( timeout 300 python bench.py --steps 12 --warmup 3 --cpu-baseline off --inflight 3 2>&1 | grep -o 'timed.*\|"value": [0-9.]*' )
