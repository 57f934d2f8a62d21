#!/bin/bash
for w in 128 256 512; do
  timeout -k 10 120 python scripts/sipp_probe.py $w 512 1500 2>&1 | tail -3
done
