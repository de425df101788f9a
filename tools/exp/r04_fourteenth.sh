#!/bin/bash
# round 4, fourteenth GPU call: where the hand-off's transport time goes with the mailbox in host memory vs device memory (BAR), with and without an
# HDP flush by the host
set -u
OUT=gpurun_out/r04_n
mkdir -p $OUT
for p in 64 3072; do
  for k in 0 3 1; do timeout 30 tools/micro/bar_mailbox $k $p 0 2>&1 | tail -1 | tee -a $OUT/bar_mailbox_breakdown.txt; done
  for k in 3 1; do timeout 30 tools/micro/bar_mailbox $k $p 1 2>&1 | tail -2 | tee -a $OUT/bar_mailbox_breakdown.txt; done
done
