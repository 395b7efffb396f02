#!/bin/bash
# tier height x mesh size through the C ABI (tools/nd_drive): us per solve, one launch per upper level
for n in 24 48 64 100 128 180 250 350 500 700 1000; do
  for t in -1 0 1 2 3 4 5 6; do
    r=$(timeout 60 tools/build/nd_drive $n 200 3 $t 0 2>&1 | grep -E "persist 0|error" | head -1 | sed -E 's/.*launches +([0-9.]+) us per solve.*/\1/; s/.*error.*/refused/')
    l=$(timeout 60 tools/build/nd_drive $n 1 3 $t 0 2>&1 | grep -E "^plane" | sed -E 's/.*, ([0-9]+) levels, tier of ([0-9]+) .*/levels \1 tier \2/')
    echo "n $n (V $((n*n))) tier_levels $t: $l  $r us"
  done
done
