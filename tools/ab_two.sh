#!/bin/bash
# A/B of LDS stack depths of the traversal kernel (workgroups per CU that fit) against the default build.
for b in 4 5; do
echo "two lanes, blocks $b"; bash tools/ab.sh "--tune 8,24,$b" base lds15 lds12 lds8 || exit 1
done
echo "one lane, blocks 5"; bash tools/ab.sh "--lanes 1 --tune 8,24,5" base lds15 lds12 lds8 || exit 1
