#!/bin/bash
for a in "--lanes 1 --pool 33554432" "--lanes 1 --pool 67108864" "--lanes 1 --pool 134217728" "--lanes 2 --pool 67108864" "--lanes 2 --pool 134217728" "--lanes 1 --pool 50331648"; do echo -n "$a  "; bash tools/ab.sh "$a" x | sed 's/^x: //' || exit 1; done
