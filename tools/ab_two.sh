#!/bin/bash
bash tools/ab.sh "" c2 d1 d2 d3 c2 d1 d2 d3
bash tools/ab.sh "--workload sunsky" base c2 d1
bash tools/ab.sh "--workload soup --spp 16" base c2 d1
