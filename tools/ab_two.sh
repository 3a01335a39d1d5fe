#!/bin/bash
bash tools/ab.sh "" base sb4 sb6 sb12 sb16 base
