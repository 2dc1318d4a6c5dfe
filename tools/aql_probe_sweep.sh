#!/bin/bash
# Sweep of tools/aql_probe.hip over grid shapes (round 6): which part of the 12.3 us from submission to the
# host seeing a word is the grid, which the command processor, which the PCIe store?  Run on the GPU box after `make aql_probe`:
#   bash tools/aql_probe_sweep.sh > gpurun_out/r06_pass_floor.jsonl
set -u
N=${N:-8000}
for shape in "1 64" "1 384" "8 384" "64 384" "128 384" "256 384" "357 768" "512 384" "714 384" "714 64" "714 256" "535 512" "357 1024" "268 1024" "1428 192" "2856 96" "4280 64"; do
    set -- $shape
    build/aql_probe build/aql_probe_kernel.hsaco $N $1 $2 host
done
for shape in "1 64" "256 384" "714 384"; do
    set -- $shape
    build/aql_probe build/aql_probe_kernel.hsaco $N $1 $2 bar
done
