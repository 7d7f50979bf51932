#!/bin/bash
# one GPU call: the lone-wave probe and the loop-unrolling variants built by scripts/exp_variants.sh / by hand
mkdir -p gpurun_out
./probes/lone_wave_probe > gpurun_out/lone_wave_probe.txt 2>&1
{
echo "== quad layout, 1 slab per workgroup (1024 samples) and 2 slabs (2048)"
python scripts/exp_run.py 1024 sbase shun
python scripts/exp_run.py 2048 sbase shun
echo "== cooperative quad, 1 and 256 samples"
python scripts/exp_run.py 1 ubase uh4
python scripts/exp_run.py 256 ubase uh4
echo "== row-lane: base"
JQ_LIB=$PWD/juqbox.jl_amd/exp/libjq_sbase.so python scripts/time_cases.py
echo "== row-lane: Horner recurrence unrolled"
JQ_LIB=$PWD/juqbox.jl_amd/exp/libjq_rlunroll.so python scripts/time_cases.py
} > gpurun_out/exp_branch.txt 2>&1
cat gpurun_out/lone_wave_probe.txt gpurun_out/exp_branch.txt
