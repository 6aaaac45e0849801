#!/bin/bash
# more K-tile schedule variants of the 4-wave fp8 tile
mkdir -p gpurun_out/r04
export MX4_SCHEDULES=6,0,2,6      # (schedules 1, 3..5, 7..11 of the sweep were removed again: within 2 % of 6 or slower, profiles/r04_notes.md §7)
for shape in "" "16384 4608 3584" "16384 3584 18944" "16384 37888 3584" "10496 3584 3584"; do
timeout 300 python tools/mx4_ksweep.py $shape 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/mx4_sched.txt
done
