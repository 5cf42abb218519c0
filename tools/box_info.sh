#!/bin/bash
# What differs between GPU boxes of the pool (round 6: the same build measures 1 350-1 440 it/s on some boxes and 1 610-1 650 on
# others while mh_search3_kernel ALONE takes 0.573 ms on all of them: the difference is how much of the front end of the next
# iteration the hardware runs beside the search of the previous one).  Prints firmware / partition / clock facts for correlation.
hostname 2>/dev/null
rocm-smi --showfwinfo 2>/dev/null | grep -i "MEC\|ME \|MES\|CE \|PFP\|RLC \|SDMA \|SMC\|VCN" | tr -s ' ' | head -12
rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -i "partition" | head -4
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4
rocm-smi --showpower --showperflevel 2>/dev/null | grep -i "power\|perf" | head -4
cat /sys/module/amdgpu/version 2>/dev/null; uname -r
nproc; uptime
