#!/bin/bash
# Round 4, GPU session 2: where the time of a generation-3 launch goes (ablations + per-wave time stamps)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
VIRTEX_AMD_LIB=$R/virtex_amd/lib/libvirtex_amd_ablate.so timeout 900 python tools/ablate_gen3.py $ABLATE_ARGS > gpurun_out/r04_s2_ablate.txt 2>&1
tail -n 150 gpurun_out/r04_s2_ablate.txt
