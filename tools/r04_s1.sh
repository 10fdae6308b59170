#!/bin/bash
# Round 4, GPU session 1: generation-3 contraction kernels -- hardware correctness (tests + race screen), per-shape table
# against generation 2, and the three schedule variants (no manual lgkmcnt(0), no priority flips, no group stagger)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "generation3 or tile_variants" -x 2>&1 | tail -5 > gpurun_out/r04_s1_tests.txt
cat gpurun_out/r04_s1_tests.txt
timeout 900 python tools/bench_gen3.py --race > gpurun_out/r04_s1_gen3_base.txt 2>&1
for v in nolgkm noprio nostagger; do
  VIRTEX_AMD_LIB=$R/virtex_amd/lib/libvirtex_amd_$v.so timeout 600 python tools/bench_gen3.py > gpurun_out/r04_s1_gen3_$v.txt 2>&1
done
tail -n 70 gpurun_out/r04_s1_gen3_base.txt
