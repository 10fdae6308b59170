#!/bin/bash
# Round 3, GPU session 2: GPU suite on the asm transposing reads + streaming 3x3 weight gradient; per-layer and step-level A/B
# against the library with the builtin reads (the drained pipeline of rounds 1-2)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export VIRTEX_AMD_FUSE_STEM_FWD=1
python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/s2_gpu_tests.txt
OLD=$R/virtex_amd/lib/libvirtex_amd_trbuiltin.so
VIRTEX_AMD_LIB=$OLD VIRTEX_AMD_WGRAD3X3=0 python tools/bench_layers.py > gpurun_out/s2_layers_builtin.txt 2>&1
python tools/bench_layers.py > gpurun_out/s2_layers_new.txt 2>&1
for r in 1 2 3; do
  VIRTEX_AMD_LIB=$OLD VIRTEX_AMD_WGRAD3X3=0 python tools/ab_step.py --rounds 1 builtin_tr >> gpurun_out/s2_ab.txt 2>> gpurun_out/s2_ab.err
  VIRTEX_AMD_WGRAD3X3=0 python tools/ab_step.py --rounds 1 asm_tr >> gpurun_out/s2_ab.txt 2>> gpurun_out/s2_ab.err
  python tools/ab_step.py --rounds 1 asm_tr+wgrad3x3 >> gpurun_out/s2_ab.txt 2>> gpurun_out/s2_ab.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 9 --warmup 3 > $R/gpurun_out/s2_prof_ks.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_ks -name "*.db" | head -1) 80 > gpurun_out/s2_kernel_stats_serial.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_ks
python bench.py --no-cpu-baseline --no-fidelity --steps 30 --warmup 10 > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err
cat gpurun_out/s2_gpu_tests.txt gpurun_out/s2_ab.txt
