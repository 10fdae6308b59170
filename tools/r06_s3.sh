#!/bin/bash
# Round 6, GPU session 3: (1) the envelope beyond 2^24 pixels per tensor (vtx_fdiv30): kernel and model tests, bench lines at
# 320 / 384 / 512 images per GPU; (2) where the folded bn3 backward loses: serial-stream kernel traces with the fold on and off.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -k "beyond_2_24" 2>&1 | tail -12 > gpurun_out/r06_s3_envelope_tests.txt
for B in 320 384 512; do
  timeout 500 python bench.py --batch $B --no-cpu-baseline --no-roofline --steps 20 --warmup 6 > gpurun_out/r06_s3_bench_b$B.json 2> gpurun_out/r06_s3_bench_b$B.err
done
cd /tmp && export TMPDIR=/tmp
VIRTEX_AMD_BN3_FOLD=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fold -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 6 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_fold.log 2>&1
VIRTEX_AMD_BN3_FOLD=0 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_nofold -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 6 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_nofold.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_fold -name "*.db" | head -1) 70 > gpurun_out/r06_s3_kernel_stats_serial_fold.txt
python tools/rocpd_stats.py $(find gpurun_out/prof_nofold -name "*.db" | head -1) 70 > gpurun_out/r06_s3_kernel_stats_serial_nofold.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_fold gpurun_out/prof_nofold
cat gpurun_out/r06_s3_envelope_tests.txt
for B in 320 384 512; do python - <<PY
import json
try:
    r = json.loads([l for l in open("gpurun_out/r06_s3_bench_b$B.json") if l.startswith("{")][-1])
    print($B, r["value"], r["ms_per_step"], r["config"]["launch"], r["config"]["peak_memory_gb"], r.get("fidelity", {}).get("backbone"))
except Exception as e:
    print($B, "no record", e); print(open("gpurun_out/r06_s3_bench_b$B.err").read()[-800:])
PY
done
