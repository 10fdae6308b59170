#!/bin/bash
# Round-end evidence on the GPU box: gpu tests, default bench line, kernel-trace summaries (default = three streams,
# and --serial-streams = every kernel un-contended, which the roofline object refers to), PMC traffic passes.
set -x
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; echo "rc=$?" >> gpurun_out/smoke.txt
cd /tmp && export TMPDIR=/tmp
# PMC passes FIRST: the default bench line below then reports roofline.traffic from a table measured on these very sources
[ -n "$SKIP_PMC" ] || rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 > $R/gpurun_out/prof_fetch.log 2>&1
[ -n "$SKIP_PMC" ] || rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 > $R/gpurun_out/prof_write.log 2>&1
cd $R
[ -n "$SKIP_PMC" ] || python tools/pmc_dump.py $(find gpurun_out/prof_fetch -name "*.db" | head -1) "" > gpurun_out/pmc_fetch.txt 2>&1
[ -n "$SKIP_PMC" ] || python tools/pmc_dump.py $(find gpurun_out/prof_write -name "*.db" | head -1) "" > gpurun_out/pmc_write.txt 2>&1
# the table bench.py's roofline.traffic reads, stamped with the hash of the kernel sources (copy both into profiles/)
[ -n "$SKIP_PMC" ] || python tools/pmc_traffic.py gpurun_out/pmc_fetch.txt gpurun_out/pmc_write.txt --json gpurun_out/traffic_table.json > gpurun_out/pmc_traffic.txt 2>&1
[ -n "$SKIP_PMC" ] || cp gpurun_out/traffic_table.json profiles/traffic_table.json
# round 6: the same command beyond the old 2^24-pixel envelope -> profiles/best_batch.json (bench.py's config.best_batch)
for B in 256 320 384 512 768 1024; do
  python bench.py --batch $B --no-cpu-baseline --no-roofline > gpurun_out/bench_b$B.json 2> gpurun_out/bench_b$B.err
done
python tools/best_batch.py gpurun_out/bench_b256.json gpurun_out/bench_b320.json gpurun_out/bench_b384.json gpurun_out/bench_b512.json gpurun_out/bench_b768.json gpurun_out/bench_b1024.json > gpurun_out/best_batch.json && cp gpurun_out/best_batch.json profiles/best_batch.json
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 10 --warmup 5 | python -c "import sys, json; r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config.best_batch as the bench line reports it:', r['config']['best_batch'])" > gpurun_out/best_batch_check.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 9 --warmup 3 > $R/gpurun_out/prof_kt.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --serial-streams --steps 9 --warmup 3 > $R/gpurun_out/prof_ks.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_kt -name "*.db" | head -1) 60 > gpurun_out/kernel_stats.txt
python tools/rocpd_stats.py $(find gpurun_out/prof_ks -name "*.db" | head -1) 60 > gpurun_out/kernel_stats_serial.txt
find gpurun_out -name "*.db" -delete
grep -h '^{' gpurun_out/prof_ks.log | tail -1 > gpurun_out/bench_serial.json
# BASELINE configs 4 and 5 (host-bound at these batch sizes): bench line with the roofline leg + kernel trace each
C4="--textual transdec_postnorm::L4_H1024_A16_F4096 --batch 128"
C5="--visual torchvision::resnet101 --textual transdec_postnorm::L1_H2048_A32_F8192 --batch 64"
# their own PMC tables (bytes per launch depend on the workload's shapes): roofline.traffic of these two lines
for C in 4 5; do
  [ -n "$SKIP_PMC" ] && break
  if [ $C = 4 ]; then CF="$C4"; else CF="$C5"; fi
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch$C -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 $CF > $R/gpurun_out/prof_fetch$C.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write$C -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 $CF > $R/gpurun_out/prof_write$C.log 2>&1
  cd $R
  python tools/pmc_dump.py $(find gpurun_out/prof_fetch$C -name "*.db" | head -1) "" > gpurun_out/pmc_fetch_config$C.txt 2>&1
  python tools/pmc_dump.py $(find gpurun_out/prof_write$C -name "*.db" | head -1) "" > gpurun_out/pmc_write_config$C.txt 2>&1
  python tools/pmc_traffic.py gpurun_out/pmc_fetch_config$C.txt gpurun_out/pmc_write_config$C.txt --json gpurun_out/traffic_table_config$C.json > gpurun_out/pmc_traffic_config$C.txt 2>&1
  cp gpurun_out/traffic_table_config$C.json profiles/traffic_table_config$C.json
  rm -rf gpurun_out/prof_fetch$C gpurun_out/prof_write$C
done
# (round 6: WITH the fidelity leg -- the bf16 step against the fp32 step on the timed batch of these two configurations)
python bench.py --no-cpu-baseline $C4 > gpurun_out/bench_config4.json 2> gpurun_out/bench_config4.err
python bench.py --no-cpu-baseline $C5 > gpurun_out/bench_config5.json 2> gpurun_out/bench_config5.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c4 -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 9 --warmup 3 $C4 > $R/gpurun_out/prof_c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c5 -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 9 --warmup 3 $C5 > $R/gpurun_out/prof_c5.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_c4 -name "*.db" | head -1) 40 > gpurun_out/kernel_stats_serial_config4.txt
python tools/rocpd_stats.py $(find gpurun_out/prof_c5 -name "*.db" | head -1) 40 > gpurun_out/kernel_stats_serial_config5.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_c4 gpurun_out/prof_c5 gpurun_out/prof_kt gpurun_out/prof_ks gpurun_out/prof_fetch gpurun_out/prof_write
# round 5: the reference's module graph through stock PyTorch-ROCm (MIOpen / hipBLASLt) on this GPU, next to the product's step
[ -n "$SKIP_STOCK" ] || timeout 420 python bench.py --no-cpu-baseline --no-fidelity --no-roofline --stock-pytorch-baseline --steps 20 --warmup 10 > gpurun_out/bench_stock_pytorch_baseline.json 2> gpurun_out/bench_stock_pytorch_baseline.err
[ -n "$SKIP_STOCK" ] || timeout 420 python bench.py --no-cpu-baseline --no-fidelity --no-roofline --stock-pytorch-baseline --steps 20 --warmup 10 $C4 > gpurun_out/bench_stock_pytorch_baseline_config4.json 2> gpurun_out/bench_stock_pytorch_baseline_config4.err
[ -n "$SKIP_STOCK" ] || timeout 420 python bench.py --no-cpu-baseline --no-fidelity --no-roofline --stock-pytorch-baseline --steps 20 --warmup 10 $C5 > gpurun_out/bench_stock_pytorch_baseline_config5.json 2> gpurun_out/bench_stock_pytorch_baseline_config5.err
# round 5: the fused conv3 backward against the launches it replaces; JPEG decode throughput of the input pipeline
timeout 300 python tools/bench_conv3_bwd.py > gpurun_out/conv3_bwd.txt 2>&1
timeout 300 python tools/bench_jpeg.py > gpurun_out/bench_jpeg.txt 2>&1
cat gpurun_out/gpu_tests.txt
