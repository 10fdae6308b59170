#!/bin/bash
# Copy the round-end evidence of gpurun_out/ (tools/round_end.sh) into profiles/ under the round's prefix:
#     bash tools/collect_evidence.sh r06
set -e
R=${1:?round prefix, e.g. r06}
cd "$(dirname "$0")/.."
G=gpurun_out
for f in bench_default bench_serial bench_config4 bench_config5 bench_b256 bench_b320 bench_b384 bench_b512 bench_b768 bench_b1024 bench_stock_pytorch_baseline bench_stock_pytorch_baseline_config4 bench_stock_pytorch_baseline_config5; do
  [ -s $G/$f.json ] && grep -h '^{' $G/$f.json | tail -1 > profiles/${R}_$f.json
done
for f in gpu_tests smoke kernel_stats kernel_stats_serial kernel_stats_serial_config4 kernel_stats_serial_config5 pmc_traffic pmc_traffic_config4 pmc_traffic_config5 bench_jpeg conv3_bwd; do
  [ -s $G/$f.txt ] && cp $G/$f.txt profiles/${R}_$f.txt
done
for f in $G/fidelity_*.json $G/parity_*.json; do
  [ -s $f ] && cp $f profiles/${R}_$(basename $f)
done
for t in traffic_table traffic_table_config4 traffic_table_config5 best_batch; do
  [ -s $G/$t.json ] && cp $G/$t.json profiles/$t.json
done
# PMC bytes against the algorithmic bytes per kernel class, from the bench lines of the same run
python tools/traffic_ratio.py profiles/${R}_bench_default.json profiles/traffic_table.json > profiles/${R}_traffic_ratio.txt 2>/dev/null || true
python tools/traffic_ratio.py profiles/${R}_bench_config4.json profiles/traffic_table_config4.json > profiles/${R}_traffic_ratio_config4.txt 2>/dev/null || true
python tools/traffic_ratio.py profiles/${R}_bench_config5.json profiles/traffic_table_config5.json > profiles/${R}_traffic_ratio_config5.txt 2>/dev/null || true
ls profiles/${R}_* | wc -l
