#!/bin/bash
# evidence_r03.sh: every bench line + rocprofv3 stats + PMC passes DESIGN.md cites for round 3 (one GPU box, ~6 min)
R="$(cd "$(dirname "$0")/.." && pwd)"; cd $R; mkdir -p gpurun_out/r03
bash tools/evidence.sh pe10x10_batch --workload pe10x10 > gpurun_out/r03/ev_pe10x10_batch.log 2>&1
bash tools/evidence.sh co100x20_batch --workload co100x20 > gpurun_out/r03/ev_co100x20_batch.log 2>&1
bash tools/evidence.sh pe10x10_4m_batch --workload pe10x10 --points 4194304 > gpurun_out/r03/ev_pe10x10_4m_batch.log 2>&1
for w in "pe8x64 --points 4194304" "pe10x10 --engine 4" "co100x20 --engine 4"; do
  n=$(echo $w | tr -d ' -' ); python bench.py --no-cpu-baseline --workload $w > gpurun_out/r03/bench_$n.json 2> /dev/null; done
python bench.py --workload lbfgs8x64 --steps 15 > gpurun_out/r03/bench_lbfgs8x64.json 2> gpurun_out/r03/bench_lbfgs8x64.log
python tools/newmethod_latency.py 2>&1 | grep newmethod > gpurun_out/r03/newmethod_latency.txt
python tools/small_n_latency.py 2>&1 | grep -v Epoch | grep "N=\|CMB" > gpurun_out/r03/small_n_latency.txt
echo done
