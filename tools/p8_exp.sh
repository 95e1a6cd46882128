#!/bin/bash
# p8_exp.sh VARIANT...: average duration of the chain kernels under rocprofv3 for experimental library builds
# (tools/build_variant.sh) — timing experiments, results of the NO* variants are garbage by construction.
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  O=$R/gpurun_out/p8exp_$v; rm -rf $O; mkdir -p $O
  PINN_HIP_LIB=$R/pinn_depthestimation_amd/libpinn_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload ns12x256 --bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1 || { echo "FAIL $v"; tail -3 $O/log.txt; exit 1; }
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    n = r["Name"]
    if "k_chain_pack" in n:
        continue
    if "k_chain" in n:
        short = n.split("k_chain_")[1].split("<")[0].split("I")[0]
        out.append("%s %.2f" % (short, float(r["AverageNs"]) / 1e6))
    elif "k_wide_" in n and float(r["AverageNs"]) > 2e5:
        short = n.split("k_wide_")[1].split("(")[0].replace(" ", "")
        out.append("%s %.2f" % (short, float(r["AverageNs"]) / 1e6))
print("; ".join(out))
PY
)"
done
