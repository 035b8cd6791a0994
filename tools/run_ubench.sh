#!/bin/bash
# Build and run the fp64 micro-benchmarks the roofline fractions lean on; raw output -> gpurun_out/r03_ubench_*.txt (copied to profiles/)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out $root/build
for b in mfma_f64_rate sep_nodes; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $root/build/$b $root/tools/ubench/$b.hip 2> $out/r03_ubench_$b.build.log || { echo "build of $b failed"; continue; }
  timeout -k 10 120 $root/build/$b > $out/r03_ubench_$b.txt 2>&1
  echo "== $b"; cat $out/r03_ubench_$b.txt
done
python3 - "$out" <<'PY'
import json, re, sys
out = sys.argv[1]
d = dict(source="tools/ubench/mfma_f64_rate.hip, tools/ubench/sep_nodes.hip (raw output beside this file)")
try:
    rates = [(l.split()[0], float(re.search(r"([\d.]+) TFLOP/s", l).group(1))) for l in open(out + "/r03_ubench_mfma_f64_rate.txt") if "TFLOP/s" in l]
    d["mfma_f64_16x16x4_tflops_best"] = max(v for k, v in rates if k.startswith("mfma"))
    d["v_fma_f64_tflops_best"] = max(v for k, v in rates if k.startswith("v_fma"))
except Exception as e:
    d["error"] = str(e)
try:
    txt = open(out + "/r03_ubench_sep_nodes.txt").read()
    m = re.search(r"one stream \(back to back\): ([\d.]+) ms", txt); n = re.search(r"two streams\s*: ([\d.]+) ms", txt)
    if m and n:
        d["matrix_then_vector_kernel_ms"] = float(m.group(1)); d["matrix_beside_vector_kernel_ms"] = float(n.group(1))
except Exception as e:
    d["error2"] = str(e)
json.dump(d, open(out + "/r03_ubench.json", "w"), indent=1)
print(json.dumps(d))
PY
