timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r2o_tests.log 2>&1; tail -3 gpurun_out/r2o_tests.log
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_sub2 -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu > $root/gpurun_out/prof_sub2.log 2>&1
cd $root
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_sub2/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("%-40s calls %4s avg %10.1f us" % (r["Name"].replace("void ","").replace("csdev::","")[:40], r["Calls"], float(r["AverageNs"])/1e3))
PY
