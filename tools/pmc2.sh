#!/bin/bash
# usage: pmc2.sh <tag> "<counters>"   (env PARTLS_KERNEL etc. pass through)
TAG=$1; CNT=$2
mkdir -p gpurun_out/$TAG
timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d gpurun_out/$TAG/pmc -- python3 bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/$TAG/pmc.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/$TAG/pmc/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:50]
    agg[k][r['Counter_Name']]=agg[k].get(r['Counter_Name'],0)+float(r['Counter_Value'])
for k,v in agg.items():
    if 'sweep' in k: print(k, {a:round(b/1e9,3) for a,b in v.items()})
PY
