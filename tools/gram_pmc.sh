#!/bin/bash
# gram_pmc.sh [N] — counter passes of the Gram kernel at D = 512 (run on the GPU box from the repo root; output under gpurun_out/gpmc/)
set -o pipefail
export TMPDIR=/tmp
N=${1:-1000000}
O=gpurun_out/gpmc; mkdir -p $O
G="python3 tools/gram_bench.py $N 512 16 3"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/mfma -- $G > $O/mfma.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/tcc -- $G > $O/tcc.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $G > $O/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq -- $G > $O/sq.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for d in ("mfma", "tcc", "fetch", "sq"):
    for f in glob.glob("gpurun_out/gpmc/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "gram_kernel" in k:
                print(d, {c: v / n[(k, c)] for c, v in acc[k].items()})
PY
