#!/bin/bash
# profile_r02.sh — the rocprofv3 passes behind profiles/r02_* (run on the GPU box from the repo root; output under gpurun_out/r02/).
# Counter passes are separate from the kernel-trace pass (the pool refuses mixing them, and FETCH_SIZE / WRITE_SIZE do not fit
# one pass).  The program after `--` is always python3 itself.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
B="python3 bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline"
B1="python3 bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- $B > $O/c3_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/c3_sq -- $B1 > $O/c3_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/c3_sq2 -- $B1 > $O/c3_sq2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c3_fetch -- $B1 > $O/c3_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c3_write -- $B1 > $O/c3_write.log 2>&1 || exit 1
G="python3 tools/gram_bench.py 1000000 512 16 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- $G > $O/c4_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/c4_pmc -- $G > $O/c4_pmc.log 2>&1 || exit 1
# the run that died inside exit() in round 1 (module-global Context + cooperative launch under the profiler): must exit 0 now
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fs_stats -- python3 tools/fullsize_timing.py > $O/fs_prof.log 2>&1; echo "fullsize_timing under rocprofv3: exit code $?" | tee $O/fs_exit.txt
python3 tools/collect_r02.py $O
