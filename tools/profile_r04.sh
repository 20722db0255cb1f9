#!/bin/bash
# profile_r04.sh — the rocprofv3 passes behind profiles/r04_* (run on the GPU box from the repo root; output under gpurun_out/r04/).
# Counter passes are separate from the kernel-trace pass (the pool refuses mixing them, and FETCH_SIZE / WRITE_SIZE do not fit
# one pass).  The program after `--` is always python3 itself.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
B="python3 bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline --no-host-inclusive"
B1="python3 bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --no-host-inclusive"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- $B > $O/c3_stats.log 2>&1 || exit 1
echo c3_stats done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/c3_sq -- $B1 > $O/c3_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/c3_sq2 -- $B1 > $O/c3_sq2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c3_fetch -- $B1 > $O/c3_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c3_write -- $B1 > $O/c3_write.log 2>&1 || exit 1
echo c3 pmc done
G="python3 tools/gram_bench.py 1000000 512 16 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- $G > $O/c4_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/c4_pmc -- $G > $O/c4_pmc.log 2>&1 || exit 1
echo c4 done
# C5: the 2^24-pattern enumeration + the BnB search with warm-started node bounds (bnb_hard leg: 170k nodes, twice + the 2^24 certification sweep)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python3 bench.py --config C5 --steps 1 --warmup 0 --no-cpu-baseline > $O/c5_stats.log 2>&1 || exit 1
echo c5 done
# beyond the register kernel (D = 340, K = 18: sweep_lazy.hip): kernel time and HBM traffic of the 2^18-pattern sweep (two launches per run)
L="python3 tools/generic_timing.py 20000 340 18"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l340_stats -- $L > $O/l340_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/l340_fetch -- $L > $O/l340_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/l340_write -- $L > $O/l340_write.log 2>&1 || exit 1
echo l340 done
# (the L340 bench line below takes its bytes per sweep from profiles/r04_d340_traffic.json: condense the passes above first)
python3 tools/collect_r04.py $O > /dev/null && cp $O/summary/r04_d340_traffic.json profiles/ || exit 1
# plain bench lines of every config (no profiler attached)
python3 bench.py --config C2 --steps 50 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
python3 bench.py --config C3 > $O/bench_c3.json 2> $O/bench_c3.err || exit 1
python3 bench.py --config C4 --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err || exit 1
python3 bench.py --config C5 --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err || exit 1
python3 bench.py --config L340 --steps 5 --warmup 1 > $O/bench_l340.json 2> $O/bench_l340.err || exit 1
echo benches done
python3 tools/collect_r04.py $O
