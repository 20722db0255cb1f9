for wb in 0 0.5 1 1.5 2 3; do for ws in 0 1 2; do
  echo -n "wb=$wb ws=$ws: "; PARTLS_CAL_WB=$wb PARTLS_CAL_WS=$ws timeout -k 10 100 python3 tools/shard_chain.py 2>&1 | grep -v amdgpu.ids | cut -c1-120 || exit 1
done; done
for seed in 20260005; do for wb in 0 1 2; do echo -n "C5 wb=$wb: "; PARTLS_CAL_WB=$wb timeout -k 10 100 python3 bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline --bnb-cap 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels_ms']['sweep'], d['roofline_fp64']['pivots_per_launch'])"; done; done
