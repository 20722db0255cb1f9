for i in 1 2; do
for lib in libpartls_hip_lz.so libpartls_hip.so; do
PARTLS_LIB=$PWD/partitionedls.jl_amd/$lib python3 bench.py --config L340 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['kernels_ms']['sweep'])"
done; done
