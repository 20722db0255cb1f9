# ab_lazy.sh [lib ...] — the L340 sweep time of several builds of the library on ONE box, twice each (libpartls_hip_<name>.so; "hip" = the shipped one)
for i in 1 2; do
for lib in ${@:-lz hip}; do
f=libpartls_hip_$lib.so; [ "$lib" = hip ] && f=libpartls_hip.so
PARTLS_LIB=$PWD/partitionedls.jl_amd/$f python3 bench.py --config L340 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['kernels_ms']['sweep'])"
done; done
