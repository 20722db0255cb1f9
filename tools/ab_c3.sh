# ab_c3.sh [lib ...] — the C3 sweep time of several builds of the library on ONE box, twice each (libpartls_hip_<name>.so; "hip" = the shipped one)
for i in 1 2; do
for lib in ${@:-hip}; do
f=libpartls_hip_$lib.so; [ "$lib" = hip ] && f=libpartls_hip.so
PARTLS_LIB=$PWD/partitionedls.jl_amd/$f python3 bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['kernels_ms']['sweep'], d['roofline']['pivots_per_launch'])"
done; done
