B="--steps 30 --warmup 30 --repeats 3 --no-size-sweep --no-cpu-baseline --kernel-events separate"
tools/profile2.sh r02_config3 bench.py --workload config3 $B && \
tools/profile2.sh r02_config5 bench.py --workload config5 $B && \
MSGW_REGTILES=0 tools/profile2.sh r02_config5_streamed bench.py --workload config5 $B && \
tools/profile2.sh r02_config2 bench.py --workload config2 --steps 1000 --warmup 100 --repeats 3 --no-cpu-baseline --kernel-events separate && \
MSGW_PERSIST=0 tools/profile2.sh r02_chain bench.py --workload config3 $B && \
MSGW_REGTILES=0 tools/profile2.sh r02_config3_streamed bench.py --workload config3 $B && \
tools/profile2.sh r02_hprop tools/run_variant.py hprop 1000000 30 && \
tools/profile2.sh r02_nz tools/run_variant.py nz 1000000 30 && \
tools/profile2.sh r02_tall301 tools/run_variant.py tall301 1000000 30
