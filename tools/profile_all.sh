# Run ON THE GPU BOX (gpurun): every rocprofv3 profile behind profiles/r03_*_summary.* (tools/profile2.sh per workload: kernel
# trace + separate --pmc passes) and the bench lines kept under profiles/r03_bench_*.json.  Afterwards, here:
# tools/summarize_prof.py <tag> for each tag, then tools/make_counter_table.py (profiles/traffic.json).
B="--steps 30 --warmup 30 --repeats 3 --no-size-sweep --no-cpu-baseline --no-streamed-leg --kernel-events separate"
tools/profile2.sh r03_config3 bench.py --workload config3 $B && \
MSGW_REGTILES=0 tools/profile2.sh r03_config3_streamed bench.py --workload config3 $B && \
tools/profile2.sh r03_config3_4e6 bench.py --workload config3 --rays-per-gpu 4000000 $B && \
tools/profile2.sh r03_config4_shard bench.py --workload config4 $B && \
tools/profile2.sh r03_config5 bench.py --workload config5 $B && \
MSGW_REGTILES=0 tools/profile2.sh r03_config5_streamed bench.py --workload config5 $B && \
tools/profile2.sh r03_config2 bench.py --workload config2 --steps 1000 --warmup 1000 --repeats 3 --no-cpu-baseline --kernel-events separate && \
MSGW_FIXED_NARROW=0 tools/profile2.sh r03_config2_wide bench.py --workload config2 --steps 1000 --warmup 1000 --repeats 3 --no-cpu-baseline --kernel-events separate && \
MSGW_PERSIST=0 tools/profile2.sh r03_chain bench.py --workload config3 $B && \
tools/profile2.sh r03_hprop tools/run_variant.py hprop 1000000 30 && \
tools/profile2.sh r03_nz tools/run_variant.py nz 1000000 30
echo PROFILES_DONE
