# Run ON THE GPU BOX (gpurun): every rocprofv3 profile behind profiles/r02_*_summary.* (tools/profile2.sh per workload) and the
# bench lines kept under profiles/r02_bench_*.json.  Afterwards, here: tools/summarize_prof.py <tag> for each tag.
B="--steps 30 --warmup 30 --repeats 3 --no-size-sweep --no-cpu-baseline --kernel-events separate"
tools/profile2.sh r02_config3 bench.py --workload config3 $B && \
MSGW_REGTILES=2 tools/profile2.sh r02_config3_res2 bench.py --workload config3 $B && \
MSGW_REGTILES=0 tools/profile2.sh r02_config3_streamed bench.py --workload config3 $B && \
tools/profile2.sh r02_config3_4e6 bench.py --workload config3 --rays-per-gpu 4000000 $B && \
tools/profile2.sh r02_config5 bench.py --workload config5 $B && \
MSGW_REGTILES=0 tools/profile2.sh r02_config5_streamed bench.py --workload config5 $B && \
tools/profile2.sh r02_config2 bench.py --workload config2 --steps 1000 --warmup 100 --repeats 3 --no-cpu-baseline --kernel-events separate && \
MSGW_PERSIST=0 tools/profile2.sh r02_chain bench.py --workload config3 $B && \
tools/profile2.sh r02_hprop tools/run_variant.py hprop 1000000 30 && \
tools/profile2.sh r02_nz tools/run_variant.py nz 1000000 30 && \
tools/profile2.sh r02_tall301 tools/run_variant.py tall301 1000000 30
mkdir -p gpurun_out/bench
python bench.py > gpurun_out/bench/r02_bench_config3.json 2> gpurun_out/bench/err.log
python bench.py --workload config5 > gpurun_out/bench/r02_bench_config5.json 2>> gpurun_out/bench/err.log
python bench.py --workload config2 > gpurun_out/bench/r02_bench_config2.json 2>> gpurun_out/bench/err.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench/r02_bench_config3_driver_args.json 2>> gpurun_out/bench/err.log
python bench.py --force-collective --no-cpu-baseline --no-size-sweep > gpurun_out/bench/r02_bench_config3_one_rank_exchange.json 2>> gpurun_out/bench/err.log
MSGW_XCH_TRANSPORT=shm python bench.py --force-collective --no-cpu-baseline --no-size-sweep > gpurun_out/bench/r02_bench_config3_one_rank_exchange_shm.json 2>> gpurun_out/bench/err.log
MSGW_EXCHANGE=0 python bench.py --force-collective --no-cpu-baseline --no-size-sweep > gpurun_out/bench/r02_bench_config3_one_rank_rccl_chain.json 2>> gpurun_out/bench/err.log
python tools/tall_probe.py 1000000 201 301 451 601 801 > gpurun_out/bench/tall_probe.txt 2>&1
python tools/variant_bench.py 1000000 f64 0.01 > gpurun_out/bench/variants_f64.txt 2>&1
python tools/variant_bench.py 1250000 f32 0.01 > gpurun_out/bench/variants_f32.txt 2>&1
echo ALLDONE
