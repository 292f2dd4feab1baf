# Run ON THE GPU BOX (gpurun): the bench lines kept under profiles/r03_bench_*.json and the variant / tall-column tables.
# All on ONE box, so that the paired figures (with / without the 1-rank exchange, narrow / wide config 2) are comparable.
O=gpurun_out/bench_r03; mkdir -p $O
python bench.py > $O/r03_bench_config3.json 2> $O/err.log
python bench.py --steps 20 --warmup 5 > $O/r03_bench_config3_driver_args.json 2>> $O/err.log
python bench.py --workload config2 > $O/r03_bench_config2.json 2>> $O/err.log
MSGW_FIXED_NARROW=0 python bench.py --workload config2 --no-cpu-baseline > $O/r03_bench_config2_two_rays_per_lane.json 2>> $O/err.log
python bench.py --workload config2 --rays-per-gpu 1000000 --no-cpu-baseline > $O/r03_bench_config2_1e6.json 2>> $O/err.log
python bench.py --workload config5 > $O/r03_bench_config5.json 2>> $O/err.log
python bench.py --workload config4 --no-cpu-baseline > $O/r03_bench_config4_shard.json 2>> $O/err.log
python bench.py --workload config4 --force-collective --no-cpu-baseline --no-size-sweep --no-streamed-leg > $O/r03_bench_config4_shard_one_rank_exchange.json 2>> $O/err.log
python bench.py --force-collective --no-cpu-baseline --no-size-sweep --no-streamed-leg > $O/r03_bench_config3_one_rank_exchange.json 2>> $O/err.log
MSGW_XCH_TRANSPORT=shm python bench.py --force-collective --no-cpu-baseline --no-size-sweep --no-streamed-leg > $O/r03_bench_config3_one_rank_exchange_shm.json 2>> $O/err.log
MSGW_EXCHANGE=0 python bench.py --force-collective --no-cpu-baseline --no-size-sweep --no-streamed-leg > $O/r03_bench_config3_one_rank_rccl_chain.json 2>> $O/err.log
MSGW_PERSIST=0 python bench.py --no-cpu-baseline --no-size-sweep --no-streamed-leg > $O/r03_bench_config3_launch_chain.json 2>> $O/err.log
python tools/variant_bench.py 1000000 f64 0.01 > $O/r03_variants_f64.txt 2>&1
python tools/variant_bench.py 1250000 f32 0.01 > $O/r03_variants_f32.txt 2>&1
python tools/tall_probe.py 1000000 201 301 451 601 801 > $O/r03_tall_probe.txt 2>&1
for k in hprop nz nz_sat hprop_nz hprop_sat hprop_f32 nz_f32 hprop_nz_f32 hprop_direct_rl; do python tools/run_variant.py $k 1000000 30; done > $O/r03_chain_variants.txt 2>&1
echo BENCH_ALL_DONE
