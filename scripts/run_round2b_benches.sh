# Round-2 (second half) bench set after the LDS-private products / camera-sorted landmarks / RBA_MIXED:
#   gpurun -- 'bash scripts/run_round2b_benches.sh'   -> gpurun_out/r2b/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/venice_driver_flags.json 2> $O/venice_driver_flags.log
python bench.py > $O/venice.json 2> $O/venice.log
python bench.py --mixed --cpu-baseline-iters 0 > $O/venice_mixed.json 2> $O/venice_mixed.log
RBA_EXPLICIT_AFTER=0 python bench.py --cpu-baseline-iters 0 --no-reference-semantics > $O/venice_matrix_free.json 2> $O/venice_matrix_free.log
RBA_HX_LDS=0 RBA_SORT_BY_CAMERA=0 python bench.py --cpu-baseline-iters 0 --no-reference-semantics > $O/venice_tile_per_wave.json 2> $O/venice_tile_per_wave.log
python bench.py --preconditioner JACOBI --cpu-baseline-iters 0 --no-reference-semantics > $O/venice_jacobi.json 2> $O/venice_jacobi.log
python bench.py --preconditioner POWER_SCHUR_COMPLEMENT --cpu-baseline-iters 0 --no-reference-semantics > $O/venice_power.json 2> $O/venice_power.log
python bench.py --workload venice-1778+tail --cpu-baseline-iters 0 > $O/venice_tail.json 2> $O/venice_tail.log
python bench.py --workload trafalgar-257 > $O/trafalgar.json 2> $O/trafalgar.log
python bench.py --workload ladybug-49 > $O/ladybug.json 2> $O/ladybug.log
python bench.py --use-double --cpu-baseline-iters 0 > $O/venice_f64.json 2> $O/venice_f64.log
python bench.py --workload final-13682 --cpu-baseline-iters 0 --steps 10 --warmup 2 > $O/final.json 2> $O/final.log
python bench.py --workload final-13682 --mixed --cpu-baseline-iters 0 --steps 10 --warmup 2 --no-reference-semantics > $O/final_mixed.json 2> $O/final_mixed.log
python bench.py --workload final-13682 --preconditioner POWER_SCHUR_COMPLEMENT --cpu-baseline-iters 0 --steps 10 --warmup 2 --no-reference-semantics > $O/final_power.json 2> $O/final_power.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --cpu-baseline-iters 0 --no-reference-semantics > $O/prof.json 2> $O/prof.log
cp $O/prof/bench_kernel_stats.csv $O/kernel_stats.csv
rm -f $O/prof/bench_kernel_trace.csv
B="python $R/bench.py --steps 4 --warmup 2 --cpu-baseline-iters 0 --no-reference-semantics"
RBA_PCG_GRAPHS=0 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p1 -- $B > /dev/null 2> $O/p1.log
RBA_PCG_GRAPHS=0 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/p2 -- $B > /dev/null 2> $O/p2.log
RBA_PCG_GRAPHS=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- $B > /dev/null 2> $O/p3.log
RBA_PCG_GRAPHS=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- $B > /dev/null 2> $O/p4.log
cd $R
python scripts/pmc_summary.py $O/p1 $O/p2 $O/p3 $O/p4 > $O/pmc_all_kernels.csv
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
bash scripts/run_pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/pmc/hx_traffic.json $O/hx_traffic.json
for f in venice_driver_flags venice venice_mixed venice_matrix_free venice_tile_per_wave venice_jacobi venice_power venice_tail trafalgar ladybug venice_f64 final final_mixed final_power; do echo $f $(python -c "import json,sys; d=json.load(open('$O/$f.json')); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['config']['cg_iterations_per_step'],1), (d['config'].get('value_reference_semantics') or {}).get('value'), d['roofline']['frac'], d['roofline']['avg_launch_ms'])"); done
