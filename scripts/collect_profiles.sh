#!/bin/bash
# Runs on the GPU box (gpurun): every rocprofv3 summary the round commits under profiles/.
# usage: bash scripts/collect_profiles.sh <tag> <build-id>
set -u
TAG=${1:-r03}; BUILD=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG; mkdir -p $O
PY=python3
# 1. the bench line itself (no profiler) + kernel stats of the same command
$PY bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err
echo "bench line done" 
rocprofv3 --kernel-trace -d $O/kt_bench -o b -- $PY bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $O/kt_bench.log 2>&1
$PY scripts/rocpd_stats.py $O/kt_bench/b_results.db > $O/bench_kernel_stats.csv
echo "bench kernel trace done"
# 2. SQ counters on the shipping kernels
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -o sq -- $PY bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $O/sq.log 2>&1
$PY scripts/pmc_summary.py $O/sq/sq_counter_collection.csv dgemm_band k_band_reduce dgemm_dl_kernel k_potrf_leaf > $O/bench_sq_counters.csv
echo "sq done"
# 3. HBM traffic
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o f -- $PY bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o w -- $PY bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $O/w.log 2>&1
$PY scripts/make_traffic_json.py $O/f $O/w 5000 1024 0 "$BUILD" > $O/hbm_traffic.json
echo "traffic done"
# 4. one rank of the 8-GPU job (128 chains, its share of every theta-step round; peers emulated), the same with the
#    round-2 replicated theta-step, and the dense-Z workload
$PY bench.py --steps 3 --no-cpu-baseline --as-rank-of 8 > $O/bench_rank8_line.json 2>/dev/null
GLMMR_MCML_THETA_SHARD=0 GLMMR_MCML_THETA_BATCH=1 $PY bench.py --steps 3 --warmup 1 --no-cpu-baseline --chains 128 > $O/bench_c128_replicated_line.json 2>/dev/null
rocprofv3 --kernel-trace -d $O/kt_rank8 -o b -- $PY bench.py --steps 2 --no-cpu-baseline --as-rank-of 8 > $O/kt_rank8.log 2>&1
$PY scripts/rocpd_stats.py $O/kt_rank8/b_results.db > $O/bench_rank8_kernel_stats.csv
$PY bench.py --steps 2 --warmup 1 --no-cpu-baseline --dense-z > $O/bench_densez_line.json 2>/dev/null
for w in 2 4; do $PY bench.py --steps 3 --no-cpu-baseline --as-rank-of $w > $O/bench_rank${w}_line.json 2>/dev/null; done
echo "rank8 / dense-z done"
# 5. theta-step alone, configs 4 and 5
rocprofv3 --kernel-trace -d $O/kt_mvn -o m -- $PY scripts/time_mvn.py 5000 1024 6 > $O/mvn.log 2>&1
$PY scripts/rocpd_stats.py $O/kt_mvn/m_results.db > $O/mvn_kernel_stats.csv
for c in cfg4 cfg5; do
  C=512; [ $c = cfg5 ] && C=1024
  rocprofv3 --kernel-trace -d $O/kt_$c -o k -- $PY scripts/time_cfg.py $c $C > $O/$c.log 2>&1
  $PY scripts/rocpd_stats.py $O/kt_$c/k_results.db > $O/${c}_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$c -o f -- $PY scripts/time_cfg.py $c $C > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w_$c -o w -- $PY scripts/time_cfg.py $c $C > /dev/null 2>&1
  $PY scripts/make_traffic_json.py $O/f_$c $O/w_$c 0 $C 0 "$BUILD" > $O/${c}_hbm_traffic.json
done
echo "cfg4/cfg5 done"
# 6. theta-step rounds: k candidates factorised side by side; the sampler's products at the sharded chain counts
for a in "5000 1024" "5000 128" "2000 256"; do $PY scripts/time_mvn_batch.py $a 1 2 4 8 >> $O/mvn_batch_timing.txt 2>&1; done
$PY scripts/ab_sampler_gemm.py 5000 1024 512 256 128 > $O/sampler_products.txt 2>&1
$PY scripts/ab_sampler_gemm.py 2000 256 >> $O/sampler_products.txt 2>&1
echo "batch / products done"
for g in 2 old 0; do for m in 1024 128; do GLMMR_MCML_CHOL_GRAPH=$g $PY scripts/time_mvn.py 5000 $m 12 2>/dev/null | sed "s/^/CHOL_GRAPH=$g  /" >> $O/mvn_graph_ab.txt; done; done
$PY scripts/rocpd_timeline.py $O/kt_mvn/m_results.db 260 > $O/mvn_timeline.txt 2>&1
echo "few chains / graph A-B / timeline done"
# keep the merge-back small: drop the raw traces
find $O -name "*_results.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls -la $O
