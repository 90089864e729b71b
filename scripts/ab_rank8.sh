for v in "X=1" "GLMMR_MCML_MVN_BATCH=0" "X=1" "GLMMR_MCML_MVN_BATCH=0"; do env $v python bench.py --steps 3 --no-cpu-baseline --as-rank-of 8 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$v', round(j['ms_per_step'],1), round(j['roofline']['gemm_share_of_step'],3), round(1e3*j['roofline']['avg_launch_ms'],1))"; done
