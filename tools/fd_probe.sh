cd $GRAFT_REPO_ROOT
one() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['steady_state']['it_per_s'],1), d['roofline_iteration'].get('sub_batch_chains'))"; }
one X=1
one DPX_BENCH_FORCE_DIST=1
one GPU_MAX_HW_QUEUES=1
one DPX_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=2
