cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4b
python tools/prof_c4.py 4 > gpurun_out/r4b/c4_default.log 2>&1
for i in 1 2; do
python tools/bench_c4.py 4 32 2>&1 | grep config >> gpurun_out/r4b/c4_hint.log
DPX_CG_NO_HINT=1 python tools/bench_c4.py 4 32 2>&1 | grep config | sed "s/^/NO HINT /" >> gpurun_out/r4b/c4_hint.log
done
cat gpurun_out/r4b/c4_default.log | head -14; cat gpurun_out/r4b/c4_hint.log
python -m pytest tests -m gpu -x -q -k "cg or config4 or ladmm or csmri" 2>&1 | tail -3
