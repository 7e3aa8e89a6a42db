set -e
python -m pytest tests/test_tq_gpu.py tests/test_batcher_gpu.py tests/test_tq_config4_gpu.py tests/test_c_consumer.py tests/test_rtcd_gpu.py tests/test_sadloop_gpu.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03_t3.txt 2>&1 || { tail -30 gpurun_out/r03_t3.txt; exit 1; }
tail -2 gpurun_out/r03_t3.txt
python bench.py --no-cpu-baseline --only-legs tq_chain,tu_batcher,sad_loop_480p > gpurun_out/r03_bench_tq.json 2> gpurun_out/r03_bench_tq.err || { tail -5 gpurun_out/r03_bench_tq.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_tq.json'))
for k,v in d["legs"]["tq_chain"]["sizes"].items():
    print(k, {a:(b["ms"], b["gpix_per_s"], b["frac_hbm_algorithmic"]) for a,b in v.items()})
print(json.dumps(d["legs"]["tu_batcher"])[:700])
s=d["legs"]["sad_loop_480p"]; print("sad_loop", s["ms"], s["blocks_per_s"], s["frac_sad_ceiling"])
PY
