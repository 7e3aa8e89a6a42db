set -e
python -m pytest tests/test_tq_gpu.py tests/test_tq_config4_gpu.py tests/test_batcher_gpu.py tests/test_rtcd_gpu.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03_tests_l.txt 2>&1 || { tail -30 gpurun_out/r03_tests_l.txt | cut -c1-300; exit 1; }
tail -2 gpurun_out/r03_tests_l.txt
python bench.py --no-cpu-baseline --only-legs tq_chain,tu_batcher,uhd_10bit > gpurun_out/r03_bench_tq.json 2>gpurun_out/r03_bench_tq.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_tq.json'))
L=d['legs']
print("tq:", {k:(v["plane_64Mpx"]["ms"], v["frame_1080p"]["ms"]) for k,v in L["tq_chain"]["sizes"].items()})
print("batch:", {k:v["ms_per_flush"] for k,v in L["tu_batcher"].items() if k.startswith("flush")})
PY
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_tq.json'))
print({k:v for k,v in d['legs']['uhd_10bit'].items() if k.startswith('encode')})
PY
