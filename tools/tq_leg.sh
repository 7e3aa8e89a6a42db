python bench.py --only-legs tq_chain --no-cpu-baseline --no-host-threads --steps 3 2>/dev/null > gpurun_out/b1.json; python - <<EOP
import json
d=json.load(open("gpurun_out/b1.json"))["legs"]
for k,v in d["tq_chain"]["sizes"].items():
    print(k, {a:(b["ms"],b.get("frac_hbm_algorithmic")) for a,b in v.items() if isinstance(b,dict) and "ms" in b})
EOP
