import json,sys
for l in sys.stdin:
    if l.startswith('{"metric"'):
        j = json.loads(l); print(round(j["value"]/1e6,1), j["ms_per_step"], j["n_gpus"], j["config"]["parallelism"][:120])
