import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][0])
kb=d["kernel_breakdown_ms_per_step"]
print(d["ms_per_step"], {k:v for k,v in kb.items() if "dW" in k or "dw" in k})
