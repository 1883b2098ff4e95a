"""Print one line per bench variant from gpurun_out/<tag>/bench_variants.jsonl."""
import json, re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\{"args": "([^"]*)", "out": (\{.*?\}\})\s*\n\}', txt, re.S):
    try:
        o = json.loads(m.group(2))
    except Exception:
        print("fail", m.group(1)); continue
    r = o.get('roofline', {})
    print(f"{m.group(1):75s} {o['value']:9.0f} it/s  {r.get('avg_launch_us',0):8.1f} us  frac {r.get('frac',0):.3f}")
