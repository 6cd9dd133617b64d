"""TTFT / TTST of the WorldPipeline cascade on one MI355X (method of the reference's evaluation/latency.py); prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terrain_diffusion_amd.latency import measure_latency
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
r = measure_latency(num_runs=n, dtype=dt)
print(json.dumps({k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()}))
