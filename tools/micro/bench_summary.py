"""One bench.py run, its line saved under gpurun_out/<tag>/bench.json, the figures that matter printed."""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "bench"
out = os.path.join(root, "gpurun_out", tag); os.makedirs(out, exist_ok=True)
r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + sys.argv[2:], capture_output=True, text=True)
open(os.path.join(out, "bench.json"), "w").write(r.stdout); open(os.path.join(out, "bench.err"), "w").write(r.stderr)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_instrumented", "value_cold", "value_block65536", "value_f32", "f32_frac")})
ro = d["roofline"]; print({k: ro[k] for k in ("achieved", "frac", "avg_kernel_ms", "avg_prep_ms", "kernel")})
for k in ("config_b", "config_c", "config_d", "config_e"):
    if k in d: print(k, d[k]["value"], d[k].get("ms_per_step"), d[k].get("fir_kernel"), (d[k].get("roofline") or {}).get("frac"), d[k].get("stage_ms", ""))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("other_configs_error"), d.get("config_d_error"))
