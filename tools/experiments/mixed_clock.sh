#!/bin/bash
# GPU box: which clock / power does `gemm_bench.py --mixed 2` actually run at?  (rocm-smi sampled while it loops)
O=gpurun_out; mkdir -p $O
for mixed in 0 2; do
  python tools/gemm_bench.py --nodes 64000 --msgs 320000 --din 512 --dm 512 --which fwd_x6w --rounds 30 --iters 400 --mixed $mixed > /tmp/g.log 2>&1 &
  pid=$!
  sleep 9
  : > /tmp/smi.txt
  while kill -0 $pid 2>/dev/null; do rocm-smi --showclocks --showpower --json >> /tmp/smi.txt 2>/dev/null; echo >> /tmp/smi.txt; done
  python - $mixed <<'PY'
import json, sys
s, p = [], []
for line in open("/tmp/smi.txt"):
    line = line.strip()
    if not line.startswith("{"): continue
    try:
        c = json.loads(line)["card0"]; f = int(c["sclk clock speed:"].strip("()Mhz")); w = float(c["Current Socket Graphics Package Power (W)"])
    except Exception: continue
    if w > 700: s.append(f); p.append(w)
s.sort(); p.sort()
res = [l for l in open("/tmp/g.log") if l.startswith("fwd_x6w")]
print(f"--mixed {sys.argv[1]}: {len(s)} samples: sclk median {s[len(s)//2] if s else None} MHz, power median {p[len(p)//2] if p else None} W | {res[-1].strip() if res else open('/tmp/g.log').read()[-300:]}")
PY
done | tee $O/r05x_mixed_clock.log
