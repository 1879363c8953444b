#!/bin/bash
# GPU box: package power and shader clock while bench.py's training steps run (rocm-smi sampled as fast as it answers, ~3 per second)
TAG=${1:-r05q}; O=gpurun_out; mkdir -p $O
for cfg in "" "--hidden 256 --graphs 32" "--model seq-great"; do
  python - "$cfg" <<'PY'
import glob, json, os, statistics, subprocess, sys, time
cfg = sys.argv[1]
steps = {"": 1500, "--hidden 256 --graphs 32": 900, "--model seq-great": 2000}[cfg]
hw = []  # (the hwmon files of card0 are not the visible GPU's on these boxes: 464 W / 2 400 MHz whatever runs) -> rocm-smi
def read(path):
    try: return int(open(path).read().strip())
    except Exception: return None
p = subprocess.Popen(f"python bench.py {cfg} --steps {steps} --warmup 20 --no-cpu-baseline --no-also --no-predict > /tmp/b.json 2>/dev/null", shell=True)
samples = []
t0 = time.time()
while p.poll() is None:
    if hw:
        d = hw[0]
        pw = read(d + "/power1_average") or read(d + "/power1_input")
        fq = read(d + "/freq1_input")
        samples.append((time.time() - t0, (fq or 0) / 1e6, (pw or 0) / 1e6))
        time.sleep(0.02)
    else:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True).stdout
        try:
            c = json.loads(out).get("card0", {})
            samples.append((time.time() - t0, int(c["sclk clock speed:"].strip("()Mhz")), float(c["Current Socket Graphics Package Power (W)"])))
        except Exception:
            pass
txt = open("/tmp/b.json").read().strip()
if not txt:
    print(f"config [{cfg}]: bench printed nothing (rc {p.returncode})"); sys.exit(0)
j = json.loads(txt.splitlines()[-1])
busy = [(f, w) for _, f, w in samples if w > 700]
src = "hwmon sysfs" if hw else "rocm-smi"
if busy:
    fs, ws = sorted(f for f, _ in busy), sorted(w for _, w in busy)
    q = lambda a, x: a[min(len(a) - 1, int(x * len(a)))]
    print(f"config [{cfg or 'c2 default'}]: {j['value']} {j['unit']}, {j['ms_per_step']} ms/step; {len(busy)} samples above 700 W ({src}): "
          f"sclk p10 / median / p90 {q(fs, .1):.0f} / {q(fs, .5):.0f} / {q(fs, .9):.0f} MHz, power p10 / median / p90 {q(ws, .1):.0f} / {q(ws, .5):.0f} / {q(ws, .9):.0f} W")
else:
    print(f"config [{cfg}]: no samples under load ({len(samples)} samples, source {src}); first: {samples[:3]}")
PY
done 2>&1 | tee $O/${TAG}_step_power.log
