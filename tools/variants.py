#!/usr/bin/env python3
"""A/B timing of alternative builds of libvamp_hip.so on the headline workload.

  python tools/variants.py build NAME=-DFLAG[,-DFLAG2] ...     (here; cross-compiles, no GPU)
  python tools/variants.py run [steps] [extra bench args...]    (on the GPU box; interleaved rounds)
  python tools/variants.py runc3 [steps] [extra args...]         (the same on tools/bench_c3.py, config 3)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "variants")
SRC = os.path.join(ROOT, "vamp_amd", "csrc", "vamp_hip.hip")

if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        os.remove(os.path.join(OUT, f))
    procs = []
    for spec in sys.argv[2:]:
        name, _, flags = spec.partition("=")
        so = os.path.join(OUT, f"lib_{name}.so")
        sys.path.insert(0, ROOT)
        from vamp_amd.build import FLAGS                      # the product's own flags; NAME=-fslp-vectorize undoes one
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", so, SRC]
        cmd += [f for f in flags.split(",") if f]
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        assert p.wait() == 0, name
    print("built", [n for n, _ in procs])
elif sys.argv[1] == "runc3":
    steps = sys.argv[2] if len(sys.argv) > 2 else "5"
    extra = sys.argv[3:]
    names = sorted(f[4:-3] for f in os.listdir(OUT) if f.startswith("lib_") and f.endswith(".so"))
    res = {n: [] for n in names}
    for rnd in range(2):
        for n in names:
            env = dict(os.environ, VAMP_HIP_LIB=os.path.join(OUT, f"lib_{n}.so"))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_c3.py"), "--steps", steps] + extra,
                                 env=env, capture_output=True, text=True)
            try:
                j = json.loads(out.stdout.strip().splitlines()[-1])
                res[n].append(j["avg_launch_ms"])
            except Exception as e:
                print(n, "failed", e, out.stderr[-400:], flush=True)
    for n in names:
        if res[n]:
            print(f"{n:24s} min {min(res[n]):8.3f} ms per half-step  all {['%.3f' % v for v in res[n]]}", flush=True)
else:
    steps = sys.argv[2] if len(sys.argv) > 2 else "3"
    extra = sys.argv[3:]
    names = sorted(f[4:-3] for f in os.listdir(OUT) if f.startswith("lib_") and f.endswith(".so"))
    res = {n: [] for n in names}
    for rnd in range(2):
        for n in names:
            env = dict(os.environ, VAMP_HIP_LIB=os.path.join(OUT, f"lib_{n}.so"))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "1",
                                  "--no-cpu-baseline", "--sustain-seconds", "0"] + extra, env=env, capture_output=True, text=True)
            try:
                j = json.loads(out.stdout.strip().splitlines()[-1])
                res[n].append(j["roofline"]["avg_launch_ms"])
                evals = j["config"]["pixels"] * j["config"]["components"] * j["roofline"]["walker_steps_per_launch"]
            except Exception as e:
                print(n, "failed", e, out.stderr[-400:], flush=True)
    for n in names:
        if res[n]:
            ms = min(res[n])
            print(f"{n:24s} min {ms:8.3f} ms  all {['%.3f' % v for v in res[n]]}  {ms * 1e-3 * 39.3e12 / evals:6.1f} slots/eval", flush=True)
