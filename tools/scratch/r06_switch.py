#!/usr/bin/env python
"""Verdict r05 weak 12: where is the switch-over between the dense matrix-pipe conv (K1m) and the load-balanced sparse walk at the
low densities?  Whole training steps of the AMOS-shaped network (BASELINE config 5's per-rank workload) at density 0.05 / 0.1 / 0.2
with the masked layers on K1m (default) or on the walk (E2E_MM_MIN_DENSITY above the density), one child process per setting.
   python tools/scratch/r06_switch.py [d ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch, bench
    dens = float(sys.argv[2])
    r = bench.step_record(torch.device("cuda"), "config5 d=%s" % dens, (128, 128, 128), 1, 16, 32, dens, 2, 7)
    print("RESULT " + json.dumps({"ms": r["ms_per_step"], "kernels": r.get("conv_kernels_per_step")}))
    sys.exit(0)
dens = [float(a) for a in sys.argv[1:]] or [0.05, 0.1, 0.2]
for d in dens:
    for rep in range(2):
        for tag, env in (("K1m", {"E2E_MM_MIN_DENSITY": "0"}), ("walk", {"E2E_MM_MIN_DENSITY": "0.99"})):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(d)], env=dict(os.environ, **env),
                                 capture_output=True, text=True, timeout=900)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print("d=%s %-18s FAILED: %s" % (d, tag, out.stderr[-300:]))
                continue
            r = json.loads(line[0][7:])
            fams = {}
            for k, v in (r["kernels"] or {}).items():
                f = k.split("<")[0]
                fams[f] = fams.get(f, 0) + v
            print("d=%-5s %-18s %7.2f ms / step   %s" % (d, tag, r["ms"], fams), flush=True)
