"""Sweep of the persistent kernel's two thresholds (PTAMD_REFILL_T x PTAMD_LEAF_T): last-launch ms of gpu_order.py."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1:] or ["scene04", "1162", "1008", "128"]
for r in (16, 20, 24, 28):
    row = []
    for l in (8, 12, 16, 20, 24):
        env = dict(os.environ, PTAMD_REFILL_T=str(r), PTAMD_LEAF_T=str(l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_order.py")] + cfg, env=env,
                             capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print("FAILED", r, l, out.stderr[-500:]); sys.exit(1)
        ms = sorted(float(x.split("ms")[0]) for x in out.stdout.strip().splitlines()[-1].split()[5:])
        row.append(f"L{l}:{ms[len(ms)//2]:.1f}")
    print(f"R{r}", " ".join(row), flush=True)
