"""A/B of the two trace variants over scenes and sizes (steady-state ms per 32-frame launch)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for scene in ("scene01", "scene02", "scene03", "scene04"):
    for (W, H) in ((1162, 1008), (3840, 2160)):
        for mk in ("0", "1"):
            env = dict(os.environ, PTAMD_MEGAKERNEL=mk)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_order.py"), scene, str(W), str(H), "32"],
                               env=env, capture_output=True, text=True, timeout=200)
            if r.returncode != 0:
                print("FAILED", scene, W, H, mk, r.stderr[-500:], flush=True); sys.exit(1)
            line = r.stdout.strip().splitlines()[-1]
            print("mega" if mk == "1" else "wave", line, flush=True)
