"""A/B of library builds under ab/: gpu_abx.py name1 name2 ... [-- scene W H frames]; prints the launches of each
build twice, interleaved, so that drift of the box shows."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
cfg = ["scene04", "1162", "1008", "128"]
if "--" in args:
    i = args.index("--"); cfg = args[i + 1:]; args = args[:i]
for rep in range(2):
    for name in args:
        env = dict(os.environ, PTAMD_LIB=os.path.join(ROOT, "ab", name + ".so"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_order.py")] + cfg, env=env,
                           capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            print("FAILED", name, r.stderr[-800:], flush=True); sys.exit(1)
        print(f"{name:12s}", r.stdout.strip().splitlines()[-1], flush=True)
