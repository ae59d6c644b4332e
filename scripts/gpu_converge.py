"""Converged scene01 render vs the reference's shipped snapshot (image-level statistics)."""
import os, sys, time
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = np.load(os.path.join(ROOT, "tests", "golden", "scene01.npz"))
ref = np.asarray(Image.open(os.path.join(ROOT, "tests/golden/reference_snapshot_1162x1008_frame24440.png")).convert("RGB")).astype(np.float64)
W, H = 1162, 1008
def blur(a, k=8):
    h, w = (a.shape[0] // k) * k, (a.shape[1] // k) * k
    return a[:h, :w].reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3))
for quirk in (1, 0):
    ctx = ptamd.Context(0)
    ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX)); ctx.build(); ctx.resize(W, H)
    ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], bool(quirk)))
    t = time.time()
    for k in range(N // 128):
        ctx.trace_frames(128 * k, ptamd.frame_seeds(1, 128 * k, 128), ptamd.PT_ACCUM_LINEAR)
    ctx.resolve(); img = ctx.display()[..., :3].astype(np.float64); dt = time.time() - t
    c = ctx.counters()
    print(f"quirk={quirk} N={N} {dt:.2f}s {c['rays']/dt/1e6:.0f} Mrays/s rmse={np.sqrt(np.mean((img-ref)**2)):.3f} "
          f"rmse_blur8={np.sqrt(np.mean((blur(img)-blur(ref))**2)):.3f} mean={img.mean():.3f} ref={ref.mean():.3f} "
          f"per-channel ours={img.mean(axis=(0,1))} ref={ref.mean(axis=(0,1))}", flush=True)
    Image.fromarray(img.astype(np.uint8)).save(os.path.join(ROOT, "gpurun_out", f"scene01_q{quirk}_{N}.png"))
    d = np.abs(img - ref).mean(axis=2)
    Image.fromarray(np.clip(d * 4, 0, 255).astype(np.uint8)).save(os.path.join(ROOT, "gpurun_out", f"scene01_q{quirk}_{N}_diff.png"))
    ctx.close()
