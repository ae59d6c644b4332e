"""The reference's one shipped output (scene01, 1162x1008, 24440 frames) was saved after its author had moved
the camera with the mouse, so the default camera leaves the images mis-registered by several pixels.  This fits
the camera (eye and look-at point, six numbers; Nelder-Mead on the RMSE of 4x4-averaged images at 256 spp) and
then renders the fitted view with many frames: the RMSE that is left is the image-level agreement of the whole
pipeline (loader, tree, traversal, shading, accumulation) with the reference's own renderer."""
import os, sys, time
import numpy as np
from PIL import Image
from scipy.optimize import minimize
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
FINAL = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = np.load(os.path.join(ROOT, "tests", "golden", "scene01.npz"))
ref = np.asarray(Image.open(os.path.join(ROOT, "tests/golden/reference_snapshot_1162x1008_frame24440.png")).convert("RGB")).astype(np.float64)
W, H = 1162, 1008
alb = ptamd.albedo_table(g["mat_table"][:, 0:3], True)
ctx = ptamd.Context(0)
ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.build(); ctx.resize(W, H)

def blur(a, k):
    h, w = (a.shape[0] // k) * k, (a.shape[1] // k) * k
    return a[:h, :w].reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3))

def render(p, frames):
    cb = g["cb"].view(ptamd.CB_RADIX).copy()
    cb["pos"][0] = p[0:3]                       # the trace kernel's view point (g_pos) moves with the camera
    ctx.set_constants(cb)
    ctx.primary_hits(ptamd.default_camera(eye=tuple(float(x) for x in p[0:3]), at=tuple(float(x) for x in p[3:6])), alb)
    ctx.reset_accumulation()
    for k in range(frames // 128):
        ctx.trace_frames(128 * k, ptamd.frame_seeds(1, 128 * k, 128), ptamd.PT_ACCUM_LINEAR)
    ctx.resolve()
    return ctx.display()[..., :3].astype(np.float64)

ref4 = blur(ref, 4)
evals = [0]
def cost(p):
    evals[0] += 1
    c = float(np.sqrt(np.mean((blur(render(p, 256), 4) - ref4) ** 2)))
    if evals[0] % 20 == 0: print(f"  eval {evals[0]} rmse4 {c:.3f} p {np.round(p, 4)}", flush=True)
    return c

p0 = np.array([0.0, 5.0, 23.0, 0.0, 5.0, 0.0])
t = time.time()
if os.environ.get("POSE"):        # POSE="ex ey ez ax ay az": skip the fit
    p = np.array([float(x) for x in os.environ["POSE"].split()])
else:
    print("default camera: rmse4", round(cost(p0), 3), flush=True)
    simplex = np.vstack([p0] + [p0 + d for d in np.diag([0.3, 0.3, 1.0, 0.3, 0.3, 0.3])])
    res = minimize(cost, p0, method="Nelder-Mead", options={"initial_simplex": simplex, "xatol": 2e-3, "fatol": 2e-3, "maxfev": 500})
    p = res.x
    print(f"fitted camera eye {np.round(p[:3], 4)} at {np.round(p[3:], 4)} rmse4 {res.fun:.3f} after {evals[0]} renders, {time.time() - t:.0f} s", flush=True)
# image regions (x0, x1, y0, y1) that isolate the three kinds of material response
REGIONS = {"metal sphere (Phong lobe)": (285, 515, 640, 880), "glass sphere (dielectric)": (640, 915, 665, 940),
           "light and ceiling": (300, 860, 40, 170), "back wall (diffuse)": (210, 950, 190, 600), "red wall": (0, 190, 100, 900), "blue wall": (970, 1162, 100, 900)}
for name, cam in (("default", p0), ("fitted", p)):
    img = render(cam, FINAL)
    print(f"{name}: {FINAL} spp  RMSE {np.sqrt(np.mean((img - ref) ** 2)):.3f}  RMSE(4x4 mean) {np.sqrt(np.mean((blur(img, 4) - ref4) ** 2)):.3f} "
          f" RMSE(8x8 mean) {np.sqrt(np.mean((blur(img, 8) - blur(ref, 8)) ** 2)):.3f}  channel means ours {np.round(img.mean(axis=(0, 1)), 2)} ref {np.round(ref.mean(axis=(0, 1)), 2)}", flush=True)
    if name == "fitted":
        for rn, (x0, x1, y0, y1) in REGIONS.items():
            a, b = img[y0:y1, x0:x1], ref[y0:y1, x0:x1]
            print(f"  {rn:28s} RMSE {np.sqrt(np.mean((a - b) ** 2)):.3f}  RMSE(8x8 mean) {np.sqrt(np.mean((blur(a, 8) - blur(b, 8)) ** 2)):.3f}  "
                  f"channel means ours {np.round(a.mean(axis=(0, 1)), 2)} ref {np.round(b.mean(axis=(0, 1)), 2)}", flush=True)
    Image.fromarray(img.astype(np.uint8)).save(os.path.join(ROOT, "gpurun_out", f"scene01_{name}_{FINAL}.png"))
    Image.fromarray(np.clip(np.abs(img - ref).mean(axis=2) * 4, 0, 255).astype(np.uint8)).save(os.path.join(ROOT, "gpurun_out", f"scene01_{name}_{FINAL}_diff.png"))
ctx.close()
