"""Rate of the reference-shaped entry point pt_render_frame (1 sample per pixel per call, RGBA8 copied to host
memory every call) next to the headless entry point on the same frames: the PCIe- and sync-inclusive figure."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ptamd, util
from ptamd import scenes
b = util.atrium()
W, H, n = 1920, 1080, int(os.environ.get('FRAMES', '512'))      # long enough that the look-ahead's first small batches and the batch left unused at the end do not set the figure
P = scenes.atrium_params(ptamd.default_params(W, H))
r = b.renderer(P)
if os.environ.get('DRAIN'):
    r.set_drain(int(os.environ['DRAIN']))
if os.environ.get('LOOKAHEAD'):
    r.set_frame_lookahead(int(os.environ['LOOKAHEAD']))     # frames per batch at most (default 32)
r.enable_counters(True); r.render(P, 0, n); rays = r.counters()['rays']; r.enable_counters(False)
for f in range(4):
    P.frameIndex = f; r.render_frame(P)
t0 = time.perf_counter()
for f in range(n):
    P.frameIndex = f; r.render_frame(P)
t1 = time.perf_counter()
# the same loop with one frame kept in flight: the copy of frame f runs beside the work of frame f + 1
ta = time.perf_counter()
for f in range(n):
    P.frameIndex = f; r.render_frame_async(P)
    if f: r.frame_fetch()
r.frame_fetch()
tb = time.perf_counter()
r.clear_accum(); r.render(P, 0, n)
t2 = time.perf_counter()
r.clear_accum(); r.render(P, 0, n)
t3 = time.perf_counter()
print('rays for %d frames: %d' % (n, rays))
print('pt_render_frame x%d: %.1f ms/frame, %.1f Mrays/s (host RGBA8 copy + sync every frame)' % (n, (t1 - t0) / n * 1e3, rays / (t1 - t0) / 1e6))
print('pt_render_frame_async + pt_frame_fetch x%d: %.1f ms/frame, %.1f Mrays/s (pinned double buffer, copy stream)' % (n, (tb - ta) / n * 1e3, rays / (tb - ta) / 1e6))
print('pt_render_headless [0,%d): %.1f ms total, %.1f Mrays/s' % (n, (t3 - t2) * 1e3, rays / (t3 - t2) / 1e6))
