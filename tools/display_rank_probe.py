"""What the display rank of an N-GPU scanline-sharded job does per frame on its own GPU: path tracing of its rows with F frames
in flight, then stitch of the N - 1 partial frames and tonemap on the default stream - everything but the transport (the
partials are standing buffers).  Beside it: the same rank tracing only.  usage: python tools/display_rank_probe.py [workload] [frames per launch]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
from tauray_amd import renderer as R, scenes, transfer
from tauray_amd.distribution import DISTRIBUTION_SCANLINE
W, H = 1920, 1080
wname = sys.argv[1] if len(sys.argv) > 1 else "sponza_teapots"
ctx = R.Context(0)
sc = scenes.WORKLOADS[wname](W, H)
opt = R.options_for_scene(sc, max_bounces=4, samples_per_pixel=1)


class StandingPartials(transfer.LocalExchange):
    def gather_to_display(self, color, dists, rank, world_size, viewports, recv_buffers, ctx):
        for r in range(1, world_size):
            shape = transfer.partial_shape(dists[r], viewports)
            if r not in self.mailbox:
                self.mailbox[r] = (ctx.alloc(shape[0] * shape[1] * shape[2] * 16).zero(), shape)
        return {r: self.mailbox[r][0] for r in range(1, world_size)}


B = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # frames per launch
for world in (2, 4, 8):
    for F in (4,):
        row = []
        for full in (False, True):
            rr = R.RtRenderer(ctx, sc, opt, (W, H), strategy=DISTRIBUTION_SCANLINE, rank=0, world_size=world, use_torch=False, frames_in_flight=F, frames_per_launch=B,
                              exchange=StandingPartials(world) if world > 1 else None)
            step = rr.render if full else rr.render_partial
            for _ in range(8):
                step()
            rr.sync()
            t0 = time.perf_counter()
            for _ in range(80):
                step()
            rr.sync()
            row.append((time.perf_counter() - t0) / (80 * B) * 1e3)
            row.append(rr.path_tracing_ms())
            rr.close()
        print(f"{wname} display rank of {world}, {B} frame(s) per launch, {F} launch(es) in flight: trace only {row[0]:.3f} ms/frame (path tracing timer {row[1]:.3f} ms), "
              f"with stitch + tonemap {row[2]:.3f} ms/frame (timer {row[3]:.3f} ms)")
