"""Does stepping K x-slabs of one scene on K concurrent streams (deep ghost zones, sb_peer_* between the
engines of this process) beat one engine stepping the whole scene?  One launch per substep has every
workgroup in the same phase at the same time and drains the chip between launches; independent streams
fill those gaps.  Prints particle-steps/s of the OWNED particles."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
sb = ge.load_package()
halo = sb.halo
W, H, steps = 1000, 1000, 960
# K <= 3: HIP multiplexes a process's streams onto 4 hardware queues, and a waiting flag kernel blocks
# whatever is queued behind it on the same hardware queue (the bounded wait then gives up)
for K, depth in ((1, 0), (2, 16), (2, 24), (3, 16)):
    exs = []
    for r in range(K):
        buf, plan = halo.slab_scene(sb, r, K, -(-W // K), H, jitter=1.0, depth=max(depth, 1))
        eng = sb.Engine(bounds_size=100000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=0)
        eng.write_buffers(buf)
        exs.append(halo.PeerExchanger(eng, plan, timeout_ms=5000) if K > 1 else eng)
    if K > 1:
        cards = [ex.card for ex in exs]
        for ex in exs:
            ex.connect(cards)
    def run(n):
        done = 0
        while done < n:
            m = min(depth, n - done) if K > 1 else n
            for ex in exs:
                ex.step(m)
            done += m
    def sync():
        for ex in exs:
            (ex.engine if K > 1 else ex).sync()
    run(96); sync()
    t0 = time.perf_counter()
    run(steps); sync()
    dt = time.perf_counter() - t0
    print("slabs %d depth %2d: %.2f us/substep, %.3e particle-steps/s" % (K, depth, dt / steps * 1e6, W * H * steps / dt), flush=True)
    for ex in exs:
        (ex.engine if K > 1 else ex).destroy()
