"""How long does one ghost-zone refresh take on the device?  One rank sends the real slab payload to itself
over RCCL on the engine stream (pack kernel -> batch_isend_irecv -> unpack kernel), timed with events.
A lower bound for the 2-GPU case (no xGMI hop), used to pick bench.py's --ghost-depth."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29612")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as ge
sb = ge.load_package()
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
W, H = 1000, 1000
for depth in (8, 16, 24, 32, 48):
    buf, plan = sb.halo.slab_scene(sb, 1, 3, W, H, jitter=1.0, depth=depth)   # an interior rank: two neighbours
    eng = sb.Engine(bounds_size=100000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=0)
    eng.write_buffers(buf)
    gp, sp, gb, sbm = plan.lists()
    eng.halo_configure(gp, sp, gb, sbm)
    segs, n_send, n_recv, offsets = plan.segments()      # one contiguous segment per neighbour and direction
    eng.halo_set_layout(*offsets)
    send = torch.zeros(n_send, device="cuda"); recv = torch.zeros(n_recv, device="cuda")
    ext = torch.cuda.ExternalStream(eng.stream(), device=torch.device("cuda", 0))
    def exchange():
        eng.halo_pack(send.data_ptr())
        with torch.cuda.stream(ext):
            ops = []
            for s in segs:      # both "neighbours" are this rank: same sizes, no xGMI hop
                (so, sn), (ro, rn) = s["send"][0], s["recv"][0]
                ops += [dist.P2POp(dist.isend, send[so:so + sn], 0), dist.P2POp(dist.irecv, recv[ro:ro + rn], 0)]
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        eng.halo_unpack(recv.data_ptr())
    for _ in range(3):
        exchange()
    eng.sync(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    with torch.cuda.stream(ext):
        e0.record()
    t0 = time.perf_counter()
    for _ in range(reps):
        exchange()
    host = (time.perf_counter() - t0) / reps
    with torch.cuda.stream(ext):
        e1.record()
    eng.sync(); torch.cuda.synchronize()
    dev = e0.elapsed_time(e1) / reps
    # and the substep cost of this rank's scene (with its ghost columns)
    ms = eng.step_timed(64) / 64
    # the real loop: `depth` substeps, one exchange, repeated (host enqueue overlaps device work)
    periods = 12
    eng.sync(); torch.cuda.synchronize()
    with torch.cuda.stream(ext):
        e0.record()
    for _ in range(periods):
        eng.step(depth)
        exchange()
    with torch.cuda.stream(ext):
        e1.record()
    eng.sync(); torch.cuda.synchronize()
    loop_us = e0.elapsed_time(e1) * 1e3 / (periods * depth)
    print("depth %2d: real loop %.2f us/substep (pure stepping %.2f) = %.1f%% over a ghost-free 1M slab at 20.0 us"
          % (depth, loop_us, ms * 1e3, 100 * (loop_us / 20.0 - 1)), flush=True)
    print("depth %2d: payload %.2f MB out, exchange %.1f us device / %.1f us host enqueue; substep %.2f us -> overhead per substep %.2f us (%.1f%%)"
          % (depth, n_send * 4 / 1e6, dev * 1e3, host * 1e6, ms * 1e3, dev * 1e3 / depth, 100 * dev / depth / ms), flush=True)
    eng.destroy()
dist.destroy_process_group()
