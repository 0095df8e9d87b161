"""How long does one ghost-zone refresh take on the device, and what does the real loop cost?

One rank (an interior slab: two neighbours) plays both of its own neighbours, so there is no xGMI hop:
a lower bound for the multi-GPU case, used to pick bench.py's --ghost-depth.  Two transports:
  rccl  pack kernel -> batch_isend_irecv to self on the engine stream -> unpack kernel
  peer  sb_peer_exchange wired to the rank's own mailbox (pack into mailbox, flag handshake, unpack)
"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29612")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import __graft_entry__ as ge
sb = ge.load_package()
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
W, H = 1000, 1000
base = sb.scenes.lattice_buffers(W, H, d=30.0, origin=(1000.0, 1000.0), jitter=1.0, layout=2)
eng = sb.Engine(bounds_size=100000.0, layout=2, max_particles=base.max_particles, max_beams=base.max_beams, collision_mode=0)
eng.write_buffers(base)
eng.step(64)
BASE_US = eng.step_timed(500) * 1e3 / 500   # the same slab without ghost columns, us per substep
eng.destroy()
del base
for depth in (12, 18, 24, 30):
    for transport in ("rccl", "peer"):
        buf, plan = sb.halo.slab_scene(sb, 1, 3, W, H, jitter=1.0, depth=depth)
        eng = sb.Engine(bounds_size=100000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=0)
        eng.write_buffers(buf)
        eng.halo_configure(*plan.lists())
        segs, n_send, n_recv, offsets = plan.segments()      # one contiguous segment per neighbour and direction
        eng.halo_set_layout(*offsets)
        ext = torch.cuda.ExternalStream(eng.stream(), device=torch.device("cuda", 0))
        if transport == "rccl":
            send = torch.zeros(n_send, device="cuda"); recv = torch.zeros(n_recv, device="cuda")
            def exchange():
                eng.halo_pack(send.data_ptr())
                with torch.cuda.stream(ext):
                    ops = []
                    for s in segs:
                        (so, sn), (ro, rn) = s["send"][0], s["recv"][0]
                        ops += [dist.P2POp(dist.isend, send[so:so + sn], 0), dist.P2POp(dist.irecv, recv[ro:ro + rn], 0)]
                    for r in dist.batch_isend_irecv(ops):
                        r.wait()
                eng.halo_unpack(recv.data_ptr())
        else:
            box, _, _ = eng.peer_mailbox()
            eng.peer_connect([box, box], [n_recv, n_recv], [s["send"][0][0] for s in segs], [s["send"][0][1] for s in segs],
                             [s["recv"][0][0] for s in segs], [0, 1], 2000)
            exchange = eng.peer_exchange
        for _ in range(3):
            exchange()
        eng.sync(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        with torch.cuda.stream(ext):
            e0.record()
        for _ in range(reps):
            exchange()
        with torch.cuda.stream(ext):
            e1.record()
        eng.sync(); torch.cuda.synchronize()
        alone_us = e0.elapsed_time(e1) * 1e3 / reps
        pure_us = eng.step_timed(64) * 1e3 / 64     # this rank's scene with its ghost columns, no exchange
        periods = 12                                # the real loop: `depth` substeps, one exchange, repeated
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
        print("depth %2d %s: payload %.2f MB, exchange alone %.1f us; loop %.2f us/substep (stepping alone %.2f) = %+.1f%% over a ghost-free slab at %.2f us"
              % (depth, transport, n_send * 4 / 1e6, alone_us, loop_us, pure_us, 100 * (loop_us / BASE_US - 1), BASE_US), flush=True)
        eng.destroy()
dist.destroy_process_group()
