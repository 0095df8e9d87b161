"""Worker for tests/test_gpu_halo.py::test_two_processes_peer_exchange (launched by torch.distributed.run).

Two OS processes, each with its own engine on cuda:0, trade ghost zones through IPC-mapped mailboxes
(sb_peer_*): the cross-process form of the exchange bench.py uses between GPUs.  The process group (gloo)
only carries the mailbox handles and the final gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sb = ge.load_package()
    halo = sb.halo
    W, H, depth, steps = 40, 48, 4, 100
    frames = int(os.environ.get("HALO_FRAMES", "0"))     # > 0: beams break, whole frames with delete passes (PeerExchanger.frame)
    kw = dict(d=30.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), strain_limit=0.02 if frames else 0.5)

    def engine_for(buf):
        e = sb.Engine(bounds_size=8000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                      collision_mode=0, path=2, tile_particles=256)
        e.write_buffers(buf)
        return e

    buf, plan = halo.slab_scene(sb, rank, world, W, H, depth=depth, **kw)
    eng = engine_for(buf)
    ex = halo.PeerExchanger(eng, plan, timeout_ms=20000)
    cards = [None] * world
    dist.all_gather_object(cards, ex.card)
    assert len({c["pid"] for c in cards}) == world
    ex.connect(cards)
    dist.barrier()
    if frames:
        for _ in range(frames):
            ex.frame()
        ex.verify()
    else:
        ex.step(steps)
    eng.sync()
    out = eng.load_buffers(buf.copy())
    gid, prt, bkey, brec = halo.gather_owned(plan, out)
    live = out.mapping[out.max_particles:out.max_particles + out.beam_count].astype(np.int64)
    own = np.zeros(out.max_beams, bool)
    own[plan.owned_beams] = True
    live = [int(k) for k in plan.global_beam_key[live[own[live]]]]
    gathered = [None] * world
    dist.all_gather_object(gathered, (gid, prt, bkey, brec.tobytes(), live))
    dist.barrier()          # nobody unmaps a mailbox a neighbour may still be writing
    eng.destroy()
    if rank == 0:
        gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
        ref = engine_for(gbuf)
        if frames:
            for _ in range(frames):
                ref.frame()
        else:
            ref.step(steps)
        want = ref.load_buffers(gbuf.copy())
        ref.destroy()
        parts = np.zeros_like(want.particles)
        alive = set()
        for g, p, _, _, lk in gathered:
            parts[g] = p
            alive |= set(lk)
        if frames:
            assert want.beam_count < gbuf.beam_count - 10, "the scene is meant to break beams"
            gl = want.mapping[want.max_particles:want.max_particles + want.beam_count].astype(np.int64)
            assert alive == set(int(k) for k in gplan.global_beam_key[gl]), "different beams removed"
        assert np.array_equal(parts.view("u4"), want.particles.view("u4")), "particles differ"
        wantb = {int(k): r.tobytes()[8:] for k, r in zip(gplan.global_beam_key, want.beams)}
        n = 0
        for _, _, bk, bb, _ in gathered:
            recs = np.frombuffer(bb, dtype=want.beams.dtype)
            for k, r in zip(bk, recs):
                assert wantb[int(k)] == r.tobytes()[8:], "beam differs"
                n += 1
        assert n == len(wantb)
        print("HALO_PEER_OK ranks=%d particles=%d beams=%d frames=%d beams left %d" % (world, parts.shape[0], n, frames, want.beam_count), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
