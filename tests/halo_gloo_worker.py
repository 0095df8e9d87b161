"""Worker for tests/test_halo_cpu.py::test_gloo_world_size_2 (launched by torch.distributed.run)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sb = ge.load_package()
    oracle = ge.load_oracle()
    from halo_oracle import CpuTorchTransport, OracleRank
    halo = sb.halo
    W, H, depth, steps = 8, 9, 4, 42
    frames = int(os.environ.get("HALO_FRAMES", "0"))     # > 0: beams break, whole frames with delete passes (Exchanger.frame)
    kw = dict(d=25.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), strain_limit=0.02 if frames else 0.5)
    buf, plan = halo.slab_scene(sb, rank, world, W, H, depth=depth, **kw)
    eng = OracleRank(oracle, buf, 1000.0)
    ex = halo.Exchanger(eng, plan, CpuTorchTransport(torch, dist))
    if frames:
        return frames_with_breaks(sb, oracle, halo, ex, eng, buf, plan, rank, world, W, H, depth, kw, frames)
    ex.step(steps)
    out = eng.load(buf)
    # 42 = 10 refresh periods + 2 substeps: ghosts are stale by two substeps now, so the cross-check must see it;
    # two more substeps complete a period and it must come back clean
    stale = halo.ghost_mismatches(plan, out, dist, torch, torch.device("cpu"))
    ex.step(2)
    fresh = halo.ghost_mismatches(plan, eng.load(buf), dist, torch, torch.device("cpu"))
    assert stale > 0 and fresh == 0, (stale, fresh)
    gid, prt, bkey, brec = halo.gather_owned(plan, out)
    gathered = [None] * world
    dist.all_gather_object(gathered, (gid, prt, bkey, brec.tobytes()))
    if rank == 0:
        gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
        ref = OracleRank(oracle, gbuf, 1000.0)
        ref.step(steps)
        want = ref.load(gbuf)
        parts = np.zeros_like(want.particles)
        for g, p, _, _ in gathered:
            parts[g] = p
        assert np.array_equal(parts.view("u4"), want.particles.view("u4")), "particles differ"
        wantb = {int(k): r.tobytes()[8:] for k, r in zip(gplan.global_beam_key, want.beams)}
        n = 0
        for _, _, bk, bb in gathered:
            recs = np.frombuffer(bb, dtype=want.beams.dtype)
            for k, r in zip(bk, recs):
                assert wantb[int(k)] == r.tobytes()[8:], "beam differs"
                n += 1
        assert n == len(wantb)
        print("HALO_GLOO_OK ranks=%d particles=%d beams=%d" % (world, parts.shape[0], n), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def live_keys(plan, buf):
    live = buf.mapping[buf.max_particles:buf.max_particles + buf.beam_count].astype(np.int64)
    own = np.zeros(buf.max_beams, bool)
    own[plan.owned_beams] = True
    return [int(k) for k in plan.global_beam_key[live[own[live]]]]


def frames_with_breaks(sb, oracle, halo, ex, eng, buf, plan, rank, world, W, H, depth, kw, frames):
    from halo_oracle import OracleRank
    for _ in range(frames):
        ex.frame()
    ex.verify()
    out = eng.load(buf)
    gid, prt, bkey, brec = halo.gather_owned(plan, out)
    gathered = [None] * world
    dist.all_gather_object(gathered, (gid, prt, bkey, brec.tobytes(), live_keys(plan, out)))
    if rank == 0:
        gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
        ref = OracleRank(oracle, gbuf, 1000.0)
        for _ in range(frames):
            ref.ref.frame()
        want = ref.load(gbuf)
        assert want.beam_count < gbuf.beam_count - 10, "the scene is meant to break beams"
        parts = np.zeros_like(want.particles)
        live = set()
        for g, p, _, _, lk in gathered:
            parts[g] = p
            live |= set(lk)
        assert np.array_equal(parts.view("u4"), want.particles.view("u4")), "particles differ"
        wantb = {int(k): r.tobytes()[8:] for k, r in zip(gplan.global_beam_key, want.beams)}
        for _, _, bk, bb, _ in gathered:
            for k, r in zip(bk, np.frombuffer(bb, dtype=want.beams.dtype)):
                assert wantb[int(k)] == r.tobytes()[8:], "beam differs"
        gl = want.mapping[want.max_particles:want.max_particles + want.beam_count].astype(np.int64)
        assert live == set(int(k) for k in gplan.global_beam_key[gl]), "different beams removed"
        print("HALO_GLOO_OK ranks=%d frames=%d beams left %d of %d" % (world, frames, want.beam_count, gbuf.beam_count), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
