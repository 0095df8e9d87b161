"""GPU halo path: several engines on ONE GPU play the ranks (a gpurun box has a single MI355X);
pack/unpack are the real HIP kernels, the wire is a device-to-device copy.  The merged result must
equal the single-engine run bit for bit.  (Real multi-process RCCL runs are the driver's.)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,depth,path", [(2, 4, 2), (3, 8, 2), (2, 2, 1)])
def test_simulated_gpu_ranks_equal_single_engine(sb, world, depth, path):
    import torch
    from halo_oracle import LocalBus, step_all
    halo = sb.halo
    W, H, steps = 40, 48, 100
    kw = dict(d=30.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), strain_limit=0.5)
    bounds = 8000.0

    def engine_for(buf):
        e = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                      collision_mode=0, path=path, tile_particles=256)
        e.write_buffers(buf)
        return e

    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref = engine_for(gbuf)
    ref.step(steps)
    want = ref.load_buffers(gbuf.copy())
    ref.destroy()

    dev = torch.device("cuda", 0)
    bus = LocalBus()
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = engine_for(buf)
        tr = bus.transport(r, lambda a, b: (torch.zeros(max(a, 1), device=dev), torch.zeros(max(b, 1), device=dev)),
                           lambda t: t.data_ptr())
        exs.append(halo.Exchanger(eng, plan, tr))
        made.append((buf, plan, eng))

    def sync():
        for _, _, e in made:
            e.sync()
        torch.cuda.synchronize()

    step_all(exs, bus, steps, lambda dst, src: dst.copy_(src), sync)
    parts = np.zeros_like(want.particles)
    beams = {}
    for buf, plan, eng in made:
        out = eng.load_buffers(buf.copy())
        gid, prt, bkey, brec = halo.gather_owned(plan, out)
        parts[gid] = prt
        for k, rec in zip(bkey, brec):
            beams[int(k)] = rec.tobytes()[8:]
        eng.destroy()
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)] == rec.tobytes()[8:]
    assert (want.particles[:, 1] == 10.0).any()
