"""soak of the cross-rank delete protocol on the real kernels: random world / depth / path / limits / velocities"""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as ge
import torch
sb = ge.load_package()
from halo_oracle import LocalBus, frame_all
halo = sb.halo
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
for case in range(40):
    world = int(rng.integers(2, 4)); depth = int(rng.integers(2, 9)); path, block = [(2, 0), (2, 1), (1, 0), (2, 3)][int(rng.integers(0, 4))]
    W, H, frames = int(rng.integers(max(depth, 10), 40)), int(rng.integers(10, 40)), int(rng.integers(1, 4))
    kw = dict(d=30.0, origin=(100.0, 11.5), jitter=float(rng.uniform(0.2, 2.0)), velocity=(float(rng.uniform(-8, 8)), float(rng.uniform(-9, -1))),
              strain_limit=float(rng.choice([0.01, 0.02, 0.04, 0.08])), yield_strain=float(rng.choice([0.2, 0.2, 0.004, 0.01])))
    def engine_for(buf):
        e = sb.Engine(bounds_size=8000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=0, path=path,
                      tile_particles=int(rng.choice([64, 256, 0])), block_substeps=block)
        e.write_buffers(buf); return e
    def live_keys(plan, out, owned_only=True):
        live = out.mapping[out.max_particles:out.max_particles + out.beam_count].astype(np.int64)
        if owned_only:
            own = np.zeros(out.max_beams, bool); own[plan.owned_beams] = True; live = live[own[live]]
        return set(int(k) for k in plan.global_beam_key[live])
    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref = engine_for(gbuf)
    for _ in range(frames): ref.frame()
    want = ref.load_buffers(gbuf.copy()); ref.destroy()
    if not np.isfinite(want.particles[:want.particle_count]).all():
        print("case", case, "skipped (non-finite)"); continue
    dev = torch.device("cuda", 0); bus = LocalBus(); exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = engine_for(buf)
        tr = bus.transport(r, lambda a, b: (torch.zeros(max(a, 1), device=dev), torch.zeros(max(b, 1), device=dev)), lambda t: t.data_ptr())
        exs.append(halo.Exchanger(eng, plan, tr)); made.append((buf, plan, eng))
    def sync():
        for _, _, e in made: e.sync()
        torch.cuda.synchronize()
    for _ in range(frames): frame_all(exs, bus, lambda dst, src: dst.copy_(src), sync)
    parts = np.zeros_like(want.particles); live = set(); ok = True
    for buf, plan, eng in made:
        out = eng.load_buffers(buf.copy())
        gid, prt, bkey, brec = halo.gather_owned(plan, out)
        parts[gid] = prt; live |= live_keys(plan, out)
        ok &= live_keys(plan, out, owned_only=False) == set(int(k) for k in plan.global_beam_key) & live_keys(gplan, want)
        eng.destroy()
    ok &= np.array_equal(parts.view("u4"), want.particles.view("u4")) and live == live_keys(gplan, want)
    print("case %2d world %d depth %d path %d block %d %dx%d frames %d limit %.2f: broke %4d of %5d  %s" % (case, world, depth, path, block, W, H, frames,
          kw["strain_limit"], gbuf.beam_count - want.beam_count, gbuf.beam_count, "OK" if ok else "MISMATCH"), flush=True)
    bad += not ok
print("mismatches:", bad)
