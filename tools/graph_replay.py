#!/usr/bin/env python3
"""Does a hipGraph shorten the substep loop?  (no.)  Diagnostic, outside the product: the engine's own
launches (sb_step, through the C ABI) are captured on the engine's stream into ONE graph and the replay is timed with
events on that stream, next to sb_step_timed on the same scene.  Until round 2 this lived inside sb_step_timed behind an
environment variable; the timed entry point is now events around the launches and nothing else.

    python tools/graph_replay.py [--width 1000 --height 1000 --steps 960 --collisions off|grid]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--height", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=960)
    ap.add_argument("--collisions", choices=["off", "grid"], default="off")
    a = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    sb = ge.load_package()
    buf = sb.scenes.lattice_buffers(a.width, a.height, d=30.0, origin=(1000.0, 1000.0), jitter=1.0, layout=2)
    bounds = float(max(a.width, a.height) * 30.0 + 2000.0)
    eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=64, layout=2, max_particles=buf.max_particles,
                    max_beams=buf.max_beams, collision_mode={"off": 0, "grid": 2}[a.collisions])
    eng.write_buffers(buf)
    eng.step(64)
    eng.sync()
    plain_ms = eng.step_timed(a.steps)
    stream = torch.cuda.ExternalStream(eng.stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
        eng.step(a.steps)              # the same launches, recorded instead of run
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        g.replay()                     # (first replay: uploads the graph)
        e0.record()
        g.replay()
        e1.record()
    e1.synchronize()
    graph_ms = e0.elapsed_time(e1)
    print("plain launches: %.2f us per substep; one graph of the same %d substeps: %.2f us per substep"
          % (plain_ms * 1e3 / a.steps, a.steps, graph_ms * 1e3 / a.steps))
    eng.destroy()


if __name__ == "__main__":
    main()
