"""Synthetic scene generators.

The only "fixtures" the reference has are its default scene and the lattice
helper that builds it: `addRectangle` in /root/reference/src/main.ts:203-214
(particle (x,y) at (ox + x*d, oy + y*d), data index x*h + y; beams to +y, +x
and both diagonals).  These functions generalise that helper.
"""
import numpy as np

from .layout import BEAM_DTYPE, LAYOUT_V1, LAYOUT_V2, Buffers


def rectangle(ox, oy, d, w, h, spring, damp, yield_strain, strain_limit, *, base=0,
              anti_diagonal=True, layout=LAYOUT_V2):
    """Vectorised `addRectangle` (main.ts:203-214).  Returns (particles (w*h,2) f32, beams
    structured array) with beams in the reference's emission order: for each particle b in
    index order, [+y, +x, diagonal(+x,+y), anti-diagonal(+x,-y)] where they exist.
    anti_diagonal=False drops the main.ts:211 beam (the "~3 beams/particle" topology of
    BASELINE config 2)."""
    xs, ys = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")  # index = x*h + y
    xs = xs.reshape(-1)
    ys = ys.reshape(-1)
    # float64 math then one rounding to f32, as JS (doubles) -> Float32Array does
    pos = np.stack([xs * float(d) + float(ox), ys * float(d) + float(oy)], axis=1).astype("<f4")
    idx = (xs * h + ys).astype(np.int64) + base
    kinds = [
        (ys < h - 1, 1, float(d)),                                  # main.ts:208
        (xs < w - 1, h, float(d)),                                  # main.ts:209
        ((ys < h - 1) & (xs < w - 1), h + 1, np.sqrt(2.0) * d),     # main.ts:210
    ]
    if anti_diagonal:
        kinds.append(((ys > 0) & (xs < w - 1), h - 1, np.sqrt(2.0) * d))  # main.ts:211
    nk = len(kinds)
    n = w * h
    valid = np.stack([k[0] for k in kinds], axis=1)          # (n, nk), row-major = emission order
    a = np.repeat(idx, nk).reshape(n, nk)
    boff = np.array([k[1] for k in kinds], dtype=np.int64)
    blen = np.array([k[2] for k in kinds], dtype=np.float64)
    b = a + boff[None, :]
    ln = np.broadcast_to(blen[None, :], (n, nk))
    sel = valid.reshape(-1)
    beams = np.zeros(int(sel.sum()), dtype=BEAM_DTYPE[layout])
    beams["a"] = a.reshape(-1)[sel]
    beams["b"] = b.reshape(-1)[sel]
    L = ln.reshape(-1)[sel].astype("<f4")
    beams["length"] = L
    beams["target_length"] = L   # Beam ctor defaults, engineMapping.ts:170-171
    beams["last_length"] = L
    beams["spring"] = spring
    beams["damp"] = damp
    beams["yield_strain"] = yield_strain
    beams["strain_break_limit"] = strain_limit
    return pos, beams


def default_scene(layout=LAYOUT_V1):
    """`oofDefaultState`, main.ts:188-246: 119 particles / 299 beams."""
    parts, beams = [], []
    base = 0

    def add(*args):
        nonlocal base
        p, b = rectangle(*args, base=base, layout=layout)
        parts.append(p)
        beams.append(b)
        base += p.shape[0]

    def free(x, y):
        nonlocal base
        parts.append(np.array([[x, y]], dtype="<f4"))
        base += 1

    add(185, 10, 60, 2, 2, 1, 50, 1, 2.5)       # main.ts:218
    add(35, 10, 60, 2, 2, 1, 50, 1, 2.5)        # :219
    add(20, 120, 30, 9, 4, 50, 700, 0.2, 0.5)   # :220
    free(445, 10)                               # :221
    free(925, 10)                               # :222
    add(400, 40, 30, 20, 2, 500, 800, 0.1, 0.5) # :223
    add(700, 400, 40, 5, 5, 3, 50, 2, 5)        # :224
    add(20, 900, 50, 2, 2, 0.05, 10, 2, 3)      # :240
    add(20, 700, 50, 2, 2, 0.1, 10, 2, 3)       # :241
    return np.concatenate(parts), np.concatenate(beams)


def default_buffers(layout=LAYOUT_V1, max_particles=None, max_beams=None):
    p, b = default_scene(layout)
    buf = Buffers(layout, max_particles or 65536, max_beams or 65536)
    buf.set_scene(p, b)
    return buf


def lattice_buffers(w, h, d=30.0, origin=(1000.0, 1000.0), spring=50.0, damp=700.0,
                    yield_strain=0.2, strain_limit=1.0e9, *, anti_diagonal=False,
                    layout=LAYOUT_V2, jitter=0.0, velocity=None, seed=1, slack=0):
    """One w x h lattice blob (BASELINE configs 2-5 are instances of this)."""
    p, b = rectangle(origin[0], origin[1], d, w, h, spring, damp, yield_strain, strain_limit,
                     anti_diagonal=anti_diagonal, layout=layout)
    pv = np.zeros((p.shape[0], 6), dtype="<f4")
    pv[:, :2] = p
    if jitter:
        pv[:, :2] += hash_uniform(seed, p.shape[0] * 2).reshape(-1, 2).astype("<f4") * np.float32(jitter)
    if velocity is not None:
        pv[:, 2:4] = np.asarray(velocity, dtype="<f4")
    buf = Buffers(layout, p.shape[0] + slack, b.shape[0] + slack)
    buf.set_scene(pv, b)
    return buf


def soup_buffers(w, h, d=40.0, origin=(1000.0, 1000.0), jitter=10.0, speed=60.0, seed=1, layout=LAYOUT_V2):
    """w*h FREE particles (no beams): a grid of spacing d jittered by +-jitter, each thrown in a random
    direction at up to `speed` units/s (hash_uniform draws).  With gravity, the floor and the walls this is a
    scene in which the whole collision path of compute.wgsl:142-170 keeps firing."""
    n = w * h
    xs, ys = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
    u = hash_uniform(seed, 4 * n).reshape(n, 4)
    pv = np.zeros((n, 6), dtype="<f4")
    pv[:, 0] = (xs.reshape(-1) * float(d) + float(origin[0]) + u[:, 0] * jitter).astype("<f4")
    pv[:, 1] = (ys.reshape(-1) * float(d) + float(origin[1]) + u[:, 1] * jitter).astype("<f4")
    pv[:, 2] = (u[:, 2] * speed).astype("<f4")
    pv[:, 3] = (u[:, 3] * speed).astype("<f4")
    buf = Buffers(layout, n, 4)
    buf.set_scene(pv, np.zeros(0, dtype=BEAM_DTYPE[layout]))
    return buf


def hash_uniform(seed, n):
    """Documented integer hash -> uniform [-1, 1) doubles (no Math.random).
    splitmix64 of (seed*2^32 + i)."""
    x = (np.uint64(seed) << np.uint64(32)) + np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 52) - 1.0


def rest_at_current_length(buf):
    """Every beam's rest length := the present distance between its endpoints, the way the reference's editor makes beams
    (a beam drawn between two particles rests at that distance).  On a jittered lattice no two beams then share a rest
    length: the material dictionary falls back to (spring, damp, yield, limit) rows + one length per beam (mode 1).
    The length is computed in binary32 like compute.wgsl:103,108: sqrt(dx*dx + dy*dy), each operation rounded."""
    B = buf.beam_count
    a, b = buf.beams["a"][:B].astype(np.int64), buf.beams["b"][:B].astype(np.int64)
    dx = buf.particles[b, 0] - buf.particles[a, 0]
    dy = buf.particles[b, 1] - buf.particles[a, 1]
    ln = np.sqrt(dx * dx + dy * dy, dtype=np.float32)
    for f in ("length", "target_length", "last_length"):
        buf.beams[f][:B] = ln
    return buf


def mix_stiffness(buf, seed=1, subticks=128):
    """BASELINE config 5: per-beam spring drawn from {1, 3, 50, 500} (the values in main.ts:218-246)
    with damping scaled so that the explicit integrator stays inside its stability envelope at
    dt = 1/subticks (per-node sum of damp*dt^2 <~ 0.5 with ~6 beams per node)."""
    B = buf.beam_count
    pick = ((hash_uniform(seed + 77, B) + 1.0) * 2.0).astype(np.int64).clip(0, 3)
    springs = np.array([1.0, 3.0, 50.0, 500.0], dtype="<f4")[pick]
    damp_cap = np.float32(0.5 / 6.0 * subticks * subticks)       # damp * dt^2 * 6 beams <= 0.5
    damps = np.minimum(springs * np.float32(14.0), damp_cap).astype("<f4")
    buf.beams["spring"][:B] = springs
    buf.beams["damp"][:B] = damps
    return buf


def blob_pile_buffers(columns, layers, bw=9, bh=4, d=30.0, gap=21.0, spring=50.0, damp=700.0,
                      yield_strain=0.2, strain_limit=0.5, bounds=None, layout=LAYOUT_V2, seed=1):
    """BASELINE config 3 as the reference's own scene does it (main.ts:218-224: lattice blobs resting on the
    floor and on each other): `layers` courses of `columns` blobs, each a bw x bh `addRectangle` lattice with
    the material of main.ts:220 (spring 50, damp 700, yield 0.2, limit 0.5), laid like bricks (every other
    course shifted by half a blob) with `gap` between the particle centres of neighbouring blobs (2r = 20 is
    touching), the bottom course on the floor.  Beams, resting contacts and the floor response all act in
    steady state.  Returns (Buffers, bounds_size)."""
    bwid, bhei = (bw - 1) * float(d), (bh - 1) * float(d)
    pitch_x, pitch_y = bwid + gap, bhei + gap
    parts, beams, base = [], [], 0
    x_left = 10.0 + 0.5 * pitch_x
    for L in range(layers):
        shift = 0.0 if L % 2 == 0 else 0.5 * pitch_x
        for c in range(columns):
            p, b = rectangle(x_left + shift + c * pitch_x, 10.0 + L * pitch_y, d, bw, bh, spring, damp,
                             yield_strain, strain_limit, base=base, anti_diagonal=True, layout=layout)
            parts.append(p)
            beams.append(b)
            base += p.shape[0]
    p = np.concatenate(parts)
    pv = np.zeros((p.shape[0], 6), dtype="<f4")
    pv[:, :2] = p
    b = np.concatenate(beams)
    need = float(max(x_left + (columns + 0.5) * pitch_x + 10.0, layers * pitch_y + 1000.0))
    S = float(bounds) if bounds else need
    if S < need:
        raise ValueError("blob pile needs a box of %g, got %g" % (need, S))
    buf = Buffers(layout, pv.shape[0], b.shape[0])
    buf.set_scene(pv, b)
    return buf, S


CONFIG3_SETTLE_FRAMES = 48
CONFIG3_TEXT = ("BASELINE config 3: %d particles / %d beams as a pile of 9x4 lattice blobs (the reference's own blob, "
                "main.ts:220: spacing 30, spring 50, damp 700, yield 0.2, break limit 0.5) laid like bricks, touching, "
                "courses resting on the floor and on each other in a %g box; gravity, floor, walls, spatial-hash "
                "collisions, subticks 64, v2 (u32) layout; settled for 48 frames (3072 substeps) before the warm-up")


def config3_buffers(particles=1_000_000, layout=LAYOUT_V2):
    """The config-3 workload of bench.py: about `particles` particles (never more) as a blob pile some twenty
    times wider than tall (deep beds burst under the collision response of compute.wgsl:164-168; shallow ones
    settle and stay a pile: tools/pile_probe.py).  Returns (Buffers, bounds_size)."""
    blobs = max(1, particles // 36)
    layers = max(1, int(round((blobs / 8.6) ** 0.5)))
    columns = max(1, blobs // layers)
    return blob_pile_buffers(columns, layers, gap=20.0, layout=layout)
