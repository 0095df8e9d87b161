"""Analytic known-answer tests pinning the CPU oracle to the WGSL text.

The reference ships no tests or golden vectors ("parity unpinned"),
so each case here derives the expected value by hand from one cited line of
/root/reference/src/shaders/compute.wgsl, in numpy float32 arithmetic.
"""
import numpy as np
import pytest

f32 = np.float32


def mk(sb, particles, beams=(), layout=1, **consts):
    """Tiny scene: particles = rows [px,py,vx,vy,ax,ay]; beams = tuples
    (a, b, length, spring, damp, yield, limit[, target, last])."""
    buf = sb.Buffers(layout, 16, 16)
    bb = np.zeros(len(beams), dtype=sb.layout.BEAM_DTYPE[layout])
    for i, t in enumerate(beams):
        a, b, length, spring, damp, ys, lim = t[:7]
        bb[i]["a"], bb[i]["b"] = a, b
        bb[i]["length"] = length
        bb[i]["target_length"] = t[7] if len(t) > 7 else length
        bb[i]["last_length"] = t[8] if len(t) > 8 else length
        bb[i]["spring"], bb[i]["damp"] = spring, damp
        bb[i]["yield_strain"], bb[i]["strain_break_limit"] = ys, lim
    buf.set_scene(np.array(particles, dtype="<f4").reshape(-1, 6), bb)
    c = dict(sb.layout.DEFAULT_CONSTANTS)
    c.update(consts)
    buf.set_physics_constants(**c)
    return buf


def run(oracle, buf, n=1, subticks=64, radius=10.0, bounds=1000.0, mode=None, frame=False):
    eng = oracle.OracleEngine(bounds, radius, subticks, buf.layout,
                              oracle.COLLIDE_ALLPAIRS if mode is None else mode)
    eng.write_buffers(buf)
    if frame:
        eng.frame()
    else:
        eng.step(n)
    out = eng.load_buffers(buf.copy())
    return out, eng


NOFORCE = dict(gravity=(0.0, 0.0), drag_coeff=0.0)


@pytest.mark.parametrize("layout", [1, 2])
def test_free_fall(sb, oracle, layout):
    """compute.wgsl:172,186-187: v_n = sum of g*dt, p_n = sum of v_k*dt (drag off)."""
    buf = mk(sb, [[500, 500, 0, 0, 0, 0]], layout=layout, drag_coeff=0.0)
    out, _ = run(oracle, buf, n=10)
    dt = f32(1) / f32(64)
    v = f32(0)
    p = f32(500)
    for _ in range(10):
        a = f32(0) + f32(-0.5)
        v = f32(v + f32(a * dt))
        p = f32(p + f32(v * dt))
    assert out.particles[0, 3] == v and out.particles[0, 1] == p
    assert out.particles[0, 0] == f32(500) and out.particles[0, 2] == 0
    assert v == f32(-0.5 * 10 / 64)


def test_acceleration_persists_one_substep(sb, oracle):
    """compute.wgsl:139,172,186-188: the stored a is added to this substep's a, then zeroed."""
    buf = mk(sb, [[500, 500, 0, 0, 3.0, -2.0]], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    dt = f32(1) / f32(64)
    assert out.particles[0, 2] == f32(f32(3.0) * dt) and out.particles[0, 3] == f32(f32(-2.0) * dt)
    assert tuple(out.particles[0, 4:6]) == (0.0, 0.0)


def test_spring_at_rest_is_zero_force(sb, oracle):
    """compute.wgsl:110: target == last == len -> force_mag = 0; nothing moves."""
    buf = mk(sb, [[100, 100, 0, 0, 0, 0], [200, 100, 0, 0, 0, 0]], [(0, 1, 100, 50, 700, 0.2, 0.5)], **NOFORCE)
    out, eng = run(oracle, buf, n=4)
    assert np.array_equal(out.particles[:2], buf.particles[:2])
    b = out.beams[0]
    assert b["stress"] == 0 and b["strain"] == 0 and b["last_length"] == f32(100)
    assert not eng.forces.any()


def test_stretched_beam_force_and_fixed_point(sb, oracle):
    """compute.wgsl:103-112,122-130,184-187.  len=110, target=100, last=105:
    force_mag = (100-110)*2 + (105-110)*3 = -35; B pulled toward A.  Damping multiplies the
    length change PER SUBSTEP (no 1/dt).  Force lands as i32(f*65536)/65536."""
    buf = mk(sb, [[100, 100, 0, 0, 0, 0], [210, 100, 0, 0, 0, 0]],
             [(0, 1, 100, 2, 3, 0.5, 10, 100, 105)], **NOFORCE)
    eng = oracle.OracleEngine(1000.0, 10.0, 64, 1, oracle.COLLIDE_ALLPAIRS)
    eng.write_buffers(buf)
    # beam phase only is not separable in S0, so check the consumed result
    eng.step(1)
    out = eng.load_buffers(buf.copy())
    fm = f32(f32(f32(100) - f32(110)) * f32(2)) + f32(f32(f32(105) - f32(110)) * f32(3))
    assert fm == f32(-35)
    dt = f32(1) / f32(64)
    nx = f32(f32(110) * f32(f32(1) / f32(110)))        # normalize(v) = v * (1 / length(v)): 1 ulp below 1
    fx = f32(fm * nx)                                   # force on B (toward A: negative x)
    to_b = f32(np.int32(np.trunc(np.float64(f32(fx * f32(65536)))))) / f32(65536)
    to_a = f32(np.int32(np.trunc(np.float64(f32(-fx * f32(65536)))))) / f32(65536)
    assert to_a == -to_b and abs(float(to_b) + 35.0) < 1e-4
    assert out.particles[0, 2] == f32(to_a * dt) and out.particles[1, 2] == f32(to_b * dt)
    b = out.beams[0]
    assert b["last_length"] == f32(110)
    assert b["stress"] == f32(fm * f32(f32(1) / f32(20)))          # :71,122
    assert b["strain"] == f32(f32(abs(f32(f32(10) * f32(f32(1) / f32(100))))) / f32(0.5))  # :112 (x * (1/length)), :123
    assert b["target_length"] == f32(100)  # |strain|=0.1 <= yield 0.5


def test_fixed_point_truncates_toward_zero(sb, oracle):
    """compute.wgsl:127-130: i32() truncates, so +f and -f contributions are exact negatives and a
    force below 2^-16 vanishes."""
    assert oracle.lib().sbo_f32_to_i32(f32(1.9999)) == 1
    assert oracle.lib().sbo_f32_to_i32(f32(-1.9999)) == -1
    assert oracle.lib().sbo_f32_to_i32(f32(3e9)) == 2**31 - 1
    assert oracle.lib().sbo_f32_to_i32(f32(-3e9)) == -2**31
    assert oracle.lib().sbo_f32_to_i32(f32(np.nan)) == 0
    # spring 1e-7 on a stretch of 10 -> force 1e-6 < 2^-16: no motion at all
    buf = mk(sb, [[100, 100, 0, 0, 0, 0], [210, 100, 0, 0, 0, 0]], [(0, 1, 100, 1e-7, 0, 5, 10, 100, 110)], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    assert np.array_equal(out.particles[:2, 2:4], np.zeros((2, 2), "f4"))


def test_zero_length_guard(sb, oracle):
    """compute.wgsl:104-107: coincident endpoints -> diff=(0,-1e-10), len=1e-10 (collisions off so
    only the beam acts)."""
    buf = mk(sb, [[100, 100, 0, 0, 0, 0], [100, 100, 0, 0, 0, 0]], [(0, 1, 100, 1, 0, 5, 10)], **NOFORCE)
    out, _ = run(oracle, buf, n=1, mode=0)
    ln = np.sqrt(f32(f32(0) * f32(0)) + f32(f32(-1e-10) * f32(-1e-10)), dtype=f32)
    fm = f32(f32(f32(100) - ln) * f32(1)) + f32(f32(f32(100) - ln) * f32(0))
    ny = f32(f32(-1e-10) * f32(f32(1) / ln))   # normalize(v) = v * (1 / length(v))
    fy = f32(fm * ny)
    dt = f32(1) / f32(64)
    # A gets -force (so +y), B gets +force (-y)
    ay = f32(np.int32(np.trunc(np.float64(f32(-fy * f32(65536)))))) / f32(65536)
    assert out.particles[0, 3] == f32(ay * dt) and out.particles[1, 3] == f32(-ay * dt)
    assert out.beams[0]["last_length"] == ln


def test_yield_updates_target(sb, oracle):
    """compute.wgsl:113-116: |strain| > yield -> target = len - yield*length*sign(strain);
    :123 strain_out uses the PRE-yield strain."""
    buf = mk(sb, [[100, 100, 0, 0, 0, 0], [250, 100, 0, 0, 0, 0]], [(0, 1, 100, 0, 0, 0.2, 10)], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    strain = f32(f32(f32(150) - f32(100)) * f32(f32(1) / f32(100)))   # :112 as x * (1/length)
    assert out.beams[0]["target_length"] == f32(f32(150) - f32(f32(f32(0.2) * f32(100)) * f32(1)))
    assert out.beams[0]["strain"] == f32(strain / f32(0.2))
    # compression side
    buf = mk(sb, [[100, 100, 0, 0, 0, 0], [150, 100, 0, 0, 0, 0]], [(0, 1, 100, 0, 0, 0.2, 10)], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    assert out.beams[0]["target_length"] == f32(f32(50) - f32(f32(f32(0.2) * f32(100)) * f32(-1)))


def test_break_threshold_strict(sb, oracle):
    """compute.wgsl:117-121 threshold is strict '>' on |len-length| vs length*limit; the beam keeps
    acting until the per-frame delete (engineWorker.ts:663-664) compacts it out."""
    P = [[100, 100, 0, 0, 0, 0], [250, 100, 0, 0, 0, 0], [100, 300, 0, 0, 0, 0], [251, 300, 0, 0, 0, 0]]
    beams = [(0, 1, 100, 0, 0, 5, 0.5), (2, 3, 100, 0, 0, 5, 0.5), (0, 2, 200, 0, 0, 5, 0.5)]
    buf = mk(sb, P, beams, **NOFORCE)
    eng = oracle.OracleEngine(1000.0, 10.0, 64, 1, oracle.COLLIDE_ALLPAIRS)
    eng.write_buffers(buf)
    eng.step(2)
    bit0, bit1 = buf.max_particles + 0, buf.max_particles + 1
    assert not (eng.delete[bit0 // 32] >> (bit0 % 32)) & 1   # |150-100| = 50 > 50 false
    assert (eng.delete[bit1 // 32] >> (bit1 % 32)) & 1       # |151-100| = 51 > 50 true
    assert eng.metadata[6] == 3                              # still counted until the frame ends
    eng.delete_pass()
    assert eng.metadata[6] == 2
    mp = eng.mapping[buf.max_particles:buf.max_particles + 2]
    assert list(mp) == [0, 2]                                # stable compaction
    assert not eng.delete.any()


def test_head_on_collision(sb, oracle):
    """compute.wgsl:155-168 for two particles 18 apart (2r=20) approaching at +-1, e=0.5, mu=0.1."""
    buf = mk(sb, [[400, 300, 1, 0, 0, 0], [418, 300, -1, 0, 0, 0]], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    dt = f32(1) / f32(64)
    ec = f32(f32(f32(0.5) + f32(1)) / f32(2))
    # particle 0: normal (1,0), u = v0 - v1 = (2,0)
    jn = f32(ec * f32(2))
    jt = min(max(f32(0), -f32(jn * f32(0.1))), f32(jn * f32(0.1)))
    v0 = f32(f32(1) - f32(f32(jn * f32(1)) + f32(jt * f32(-0.0))))
    overlap = f32(f32(f32(10) * f32(2)) - f32(18))
    a0 = f32(f32(0) - f32(f32(f32(f32(1) * overlap) / f32(2)) / f32(dt * dt)))
    v0 = f32(v0 + f32(a0 * dt))
    p0 = f32(f32(400) + f32(v0 * dt))
    assert out.particles[0, 2] == v0 and out.particles[0, 0] == p0
    # symmetric partner
    assert out.particles[1, 2] == -v0 and out.particles[1, 0] == f32(f32(418) + f32(-v0 * dt))
    # the a-term is exactly a half-overlap position shift plus an overlap/(2dt) velocity kick
    assert f32(a0 * dt) == f32(-64)


def test_coincident_particles_shift(sb, oracle):
    """compute.wgsl:151-154: dist==0 -> p.y += sign(f32(index) - f32(other_index))."""
    buf = mk(sb, [[400, 300, 0, 0, 0, 0], [400, 300, 0, 0, 0, 0]], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    assert out.particles[0, 1] == f32(299) and out.particles[1, 1] == f32(301)


def test_separating_overlap_clamp_low_gt_high(sb, oracle):
    """compute.wgsl:159-161 with jn<0: clamp(x, +|m|, -|m|) = min(max(x,lo),hi) = hi."""
    buf = mk(sb, [[400, 300, -1, 0.5, 0, 0], [418, 300, 1, 0, 0, 0]], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    ec = f32(0.75)
    jn = f32(ec * f32(-2))
    mf = f32(jn * f32(0.1))          # negative
    ut = f32(f32(f32(-2) * f32(-0.0)) + f32(f32(0.5) * f32(1)))
    jt = min(max(ut, -mf), mf)       # = mf
    assert jt == mf
    dt = f32(1) / f32(64)
    vy = f32(f32(0.5) - f32(f32(jn * f32(0)) + f32(jt * f32(1))))
    assert out.particles[0, 3] == vy  # a.y = 0


def test_wall_hit_and_one_sided_friction(sb, oracle):
    """compute.wgsl:190-199: clamp to [r, S-r]; v.x *= -e_b; a.y -= min(a.y(=0), sign(v.y)*mu_b*|v.x|*(1+e_b))
    which only bites when the product is negative (v.y < 0)."""
    # moving left into the x=r wall with downward v.y
    buf = mk(sb, [[10.5, 500, -64, -1, 0, 0]], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    dt = f32(1) / f32(64)
    px = f32(f32(10.5) + f32(f32(-64) * dt))
    assert px < 10
    x = f32(f32(f32(f32(-1) * f32(0.2)) * f32(64)) * f32(f32(1) + f32(0.5)))
    assert out.particles[0, 0] == f32(10)
    assert out.particles[0, 2] == f32(f32(-64) * f32(-0.5))
    assert out.particles[0, 5] == f32(f32(0) - min(f32(0), x)) and out.particles[0, 5] > 0
    # upward v.y: min(0, positive) = 0 -> no friction
    buf = mk(sb, [[10.5, 500, -64, 1, 0, 0]], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    assert out.particles[0, 5] == 0
    # y wall, top
    buf = mk(sb, [[500, 989.5, -2, 64, 0, 0]], **NOFORCE)
    out, _ = run(oracle, buf, n=1)
    assert out.particles[0, 1] == f32(990) and out.particles[0, 3] == f32(-32)
    assert out.particles[0, 4] == f32(f32(0) - min(f32(0), f32(f32(f32(f32(-1) * f32(0.2)) * f32(64)) * f32(1.5))))


def test_drag(sb, oracle):
    """compute.wgsl:174-176: a -= c * pow(|v|, k) * normalize(v), componentwise."""
    buf = mk(sb, [[500, 500, 3, -4, 0, 0]], gravity=(0.0, 0.0), drag_coeff=0.001, drag_exp=2.0)
    out, _ = run(oracle, buf, n=1)
    dt = f32(1) / f32(64)
    L = np.sqrt(f32(f32(9) + f32(16)), dtype=f32)
    inv = f32(f32(1) / L)   # normalize(v) = v * (1 / length(v))
    ax = f32(f32(0) - f32(f32(f32(0.001) * f32(9)) * f32(f32(3) * inv)))
    ay = f32(f32(0) - f32(f32(f32(0.001) * f32(16)) * f32(f32(-4) * inv)))
    assert out.particles[0, 2] == f32(f32(3) + f32(ax * dt))
    assert out.particles[0, 3] == f32(f32(-4) + f32(ay * dt))


def test_pow_general_exponent(oracle):
    L = oracle.lib()
    for x in [0.0, 1e-3, 0.5, 1.0, 2.0, 7.25, 123.456, 1e4]:
        for y in [1.0, 1.5, 2.0, 2.5, 3.0, 3.7, 4.0]:
            got = L.sbo_pow(f32(x), f32(y))
            exp = np.float64(f32(x)) ** np.float64(f32(y))
            assert got == pytest.approx(exp, rel=2e-7, abs=0), (x, y)
    assert L.sbo_pow(f32(3), f32(2)) == 9 and L.sbo_pow(f32(0), f32(1.5)) == 0


def test_user_force_and_mouse_grab(sb, oracle):
    """compute.wgsl:178-181."""
    buf = mk(sb, [[500, 500, 1, 2, 0, 0], [800, 800, 0, 0, 0, 0]], drag_coeff=0.0)
    buf.user_strength = 2.0
    buf.set_user_input(applied_force=(0.25, -0.5), mouse_pos=(530, 540), mouse_vel=(5, 6), mouse_active=True)
    out, _ = run(oracle, buf, n=1)
    dt = f32(1) / f32(64)
    g = (f32(0), f32(-0.5))
    # particle 0 is 50 < 100 from the mouse; particle 1 is not
    ax = f32(f32(g[0] + f32(f32(0.25) * f32(2))) + f32(f32(f32(f32(5) - f32(1)) * f32(2)) - g[0]))
    ay = f32(f32(g[1] + f32(f32(-0.5) * f32(2))) + f32(f32(f32(f32(6) - f32(2)) * f32(2)) - g[1]))
    assert out.particles[0, 2] == f32(f32(1) + f32(ax * dt)) and out.particles[0, 3] == f32(f32(2) + f32(ay * dt))
    ay1 = f32(g[1] + f32(f32(-0.5) * f32(2)))
    assert out.particles[1, 2] == f32(f32(f32(0.25) * f32(2)) * dt) and out.particles[1, 3] == f32(ay1 * dt)


def test_ping_pong_parity(sb, oracle):
    """engineWorker.ts:90,655-661: even substep counts end in A."""
    buf = mk(sb, [[500, 500, 0, 0, 0, 0]])
    eng = oracle.OracleEngine(1000.0, 10.0, 3, 1)
    assert eng.subticks == 4
    eng.write_buffers(buf)
    eng.frame()
    assert eng.final_in_b == 0


def test_grid_equals_allpairs(sb, oracle):
    """The grid broad phase must reproduce compute.wgsl:144-170 bit for bit (same pair set, same
    ascending-slot summation order)."""
    rng = np.random.default_rng(7)
    P = 600
    pts = np.zeros((P, 6), "f4")
    pts[:, :2] = rng.uniform(10, 390, (P, 2))
    pts[:, 2:4] = rng.uniform(-3, 3, (P, 2))
    pts[5, :2] = pts[9, :2]  # one coincident pair
    buf = sb.Buffers(2, P, 4)
    buf.set_scene(pts, np.zeros(0, sb.layout.BEAM_DTYPE[2]))
    res = []
    for mode in (oracle.COLLIDE_ALLPAIRS, oracle.COLLIDE_GRID):
        eng = oracle.OracleEngine(400.0, 10.0, 64, 2, mode)
        eng.write_buffers(buf)
        eng.step(16)
        res.append(eng.load_buffers(buf.copy()).particles.copy())
    assert np.array_equal(res[0].view("u4"), res[1].view("u4"))
    assert not np.array_equal(res[0], pts)


def test_threads_do_not_change_result(sb, oracle):
    buf = sb.scenes.default_buffers(1)
    out = []
    for th in (1, 4):
        eng = oracle.OracleEngine(1000.0, 10.0, 64, 1, oracle.COLLIDE_ALLPAIRS, threads=th)
        eng.write_buffers(buf)
        eng.frame()
        o = eng.load_buffers(buf.copy())
        out.append((o.particles.tobytes(), o.beams.tobytes()))
    assert out[0] == out[1]
